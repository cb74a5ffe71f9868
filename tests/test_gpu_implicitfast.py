"""GPU parity of the implicitfast integrator (include/fmj.h FMJ_INT_IMPLICITFAST; reference mjcf.py:1342-1347 forwards
``simulation_options.integrator`` to MuJoCo): the velocity gains of the unclamped actuators join the joint damping on the diagonal of
the matrix the velocity update is solved with.  The oracle's version is pinned by closed-form recurrences (tests/test_oracle_kat.py);
here the four HIP kernels - two envs per wave or one, without and with constraints - are compared with it."""
import numpy as np
import pytest

from parity_metrics import relerr, group_relerr, qvel_groups

pytestmark = pytest.mark.gpu


def _servo_model(maker, integrator, kv=2e-3, clamp=False, **kw):
    """A zoo model whose velocity actuators have a gain (the zoo's own kv is 0, where implicitfast and Euler coincide)."""
    import farms_mujoco_amd.model as mm
    kw = dict(kw); solver = kw.pop('solver', None)
    m = getattr(mm, maker)(**kw)
    if solver:
        m.solver = mm.SOLVERS[solver]; m.solver_iterations = 100
    for a, tag in enumerate(m.actuator_tags):
        if tag == 'velocity':
            m.actuator_gain[a] = kv; m.actuator_bias[a, 2] = -kv
            if clamp and a % 2:                             # every other servo saturates: no velocity derivative for those
                m.actuator_forcelimited[a] = 1; m.actuator_forcerange[a] = (-1e-4, 1e-4)
    m.integrator = mm.INTEGRATORS[integrator.lower()]
    return m


def _state(m, n, seed, vscale=0.3):
    import farms_mujoco_amd.model as mm
    rng = np.random.default_rng(seed)
    qpos, qvel, _ = mm.synthetic_batch(m, n, seed=seed)
    qvel = qvel + rng.normal(size=qvel.shape)*vscale
    ctrl = np.zeros((n, m.nu))
    for a, tag in enumerate(m.actuator_tags):
        if tag == 'position':
            ctrl[:, a] = qpos[:, m.jnt_qposadr[m.actuator_jntid[a]]]
        if tag == 'velocity':
            ctrl[:, a] = rng.uniform(-2, 2, n)
    return qpos, qvel, ctrl


def _run(m, qpos, qvel, ctrl, n_steps):
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    phys = BatchedPhysics(m, qpos.shape[0], 'cuda:0')
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    d.ctrl[:] = torch.as_tensor(ctrl, dtype=torch.float32)
    r64 = lambda t: t.cpu().numpy().astype(np.float64)
    ins = r64(d.qpos), r64(d.qvel), r64(d.ctrl)
    phys.step(n_steps)
    torch.cuda.synchronize()
    assert int(d.status.abs().sum()) == 0
    return phys, ins, r64(d.qpos), r64(d.qvel)


@pytest.mark.parametrize('clamp', [False, True])
@pytest.mark.parametrize('kw,dual', [({}, '1'), ({}, '0'), (dict(contacts=True, limits=True, spawn_z=0.045), '1'), (dict(contacts=True, limits=True, spawn_z=0.045), '0'),
                                     (dict(contacts=True, limits=True, spawn_z=0.045, solver='newton'), '0')],
                         ids=['two_per_wave', 'one_per_wave', 'two_per_wave_constrained', 'one_per_wave_constrained', 'one_per_wave_newton'])
def test_implicitfast_step_matches_oracle(oracle, monkeypatch, kw, dual, clamp):
    """One step and 200 steps from random states with random velocity-servo targets.  qvel is held to the same fp32-storage floor as
    the Euler step (tests/test_gpu_step_parity.py), and the test has teeth: the Euler result of the same inputs is further from the
    oracle's implicitfast than the bound allows."""
    n, maker, tpe = 32, 'salamander33', 32 if dual == '1' else 64
    monkeypatch.setenv('FMJ_DUAL', dual)                 # '0': the one-env kernels step the same model
    m = _servo_model(maker, 'implicitfast', clamp=clamp, **kw)
    # (the primal solvers are held to states of walking speed, as in tests/test_gpu_newton.py: from the violent ones an fp32 Newton
    # iteration and the fp64 one stop at different iterates - 4 % of qvel, under Euler as under implicitfast)
    qpos, qvel, ctrl = _state(m, n, 11, vscale=0.02 if kw.get('solver') else 0.3)
    phys, (q32, v32, c32), q1, v1 = _run(m, qpos, qvel, ctrl, 1)
    assert phys.kernel_info()['threads_per_env'] == tpe
    ref = oracle.step(m, q32, v32, ctrl=c32)
    with oracle.fp32_storage():
        floor = oracle.step(m, q32, v32, ctrl=c32)
    groups = qvel_groups(m)
    err = group_relerr(v1, ref['qvel'], groups); fl = group_relerr(floor['qvel'], ref['qvel'], groups)
    print(maker, 'clamp', clamp, 'qvel err', err, 'floor', fl)
    # with constraints the 50-sweep PGS iterate adds its own fp32 error (tests/test_gpu_contacts.py bounds a single step from slow random
    # states by 2e-3; these states move 6x faster and measure 0.8e-3 .. 2.1e-3, the one-env and the two-env kernel bitwise alike)
    bound = lambda floor: max(6*floor + 2e-6, 5e-3 if kw else 0.0)
    assert err < bound(fl), (err, fl)
    me = _servo_model(maker, 'Euler', clamp=clamp, **kw)
    eul = oracle.step(me, q32, v32, ctrl=c32)
    gap = group_relerr(eul['qvel'], ref['qvel'], groups)
    assert gap > 20*bound(fl), (gap, fl)             # implicitfast is not Euler on these inputs ...
    with oracle.fp32_storage():
        fle = group_relerr(oracle.step(me, q32, v32, ctrl=c32)['qvel'], eul['qvel'], groups)
    _, _, _, ve = _run(me, qpos, qvel, ctrl, 1)
    assert group_relerr(ve, eul['qvel'], groups) < bound(fle)     # ... and the device's Euler still is Euler
    # 200 steps (swimmers in free space / the walker settling on its feet)
    _, (q32, v32, c32), q200, v200 = _run(m, qpos, qvel, ctrl, 200)
    ref = oracle.step(m, q32, v32, ctrl=c32, n_steps=200)
    e = relerr(q200, ref['qpos'])
    print('qpos after 200 steps', e)
    assert e < (1e-4 if not kw else 1e-3), e


@pytest.mark.parametrize('substeps', [1, 2])
def test_implicitfast_fused_run_matches_oracle(oracle, substeps):
    """The fused rollout (Simulation.run's loop in one launch) under implicitfast, against the oracle's fused restatement: the wave
    controller drives the position actuators, the velocity servos hold 0 - a damper that implicitfast integrates implicitly."""
    import torch
    import farms_mujoco_amd.model as mm
    from test_gpu_fused_parity import _make_sim, _oracle_initial_state, _swim_water

    def servo(m, integrator='implicitfast'):
        for a, tag in enumerate(m.actuator_tags):
            if tag == 'velocity':
                m.actuator_gain[a] = 2e-3; m.actuator_bias[a, 2] = -2e-3
        m.integrator = mm.INTEGRATORS[integrator]

    n, T = 16, 200
    sim, m, psi = _make_sim(n, T, model_hook=servo, substeps=substeps)      # (with sub-steps: the same instantiation of the two-env kernel carries both options)
    assert sim.task.fusable() and m.integrator == 3
    st = _oracle_initial_state(oracle, sim, m)
    swim, water = _swim_water(sim)
    sim.run(fused=True)
    torch.cuda.synchronize()
    c = sim.task._controller
    wave = dict(amplitude=c.amplitude.cpu().numpy(), phase_lag=c.phase_lag.cpu().numpy(), env_phase=c.env_phase.cpu().numpy(), frequency=c.frequency)
    ref = oracle.run_fused(m, st, T, swim=swim, water=water, buffer_size=T, controller=1, wave=wave, n_threads=8, substeps=substeps)
    d = sim.physics.data
    assert int(d.status.abs().sum()) == 0
    links = sim.task.data.sensors.links.array.cpu().numpy()
    errs = dict(qpos=relerr(d.qpos.cpu().numpy(), ref['qpos']), links=relerr(links, ref['links']))
    print(errs)
    assert errs['qpos'] < 1e-4 and errs['links'] < 1e-4, errs
    m.integrator = 0                                     # the same rollout under Euler ends somewhere else
    eul = oracle.run_fused(m, st, T, swim=swim, water=water, buffer_size=T, controller=1, wave=wave, n_threads=8, substeps=substeps)
    # (explicit servos of this gain on the light limbs blow up at h = 1e-3; at h = 5e-4 they hold and end 4.7e-4 away - the device is at 2e-6)
    assert not np.isfinite(eul['qpos']).all() or relerr(eul['qpos'], ref['qpos']) > 2e-4
