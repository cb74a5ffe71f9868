"""Other morphologies through the same HIP path (BASELINE config 5: eel + centipede, variable link count):
each morphology gets its own context (bucketed batching: no padding, no masks), checked against the oracle."""
import numpy as np
import pytest

from parity_metrics import relerr as _relerr, group_relerr, qpos_groups, qvel_groups, link_row_groups

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('maker', ['eel', 'centipede', 'eel40', 'eel58'])
def test_step_parity_other_morphologies(oracle, maker):
    """eel40 / eel58: dof chains of 46 and 64 - register rows longer than 32, the unconstrained one-env kernel's MAXD 48 / 64 builds."""
    import torch
    import farms_mujoco_amd.model as mm
    from farms_mujoco_amd.physics import BatchedPhysics
    m = mm.eel(n_joints=int(maker[3:])) if maker.startswith('eel') and len(maker) > 3 else getattr(mm, maker)()
    n, T = 16, 300
    qpos, qvel, psi = mm.synthetic_batch(m, n, seed=4)
    amp, lag = mm.wave_controller_params(m, amplitude=0.25)
    t = np.arange(T)[:, None, None]*m.timestep
    tape = amp[None, None, :]*np.sin(2*np.pi*1.5*t - lag[None, None, :] + psi[None, :, None])
    phys = BatchedPhysics(m, n)
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    tape_t = torch.as_tensor(tape, dtype=torch.float32, device='cuda').contiguous()
    phys.step(1, ctrl_tape=tape_t[:1].contiguous())
    torch.cuda.synchronize()
    q32 = torch.as_tensor(qpos, dtype=torch.float32).numpy().astype(np.float64)
    ref1 = oracle.step(m, q32, qvel, ctrl=tape_t[:1].cpu().numpy().astype(np.float64), n_steps=1, ctrl_step_stride=n*m.nu)
    # the first step starts from rest with a ctrl jump (qacc ~ 5e4 rad/s^2).  The joint-space inertia of these long, light
    # chains is ill-conditioned (scaled condition number ~1e5): storing it in fp32 alone moves the solution by 1e-4 .. 5e-4 of
    # its maximum (oracle.fp32_storage: the fp64 oracle with M / H rounded to fp32, nothing else).  The velocity bound is
    # stated against that floor, per component, instead of a fitted number.
    with oracle.fp32_storage():
        floor1 = oracle.step(m, q32, qvel, ctrl=tape_t[:1].cpu().numpy().astype(np.float64), n_steps=1, ctrl_step_stride=n*m.nu)
    for k, tol in (('xpos', 2e-6), ('xquat', 2e-6), ('sensordata', 5e-5)):
        assert _relerr(getattr(d, k).cpu().numpy(), ref1[k]) < tol, (maker, k, _relerr(getattr(d, k).cpu().numpy(), ref1[k]))
    for k, groups in (('qvel', qvel_groups(m)), ('qpos', qpos_groups(m))):
        err = group_relerr(getattr(d, k).cpu().numpy(), ref1[k], groups); fl = group_relerr(floor1[k], ref1[k], groups)
        print(maker, k, 'first step: per-component err', err, 'fp32-storage floor', fl)
        assert err < 6*fl + 1e-6, (maker, k, err, fl)
    phys.step(T - 1, ctrl_tape=tape_t[1:].contiguous())
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, qvel, ctrl=tape_t.cpu().numpy().astype(np.float64), n_steps=T, ctrl_step_stride=n*m.nu, n_threads=8)
    assert int(d.status.abs().sum()) == 0
    err = _relerr(d.qpos.cpu().numpy(), ref['qpos'])
    print(maker, 'nv', m.nv, 'qpos rel err after', T, 'steps:', err)
    if m.nv <= 61:
        assert err < 1e-4
    else:
        # a chain of 58 light links: the scaled condition number of its joint-space inertia is > 1e6, and storing M / H in fp32 alone
        # moves the first step by 0.18 per component (printed above) - the 300-step bound is stated against that floor's own rollout
        with oracle.fp32_storage():
            fl = oracle.step(m, q32, qvel, ctrl=tape_t.cpu().numpy().astype(np.float64), n_steps=T, ctrl_step_stride=n*m.nu, n_threads=8)
        flo = _relerr(fl['qpos'], ref['qpos'])
        print(maker, 'fp32-storage floor of the same rollout:', flo)
        assert err < max(1e-4, 6*flo) and err < 1e-3      # INTEGRATION.md "Model size": beyond nv ~ 60 the 1e-4 statement is 1e-3


def _bucket_sim(maker, n, T, ring, env_offset=0, seed=9, twins=()):
    """One morphology bucket of the mixed batch: fused swimming with drag, wave controller, inputs keyed by global env
    index."""
    import torch
    import farms_mujoco_amd.model as mm
    from farms_mujoco_amd.options import SimulationOptions, ArenaOptions, AnimatOptions, WaterOptions
    from farms_mujoco_amd.control import WaveController
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = mm.eel(n_joints=int(maker[3:])) if maker.startswith('eel') and len(maker) > 3 else getattr(mm, maker)()
    qpos, qvel, psi = mm.synthetic_batch(m, n, seed=seed, env_offset=env_offset)
    for e in twins:
        qpos[e] = qpos[0]; qvel[e] = qvel[0]; psi[e] = psi[0]
    sim = Simulation.from_sdf(SimulationOptions(timestep=m.timestep, n_iterations=T), AnimatOptions.from_model(m),
                              ArenaOptions(water=WaterOptions(height=0.0)), model=m, n_envs=n,
                              controller=WaveController(m, psi, frequency=1.5), buffer_size=ring)
    sim.reset()
    d = sim.physics.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    return sim, m


def _bucket_oracle(oracle, sim, m, T, ring, envs):
    """The oracle's fused loop on a sample of a bucket's envs, from the bucket's own fp32 inputs."""
    d = sim.physics.data
    q = d.qpos[envs].cpu().numpy().astype(np.float64); v = d.qvel[envs].cpu().numpy().astype(np.float64)
    st = dict(qpos=q, qvel=v)
    fds = [oracle.forward_debug(m, q[i], v[i]) for i in range(len(envs))]
    for k in ('xpos', 'xquat', 'xipos'):
        st[k] = np.array([fd[k] for fd in fds])
    sd = np.array([fd['sensordata'] for fd in fds]); sd[:, 6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
    st['sensordata'] = sd
    h = sim.task._callbacks[0].handler
    c = sim.task._controller
    water = dict(surface=h.water._surface, velocity=h.water._velocity, viscosity=h.water._viscosity, gravity=-9.81, use_buoyancy=h.buoyancy)
    wave = dict(amplitude=c.amplitude.cpu().numpy(), phase_lag=c.phase_lag.cpu().numpy(), env_phase=c.env_phase[envs].cpu().numpy(),
                frequency=c.frequency)
    return oracle.run_fused(m, st, T, swim=h.swim_dict(), water=water, buffer_size=ring, controller=1, wave=wave, n_threads=8)


# Absolute caps next to the "6 x floor" rule (round 4): the per-component metric divides by the size of each entry, and the entries of a
# long eel's rows that are nearly zero (lateral velocities of the head links, torques of 1e-9 N m on the tail) carry relative errors of
# tens of percent in ANY fp32 run - the fp64 oracle with its mass matrix rounded to fp32 shows 5 - 18 % there.  What such a run can be
# held to in absolute terms is the whole-tensor figure max |a - b| / max |b|:
CAP48 = dict(qpos=2e-3, links=1.5e-2, xfrc=8e-3)      # measured 7.6e-4, 5.4e-3, 2.6e-3; the fp32-storage floor run: 3.8e-4, 1.7e-3, 8.7e-4


def test_fused_swim_of_a_long_eel(oracle):
    """The fused loop (rows, drag, wave controller) of an eel of 48 joints - a dof chain of 54: the MAXD 56 build of the
    unconstrained one-env kernel - against the oracle's fused loop, bounds against the fp32-storage floor like the mixed batch."""
    import torch
    T = 40
    sim, m = _bucket_sim('eel48', 12, T, T)
    assert m.nv == 54
    ref = _bucket_oracle(oracle, sim, m, T, T, list(range(12)))
    with oracle.fp32_storage():
        flo = _bucket_oracle(oracle, sim, m, T, T, list(range(12)))
    sim.run(fused=True)
    torch.cuda.synchronize()
    assert int(sim.physics.data.status.abs().sum()) == 0
    sens = sim.task.data.sensors
    got = dict(qpos=sim.physics.data.qpos.cpu().numpy(), links=sens.links.array.cpu().numpy(), xfrc=sens.xfrc.array.cpu().numpy())
    groups = dict(qpos=qpos_groups(m), links=link_row_groups(), xfrc=[slice(0, 3), slice(3, 6)])
    for k in got:
        err = group_relerr(got[k], ref[k], groups[k]); fl = group_relerr(flo[k], ref[k], groups[k])
        whole, whole_fl = _relerr(got[k], ref[k]), _relerr(flo[k], ref[k])
        print('eel48', k, 'per-component err', err, 'fp32-storage floor', fl, ' whole-tensor err', whole, 'floor', whole_fl)
        assert err < 6*fl + 1e-6, (k, err, fl)
        assert whole < CAP48[k], (k, whole)
    assert np.abs(sens.links.array.cpu().numpy()[-1, :, :, 14:17]).max() > 1e-3      # it swims


def test_mixed_batch_bucketed(oracle):
    """Mixed-morphology batch = one fused simulation per morphology bucket, launched back to back on the same
    stream; each bucket matches the oracle (state and logged rows) and is unaffected by the presence of the other."""
    import torch
    T = 40
    sims = [_bucket_sim(maker, n, T, T) for maker, n in (('eel', 24), ('centipede', 8))]
    refs = [_bucket_oracle(oracle, sim, m, T, T, list(range(sim.physics.n_envs))) for sim, m in sims]
    with oracle.fp32_storage():       # the floor: the same fp64 loop with its stored M / H rounded to fp32 and nothing else
        floors = [_bucket_oracle(oracle, sim, m, T, T, list(range(sim.physics.n_envs))) for sim, m in sims]
    for sim, _ in sims:
        sim.run(fused=True)
    torch.cuda.synchronize()
    for (sim, m), ref, flo in zip(sims, refs, floors):
        assert int(sim.physics.data.status.abs().sum()) == 0
        sens = sim.task.data.sensors
        got = dict(qpos=sim.physics.data.qpos.cpu().numpy(), links=sens.links.array.cpu().numpy(), xfrc=sens.xfrc.array.cpu().numpy(),
                   joints=sens.joints.array.cpu().numpy())
        groups = dict(qpos=qpos_groups(m), links=link_row_groups(), xfrc=[slice(0, 3), slice(3, 6)], joints=[slice(0, 1), slice(1, 2), slice(8, 9)])
        errs = {k: group_relerr(got[k], ref[k], groups[k]) for k in got}
        fls = {k: group_relerr(flo[k], ref[k], groups[k]) for k in got}
        print(m.name, 'err', errs, 'fp32-storage floor', fls)
        # the rows carry velocities and forces that start from rest with a ctrl jump; on these long light chains they are bounded
        # by what fp32 storage of the inertia matrix alone costs over the same 40 steps (per component), qpos by the 1e-4 target
        assert _relerr(got['qpos'], ref['qpos']) < 1e-4
        for k in got:
            assert errs[k] < 6*fls[k] + 1e-6, (m.name, k, errs[k], fls[k])
        assert np.abs(sens.links.array.cpu().numpy()[-1, :, :, 14:17]).max() > 1e-3      # it swims
    # a bucket run alone gives bitwise the same rows
    alone, _ = _bucket_sim('eel', 24, T, T)
    alone.run(fused=True)
    torch.cuda.synchronize()
    assert torch.equal(alone.task.data.sensors.links.array, sims[0][0].task.data.sensors.links.array)


def test_buckets_side_by_side_equal_back_to_back():
    """BucketedSimulation (one HIP stream per morphology bucket, launches overlapping on the device) logs bitwise the
    rows and reaches bitwise the state of the same buckets stepped one after the other on the caller's stream."""
    import torch
    from farms_mujoco_amd.simulation.buckets import BucketedSimulation
    T, ring = 120, 40
    a = [_bucket_sim(maker, n, T, ring)[0] for maker, n in (('eel', 300), ('centipede', 200))]
    b = [_bucket_sim(maker, n, T, ring)[0] for maker, n in (('eel', 300), ('centipede', 200))]
    batch = BucketedSimulation(a, overlap=True)
    assert batch.n_envs == 500
    for _ in range(T // ring):
        assert batch.step_fused(ring) == ring
        for s in b:
            s.step_fused(ring)
    torch.cuda.synchronize()
    batch.check_invalid_state()
    for x, y in zip(a, b):
        assert torch.equal(x.physics.data.qpos, y.physics.data.qpos) and torch.equal(x.physics.data.qvel, y.physics.data.qvel)
        for k in ('links', 'joints', 'xfrc'):
            assert torch.equal(getattr(x.task.data.sensors, k).array, getattr(y.task.data.sensors, k).array), k
        assert x.task.iteration == T


def test_full_size_config4_mixed_properties(oracle):
    """BASELINE configs[4] at the per-GPU size bench.py runs (2048 eels + 2048 centipedes, fused with drag, 300 steps,
    ring of 100): no warning bits; per bucket an 8-env sample matches the oracle (qpos, link rows, xfrc rows); envs with
    identical inputs give identical rows wherever they sit in their bucket; the second half of each bucket run on its
    own reproduces the full run bitwise (sharding invariance)."""
    import torch
    T, ring, N = 300, 100, 2048
    for maker in ('eel', 'centipede'):
        twins = [1, N//2 + 1, N - 1]
        sim, m = _bucket_sim(maker, N, T, ring, twins=twins)
        sample = [0, 2, 3, 777, 1024, 1500, 2046, 2047]
        ref = _bucket_oracle(oracle, sim, m, T, ring, sample)
        sim.run(fused=True)
        torch.cuda.synchronize()
        d = sim.physics.data
        assert int(d.status.abs().sum()) == 0
        q = d.qpos.cpu().numpy(); links = sim.task.data.sensors.links.array.cpu().numpy(); xfrc = sim.task.data.sensors.xfrc.array.cpu().numpy()
        assert np.isfinite(q).all() and np.isfinite(links).all()
        for e in twins:
            assert np.array_equal(q[e], q[0]) and np.array_equal(links[:, e], links[:, 0])
        errs = dict(qpos=_relerr(q[sample], ref['qpos']), links=_relerr(links[:, sample], ref['links']), xfrc=_relerr(xfrc[:, sample], ref['xfrc']))
        print(maker, 'full-size sample vs oracle after', T, 'steps:', errs)
        assert errs['qpos'] < 1e-4 and errs['links'] < 3e-4 and errs['xfrc'] < 2e-3, (maker, errs)
        half = _half_run(maker, N, T, ring, twins)
        assert np.array_equal(half.physics.data.qpos.cpu().numpy(), q[N//2:])
        assert np.array_equal(half.task.data.sensors.links.array.cpu().numpy(), links[:, N//2:])
        assert np.abs(q[:, 0] - sim.physics.model.key_qpos[0]).mean() > 0.005          # they swam


def _half_run(maker, N, T, ring, twins):
    """The second half [N/2, N) of a bucket as its own simulation, with the same per-env inputs as in the full batch."""
    import torch
    import farms_mujoco_amd.model as mm
    m = getattr(mm, maker)()
    qpos, qvel, psi = mm.synthetic_batch(m, N, seed=9)
    for e in twins:
        qpos[e] = qpos[0]; qvel[e] = qvel[0]; psi[e] = psi[0]
    sim, _ = _bucket_sim(maker, N//2, T, ring, env_offset=N//2)
    d = sim.physics.data
    d.qpos[:] = torch.as_tensor(qpos[N//2:], dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel[N//2:], dtype=torch.float32)
    sim.task._controller.env_phase[:] = torch.as_tensor(psi[N//2:], dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    sim.run(fused=True)
    torch.cuda.synchronize()
    return sim


@pytest.mark.parametrize('n_envs', [1, 3])
def test_longest_chain_two_per_wave(oracle, n_envs):
    """Edge of the two-envs-per-wave kernel: a fixed-base chain of 31 hinges (nbody 32, nv 31, dof depth 31: the
    MAXD = 32 instantiation, every lane of a half in use) with 1 and 3 envs (a wave whose upper half is idle)."""
    import torch
    from farms_mujoco_amd.model import ModelBuilder
    from farms_mujoco_amd.physics import BatchedPhysics
    b = ModelBuilder('chain31', timestep=1e-3)
    rng = np.random.default_rng(7)
    parent = 'world'
    for i in range(31):
        ax = [(0, 0, 1), (0, 1, 0), (1, 0, 0)][i % 3]
        b.add_body(f'l{i}', parent, pos=(0.03, 0, 0) if i else (0, 0, 0.5), mass=0.02, ipos=(0.015, 0, 0),
                   inertia=(2e-6, 4e-6, 4e-6), joint='hinge', axis=ax, damping=2e-4, stiffness=0.01 if i % 4 == 0 else 0.0)
        b.add_position_actuator(f'joint_l{i}', kp=0.02)
        parent = f'l{i}'
    m = b.compile()
    assert m.nbody == 32 and m.nv == 31
    phys = BatchedPhysics(m, n_envs)
    assert phys.kernel_info()['threads_per_env'] == 32
    d = phys.data
    qpos = rng.uniform(-0.3, 0.3, (n_envs, m.nq)); qvel = rng.normal(size=(n_envs, m.nv))*0.3
    ctrl = rng.uniform(-0.3, 0.3, (n_envs, m.nu))
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    d.ctrl[:] = torch.as_tensor(ctrl, dtype=torch.float32)
    r64 = lambda t: t.cpu().numpy().astype(np.float64)
    q32, v32, c32, s32 = r64(d.qpos), r64(d.qvel), r64(d.ctrl), r64(d.qpos_spring)
    phys.step(30)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=c32, qpos_spring=s32, n_steps=30)
    assert int(d.status.abs().sum()) == 0
    # a 31-link whip is badly conditioned in fp32 (mass matrix condition number ~1e6): looser than the animal models
    for k, tol in (('qpos', 3e-4), ('qvel', 5e-3), ('xpos', 3e-4)):
        e = np.abs(r64(getattr(d, k)) - ref[k]).max()/max(np.abs(ref[k]).max(), 1e-9)
        assert e < tol, (k, e)
