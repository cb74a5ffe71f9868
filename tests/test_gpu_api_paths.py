"""GPU tests of the API-level paths around the step (b): Simulation.run / iterator on the
reference's own 1-env configuration (BASELINE configs[0]), torque control + spring references, run-time actuator
disabling, the bad-state policy (status bit -> frozen env -> PhysicsError), sensor-row subsets, the per-stage
comparison of the mass matrix and the bias force, and the single-link drag_forces operator."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max()/max(np.abs(b).max(), 1e-12)


def _f64(t):
    return t.cpu().numpy().astype(np.float64)


def _swim_sim(n_envs, n_iterations, buffer_size=None, seed=0, data=None, handle_exceptions=False, controller='wave',
              swimming_links=None):
    import torch
    from farms_mujoco_amd.model import salamander33, synthetic_batch
    from farms_mujoco_amd.options import SimulationOptions, ArenaOptions, AnimatOptions, WaterOptions
    from farms_mujoco_amd.control import WaveController
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = salamander33()
    qpos, qvel, psi = synthetic_batch(m, n_envs, seed=seed)
    kw = dict(data=data) if data is not None else {}
    ao = AnimatOptions.from_model(m)
    if swimming_links is not None:
        for link in ao.morphology.links:
            link.swimming = link.swimming and link.name in swimming_links
    sim = Simulation.from_sdf(SimulationOptions(timestep=m.timestep, n_iterations=n_iterations), ao,
                              ArenaOptions(water=WaterOptions(height=0.0)), model=m, n_envs=n_envs,
                              controller=WaveController(m, psi) if controller == 'wave' else controller,
                              buffer_size=buffer_size or n_iterations, handle_exceptions=handle_exceptions, **kw)
    sim.reset()
    d = sim.physics.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32)
    d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    return sim, m, psi


def _oracle_state(oracle, sim, m):
    d = sim.physics.data
    st = dict(qpos=_f64(d.qpos), qvel=_f64(d.qvel))
    xp, xq, xi, sd = [], [], [], []
    for e in range(st['qpos'].shape[0]):
        o = oracle.forward_debug(m, st['qpos'][e], st['qvel'][e])
        s = o['sensordata'].copy(); s[6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
        xp.append(o['xpos']); xq.append(o['xquat']); xi.append(o['xipos']); sd.append(s)
    st.update(xpos=np.array(xp), xquat=np.array(xq), xipos=np.array(xi), sensordata=np.array(sd))
    return st


def _swim_water_wave(sim):
    h = sim.task._callbacks[0].handler
    c = sim.task._controller
    water = dict(surface=h.water._surface, velocity=h.water._velocity, viscosity=h.water._viscosity, gravity=-9.81,
                 use_buoyancy=h.buoyancy)
    wave = dict(amplitude=c.amplitude.cpu().numpy(), phase_lag=c.phase_lag.cpu().numpy(),
                env_phase=c.env_phase.cpu().numpy(), frequency=c.frequency)
    return h.swim_dict(), water, wave


# ---- BASELINE configs[0]: one salamander, one env, through the reference's loop -----------------------------------

@pytest.mark.parametrize('driver', ['run', 'iterator'])
def test_config0_single_env_run_and_iterator(oracle, driver):
    """The reference's own configuration: ONE env stepped operator by operator by Simulation.run(fused=False) /
    Simulation.iterator() (reference simulation.py:148-179: before_step -> physics.step -> after_step per iteration),
    1000 iterations, against the fp64 oracle: qpos within the north-star 1e-4, the logged rows too."""
    import torch
    T = 1000
    sim, m, psi = _swim_sim(1, T)
    st = _oracle_state(oracle, sim, m)
    swim, water, wave = _swim_water_wave(sim)
    if driver == 'run':
        sim.run(fused=False)
    else:
        seen = []
        for it in sim.iterator(show_progress=False, verbose=False):
            assert it == sim.task.iteration == sim.iteration       # the caller sees the iteration before it is stepped
            seen.append(it)
        assert seen == list(range(T))
    torch.cuda.synchronize()
    assert sim.task.iteration == T and sim.task.sim_iteration == T
    ref = oracle.run_fused(m, st, T, swim=swim, water=water, buffer_size=T, controller=1, wave=wave)
    d = sim.physics.data
    assert int(d.status.abs().sum()) == 0
    sens = sim.task.data.sensors
    errs = dict(qpos=_relerr(_f64(d.qpos), ref['qpos']), links=_relerr(sens.links.array.cpu().numpy(), ref['links']),
                joints=_relerr(sens.joints.array.cpu().numpy(), ref['joints']), xfrc=_relerr(sens.xfrc.array.cpu().numpy(), ref['xfrc']))
    print(driver, errs)
    assert errs['qpos'] < 1e-4 and errs['links'] < 2e-4 and errs['joints'] < 1e-3 and errs['xfrc'] < 2e-3, errs
    assert abs(float(d.time[0]) - T*m.timestep) < 1e-4


def test_fused_wave_updates_ctrl(oracle):
    """After a fused launch physics.data.ctrl holds the command of the launch's last iteration (what step_control,
    reference task.py:288-346, leaves there), so host callbacks that read it do not see stale zeros."""
    import torch
    n, T = 5, 37
    sim, m, psi = _swim_sim(n, T)
    sim.run(fused=True)
    torch.cuda.synchronize()
    c = sim.task._controller
    want = torch.zeros_like(sim.physics.data.ctrl)
    want[:, sim.task.maps['ctrl']['pos']] = c.positions(iteration=T - 1, time=(T - 1)*m.timestep, timestep=m.timestep)
    assert float((sim.physics.data.ctrl - want).abs().max()) < 5e-6
    assert float(want.abs().max()) > 0.1


# ---- a8: torque control, spring references, run-time actuator disabling --------------------------------------------

class _TorqueController:
    """AnimatController interface (reference task.py:292-346) commanding torques on some joints, positions on others,
    and moving the spring reference of the torque joints."""
    fusable = False

    def __init__(self, m, pos_joints, trq_joints, n_envs, device='cuda:0'):
        from farms_mujoco_amd.control import ControlType
        self.joints_names = {ControlType.POSITION: list(pos_joints), ControlType.VELOCITY: [], ControlType.TORQUE: list(trq_joints)}
        self.muscles_names = []
        self.n_envs, self.device = n_envs, device
        self.steps = 0

    def step(self, iteration, time, timestep):
        self.steps += 1

    def positions(self, iteration, time, timestep):
        import torch
        k = torch.arange(len(self.joints_names[0]), device=self.device, dtype=torch.float32)
        e = torch.arange(self.n_envs, device=self.device, dtype=torch.float32)
        return 0.25*torch.sin(2*np.pi*1.5*time + 0.4*k[None, :] + 0.3*e[:, None])

    def torques(self, iteration, time, timestep):
        import torch
        k = torch.arange(len(self.joints_names[2]), device=self.device, dtype=torch.float32)
        e = torch.arange(self.n_envs, device=self.device, dtype=torch.float32)
        return 2e-3*torch.cos(2*np.pi*2.0*time + 0.7*k[None, :] - 0.2*e[:, None])

    def springrefs(self, iteration, time, timestep):
        return {j: 0.1*np.sin(2*np.pi*time + i) for i, j in enumerate(self.joints_names[2])}


def _torque_model():
    """Salamander with passive stiffness on the limb joints so that qpos_spring matters."""
    import farms_mujoco_amd.model as mm
    m = mm.salamander33()
    for j, name in enumerate(m.joint_names):
        if name.startswith('joint_leg_'):
            m.jnt_stiffness[j] = 2e-3
    return m


def test_torque_control_springrefs_and_disabled_actuators(oracle):
    """step_control's torque branch (reference task.py:323-346): ctrl[trq idx] = torques * units.torques and
    qpos_spring[joint] = springref every iteration, with the position / velocity actuators of the torque-only motors
    switched off by forcerange = [0, 0] at initialize_control (task.py:253-286).  The unfused Simulation loop against
    the oracle stepping the same per-step ctrl and spring references with the same rewritten force ranges."""
    import torch
    from farms_mujoco_amd.options import SimulationOptions, AnimatOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    from farms_mujoco_amd.units import SimulationUnitScaling
    m = _torque_model()
    n, T = 4, 120
    pos_joints = [j for j in m.hinge_joint_names() if j.startswith('joint_body_')]
    trq_joints = [j for j in m.hinge_joint_names() if j.startswith('joint_leg_')]
    motors = [AnimatOptions.motor(j, control_types=('position',), gains=(1.0, 0.0)) for j in pos_joints] + \
             [AnimatOptions.motor(j, control_types=('torque',), gains=(0.1, 0.0)) for j in trq_joints]
    ao = AnimatOptions(name='salamander33', motors=motors)
    units = SimulationUnitScaling()
    ctl = _TorqueController(m, pos_joints, trq_joints, n)
    sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T, units=units), n_envs=n,
                     controller=ctl, animat_options=ao, buffer_size=T)
    fl0 = m.actuator_forcelimited.copy()
    sim.reset()
    assert not sim.task.fusable()
    # initialize_control rewrote the model: pos/vel actuators of the torque-only motors are force-limited to [0, 0]
    for j in trq_joints:
        ids = sim.task.maps['ctrl']['jntname2actid'][j]
        assert m.actuator_forcelimited[ids['pos']] == 1 and tuple(m.actuator_forcerange[ids['pos']]) == (0.0, 0.0)
        assert fl0[ids['pos']] == 0
    d = sim.physics.data
    rng = np.random.default_rng(2)
    q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 7:] += rng.uniform(-0.2, 0.2, (n, m.nq - 7)); q0[:, 2] = 1.0
    d.qpos[:] = torch.as_tensor(q0, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    q32, v32 = _f64(d.qpos), _f64(d.qvel)
    sim.run(fused=False)
    torch.cuda.synchronize()
    assert ctl.steps == T and sim.task.iteration == T
    # the oracle, one step at a time with the controller's commands (m already carries the rewritten force ranges)
    pos_idx = [m.actuator_names.index(f'actuator_position_{j}') for j in pos_joints]
    trq_idx = [m.actuator_names.index(f'actuator_torque_{j}') for j in trq_joints]
    qs = np.tile(m.qpos_spring, (n, 1))
    q, v = q32.copy(), v32.copy()
    for it in range(T):
        t = it*m.timestep
        ctrl = np.zeros((n, m.nu))
        ctrl[:, pos_idx] = ctl.positions(it, t, m.timestep).cpu().numpy()
        ctrl[:, trq_idx] = ctl.torques(it, t, m.timestep).cpu().numpy()*units.torques
        for joint, value in ctl.springrefs(it, t, m.timestep).items():
            qs[:, m.jnt_qposadr[m.joint_names.index(joint)]] = value
        o = oracle.step(m, q, v, ctrl=ctrl, qpos_spring=qs)
        q, v = o['qpos'], o['qvel']
    assert int(d.status.abs().sum()) == 0
    err = _relerr(_f64(d.qpos), q)
    print('torque-control qpos rel err after', T, 'steps:', err)
    assert err < 1e-4
    # the position actuators of the torque joints produced exactly zero force; their torque actuators did not
    sd = d.sensordata.cpu().numpy(); adr = 6*(m.nbody - 1) + 3*m.n_sensor_joints
    pos_of_trq = [m.actuator_names.index(f'actuator_position_{j}') for j in trq_joints]
    assert np.all(sd[:, [adr + a for a in pos_of_trq]] == 0.0)
    assert np.abs(sd[:, [adr + a for a in trq_idx]]).max() > 1e-4
    assert _relerr(sd, o['sensordata']) < 2e-3
    # had the position actuators stayed on (kp = 0.1 against a 0.2 rad offset), the limb joints would have moved differently
    m2 = _torque_model()
    assert m2.actuator_forcelimited[pos_of_trq].sum() == 0
    o2 = oracle.step(m2, q32, v32, ctrl=np.zeros((n, m.nu)), n_steps=20)
    o1 = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)), n_steps=20)
    assert np.abs(o2['qpos'] - o1['qpos']).max() > 1e-3


def test_set_actuator_forcerange_roundtrip(oracle):
    """fmj_set_actuator_forcerange: limits take effect on the next launch and can be lifted again."""
    import torch
    from farms_mujoco_amd.model import salamander33
    from farms_mujoco_amd.physics import BatchedPhysics
    m = salamander33()
    n = 3
    phys = BatchedPhysics(m, n)
    a = m.actuator_names.index('actuator_position_joint_body_4')
    ctrl = np.zeros((n, m.nu)); ctrl[:, a] = 0.5
    phys.data.ctrl[:] = torch.as_tensor(ctrl, dtype=torch.float32)
    adr = 6*(m.nbody - 1) + 3*m.n_sensor_joints
    phys.step(1)
    f_free = phys.data.sensordata[:, adr + a].cpu().numpy().copy()
    assert np.all(np.abs(f_free) > 0.3)
    lim = np.zeros(m.nu, np.int32); rng_ = np.zeros((m.nu, 2))
    lim[a] = 1; rng_[a] = (-0.05, 0.05)
    phys.reset(); phys.data.ctrl[:] = torch.as_tensor(ctrl, dtype=torch.float32)
    phys.set_actuator_forcerange(lim, rng_)
    phys.step(1)
    assert np.allclose(phys.data.sensordata[:, adr + a].cpu().numpy(), 0.05)
    ref = oracle.step(m, np.tile(m.key_qpos, (n, 1)), np.zeros((n, m.nv)), ctrl=ctrl)     # m carries the new range
    assert _relerr(_f64(phys.data.qvel), ref['qvel']) < 5e-4       # one step from rest with a ctrl jump: see test_gpu_step_parity
    phys.reset(); phys.data.ctrl[:] = torch.as_tensor(ctrl, dtype=torch.float32)
    phys.set_actuator_forcerange(np.zeros(m.nu, np.int32), np.zeros((m.nu, 2)))
    phys.step(1)
    assert np.array_equal(phys.data.sensordata[:, adr + a].cpu().numpy(), f_free)


# ---- a1: bad state -> status bit -> frozen env -> PhysicsError --------------------------------------------------------

@pytest.mark.parametrize('fused', [True, False])
def test_nan_state_freezes_env_and_raises(oracle, fused):
    """SURVEY 5 / reference simulation.py:157-161: a non-finite qvel in ONE env sets its status bit; that env is frozen
    (state left as it was, no rows written) while every other env steps exactly as if nothing had happened;
    Simulation.run raises PhysicsError, or returns quietly with handle_exceptions=True."""
    import torch
    from farms_mujoco_amd.physics import PhysicsError
    import farms_mujoco_amd._lib as L
    n, T, bad = 7, 60, 4               # an odd batch: env 4 shares its wave with env 5, env 6 has an idle partner half
    clean, m, _ = _swim_sim(n, T)
    clean.run(fused=fused)
    for handle in (False, True):
        sim, _, _ = _swim_sim(n, T, handle_exceptions=handle)
        sim.physics.data.qvel[bad, 9] = float('nan')
        q_before = sim.physics.data.qpos[bad].clone()
        if handle:
            sim.run(fused=fused)                                   # logged, swallowed (simulation.py:159-160)
        else:
            with pytest.raises(PhysicsError):
                sim.run(fused=fused)
        torch.cuda.synchronize()
        d = sim.physics.data
        st = d.status.cpu().numpy()
        assert st[bad] & 2 and not st[np.arange(n) != bad].any()   # FMJ_WARN_BADQVEL on the bad env only
        assert torch.equal(d.qpos[bad], q_before)                  # frozen: not integrated
        links = sim.task.data.sensors.links.array.cpu().numpy()
        # frozen: no rows written (the operator-by-operator loop logs row 0 before its first step finds the bad value)
        assert np.all(links[0 if fused else 1:, bad] == 0.0)
        if fused:                                                  # the launch went through: the others are untouched by it
            others = np.arange(n) != bad
            assert torch.equal(d.qpos[others], clean.physics.data.qpos[others])
            assert np.array_equal(links[:, others], clean.task.data.sensors.links.array.cpu().numpy()[:, others])
        else:
            assert sim.task.iteration <= sim.check_every           # the host noticed at its first look at the status words


def test_iterator_raises_physics_error(oracle):
    """Simulation.iterator (reference simulation.py:164-179) re-raises PhysicsError whatever handle_exceptions says."""
    from farms_mujoco_amd.physics import PhysicsError
    sim, m, _ = _swim_sim(3, 50, handle_exceptions=True)
    sim.check_every = 10
    sim.physics.data.qpos[1, 2] = float('inf')
    done = []
    with pytest.raises(PhysicsError):
        for it in sim.iterator(show_progress=False, verbose=False):
            done.append(it)
    assert done == list(range(10))
    assert int(sim.physics.data.status[1]) & 1                     # FMJ_WARN_BADQPOS


def test_blowup_mid_launch_freezes_at_last_finite_state(oracle):
    """An env that diverges inside a fused launch (a huge position gain) is stopped at the step whose acceleration /
    velocity left the finite range: its state stays finite, later ring rows are not written, the status word says why,
    the neighbour in the same wavefront is bitwise unaffected."""
    import torch
    from farms_mujoco_amd.model import salamander33, synthetic_batch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = salamander33()
    n, T = 4, 400
    qpos, qvel, psi = synthetic_batch(m, n)
    a = m.actuator_names.index('actuator_velocity_joint_body_3')
    tape = np.zeros((T, n, m.nu), np.float32)

    def run(poison):
        mm_ = salamander33()
        if poison:
            mm_.actuator_gain[a] = -5e3; mm_.actuator_bias[a, 2] = 5e3     # negative damping: exponential blow-up
        phys = BatchedPhysics(mm_, n)
        phys.data.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32)
        phys.data.qvel[:] = torch.as_tensor(qvel + 0.1, dtype=torch.float32)
        phys.step(T, ctrl_tape=torch.as_tensor(tape, device='cuda'))
        torch.cuda.synchronize()
        return phys
    bad = run(True)
    st = bad.data.status.cpu().numpy()
    assert (st & 6).all()                                          # BADQVEL or BADQACC everywhere
    assert torch.isfinite(bad.data.qpos).all() and torch.isfinite(bad.data.qvel).all()
    assert float(bad.data.time.max()) < T*m.timestep               # stopped early: time counts the steps actually taken
    good = run(False)
    assert int(good.data.status.abs().sum()) == 0 and abs(float(good.data.time[0]) - T*m.timestep) < 1e-4


@pytest.mark.parametrize('maker', ['salamander33', 'centipede'])      # two-env kernel / one-env kernel
def test_root_position_past_the_limit_mid_launch_is_not_committed(maker):
    """A root position about to leave the finite range (|x| > 1e10, mj_checkPos) in the middle of a launch is never stored: the
    env freezes at its last good state (include/fmj.h freeze contract) - with BADQPOS, or with BADQACC / BADQVEL if the fp32
    dynamics at |x| ~ 1e10 (one ulp = 1 km) gives up first - and the other envs go on."""
    import torch
    import farms_mujoco_amd.model as mm
    from farms_mujoco_amd.physics import BatchedPhysics
    m = getattr(mm, maker)()
    n, T = 4, 20
    qpos, qvel, _ = mm.synthetic_batch(m, n)
    qpos[1, 0] = 1e10 - 4e7; qvel[1, 0] = 9e9           # env 1 crosses 1e10 at its fifth step of 1e-3 s (9e6 m per step)
    phys = BatchedPhysics(m, n)
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    phys.step(T)
    torch.cuda.synchronize()
    st = d.status.cpu().numpy()
    assert st[1] & 7 and not st[[0, 2, 3]].any()
    assert float(d.qpos[1, 0].abs()) <= 1e10 and torch.isfinite(d.qpos).all()
    assert float(d.time[1]) < T*m.timestep - 1e-6 and abs(float(d.time[0]) - T*m.timestep) < 1e-5
    # ... and what it keeps is exactly the state after its last completed step: qpos, qvel AND time of step k, nothing of step k + 1
    # (round 4: the root position is tested before anything of the step is committed)
    k = int(round(float(d.time[1])/m.timestep))
    assert 0 < k < T
    phys2 = BatchedPhysics(m, n)
    d2 = phys2.data
    d2.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d2.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    phys2.step(k)
    torch.cuda.synchronize()
    assert int(d2.status[1]) == 0
    assert torch.equal(d.qpos[1], d2.qpos[1]) and torch.equal(d.qvel[1], d2.qvel[1])


# ---- ADVICE: links / xfrc subsets ----------------------------------------------------------------------------------------

def test_links_and_xfrc_subset_rows(oracle):
    """AnimatData may log a subset of the links, and a different subset / order of xfrc rows (allowed by the reference's
    get_physics2data_maps): the env stride of every row address is the tensor's real row count, in the fused loop and in
    the standalone drag operator, and nothing is written outside the tensors."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.model import salamander33
    m = salamander33()
    n, T = 6, 40
    links = [b for b in m.body_names[1:] if b.startswith('body_')]              # 12 of 28 links, all of them swimming
    xfrc = list(reversed(links)) + ['leg_front_L_3']                             # another order, one extra dry row
    nx = len(xfrc)

    def make(fused):
        data = AnimatData(m.timestep, T, n, links, m.hinge_joint_names(), xfrc=xfrc)
        flat = torch.full((T*n*nx*6 + 8192,), 7.0, device='cuda')               # sentinel everywhere + a guard behind the tensor
        data.sensors.xfrc.array = flat[:T*n*nx*6].view(T, n, nx, 6)
        sim, _, _ = _swim_sim(n, T, data=data, swimming_links=links)
        st = _oracle_state(oracle, sim, m)
        sim.run(fused=fused)
        torch.cuda.synchronize()
        return sim, flat, st
    sim_f, flat_f, st = make(True)
    sim_u, flat_u, _ = make(False)
    assert float((flat_f[T*n*nx*6:] - 7.0).abs().max()) == 0.0 and float((flat_u[T*n*nx*6:] - 7.0).abs().max()) == 0.0
    xf = sim_f.task.data.sensors.xfrc.array.cpu().numpy(); xu = sim_u.task.data.sensors.xfrc.array.cpu().numpy()
    assert np.all(xf[..., nx - 1, :] == 7.0)                                     # the dry extra row is never written
    assert np.abs(xf[-1, :, :nx - 1] - 7.0).min() > 1e-6                         # every swimming link got its row
    assert _relerr(xf, xu) < 5e-4
    swim, water, wave = _swim_water_wave(sim_f)
    links_body = [m.body_names.index(b) for b in links]
    ref = oracle.run_fused(m, st, T, swim=swim, water=water, buffer_size=T, controller=1, wave=wave, links_body=links_body,
                           n_xfrc=nx)
    assert _relerr(sim_f.task.data.sensors.links.array.cpu().numpy(), ref['links']) < 1e-4
    want = ref['xfrc'].copy(); want[..., nx - 1, :] = 7.0
    assert _relerr(xf, want) < 1e-3
    assert _relerr(_f64(sim_f.physics.data.qpos), ref['qpos']) < 1e-4


# ---- per-stage parity: mass matrix and bias force ---------------------------------------------------------------------------

@pytest.mark.parametrize('maker', ['salamander33', 'eel', 'centipede'])
def test_mass_matrix_and_bias_stage(oracle, maker):
    """SURVEY 4 per-stage check: H = M + diag(armature + h damping) as the step assembles it, and qfrc_smooth =
    passive - bias (actuation off), against the oracle's CRBA / RNE: every entry of H to 1e-5 of sqrt(H_ii H_jj) (the per-stage
    bound SURVEY 4 asks for; measured 2e-6 .. 5.5e-6), the smooth force to 2e-5 of the largest generalized force (measured 1e-7).
    The solve that follows amplifies these by the conditioning of H (light limb links on a heavy trunk), which is why the
    single-step qvel / qacc bound of test_gpu_step_parity is 1e-4 and not 1e-5."""
    import torch
    import farms_mujoco_amd.model as mm
    from farms_mujoco_amd.physics import BatchedPhysics
    from farms_mujoco_amd import _lib
    m = getattr(mm, maker)()
    n = 8
    rng = np.random.default_rng(5)
    qpos = np.tile(m.key_qpos, (n, 1)); qpos[:, 7:] += rng.uniform(-0.5, 0.5, (n, m.nq - 7))
    quat = rng.normal(size=(n, 4)); qpos[:, 3:7] = quat/np.linalg.norm(quat, axis=1, keepdims=True)
    qvel = rng.normal(size=(n, m.nv))*0.5
    phys = BatchedPhysics(m, n)
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    q32, v32 = _f64(d.qpos), _f64(d.qvel)
    rs = ctypes.c_int32()
    H = torch.zeros(n, m.nv, 32, device='cuda'); qf = torch.zeros(n, m.nv, device='cuda')
    c = phys._cdata()
    _lib.check(phys._lib.fmj_forward_debug(phys._ctx, ctypes.byref(c), 1, H.data_ptr(), ctypes.byref(rs), qf.data_ptr(), None))
    torch.cuda.synchronize()
    rs = rs.value
    Hrows = H.cpu().numpy().ravel()[:n*m.nv*rs].reshape(n, m.nv, rs)            # packed [n_envs, nv, rs]
    depth = np.zeros(m.nv, int)
    for i in range(m.nv):
        depth[i] = 0 if m.dof_parentid[i] < 0 else depth[m.dof_parentid[i]] + 1
    worst_M = worst_b = 0.0
    for e in range(n):
        o = oracle.forward_debug(m, q32[e], v32[e])
        Href = o['M'] + np.diag(m.timestep*m.dof_damping)       # armature is inside the oracle's M already
        scale = np.sqrt(np.outer(np.diag(Href), np.diag(Href)))
        for i in range(m.nv):
            j = i
            while j >= 0:
                worst_M = max(worst_M, abs(Hrows[e, i, depth[j]] - Href[i, j])/scale[i, j])
                j = m.dof_parentid[j]
        want = o['qfrc_passive'] - o['qfrc_bias']
        worst_b = max(worst_b, np.abs(qf[e].cpu().numpy() - want).max()/np.abs(want).max())
    print(maker, 'H rel err (vs sqrt(Hii Hjj))', worst_M, 'qfrc_smooth rel err', worst_b)
    assert worst_M < 1e-5 and worst_b < 2e-5


# ---- drag_forces, the single-link free function ----------------------------------------------------------------------------

def test_drag_forces_single_link(oracle):
    """drag_forces(iteration, data_links, links_index, data_xfrc, xfrc_index, ...) (reference drag.pyx:152-167) for one
    link in every env equals that link's row of SwimmingHandler.step, and reports which envs had the link in water."""
    import torch
    from farms_mujoco_amd.swimming.drag import drag_forces
    n = 9
    sim, m, _ = _swim_sim(n, 4)
    sim.physics.data.qpos[::2, 2] = 0.05                           # every other animal above the surface
    sim.physics.forward(disable_actuation=True)
    sim.physics.data.qvel[:] = 0.2*torch.randn(n, m.nv, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    sim.physics.step(2)
    sim.task.update_sensors(sim.physics)
    h = sim.task._callbacks[0].handler
    X = sim.task.data.sensors.xfrc.array
    X[0] = 3.0
    h.step(0)
    want = X[0].clone()
    X[0] = 3.0
    for i in (0, 5, h.n_links - 1):
        wet = drag_forces(0, sim.task.data.sensors.links, int(h.links_indices[i]), sim.task.data.sensors.xfrc, int(h.xfrc_indices[i]),
                          h.links_coefficients[i], water=h.water, mass=h.masses[i], height=h.heights[i], density=h.densities[i],
                          use_buoyancy=h.buoyancy)
        torch.cuda.synchronize()
        row = X[0, :, int(h.xfrc_indices[i])]
        assert torch.allclose(row, want[:, int(h.xfrc_indices[i])], rtol=2e-6, atol=1e-9)
        z = sim.task.data.sensors.links.array[0, :, int(h.links_indices[i]), 2]
        assert torch.equal(wet, z <= h.water._surface) and bool(wet.any()) and not bool(wet.all())
        assert torch.all(row[~wet] == 3.0)


def test_postprocess_writes_the_log_and_options(tmp_path):
    """Simulation.postprocess (reference simulation.py:181-213): the sensor log of the iterations run (``simulation.hdf5``, or
    ``.npz`` with the same keys where h5py is absent), the simulation and animat options as YAML; the log reads back
    into an AnimatData with the rows the run left in the ring buffer; save_mjcf_xml writes the compiled model."""
    import os
    import torch
    import yaml
    from farms_mujoco_amd.data import AnimatData
    sim, m, _ = _swim_sim(3, 30)
    sim.run(fused=True)
    torch.cuda.synchronize()
    sim.postprocess(iteration=sim.iteration, log_path=str(tmp_path))
    logs = [f for f in os.listdir(tmp_path) if f.startswith('simulation.') and f.rsplit('.', 1)[1] in ('hdf5', 'npz')]
    assert len(logs) == 1
    back = AnimatData.from_file(str(tmp_path / logs[0]))
    for k in ('links', 'joints', 'xfrc'):
        a = getattr(sim.task.data.sensors, k); b = getattr(back.sensors, k)
        assert list(map(str, a.names)) == list(map(str, b.names))
        assert np.array_equal(a.array.cpu().numpy(), b.array.cpu().numpy()), k
    assert abs(back.timestep - m.timestep) < 1e-15
    opts = yaml.safe_load(open(tmp_path / 'simulation_options.yaml'))
    assert opts['n_iterations'] == 30 and abs(opts['timestep'] - m.timestep) < 1e-15
    assert os.path.exists(tmp_path / 'animat_options.yaml')
    xml = open(sim.save_mjcf_xml(str(tmp_path / 'model.xml'))).read()
    assert xml.count('<body ') == m.nbody - 1 and 'solver="PGS"' in xml


def test_host_callbacks_with_one_launch_per_iteration():
    """Round 5: with a host callback an iteration of Simulation.run is ONE launch for a swimming model - the
    step's launch also writes the next iteration's rows, drag and xfrc_applied (fmj_fused_args::rows_ahead) - instead of the sensors'
    launch plus the step's.  The callback must see what it saw before: the rows of ITS iteration, the drag already in xfrc_applied, and
    what it adds to xfrc_applied must reach the step.  Compared with the two-launch path: equal to the fp32 rounding of the drag operator
    (the standalone operator and the in-kernel drag differ in the last bit, tests/test_gpu_fused_parity.py)."""
    import torch
    from farms_mujoco_amd.simulation.task import TaskCallback

    class Push(TaskCallback):
        def __init__(self):
            super().__init__()
            self.seen = []

        def before_step(self, task, action, physics):
            idx = task.iteration % task.buffer_size
            row = task.data.sensors.links.array[idx]
            self.seen.append((task.iteration, row[:, 3, :3].clone(), task.data.sensors.joints.array[idx][:, 5, 0].clone(),
                              physics.data.xfrc_applied[:, 4, :3].clone()))
            if 5 <= task.iteration < 10:
                physics.data.xfrc_applied[:, 7, 1] += 0.02           # a sideways push on one link, on top of its drag

    def run(ahead):
        sim, m, _ = _swim_sim(6, 40, controller='wave')
        cb = Push()
        sim.task._callbacks.append(cb)
        assert not sim.task.fusable()
        sim._ahead_ok = None if ahead else False
        sim.run()
        torch.cuda.synchronize()
        assert (sim._ahead_ok is True) == ahead
        s = sim.task.data.sensors
        return sim, cb, {k: getattr(s, k).array.cpu().numpy() for k in ('links', 'joints', 'xfrc')}
    sim_a, cb_a, rows_a = run(True)
    sim_b, cb_b, rows_b = run(False)
    close = lambda a, b, tol=5e-4: _relerr(_f64(a) if hasattr(a, 'cpu') else a, _f64(b) if hasattr(b, 'cpu') else b) < tol
    for k in rows_a:
        assert close(rows_a[k], rows_b[k]), (k, _relerr(rows_a[k], rows_b[k]))
    for f in ('qpos', 'qvel', 'xfrc_applied', 'ctrl', 'sensordata'):
        assert close(getattr(sim_a.physics.data, f), getattr(sim_b.physics.data, f)), f
    assert torch.equal(sim_a.physics.data.time, sim_b.physics.data.time)
    assert len(cb_a.seen) == len(cb_b.seen) == 40
    for (ia, la, ja, xa), (ib, lb, jb, xb) in zip(cb_a.seen, cb_b.seen):
        assert ia == ib and close(la, lb) and close(ja, jb) and close(xa, xb), ia
    assert torch.equal(cb_a.seen[0][1], cb_b.seen[0][1])           # iteration 0: both read the reset's rows, written by the same operator
    assert float(cb_a.seen[7][3].abs().max()) > 0                  # the callback saw the drag of its iteration in xfrc_applied
    # ... and the push moved the animal: the same run without it ends elsewhere
    sim_c, _, _ = _swim_sim(6, 40, controller='wave')
    sim_c.run(fused=False)
    assert not torch.equal(sim_c.physics.data.qpos, sim_a.physics.data.qpos)
    # a callback that edits the state itself opts out of the look-ahead
    cb_a.writes_state = True
    assert not sim_a.task.rows_ahead_ok(sim_a.physics)


def test_before_step_operator_equals_the_three_operators(oracle):
    """fmj_before_step (ABI 6) = fmj_physics2data + fmj_contacts2data + fmj_drag in one launch, bit for bit: swimming (rows + drag +
    xfrc_applied) and walking (rows + contact rows), full rows and links-only rows."""
    import torch
    from farms_mujoco_amd import _lib
    from farms_mujoco_amd.simulation.physics import physics2data
    from farms_mujoco_amd.sensors.sensors import cycontacts2data
    # swimming
    sim, m, _ = _swim_sim(5, 4)
    sim.step_fused(3)                                    # a state with velocities
    phys, task = sim.physics, sim.task
    h = task._callbacks[0].handler
    s = task.data.sensors

    def three(index, links_only):
        rows = _lib.CRows(); rows.links = s.links.array[index].data_ptr(); rows.joints = s.joints.array[index].data_ptr()
        c = phys._cdata(); u = task.units.as_c()
        _lib.check(phys._lib.fmj_physics2data(phys._ctx, ctypes.byref(c), ctypes.byref(rows), ctypes.byref(u), int(links_only),
                                              ctypes.c_void_p(torch.cuda.current_stream(phys.device).cuda_stream)))
        h.step(index)
    for links_only in (False, True):
        for arr in (s.links.array, s.joints.array, s.xfrc.array):
            arr.zero_()
        phys.data.xfrc_applied.zero_()
        three(1, links_only)
        want = [a[1].clone() for a in (s.links.array, s.joints.array, s.xfrc.array)] + [phys.data.xfrc_applied.clone()]
        for arr in (s.links.array, s.joints.array, s.xfrc.array):
            arr.zero_()
        phys.data.xfrc_applied.zero_()
        physics2data(phys, 1, task.data, task.maps, task.units, links_only=links_only, swimming=h)
        got = [a[1] for a in (s.links.array, s.joints.array, s.xfrc.array)] + [phys.data.xfrc_applied]
        for w, g, name in zip(want, got, ('links', 'joints', 'xfrc', 'xfrc_applied')):
            assert torch.equal(w, g), (name, links_only)
        assert float(want[0].abs().max()) > 0 and float(want[3].abs().max()) > 0 and (links_only or float(want[1].abs().max()) > 0)
    # walking: contact rows
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.model import salamander33
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    mw = salamander33(contacts=True, limits=True, spawn_z=0.03)
    pairs = [(b, '') for b in mw.body_names[1:] if b.endswith('_3')] + [('world', '')]
    data = AnimatData(mw.timestep, 4, 3, mw.body_names[1:], mw.hinge_joint_names(), contacts=pairs)
    simw = Simulation(mw, mw.body_names[1], SimulationOptions(timestep=mw.timestep, n_iterations=4), n_envs=3, data=data, buffer_size=4)
    simw.reset()
    simw.physics.step(30)
    sw = data.sensors
    rows = _lib.CRows(); rows.links = sw.links.array[2].data_ptr(); rows.joints = sw.joints.array[2].data_ptr()
    c = simw.physics._cdata(); u = simw.task.units.as_c()
    _lib.check(simw.physics._lib.fmj_physics2data(simw.physics._ctx, ctypes.byref(c), ctypes.byref(rows), ctypes.byref(u), 0,
                                                  ctypes.c_void_p(torch.cuda.current_stream(simw.physics.device).cuda_stream)))
    cycontacts2data(physics=simw.physics, iteration=2, data=sw.contacts, geompair2data=simw.task.maps['sensors']['geompair2data'],
                    meters=simw.task.units.meters, newtons=simw.task.units.newtons)
    want = [a[2].clone() for a in (sw.links.array, sw.joints.array, sw.contacts.array)]
    for a in (sw.links.array, sw.joints.array, sw.contacts.array):
        a.zero_()
    physics2data(simw.physics, 2, data, simw.task.maps, simw.task.units)
    for w, a, name in zip(want, (sw.links.array, sw.joints.array, sw.contacts.array), ('links', 'joints', 'contacts')):
        assert torch.equal(w, a[2]), name
    assert float(want[2].abs().max()) > 1e-3
