"""Config 4 (BASELINE): joint limits + plane contacts, pyramidal cone, PGS — HIP path vs the fp64 oracle.
Contact parity versus MuJoCo itself is unverifiable here (SURVEY §7 hard parts); the oracle restates MuJoCo's
published soft-constraint model and is what the HIP path is held to."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max()/max(np.abs(b).max(), 1e-12)


def _walker(spawn_z=0.045):
    from farms_mujoco_amd.model import salamander33
    return salamander33(contacts=True, limits=True, spawn_z=spawn_z)


def _trot_tape(m, n, T, seed=0):
    """Trot-like position control: axial wave + diagonal limb pairs swinging in antiphase."""
    rng = np.random.default_rng(seed)
    psi = rng.uniform(0, 2*np.pi, n)
    t = np.arange(T)[:, None, None]*m.timestep
    tape = np.zeros((T, n, m.nu))
    for a in range(m.nu):
        if m.actuator_tags[a] != 'position':
            continue
        name = m.joint_names[m.actuator_jntid[a]]
        if name.startswith('joint_body_'):
            k = int(name.split('_')[-1])
            tape[:, :, a] = 0.2*np.sin(2*np.pi*1.0*t[:, :, 0] - 2*np.pi*k/11 + psi[None, :])
        elif name.endswith('_1'):      # shoulder pitch
            ph = 0.0 if ('front_L' in name or 'hind_R' in name) else np.pi
            tape[:, :, a] = 0.3*np.sin(2*np.pi*1.0*t[:, :, 0] + ph + psi[None, :])
    return tape


def _set(phys, qpos, qvel):
    import torch
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    return d.qpos.cpu().numpy().astype(np.float64), d.qvel.cpu().numpy().astype(np.float64)


def test_single_step_with_contacts_and_limit(oracle):
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _walker()
    n = 16
    rng = np.random.default_rng(0)
    qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 7:] += rng.uniform(-0.1, 0.1, (n, m.nq - 7))
    qpos[:, 7 + 3] = 1.25                         # one spine joint past its +1.2 rad limit
    qvel = rng.normal(size=(n, m.nv))*0.05
    phys = BatchedPhysics(m, n)
    q32, v32 = _set(phys, qpos, qvel)
    phys.step(1)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
    d = phys.data
    assert int(d.status.abs().sum()) == 0
    ncon_ref = np.array([oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu))['ncon'] for e in range(n)])
    assert np.array_equal(d.ncon.cpu().numpy(), ncon_ref) and ncon_ref.max() >= 3
    for k, tol in (('xpos', 2e-6), ('sensordata', 1e-3), ('qvel', 1e-3), ('qpos', 1e-5)):
        assert _relerr(getattr(d, k).cpu().numpy(), ref[k]) < tol, (k, _relerr(getattr(d, k).cpu().numpy(), ref[k]))
    # contact forces (mj_contactForce equivalent) against the oracle's pyramid forces
    for e in range(3):
        fd = oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu))
        f = fd['efc_force'][fd['nefc'] - 4*fd['ncon']:fd['nefc']].reshape(-1, 4)
        got = d.contact.cpu().numpy()[e, :fd['ncon']]
        assert np.allclose(got[:, 12], f.sum(1), rtol=5e-3, atol=1e-4)
        assert np.allclose(got[:, :3], fd['contact'][:fd['ncon'], :3], atol=1e-6)        # contact positions
        assert np.allclose(got[:, 3:6], [0, 0, 1], atol=1e-7)


def test_many_contacts_more_rows_than_lanes(oracle):
    """Belly on the ground: 24+ contacts = 96+ pyramid rows, more than the 64 rows the explicit PGS matrix holds, so
    the matrix-free PGS path runs; envs with few contacts in the same batch take the explicit path."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _walker(spawn_z=0.01)
    n = 8
    rng = np.random.default_rng(3)
    qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 7:] += rng.uniform(-0.05, 0.05, (n, m.nq - 7))
    qpos[n//2:, 2] = 0.045                         # second half stands on its feet
    qvel = rng.normal(size=(n, m.nv))*0.02
    phys = BatchedPhysics(m, n)
    q32, v32 = _set(phys, qpos, qvel)
    phys.step(1)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
    d = phys.data
    assert int(d.status.abs().sum()) == 0
    fds = [oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu)) for e in range(n)]
    ncon_ref = np.array([fd['ncon'] for fd in fds])
    assert np.array_equal(d.ncon.cpu().numpy(), ncon_ref)
    assert max(fd['nefc'] for fd in fds) > 64 and min(fd['nefc'] for fd in fds) <= 64
    for k, tol in (('xpos', 2e-6), ('qvel', 2e-3), ('qpos', 1e-5)):
        assert _relerr(getattr(d, k).cpu().numpy(), ref[k]) < tol, (k, _relerr(getattr(d, k).cpu().numpy(), ref[k]))
    for e in (0, n - 1):
        fd = fds[e]
        f = fd['efc_force'][fd['nefc'] - 4*fd['ncon']:fd['nefc']].reshape(-1, 4)
        got = d.contact.cpu().numpy()[e, :fd['ncon']]
        assert np.allclose(got[:, 12], f.sum(1), rtol=2e-2, atol=2e-4), np.abs(got[:, 12] - f.sum(1)).max()


def test_joint_limit_holds(oracle):
    """Position actuator drives a spine joint to 1.5 rad; the +1.2 rad limit stops it and jointlimitfrc reports the
    constraint force (row and sensordata), matching the oracle."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    from farms_mujoco_amd.model import salamander33
    m = salamander33(contacts=False, limits=True, spawn_z=1.0)
    n, T = 4, 400
    ctrl = np.zeros((n, m.nu)); a = m.actuator_names.index('actuator_position_joint_body_6'); ctrl[:, a] = 3.0
    phys = BatchedPhysics(m, n)
    q32, v32 = _set(phys, np.tile(m.qpos0, (n, 1)), np.zeros((n, m.nv)))
    d = phys.data
    d.ctrl[:] = torch.as_tensor(ctrl, dtype=torch.float32)
    phys.step(T)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=ctrl, n_steps=T)
    j = m.joint_names.index('joint_body_6'); qa = m.jnt_qposadr[j]
    sj = j - 1
    lim = d.sensordata.cpu().numpy()[:, 6*(m.nbody - 1) + 3*sj + 2]
    lim_ref = ref['sensordata'][:, 6*(m.nbody - 1) + 3*sj + 2]
    print('q', d.qpos.cpu().numpy()[0, qa], ref['qpos'][0, qa], 'limit force', lim[0], lim_ref[0])
    # MuJoCo's default solref/solimp make the limit soft: it yields until K*imp*penetration/R balances the
    # actuator; what matters here is that both implementations agree on where that is
    assert ref['qpos'][0, qa] > 1.2 and lim_ref[0] > 0.1
    assert _relerr(d.qpos.cpu().numpy(), ref['qpos']) < 1e-3
    assert np.allclose(lim, lim_ref, rtol=2e-2, atol=1e-3)


def test_walking_rollout(oracle):
    """Trot-like walking on the plane.  Contact switching is discontinuous and amplifies fp32 rounding (and the
    fp32 PGS can stop one sweep earlier or later than the fp64 one), so the bound is looser than the swimming
    config's 1e-4: after 100 steps every env is within 5e-3 and the median env within 5e-4 of the oracle; after
    300 steps nothing has blown up and the plane still carries the animal."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _walker()
    n, T = 8, 100
    tape = _trot_tape(m, n, T)
    phys = BatchedPhysics(m, n)
    q32, v32 = _set(phys, np.tile(m.qpos0, (n, 1)), np.zeros((n, m.nv)))
    tape_t = torch.as_tensor(tape, dtype=torch.float32, device='cuda').contiguous()
    phys.step(T, ctrl_tape=tape_t)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=tape_t.cpu().numpy().astype(np.float64), n_steps=T, ctrl_step_stride=n*m.nu, n_threads=8)
    d = phys.data
    assert int(d.status.abs().sum()) == 0 and int(ref['status'].sum()) == 0
    e = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max(1)
    print('walking qpos abs err after', T, 'steps per env:', e, 'ncon', d.ncon.cpu().numpy())
    assert e.max() < 5e-3 and np.median(e) < 5e-4
    phys.step(200, ctrl_tape=torch.as_tensor(_trot_tape(m, n, 300)[100:], dtype=torch.float32, device='cuda').contiguous())
    torch.cuda.synchronize()
    assert int(d.status.abs().sum()) == 0
    assert float(d.qpos[:, 2].min()) > 0.0 and float(d.qpos[:, 2].max()) < 0.1      # the plane holds the animal up


def test_contacts2data_rows(oracle):
    """cycontacts2data through the task/readout layer: rows equal the C restatement of reference sensors.pyx:20-190
    (oracle/fmj_oracle.c: fmjo_contacts2data) applied to the same contact list, units scaled, force-weighted contact
    position."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.physics import BatchedPhysics
    from farms_mujoco_amd.simulation.physics import get_sensor_maps, get_physics2data_maps, physics2data
    from farms_mujoco_amd.units import SimulationUnitScaling
    m = _walker(spawn_z=0.04)
    n = 4
    phys = BatchedPhysics(m, n)
    _set(phys, np.tile(m.qpos0, (n, 1)), np.zeros((n, m.nv)))
    phys.step(30)
    links = m.body_names[1:]
    pairs = [(b, '') for b in links if b.endswith('_3') or b in ('body_0', 'body_5', 'body_11')]
    data = AnimatData(m.timestep, 2, n, links, m.hinge_joint_names(), contacts=pairs)
    units = SimulationUnitScaling(meters=2.0, seconds=1.0, kilograms=3.0)
    maps = {'sensors': get_sensor_maps(phys)}
    get_physics2data_maps(phys, data.sensors, maps['sensors'])
    physics2data(phys, 1, data, maps, units)
    torch.cuda.synchronize()
    ncon = phys.data.ncon.cpu().numpy(); contact = phys.data.contact.cpu().numpy()
    assert ncon.min() >= 1
    con = oracle.contacts_from_hip(contact)
    plane = int(np.nonzero(m.geom_type == 0)[0][0])
    assert np.all(con[0, :ncon[0], 15] == plane) and np.all(m.geom_bodyid[con[0, :ncon[0], 16].astype(int)] > 0)
    want = oracle.contacts2data(con, ncon, maps['sensors']['geompair2data'], len(pairs), units.meters, units.newtons)
    got = data.sensors.contacts.array[1].cpu().numpy()
    assert np.allclose(got, want, rtol=1e-5, atol=1e-7)
    assert np.abs(got[..., 2]).max() > 0            # some vertical reaction force was logged
    assert float(data.sensors.contacts.array[0].abs().max()) == 0.0


def test_contact_pair_sensors_and_signs(oracle):
    """The four keys of reference sensors.pyx:163-169 on plane contacts (geom1 = the arena plane on the world body,
    geom2 = the link geom): a (link, '') sensor and a (link, 'world') body-pair sensor read the force ON the link
    (sign +1), a ('world', link) pair sensor and the ('world', '') sensor read its reaction (sign -1); the body-pair
    keys are every ordered geom pair of the two bodies (reference physics.py:367-374)."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.physics import BatchedPhysics
    from farms_mujoco_amd.simulation.physics import get_sensor_maps, get_physics2data_maps, physics2data
    from farms_mujoco_amd.units import SimulationUnitScaling
    m = _walker(spawn_z=0.03)
    n = 6
    phys = BatchedPhysics(m, n)
    rng = np.random.default_rng(4)
    qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 7:] += rng.uniform(-0.1, 0.1, (n, m.nq - 7))
    _set(phys, qpos, 0.02*rng.normal(size=(n, m.nv)))
    phys.step(40)
    foot, trunk = 'leg_front_L_3', 'body_5'
    pairs = [(foot, ''), (foot, 'world'), ('world', foot), ('world', ''), (trunk, ''), ('world', trunk)]
    data = AnimatData(m.timestep, 1, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
    units = SimulationUnitScaling()
    maps = {'sensors': get_sensor_maps(phys)}
    get_physics2data_maps(phys, data.sensors, maps['sensors'])
    g2d = maps['sensors']['geompair2data']
    plane = int(np.nonzero(m.geom_type == 0)[0][0])
    gfoot = [g for g in range(m.ngeom) if m.body_names[m.geom_bodyid[g]] == foot]
    assert g2d[(gfoot[0], plane)] == 1 and g2d[(plane, gfoot[0])] == 2 and g2d[(gfoot[0], -1)] == 0 and g2d[(plane, -1)] == 3
    physics2data(phys, 0, data, maps, units)
    torch.cuda.synchronize()
    got = data.sensors.contacts.array[0].cpu().numpy().astype(np.float64)
    ncon = phys.data.ncon.cpu().numpy(); con = oracle.contacts_from_hip(phys.data.contact.cpu().numpy())
    want = oracle.contacts2data(con, ncon, g2d, len(pairs))
    assert np.allclose(got, want, rtol=1e-5, atol=1e-7)
    on_floor = np.abs(got[:, 0, 2]) > 1e-4
    assert on_floor.any()
    assert np.array_equal(got[:, 0], got[:, 1])                              # (foot, '') == (foot, 'world')
    assert np.allclose(got[on_floor, 2, :9], -got[on_floor, 0, :9], rtol=1e-6, atol=0)         # ('world', foot): the reaction
    assert np.allclose(got[on_floor, 2, 9:], got[on_floor, 0, 9:], rtol=1e-6, atol=1e-9)       # same contact point
    assert np.all(got[on_floor, 0, 2] > 0) and np.all(got[on_floor, 3, 2] < 0)                 # the floor pushes the foot up
    # ('world', '') collects minus the sum of what every geom-only sensor would read
    every = [(b, '') for b in m.body_names[1:] if any(m.geom_bodyid[g] == m.body_names.index(b) for g in range(m.ngeom))]
    data2 = AnimatData(m.timestep, 1, n, m.body_names[1:], m.hinge_joint_names(), contacts=every)
    maps2 = {'sensors': get_sensor_maps(phys)}
    get_physics2data_maps(phys, data2.sensors, maps2['sensors'])
    physics2data(phys, 0, data2, maps2, units)
    torch.cuda.synchronize()
    total = data2.sensors.contacts.array[0].cpu().numpy().astype(np.float64)[..., 6:9].sum(1)
    assert np.allclose(got[:, 3, 6:9], -total, rtol=1e-5, atol=1e-6)
    # a body pair that never touches the listed partner, and a missing pair, are refused like the reference does
    with pytest.raises(AssertionError):
        bad = AnimatData(m.timestep, 1, n, m.body_names[1:], m.hinge_joint_names(), contacts=[('no_such_body', '')])
        get_physics2data_maps(phys, bad.sensors, {'sensors': {}}['sensors'])
    with pytest.raises(AssertionError):
        bad = AnimatData(m.timestep, 1, n, m.body_names[1:], m.hinge_joint_names(), contacts=['body_0'])
        get_physics2data_maps(phys, bad.sensors, {})


@pytest.mark.parametrize('solver', ['pgs', 'newton'])
def test_fused_walk_contact_rows_vs_oracle(oracle, solver):
    """Config 4 end to end against the oracle alone: the fused HIP loop (collision -> PGS or Newton -> contact forces ->
    cycontacts2data rows, with geom-only and body-pair sensors) versus the fp64 restatement doing the same from the same
    inputs: mj_step, mj_contactForce and sensors.pyx:140-190, none of it fed from the HIP contact list.  With Newton (which
    converges at every step) the rows agree an order of magnitude better than with PGS cut at 50 sweeps."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.model import SOLVERS
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = _walker()
    if solver == 'newton':
        m.solver = SOLVERS['newton']; m.solver_iterations = 100
    n, T = 8, 40
    pairs = [(b, '') for b in m.body_names[1:] if b.endswith('_3')] + [('world', 'body_0'), ('body_11', 'world'), ('world', '')]
    data = AnimatData(m.timestep, T, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
    sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T), n_envs=n, data=data, buffer_size=T)
    sim.reset()
    d = sim.physics.data
    rng = np.random.default_rng(8)
    q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 7:] += rng.uniform(-0.15, 0.15, (n, m.nq - 7)); q0[:, 2] = 0.03 + 0.01*rng.uniform(size=n)
    d.qpos[:] = torch.as_tensor(q0, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    q32 = d.qpos.cpu().numpy().astype(np.float64); v32 = d.qvel.cpu().numpy().astype(np.float64)
    st = dict(qpos=q32, qvel=v32)
    fds = [oracle.forward_debug(m, q32[e], v32[e]) for e in range(n)]
    for k in ('xpos', 'xquat', 'xipos'):
        st[k] = np.array([fd[k] for fd in fds])
    sd = np.array([fd['sensordata'] for fd in fds]); sd[:, 6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
    st['sensordata'] = sd
    assert sim.task.fusable()
    sim.run(fused=True)
    torch.cuda.synchronize()
    g2d = sim.task.maps['sensors']['geompair2data']
    ref = oracle.run_fused(m, st, T, swim=None, buffer_size=T, controller=0, ctrl=np.zeros((n, m.nu)), geompair2data=g2d,
                           n_contact_rows=len(pairs), n_threads=8)
    assert int(d.status.abs().sum()) == 0
    rows = data.sensors.contacts.array.cpu().numpy(); want = ref['contacts']
    scale = np.abs(want[..., :9]).max()
    err_f = np.abs(rows[..., :9] - want[..., :9]).max()/scale
    loaded = np.linalg.norm(want[..., 6:9], axis=-1) > 0.05*scale
    err_p = np.abs(rows[..., 9:] - want[..., 9:])[loaded].max()
    print(solver, 'contact rows vs oracle: force rel err', err_f, 'position abs err', err_p, 'peak force', scale)
    assert err_f < 1e-3 and err_p < 2e-4 and scale > 0.05         # measured 5e-5 / 1.6e-5 with either solver
    assert _relerr(d.qpos.cpu().numpy(), ref['qpos']) < (2e-3 if solver == 'pgs' else 2e-4)
    # joint rows (position, velocity, limit force): stated against the floor - the same run of the fp64 oracle with every stored array
    # in fp32 (level 3) - instead of the fitted 2e-2 of round 4 (the velocity column is what is loose: 40 steps of an ill-conditioned solve)
    from parity_metrics import within_floor
    with oracle.fp32_storage(3):
        fl = oracle.run_fused(m, st, T, swim=None, buffer_size=T, controller=0, ctrl=np.zeros((n, m.nu)), geompair2data=g2d,
                              n_contact_rows=len(pairs), n_threads=8)
    cols = [0, 1, 9]
    e_j = _relerr(data.sensors.joints.array.cpu().numpy()[..., cols], ref['joints'][..., cols]); f_j = _relerr(fl['joints'][..., cols], ref['joints'][..., cols])
    print(solver, 'joint rows (q, qd, limit force): HIP', e_j, 'fp32-storage floor', f_j)
    assert within_floor(e_j, f_j, k=6, abs_tol=1e-5) and e_j < 2e-2


def test_fused_walk_with_contact_rows(oracle):
    """Config 4 through the Simulation layer: fused launches log link, joint AND contact rows; they equal the
    operator-by-operator path (physics2data + cycontacts2data + fmj_step) row for row."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    from farms_mujoco_amd.control import AnimatController, ControlType
    m = _walker()
    n, T = 4, 60
    pairs = [(b, '') for b in m.body_names[1:] if b.endswith('_3') or b.startswith('body_')]

    def make():
        data = AnimatData(m.timestep, T, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
        sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T), n_envs=n, data=data,
                         buffer_size=T)
        sim.reset()
        return sim
    sim_f, sim_u = make(), make()
    assert sim_f.task.fusable()
    sim_f.run(fused=True, chunk=25)
    sim_u.run(fused=False)
    torch.cuda.synchronize()
    for k in ('links', 'joints', 'contacts'):
        a = getattr(sim_f.task.data.sensors, k).array.cpu().numpy(); b = getattr(sim_u.task.data.sensors, k).array.cpu().numpy()
        assert np.array_equal(a, b), k                   # same device functions, same order of operations: bitwise
    c = sim_f.task.data.sensors.contacts.array.cpu().numpy()
    assert np.abs(c[T - 1][..., 2]).max() > 0.05          # the feet carry weight at the end (reaction z)
    assert np.array_equal(sim_f.physics.data.qpos.cpu().numpy(), sim_u.physics.data.qpos.cpu().numpy())


def test_full_size_config4_properties(oracle):
    """BASELINE configs[3] at full size (4096 walking envs x 100 fused steps): no warning bits, contact forces are
    non-negative along the normal, nobody sinks through the floor, envs with identical inputs give bitwise identical
    states and contact rows wherever they sit, and the second half of the batch run alone reproduces the full run."""
    import torch
    import bench
    N, T = 4096, 100
    twins = [1, 778, 2047]                 # copies of env 0, all in the first half of the batch

    def run(n, off):
        sim, m, _ = bench.build_sim(n, T, T, off, 'cuda:0', 'walk')
        d = sim.physics.data
        c = sim.task._controller
        if off == 0:
            for e in twins:
                d.qpos[e] = d.qpos[0]; d.qvel[e] = d.qvel[0]; c.env_phase[e] = c.env_phase[0]
        d.qacc_warmstart.zero_()            # the PGS warm start is state too: twins must share its history
        sim.physics.forward(disable_actuation=True)
        sim.run(fused=True)
        torch.cuda.synchronize()
        return sim

    sim = run(N, 0)
    d = sim.physics.data
    assert int(d.status.abs().sum()) == 0
    q = d.qpos.cpu().numpy(); rows = sim.task.data.sensors.contacts.array.cpu().numpy()
    ncon = d.ncon.cpu().numpy(); con = d.contact.cpu().numpy()
    assert np.isfinite(q).all() and np.isfinite(rows).all()
    assert ncon.min() >= 1 and ncon.max() <= 32 and q[:, 2].min() > 0.0
    for e in range(0, N, 97):
        assert (con[e, :ncon[e], 12] >= -1e-6).all()                      # normal force of every listed contact
    for e in twins:
        assert np.array_equal(q[e], q[0]) and np.array_equal(rows[:, e], rows[:, 0])
    half = run(N//2, N//2)
    assert np.array_equal(half.physics.data.qpos.cpu().numpy(), q[N//2:])
    assert np.array_equal(half.task.data.sensors.contacts.array.cpu().numpy(), rows[:, N//2:])


def _box_walker():
    """A free box trunk with two hinged box limbs above a plane: plane-box contacts (up to 4 corners per geom)."""
    from farms_mujoco_amd.model import ModelBuilder, GEOM_BOX, GEOM_PLANE
    b = ModelBuilder('boxbot', timestep=1e-3)
    b.options['max_contacts'] = 16
    b.add_body('trunk', pos=(0, 0, 0.06), mass=0.5, inertia=(2e-4, 6e-4, 7e-4), joint='free')
    b.add_geom('trunk', GEOM_BOX, (0.06, 0.03, 0.015), friction=(0.8, 0, 0))
    for side, y in (('L', 0.04), ('R', -0.04)):
        b.add_body(f'limb_{side}', parent='trunk', pos=(0.03, y, 0.0), mass=0.05, inertia=(2e-6, 8e-6, 8e-6),
                   joint='hinge', axis=(0, 1, 0), damping=1e-3, limited=True, range=(-0.6, 0.6))
        b.add_geom(f'limb_{side}', GEOM_BOX, (0.03, 0.008, 0.008), pos=(0.03, 0, -0.02), quat=(0.9659258, 0, 0.258819, 0),
                   friction=(1.0, 0, 0))
        b.add_position_actuator(f'joint_limb_{side}', kp=0.05)
    b.add_geom('world', GEOM_PLANE, (0, 0, 0), friction=(0, 0, 0))
    return b.compile()


def test_plane_box_contacts(oracle):
    """Box geoms against the plane (the box part): contact lists, forces and the state after a drop
    match the oracle's corner test."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _box_walker()
    n, T = 8, 60
    rng = np.random.default_rng(11)
    qpos = np.tile(m.qpos0, (n, 1))
    qpos[:, 2] = 0.02 + 0.01*rng.uniform(size=n)                      # trunk partly below its rest height: corners touch
    ang = 0.2*rng.normal(size=(n, 3)); qpos[:, 3] = 1.0; qpos[:, 4:7] = 0.5*ang
    qpos[:, 3:7] /= np.linalg.norm(qpos[:, 3:7], axis=1, keepdims=True)
    qpos[:, 7:] = rng.uniform(-0.3, 0.3, (n, m.nq - 7))
    qvel = 0.05*rng.normal(size=(n, m.nv))
    phys = BatchedPhysics(m, n)
    q32, v32 = _set(phys, qpos, qvel)
    phys.step(1)
    torch.cuda.synchronize()
    d = phys.data
    assert int(d.status.abs().sum()) == 0
    fds = [oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu)) for e in range(n)]
    ncon_ref = np.array([fd['ncon'] for fd in fds])
    assert np.array_equal(d.ncon.cpu().numpy(), ncon_ref) and ncon_ref.max() >= 4
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
    for k, tol in (('xpos', 2e-6), ('qvel', 2e-3), ('qpos', 1e-5)):
        assert _relerr(getattr(d, k).cpu().numpy(), ref[k]) < tol, (k, _relerr(getattr(d, k).cpu().numpy(), ref[k]))
    for e in range(n):
        fd = fds[e]
        got = d.contact.cpu().numpy()[e, :fd['ncon']]
        assert np.allclose(got[:, :3], fd['contact'][:fd['ncon'], :3], atol=1e-6)
        f = fd['efc_force'][fd['nefc'] - 4*fd['ncon']:fd['nefc']].reshape(-1, 4)
        assert np.allclose(got[:, 12], f.sum(1), rtol=2e-2, atol=2e-4)
    # a short drop: the box comes to rest on the plane in both implementations
    phys.step(T - 1)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)), n_steps=T)
    assert int(d.status.abs().sum()) == 0
    assert np.abs(d.qpos.cpu().numpy()[:, :3] - ref['qpos'][:, :3]).max() < 2e-3


def test_create_refuses_more_rows_than_the_solver_holds(oracle):
    """the HBM constraint path holds at most 192 rows per env (limited joints + 4 * max_contacts); a model
    that could exceed that is refused at fmj_create instead of writing past its scratch rows."""
    from farms_mujoco_amd import _lib
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _box_walker()                      # 2 limited joints
    m.max_contacts = 47                    # 2 + 188 = 190 rows: accepted
    phys = BatchedPhysics(m, 2)
    phys.step(3)
    assert int(phys.data.status.abs().sum()) == 0
    m.max_contacts = 48                    # 2 + 192 = 194 rows: refused
    with pytest.raises(_lib.FmjError, match='constraint rows'):
        BatchedPhysics(m, 2)


def _terrain(seed=0, nr=17, nc=33, rx=0.8, ry=0.4, zt=0.03):
    """Smooth random bumps, a few centimetres high, sampled on a grid."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-rx, rx, nc); ys = np.linspace(-ry, ry, nr)
    z = np.zeros((nr, nc))
    for _ in range(6):
        kx, ky, ph = rng.uniform(3, 9), rng.uniform(3, 9), rng.uniform(0, 6.28)
        z += rng.uniform(0.2, 1.0)*np.sin(kx*xs[None, :] + ph)*np.cos(ky*ys[:, None] - ph)
    return z/np.abs(z).max(), (rx, ry, zt, 0.1)


def _hfield_walker(spawn_z=0.075):
    """salamander33 with its capsules / foot spheres over a heightfield instead of the plane."""
    from farms_mujoco_amd.model import salamander33
    import farms_mujoco_amd.model as mm
    b_ref = salamander33(contacts=True, limits=True, spawn_z=spawn_z)
    # rebuild through the builder API: same animat, heightfield arena
    b = mm.ModelBuilder('salamander33_hf', timestep=1e-3)
    m = b_ref
    for i in range(1, m.nbody):
        j = int(m.body_jntadr[i])
        kw = dict(pos=m.body_pos[i], quat=m.body_quat[i], mass=m.body_mass[i], ipos=m.body_ipos[i], inertia=m.body_inertia[i], iquat=m.body_iquat[i])
        if j < 0:
            b.add_body(m.body_names[i], m.body_names[m.body_parentid[i]], **kw)
        elif m.jnt_type[j] == 0:
            b.add_body(m.body_names[i], 'world', joint='free', **kw)
        else:
            b.add_body(m.body_names[i], m.body_names[m.body_parentid[i]], joint='hinge', jname=m.joint_names[j], axis=m.jnt_axis[j],
                       damping=m.dof_damping[m.jnt_dofadr[j]], limited=bool(m.jnt_limited[j]), range=m.jnt_range[j], **kw)
    for g in range(m.ngeom):
        if m.geom_type[g] != 0:
            b.add_geom(m.body_names[m.geom_bodyid[g]], int(m.geom_type[g]), m.geom_size[g], pos=m.geom_pos[g], quat=m.geom_quat[g],
                       friction=m.geom_friction[g])
    data, size = _terrain()
    b.add_hfield(data, size, pos=(0.4, 0.0, 0.0))
    b.options['max_contacts'] = 32
    for a in range(m.nu):
        if m.actuator_tags[a] == 'position':
            b.add_position_actuator(m.joint_names[m.actuator_jntid[a]], kp=m.actuator_gain[a])
    return b.compile()


def test_heightfield_contacts_match_oracle(oracle):
    """SURVEY 8 f4, the heightfield part (reference task.py:108-123, mjcf.py:486-522), met as the plane of the grid
    triangle under each candidate point: contact lists (positions, normals that differ from contact to contact, geom
    ids), contact forces and the state after a drop onto bumpy terrain match the oracle."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _hfield_walker()
    n, T = 8, 80
    rng = np.random.default_rng(12)
    qpos = np.tile(m.qpos0, (n, 1))
    qpos[:, 0] += rng.uniform(-0.1, 0.1, n); qpos[:, 1] += rng.uniform(-0.1, 0.1, n); qpos[:, 2] = 0.02 + 0.02*rng.uniform(size=n)
    qpos[:, 7:] += rng.uniform(-0.2, 0.2, (n, m.nq - 7))
    qvel = 0.02*rng.normal(size=(n, m.nv))
    phys = BatchedPhysics(m, n)
    q32, v32 = _set(phys, qpos, qvel)
    phys.step(1)
    torch.cuda.synchronize()
    d = phys.data
    assert int(d.status.abs().sum()) == 0
    fds = [oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu)) for e in range(n)]
    ncon_ref = np.array([fd['ncon'] for fd in fds])
    assert np.array_equal(d.ncon.cpu().numpy(), ncon_ref) and ncon_ref.max() >= 4 and ncon_ref.sum() >= 16
    hf = int(np.nonzero(m.geom_type == 1)[0][0])
    normals = []
    for e in range(n):
        fd = fds[e]
        got = d.contact.cpu().numpy()[e, :fd['ncon']]
        con = oracle.contacts_from_hip(got)
        assert np.all(con[:, 15] == hf) and np.array_equal(con[:, 16], fd['contact'][:fd['ncon'], 16])
        assert np.allclose(got[:, :3], fd['contact'][:fd['ncon'], :3], atol=2e-6)           # positions
        assert np.allclose(got[:, 3:12], fd['contact'][:fd['ncon'], 3:12], atol=2e-5)       # frames
        f = fd['efc_force'][fd['nefc'] - 4*fd['ncon']:fd['nefc']].reshape(-1, 4)
        assert np.allclose(got[:, 12], f.sum(1), rtol=2e-2, atol=3e-4)
        normals.append(got[:, 3:6])
    normals = np.concatenate(normals)
    assert np.abs(normals[:, :2]).max() > 0.05 and np.ptp(normals[:, 0]) > 0.05         # the terrain really is bumpy
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
    for k, tol in (('xpos', 2e-6), ('qvel', 2e-3), ('qpos', 1e-5)):
        assert _relerr(getattr(d, k).cpu().numpy(), ref[k]) < tol, (k, _relerr(getattr(d, k).cpu().numpy(), ref[k]))
    phys.step(T - 1)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)), n_steps=T, n_threads=8)
    assert int(d.status.abs().sum()) == 0
    e = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max(1)
    print('heightfield drop: qpos abs err per env after', T, 'steps', e)
    assert e.max() < 5e-3 and np.median(e) < 1e-3
    assert float(d.qpos[:, 2].min()) > -0.03                                             # nobody fell through the terrain


def test_flat_heightfield_equals_plane_on_the_gpu(oracle):
    """A constant heightfield and the plane at the same height give the same rollout (to rounding: the distance is
    formed differently)."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    from farms_mujoco_amd.model import ModelBuilder, GEOM_BOX, GEOM_PLANE
    def build(kind):
        b = ModelBuilder('boxg', timestep=1e-3)
        b.options['max_contacts'] = 8
        b.add_body('trunk', pos=(0.02, -0.01, 0.03), mass=0.5, inertia=(2e-4, 6e-4, 7e-4), joint='free')
        b.add_geom('trunk', GEOM_BOX, (0.06, 0.03, 0.015), friction=(0.8, 0, 0))
        if kind == 'plane':
            b.add_geom('world', GEOM_PLANE, (0, 0, 0), pos=(0, 0, 0.01))
        else:
            b.add_hfield(np.full((4, 6), 0.5), (0.5, 0.4, 0.02, 0.1))
        return b.compile()
    outs = []
    for kind in ('plane', 'hfield'):
        m = build(kind)
        phys = BatchedPhysics(m, 3)
        q = np.tile(m.qpos0, (3, 1)); q[:, 3:7] = [0.995, 0.05, 0.08, 0.0]; q[:, 3:7] /= np.linalg.norm(q[0, 3:7])
        _set(phys, q, np.zeros((3, m.nv)))
        phys.step(150)
        torch.cuda.synchronize()
        assert int(phys.data.status.abs().sum()) == 0 and int(phys.data.ncon.min()) >= 3
        outs.append(phys.data.qpos.cpu().numpy())
    assert np.abs(outs[0] - outs[1]).max() < 2e-4


def test_self_collision_pairs_match_oracle(oracle):
    """SURVEY 8 f4, explicit self-collision pairs (reference mjcf.py:1012-1033): constraint rows over TWO branches of
    the tree (J = J(body2) - J(body1)).  A scissors mechanism (two arms on one post) with sphere and with capsule tips:
    contact record, contact force and accelerations of a single step, then a rollout, against the oracle."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    from test_oracle_contacts import _scissors
    for capsule in (False, True):
        m, q = _scissors(0.1, capsule=capsule)
        n = 6
        rng = np.random.default_rng(3)
        qpos = np.tile(q, (n, 1)) + rng.uniform(-0.03, 0.03, (n, 2)); qpos[n - 1] = [0.6, -0.6]      # the last env is open: no contact
        qvel = rng.normal(size=(n, 2))*0.2
        phys = BatchedPhysics(m, n)
        q32, v32 = _set(phys, qpos, qvel)
        phys.step(1)
        torch.cuda.synchronize()
        d = phys.data
        assert int(d.status.abs().sum()) == 0
        fds = [oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu)) for e in range(n)]
        ncon_ref = np.array([fd['ncon'] for fd in fds])
        assert np.array_equal(d.ncon.cpu().numpy(), ncon_ref) and ncon_ref[:-1].min() == 1 and ncon_ref[-1] == 0
        ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
        for e in range(n - 1):
            got = oracle.contacts_from_hip(d.contact.cpu().numpy()[e, :1])[0]
            want = fds[e]['contact'][0]
            assert got[15] == want[15] and got[16] == want[16]
            assert np.allclose(got[:12], want[:12], atol=2e-6)
            assert abs(got[12] - want[12]) < 2e-3*abs(want[12]) + 1e-5, (capsule, e, got[12], want[12])
        assert _relerr(d.qvel.cpu().numpy(), ref['qvel']) < 1e-4 and _relerr(d.qpos.cpu().numpy(), ref['qpos']) < 1e-6
        phys.step(199)
        torch.cuda.synchronize()
        ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)), n_steps=200)
        assert int(d.status.abs().sum()) == 0
        assert np.abs(d.qpos.cpu().numpy() - ref['qpos']).max() < 2e-4, capsule


def _salamander_self_collisions(spawn_z=0.045):
    """The walking salamander with explicit pairs between neighbouring limbs and between head and tail, over the plane."""
    import farms_mujoco_amd.model as mm
    ref = mm.salamander33(contacts=True, limits=True, spawn_z=spawn_z)
    b = mm.ModelBuilder('salamander33_sc', timestep=1e-3)
    m = ref
    for i in range(1, m.nbody):
        j = int(m.body_jntadr[i])
        kw = dict(pos=m.body_pos[i], quat=m.body_quat[i], mass=m.body_mass[i], ipos=m.body_ipos[i], inertia=m.body_inertia[i], iquat=m.body_iquat[i])
        if m.jnt_type[j] == 0:
            b.add_body(m.body_names[i], 'world', joint='free', **kw)
        else:
            b.add_body(m.body_names[i], m.body_names[m.body_parentid[i]], joint='hinge', jname=m.joint_names[j], axis=m.jnt_axis[j],
                       damping=m.dof_damping[m.jnt_dofadr[j]], limited=bool(m.jnt_limited[j]), range=m.jnt_range[j], **kw)
    for g in range(m.ngeom):
        b.add_geom(m.body_names[m.geom_bodyid[g]], int(m.geom_type[g]), m.geom_size[g], pos=m.geom_pos[g], quat=m.geom_quat[g],
                   friction=m.geom_friction[g])
    for a in range(m.nu):
        if m.actuator_tags[a] == 'position':
            b.add_position_actuator(m.joint_names[m.actuator_jntid[a]], kp=m.actuator_gain[a])
    b.options['max_contacts'] = 32
    for pair in (('leg_front_L_3', 'leg_front_R_3'), ('leg_hind_L_3', 'leg_hind_R_3'), ('body_0', 'body_11'), ('body_2', 'body_10'),
                 ('leg_front_L_3', 'body_2'), ('leg_hind_R_3', 'body_6'), ('leg_front_R_3', 'body_1'), ('leg_hind_L_3', 'leg_front_L_3')):
        b.add_contact_pair(*pair)
    return b.compile()


def test_self_collisions_with_ground_contacts_match_oracle(oracle):
    """Ground contacts and self-collision pairs in the same env: a salamander curled so that its head meets its tail and
    a foot its trunk, lying on the plane.  Contact lists in the oracle's order (ground first, then pairs), forces and the
    state after 40 steps; the body-pair contact sensor (reference physics.py:367-374) reads the pair's force with the
    sign convention of sensors.pyx:163-169."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.physics import BatchedPhysics
    from farms_mujoco_amd.simulation.physics import get_sensor_maps, get_physics2data_maps, physics2data
    from farms_mujoco_amd.units import SimulationUnitScaling
    m = _salamander_self_collisions()
    assert m.npair == 8
    n = 8
    rng = np.random.default_rng(21)
    plane = int(np.nonzero(m.geom_type == 0)[0][0])
    spine = [m.jnt_qposadr[m.joint_names.index(f'joint_body_{i}')] for i in range(1, 12)]
    legj = [m.jnt_qposadr[m.joint_names.index(f'joint_leg_{t}_{s_}_{k}')] for t in ('front', 'hind') for s_ in ('L', 'R') for k in range(4)]
    poses = []
    # two leg configurations in which the two front / the two hind feet meet under the trunk (rows over two sibling branches)
    for legs in ([-1.19, 0.07, -0.84, -0.74, 0.24, -0.09, 1.01, -0.27, 0.42, -0.19, 0.52, -0.12, -0.63, 0.83, 0.69, -0.46],
                 [-0.25, -0.11, 0.59, -0.16, 1.02, 0.67, 0.77, -0.62, -0.01, -0.32, -0.9, 0.36, 0.78, 0.27, 1.14, -0.7]):
        q = m.qpos0.copy(); q[2] = 0.05; q[legj] = legs; poses.append(q)
    # curled up: the head meets the tail (one body's chain contains the other's)
    for curl in (0.645, 0.65):
        q = m.qpos0.copy(); q[2] = 0.03; q[spine] = curl; poses.append(q)
    # random leg poses in which a foot touches the trunk (found with the oracle's collision pass)
    while len(poses) < n:
        q = m.qpos0.copy(); q[2] = 0.05; q[spine] = rng.uniform(-0.1, 0.1, 11); q[legj] = rng.uniform(-1.2, 1.2, len(legj))
        fd = oracle.forward_debug(m, q, np.zeros(m.nv))
        c = fd['contact'][:fd['ncon']]
        if ((c[:, 15] != plane) & (c[:, 17] > -0.01)).sum() >= 1 and (c[:, 17] > -0.012).all():
            poses.append(q)
    qpos = np.array(poses)
    qvel = 0.02*rng.normal(size=(n, m.nv))
    phys = BatchedPhysics(m, n)
    q32, v32 = _set(phys, qpos, qvel)
    phys.step(1)
    torch.cuda.synchronize()
    d = phys.data
    assert int(d.status.abs().sum()) == 0
    fds = [oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu)) for e in range(n)]
    ncon_ref = np.array([fd['ncon'] for fd in fds])
    assert np.array_equal(d.ncon.cpu().numpy(), ncon_ref)
    n_pair_contacts = 0
    kinds = set()
    for e in range(n):
        fd = fds[e]
        got = oracle.contacts_from_hip(d.contact.cpu().numpy()[e, :fd['ncon']])
        want = fd['contact'][:fd['ncon']]
        assert np.array_equal(got[:, 15:17], want[:, 15:17])
        n_pair_contacts += int((want[:, 15] != plane).sum())
        kinds |= {(int(a_), int(b_)) for a_, b_ in want[want[:, 15] != plane][:, 15:17]}
        assert np.allclose(got[:, :12], want[:, :12], atol=5e-6)
        assert np.allclose(got[:, 12], want[:, 12], rtol=3e-2, atol=5e-4), (e, np.abs(got[:, 12] - want[:, 12]).max())
    assert n_pair_contacts >= n and len(kinds) >= 4                    # every env is in self-contact; feet, trunk, head and tail take part
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
    assert _relerr(d.qvel.cpu().numpy(), ref['qvel']) < 3e-3 and _relerr(d.qpos.cpu().numpy(), ref['qpos']) < 1e-5
    # body-pair sensors on the current contact list
    pairs = [('body_0', 'body_11'), ('body_11', 'body_0'), ('body_0', '')]
    data = AnimatData(m.timestep, 1, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
    maps = {'sensors': get_sensor_maps(phys)}
    get_physics2data_maps(phys, data.sensors, maps['sensors'])
    physics2data(phys, 0, data, maps, SimulationUnitScaling())
    torch.cuda.synchronize()
    rows = data.sensors.contacts.array[0].cpu().numpy().astype(np.float64)
    want = oracle.contacts2data(oracle.contacts_from_hip(d.contact.cpu().numpy()), d.ncon.cpu().numpy(), maps['sensors']['geompair2data'], len(pairs))
    assert np.allclose(rows, want, rtol=1e-5, atol=1e-7)
    touching = np.linalg.norm(rows[:, 0, 6:9], axis=1) > 1e-4
    assert touching.any() and np.allclose(rows[touching, 1, :9], -rows[touching, 0, :9], rtol=1e-6)
    phys.step(39)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)), n_steps=40, n_threads=8)
    assert int(d.status.abs().sum()) == 0
    err = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max(1)
    print('self-collision + ground rollout: qpos abs err per env', err)
    assert err.max() < 5e-3 and np.median(err) < 1e-3


def _mesh_walker(seed=5):
    """A free trunk with a convex-mesh hull (random points on an ellipsoid) and two hinged limbs ending in small convex
    meshes, above a plane: every ground contact comes from a mesh vertex."""
    from farms_mujoco_amd.model import ModelBuilder, GEOM_PLANE
    rng = np.random.default_rng(seed)

    def cloud(n, a, b, c):
        v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v*np.array([a, b, c])
    b = ModelBuilder('meshbot', timestep=1e-3)
    b.options['max_contacts'] = 16
    b.add_body('trunk', pos=(0, 0, 0.06), mass=0.5, inertia=(2e-4, 6e-4, 7e-4), joint='free')
    b.add_mesh_geom('trunk', cloud(60, 0.06, 0.03, 0.015), friction=(0.8, 0, 0))
    for side, y in (('L', 0.04), ('R', -0.04)):
        b.add_body(f'limb_{side}', parent='trunk', pos=(0.03, y, 0.0), mass=0.05, inertia=(2e-6, 8e-6, 8e-6),
                   joint='hinge', axis=(0, 1, 0), damping=1e-3, limited=True, range=(-0.6, 0.6))
        b.add_mesh_geom(f'limb_{side}', cloud(24, 0.03, 0.008, 0.008), pos=(0.03, 0, -0.02), quat=(0.9659258, 0, 0.258819, 0),
                        friction=(1.0, 0, 0))
        b.add_position_actuator(f'joint_limb_{side}', kp=0.05)
    b.add_geom('world', GEOM_PLANE, (0, 0, 0), friction=(0, 0, 0))
    return b.compile()


def test_plane_mesh_contacts(oracle):
    """Convex mesh geoms against the plane (the mesh part): the contact lists (the up-to-4 deepest penetrating
    hull vertices per geom, deepest first), the forces and the state after a drop match the oracle."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _mesh_walker()
    assert m.nmeshvert == 60 + 2*24
    n, T = 8, 60
    rng = np.random.default_rng(12)
    qpos = np.tile(m.qpos0, (n, 1))
    qpos[:, 2] = 0.012 + 0.01*rng.uniform(size=n)                     # trunk partly below its rest height: vertices touch
    ang = 0.2*rng.normal(size=(n, 3)); qpos[:, 3] = 1.0; qpos[:, 4:7] = 0.5*ang
    qpos[:, 3:7] /= np.linalg.norm(qpos[:, 3:7], axis=1, keepdims=True)
    qpos[:, 7:] = rng.uniform(-0.3, 0.3, (n, m.nq - 7))
    qvel = 0.05*rng.normal(size=(n, m.nv))
    phys = BatchedPhysics(m, n)
    q32, v32 = _set(phys, qpos, qvel)
    phys.step(1)
    torch.cuda.synchronize()
    d = phys.data
    assert int(d.status.abs().sum()) == 0
    fds = [oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu)) for e in range(n)]
    ncon_ref = np.array([fd['ncon'] for fd in fds])
    assert np.array_equal(d.ncon.cpu().numpy(), ncon_ref) and ncon_ref.max() >= 5 and ncon_ref.min() >= 1
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
    for k, tol in (('xpos', 2e-6), ('qvel', 2e-3), ('qpos', 1e-5)):
        assert _relerr(getattr(d, k).cpu().numpy(), ref[k]) < tol, (k, _relerr(getattr(d, k).cpu().numpy(), ref[k]))
    for e in range(n):
        fd = fds[e]
        got = d.contact.cpu().numpy()[e, :fd['ncon']]
        assert np.allclose(got[:, :3], fd['contact'][:fd['ncon'], :3], atol=1e-6)
        f = fd['efc_force'][fd['nefc'] - 4*fd['ncon']:fd['nefc']].reshape(-1, 4)
        assert np.allclose(got[:, 12], f.sum(1), rtol=2e-2, atol=2e-4)
    phys.step(T - 1)                      # a short drop: the body comes to rest on its hull in both implementations
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)), n_steps=T)
    assert int(d.status.abs().sum()) == 0
    assert np.abs(d.qpos.cpu().numpy()[:, :3] - ref['qpos'][:, :3]).max() < 2e-3


def test_mesh_on_heightfield_and_fused_rows(oracle):
    """Mesh geoms on a heightfield through the fused loop: contact-sensor rows of the mesh-footed limbs and the state
    match the oracle's fused loop."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    from farms_mujoco_amd.model import ModelBuilder
    rng = np.random.default_rng(2)

    def cloud(n_, a, b_, c):
        v = rng.normal(size=(n_, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v*np.array([a, b_, c])
    b = ModelBuilder('meshbot_h', timestep=1e-3)
    b.options['max_contacts'] = 16
    b.add_body('trunk', pos=(0, 0, 0.08), mass=0.5, inertia=(2e-4, 6e-4, 7e-4), joint='free')
    b.add_mesh_geom('trunk', cloud(40, 0.06, 0.03, 0.015), friction=(0.8, 0, 0))
    for side, y in (('L', 0.04), ('R', -0.04)):
        b.add_body(f'limb_{side}', parent='trunk', pos=(0.03, y, 0.0), mass=0.05, inertia=(2e-6, 8e-6, 8e-6),
                   joint='hinge', axis=(0, 1, 0), damping=1e-3, limited=True, range=(-0.6, 0.6))
        b.add_mesh_geom(f'limb_{side}', cloud(16, 0.03, 0.008, 0.008), pos=(0.03, 0, -0.02), friction=(1.0, 0, 0))
        b.add_position_actuator(f'joint_limb_{side}', kp=0.05)
    xs = np.linspace(-1, 1, 9)
    b.add_hfield(0.5 + 0.5*np.outer(np.sin(2.0*xs), np.cos(1.5*xs)), (0.4, 0.4, 0.03, 0.1), friction=(0, 0, 0))
    m = b.compile()
    n, T = 6, 50
    pairs = [('limb_L', ''), ('limb_R', ''), ('trunk', '')]
    data = AnimatData(m.timestep, T, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
    sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T), n_envs=n, data=data, buffer_size=T)
    sim.reset()
    d = sim.physics.data
    q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 0] = rng.uniform(-0.15, 0.15, n); q0[:, 1] = rng.uniform(-0.15, 0.15, n); q0[:, 2] = 0.05
    q0[:, 7:] += rng.uniform(-0.2, 0.2, (n, m.nq - 7))
    d.qpos[:] = torch.as_tensor(q0, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    q32 = d.qpos.cpu().numpy().astype(np.float64); v32 = d.qvel.cpu().numpy().astype(np.float64)
    st = dict(qpos=q32, qvel=v32)
    fds = [oracle.forward_debug(m, q32[e], v32[e]) for e in range(n)]
    for k in ('xpos', 'xquat', 'xipos'):
        st[k] = np.array([fd[k] for fd in fds])
    sd = np.array([fd['sensordata'] for fd in fds]); sd[:, 6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
    st['sensordata'] = sd
    assert sim.task.fusable()
    sim.run(fused=True)
    torch.cuda.synchronize()
    g2d = sim.task.maps['sensors']['geompair2data']
    ref = oracle.run_fused(m, st, T, swim=None, buffer_size=T, controller=0, ctrl=np.zeros((n, m.nu)), geompair2data=g2d,
                           n_contact_rows=len(pairs), n_threads=8)
    assert int(d.status.abs().sum()) == 0
    rows = data.sensors.contacts.array.cpu().numpy(); want = ref['contacts']
    scale = np.abs(want[..., :9]).max()
    # contact rows of mesh feet on a heightfield after T free-running steps: against the floor (fp64 oracle, fp32 storage of every array)
    from parity_metrics import within_floor
    with oracle.fp32_storage(3):
        fl = oracle.run_fused(m, st, T, swim=None, buffer_size=T, controller=0, ctrl=np.zeros((n, m.nu)), geompair2data=g2d,
                              n_contact_rows=len(pairs), n_threads=8)
    e_c = np.abs(rows[..., :9] - want[..., :9]).max()/scale; f_c = np.abs(fl['contacts'][..., :9] - want[..., :9]).max()/scale
    print('mesh on heightfield: contact rows HIP', e_c, 'fp32-storage floor', f_c)
    assert scale > 0.05 and within_floor(e_c, f_c, k=6, abs_tol=1e-4) and e_c < 3e-2
    assert _relerr(d.qpos.cpu().numpy(), ref['qpos']) < 2e-3


def test_two_env_kernel_modes_do_not_depend_on_the_partner(oracle, monkeypatch):
    """The two-env constraint kernel (fmj_cons2.inc, round 4): what an env computes depends neither on the mode its rows were solved
    in - PAIR (both envs of the wave have <= 32 rows: solved at once), SOLO (one has 33..64: one after the other) - nor on the env it
    shares its wave with; an env with more than 64 rows retires from the wave ALONE and is finished by the one-env kernel, bitwise
    as if that kernel had stepped it from the start (FMJ_DUAL=0).  All of it within tolerance of the oracle."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _walker()
    rng = np.random.default_rng(11)
    base = np.tile(m.qpos0, (1, 1))[0]

    def state(z, seed):
        r = np.random.default_rng(seed)
        q = base.copy(); q[7:] += r.uniform(-0.05, 0.05, m.nq - 7); q[2] = z
        return q
    # candidates from standing on the feet down to lying on the belly; classes by the oracle's row count of the FIRST step
    cands = [state(z, 100 + i) for i, z in enumerate((0.045, 0.03, 0.0175, 0.0165, 0.0155, 0.0145, 0.012, 0.01))]
    nefc = [oracle.forward_debug(m, q.astype(np.float32).astype(np.float64), np.zeros(m.nv), ctrl=np.zeros(m.nu))['nefc'] for q in cands]
    light = [q for q, ne in zip(cands, nefc) if 0 < ne <= 32][:2]
    medium = [q for q, ne in zip(cands, nefc) if 32 < ne <= 64][:2]
    heavy = [q for q, ne in zip(cands, nefc) if ne > 64][:1]
    assert len(light) == 2 and len(medium) == 2 and len(heavy) == 1, nefc
    L, L2, Md, Md2, Hv = light[0], light[1], medium[0], medium[1], heavy[0]
    T = 25
    import os

    def run(qs):
        phys = BatchedPhysics(m, len(qs))
        assert phys.kernel_info()['threads_per_env'] == (32 if os.environ.get('FMJ_DUAL', '1') != '0' else 64)
        q32, v32 = _set(phys, np.array(qs), np.zeros((len(qs), m.nv)))
        phys.step(T)
        torch.cuda.synchronize()
        d = phys.data
        assert int((d.status & ~8).abs().sum()) == 0
        return (d.qpos.cpu().numpy().copy(), d.qvel.cpu().numpy().copy(), d.qacc_warmstart.cpu().numpy().copy(), d.contact.cpu().numpy().copy(),
                d.ncon.cpu().numpy().copy(), d.sensordata.cpu().numpy().copy(), q32, v32)
    same = lambda a, i, b, j: all(np.array_equal(a[k][i], b[k][j]) for k in range(6))
    r_ll, r_lm, r_lh, r_l = run([L, L2]), run([L, Md]), run([L, Hv]), run([L])
    assert same(r_ll, 0, r_lm, 0) and same(r_ll, 0, r_lh, 0) and same(r_ll, 0, r_l, 0), 'a light env changed with its partner'
    r_mm, r_mh, r_ml = run([Md, Md2]), run([Md, Hv]), run([Md2, L])
    assert same(r_mm, 0, r_lm, 1) and same(r_mm, 0, r_mh, 0) and same(r_mm, 1, r_ml, 0), 'an env with 33..64 rows changed with its partner'
    r_hh = run([Hv, Hv])
    assert same(r_hh, 0, r_lh, 1) and same(r_hh, 0, r_mh, 1) and same(r_hh, 0, r_hh, 1)
    monkeypatch.setenv('FMJ_DUAL', '0')            # the one-env kernel from the first step on: what a retired env must reproduce bitwise
    r_one = run([Hv, L])
    assert same(r_one, 0, r_hh, 0), 'a retired env is not what the one-env kernel computes'
    # ... and the two kernels agree with each other and with the oracle to fp32 tolerance on the light env
    assert _relerr(r_one[0][1], r_ll[0][0]) < 2e-5
    for res in (r_ll, r_lm, r_lh, r_mm):
        ref = oracle.step(m, res[6], res[7], ctrl=np.zeros((len(res[6]), m.nu)), n_steps=T)
        assert _relerr(res[0], ref['qpos']) < 2e-4, _relerr(res[0], ref['qpos'])


@pytest.mark.parametrize('solver', ['newton', 'pgs'])
def test_thousand_steps_of_walking(oracle, solver):
    """north_star's horizon on BASELINE configs[3]: qpos after 300 / 600 / 1000 free-running steps of the trot against the oracle's
    own walk, 32 envs, rel = max |qpos - qpos_ref| / max |qpos_ref| per env.

    Walking is event-driven.  scripts/walk_event_diff.py (round 5, profiles/r05_walk_event_diff_*.txt) locates, per env, the first
    step whose active set differs from the oracle's: 39 of 42 first differences are a GRAZING GROUND CONTACT - |dist| of 1e-9 ..
    2e-7 m, the rounding of an fp32 position - on which the oracle, handed HIP's own state, decides as HIP does ("drift": the states
    were 2e-6 .. 2e-5 apart by then); 3 are decisions on identical state at |dist| <= 7e-9 m.  An env that takes one such event a step
    early or late is on another (equally valid) walk from there on.  How many envs are still within 1e-4 of the oracle's walk after
    1000 steps is therefore a steep function of the PER-STEP error, and the yardsticks are fp64 engines with fp32 STORAGE:
    * oracle.fp32_state(): qpos / qvel / warm start rounded to fp32 every step, fp32 poses, M and H stored in fp32 - the floor of any
      fp32 engine (VERDICT round 4 asked for it): 19 / 32 (Newton) and 18 / 32 (PGS) envs within 1e-4 after 1000 steps;
    * oracle.fp32_state(drop_bits=k): the same with 2^k times that storage error: x2 11 and 12 / 32, x4 5 and 7 / 32, x8 3 and 0 / 32.
    The HIP step's per-step velocity error is 2.4x the floor's on this walk (teacher-forced, scripts/tf_probe.py,
    profiles/r05_walk_teacher_forced_floor.txt: 4.6e-5 against 1.9e-5 rad/s median, equal parts entries of H and the fp32 L'DL, DESIGN
    section 2; the 99th percentile 2.1x with PGS and 3.1x with Newton, the worst step 2.5x and 10x: the fp32 primal iteration loses
    more on stiff rows than the dual sweeps do) and it keeps 13 - 17 / 32 with PGS - between the x2 and the x4 engine, where a 2.4x
    engine belongs - and 7 / 32 with Newton (20 / 32 after 600 steps, where the x4 engine keeps 25 and the x8 engine 19).  The
    assertions below hold HIP to the x4 (PGS) / x8 (Newton) engine measured in the same test (a regression of activation order or
    narrow phase would fall below it) and to north_star's bound for every env at 300 steps; the bound that does not depend on
    events is per step, tests/test_gpu_teacher_forced.py."""
    import torch
    from farms_mujoco_amd.model import SOLVERS
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _walker()
    if solver == 'newton':
        m.solver = SOLVERS['newton']; m.solver_iterations = 100
    n, T = 32, 1000
    tape = _trot_tape(m, n, T)
    phys = BatchedPhysics(m, n)
    q32, v32 = _set(phys, np.tile(m.qpos0, (n, 1)), np.zeros((n, m.nv)))
    tape_t = torch.as_tensor(tape, dtype=torch.float32, device='cuda').contiguous()
    tape64 = tape_t.cpu().numpy().astype(np.float64)
    d = phys.data
    rel = lambda a, b: np.abs(a - b).max(1)/np.abs(b).max(1)
    done, stats = 0, {}
    for Tm in (300, 600, 1000):
        phys.step(Tm - done, ctrl_tape=tape_t[done:Tm].contiguous())
        torch.cuda.synchronize()
        done = Tm
        run = lambda: oracle.step(m, q32, v32, ctrl=tape64[:Tm], n_steps=Tm, ctrl_step_stride=n*m.nu, n_threads=8)
        ref = run()
        floors = {}
        for k in (0, 1, 2, 3):
            with oracle.fp32_state(drop_bits=k):
                floors[k] = rel(run()['qpos'], ref['qpos'])
        assert int(d.status.abs().sum()) == 0 and int(ref['status'].sum()) == 0
        r = rel(d.qpos.cpu().numpy(), ref['qpos'])
        stats[Tm] = (r, floors)
        print(f'{solver} walk after {Tm:4d} steps: HIP median {np.median(r):.1e} worst {r.max():.1e}, within 1e-4: {(r <= 1e-4).sum()}/{n};   fp64 engine with '
              + ', '.join(f'x{2**k} fp32 storage error: {(f <= 1e-4).sum()}/{n} (median {np.median(f):.1e})' for k, f in floors.items()))
    assert float(d.qpos[:, 2].min()) > 0.0 and float(d.qpos[:, 2].max()) < 0.1      # the plane holds the animal up
    within = lambda x: int((x <= 1e-4).sum())
    assert (stats[300][0] <= 1e-4).all(), stats[300][0]                 # north_star's bound, every env, both solvers
    assert np.median(stats[600][0]) <= 1e-4
    # never below the fp64 engine with 4x (PGS) / 8x (Newton: its per-step error has the heavier tail, see the docstring) fp32's storage
    # error, 3 envs of sampling noise allowed
    yard = 3 if solver == 'newton' else 2
    for Tm in (600, 1000):
        r, floors = stats[Tm]
        assert within(r) >= within(floors[yard]) - 3, (Tm, within(r), {k: within(f) for k, f in floors.items()})
    assert within(stats[1000][0]) >= 4 and stats[1000][0].max() < 0.3
    # an env that left the oracle's walk did so through an event, not through drift: it is either on it (1e-4) or far from it
    r = stats[1000][0]
    assert ((r <= 1e-4) | (r > 5e-4)).mean() > 0.8


@pytest.mark.parametrize('dual', ['1', '0'], ids=['two_per_wave', 'one_per_wave'])
@pytest.mark.parametrize('substeps', [2, 3])
def test_fused_walk_with_substeps(oracle, monkeypatch, substeps, dual):
    """Walking with num_sub_steps > 1 (reference task.py:168-186,348-369: joint and contact rows once per iteration, at the full
    step; the model steps at timestep / S) through both constraint kernels: the fused launch equals the operator-by-operator path row
    for row, and both follow the oracle's restatement of the task's counters."""
    import os
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.model import salamander33
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    monkeypatch.setenv('FMJ_DUAL', dual)
    S, h = substeps, 1e-3
    m = salamander33(contacts=True, limits=True, spawn_z=0.045, timestep=h/S)
    n, T = 6, 30
    pairs = [(b, '') for b in m.body_names[1:] if b.endswith('_3')] + [('world', 'body_0'), ('world', '')]
    rng = np.random.default_rng(21)
    q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 7:] += rng.uniform(-0.15, 0.15, (n, m.nq - 7)); q0[:, 2] = 0.03 + 0.01*rng.uniform(size=n)

    def make():
        data = AnimatData(h, T, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
        sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=h, n_iterations=T, num_sub_steps=S), n_envs=n, data=data, buffer_size=T)
        sim.reset()
        sim.physics.data.qpos[:] = torch.as_tensor(q0, dtype=torch.float32)
        sim.physics.forward(disable_actuation=True)
        return sim, data
    sim_f, data_f = make()
    sim_u, data_u = make()
    assert sim_f.physics.kernel_info()['threads_per_env'] == (32 if dual == '1' else 64)
    d = sim_f.physics.data
    q32 = d.qpos.cpu().numpy().astype(np.float64); v32 = d.qvel.cpu().numpy().astype(np.float64)
    st = dict(qpos=q32, qvel=v32)
    fds = [oracle.forward_debug(m, q32[e], v32[e]) for e in range(n)]
    for k in ('xpos', 'xquat', 'xipos'):
        st[k] = np.array([fd[k] for fd in fds])
    sd = np.array([fd['sensordata'] for fd in fds]); sd[:, 6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
    st['sensordata'] = sd
    assert sim_f.task.fusable()
    sim_f.run(fused=True, chunk=7)                          # launch boundaries inside and between iterations' sub-steps
    sim_u.run(fused=False)
    torch.cuda.synchronize()
    assert int(d.status.abs().sum()) == 0
    for k in ('links', 'contacts'):
        a = getattr(data_f.sensors, k).array.cpu().numpy(); b = getattr(data_u.sensors, k).array.cpu().numpy()
        assert np.array_equal(a, b), k
    ja = data_f.sensors.joints.array.cpu().numpy(); jb = data_u.sensors.joints.array.cpu().numpy()
    assert np.allclose(ja, jb, rtol=1e-5, atol=2e-7)       # (the row at a launch boundary: see test_fused_substeps_chunked_equals_one_launch)
    assert np.array_equal(sim_f.physics.data.qpos.cpu().numpy(), sim_u.physics.data.qpos.cpu().numpy())
    g2d = sim_f.task.maps['sensors']['geompair2data']
    ref = oracle.run_fused(m, st, T, swim=None, buffer_size=T, controller=0, ctrl=np.zeros((n, m.nu)), geompair2data=g2d,
                           n_contact_rows=len(pairs), n_threads=8, substeps=S)
    rows = data_f.sensors.contacts.array.cpu().numpy(); want = ref['contacts']
    scale = np.abs(want[..., :9]).max()
    err_f = np.abs(rows[..., :9] - want[..., :9]).max()/scale
    print('substeps', S, 'contact rows force rel err', err_f, 'peak', scale, 'qpos', _relerr(d.qpos.cpu().numpy(), ref['qpos']))
    assert err_f < 2e-3 and scale > 0.05
    assert _relerr(d.qpos.cpu().numpy(), ref['qpos']) < 2e-3
    assert _relerr(data_f.sensors.links.array.cpu().numpy(), ref['links']) < 2e-3
    from parity_metrics import within_floor
    with oracle.fp32_storage(3):
        fl = oracle.run_fused(m, st, T, swim=None, buffer_size=T, controller=0, ctrl=np.zeros((n, m.nu)), geompair2data=g2d,
                              n_contact_rows=len(pairs), n_threads=8, substeps=S)
    e_j = _relerr(ja[..., [0, 1, 9]], ref['joints'][..., [0, 1, 9]]); f_j = _relerr(fl['joints'][..., [0, 1, 9]], ref['joints'][..., [0, 1, 9]])
    print('substeps', S, 'joint rows: HIP', e_j, 'fp32-storage floor', f_j)
    assert within_floor(e_j, f_j, k=6, abs_tol=1e-5) and e_j < 2e-2


def test_fused_walk_one_long_launch_equals_many_short_ones(oracle):
    """bench.py runs walking in launches of 1000 steps (round 4).  An env that stays within what the two-env kernel holds on chip
    (<= 64 rows, <= 16 contacts) gets the same rows and state from one long launch as from the same run cut into short launches,
    bit for bit.  An env that is RETIRED in the middle of a launch is stepped by the one-env kernel until that launch ends - to the
    end of the run in the long launch, to the next boundary in the short ones, where it rejoins the two-env kernel if its rows
    allow: the two kernels agree to fp32 rounding, not bitwise, so such an env's two runs differ like two fp32 implementations do."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = _walker(spawn_z=0.06)
    n, T = 16, 240
    pairs = [(b, '') for b in m.body_names[1:] if b.endswith('_3') or b.startswith('body_')]
    rng = np.random.default_rng(3)
    q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 7:] += rng.uniform(-0.3, 0.3, (n, m.nq - 7)); q0[:, 2] = 0.03 + 0.03*rng.uniform(size=n)
    belly = np.arange(n) % 4 == 0
    q0[belly, 2] = 0.012                                   # every fourth animal starts on its belly: more than 16 contacts at once

    def run(chunk):
        data = AnimatData(m.timestep, T, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
        sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T), n_envs=n, data=data, buffer_size=T)
        sim.reset()
        sim.physics.data.qpos[:] = torch.as_tensor(q0, dtype=torch.float32)
        sim.physics.forward(disable_actuation=True)
        assert sim.physics.kernel_info()['threads_per_env'] == 32
        sim.run(fused=True, chunk=chunk)
        torch.cuda.synchronize()
        assert int((sim.physics.data.status & ~8).abs().sum()) == 0
        s = data.sensors
        return dict(qpos=sim.physics.data.qpos.cpu().numpy(), qvel=sim.physics.data.qvel.cpu().numpy(), links=s.links.array.cpu().numpy(),
                    joints=s.joints.array.cpu().numpy(), contacts=s.contacts.array.cpu().numpy()), s.contacts.array.cpu().numpy()
    (long_, c_long), (short, _) = run(T), run(20)
    same = np.array([all(np.array_equal(long_[k][..., e, :, :] if long_[k].ndim == 4 else long_[k][e], short[k][..., e, :, :] if short[k].ndim == 4 else short[k][e])
                         for k in long_) for e in range(n)])
    print('envs bitwise equal between one launch of', T, 'and launches of 20:', same.astype(int), ' belly-landers:', belly.astype(int))
    assert same[~belly].all()                              # envs the two-env kernel keeps: bit for bit
    assert not same[belly].all()                           # the test does reach the retirement path ...
    # ... and each of its two routes is the oracle's walk: link positions of every env over the first 120 iterations (measured 2e-6 for both
    # routes, retired envs and kept ones alike)
    TO = 120
    q32 = q0.astype(np.float32).astype(np.float64); v32 = np.zeros((n, m.nv))
    st = dict(qpos=q32, qvel=v32)
    fds = [oracle.forward_debug(m, q32[e], v32[e]) for e in range(n)]
    for k in ('xpos', 'xquat', 'xipos'):
        st[k] = np.array([fd[k] for fd in fds])
    sd = np.array([fd['sensordata'] for fd in fds]); sd[:, 6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
    st['sensordata'] = sd
    want = oracle.run_fused(m, st, TO, swim=None, buffer_size=TO, controller=0, ctrl=np.zeros((n, m.nu)), n_threads=8)['links']
    e_long = np.abs(long_['links'][:TO, ..., :3] - want[..., :3]).max(); e_short = np.abs(short['links'][:TO, ..., :3] - want[..., :3]).max()
    print('link positions against the oracle over', TO, 'iterations: one launch', e_long, 'launches of 20', e_short)
    assert e_long < 2e-5 and e_short < 2e-5
    # Between themselves the routes of a retired env are two fp32 implementations of the same walk: together to 1e-3 of the rows until a
    # contact event falls differently (round 5, the matrix-core A: one belly-lander's routes part after iteration ~170 - angular velocity
    # 0.57 rad/s, pose 3e-4 at iteration 240; before that the largest difference is 7e-4 of the angular velocities, 5e-6 of the poses)
    r160 = _relerr(long_['links'][:160], short['links'][:160]); rq = _relerr(long_['qpos'], short['qpos'])
    print('one launch against launches of 20: rows of the first 160 iterations', r160, 'final qpos', rq)
    assert r160 < 1e-3 and rq < 1e-2
    assert np.abs(c_long[-1][..., 2]).max() > 0.02         # the animals rest on the floor at the end


def test_env_with_more_contacts_than_the_chip_holds_at_a_launch_boundary(oracle):
    """an env that ENTERS a fused launch with more than 16 contacts is retired by the two-env kernel before
    its first step and finished by the one-env kernel, which never writes the launch's first links row of an env it resumes - so the
    two-env kernel has to, before it lets the env go.  Animals lying on their bellies (20 contacts from the reset's mj_forward on, 17+
    for the first steps), launches of 2 steps so that several boundaries fall while the list is that long; every links / joints /
    contact row is compared with the ORACLE's run (a second GPU run would skip the same rows)."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = _walker(spawn_z=0.06)
    n, T = 6, 12
    pairs = [(b, '') for b in m.body_names[1:] if b.endswith('_3') or b.startswith('body_')]
    rng = np.random.default_rng(3)
    q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 7:] += rng.uniform(-0.05, 0.05, (n, m.nq - 7))
    q0[:, 2] = 0.012                                        # on the belly
    q0[n - 1, 2] = 0.04                                     # one animal on its feet: the partner of a retired half
    data = AnimatData(m.timestep, T, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
    sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T), n_envs=n, data=data, buffer_size=T)
    sim.reset()
    d = sim.physics.data
    d.qpos[:] = torch.as_tensor(q0, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    assert sim.physics.kernel_info()['threads_per_env'] == 32
    ncon0 = d.ncon.cpu().numpy().copy()
    assert (ncon0[:n - 1] > 16).all() and ncon0[n - 1] <= 16, ncon0
    q32 = d.qpos.cpu().numpy().astype(np.float64); v32 = d.qvel.cpu().numpy().astype(np.float64)
    st = dict(qpos=q32, qvel=v32)
    fds = [oracle.forward_debug(m, q32[e], v32[e]) for e in range(n)]
    for k in ('xpos', 'xquat', 'xipos'):
        st[k] = np.array([fd[k] for fd in fds])
    sd = np.array([fd['sensordata'] for fd in fds]); sd[:, 6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
    st['sensordata'] = sd
    seen = []
    while sim.task.sim_iteration < T:
        seen.append(d.ncon.cpu().numpy().copy())
        sim.step_fused(2)
    torch.cuda.synchronize()
    assert int((d.status & ~8).abs().sum()) == 0
    assert sum(int((s_[:n - 1] > 16).any()) for s_ in seen) >= 2, seen      # boundaries with a list too long for the chip: the first and later ones
    g2d = sim.task.maps['sensors']['geompair2data']
    ref = oracle.run_fused(m, st, T, swim=None, buffer_size=T, controller=0, ctrl=np.zeros((n, m.nu)), geompair2data=g2d,
                           n_contact_rows=len(pairs), n_threads=4)
    links = data.sensors.links.array.cpu().numpy()
    assert np.abs(links[:, :, :, 7:10]).min(axis=(2, 3)).max() < 1.0 and np.abs(links[..., 3:7]).max(axis=(2, 3)).min() > 0.5, 'a links row was never written'
    for it in range(T):
        assert _relerr(links[it], ref['links'][it]) < 2e-3, (it, _relerr(links[it], ref['links'][it]))
    assert _relerr(links[0], ref['links'][0]) < 1e-6        # row 0 holds the reset's poses: no dynamics in it yet
    rows = data.sensors.contacts.array.cpu().numpy(); want = ref['contacts']
    scale = np.abs(want[..., :9]).max()
    assert np.abs(rows[..., :9] - want[..., :9]).max()/scale < 2e-2 and scale > 0.05
