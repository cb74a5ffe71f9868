"""The two solver options round 4 still refused on the device (the reference forwards `cone`, `solver` and
`noslip_iterations` independently, mjcf.py:1342-1353,1392-1403):
  * PGS with the ELLIPTIC cone - MuJoCo's block update per contact (ray update + friction QCQP), restated by the oracle's
    pgs_elliptic_block - also on models with explicit self-collision pairs;
  * the NOSLIP post-pass (oracle: noslip()) after PGS, and after Newton (which the device then runs on the dual problem).
Every test compares the HIP path with the fp64 oracle on identical fp32 inputs."""
import numpy as np
import pytest

from parity_metrics import group_relerr, qvel_groups

pytestmark = pytest.mark.gpu


def _walker(solver='pgs', cone='pyramidal', impratio=1.0, noslip=0, spawn_z=0.045, **kw):
    from farms_mujoco_amd.model import salamander33, SOLVERS, CONES
    m = salamander33(contacts=True, limits=True, spawn_z=spawn_z, **kw)
    m.solver = SOLVERS[solver]; m.cone = CONES[cone]; m.impratio = impratio
    m.noslip_iterations = noslip; m.noslip_tolerance = 1e-10
    if solver != 'pgs':
        m.solver_iterations = 100
    return m


def _set(phys, qpos, qvel, warm=None):
    import torch
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    if warm is not None:
        d.qacc_warmstart[:] = torch.as_tensor(warm, dtype=torch.float32)
    r = lambda t: t.cpu().numpy().astype(np.float64)
    return r(d.qpos), r(d.qvel), r(d.qacc_warmstart)


def _states(m, n, seed, belly=2):
    rng = np.random.default_rng(seed)
    qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 7:] += rng.uniform(-0.1, 0.1, (n, m.nq - 7))
    qpos[:belly, 2] = 0.012                      # bellies pressed in: many contacts
    qpos[:, 7 + 3] = 1.25                        # a joint past its limit
    qvel = rng.normal(size=(n, m.nv))*0.05; qvel[:, :2] += 0.3      # sliding feet
    return qpos, qvel, rng.normal(size=(n, m.nv))*5.0


def _single_step(oracle, m, n=10, seed=1, rows_per_contact=4):
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    qpos, qvel, warm = _states(m, n, seed)
    phys = BatchedPhysics(m, n)
    q32, v32, w32 = _set(phys, qpos, qvel, warm)
    rows, _ = phys.step_debug(want_pgs=False)
    torch.cuda.synchronize()
    d = phys.data
    assert int(d.status.abs().sum()) == 0
    o = oracle.step_tf(m, q32, v32, ctrl=np.zeros((n, m.nu)), warmstart=w32, want_AR=False)
    with oracle.fp32_storage():
        fl = oracle.step_tf(m, q32, v32, ctrl=np.zeros((n, m.nu)), warmstart=w32, want_AR=False)
    assert np.array_equal(d.ncon.cpu().numpy(), o['ncon'])
    rows = rows.cpu().numpy().astype(np.float64)
    con = oracle.contacts_from_hip(d.contact.cpu().numpy())
    worst = worst_c = 0.0
    for e in range(n):
        ne, nc = int(o['nefc'][e]), int(o['ncon'][e])
        f_h, f_o = rows[e, :ne, 4], o['efc'][e, :ne, 0]
        fs = max(np.abs(f_o).max(), 1e-2)
        worst = max(worst, np.abs(f_h - f_o).max()/fs)
        worst_c = max(worst_c, np.abs(con[e, :nc, 12:15] - o['contact'][e, :nc, 12:15]).max()/fs)
    err = group_relerr(d.qvel.cpu().numpy(), o['qvel'], qvel_groups(m)); floor = group_relerr(fl['qvel'], o['qvel'], qvel_groups(m))
    return dict(rows=worst, contact=worst_c, qvel=err, floor=floor, nefc=o['nefc'], ncon=o['ncon'], con=con, phys=phys, o=o)


@pytest.mark.parametrize('impratio', [1.0, 4.0])
def test_elliptic_cone_with_pgs_single_step(oracle, impratio):
    """mj_solPGS with elliptic cones: sliding feet (forces on the cone), bellies with more rows than a wave holds, a joint past its
    limit.  PGS cut at 50 sweeps is an unconverged map: device and oracle must walk through the SAME sweeps, so the forces agree like
    the pyramidal PGS's do (teacher-forced walking: 1.2e-4 at the 99th percentile)."""
    m = _walker(cone='elliptic', impratio=impratio)
    r = _single_step(oracle, m)
    print('elliptic PGS, impratio', impratio, {k: r[k] for k in ('rows', 'contact', 'qvel', 'floor')}, 'rows per env', r['nefc'])
    assert r['nefc'].max() > 50 and r['nefc'].min() >= 4
    assert r['rows'] < 3e-3 and r['contact'] < 3e-3
    assert r['qvel'] < 6*r['floor'] + 1e-6
    con = r['con']; mf = np.asarray(m.geom_friction)
    for e in range(len(r['ncon'])):
        nc = int(r['ncon'][e])
        fn, ft = con[e, :nc, 12], np.hypot(con[e, :nc, 13], con[e, :nc, 14])
        mus = np.maximum(mf[con[e, :nc, 16].astype(int), 0], mf[con[e, :nc, 15].astype(int), 0])
        assert (fn >= 0).all() and (ft <= mus*fn*(1 + 1e-4) + 1e-6).all()              # friction inside the cone


def test_elliptic_cone_with_pgs_walk(oracle):
    """100 free-running steps of the trot, fused launches with contact rows, PGS + elliptic cones, against the oracle's walk."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    from test_gpu_contacts import _trot_tape
    m = _walker(cone='elliptic')
    n, T = 6, 100
    tape = _trot_tape(m, n, T).astype(np.float32)
    from farms_mujoco_amd.physics import BatchedPhysics
    phys = BatchedPhysics(m, n)
    q32, v32, _ = _set(phys, np.tile(m.qpos0, (n, 1)), np.zeros((n, m.nv)))
    tape_t = torch.as_tensor(tape, device='cuda').contiguous()
    phys.step(T, ctrl_tape=tape_t)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=tape.astype(np.float64), n_steps=T, ctrl_step_stride=n*m.nu, n_threads=6)
    d = phys.data
    assert int(d.status.abs().sum()) == 0 and int(ref['status'].sum()) == 0
    rel = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max(1)/np.abs(ref['qpos']).max(1)
    print('elliptic PGS walk, qpos rel err per env after', T, 'steps:', rel, 'ncon', d.ncon.cpu().numpy())
    assert rel.max() < 1e-4 and int(d.ncon.max()) >= 2


def test_elliptic_cone_with_pairs(oracle):
    """The refusal "no elliptic cone on a model with explicit pairs" is gone: the salamander with self-collision pairs, limbs crossed
    under the trunk / curled up head to tail (the poses of test_gpu_newton.py::test_newton_and_cg_with_self_collision_pairs), PGS with
    elliptic cones; contact-frame forces (the friction-0 pairs leave the split among a contact's rows undetermined, not their sum)."""
    import torch
    from farms_mujoco_amd.model import CONES
    from farms_mujoco_amd.physics import BatchedPhysics
    from test_gpu_contacts import _salamander_self_collisions
    m = _salamander_self_collisions()
    m.cone = CONES['elliptic']
    plane = int(np.nonzero(m.geom_type == 0)[0][0])
    spine = [m.jnt_qposadr[m.joint_names.index(f'joint_body_{i}')] for i in range(1, 12)]
    legj = [m.jnt_qposadr[m.joint_names.index(f'joint_leg_{t}_{s_}_{k}')] for t in ('front', 'hind') for s_ in ('L', 'R') for k in range(4)]
    poses = []
    for legs in ([-1.19, 0.07, -0.84, -0.74, 0.24, -0.09, 1.01, -0.27, 0.42, -0.19, 0.52, -0.12, -0.63, 0.83, 0.69, -0.46],
                 [-0.25, -0.11, 0.59, -0.16, 1.02, 0.67, 0.77, -0.62, -0.01, -0.32, -0.9, 0.36, 0.78, 0.27, 1.14, -0.7]):
        q = m.qpos0.copy(); q[2] = 0.05; q[legj] = legs; poses.append(q)
    for curl in (0.645, 0.65):
        q = m.qpos0.copy(); q[2] = 0.03; q[spine] = curl; poses.append(q)
    qpos = np.array(poses); n = len(poses)
    rng = np.random.default_rng(21)
    phys = BatchedPhysics(m, n)
    q32, v32, w32 = _set(phys, qpos, 0.02*rng.normal(size=(n, m.nv)))
    phys.step_debug(want_pgs=False)
    torch.cuda.synchronize()
    d = phys.data
    o = oracle.step_tf(m, q32, v32, ctrl=np.zeros((n, m.nu)), warmstart=w32, want_AR=False)
    assert int(d.status.abs().sum()) == 0 and np.array_equal(d.ncon.cpu().numpy(), o['ncon'])
    npair = [int((o['contact'][e, :o['ncon'][e], 15] != plane).sum()) for e in range(n)]
    assert min(npair) >= 1, npair
    con = oracle.contacts_from_hip(d.contact.cpu().numpy())
    worst = 0.0
    for e in range(n):
        nc = int(o['ncon'][e])
        fs = max(np.abs(o['contact'][e, :nc, 12]).max(), 1e-2)
        worst = max(worst, np.abs(con[e, :nc, 12:15] - o['contact'][e, :nc, 12:15]).max()/fs)
    err = group_relerr(d.qvel.cpu().numpy(), o['qvel'], qvel_groups(m))
    print('elliptic PGS with pairs: contact-frame forces', worst, 'qvel per component', err, 'pair contacts per env', npair)
    assert worst < 5e-3 and err < 3e-2


@pytest.mark.parametrize('cone', ['pyramidal', 'elliptic'])
def test_noslip_single_step(oracle, cone):
    """noslip_iterations > 0 after PGS: the friction forces are re-solved without the regulariser, normal and limit forces stay.
    Device against oracle, and the pass really moved the forces (against the same step without it)."""
    m = _walker(cone=cone, noslip=10)
    r = _single_step(oracle, m, seed=2)
    m0 = _walker(cone=cone, noslip=0)
    r0 = _single_step(oracle, m0, seed=2)
    moved = max(np.abs(r['o']['efc'][e, :int(r['nefc'][e]), 0] - r0['o']['efc'][e, :int(r['nefc'][e]), 0]).max() for e in range(len(r['nefc'])))
    print(cone, 'noslip:', {k: r[k] for k in ('rows', 'contact', 'qvel', 'floor')}, 'forces moved by', moved)
    assert moved > 1e-3
    assert r['rows'] < 3e-3 and r['contact'] < 3e-3
    assert r['qvel'] < 6*r['floor'] + 1e-5


def test_noslip_holds_a_slab_on_an_incline(oracle):
    """The purpose of the option, on the device: a slab at rest on an incline it can hold creeps under the soft contact, and stops
    creeping with noslip (tests/test_oracle_solvers.py::test_noslip_stops_the_creep_of_a_sticking_contact is the oracle's side)."""
    import torch
    from farms_mujoco_amd.model import ModelBuilder, GEOM_BOX, GEOM_PLANE
    from farms_mujoco_amd.physics import BatchedPhysics
    th, mu, yaw, g = 0.3, 0.6, 0.3, 9.81
    vt = {}
    for ns in (0, 50):
        b = ModelBuilder('slab', timestep=1e-3, gravity=(g*np.sin(th)*np.cos(yaw), g*np.sin(th)*np.sin(yaw), -g*np.cos(th)))
        b.add_body('slab', 'world', pos=(0, 0, 0.02), mass=1.0, inertia=(4e-3, 4e-3, 8e-3), joint='free')
        b.add_geom('slab', GEOM_BOX, (0.1, 0.1, 0.02), friction=(mu, 0, 0))
        b.add_geom('world', GEOM_PLANE, (0, 0, 0), friction=(mu, 0, 0))
        b.options['max_contacts'] = 8
        m = b.compile()
        m.solver_iterations = 200; m.solver_tolerance = 1e-10; m.noslip_iterations = ns; m.noslip_tolerance = 1e-12
        phys = BatchedPhysics(m, 2)
        q32, v32, _ = _set(phys, np.tile(m.qpos0, (2, 1)), np.zeros((2, m.nv)))
        phys.step(300)
        torch.cuda.synchronize()
        v = phys.data.qvel.cpu().numpy()
        ref = oracle.step(m, q32, v32, n_steps=300)
        assert int(phys.data.status.abs().sum()) == 0 and int(phys.data.ncon[0]) == 4
        vt[ns] = (float(np.hypot(v[0, 0], v[0, 1])), float(np.hypot(ref['qvel'][0, 0], ref['qvel'][0, 1])))
    print('creep velocity (device, oracle): without noslip', vt[0], 'with', vt[50])
    assert vt[0][0] > 1e-5 and abs(vt[0][0] - vt[0][1]) < 0.05*vt[0][1]
    assert vt[50][0] < 2e-2*vt[0][0]


def test_noslip_after_newton_runs_on_the_dual_problem(oracle):
    """Newton + noslip: the post-pass works on the dual matrices, which the primal solvers never form; the device then solves the whole
    step on the dual problem (PGS to the solver's tolerance: the same minimiser) and says so (fmj_solver_info, a warning)."""
    m = _walker(solver='newton', noslip=10)
    m.solver_tolerance = 1e-10
    with pytest.warns(UserWarning, match='dual problem'):
        r = _single_step(oracle, m, n=6, seed=3)
    assert r['phys'].solver_requested == 'Newton' and r['phys'].solver_effective == 'PGS'
    print('Newton + noslip:', {k: r[k] for k in ('rows', 'contact', 'qvel', 'floor')})
    assert r['rows'] < 5e-3 and r['qvel'] < 6*r['floor'] + 1e-4
