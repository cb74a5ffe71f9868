"""The Newton solver of the HIP constraint path (solver = 'Newton', reference mjcf.py:1348-1359: the reference's own fallback)
against the oracle's Newton (oracle/fmj_oracle.c solve_primal), itself cross-checked against a converged PGS on CPU
(tests/test_oracle_solvers.py)."""
import copy

import numpy as np
import pytest

from parity_metrics import group_relerr, qvel_groups

pytestmark = pytest.mark.gpu


def _walker(spawn_z=0.045, solver='newton'):
    from farms_mujoco_amd.model import salamander33, SOLVERS
    m = salamander33(contacts=True, limits=True, spawn_z=spawn_z)
    m.solver = SOLVERS[solver]; m.solver_iterations = 100
    return m


def _set(phys, qpos, qvel, warm=None):
    import torch
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    if warm is not None:
        d.qacc_warmstart[:] = torch.as_tensor(warm, dtype=torch.float32)
    r64 = lambda t: t.cpu().numpy().astype(np.float64)
    return r64(d.qpos), r64(d.qvel), r64(d.qacc_warmstart)


@pytest.mark.parametrize('solver', ['newton', 'cg'])
def test_newton_single_step_forces_and_kkt(oracle, solver):
    """Feet on the floor, bellies pressed in (more than 64 rows), a joint past its limit: contact list, forces and velocity of one
    step against the oracle's Newton; the forces satisfy the KKT conditions of the fp64 problem."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = _walker(solver=solver)
    n = 12
    rng = np.random.default_rng(0)
    qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 7:] += rng.uniform(-0.1, 0.1, (n, m.nq - 7))
    qpos[:4, 2] = 0.012                       # belly contacts
    qpos[:, 7 + 3] = 1.25                     # a spine joint past its +1.2 rad limit
    qvel = rng.normal(size=(n, m.nv))*0.05
    phys = BatchedPhysics(m, n)
    q32, v32, w32 = _set(phys, qpos, qvel, rng.normal(size=(n, m.nv))*5.0)
    rows, _ = phys.step_debug(want_pgs=False)
    torch.cuda.synchronize()
    d = phys.data
    assert int(d.status.abs().sum()) == 0
    o = oracle.step_tf(m, q32, v32, ctrl=np.zeros((n, m.nu)), warmstart=w32)
    with oracle.fp32_storage():
        fl = oracle.step_tf(m, q32, v32, ctrl=np.zeros((n, m.nu)), warmstart=w32, want_AR=False)
    assert np.array_equal(d.ncon.cpu().numpy(), o['ncon']) and o['nefc'].max() > 64 and o['nefc'].min() <= 40
    rows = rows.cpu().numpy().astype(np.float64)
    worst = 0.0
    for e in range(n):
        ne = int(o['nefc'][e])
        f_h = rows[e, :ne, 4]; f_o = o['efc'][e, :ne, 0]; b = o['efc'][e, :ne, 1]; AR = o['AR'][e, :ne, :ne]
        fs = max(np.abs(f_o).max(), 1e-2); bs = max(np.abs(b).max(), 1.0)
        worst = max(worst, np.abs(f_h - f_o).max()/fs)
        r = AR @ f_h + b
        assert f_h.min() >= 0.0 and r.min() > -2e-3*bs and np.abs(f_h*r).max() < 2e-3*fs*bs, (e, r.min()/bs, np.abs(f_h*r).max()/(fs*bs))
    err = group_relerr(d.qvel.cpu().numpy(), o['qvel'], qvel_groups(m)); floor = group_relerr(fl['qvel'], o['qvel'], qvel_groups(m))
    print(solver, 'single step: forces', worst, 'qvel per component', err, 'fp32-storage floor', floor)
    assert worst < 2e-3
    assert err < 6*floor + 1e-6
    assert np.abs(d.qacc_warmstart.cpu().numpy() - o['warmstart']).max() < 2e-3*np.abs(o['warmstart']).max()


@pytest.mark.parametrize('solver', ['newton', 'cg'])
def test_newton_walk_follows_the_oracle(oracle, solver):
    """300 steps of the trot with the Newton solver, fused loop with contact rows: no warning bits, the floor carries the animal,
    and since Newton converges at every step (unlike PGS cut at 50 sweeps) the walk stays on the oracle's."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    from test_gpu_contacts import _trot_tape
    m = _walker(solver=solver)
    n, T = 8, 300
    tape = _trot_tape(m, n, T)
    phys = BatchedPhysics(m, n)
    q32, v32, _ = _set(phys, np.tile(m.qpos0, (n, 1)), np.zeros((n, m.nv)))
    tape_t = torch.as_tensor(tape, dtype=torch.float32, device='cuda').contiguous()
    phys.step(T, ctrl_tape=tape_t)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=tape_t.cpu().numpy().astype(np.float64), n_steps=T, ctrl_step_stride=n*m.nu, n_threads=8)
    with oracle.fp32_storage():
        fl = oracle.step(m, q32, v32, ctrl=tape_t.cpu().numpy().astype(np.float64), n_steps=T, ctrl_step_stride=n*m.nu, n_threads=8)
    d = phys.data
    assert int(d.status.abs().sum()) == 0 and int(ref['status'].sum()) == 0
    e = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max(1); f = np.abs(fl['qpos'] - ref['qpos']).max(1)
    print(solver, 'walk, qpos abs err per env after', T, 'steps:', e, 'fp32-storage floor run:', f)
    assert float(d.qpos[:, 2].min()) > 0.0 and float(d.qpos[:, 2].max()) < 0.1
    # 10 - 20x the fp32-storage floor of the same 300 steps (the order in which the Hessian rows are summed moves it between builds),
    # an order of magnitude inside the bounds of the PGS walk (5e-4 median, 5e-3 worst)
    assert np.median(e) < 1e-4 and e.max() < 1e-3 and np.median(e) < 40*np.median(f)


def test_newton_with_mesh_feet_on_a_heightfield(oracle):
    """The reference's usual configuration - SDF mesh collisions (mjcf.py:270-413) under its fallback solver Newton (:1348-1359) - here
    on a heightfield: 60 steps of standing / settling against the oracle's Newton."""
    import torch
    from farms_mujoco_amd.model import salamander33, SOLVERS
    from farms_mujoco_amd.physics import BatchedPhysics
    m = salamander33(contacts=True, limits=True, spawn_z=0.05, mesh_feet=True, terrain='hfield')
    m.solver = SOLVERS['newton']; m.solver_iterations = 100
    n, T = 6, 60
    rng = np.random.default_rng(3)
    qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 7:] += rng.uniform(-0.1, 0.1, (n, m.nq - 7)); qpos[:, :2] += rng.uniform(-0.3, 0.3, (n, 2))
    phys = BatchedPhysics(m, n)
    q32, v32, _ = _set(phys, qpos, np.zeros((n, m.nv)))
    phys.step(T)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)), n_steps=T)
    d = phys.data
    assert int(d.status.abs().sum()) == 0 and int(ref['status'].sum()) == 0
    fds = [oracle.forward_debug(m, ref['qpos'][e], ref['qvel'][e], ctrl=np.zeros(m.nu)) for e in range(n)]
    assert max(fd['ncon'] for fd in fds) >= 4
    e = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max(1)
    print('Newton, mesh feet on a heightfield, qpos abs err per env after', T, 'steps:', e)
    assert e.max() < 1e-3


def test_newton_and_cg_refused_for_pairs():
    from farms_mujoco_amd.model import salamander33, SOLVERS
    from farms_mujoco_amd.physics import BatchedPhysics
    from farms_mujoco_amd._lib import FmjError
    for solver in ('newton', 'cg'):
        m = salamander33(contacts=True, limits=True, spawn_z=0.045, self_collisions=True)
        m.solver = SOLVERS[solver]
        with pytest.raises(FmjError):
            BatchedPhysics(m, 2)
