"""GPU parity of the RK4 integrator (round 5): fmj_step's four forward launches + fmj_rk4_stage_kernel against the oracle's mj_RungeKutta
restatement (oracle/fmj_oracle.c rk4) on identical inputs; the reference forwards simulation_options.integrator (mjcf.py:1360-1365)."""
import numpy as np
import pytest

from parity_metrics import relerr as _relerr

pytestmark = pytest.mark.gpu


def _physics(m, n):
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    return BatchedPhysics(m, n, 'cuda:0'), torch


def _state(m, n, seed):
    from test_gpu_step_parity import _rand_state
    return _rand_state(m, n, seed, qscale=0.2, vscale=0.3)


def test_rk4_step_matches_oracle_swimming(oracle):
    """One step: every field mj_step leaves (state, qacc of the last pass, sensordata of the first, poses of the last); then 200 steps of
    the free-swimming animal with external forces held over each step."""
    from farms_mujoco_amd.model import salamander33
    m = salamander33()
    m.integrator = 1
    n = 32
    phys, torch = _physics(m, n)
    assert phys.rk4 and phys.kernel_info()['threads_per_env'] == 64          # the one-env kernel runs the passes
    qpos, qvel, ctrl = _state(m, n, 5)
    xf = np.random.default_rng(2).normal(size=(n, m.nbody, 6))*0.01; xf[:, 0] = 0
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    d.ctrl[:] = torch.as_tensor(ctrl, dtype=torch.float32); d.xfrc_applied[:] = torch.as_tensor(xf, dtype=torch.float32)
    r64 = lambda t: t.cpu().numpy().astype(np.float64)
    q32, v32, c32, x32 = r64(d.qpos), r64(d.qvel), r64(d.ctrl), r64(d.xfrc_applied)
    phys.step(1)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=c32, xfrc_applied=x32)
    m.integrator = 0
    eul = oracle.step(m, q32, v32, ctrl=c32, xfrc_applied=x32)
    m.integrator = 1
    assert int(d.status.abs().sum()) == 0
    errs = {k: _relerr(getattr(d, k).cpu().numpy(), ref[k]) for k in ('qpos', 'qvel', 'qacc', 'xpos', 'xquat', 'xipos', 'sensordata')}
    print('RK4, one step against the oracle:', errs, ' (RK4 against Euler in the oracle: qvel', _relerr(eul['qvel'], ref['qvel']), ')')
    for k, tol in (('qpos', 2e-6), ('qvel', 2e-4), ('qacc', 2e-3), ('xpos', 2e-6), ('xquat', 2e-6), ('xipos', 2e-6), ('sensordata', 2e-4)):
        assert errs[k] < tol, (k, errs[k])
    assert _relerr(eul['qvel'], ref['qvel']) > 10*errs['qvel']             # the test tells the integrators apart
    assert abs(float(d.time[0]) - m.timestep) < 1e-9
    T = 200
    phys.step(T)
    torch.cuda.synchronize()
    ref = oracle.step(m, ref['qpos'], ref['qvel'], ctrl=c32, xfrc_applied=x32, n_steps=T)
    e_q = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max(); e_v = _relerr(d.qvel.cpu().numpy(), ref['qvel'])
    print('after', T, 'more steps: qpos abs', e_q, 'qvel rel', e_v)
    assert int(d.status.abs().sum()) == 0 and e_q < 5e-5 and e_v < 2e-3


def test_rk4_walking_with_contacts_matches_oracle(oracle):
    """Limits + ground contacts (PGS): every pass makes its own contacts and solves its own rows from the step's warm start; 60 steps of
    the walker settling on the floor."""
    from test_gpu_contacts import _walker, _set
    m = _walker()
    m.integrator = 1
    n, T = 8, 60
    phys, torch = _physics(m, n)
    rng = np.random.default_rng(4)
    q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 7:] += rng.uniform(-0.1, 0.1, (n, m.nq - 7)); q0[:, 2] = 0.035 + 0.01*rng.uniform(size=n)
    q32, v32 = _set(phys, q0, np.zeros((n, m.nv)))
    phys.step(T)
    torch.cuda.synchronize()
    d = phys.data
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)), n_steps=T)
    e_q = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max()
    print('RK4 walker,', T, 'steps: qpos abs', e_q, 'contacts', d.ncon.cpu().numpy())
    assert int((d.status & ~8).abs().sum()) == 0 and int(d.ncon.sum()) > 0
    assert e_q < 2e-4
    assert _relerr(d.qacc_warmstart.cpu().numpy(), ref['qacc_warmstart']) < 5e-2 if 'qacc_warmstart' in ref else True


def test_rk4_through_the_simulation_api(oracle):
    """SimulationOptions(integrator='RK4'): Simulation.run() takes the per-iteration path by itself (fmj_step_fused refuses RK4), rows and
    drag written by fmj_before_step, the wave controller evaluated on the host; the log matches the oracle's run of the same loop."""
    import torch
    from farms_mujoco_amd import _lib
    from farms_mujoco_amd.model import salamander33
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = salamander33()
    m.integrator = 1
    n, T = 4, 40
    sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T, integrator='RK4'), n_envs=n, buffer_size=T)
    assert not sim.task.fusable()
    sim.run()
    torch.cuda.synchronize()
    assert sim.task.iteration == T and int(sim.physics.data.status.abs().sum()) == 0
    assert abs(float(sim.physics.data.time[0]) - T*m.timestep) < 1e-6
    with pytest.raises(_lib.FmjError, match='RK4'):
        sim.task.host_step_only = False
        sim.task.sim_iteration = 0; sim.task.iteration = 0
        sim.step_fused(1)
