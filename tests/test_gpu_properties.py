"""Oracle-independent properties of the HIP step on the GPU: conservation laws, run-to-run determinism, long horizons."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_swim_is_equivariant_under_horizontal_motions():
    """Gravity is vertical and the water surface horizontal, so translating an env horizontally and turning it about z
    must turn and translate its whole trajectory and leave the joint angles alone (fp32 rounding aside).  400 fused
    steps of swimming with drag and buoyancy; envs come in (original, moved) pairs."""
    import torch
    from farms_mujoco_amd.model import salamander33, synthetic_batch
    from farms_mujoco_amd.options import SimulationOptions, ArenaOptions, AnimatOptions, WaterOptions
    from farms_mujoco_amd.control import WaveController
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = salamander33()
    n, T = 16, 400
    qpos, qvel, psi = synthetic_batch(m, n)
    rng = np.random.default_rng(5)
    th = rng.uniform(-np.pi, np.pi, n//2); dxy = rng.uniform(-2.0, 2.0, (n//2, 2))
    for k in range(n//2):
        a, b = 2*k, 2*k + 1
        c, s_ = np.cos(th[k]), np.sin(th[k])
        Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
        qz = np.array([np.cos(th[k]/2), 0, 0, np.sin(th[k]/2)])
        qpos[b] = qpos[a]; qvel[b] = qvel[a]; psi[b] = psi[a]
        qpos[b, :3] = Rz @ qpos[a, :3] + [dxy[k, 0], dxy[k, 1], 0]
        w0, x0, y0, z0 = qpos[a, 3:7]; w1, x1, y1, z1 = qz
        qpos[b, 3:7] = [w1*w0 - x1*x0 - y1*y0 - z1*z0, w1*x0 + x1*w0 + y1*z0 - z1*y0,
                        w1*y0 - x1*z0 + y1*w0 + z1*x0, w1*z0 + x1*y0 - y1*x0 + z1*w0]      # qz * q
        qvel[b, :3] = Rz @ qvel[a, :3]          # free-joint linear velocity is in world axes, angular in body axes
    sim = Simulation.from_sdf(SimulationOptions(timestep=m.timestep, n_iterations=T), AnimatOptions.from_model(m),
                              ArenaOptions(water=WaterOptions(height=0.0, drag=True, buoyancy=True, viscosity=1.0)),
                              model=m, n_envs=n, controller=WaveController(m, psi), buffer_size=T)
    sim.reset()
    d = sim.physics.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    sim.run(fused=True)
    torch.cuda.synchronize()
    assert int(d.status.abs().sum()) == 0
    q = d.qpos.cpu().numpy().astype(np.float64)
    links = sim.task.data.sensors.links.array.cpu().numpy().astype(np.float64)      # [T, n, n_links, 20]
    for k in range(n//2):
        a, b = 2*k, 2*k + 1
        c, s_ = np.cos(th[k]), np.sin(th[k])
        Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
        assert np.abs(q[a, 7:] - q[b, 7:]).max() < 2e-4                             # joint angles
        assert np.abs(Rz @ q[a, :3] + [dxy[k, 0], dxy[k, 1], 0] - q[b, :3]).max() < 2e-4
        pa = links[:, a, :, 0:3]; pb = links[:, b, :, 0:3]                          # link CoM positions over time
        assert np.abs(pa @ Rz.T + [dxy[k, 0], dxy[k, 1], 0] - pb).max() < 3e-4
        va = links[:, a, :, 14:17]; vb = links[:, b, :, 14:17]                      # CoM linear velocities
        assert np.abs(va @ Rz.T - vb).max() < 3e-3
    assert np.abs(q[:, :2] - qpos[:, :2]).max() > 0.005                             # they did move


@pytest.mark.parametrize('workload', ['swim', 'walk'])
def test_run_to_run_determinism(workload):
    """The same launch twice gives bitwise the same state and rows (no atomics, no uninitialised reads, the HBM
    scratch of the constraint path included)."""
    import torch
    import bench
    outs = []
    for _ in range(2):
        sim, m, _ = bench.build_sim(512, 200, 100, 0, 'cuda:0', workload)
        if workload == 'walk':
            sim.physics.data.qpos[:64, 2] = 0.012           # some envs on their belly: > 60 constraint rows
            sim.physics.data.qacc_warmstart.zero_()
            sim.physics.forward(disable_actuation=True)
        sim.run(fused=True)
        torch.cuda.synchronize()
        d = sim.physics.data
        outs.append((d.qpos.cpu().numpy().copy(), d.qvel.cpu().numpy().copy(),
                     sim.task.data.sensors.links.array.cpu().numpy().copy(), d.status.cpu().numpy().copy()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert np.isfinite(outs[0][0]).all()


def test_long_horizon_swim_parity(oracle):
    """4000 steps (4 s of swimming) against the fp64 oracle: the error stays far below the 1e-4 north-star bound set
    for 1000 steps."""
    import torch
    from farms_mujoco_amd.model import salamander33, wave_controller_params
    from farms_mujoco_amd.physics import BatchedPhysics
    m = salamander33()
    n, T = 8, 4000
    rng = np.random.default_rng(3)
    qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 7:] += rng.uniform(-0.05, 0.05, (n, m.nq - 7))
    phys = BatchedPhysics(m, n)
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel.zero_()
    q32 = d.qpos.cpu().numpy().astype(np.float64)
    amp, lag = wave_controller_params(m)
    t = np.arange(T)[:, None, None]*m.timestep
    tape = torch.as_tensor(amp[None, None, :]*np.sin(2*np.pi*t - lag[None, None, :] + rng.uniform(0, 6.28, (1, n, 1))),
                           dtype=torch.float32, device='cuda').contiguous()
    phys.step(T, ctrl_tape=tape)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, np.zeros((n, m.nv)), ctrl=tape.cpu().numpy().astype(np.float64), n_steps=T,
                      ctrl_step_stride=n*m.nu, n_threads=8)
    assert int(d.status.abs().sum()) == 0
    err = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max()/np.abs(ref['qpos']).max()
    print('qpos rel err after 4000 steps:', err)
    assert err < 1e-4, err


def test_twenty_seconds_of_swimming_stay_sane():
    """20 000 fused steps (20 s of simulated swimming, ring of 100 rows wrapping 200 times) on 1024 envs: no warning
    bit, everything finite, joints inside a sane range, the animals keep moving, the ring holds the last 100 rows."""
    import torch
    import bench
    sim, m, _ = bench.build_sim(1024, 20000, 100, 0, 'cuda:0', 'swim')
    sim.run(fused=True)
    torch.cuda.synchronize()
    d = sim.physics.data
    assert int(d.status.abs().sum()) == 0
    q = d.qpos.cpu().numpy(); v = d.qvel.cpu().numpy()
    assert np.isfinite(q).all() and np.isfinite(v).all()
    assert np.abs(q[:, 7:]).max() < 1.5 and np.abs(v).max() < 50.0
    assert np.abs(np.linalg.norm(q[:, 3:7], axis=1) - 1.0).max() < 1e-5
    links = sim.task.data.sensors.links.array.cpu().numpy()
    assert np.isfinite(links).all()
    speed = np.linalg.norm(links[:, :, 0, 14:17], axis=-1)              # head link CoM speed over the last 100 steps
    assert 0.005 < speed.mean() < 2.0
    assert sim.task.iteration == 20000
