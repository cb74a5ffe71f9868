"""Other morphologies through the same HIP path (BASELINE config 5: eel + centipede, variable link count):
each morphology gets its own context (bucketed batching: no padding, no masks), checked against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max()/max(np.abs(b).max(), 1e-12)


@pytest.mark.parametrize('maker', ['eel', 'centipede'])
def test_step_parity_other_morphologies(oracle, maker):
    import torch
    import farms_mujoco_amd.model as mm
    from farms_mujoco_amd.physics import BatchedPhysics
    m = getattr(mm, maker)()
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
    # the first step starts from rest with a ctrl jump (qacc ~ 5e4 rad/s^2); the long, light chains of these
    # morphologies amplify fp32 rounding of M in the solve more than the salamander does -> 2e-3 on qvel
    for k, tol in (('xpos', 2e-6), ('xquat', 2e-6), ('sensordata', 5e-5), ('qvel', 2e-3), ('qpos', 2e-4)):
        assert _relerr(getattr(d, k).cpu().numpy(), ref1[k]) < tol, (maker, k, _relerr(getattr(d, k).cpu().numpy(), ref1[k]))
    phys.step(T - 1, ctrl_tape=tape_t[1:].contiguous())
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, qvel, ctrl=tape_t.cpu().numpy().astype(np.float64), n_steps=T, ctrl_step_stride=n*m.nu, n_threads=8)
    assert int(d.status.abs().sum()) == 0
    err = _relerr(d.qpos.cpu().numpy(), ref['qpos'])
    print(maker, 'nv', m.nv, 'qpos rel err after', T, 'steps:', err)
    assert err < 1e-4


def test_mixed_batch_bucketed(oracle):
    """Mixed-morphology batch = one fused simulation per morphology bucket, launched back to back on the same
    stream; each bucket matches the oracle and is unaffected by the presence of the other."""
    import torch
    import farms_mujoco_amd.model as mm
    from farms_mujoco_amd.options import SimulationOptions, ArenaOptions, AnimatOptions, WaterOptions
    from farms_mujoco_amd.control import WaveController
    from farms_mujoco_amd.simulation.simulation import Simulation
    T = 40
    sims = []
    for maker, n in (('eel', 24), ('centipede', 8)):
        m = getattr(mm, maker)()
        _, _, psi = mm.synthetic_batch(m, n, seed=9)
        sim = Simulation.from_sdf(SimulationOptions(timestep=m.timestep, n_iterations=T), AnimatOptions.from_model(m),
                                  ArenaOptions(water=WaterOptions(height=0.0)), model=m, n_envs=n,
                                  controller=WaveController(m, psi, frequency=1.5), buffer_size=T)
        sim.reset()
        sims.append((sim, m, psi))
    for sim, _, _ in sims:
        sim.run(fused=True)
    torch.cuda.synchronize()
    for sim, m, psi in sims:
        assert int(sim.physics.data.status.abs().sum()) == 0
        links = sim.task.data.sensors.links.array.cpu().numpy()
        assert np.isfinite(links).all() and np.abs(links[-1, :, :, 14:17]).max() > 1e-3      # it swims


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
