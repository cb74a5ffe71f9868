"""Checkpoint / resume of a running simulation: ``Simulation.save_state`` in the middle of a
run, ``load_state`` into a fresh Simulation, and the rest of the run - state, warm start, counters, controller state and every ring
buffer row - is what the uninterrupted run produces, bit for bit.  Swimming (wave controller inside the launch), swimming with the
oscillator-network controller (its phases are part of the checkpoint) and walking (warm start, contacts, the two-env constraint
kernel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(sim):
    s = sim.task.data.sensors
    out = {k: getattr(s, k).array.cpu().numpy().copy() for k in ('links', 'joints', 'xfrc')}
    if s.contacts.names:
        out['contacts'] = s.contacts.array.cpu().numpy().copy()
    return out


def _state(sim):
    return sim.physics.get_state()


def _swim(n, T, controller='wave', seed=3):
    import torch
    from farms_mujoco_amd.control import NetworkController, WaveController, salamander_network
    from farms_mujoco_amd.model import salamander33, synthetic_batch
    from farms_mujoco_amd.options import AnimatOptions, ArenaOptions, SimulationOptions, WaterOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = salamander33()
    qpos, qvel, psi = synthetic_batch(m, n, seed=seed)
    c = WaveController(m, psi) if controller == 'wave' else NetworkController(m, salamander_network(m), n, env_phase=psi)
    sim = Simulation.from_sdf(SimulationOptions(timestep=m.timestep, n_iterations=T), AnimatOptions.from_model(m),
                              ArenaOptions(water=WaterOptions(height=0.0)), model=m, n_envs=n, controller=c, buffer_size=T)
    sim.reset()
    d = sim.physics.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    return sim


def _walk(n, T):
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.model import salamander33
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = salamander33(contacts=True, limits=True, spawn_z=0.045)
    pairs = [(b, '') for b in m.body_names[1:] if b.endswith('_3')] + [('world', '')]
    data = AnimatData(m.timestep, T, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
    sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T), n_envs=n, data=data, buffer_size=T)
    sim.reset()
    rng = np.random.default_rng(5)
    q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 7:] += rng.uniform(-0.2, 0.2, (n, m.nq - 7)); q0[:, 2] = 0.03 + 0.01*rng.uniform(size=n)
    sim.physics.data.qpos[:] = torch.as_tensor(q0, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    return sim


@pytest.mark.parametrize('case', ['swim_wave', 'swim_network', 'walk'])
def test_save_load_resumes_bitwise(tmp_path, case):
    import torch
    n, T, cut = 6, 80, 33
    make = (lambda: _walk(n, T)) if case == 'walk' else (lambda: _swim(n, T, 'wave' if case == 'swim_wave' else 'network'))
    whole = make()
    whole.step_fused(cut); whole.step_fused(T - cut)
    first = make()
    first.step_fused(cut)
    ck = first.save_state(str(tmp_path/'state.npz'))
    del first
    resumed = make()
    resumed.step_fused(5)                                   # a context that has already moved on: load_state overwrites all of it
    resumed.load_state(ck)
    assert (resumed.task.iteration, resumed.task.sim_iteration) == (cut, cut)
    resumed.step_fused(T - cut)
    torch.cuda.synchronize()
    a, b = _state(whole), _state(resumed)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    ra, rb = _rows(whole), _rows(resumed)
    for k in ra:
        assert np.array_equal(ra[k], rb[k]), k              # the rows before the cut came from the file, the others from the resumed run
    assert int(np.abs(a['status']).sum()) == 0 and np.abs(ra['links']).max() > 0
    if case == 'walk':
        assert np.abs(ra['contacts'][..., 2]).max() > 1e-3 and np.abs(a['qacc_warmstart']).max() > 0
    if case == 'swim_network':
        for k in ('phase', 'amp', 'damp'):
            assert torch.equal(getattr(whole.task._controller, k), getattr(resumed.task._controller, k)), k


def test_load_state_refuses_another_batch(tmp_path):
    sim = _swim(4, 10)
    ck = sim.save_state(str(tmp_path/'s.npz'))
    other = _swim(6, 10)
    with pytest.raises(ValueError, match='checkpoint of 4 envs'):
        other.load_state(ck)


def test_controller_timestep_follows_the_task():
    """a NetworkController built without ``timestep=`` advances by task.timestep once per iteration, also when the
    model steps at timestep / num_sub_steps; an explicit mismatch is refused."""
    from farms_mujoco_amd.control import NetworkController, salamander_network
    from farms_mujoco_amd.model import salamander33
    from farms_mujoco_amd.options import AnimatOptions, ArenaOptions, SimulationOptions, WaterOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    h, S, n = 1e-3, 2, 2
    m = salamander33(timestep=h/S)

    def make(**kw):
        c = NetworkController(m, salamander_network(m), n, **kw)
        sim = Simulation.from_sdf(SimulationOptions(timestep=h, n_iterations=4, num_sub_steps=S), AnimatOptions.from_model(m),
                                  ArenaOptions(water=WaterOptions(height=0.0)), model=m, n_envs=n, controller=c, buffer_size=4)
        return sim, c
    sim, c = make()
    assert c.timestep == pytest.approx(h/S)
    sim.reset()
    assert c.timestep == pytest.approx(h)
    sim, c = make(timestep=h)
    sim.reset()
    assert c.timestep == pytest.approx(h)
    sim, c = make(timestep=h/S)
    with pytest.raises(ValueError, match='controller.timestep'):
        sim.reset()
