"""ExperimentTask / Simulation sequencing logic (no GPU): iteration counters, ring index, sub-step quirk,
kwargs strictness — mirrored from reference farms_mujoco/simulation/task.py."""
import numpy as np
import pytest

from farms_mujoco_amd.simulation.task import ExperimentTask, TaskCallback, duration2nit


class _FakePhysics:
    pass


class _Recorder(TaskCallback):
    def __init__(self, substep=False):
        super().__init__(substep=substep)
        self.before, self.after = [], []

    def before_step(self, task, action, physics): self.before.append((task.sim_iteration, task.iteration))
    def after_step(self, task, physics): self.after.append((task.sim_iteration, task.iteration))


def _task(**kw):
    t = ExperimentTask(base_link='b', n_iterations=kw.pop('n_iterations', 6), timestep=1e-3, **kw)
    t.update_sensors = lambda physics, links_only=False, swimming=None: t._sensor_calls.append((t.iteration % t.buffer_size, links_only))
    t._sensor_calls = []
    return t


def test_kwargs_strictness():
    with pytest.raises(AssertionError):
        ExperimentTask(base_link='b', n_iterations=1, timestep=1e-3, bogus=1)      # task.py:73
    assert duration2nit(1.0, 1e-3) == 1000


def test_counters_no_substeps():
    cb = _Recorder()
    t = _task(callbacks=[cb], buffer_size=4)
    for _ in range(6):
        t.before_step(None, _FakePhysics()); t.after_step(_FakePhysics())
    assert t.iteration == 6 and t.sim_iteration == 6
    assert [c[0] for c in t._sensor_calls] == [0, 1, 2, 3, 0, 1]          # ring index = iteration % buffer_size
    assert all(not lo for _, lo in t._sensor_calls)
    assert cb.before == [(i, i) for i in range(6)]
    assert t.get_termination(_FakePhysics()) == 1


def test_substep_quirk_matches_reference():
    """With substeps > 1 the reference advances `iteration` after the first sub-step of each group
    (task.py:352-355 tests (sim_iteration + 1) % substeps after incrementing)."""
    sub = 3
    cb, cbs = _Recorder(), _Recorder(substep=True)
    t = _task(n_iterations=4, substeps=sub, callbacks=[cb, cbs])
    its = []
    for _ in range(4*sub - 1):
        t.before_step(None, _FakePhysics()); t.after_step(_FakePhysics())
        its.append(t.iteration)
    assert its[:7] == [0, 1, 1, 1, 2, 2, 2]
    # full-step callback only on sim_iteration % substeps == 0, sub-step callback every step
    assert [s for s, _ in cb.before] == [0, 3, 6, 9]
    assert len(cbs.before) == 4*sub - 1
    # sensors: full rows on full steps, links_only rows on sub-steps because a callback asked for sub-steps
    assert [lo for _, lo in t._sensor_calls[:4]] == [False, True, True, False]


def test_substeps_past_the_last_iteration_write_nothing():
    """run() executes all n_iterations * substeps steps; the reference never executes the sub-steps whose task.iteration
    has reached n_iterations (its assert at task.py:170).  They run here, but write no rows and call no sub-step callback: ring index
    n_iterations % buffer_size would be row 0 of a full log."""
    sub, n_it = 3, 4
    cb, cbs = _Recorder(), _Recorder(substep=True)
    t = _task(n_iterations=n_it, substeps=sub, callbacks=[cb, cbs], buffer_size=n_it)
    for _ in range(n_it*sub):
        t.before_step(None, _FakePhysics()); t.after_step(_FakePhysics())
    assert t.sim_iteration == n_it*sub and t.iteration == n_it
    assert len(cbs.before) == n_it*sub - 1 and all(it < n_it for _, it in cbs.before)
    assert len(t._sensor_calls) == n_it*sub - 1
    assert [i for i, lo in t._sensor_calls if i == 0] == [0, 0]            # row 0: the first full step and its first sub-step, nothing later


def test_fusable_rules():
    class DevCb(TaskCallback):
        fusable = True
    assert _task().fusable()
    assert _task(callbacks=[DevCb()]).fusable()
    assert not _task(callbacks=[_Recorder()]).fusable()        # arbitrary host callbacks need per-step launches
    assert _task(callbacks=[DevCb()], substeps=2).fusable()     # round 4: the fused kernel sequences sub-steps itself (include/fmj.h)
    assert not _task(callbacks=[_Recorder(substep=True)], substeps=2).fusable()


def test_units_and_options():
    from farms_mujoco_amd.units import SimulationUnitScaling
    from farms_mujoco_amd.options import SimulationOptions, WaterOptions
    u = SimulationUnitScaling(meters=2.0, seconds=0.5, kilograms=3.0)
    assert u.velocity == 4.0 and u.acceleration == 8.0 and u.newtons == 24.0 and u.torques == 48.0
    assert u.angular_velocity == 2.0 and u.inertia == 12.0
    with pytest.raises(AssertionError):
        SimulationOptions(nope=1)
    with pytest.raises(AssertionError):
        WaterOptions(nope=1)
    assert SimulationOptions().integrator == 'Euler'


def test_synthetic_batch_is_keyed_by_global_index():
    """Env e's inputs depend only on its global index, so any contiguous sharding reproduces them."""
    from farms_mujoco_amd.model import salamander33, synthetic_batch
    m = salamander33()
    q, v, psi = synthetic_batch(m, 12, seed=3)
    q2, v2, psi2 = synthetic_batch(m, 5, seed=3, env_offset=4)
    assert np.array_equal(q[4:9], q2) and np.array_equal(psi[4:9], psi2)


def test_animat_data_log_round_trip(tmp_path):
    """AnimatData.to_file / from_file (reference simulation.py:200-203 writes simulation.hdf5; .npz when h5py is
    absent): arrays, names and the iteration cut survive."""
    import torch
    from farms_mujoco_amd.data import AnimatData
    d = AnimatData(1e-3, 6, 3, ['a', 'b'], ['j0'], contacts=[('a', ''), ('b', 'a')], device='cpu')
    g = torch.Generator().manual_seed(0)
    for k in ('links', 'joints', 'xfrc', 'contacts'):
        arr = getattr(d.sensors, k).array
        arr.copy_(torch.rand(arr.shape, generator=g))
    path = d.to_file(str(tmp_path/'simulation.hdf5'), iteration=4)
    back = AnimatData.from_file(path)
    assert back.timestep == 1e-3 and back.buffer_size == 4 and back.n_envs == 3
    assert back.sensors.links.names == ['a', 'b'] and back.sensors.joints.names == ['j0']
    for k in ('links', 'joints', 'xfrc', 'contacts'):
        assert torch.equal(getattr(back.sensors, k).array, getattr(d.sensors, k).array[:4])
    assert len(back.sensors.contacts.names) == 2


class _StatusPhysics:
    """Stands in for BatchedPhysics in the host-logic tests: counts steps, reports a bad env from step `bad_at` on."""
    def __init__(self, bad_at=None):
        self.steps, self.bad_at = 0, bad_at

    def step(self, n=1):
        self.steps += n

    def check_invalid_state(self):
        from farms_mujoco_amd.physics import PhysicsError
        if self.bad_at is not None and self.steps >= self.bad_at:
            raise PhysicsError('bad simulation state in 1 env(s); first env 0 status bits 2')


def _host_sim(n_iterations=30, handle_exceptions=False, bad_at=None, check_every=5, substeps=1):
    """Simulation with the device side stubbed out (constructor bypassed): sequencing and error policy only."""
    from farms_mujoco_amd.simulation.simulation import Simulation
    sim = Simulation.__new__(Simulation)
    sim.physics = _StatusPhysics(bad_at)
    sim.handle_exceptions = handle_exceptions
    sim.check_every = check_every
    sim.task = _task(n_iterations=n_iterations, substeps=substeps)
    sim._needs_reset = False
    return sim


def test_run_physics_error_policy():
    """reference simulation.py:153-161: run() logs and re-raises PhysicsError, or returns when handle_exceptions."""
    from farms_mujoco_amd.physics import PhysicsError
    sim = _host_sim(bad_at=12)
    with pytest.raises(PhysicsError):
        sim.run(fused=False)
    assert sim.physics.steps == 15 and sim.task.iteration == 15        # noticed at the first status look after step 12
    sim = _host_sim(bad_at=12, handle_exceptions=True)
    assert sim.run(fused=False) is None and sim.physics.steps == 15
    ok = _host_sim()
    ok.run(fused=False)
    assert ok.physics.steps == 30 and ok.task.iteration == 30


def test_iterator_physics_error_policy():
    """reference simulation.py:164-179: iterator() yields the iteration before stepping it and always re-raises."""
    from farms_mujoco_amd.physics import PhysicsError
    sim = _host_sim(bad_at=7, handle_exceptions=True, substeps=2, n_iterations=10)
    seen = []
    with pytest.raises(PhysicsError):
        for it in sim.iterator(show_progress=False, verbose=False):
            seen.append(it)
    assert seen == list(range(5)) and sim.physics.steps == 10           # 5 iterations x 2 sub-steps, checked every 5
    ok = _host_sim(n_iterations=4)
    assert list(ok.iterator()) == [0, 1, 2, 3] and ok.physics.steps == 4


def test_unsupported_solver_options_are_refused():
    """integrator / cone / solver other than Euler / pyramidal / PGS, CG or Newton must not silently run different physics
    (the reference forwards them to MuJoCo, mjcf.py:1342-1365)."""
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.mjcf import check_supported_options
    check_supported_options(SimulationOptions())
    check_supported_options(SimulationOptions(solver='Newton'))        # round 3: the device has MuJoCo's Newton and CG solvers too
    check_supported_options(SimulationOptions(solver='CG'))
    check_supported_options(SimulationOptions(solver='Newton', cone='elliptic'))
    check_supported_options(SimulationOptions(integrator='implicitfast'))     # round 4
    check_supported_options(SimulationOptions(noslip_iterations=3), compile_only=True)      # the model compiler forwards it (MJCF export)
    check_supported_options(None)
    check_supported_options(SimulationOptions(cone='elliptic', solver='PGS'))     # round 5: MuJoCo's elliptic PGS on the device
    check_supported_options(SimulationOptions(noslip_iterations=3))               # round 5: the noslip post-pass
    check_supported_options(SimulationOptions(integrator='RK4'))                  # round 5: mj_RungeKutta through fmj_step
    for kw, word in ((dict(integrator='implicit'), 'Coriolis'), (dict(solver='SOR'), 'pgs')):
        with pytest.raises(NotImplementedError, match=word):      # a refusal says why
            check_supported_options(SimulationOptions(**kw))


def test_task_spec_hooks_and_restart_flag():
    """reference task.py:371-412: the spec hooks fan out to the callbacks, action_spec concatenates what they return;
    restart needs an application (task.py:91-94)."""
    class Cb(TaskCallback):
        def __init__(self): super().__init__(); self.calls = []
        def action_spec(self, task, physics): self.calls.append('a'); return ['u']
        def step_spec(self, task, physics): self.calls.append('s')
        def get_observation(self, task, physics): self.calls.append('o')
        def observation_spec(self, task, physics): self.calls.append('os')
    a, b = Cb(), Cb()
    t = _task(callbacks=[a, b])
    assert t.action_spec(_FakePhysics()) == ['u', 'u']
    t.step_spec(_FakePhysics()); t.get_observation(_FakePhysics()); t.observation_spec(_FakePhysics())
    assert a.calls == b.calls == ['a', 's', 'o', 'os']
    closed = []
    t2 = _task(n_iterations=2)
    t2.set_app(type('App', (), {'close': lambda self: closed.append(1)})())
    for _ in range(2):
        t2.before_step(None, _FakePhysics()); t2.after_step(_FakePhysics())
    assert closed == [1]                                                  # task.py:358-364: the app is closed at the end
    with pytest.raises(AssertionError):
        ExperimentTask(base_link='b', n_iterations=1, timestep=1e-3, restart=True).initialize_episode(_FakePhysics())
