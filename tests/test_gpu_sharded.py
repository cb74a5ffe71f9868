"""The in-process multi-device object (SURVEY section 8(b) "device(s)", 8(e)): ``ShardedSimulation`` splits
one batch into contiguous env ranges, one context + stream per device, no collective.  A one-GPU box names its device several times:
the shards then run side by side on their own HIP streams, which exercises the same code path as G devices.  Results are bitwise
those of one shard holding every env - swimming (configs[1] / [2]), walking (configs[3]) and the mixed batch, bucket then split
(configs[4])."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _swim_factory(T, morphology='salamander33'):
    import bench

    def factory(lo, hi, device):
        return bench.build_sim(hi - lo, T, T, lo, str(device), morphology=morphology)[0]
    return factory


def _rows(sim):
    s = sim.task.data.sensors
    return {k: getattr(s, k).array.cpu().numpy() for k in ('links', 'joints', 'xfrc')}


@pytest.mark.parametrize('n_shards', [2, 3])
def test_sharded_swim_equals_one_shard_bitwise(n_shards):
    import torch
    from farms_mujoco_amd.sharding import ShardedSimulation, shard_range
    n, T = 22, 60
    one = ShardedSimulation(_swim_factory(T), n, ['cuda:0'])
    many = ShardedSimulation(_swim_factory(T), n, ['cuda:0']*n_shards)
    assert many.ranges == [shard_range(n, g, n_shards) for g in range(n_shards)]
    one.run(chunk=25); many.run(chunk=25)
    one.synchronize(); many.synchronize()
    for f in ('qpos', 'qvel', 'time', 'sensordata'):
        assert np.array_equal(one.gather(f), many.gather(f)), f
    assert int(np.abs(many.gather('status')).sum()) == 0
    whole = _rows(one.shards[0])
    for (lo, hi), sh in zip(many.ranges, many.shards):
        part = _rows(sh)
        for k in whole:
            assert np.array_equal(whole[k][:, lo:hi], part[k]), k
    assert np.abs(whole['xfrc']).max() > 0


def test_sharded_walk_equals_one_shard_bitwise():
    import torch
    from farms_mujoco_amd.data import AnimatData
    from farms_mujoco_amd.model import salamander33
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.sharding import ShardedSimulation
    from farms_mujoco_amd.simulation.simulation import Simulation
    m = salamander33(contacts=True, limits=True, spawn_z=0.045)
    n, T = 12, 80
    rng = np.random.default_rng(11)
    q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 7:] += rng.uniform(-0.2, 0.2, (n, m.nq - 7)); q0[:, 2] = 0.03 + 0.01*rng.uniform(size=n)
    pairs = [(b, '') for b in m.body_names[1:] if b.endswith('_3')]

    def factory(lo, hi, device):
        data = AnimatData(m.timestep, T, hi - lo, m.body_names[1:], m.hinge_joint_names(), contacts=pairs, device=str(device))
        # order_by_contacts pairs neighbours of a per-shard order in the two-env kernel; a wave's two envs never influence each other
        # (test_two_env_kernel_modes_do_not_depend_on_the_partner), so the sharding cannot show in the results
        sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T), n_envs=hi - lo, data=data, buffer_size=T,
                         device=str(device))
        sim.reset()
        sim.physics.data.qpos[:] = torch.as_tensor(q0[lo:hi], dtype=torch.float32)
        sim.physics.forward(disable_actuation=True)
        return sim
    one = ShardedSimulation(factory, n, ['cuda:0'])
    two = ShardedSimulation(factory, n, ['cuda:0', 'cuda:0'])
    one.run(chunk=40); two.run(chunk=40)
    one.synchronize(); two.synchronize()
    assert int((one.gather('status') & ~8).sum()) == 0
    for f in ('qpos', 'qvel', 'qacc_warmstart', 'ncon'):
        assert np.array_equal(one.gather(f), two.gather(f)), f
    c1 = one.shards[0].task.data.sensors.contacts.array.cpu().numpy()
    c2 = np.concatenate([s.task.data.sensors.contacts.array.cpu().numpy() for s in two.shards], axis=1)
    assert np.array_equal(c1, c2) and np.abs(c1[..., 2]).max() > 1e-3


def test_sharded_mixed_bucket_then_split():
    """configs[4] across devices: every morphology bucket is split like the whole batch (eels [lo, hi) of the eels, centipedes [lo, hi)
    of the centipedes on shard g); per bucket the result is bitwise the unsplit bucket's."""
    import bench
    from farms_mujoco_amd.sharding import ShardedSimulation, shard_range
    from farms_mujoco_amd.simulation.buckets import BucketedSimulation
    n_eel, n_cen, T, G = 10, 6, 40, 2

    def factory_for(G_):
        def factory(lo, hi, device):            # lo, hi index the whole batch of n_eel + n_cen envs; the buckets are split by the same rule
            g = [r for r in range(G_) if shard_range(n_eel + n_cen, r, G_) == (lo, hi)][0]
            (el, eh), (cl, ch) = shard_range(n_eel, g, G_), shard_range(n_cen, g, G_)
            return BucketedSimulation([bench.build_sim(eh - el, T, T, el, str(device), morphology='eel')[0],
                                       bench.build_sim(ch - cl, T, T, n_eel + cl, str(device), morphology='centipede')[0]])
        return factory
    one = ShardedSimulation(factory_for(1), n_eel + n_cen, ['cuda:0'])
    two = ShardedSimulation(factory_for(G), n_eel + n_cen, ['cuda:0']*G)
    for sh in (one, two):
        sh.run(chunk=20); sh.synchronize()
    for f in ('qpos', 'qvel'):
        a, b = one.gather(f), two.gather(f)
        assert len(a) == len(b) == 2
        for x, y in zip(a, b):
            assert np.array_equal(x, y), f
    assert one.gather('qpos')[0].shape[0] == n_eel and one.gather('qpos')[1].shape[0] == n_cen
