"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): every rank owns a contiguous env range, inputs
are keyed by global env index, results are independent of the sharding, timing is MAX-reduced.  The per-env
work here is done by the CPU oracle (this test has no GPU); on the GPU box each rank runs the HIP path."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from farms_mujoco_amd.sharding import shard_range, max_over_ranks


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from farms_mujoco_amd.model import salamander33, synthetic_batch, wave_controller_params
    from oracle import oracle
    m = salamander33()
    lo, hi = shard_range(n_total, rank, world)
    qpos, qvel, psi = synthetic_batch(m, hi - lo, seed=1, env_offset=lo)
    amp, lag = wave_controller_params(m)
    ctrl = amp[None, :]*np.sin(-lag[None, :] + psi[:, None])
    o = oracle.step(m, qpos, qvel, ctrl=ctrl, n_steps=5)
    dist.barrier()
    tmax = max_over_ranks(float(rank + 1))          # stands in for the per-rank wall time
    gathered = [torch.zeros(h - l, m.nq, dtype=torch.float64) for l, h in (shard_range(n_total, r, world) for r in range(world))]
    dist.all_gather(gathered, torch.from_numpy(o['qpos']))      # test-only gather; the product path has none
    if rank == 0:
        out.put((tmax, torch.cat(gathered).numpy()))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_rank():
    from farms_mujoco_amd.model import salamander33, synthetic_batch, wave_controller_params
    from oracle import oracle
    oracle.build()
    n_total, world = 10, 2
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, out)) for r in range(world)]
    for p in procs: p.start()
    tmax, qpos_sharded = out.get(timeout=120)
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert tmax == float(world)                     # MAX over ranks
    m = salamander33()
    qpos, qvel, psi = synthetic_batch(m, n_total, seed=1)
    amp, lag = wave_controller_params(m)
    ref = oracle.step(m, qpos, qvel, ctrl=amp[None, :]*np.sin(-lag[None, :] + psi[:, None]), n_steps=5)
    assert np.array_equal(ref['qpos'], qpos_sharded)      # bitwise: sharding does not change any env
