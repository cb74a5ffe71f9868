"""Environment sharding across GPUs: independent envs, contiguous ranges, NO data-path collective.

``torch.distributed`` is only used for the barrier and the max-over-ranks wall time of the benchmark contract."""
import os


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous env range [lo, hi) owned by ``rank``: GPU g of G owns [g*N/G, (g+1)*N/G)."""
    assert 0 <= rank < world
    return n_total*rank//world, n_total*(rank + 1)//world


def dist_env():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a python float over the default process group (identity when not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class ShardedSimulation:
    """One logical batch of independent environments over several devices, driven from ONE process (``Simulation``
    takes ``device(s)``).  Device g of G owns the contiguous env range ``shard_range(n_envs, g, G)``; a shard is whatever
    ``factory(env_lo, env_hi, device)`` returns for that range - a :class:`~farms_mujoco_amd.simulation.simulation.Simulation`, or a
    :class:`~farms_mujoco_amd.simulation.buckets.BucketedSimulation` for mixed morphologies (BASELINE configs[4]: every morphology
    bucket is split across the devices the same way, "bucket, then split") - with its inputs keyed by GLOBAL env index, so that
    results do not depend on G (tests/test_gpu_sharded.py: bitwise against one shard holding every env).

    No thread per device and no collective: launches are asynchronous, so one host thread queues a chunk on every device in turn
    (its own context and stream each) and only then looks at the status words.  ``devices`` may name one device several times
    (shards side by side on their own HIP streams: how the multi-device path is tested on a one-GPU box)."""

    def __init__(self, factory, n_envs: int, devices):
        import torch
        self.devices = [torch.device(d) for d in devices]
        assert self.devices, 'at least one device'
        self.n_envs = int(n_envs)
        G = len(self.devices)
        self.ranges = [shard_range(self.n_envs, g, G) for g in range(G)]
        self.shards = []
        for (lo, hi), dev in zip(self.ranges, self.devices):
            with torch.cuda.device(dev):
                self.shards.append(factory(lo, hi, dev) if hi > lo else None)
        # shards that share a device get a stream each (and the caller's stream is forked / joined around them)
        shared = len({str(d) for d in self.devices}) < G
        self._streams = [torch.cuda.Stream(device=d) if shared else None for d in self.devices]

    def _each(self):
        return [(s, d, st) for s, d, st in zip(self.shards, self.devices, self._streams) if s is not None]

    def _on(self, fn):
        """fn(shard) on every shard, each on its own device (and stream, where shards share a device); returns the results."""
        import torch
        out, joins = [], []
        for s, d, st in self._each():
            with torch.cuda.device(d):
                if st is None:
                    out.append(fn(s))
                else:
                    cur = torch.cuda.current_stream(d)
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        out.append(fn(s))
                    joins.append((cur, st))
        for cur, st in joins:
            cur.wait_stream(st)
        return out

    def reset(self):
        self._on(lambda s: s.reset() if hasattr(s, 'reset') else [b.reset() for b in s.simulations])

    def step_fused(self, n_steps: int) -> int:
        """``n_steps`` iterations on every shard: one fused launch per shard (and bucket), queued back to back from this thread."""
        return max(self._on(lambda s: s.step_fused(n_steps)))

    def check_invalid_state(self):
        import torch
        for s, d, _ in self._each():
            with torch.cuda.device(d):
                (s.physics if hasattr(s, 'physics') else s).check_invalid_state()

    def synchronize(self):
        import torch
        for d in {str(d): d for d in self.devices}.values():
            torch.cuda.synchronize(d)

    def run(self, chunk=None):
        """The headless loop of reference simulation.py:148-161 over all shards: chunks of fused launches, every device busy at once,
        status words read once per chunk."""
        sims = [x for s, _, _ in self._each() for x in (s.simulations if hasattr(s, 'simulations') else [s])]
        for s in sims:
            if s._needs_reset:
                s.reset()
            assert s.task.fusable(), 'ShardedSimulation.run drives fused launches; run a shard with host callbacks through its own Simulation.run'
        task = sims[0].task
        chunk = min(chunk or task.buffer_size, task.buffer_size)
        while task.sim_iteration < task.sim_iterations:
            self.step_fused(chunk)
            self.check_invalid_state()

    def gather(self, field: str):
        """A ``physics.data`` field of the whole batch on the host, envs in global order (tests / checkpoints; no device traffic between
        shards).  Bucketed shards (mixed morphologies) return a list with one array per bucket."""
        import numpy as np
        per_bucket = None
        for s, _, _ in self._each():
            sims = s.simulations if hasattr(s, 'simulations') else [s]
            if per_bucket is None:
                per_bucket = [[] for _ in sims]
            for parts, x in zip(per_bucket, sims):
                parts.append(getattr(x.physics.data, field).detach().cpu().numpy())
        out = [np.concatenate(p) for p in per_bucket]
        return out if hasattr(self._each()[0][0], 'simulations') else out[0]
