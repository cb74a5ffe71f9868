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
