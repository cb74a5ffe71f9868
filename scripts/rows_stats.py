"""Row-count statistics of the walking workload (BASELINE configs[3]): how the two-env constraint kernel's three modes would split
the (wave, step) pairs - PAIR (both envs <= 32 rows), SOLO (some env 33..64), BAIL (more than 64 rows / 16 contacts).
Rows of an env = active limit sides + 4 * contacts, sampled at the end of every launch; waves pair neighbours of the launch order
(envs sorted by contact count).  usage (GPU box): python scripts/rows_stats.py [workload] [launch length]"""
import os
import sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
import numpy as np
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'walk'
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sim, m, _ = bench.build_sim(4096, 1 << 30, L, 0, 'cuda:0', wl)
lim = np.asarray(m.jnt_limited) != 0
lo = torch.as_tensor(np.asarray(m.jnt_range)[lim, 0], device='cuda:0', dtype=torch.float32)
hi = torch.as_tensor(np.asarray(m.jnt_range)[lim, 1], device='cuda:0', dtype=torch.float32)
qadr = torch.as_tensor(np.asarray(m.jnt_qposadr)[lim], device='cuda:0')
for k in range(20):
    sim.step_fused(L)
    d = sim.physics.data
    q = d.qpos[:, qadr]
    nlim = ((q - lo) < 0).sum(1) + ((hi - q) < 0).sum(1)
    nefc = (nlim + 4*d.ncon).cpu().numpy()
    order = np.argsort(-d.ncon.cpu().numpy(), kind='stable')
    a, b = nefc[order[0::2]], nefc[order[1::2]]
    pair = ((a <= 32) & (b <= 32)).mean(); bail = ((a > 64) | (b > 64)).mean()
    a2, b2 = nefc[0::2], nefc[1::2]
    pair_unsorted = ((a2 <= 32) & (b2 <= 32)).mean()
    print(f'steps {L*(k+1):5d}: rows mean {nefc.mean():5.1f} p50 {np.median(nefc):4.0f} p90 {np.percentile(nefc, 90):4.0f} max {nefc.max():3d}  <=32: {(nefc <= 32).mean()*100:5.1f}%  '
          f'waves PAIR {pair*100:5.1f}% SOLO {(1 - pair - bail)*100:5.1f}% BAIL {bail*100:5.1f}%  (unsorted pairing: PAIR {pair_unsorted*100:5.1f}%)  nlim mean {nlim.float().mean().item():.2f}')
