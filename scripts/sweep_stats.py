"""How many PGS sweeps a walking step runs (BASELINE configs[3]): the sweeps until MuJoCo's stopping test (improvement * scale <
tolerance) fires, per env, sampled with fmj_step_debug at several points of a walk.  usage (GPU box): python scripts/sweep_stats.py [workload]"""
import os
import sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
import numpy as np
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'walk'
n = 4096
sim, m, _ = bench.build_sim(n, 1 << 30, 100, 0, 'cuda:0', wl)
for k in range(8):
    sim.step_fused(100)
    rows, imp = sim.physics.step_debug()
    imp = imp.cpu().numpy()
    ran = (~np.isnan(imp)).sum(1)                 # sweeps written = sweeps run
    kind = rows[:, :, 5].cpu().numpy() if rows.shape[2] > 5 else None
    nefc = (rows[:, :, 2].cpu().numpy() != 0).sum(1)           # R of a live row is > 0
    h = np.bincount(ran, minlength=imp.shape[1] + 1)
    print(f'after {100*(k+1)+k:4d} steps: sweeps mean {ran.mean():5.1f} p50 {np.median(ran):3.0f} p90 {np.percentile(ran, 90):3.0f} at cap ({imp.shape[1]}): {(ran >= imp.shape[1]).mean()*100:5.1f}%   '
          f'rows mean {nefc.mean():5.1f};  sweeps by rows: ' + ' '.join(f'{lo}-{lo+7}:{ran[(nefc >= lo) & (nefc < lo + 8)].mean() if ((nefc >= lo) & (nefc < lo + 8)).any() else 0:.0f}' for lo in range(0, 64, 8)))
    pairs = np.maximum(ran[0::2], ran[1::2])
    print(f'      per wave (max of the two envs): mean {pairs.mean():5.1f};  histogram of sweeps (x5): ' + ' '.join(str(h[i:i+5].sum()) for i in range(0, len(h), 5)))
