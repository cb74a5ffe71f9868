"""A/B helper: the state a workload reaches after some fused launches, saved for a bitwise comparison between two builds of the library
(FMJ_SO=<path> python scripts/ab_state.py walk out.npz [launches] [steps per launch]; scripts/ab_state.py --diff a.npz b.npz)."""
import os
import sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')))
import numpy as np

if sys.argv[1] == '--diff':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(f'{k}: {"bitwise equal" if same else "DIFFERENT"}  max abs diff {np.nanmax(d):.3e}  envs differing {int((d.reshape(d.shape[0], -1) > 0).any(1).sum())} of {d.shape[0]}')
    sys.exit(0)
import torch
import bench
wl, out = sys.argv[1], sys.argv[2]
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = int(sys.argv[4]) if len(sys.argv) > 4 else 100
sim, m, _ = bench.build_sim(int(os.environ.get('FMJ_ENVS', '4096')), 1 << 30, L, 0, 'cuda:0', wl)
for _ in range(launches):
    sim.step_fused(L)
torch.cuda.synchronize()
d = sim.physics.data
np.savez(out, qpos=d.qpos.cpu().numpy(), qvel=d.qvel.cpu().numpy(), ncon=d.ncon.cpu().numpy())
print('saved', out)
