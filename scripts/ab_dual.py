"""A/B of the two generations of the two-env kernel on the same inputs (FMJ_DUAL=1 -> fmj_dual.inc, default -> fmj_dual2.inc):
python scripts/ab_dual.py [n_envs] [n_steps].  Prints the largest difference per field and per env."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from farms_mujoco_amd.model import salamander33, synthetic_batch
from farms_mujoco_amd.physics import BatchedPhysics

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = salamander33()
qpos, qvel, psi = synthetic_batch(m, n)
rng = np.random.default_rng(0)
qvel = qvel + 0.1*rng.normal(size=qvel.shape)
ctrl = 0.2*rng.normal(size=(n, m.nu))
out = {}
for gen in ('1', '2'):
    os.environ['FMJ_DUAL'] = gen
    phys = BatchedPhysics(m, n)
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    d.ctrl[:] = torch.as_tensor(ctrl, dtype=torch.float32)
    phys.step(T)
    torch.cuda.synchronize()
    out[gen] = {k: getattr(d, k).cpu().numpy().copy() for k in ('qpos', 'qvel', 'qacc', 'xpos', 'xquat', 'xipos', 'sensordata', 'status', 'time')}
for k in out['1']:
    a, b = out['1'][k].astype(np.float64), out['2'][k].astype(np.float64)
    a = a.reshape(n, -1); b = b.reshape(n, -1)
    with np.errstate(invalid='ignore'):
        err = np.abs(a - b).max(1)/np.maximum(np.abs(a).max(1), 1e-30)
    print(f'{k:11s} per-env rel diff', np.array2string(err, precision=2), 'nan:', np.isnan(b).any(1).astype(int))
    if k == 'qacc':
        with np.errstate(invalid='ignore'):
            e = np.abs(a - b)/np.maximum(np.abs(a).max(1, keepdims=True), 1e-30)
        for env in range(min(n, 4)):
            print('   qacc env', env, 'worst dofs', np.argsort(-np.nan_to_num(e[env], nan=1e9))[:6], np.array2string(np.sort(np.nan_to_num(e[env], nan=1e9))[::-1][:6], precision=2))
