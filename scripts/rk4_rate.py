"""Throughput of fmj_step with the Euler and the RK4 integrator on 4096 swimming salamanders (GPU box): python scripts/rk4_rate.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from farms_mujoco_amd.model import salamander33, synthetic_batch
from farms_mujoco_amd.physics import BatchedPhysics
for integ, name in ((0, 'Euler (fmj_step, one launch per step)'), (1, 'RK4 (fmj_step, eight launches per step)')):
    m = salamander33(); m.integrator = integ
    n = 4096
    phys = BatchedPhysics(m, n, 'cuda:0')
    q, v, _ = synthetic_batch(m, n, seed=0)
    phys.data.qpos[:] = torch.as_tensor(q, dtype=torch.float32); phys.data.qvel[:] = torch.as_tensor(v, dtype=torch.float32)
    phys.step(20); torch.cuda.synchronize()
    t = time.perf_counter(); phys.step(200); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(name, f'{n*200/dt/1e6:.1f} M env-steps/s')
