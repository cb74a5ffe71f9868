import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for n in (256, 1024, 2048, 3072, 4096, 5120, 8192, 16384):
    sim, m, _ = bench.build_sim(n, 700, 100, 0, 'cuda:0')
    sim.step_fused(100); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): sim.step_fused(100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0)/500
    print(f'n_envs {n:6d}  us/step {dt*1e6:8.2f}  env-steps/s {n/dt/1e6:8.2f} M', flush=True)
