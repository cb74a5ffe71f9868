import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for n in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else (1024, 2048, 4096, 6144, 8192, 16384, 65536):
    sim, m, _ = bench.build_sim(n, 1 << 30, 100, 0, 'cuda:0')
    for _ in range(10): sim.step_fused(100)
    torch.cuda.synchronize()
    reps = max(10, min(200, int(0.3/(n*100/250e6))))
    t0 = time.perf_counter()
    for _ in range(reps): sim.step_fused(100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0)/(100*reps)
    del sim
    print(f'n_envs {n:6d}  us/step {dt*1e6:8.2f}  env-steps/s {n/dt/1e6:8.2f} M', flush=True)
