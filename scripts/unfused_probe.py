"""What an iteration of Simulation.run(fused=False) costs (4096 swimmers): wall time per iteration, and - under rocprofv3 --kernel-trace
--stats - the two kernels' durations.  python scripts/unfused_probe.py [n_envs] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sim, m, _ = bench.build_sim(n, T + 50, 100, 0, 'cuda:0')
sim.task.n_iterations = 50; sim.task.sim_iterations = 50
sim.run(fused=False)
sim.task.n_iterations = T + 50; sim.task.sim_iterations = T + 50
torch.cuda.synchronize()
t0 = time.perf_counter()
sim.run(fused=False)
t1 = time.perf_counter()            # host done queueing
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{n} envs, {T} iterations: host queued in {(t1 - t0)/T*1e6:.1f} us / iteration, device done after {(t2 - t0)/T*1e6:.1f} us / iteration '
      f'-> {n*T/(t2 - t0)/1e6:.1f} M env-steps/s')
