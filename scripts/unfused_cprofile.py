import os, sys, cProfile, pstats
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
n, T = 4096, 400
sim, m, _ = bench.build_sim(n, T + 50, 100, 0, 'cuda:0')
sim.task.n_iterations = 50; sim.task.sim_iterations = 50
sim.run(fused=False)
sim.task.n_iterations = T + 50; sim.task.sim_iterations = T + 50
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
sim.run(fused=False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
