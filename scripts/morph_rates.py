import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
for morph in ('eel', 'centipede', 'salamander33'):
    sim, m, _ = bench.build_sim(2048, 3000, 100, 0, 'cuda:0', morphology=morph)
    for _ in range(5): sim.step_fused(100)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): sim.step_fused(100)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(morph, 'nv', m.nv, 'nbody', m.nbody, f'{2048*1000/dt/1e6:.1f} M env-steps/s', sim.physics.kernel_info())
