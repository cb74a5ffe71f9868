import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench, numpy as np
sim, m, _ = bench.build_sim(4096, 1000, 100, 0, 'cuda:0', 'walk')
import time
for k in range(10):
    torch.cuda.synchronize(); t0 = time.time()
    sim.step_fused(100); torch.cuda.synchronize(); dt = time.time() - t0
    nc = sim.physics.data.ncon.cpu().numpy()
    print(f'steps {100*(k+1):4d}: {dt*1e3:6.1f} ms  ncon mean {nc.mean():5.1f} p50 {np.median(nc):4.0f} p90 {np.percentile(nc,90):4.0f} max {nc.max():3d}  >15 contacts: {(nc>15).mean()*100:5.1f}%  z mean {sim.physics.data.qpos[:,2].mean().item():.3f}')
