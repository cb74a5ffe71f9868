"""Long-horizon soak on the GPU: many fused launches of the bench workloads, then invariants (no warning bits, finite
state, unit root quaternions, animals still inside a sane box).  usage: python scripts/soak.py [steps=50000]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
L = int(os.environ.get('FMJ_SOAK_LAUNCH', '1000'))      # steps per launch (= ring length)
workloads = [(w, 4096) for w in os.environ['FMJ_SOAK'].split(',')] if 'FMJ_SOAK' in os.environ else (('swim', 4096), ('walk', 4096), ('swim', 8191))
for workload, n in workloads:
    sim, m, _ = bench.build_sim(n, 1 << 30, L, 0, 'cuda:0', workload)
    t0 = time.perf_counter()
    for _ in range(max(1, (steps if workload == 'swim' else steps//4)//L)):
        sim.step_fused(L)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d = sim.physics.data
    bad = int((d.status != 0).sum())
    q = d.qpos
    fin = bool(torch.isfinite(q).all()) and bool(torch.isfinite(d.qvel).all())
    qn = torch.linalg.norm(q[:, 3:7], dim=1)
    rows = sim.task.data.sensors.links.array
    print(f'{workload} x{n}: {sim.task.iteration} steps in {dt:.1f} s, envs with status bits {bad}, finite {fin}, '
          f'|quat| in [{qn.min().item():.7f}, {qn.max().item():.7f}], z in [{q[:, 2].min().item():.3f}, {q[:, 2].max().item():.3f}], '
          f'|xy| max {q[:, :2].abs().max().item():.2f}, |qvel| max {d.qvel.abs().max().item():.2f}, rows finite {bool(torch.isfinite(rows).all())}', flush=True)
    assert bad == 0 and fin
    del sim
