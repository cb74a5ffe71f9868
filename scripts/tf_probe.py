"""Diagnostic: teacher-forced per-step errors of the HIP constraint solve along the oracle's trot (any solver / cone):
python scripts/tf_probe.py newton elliptic"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from farms_mujoco_amd.model import salamander33, SOLVERS, CONES
from farms_mujoco_amd.physics import BatchedPhysics
from oracle import oracle
from test_gpu_contacts import _trot_tape
solver, cone = (sys.argv[1:] + ['newton', 'elliptic'])[:2]
m = salamander33(contacts=True, limits=True, spawn_z=0.045)
m.solver = SOLVERS[solver]; m.cone = CONES[cone]
if solver != 'pgs': m.solver_iterations = 100
n, T = 8, int(os.environ.get('T', '300'))
tape = _trot_tape(m, n, T)
phys = BatchedPhysics(m, n); d = phys.data
f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=d.qpos.device)
r64 = lambda t: t.cpu().numpy().astype(np.float64)
q = np.tile(m.qpos0, (n, 1)); v = np.zeros((n, m.nv)); w = np.zeros((n, m.nv))
worst = []; floor = []
from parity_metrics import group_relerr, qvel_groups
groups = qvel_groups(m)
for t in range(T):
    d.qpos[:] = f32(q); d.qvel[:] = f32(v); d.qacc_warmstart[:] = f32(w); d.ctrl[:] = f32(tape[t])
    q32, v32, w32, c32 = r64(d.qpos), r64(d.qvel), r64(d.qacc_warmstart), r64(d.ctrl)
    rows, imp = phys.step_debug()
    o = oracle.step_tf(m, q32, v32, ctrl=c32, warmstart=w32, want_AR=False)
    with oracle.fp32_state():
        fl = oracle.step_tf(m, q32, v32, ctrl=c32, warmstart=w32, want_AR=False)
    rows = r64(rows); imp = r64(imp); qv = r64(d.qvel)
    for e in range(n):
        ne = int(o['nefc'][e])
        if int(d.ncon[e]) != int(o['ncon'][e]):
            print(t, e, 'ncon differs', int(d.ncon[e]), int(o['ncon'][e])); continue
        fs = max(np.abs(o['efc'][e, :ne, 0]).max(), 1e-2) if ne else 1.0
        ef = np.abs(rows[e, :ne, 4] - o['efc'][e, :ne, 0]).max()/fs if ne else 0.0
        ev = np.abs(qv[e] - o['qvel'][e]).max()/max(np.abs(o['qvel'][e]).max(), 1e-3)
        worst.append((ef, ev, t, e, imp[e, 0], o['iterations'][e]))
        floor.append((group_relerr(qv[e], o['qvel'][e], groups), group_relerr(fl['qvel'][e], o['qvel'][e], groups),
                      np.abs(qv[e] - o['qvel'][e]).max(), np.abs(fl['qvel'][e] - o['qvel'][e]).max()))
    tch = oracle.step_tf(m, q, v, ctrl=tape[t], warmstart=w, want_AR=False)
    q, v, w = tch['qpos'], tch['qvel'], tch['warmstart']
W = np.array(worst)
print(f'{solver} {cone}: per-step force err median {np.median(W[:,0]):.2e} 99% {np.percentile(W[:,0],99):.2e} max {W[:,0].max():.2e}; qvel err median {np.median(W[:,1]):.2e} max {W[:,1].max():.2e}')
F = np.array(floor)
for k, nm in enumerate(('qvel per component HIP', 'qvel per component fp32-state floor', 'qvel abs HIP', 'qvel abs floor')):
    print(f'  {nm}: median {np.median(F[:,k]):.2e} 90% {np.percentile(F[:,k],90):.2e} 99% {np.percentile(F[:,k],99):.2e} max {F[:,k].max():.2e}')
for r in W[np.argsort(-W[:, 0])[:8]]:
    print('  force err %.2e qvel err %.2e step %d env %d device iters %g oracle iters %g' % tuple(r))
