"""Diagnostic: iterations and line-search evaluations per step of the device Newton / CG solvers on the trot (fmj_step_debug),
next to the oracle's iteration counts on the same walk."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from farms_mujoco_amd.model import salamander33, SOLVERS
from farms_mujoco_amd.physics import BatchedPhysics
from oracle import oracle
from test_gpu_contacts import _trot_tape
for solver in sys.argv[1:] or ['newton', 'cg']:
    m = salamander33(contacts=True, limits=True, spawn_z=0.045)
    m.solver = SOLVERS[solver]; m.solver_iterations = 100
    n, T = 8, 300
    tape = _trot_tape(m, n, T)
    phys = BatchedPhysics(m, n)
    d = phys.data
    its, evs = [], []
    q = np.tile(m.qpos0, (n, 1)); v = np.zeros((n, m.nv)); w = np.zeros((n, m.nv)); oits = []
    for t in range(T):
        d.ctrl[:] = torch.as_tensor(tape[t], dtype=torch.float32)
        _, imp = phys.step_debug()
        imp = imp.cpu().numpy()
        its.append(imp[:, 0]); evs.append(imp[:, 1])
        o = oracle.step_tf(m, q, v, ctrl=tape[t], warmstart=w, want_AR=False)
        q, v, w = o['qpos'], o['qvel'], o['warmstart']; oits.append(o['iterations'])
    its = np.array(its); evs = np.array(evs); oits = np.array(oits)
    e = np.abs(d.qpos.cpu().numpy() - q).max(1)
    print(f'{solver}: device iterations median {np.median(its)} mean {its.mean():.2f} max {its.max()}, line-search evaluations mean {evs.mean():.1f}; '
          f'oracle iterations median {np.median(oits)} mean {oits.mean():.2f} max {oits.max()}; qpos err median {np.median(e):.2e} max {e.max():.2e}')
