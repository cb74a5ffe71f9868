"""Debug: which env of test_two_env_kernel_modes_do_not_depend_on_the_partner changes with its partner, at which step, by how much."""
import os, sys
root = os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np, torch
import test_gpu_contacts as tc
from oracle import oracle
from farms_mujoco_amd.physics import BatchedPhysics
m = tc._walker()
base = np.tile(m.qpos0, (1, 1))[0]
def state(z, seed):
    r = np.random.default_rng(seed)
    q = base.copy(); q[7:] += r.uniform(-0.05, 0.05, m.nq - 7); q[2] = z
    return q
cands = [state(z, 100 + i) for i, z in enumerate((0.045, 0.03, 0.0175, 0.0165, 0.0155, 0.0145, 0.012, 0.01))]
nefc = [oracle.forward_debug(m, q.astype(np.float32).astype(np.float64), np.zeros(m.nv), ctrl=np.zeros(m.nu))['nefc'] for q in cands]
light = [q for q, ne in zip(cands, nefc) if 0 < ne <= 32][:2]
medium = [q for q, ne in zip(cands, nefc) if 32 < ne <= 64][:2]
heavy = [q for q, ne in zip(cands, nefc) if ne > 64][:1]
L, L2, Md, Md2, Hv = light[0], light[1], medium[0], medium[1], heavy[0]
print('nefc', nefc)
def run(qs, T):
    phys = BatchedPhysics(m, len(qs))
    tc._set(phys, np.array(qs), np.zeros((len(qs), m.nv)))
    out = []
    for t in range(T):
        phys.step(1); torch.cuda.synchronize(); d = phys.data
        out.append((d.qpos.cpu().numpy().copy(), d.qvel.cpu().numpy().copy(), d.ncon.cpu().numpy().copy()))
    return out
T = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for name, (qa, ia), (qb, ib) in (('Md: [Md,Md2][0] vs [L,Md][1]', ([Md, Md2], 0), ([L, Md], 1)), ('Md: [Md,Md2][0] vs [Md,Hv][0]', ([Md, Md2], 0), ([Md, Hv], 0)),
                                 ('Md2: [Md,Md2][1] vs [Md2,L][0]', ([Md, Md2], 1), ([Md2, L], 0)), ('L: [L,L2][0] vs [L,Md][0]', ([L, L2], 0), ([L, Md], 0))):
    a, b = run(qa, T), run(qb, T)
    first = next((t for t in range(T) if not (np.array_equal(a[t][0][ia], b[t][0][ib]) and np.array_equal(a[t][1][ia], b[t][1][ib]))), None)
    print(name, 'first differing step', first, '' if first is None else f'qvel diff {np.abs(a[first][1][ia] - b[first][1][ib]).max():.3e} ncon {a[first][2][ia]} {b[first][2][ib]}  ncon before: {a[max(first-1,0)][2]} {b[max(first-1,0)][2]}')

print('--- rows of the first differing step')
def upto(qs, T):
    phys = BatchedPhysics(m, len(qs))
    tc._set(phys, np.array(qs), np.zeros((len(qs), m.nv)))
    if T: phys.step(T)
    rows, imp = phys.step_debug()
    torch.cuda.synchronize()
    return rows.cpu().numpy(), imp.cpu().numpy(), phys.data.qvel.cpu().numpy().copy()
S = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ra, ia_, va = upto([Md, Md2], S); rb_, ib_, vb = upto([L, Md], S)
A, B = ra[0], rb_[1]
n = int((A[:, 2] != 0).sum()); print('rows', n, int((B[:, 2] != 0).sum()))
for c, nm in enumerate(('pos', 'aref', 'R', 'b', 'force', 'R0', 'type', 'mu')):
    d = np.abs(A[:, c] - B[:, c]); print(f'  col {nm}: max diff {np.nanmax(d):.3e} at row {int(np.nanargmax(d))}')
print('  improvement sweeps 0..4:', ia_[0][:5], ib_[1][:5])
print('  first sweep with a different improvement:', next((i for i in range(ia_.shape[1]) if not (ia_[0][i] == ib_[1][i] or (np.isnan(ia_[0][i]) and np.isnan(ib_[1][i])))), None))
print('  improvements 4..12 SOLO run:', ia_[0][4:12]); print('  improvements 4..12 PAIR run:', ib_[1][4:12])
print('  rows with different forces:', np.nonzero(A[:, 4] != B[:, 4])[0], 'sweeps run', int((~np.isnan(ia_[0])).sum()), int((~np.isnan(ib_[1])).sum()))
print('  qvel diff', np.abs(va[0] - vb[1]).max())
