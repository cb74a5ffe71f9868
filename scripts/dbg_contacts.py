import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from farms_mujoco_amd.model import salamander33
from farms_mujoco_amd.physics import BatchedPhysics
from oracle import oracle
np.set_printoptions(precision=5, linewidth=220)
m = salamander33(contacts=True, limits=True, spawn_z=float(sys.argv[1]) if len(sys.argv) > 1 else 0.045)
n = 4
rng = np.random.default_rng(0)
qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 7:] += rng.uniform(-0.1, 0.1, (n, m.nq-7))
qpos[:, 7+3] = 1.25      # push one spine joint past its +1.2 limit
qvel = rng.normal(size=(n, m.nv))*0.05
ctrl = np.zeros((n, m.nu))
phys = BatchedPhysics(m, n)
print(phys.kernel_info())
d = phys.data
d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1
phys.step(T); torch.cuda.synchronize()
q32 = torch.as_tensor(qpos, dtype=torch.float32).numpy().astype(np.float64); v32 = torch.as_tensor(qvel, dtype=torch.float32).numpy().astype(np.float64)
ref = oracle.step(m, q32, v32, ctrl=ctrl, n_steps=T)
print('status', d.status.cpu().numpy(), 'ncon gpu', d.ncon.cpu().numpy())
fd = oracle.forward_debug(m, q32[0], v32[0], ctrl=ctrl[0])
print('oracle step-0 ncon', fd['ncon'], 'nefc', fd['nefc'])
for name in ('xpos', 'sensordata', 'qvel', 'qpos'):
    a = getattr(d, name).cpu().numpy().astype(np.float64); b = ref[name]
    e = np.abs(a-b)
    print(name, 'max abs err', e.max(), 'ref max', np.abs(b).max(), 'at', np.unravel_index(e.argmax(), e.shape))
if T == 1:
    print('efc_force ref', fd['efc_force'][:fd['nefc']])
    c = d.contact.cpu().numpy()[0, :fd['ncon']]
    print('gpu contact force (n,t1,t2)', c[:, 12:15].ravel())
    f = fd['efc_force'][fd['nefc']-4*fd['ncon']:fd['nefc']].reshape(-1, 4)
    print('ref contact normal force', f.sum(1))
    sl = 6*28 + 2
    print('limit frc gpu', d.sensordata.cpu().numpy()[0, 6*28:6*28+81].reshape(-1, 3)[:, 2])
    print('limit frc ref', ref['sensordata'][0, 6*28:6*28+81].reshape(-1, 3)[:, 2])
