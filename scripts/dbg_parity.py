import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from farms_mujoco_amd.model import salamander33
from farms_mujoco_amd.physics import BatchedPhysics
from oracle import oracle
np.set_printoptions(precision=4, linewidth=200, suppress=False)
m = salamander33()
n = 4
rng = np.random.default_rng(1)
qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 7:] += rng.uniform(-0.3, 0.3, (n, m.nq-7))
q = rng.normal(size=(n, 4)); qpos[:, 3:7] = q/np.linalg.norm(q, axis=1, keepdims=True)
qvel = rng.normal(size=(n, m.nv))*0.5
ctrl = rng.uniform(-0.5, 0.5, (n, m.nu))
mode = sys.argv[1] if len(sys.argv) > 1 else 'full'
if mode == 'novel': qvel[:] = 0
if mode == 'noctrl': ctrl[:] = 0
phys = BatchedPhysics(m, n)
d = phys.data
d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
d.ctrl[:] = torch.as_tensor(ctrl, dtype=torch.float32)
phys.step(1); torch.cuda.synchronize()
ref = oracle.step(m, qpos, qvel, ctrl=ctrl)
for name in ('xpos', 'xquat', 'xipos', 'sensordata', 'qacc', 'qvel', 'qpos'):
    a = getattr(d, name).cpu().numpy().astype(np.float64); b = ref[name]
    e = np.abs(a-b)
    print(name, 'max abs err', e.max(), 'ref max', np.abs(b).max(), 'argmax', np.unravel_index(e.argmax(), e.shape))
a = d.qacc.cpu().numpy()[0]; b = ref['qacc'][0]
print('qacc gpu', a); print('qacc ref', b); print('diff', a-b)
sd = d.sensordata.cpu().numpy()[0]; sr = ref['sensordata'][0]
print('sd diff (first 30)', (sd-sr)[:30])
print('act diff', (sd-sr)[6*28+3*27:][:12])
