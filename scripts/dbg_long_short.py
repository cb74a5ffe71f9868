"""Debug of test_fused_walk_one_long_launch_equals_many_short_ones: which env differs first, when, by how much."""
import os, sys
root = os.environ.get('GRAFT_REPO_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np, torch
import test_gpu_contacts as tc
from farms_mujoco_amd.data import AnimatData
from farms_mujoco_amd.options import SimulationOptions
from farms_mujoco_amd.simulation.simulation import Simulation
m = tc._walker(spawn_z=0.06)
n, T = 16, 240
pairs = [(b, '') for b in m.body_names[1:] if b.endswith('_3') or b.startswith('body_')]
rng = np.random.default_rng(3)
q0 = np.tile(m.key_qpos, (n, 1)); q0[:, 7:] += rng.uniform(-0.3, 0.3, (n, m.nq - 7)); q0[:, 2] = 0.03 + 0.03*rng.uniform(size=n)
belly = np.arange(n) % 4 == 0
q0[belly, 2] = 0.012
def run(chunk):
    data = AnimatData(m.timestep, T, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
    sim = Simulation(m, m.body_names[1], SimulationOptions(timestep=m.timestep, n_iterations=T), n_envs=n, data=data, buffer_size=T)
    sim.reset()
    sim.physics.data.qpos[:] = torch.as_tensor(q0, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    sim.run(fused=True, chunk=chunk)
    torch.cuda.synchronize()
    return data.sensors.links.array.cpu().numpy(), data.sensors.contacts.array.cpu().numpy()
(la, ca), (lb, cb) = run(T), run(20)
for e in range(n):
    d = np.abs(la[:, e] - lb[:, e]).reshape(T, -1).max(1)
    first = np.nonzero(d > 0)[0]
    nc = (np.abs(ca[:, e, :, 2]) > 0).sum(1)
    print(f'env {e:2d} belly {int(belly[e])}: first differing iteration {first[0] if len(first) else None}  max diff {d.max():.3e}  contact sensors active around it: {nc[max(first[0]-2,0):first[0]+2] if len(first) else ""}')

# each route against the fp64 oracle: link rows over the first TO steps
from oracle import oracle
TO = int(os.environ.get('TO', '120'))
q32 = q0.astype(np.float32).astype(np.float64); v32 = np.zeros((n, m.nv))
st = dict(qpos=q32, qvel=v32)
fds = [oracle.forward_debug(m, q32[e], v32[e]) for e in range(n)]
for k in ('xpos', 'xquat', 'xipos'):
    st[k] = np.array([fd[k] for fd in fds])
sd = np.array([fd['sensordata'] for fd in fds]); sd[:, 6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
st['sensordata'] = sd
ref = oracle.run_fused(m, st, TO, swim=None, buffer_size=TO, controller=0, ctrl=np.zeros((n, m.nu)), n_threads=8)
want = ref['links']          # [TO, n, nlinks, 20] ?
print('oracle links', want.shape, 'device', la.shape)
for e in range(n):
    ea = np.abs(la[:TO, e, :, :3] - want[:, e, :, :3]).max(); eb = np.abs(lb[:TO, e, :, :3] - want[:, e, :, :3]).max()
    print(f'env {e:2d} belly {int(belly[e])}: link positions against the oracle over {TO} steps: long launch {ea:.3e}  launches of 20 {eb:.3e}')
for k, (x, y) in (('links', (la, lb)),):
    print('relerr', k, np.abs(x - y).max() / (np.abs(y).max() + 1e-12))
    e = 4
    d = np.abs(x[:, e] - y[:, e])          # [T, nlinks, 20]
    t, l, c = np.unravel_index(np.argmax(d), d.shape)
    print('env 4 worst at iteration', t, 'link', l, 'column', c, 'values', x[t, e, l, c], y[t, e, l, c])
    for c0, c1, nm in ((0, 3, 'pos'), (3, 7, 'quat'), (7, 10, 'com pos'), (10, 14, 'com quat'), (14, 17, 'lin vel'), (17, 20, 'ang vel')):
        print('   ', nm, 'max diff', d[..., c0:c1].max(), 'by time (every 40):', d[::40, :, c0:c1].max((1, 2)))
