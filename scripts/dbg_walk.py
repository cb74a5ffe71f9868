import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch, time
from farms_mujoco_amd.physics import BatchedPhysics
from oracle import oracle
from test_gpu_contacts import _walker, _trot_tape
m = _walker()
n = 8
for T in (20, 50, 100, 200, 300):
    tape = _trot_tape(m, n, T)
    phys = BatchedPhysics(m, n)
    d = phys.data
    d.qpos[:] = torch.as_tensor(np.tile(m.qpos0, (n, 1)), dtype=torch.float32)
    q32 = d.qpos.cpu().numpy().astype(np.float64)
    tape_t = torch.as_tensor(tape, dtype=torch.float32, device='cuda').contiguous()
    phys.step(T, ctrl_tape=tape_t); torch.cuda.synchronize()
    ref = oracle.step(m, q32, np.zeros((n, m.nv)), ctrl=tape_t.cpu().numpy().astype(np.float64), n_steps=T, ctrl_step_stride=n*m.nu, n_threads=8)
    e = np.abs(d.qpos.cpu().numpy() - ref['qpos'])
    print(T, 'max err', e.max(), 'per env', e.max(1).round(5), 'ncon', d.ncon.cpu().numpy())
# timing at 4096 envs
n = 4096
phys = BatchedPhysics(m, n)
phys.data.qpos[:] = torch.as_tensor(np.tile(m.qpos0, (n, 1)), dtype=torch.float32)
T = 100
tape_t = torch.as_tensor(_trot_tape(m, n, T), dtype=torch.float32, device='cuda').contiguous()
phys.step(T, ctrl_tape=tape_t); torch.cuda.synchronize()
t0 = time.perf_counter(); phys.step(T, ctrl_tape=tape_t); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('config 4: 4096 envs', T, 'steps', dt, 's ->', n*T/dt/1e6, 'M env-steps/s', 'mean ncon', phys.data.ncon.float().mean().item(), 'status', int(phys.data.status.abs().sum()))
