import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from farms_mujoco_amd.model import salamander33, synthetic_batch, wave_controller_params
from farms_mujoco_amd.physics import BatchedPhysics
from oracle import oracle
m = salamander33()
n = 4
qpos, qvel, psi = synthetic_batch(m, n)
amp, lag = wave_controller_params(m)
T = 50
t = np.arange(T)[:, None, None]*m.timestep
tape = amp[None, None, :]*np.sin(2*np.pi*t - lag[None, None, :] + psi[None, :, None])
tape_t = torch.as_tensor(tape, dtype=torch.float32, device='cuda').contiguous()
for mode in ('single-launch', 'per-step'):
    for T_ in (1, 2, 3, 5, 10, 50):
        phys = BatchedPhysics(m, n)
        d = phys.data
        d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
        if mode == 'single-launch':
            phys.step(T_, ctrl_tape=tape_t[:T_].contiguous())
        else:
            for s in range(T_):
                d.ctrl[:] = tape_t[s]
                phys.step(1)
        torch.cuda.synchronize()
        o = oracle.step(m, qpos, qvel, ctrl=tape[:T_], n_steps=T_, ctrl_step_stride=n*m.nu)
        e = np.abs(d.qpos.cpu().numpy() - o['qpos']).max(); ev = np.abs(d.qvel.cpu().numpy() - o['qvel']).max()
        print(mode, T_, 'qpos err', e, 'qvel err', ev, 'ref qvel max', np.abs(o['qvel']).max(), 'status', d.status.cpu().numpy())
