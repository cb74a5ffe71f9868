"""A/B of the two generations of the two-env kernel through the fused loop: python scripts/ab_dual_fused.py [n_envs] [T] [drag 0/1] [amp]."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from farms_mujoco_amd.model import salamander33, synthetic_batch
from farms_mujoco_amd.options import SimulationOptions, ArenaOptions, AnimatOptions, WaterOptions
from farms_mujoco_amd.control import WaveController
from farms_mujoco_amd.simulation.simulation import Simulation

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 5
drag = int(sys.argv[3]) if len(sys.argv) > 3 else 1
amp = float(sys.argv[4]) if len(sys.argv) > 4 else 0.3
m = salamander33()
qpos, qvel, psi = synthetic_batch(m, n)
out = {}
for gen in ('1', '2'):
    os.environ['FMJ_DUAL'] = gen
    sim = Simulation.from_sdf(SimulationOptions(timestep=m.timestep, n_iterations=T), AnimatOptions.from_model(m),
                              ArenaOptions(water=WaterOptions(height=0.0 if drag else None, drag=bool(drag))), model=m, n_envs=n,
                              controller=WaveController(m, psi, amplitude=amp), buffer_size=T)
    sim.reset()
    d = sim.physics.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    sim.step_fused(T)
    torch.cuda.synchronize()
    out[gen] = {k: getattr(d, k).cpu().numpy().copy() for k in ('qpos', 'qvel', 'qacc', 'sensordata', 'status', 'ctrl')}
    for k in ('links', 'joints', 'xfrc'):
        out[gen][k] = getattr(sim.task.data.sensors, k).array.cpu().numpy().copy().transpose(1, 0, 2, 3)
print('drag', drag, 'amp', amp, 'T', T)
for k in out['1']:
    a, b = out['1'][k].astype(np.float64).reshape(n, -1), out['2'][k].astype(np.float64).reshape(n, -1)
    with np.errstate(invalid='ignore'):
        err = np.abs(a - b).max(1)/np.maximum(np.abs(a).max(1), 1e-30)
    print(f'{k:11s} per-env rel diff', np.array2string(err, precision=2), 'nan:', np.isnan(b).any(1).astype(int))
print('status', out['1']['status'], out['2']['status'])
if T <= 3:
    x1, x2 = out['1']['xfrc'], out['2']['xfrc']
    print('xfrc env1 it0 dual1', np.array2string(x1[1, 0, :4], precision=3)); print('xfrc env1 it0 dual2', np.array2string(x2[1, 0, :4], precision=3))
    if T > 1:
        print('xfrc env1 it1 dual1', np.array2string(x1[1, 1, :4], precision=3)); print('xfrc env1 it1 dual2', np.array2string(x2[1, 1, :4], precision=3))
