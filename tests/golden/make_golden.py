"""Generates tests/golden/salamander33_fused.npz from the fp64 CPU oracle (the reference itself cannot run here:
mujoco / dm_control / farms_core are absent).  Inputs are rounded to fp32 first so the GPU and the
oracle see identical numbers.    python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from farms_mujoco_amd.model import salamander33, synthetic_batch, wave_controller_params   # noqa: E402
from farms_mujoco_amd.options import AnimatOptions   # noqa: E402
from oracle import oracle   # noqa: E402


def swim_arrays(m):
    sw = m.swimming
    idx = np.array([s['body'] - 1 for s in sw], np.int32)
    return dict(links_index=idx, xfrc_index=idx, body_index=np.array([s['body'] for s in sw], np.int32),
                coefficients=np.array([s['drag_coefficients'] for s in sw]), masses=m.body_mass[[s['body'] for s in sw]],
                heights=np.array([s['height'] for s in sw]), densities=np.array([s['density'] for s in sw]))


def main():
    m = salamander33()
    n, T = 4, 60
    qpos, qvel, psi = synthetic_batch(m, n, seed=42)
    qpos = qpos.astype(np.float32).astype(np.float64); psi = psi.astype(np.float32).astype(np.float64)
    amp, lag = wave_controller_params(m)
    amp = amp.astype(np.float32).astype(np.float64); lag = lag.astype(np.float32).astype(np.float64)
    xp, xq, xi, sd = [], [], [], []
    for e in range(n):
        o = oracle.forward_debug(m, qpos[e], qvel[e])
        s = o['sensordata'].copy(); s[6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
        xp.append(o['xpos']); xq.append(o['xquat']); xi.append(o['xipos']); sd.append(s)
    st = dict(qpos=qpos, qvel=qvel, xpos=np.array(xp), xquat=np.array(xq), xipos=np.array(xi), sensordata=np.array(sd))
    water = dict(surface=0.0, velocity=[0.0, 0.0, 0.0], viscosity=1.0, gravity=-9.81, use_buoyancy=True)
    ref = oracle.run_fused(m, st, T, swim=swim_arrays(m), water=water, buffer_size=T, controller=1,
                           wave=dict(amplitude=amp, phase_lag=lag, env_phase=psi, frequency=1.0))
    keep = [0, 1, 20, T - 1]
    np.savez_compressed(os.path.join(os.path.dirname(__file__), 'salamander33_fused.npz'),
                        n_steps=T, seed=42, keep=keep, qpos0=qpos, env_phase=psi,
                        qpos=ref['qpos'], qvel=ref['qvel'], links=ref['links'][keep], joints=ref['joints'][keep],
                        xfrc=ref['xfrc'][keep])


if __name__ == '__main__':
    main()
