"""The oracle's RK4 (mj_RungeKutta restated, oracle/fmj_oracle.c rk4): order of convergence against its own fine-step run, and what a step
leaves in the derived fields (sensordata of the first pass, poses of the last) - reference mjcf.py:1360-1365 forwards the integrator."""
import numpy as np
import pytest

from farms_mujoco_amd.model import salamander33, synthetic_batch


def _run(oracle, integrator, h, T=0.032):
    m = salamander33(timestep=h)
    m.integrator = integrator
    q, v, _ = synthetic_batch(m, 2, seed=1)
    ctrl = np.zeros((2, m.nu)); ctrl[:, :27] = 0.2*np.sin(np.arange(27))
    return oracle.step(m, q.astype(np.float64), v.astype(np.float64), ctrl=ctrl, n_steps=int(round(T/h)))


def test_rk4_is_fourth_order_and_euler_first(oracle):
    ref = _run(oracle, 1, 1e-3/16)
    for integrator, order in ((0, 1), (1, 4)):
        e = [np.abs(_run(oracle, integrator, h)['qpos'] - ref['qpos']).max() for h in (1e-3, 5e-4, 2.5e-4)]
        print('integrator', integrator, 'errors', e, 'ratios', e[0]/e[1], e[1]/e[2])
        for a, b in ((e[0], e[1]), (e[1], e[2])):
            assert 0.75*2**order < a/b < 1.3*2**order, (integrator, e)
    assert np.abs(_run(oracle, 1, 1e-3)['qpos'] - ref['qpos']).max() < 2e-5      # measured 9e-6 (Euler: 5.7e-3)


def test_rk4_leaves_first_pass_sensors_and_last_pass_poses(oracle):
    """mj_step with RK4: mj_forward, then three mj_forwardSkip(skipsensor) at the stage states - sensordata (framelinvel / frameangvel,
    joint sensors) stays the first pass's, xpos / xquat are the last pass's (the state X[3] = X[0] (+) h F[2])."""
    m = salamander33()
    m.integrator = 1
    q, v, _ = synthetic_batch(m, 1, seed=2)
    q = q.astype(np.float64); v = v.astype(np.float64) + 0.3
    out = oracle.step(m, q, v, n_steps=1)
    f0 = oracle.forward_debug(m, q[0], v[0])
    assert np.allclose(out['sensordata'][0][:6*(m.nbody - 1)], f0['sensordata'][:6*(m.nbody - 1)], rtol=0, atol=1e-12)
    assert np.abs(out['xpos'][0] - f0['xpos']).max() > 1e-5            # not the first pass's poses ...
    m.integrator = 0
    eu = oracle.step(m, q, v, n_steps=1)
    assert np.abs(eu['xpos'][0] - f0['xpos']).max() < 1e-12            # ... which is what an Euler step leaves
