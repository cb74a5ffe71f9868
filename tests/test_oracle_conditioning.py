"""Why single-step acceleration parity of an fp32 composite-rigid-body step cannot be 1e-5 on these models (no GPU needed).

The joint-space inertia H = M + h B of a long chain of light links on a heavy trunk is ill-conditioned.  The test rounds every
stored entry of H to fp32 - and does nothing else: exact assembly before, exact fp64 solve after - and measures how far the
solution of H x = f moves.  That is a floor for ANY method that holds H in fp32, however carefully it computes the entries; the
GPU parity tests (test_gpu_step_parity, test_gpu_morphologies, test_gpu_teacher_forced) state their velocity / acceleration
bounds as small multiples of this floor (oracle.fp32_storage runs the same rounding inside the oracle's step) instead of
numbers fitted to a run."""
import numpy as np
import pytest

from parity_metrics import fp32_floor_of_solve, scaled_condition


@pytest.mark.parametrize('maker,cond_min,floor_min', [('salamander33', 1e4, 3e-6), ('eel', 5e4, 3e-5), ('centipede', 5e4, 8e-5)])
def test_fp32_floor_of_the_joint_space_solve(oracle, maker, cond_min, floor_min):
    import farms_mujoco_amd.model as mm
    m = getattr(mm, maker)()
    n = 16
    rng = np.random.default_rng(5)
    qpos = np.tile(m.key_qpos, (n, 1)); qpos[:, 7:] += rng.uniform(-0.5, 0.5, (n, m.nq - 7))
    quat = rng.normal(size=(n, 4)); qpos[:, 3:7] = quat/np.linalg.norm(quat, axis=1, keepdims=True)
    qvel = rng.normal(size=(n, m.nv))*0.5
    q32 = qpos.astype(np.float32).astype(np.float64); v32 = qvel.astype(np.float32).astype(np.float64)
    floors, conds = [], []
    for e in range(n):
        o = oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu))
        H = o['M'] + np.diag(m.timestep*m.dof_damping)
        f, x = fp32_floor_of_solve(H, o['qfrc_smooth'])
        floors.append(f.max()/np.abs(x).max()); conds.append(scaled_condition(H))
    # the oracle's own knob reproduces the floor inside a full step
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
    with oracle.fp32_storage():
        low = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
    knob = np.abs(low['qacc'] - ref['qacc']).max(1)/np.abs(ref['qacc']).max(1)
    print(maker, 'scaled condition number median %.1e' % np.median(conds), 'floor / max|qacc|: median %.1e max %.1e' % (np.median(floors), np.max(floors)),
          'through the step: median %.1e max %.1e' % (np.median(knob), knob.max()))
    assert np.median(conds) > cond_min
    assert np.median(floors) > floor_min and np.median(knob) > floor_min       # 2e-5 is out of reach of an fp32 H on eel / centipede
    assert 0.2 < np.median(knob)/np.median(floors) < 5.0
    assert np.abs(low['xpos'] - ref['xpos']).max() == 0.0                      # nothing but the stored matrices was touched
