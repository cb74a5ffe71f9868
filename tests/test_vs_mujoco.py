"""THE PIN: the fp64 oracle against MuJoCo itself ("an importorskip('mujoco') comparison is kept for the day a
wheel is available").  Neither this container nor the GPU box has ``mujoco`` (ordinary missing dependency), so
here the whole module SKIPS; on any machine that has it,

    pip install mujoco && python -m pytest tests/test_vs_mujoco.py -q          # the comparison itself
    python tests/golden/make_golden_mujoco.py                                  # + fixtures tests/golden/mujoco_*.npz

turns "parity unpinned" into a statement: ``model2mjcf_xml(m)`` is loaded by MuJoCo and stepped (``mujoco.mj_step``, what the
reference's ``Environment.step`` runs: simulation.py:83-89,156) next to ``oracle.step`` from identical qpos / qvel / ctrl /
xfrc_applied / qpos_spring, and every quantity the hot path reads or logs is compared: state, poses, sensordata, the constraint
rows (efc_pos / aref / R / force), the contact list and ``mj_contactForce`` (sensors.pyx:70).  CPU only; never a dependency of
the ``-m gpu`` tests (those read the fixtures when they exist).

Tolerances are fp64-against-fp64: 1e-9 relative where both sides run the same finite algorithm (kinematics, CRBA, RNE, PGS with
a fixed sweep count), 1e-6 where an iterative solver stops on a tolerance (Newton / CG), 1e-6 ... 1e-4 after N_LONG steps.
Cases marked inexact in mujoco_pin.CASES (mesh / heightfield narrow phase) compare everything except contact geometry strictly
and report the contact-point deviation.
"""
import numpy as np
import pytest

mujoco = pytest.importorskip('mujoco', reason="mujoco is not installed: the oracle stays PARITY UNPINNED here; "
                             "`pip install mujoco && python -m pytest tests/test_vs_mujoco.py && python tests/golden/make_golden_mujoco.py` pins it")

import mujoco_pin as mp      # noqa: E402


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.abs(a - b).max()/max(np.abs(b).max(), 1e-12)) if a.size else 0.0


@pytest.mark.parametrize('name', list(mp.CASES))
def test_model_constants_match_mujoco(name):
    """What MuJoCo's compiler derives from the exported XML equals what this package's compiler put in the model: tree indices,
    inertials, the constants of mj_setConst (dof_invweight0, body_invweight0, stat.meaninertia) that scale every regulariser."""
    from farms_mujoco_amd.simulation.mjcf import model2mjcf_xml
    m = mp.case_model(name)
    mj = mujoco.MjModel.from_xml_string(model2mjcf_xml(m, fusestatic=False))
    assert (mj.nbody, mj.njnt, mj.nq, mj.nv, mj.nu, mj.ngeom, mj.nM) == (m.nbody, m.njnt, m.nq, m.nv, m.nu, m.ngeom, m.nM)
    assert np.array_equal(mj.body_parentid, m.body_parentid) and np.array_equal(mj.dof_parentid, m.dof_parentid)
    assert np.array_equal(mj.jnt_type, m.jnt_type) and np.array_equal(mj.geom_type, m.geom_type) and np.array_equal(mj.geom_bodyid, m.geom_bodyid)
    for k in ('body_mass', 'body_inertia', 'body_ipos', 'body_pos', 'qpos0', 'dof_damping', 'dof_armature', 'jnt_stiffness'):
        assert rel(getattr(mj, k), getattr(m, k)) < 1e-12, k
    assert rel(mj.dof_invweight0, m.dof_invweight0) < 1e-9
    assert rel(mj.body_invweight0, m.body_invweight0) < 1e-9
    assert abs(mj.stat.meaninertia - m.meaninertia) < 1e-9*m.meaninertia
    assert mj.nsensordata == m.nsensordata
    assert [mujoco.mj_id2name(mj, mujoco.mjtObj.mjOBJ_SENSOR, i) for i in range(mj.nsensor)] == m.sensor_names()


@pytest.mark.parametrize('name', list(mp.CASES))
def test_one_step_matches_mujoco(oracle, name):
    m = mp.case_model(name)
    exact = mp.CASES[name][1]
    inp = mp.case_inputs(name, m, oracle)
    ref, mj = mp.mujoco_step(m, inp, 1)
    o = oracle.step_tf(m, inp['qpos'], inp['qvel'], ctrl=inp['ctrl'] if m.nu else None, qpos_spring=inp['qpos_spring'],
                       xfrc_applied=inp['xfrc_applied'], want_AR=False)
    fwd = oracle.step(m, inp['qpos'], inp['qvel'], ctrl=inp['ctrl'] if m.nu else None, qpos_spring=inp['qpos_spring'],
                      xfrc_applied=inp['xfrc_applied'])
    iterative = int(getattr(m, 'solver', 0)) != 0
    tol = 1e-6 if iterative else 1e-9
    for k in ('xpos', 'xquat', 'xipos'):
        assert rel(fwd[k], ref[k]) < 1e-10, (name, k, rel(fwd[k], ref[k]))
    if exact:
        assert np.array_equal(o['ncon'], ref['ncon']) and np.array_equal(o['nefc'], ref['nefc']), (name, o['ncon'], ref['ncon'], o['nefc'], ref['nefc'])
    for e in range(mp.N_ENVS):
        a = mp.sort_contacts(o['contact'][e], o['ncon'][e]); b = mp.sort_contacts(ref['contact'][e], ref['ncon'][e])
        if not exact:
            if len(a) == len(b) and len(a):
                print(name, 'env', e, 'contact-point deviation from MuJoCo', np.abs(a[:, :3] - b[:, :3]).max(), 'dist', np.abs(a[:, 17] - b[:, 17]).max())
            continue
        if len(a):
            assert np.array_equal(a[:, 15:17], b[:, 15:17]), (name, e, 'geom pairs')
            assert np.abs(a[:, :3] - b[:, :3]).max() < 1e-10 and np.abs(a[:, 3:12] - b[:, 3:12]).max() < 1e-10 and np.abs(a[:, 17] - b[:, 17]).max() < 1e-10, (name, e, 'contact geometry')
            fs = max(np.abs(b[:, 12:15]).max(), 1e-3)
            assert np.abs(a[:, 12:15] - b[:, 12:15]).max() < tol*1e2*fs, (name, e, 'mj_contactForce', a[:, 12:15], b[:, 12:15])
        ne = int(ref['nefc'][e])
        if ne and (o['ncon'][e] == 0 or not int(getattr(m, 'npair', 0))):       # same row order whenever the contact order cannot differ
            fs = max(np.abs(ref['efc_force'][e, :ne]).max(), 1e-3)
            assert np.abs(o['efc'][e, :ne, 0] - ref['efc_force'][e, :ne]).max() < tol*1e2*fs, (name, e, 'efc_force')
            assert rel(o['efc'][e, :ne, 3], ref['efc_aref'][e, :ne]) < 1e-9, (name, e, 'efc_aref')
            assert rel(o['efc'][e, :ne, 2], ref['efc_R'][e, :ne]) < 1e-9, (name, e, 'efc_R')
    if exact:
        for k in ('qpos', 'qvel', 'sensordata'):
            assert rel(fwd[k], ref[k]) < tol*10, (name, k, rel(fwd[k], ref[k]))
        assert rel(o['warmstart'], ref['qacc']) < tol*1e2, (name, 'qacc')


@pytest.mark.parametrize('name', [n for n, (_, exact) in mp.CASES.items() if exact])
def test_long_rollout_matches_mujoco(oracle, name):
    """north_star's horizon: qpos after N_LONG steps (constant ctrl; the constraint solver's warm start carried on both sides)."""
    m = mp.case_model(name)
    inp = mp.case_inputs(name, m, oracle)
    ref, _ = mp.mujoco_step(m, inp, mp.N_LONG)
    o = oracle.step(m, inp['qpos'], inp['qvel'], ctrl=inp['ctrl'] if m.nu else None, qpos_spring=inp['qpos_spring'],
                    xfrc_applied=inp['xfrc_applied'], n_steps=mp.N_LONG)
    e = rel(o['qpos'], ref['qpos'])
    print(name, f'qpos after {mp.N_LONG} steps: oracle vs MuJoCo {e:.3e}')
    assert e < 1e-4, (name, e)
