"""Host-side checks that need no GPU: model compiler, sensor layout, C-ABI exports."""
import ctypes
import os
import re

import numpy as np
import pytest

from farms_mujoco_amd.model import salamander33, eel, centipede, ModelBuilder, JNT_FREE, np_mass_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_salamander33_sizes():
    m = salamander33()
    assert (m.nbody, m.njnt, m.nq, m.nv, m.nu) == (29, 28, 34, 33, 81)
    assert m.nsensordata == 28*6 + 27*3 + 81 == 330
    assert len(m.swimming) == 28
    assert salamander33(full_actuators=False).nu == 27
    names = m.sensor_names()
    assert names[0] == 'framelinvel_body_0' and names[1] == 'frameangvel_body_0'
    assert sum(n.startswith('actuatorfrc_position_') for n in names) == 27
    assert m.actuator_names[0] == 'actuator_position_joint_body_1' and m.actuator_names[2] == 'actuator_torque_joint_body_1'


@pytest.mark.parametrize('maker,nv', [(salamander33, 33), (eel, 26), (centipede, 61)])
def test_dfs_preorder_and_tables(maker, nv):
    m = maker()
    assert m.nv == nv and m.nbody <= 64
    sub = np.ones(m.nbody, int)
    for b in range(m.nbody - 1, 1, -1):
        sub[m.body_parentid[b]] += sub[b]
    for b in range(2, m.nbody):      # subtree of every body is the contiguous id range [b, b + size)
        p = m.body_parentid[b]
        assert p < b < p + sub[p]
    # sparse M bookkeeping: row i holds i and its dof ancestors
    n = 0
    for i in range(m.nv):
        assert m.dof_Madr[i] == n
        j = i
        while j >= 0:
            n += 1; j = m.dof_parentid[j]
    assert n == m.nM
    M = np_mass_matrix(m, m.qpos0)
    assert np.all(np.linalg.eigvalsh(M) > 0)
    assert np.allclose(m.dof_invweight0[6:], np.diag(np.linalg.inv(M))[6:])


def test_builder_reorders_to_dfs():
    b = ModelBuilder('t')
    b.add_body('root', 'world', mass=1, inertia=(1, 1, 1), joint='free')
    b.add_body('s1', 'root', pos=(1, 0, 0), mass=1, inertia=(1, 1, 1), joint='hinge')
    b.add_body('s2', 's1', pos=(1, 0, 0), mass=1, inertia=(1, 1, 1), joint='hinge')
    b.add_body('leg', 's1', pos=(0, 1, 0), mass=1, inertia=(1, 1, 1), joint='hinge')     # added late, nests under s1
    m = b.compile()
    assert m.body_names == ['world', 'root', 's1', 's2', 'leg']
    b2 = ModelBuilder('t2')
    b2.add_body('root', 'world', mass=1, inertia=(1, 1, 1), joint='free')
    b2.add_body('a', 'root', mass=1, inertia=(1, 1, 1), joint='hinge')
    b2.add_body('b', 'root', mass=1, inertia=(1, 1, 1), joint='hinge')
    b2.add_body('a1', 'a', mass=1, inertia=(1, 1, 1), joint='hinge')
    m2 = b2.compile()
    assert m2.body_names == ['world', 'root', 'a', 'a1', 'b']
    assert list(m2.dof_parentid[6:]) == [5, 6, 5]


def test_fullinertia_diagonalisation():
    b = ModelBuilder('t')
    I = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, -0.2], [0.1, -0.2, 1.0]])
    b.add_body('r', 'world', mass=1, fullinertia=(I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]), joint='free')
    m = b.compile()
    from farms_mujoco_amd.model import quat2mat
    R = quat2mat(m.body_iquat[1])
    assert np.allclose(R @ np.diag(m.body_inertia[1]) @ R.T, I)


def test_header_symbols_exported():
    """The shared library loads and exports every function include/fmj.h declares (no compute calls)."""
    from farms_mujoco_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'fmj.h')).read()
    declared = set(re.findall(r'^\s*(?:int|void|const char\*)\s+(fmj_\w+)\s*\(', hdr, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    if not os.path.exists(_lib.SO_PATH):
        pytest.skip('libfmj_hip.so not built (run __graft_entry__.build())')
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.fmj_abi_version() == _lib.ABI_VERSION == 6
    assert _lib.sc('LINK_SIZE') == 20 and _lib.sc('XFRC_TORQUE') == 3 and lib.fmj_sc(b'nope') == -1


def test_ctypes_struct_matches_header_order():
    """fmj_model field order in the ctypes mirror == the header (a silent mismatch would corrupt every table)."""
    from farms_mujoco_amd.model import _CModel
    hdr = open(os.path.join(ROOT, 'include', 'fmj.h')).read()
    body = hdr[hdr.index('typedef struct fmj_model {'):hdr.index('} fmj_model;')]
    fields = []
    for line in body.splitlines():
        line = line.split('/*')[0].strip()
        m = re.match(r'(?:const\s+)?(?:int32_t|double)\s*\*?\s*([\w\s,\[\]\*]+);', line)
        if m:
            for f in m.group(1).split(','):
                fields.append(re.sub(r'\[.*\]|\*|\s', '', f))
    assert fields == [f[0] for f in _CModel._fields_], (fields, [f[0] for f in _CModel._fields_])


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a GPU the product path raises instead of computing something else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from farms_mujoco_amd import _lib
    from farms_mujoco_amd.physics import BatchedPhysics
    with pytest.raises(_lib.FmjError):
        BatchedPhysics(salamander33(), 4)
    if os.path.exists(_lib.SO_PATH):
        lib = _lib.load()
        m = salamander33(); c = m.as_c(); ctx = ctypes.c_void_p()
        rc = lib.fmj_create(ctypes.byref(c), 4, 0, ctypes.byref(ctx))
        assert rc == 4 and b'no HIP device' in lib.fmj_last_error()       # FMJ_ERR_NODEVICE


def test_model_size_limits_are_refused_with_a_message_that_names_them():
    """One wavefront per environment: more than 64 bodies / dofs, or a dof chain longer than 32, is FMJ_ERR_UNSUPPORTED with the limit
    in the message ('Model size'; the reference itself has no such limit, mjcf.py:1327-1328).  fmj_create validates
    the model before it looks for a device, so this runs without a GPU."""
    from farms_mujoco_amd import _lib
    from farms_mujoco_amd.model import eel, SOLVERS
    if not os.path.exists(_lib.SO_PATH):
        pytest.skip('libfmj_hip.so not built')
    lib = _lib.load()

    def create(m):
        c = m.as_c(); ctx = ctypes.c_void_p()
        rc = lib.fmj_create(ctypes.byref(c), 4, 0, ctypes.byref(ctx))
        if rc == 0:
            lib.fmj_destroy(ctx)
        return rc, lib.fmj_last_error().decode()
    rc, msg = create(eel(n_joints=70))                       # 71 bodies, nv 76
    assert rc == 2 and '64' in msg and 'wavefront' in msg, (rc, msg)
    rc, msg = create(eel(n_joints=40))                       # nv 46, dof chain of 46: accepted without constraints (rows of up to 64, round 3)
    assert rc in (0, 4), (rc, msg)
    e40 = eel(n_joints=40)                                   # ... but not with limits / contacts (row of 32 there)
    for k in ('jnt_limited',):
        setattr(e40, k, np.ones_like(getattr(e40, k)))
    e40.jnt_range = np.tile([-1.0, 1.0], (e40.njnt, 1)).astype(float)
    rc, msg = create(e40)
    assert rc == 2 and 'chain longer than 32' in msg, (rc, msg)
    m = salamander33(contacts=True, limits=True)
    m.solver = 7
    rc, msg = create(m)
    assert rc == 2 and 'FMJ_SOLVER_PGS, FMJ_SOLVER_CG or FMJ_SOLVER_NEWTON' in msg, (rc, msg)
    m = salamander33(contacts=True, limits=True, self_collisions=True)
    m.solver = SOLVERS['newton']; m.cone = 1
    rc, msg = create(m)                                      # round 5: accepted (solved on the dual problem: elliptic block update + pairs)
    assert rc in (0, 4), (rc, msg)
    m = salamander33(contacts=True, limits=True)
    m.solver = SOLVERS['newton']; m.geom_friction = m.geom_friction*0.0
    rc, msg = create(m)                                      # frictionless contacts under Newton: accepted (solved on the dual problem)
    assert rc in (0, 4), (rc, msg)                           # 4 = FMJ_ERR_NODEVICE on a box without a GPU: the model itself passed
    m = salamander33(contacts=True, limits=True)
    m.cone = 1                                               # round 5: PGS with the elliptic cone is implemented
    rc, msg = create(m)
    assert rc in (0, 4), (rc, msg)
    m = salamander33()
    for integ, ok in ((0, True), (3, True), (1, True), (2, False), (9, False)):     # Euler, implicitfast, RK4 (round 5) | implicit, junk
        m.integrator = integ
        rc, msg = create(m)
        assert (rc in (0, 4)) if ok else (rc == 2 and 'FMJ_INT_IMPLICITFAST' in msg and 'Coriolis' in msg), (integ, rc, msg)
    m = salamander33(contacts=True, limits=True)
    m.noslip_iterations = 3                                  # round 5: the noslip post-pass is implemented
    rc, msg = create(m)
    assert rc in (0, 4), (rc, msg)
    m.noslip_iterations = -1
    rc, msg = create(m)
    assert rc != 0 and (rc == 4 or 'noslip' in msg), (rc, msg)
