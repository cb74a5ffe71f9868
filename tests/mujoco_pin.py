"""The models and states on which the fp64 oracle is to be PINNED against MuJoCo itself.

Shared by
  * tests/test_vs_mujoco.py          - ``pytest.importorskip('mujoco')``: steps ``model2mjcf_xml(m)`` in MuJoCo next to the oracle;
  * tests/golden/make_golden_mujoco.py - writes tests/golden/mujoco_<case>.npz (MuJoCo's outputs as committed fixtures);
  * tests/test_golden.py             - compares the oracle (CPU) and the HIP path (``-m gpu``) with those fixtures when present;
  * tests/test_mjcf_export.py        - every case's XML parses and holds what the model holds (no MuJoCo needed).

Nothing here imports ``mujoco`` at module level.  The reference reaches MuJoCo at
reference farms_mujoco/simulation/simulation.py:53,83-89,156 (``Environment.step`` -> ``mujoco.mj_step``) and
reference farms_mujoco/sensors/sensors.pyx:70 (``mj_contactForce``).
"""
import numpy as np


def _salamander(**kw):
    import farms_mujoco_amd.model as mm
    solver = kw.pop('solver', None)
    cone = kw.pop('cone', None)
    m = mm.salamander33(**kw)
    if solver:
        m.solver = mm.SOLVERS[solver]; m.solver_iterations = 100
    if cone:
        m.cone = mm.CONES[cone]
    return m


def _servo_salamander(integrator, kv=2e-3, **kw):
    """Velocity servos with a gain (the zoo's own kv is 0, where implicitfast and Euler coincide) and the integrator under test."""
    import farms_mujoco_amd.model as mm
    m = _salamander(**kw)
    for a, tag in enumerate(m.actuator_tags):
        if tag == 'velocity':
            m.actuator_gain[a] = kv; m.actuator_bias[a, 2] = -kv
    m.integrator = mm.INTEGRATORS[integrator]
    return m


def _tree(seed, **kw):
    from test_gpu_random_trees import random_tree
    return random_tree(seed, **kw)


def _morph(name):
    import farms_mujoco_amd.model as mm
    return getattr(mm, name)()


# name -> (builder, exact): ``exact`` False marks the cases whose narrow phase is this package's own construction
# (convex mesh against the ground: the deepest hull vertices; heightfield: the plane of the cell) rather than MuJoCo's
# mjc_PlaneConvex / prism test (include/fmj.h:160-164): contact points may legitimately differ there.
CASES = {
    'salamander33_swim': (lambda: _salamander(), True),
    'salamander33_walk_pgs': (lambda: _salamander(contacts=True, limits=True, spawn_z=0.045), True),
    'salamander33_walk_newton': (lambda: _salamander(contacts=True, limits=True, spawn_z=0.045, solver='newton'), True),
    'salamander33_walk_cg': (lambda: _salamander(contacts=True, limits=True, spawn_z=0.045, solver='cg'), True),
    'salamander33_walk_elliptic': (lambda: _salamander(contacts=True, limits=True, spawn_z=0.045, solver='newton', cone='elliptic'), True),
    'salamander33_walk_pairs': (lambda: _salamander(contacts=True, limits=True, spawn_z=0.045, self_collisions=True), True),
    'salamander33_walk_hfield': (lambda: _salamander(contacts=True, limits=True, spawn_z=0.045, terrain='hfield'), False),
    'salamander33_walk_mesh': (lambda: _salamander(contacts=True, limits=True, spawn_z=0.045, mesh_feet=True), False),
    'salamander33_swim_implicitfast': (lambda: _servo_salamander('implicitfast'), True),
    'salamander33_walk_implicitfast': (lambda: _servo_salamander('implicitfast', contacts=True, limits=True, spawn_z=0.045), True),
    'eel': (lambda: _morph('eel'), True),
    'centipede': (lambda: _morph('centipede'), True),
    'tree_0': (lambda: _tree(0), True), 'tree_1': (lambda: _tree(1), True), 'tree_2': (lambda: _tree(2), True),
    'tree_3': (lambda: _tree(3), True), 'tree_7': (lambda: _tree(7), True),
    'tree_contacts_100': (lambda: _tree(100, contacts=True), True), 'tree_contacts_101': (lambda: _tree(101, contacts=True), True),
    'tree_contacts_104': (lambda: _tree(104, contacts=True), True), 'tree_contacts_107': (lambda: _tree(107, contacts=True), True),
}
N_ENVS = 3
N_LONG = 1000      # north_star: qpos within 1e-4 of CPU MuJoCo after 1000 steps


def case_model(name):
    return CASES[name][0]()


def case_inputs(name, m, oracle=None):
    """Seeded inputs of a case, already rounded to fp32 (what the HIP path can hold), as float64 arrays: qpos, qvel, ctrl,
    xfrc_applied, qpos_spring [N_ENVS, ...].  Walking cases start from the state the ORACLE reaches after 300 steps of the trot
    (feet on the ground, some joints near their limits) when an oracle is given, else from the spawn pose."""
    import zlib
    import farms_mujoco_amd.model as mm
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    n = N_ENVS
    if name.startswith('salamander33') or name in ('eel', 'centipede'):
        qpos, qvel, _ = mm.synthetic_batch(m, n, seed=7)
    else:
        qpos = np.tile(m.qpos0, (n, 1)) + rng.uniform(-0.4, 0.4, (n, m.nq))
        for j in range(m.njnt):
            if m.jnt_type[j] == 0:
                a = m.jnt_qposadr[j]; q = rng.normal(size=(n, 4)); qpos[:, a + 3:a + 7] = q/np.linalg.norm(q, axis=1, keepdims=True)
                if 'contacts' in name:
                    qpos[:, a + 2] = rng.uniform(-0.05, 0.1, n)
        qvel = rng.normal(size=(n, m.nv))*0.3
    ctrl = rng.uniform(-0.3, 0.3, (n, max(m.nu, 1)))[:, :m.nu]
    xf = rng.normal(size=(n, m.nbody, 6))*0.02; xf[:, 0] = 0
    qs = np.tile(m.qpos_spring, (n, 1))
    if 'walk' in name:
        xf[:] = 0
        amp, lag = mm.trot_controller_params(m)
        ctrl = np.tile(amp*np.sin(-lag), (n, 1))
        if oracle is not None:
            o = oracle.step(m, qpos, qvel, ctrl=ctrl, n_steps=300)
            qpos, qvel = o['qpos'], o['qvel']
    r = lambda a: np.ascontiguousarray(a, np.float32).astype(np.float64)
    return dict(qpos=r(qpos), qvel=r(qvel), ctrl=r(ctrl), xfrc_applied=r(xf), qpos_spring=r(qs))


def mujoco_step(m, inp, n_steps=1, warmstart=None):
    """``mujoco.mj_step`` x n_steps on ``model2mjcf_xml(m, fusestatic=False)`` for every env of ``inp`` (the reference's
    ``Environment.step`` -> ``Physics.step``, simulation.py:83-89,156).  Returns what the oracle's ``step`` / ``step_tf`` return,
    MuJoCo's names: qpos, qvel, xpos, xquat, xipos, sensordata (of the last forward pass), and from that pass nefc, ncon,
    efc_force / efc_aref / efc_R / efc_pos, contacts as [pos(3) frame(9) mj_contactForce(3) geom1 geom2 dist], qacc (= the
    next warm start).  Needs ``mujoco``."""
    import mujoco
    from farms_mujoco_amd.simulation.mjcf import model2mjcf_xml
    mj = mujoco.MjModel.from_xml_string(model2mjcf_xml(m, fusestatic=False))
    assert (mj.nbody, mj.nq, mj.nv, mj.nu, mj.nsensordata) == (m.nbody, m.nq, m.nv, m.nu, m.nsensordata), 'the export changed the model'
    n = inp['qpos'].shape[0]
    out = {k: [] for k in ('qpos', 'qvel', 'xpos', 'xquat', 'xipos', 'sensordata', 'nefc', 'ncon', 'efc_force', 'efc_aref', 'efc_R',
                           'efc_pos', 'contact', 'qacc')}
    for e in range(n):
        d = mujoco.MjData(mj)
        d.qpos[:] = inp['qpos'][e]; d.qvel[:] = inp['qvel'][e]
        if m.nu:
            d.ctrl[:] = inp['ctrl'][e]
        d.xfrc_applied[:] = inp['xfrc_applied'][e]
        mj.qpos_spring[:] = inp['qpos_spring'][e]
        if warmstart is not None:
            d.qacc_warmstart[:] = warmstart[e]
        for _ in range(n_steps):
            mujoco.mj_step(mj, d)
        con = np.zeros((max(int(m.max_contacts), 1), 18))
        for i in range(d.ncon):
            c = d.contact[i]
            f6 = np.zeros(6)
            mujoco.mj_contactForce(mj, d, i, f6)                  # reference sensors.pyx:70
            con[i, :3] = c.pos; con[i, 3:12] = np.asarray(c.frame).ravel(); con[i, 12:15] = f6[:3]
            con[i, 15] = c.geom1 if hasattr(c, 'geom1') else c.geom[0]; con[i, 16] = c.geom2 if hasattr(c, 'geom2') else c.geom[1]
            con[i, 17] = c.dist
        ne = int(d.nefc)
        pad = lambda a: np.concatenate([np.asarray(a, float)[:ne], np.zeros(max(0, 2*m.njnt + 4*max(int(m.max_contacts), 0) - ne))])
        for k, v in (('qpos', d.qpos), ('qvel', d.qvel), ('xpos', d.xpos), ('xquat', d.xquat), ('xipos', d.xipos),
                     ('sensordata', d.sensordata), ('nefc', ne), ('ncon', int(d.ncon)), ('efc_force', pad(d.efc_force)),
                     ('efc_aref', pad(d.efc_aref)), ('efc_R', pad(d.efc_R)), ('efc_pos', pad(d.efc_pos)), ('contact', con),
                     ('qacc', d.qacc)):
            out[k].append(np.array(v, float).copy() if not np.isscalar(v) else v)
    return {k: np.array(v) for k, v in out.items()}, mj


def sort_contacts(con, ncon):
    """Contacts in a canonical order (geom pair, then position): MuJoCo and this package may list the same set differently
    when explicit pairs are present."""
    c = np.asarray(con)[:int(ncon)]
    if len(c) == 0:
        return c
    key = np.lexsort((np.round(c[:, 2], 9), np.round(c[:, 1], 9), np.round(c[:, 0], 9), c[:, 16], c[:, 15]))
    return c[key]
