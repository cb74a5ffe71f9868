"""Batched-model description (host side).

Counterpart of the compiled ``mjModel`` that the reference obtains with
``mjcf.Physics.from_mjcf_model`` (reference farms_mujoco/simulation/simulation.py:53) from the
MJCF tree ``farms_mujoco/simulation/mjcf.py`` generates.  Field names follow MuJoCo's mjModel so a
user of ``physics.model.*`` finds the same attributes (``body_mass``, ``jnt_range``, ``dof_damping``,
``actuator_forcerange`` ... as touched at reference task.py:97-101,254-286, drag.pyx:359-372).

The supported subset is exactly what mjcf.py emits for an animat: a kinematic
tree whose bodies carry 0 or 1 joint (free / hinge / slide), explicit inertials, joint
spring-dampers, three actuators per joint (position / velocity / motor, mjcf.py:819-854), optional
collision geoms against a plane.  The arrays are fp64 numpy on the host; the HIP library keeps an
fp32 device copy (include/fmj.h: fmj_model).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 0, 1, 2, 3, 5, 6, 7
ABI_VERSION = 6

DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


# --------------------------------------------------------------------------------------------
# small quaternion helpers (w, x, y, z)

def quat_mul(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return np.array([
        a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
        a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
        a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1],
        a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0],
    ])


def quat2mat(q):
    w, x, y, z = np.asarray(q, float)
    return np.array([
        [w*w + x*x - y*y - z*z, 2*(x*y - w*z), 2*(x*z + w*y)],
        [2*(x*y + w*z), w*w - x*x + y*y - z*z, 2*(y*z - w*x)],
        [2*(x*z - w*y), 2*(y*z + w*x), w*w - x*x - y*y + z*z],
    ])


def axisangle2quat(axis, angle):
    axis = np.asarray(axis, float)
    return np.concatenate([[np.cos(angle/2)], axis*np.sin(angle/2)])


def euler2quat(euler):
    """xyz (extrinsic roll-pitch-yaw as SDF poses use) -> wxyz; mjcf.py euler2mjcquat role."""
    r, p, y = euler
    qx = axisangle2quat([1, 0, 0], r)
    qy = axisangle2quat([0, 1, 0], p)
    qz = axisangle2quat([0, 0, 1], y)
    return quat_mul(qz, quat_mul(qy, qx))


def mat2quat(R):
    """Rotation matrix -> unit quaternion wxyz."""
    R = np.asarray(R, float)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)*2
        q = np.array([0.25*s, (R[2, 1]-R[1, 2])/s, (R[0, 2]-R[2, 0])/s, (R[1, 0]-R[0, 1])/s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i+1) % 3, (i+2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k])*2
        q = np.zeros(4)
        q[0] = (R[k, j]-R[j, k])/s
        q[1+i] = 0.25*s
        q[1+j] = (R[j, i]+R[i, j])/s
        q[1+k] = (R[k, i]+R[i, k])/s
    return q/np.linalg.norm(q)


# --------------------------------------------------------------------------------------------

@dataclass
class _Body:
    name: str
    parent: int
    pos: np.ndarray
    quat: np.ndarray
    mass: float
    ipos: np.ndarray
    iquat: np.ndarray
    inertia: np.ndarray
    joint: Optional[dict] = None
    swimming: Optional[dict] = None
    geoms: List[dict] = field(default_factory=list)


class ModelBuilder:
    """Incrementally describe an animat, then :meth:`compile` it to flat arrays.

    Plays the role of ``sdf2mjcf`` + MuJoCo's XML compiler (reference mjcf.py:647-1035) without the
    dm_control dependency: bodies are added parent-first, exactly as ``add_link_recursive``
    (mjcf.py:603-644) walks the SDF tree.
    """

    def __init__(self, name: str = 'animat', timestep: float = 1e-3, gravity=(0.0, 0.0, -9.81)):
        self.name = name
        self.timestep = float(timestep)
        self.gravity = np.asarray(gravity, float)
        self.bodies: List[_Body] = [_Body('world', -1, np.zeros(3), np.array([1., 0, 0, 0]), 0.0,
                                          np.zeros(3), np.array([1., 0, 0, 0]), np.zeros(3))]
        self.actuators: List[dict] = []
        self.hfield: Optional[dict] = None
        self.pairs: List[dict] = []
        self.options = dict(solver_iterations=50, max_contacts=0, impratio=1.0, solver_tolerance=1e-8, solver='PGS', cone='pyramidal',
                            ls_iterations=50, ls_tolerance=0.01, noslip_iterations=0, noslip_tolerance=1e-6)

    def body_id(self, name: str) -> int:
        for i, b in enumerate(self.bodies):
            if b.name == name:
                return i
        raise KeyError(name)

    def add_body(self, name, parent='world', pos=(0, 0, 0), quat=(1, 0, 0, 0), mass=1e-10,
                 ipos=(0, 0, 0), inertia=(1e-12, 1e-12, 1e-12), fullinertia=None, iquat=(1, 0, 0, 0),
                 joint=None, **jnt):
        """Add a body.  ``joint`` in {None, 'free', 'hinge', 'slide'}; joint kwargs: ``jname, axis,
        jpos, stiffness, damping, armature, limited, range, solreflimit, solimplimit, margin, qpos0``
        (reference mjcf.py:181-212,1432-1444).  Massless default = mjcf.py:591-598."""
        pid = parent if isinstance(parent, int) else self.body_id(parent)
        iquat = np.asarray(iquat, float)
        inertia = np.asarray(inertia, float)
        if fullinertia is not None:
            # MuJoCo compiler: diagonalise fullinertia -> principal moments + iquat (mjcf.py:540-589)
            ixx, iyy, izz, ixy, ixz, iyz = fullinertia
            I = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]], float)
            evals, evecs = np.linalg.eigh(I)
            if np.linalg.det(evecs) < 0:
                evecs[:, 0] *= -1
            inertia = evals
            iquat = mat2quat(evecs)
        jd = None
        if joint is not None:
            assert joint in ('free', 'hinge', 'slide'), joint
            jd = dict(type={'free': JNT_FREE, 'hinge': JNT_HINGE, 'slide': JNT_SLIDE}[joint],
                      name=jnt.pop('jname', f'joint_{name}' if joint != 'free' else f'root_{name}'),
                      axis=np.asarray(jnt.pop('axis', (0, 0, 1)), float),
                      pos=np.asarray(jnt.pop('jpos', (0, 0, 0)), float),
                      stiffness=float(jnt.pop('stiffness', 0.0)), damping=float(jnt.pop('damping', 0.0)),
                      armature=float(jnt.pop('armature', 0.0)), limited=bool(jnt.pop('limited', False)),
                      range=np.asarray(jnt.pop('range', (0.0, 0.0)), float),
                      solref=np.asarray(jnt.pop('solreflimit', DEFAULT_SOLREF), float),
                      solimp=np.asarray(jnt.pop('solimplimit', DEFAULT_SOLIMP), float),
                      margin=float(jnt.pop('margin', 0.0)), qpos0=float(jnt.pop('qpos0', 0.0)))
            if jd['type'] != JNT_FREE:
                n = np.linalg.norm(jd['axis'])
                assert n > 0
                jd['axis'] = jd['axis']/n
        assert not jnt, jnt
        q = np.asarray(quat, float)
        self.bodies.append(_Body(name, pid, np.asarray(pos, float), q/np.linalg.norm(q), float(mass),
                                 np.asarray(ipos, float), iquat/np.linalg.norm(iquat), inertia, jd))
        return len(self.bodies) - 1

    def set_swimming(self, body, density=1000.0, drag_coefficients=None, height=None):
        """Per-link swimming options (AnimatOptions.morphology.links[*], reference drag.pyx:353-385)."""
        b = self.bodies[body if isinstance(body, int) else self.body_id(body)]
        b.swimming = dict(density=float(density),
                          drag_coefficients=np.asarray(drag_coefficients, float).reshape(2, 3),
                          height=height)

    def add_geom(self, body, gtype, size, pos=(0, 0, 0), quat=(1, 0, 0, 0), friction=(0, 0, 0),
                 solref=DEFAULT_SOLREF, solimp=DEFAULT_SOLIMP):
        b = self.bodies[body if isinstance(body, int) else self.body_id(body)]
        s = np.zeros(3); s[:len(size)] = size
        b.geoms.append(dict(type=int(gtype), size=s, pos=np.asarray(pos, float), quat=np.asarray(quat, float),
                            friction=np.asarray(friction, float), solref=np.asarray(solref, float),
                            solimp=np.asarray(solimp, float)))

    def add_mesh_geom(self, body, vertices, pos=(0, 0, 0), quat=(1, 0, 0, 0), friction=(0, 0, 0), solref=DEFAULT_SOLREF,
                      solimp=DEFAULT_SOLIMP, hull=True):
        """Convex mesh collision geom (reference mjcf.py:270-413: every SDF mesh collision becomes a MuJoCo mesh geom,
        which collides as its convex hull).  ``vertices`` [n, 3] in the geom frame; with ``hull`` they are reduced to the
        vertices of their convex hull (scipy), in their original order."""
        v = np.ascontiguousarray(vertices, float).reshape(-1, 3)
        assert len(v) >= 1, 'a mesh needs vertices'
        if hull and len(v) > 4:
            try:
                from scipy.spatial import ConvexHull
                v = v[np.sort(ConvexHull(v).vertices)]
            except Exception:          # degenerate (flat / collinear) clouds keep every vertex
                pass
        self.add_geom(body, GEOM_MESH, (0, 0, float(np.linalg.norm(v, axis=1).max())), pos=pos, quat=quat, friction=friction,
                      solref=solref, solimp=solimp)
        b = self.bodies[body if isinstance(body, int) else self.body_id(body)]
        b.geoms[-1]['vertices'] = v

    def add_hfield(self, data, size, pos=(0, 0, 0), quat=(1, 0, 0, 0), friction=(0, 0, 0), solref=DEFAULT_SOLREF,
                   solimp=DEFAULT_SOLIMP):
        """World-attached heightfield (reference mjcf.py:486-522): ``data`` [nrow, ncol] with rows along +y and columns
        along +x, ``size`` = (x radius, y radius, elevation scale, base depth) as MuJoCo's hfield asset; elevation =
        data * size[2]."""
        data = np.ascontiguousarray(data, float)
        assert data.ndim == 2 and min(data.shape) >= 2 and self.hfield is None, 'one heightfield with at least 2 x 2 samples'
        self.hfield = dict(data=data, size=np.asarray(size, float).reshape(4))
        self.add_geom('world', GEOM_HFIELD, (0, 0, 0), pos=pos, quat=quat, friction=friction, solref=solref, solimp=solimp)

    def add_contact_pair(self, body1, body2, friction=0.0, solref=DEFAULT_SOLREF, solimp=DEFAULT_SOLIMP):
        """Explicit contact pairs between every collision geom of ``body1`` and every one of ``body2`` (MJCF contact/pair,
        condim 3, the loop of reference mjcf.py:1012-1033 over morphology.self_collisions).  Resolved at compile()."""
        self.pairs.append(dict(body1=body1, body2=body2, friction=float(friction), solref=np.asarray(solref, float),
                               solimp=np.asarray(solimp, float)))

    def add_joint_actuators(self, joint_name, kp=0.0, kv=0.0, forcerange=None, pos_limits=None, vel_limits=None):
        """The position / velocity / motor triple of reference mjcf.py:819-866.  ``forcerange`` = the motor's
        ``limits_torque`` applied to all three (:855-865); ``pos_limits`` / ``vel_limits`` = dict(ctrllimited, ctrlrange,
        forcelimited, forcerange) of the ``act_pos_*`` / ``act_vel_*`` options (:675-684,829-832,844-847)."""
        none = dict(ctrllimited=False, ctrlrange=(0.0, 0.0), forcelimited=False, forcerange=(0.0, 0.0))
        for tag, gain, bias, lim in (('position', kp, (0.0, -kp, 0.0), pos_limits), ('velocity', kv, (0.0, 0.0, -kv), vel_limits),
                                     ('torque', 1.0, (0.0, 0.0, 0.0), None)):
            lim = dict(none, **(lim or {}))
            if forcerange is not None:
                lim.update(forcelimited=True, forcerange=tuple(forcerange))
            self.actuators.append(dict(name=f'actuator_{tag}_{joint_name}', joint=joint_name, tag=tag,
                                       gain=float(gain), bias=bias, ctrllimited=bool(lim['ctrllimited']),
                                       ctrlrange=tuple(lim['ctrlrange']), forcelimited=bool(lim['forcelimited']),
                                       forcerange=tuple(lim['forcerange'])))

    def add_position_actuator(self, joint_name, kp):
        """Compact layout: only the position actuator (SURVEY Appendix D allows nu = n_joints)."""
        self.actuators.append(dict(name=f'actuator_position_{joint_name}', joint=joint_name, tag='position',
                                   gain=float(kp), bias=(0.0, -float(kp), 0.0), ctrllimited=False,
                                   ctrlrange=(0.0, 0.0), forcelimited=False, forcerange=(0.0, 0.0)))

    def compile(self) -> 'Model':
        return Model._from_builder(self)


# --------------------------------------------------------------------------------------------

_I = ctypes.POINTER(ctypes.c_int32)
_D = ctypes.POINTER(ctypes.c_double)
_CMODEL_FIELDS = (
        [('abi_version', ctypes.c_int32)] +
        [(n, ctypes.c_int32) for n in ('nbody', 'njnt', 'nq', 'nv', 'nu', 'ngeom', 'nM')] +
        [('timestep', ctypes.c_double), ('gravity', ctypes.c_double*3)] +
        [(n, _I) for n in ('body_parentid', 'body_rootid', 'body_jntadr', 'body_dofadr', 'body_dofnum')] +
        [(n, _D) for n in ('body_pos', 'body_quat', 'body_ipos', 'body_iquat', 'body_mass', 'body_inertia')] +
        [(n, _I) for n in ('jnt_type', 'jnt_qposadr', 'jnt_dofadr', 'jnt_bodyid')] +
        [(n, _D) for n in ('jnt_pos', 'jnt_axis', 'jnt_stiffness')] +
        [('jnt_limited', _I)] +
        [(n, _D) for n in ('jnt_range', 'jnt_solref', 'jnt_solimp', 'jnt_margin', 'qpos0')] +
        [(n, _I) for n in ('dof_bodyid', 'dof_jntid', 'dof_parentid', 'dof_Madr')] +
        [(n, _D) for n in ('dof_armature', 'dof_damping', 'dof_invweight0')] +
        [('actuator_jntid', _I), ('actuator_gain', _D), ('actuator_bias', _D), ('actuator_ctrllimited', _I),
         ('actuator_ctrlrange', _D), ('actuator_forcelimited', _I), ('actuator_forcerange', _D)] +
        [('geom_type', _I), ('geom_bodyid', _I)] +
        [(n, _D) for n in ('geom_size', 'geom_pos', 'geom_quat', 'geom_friction', 'geom_solref', 'geom_solimp',
                           'body_invweight0')] +
        [('hfield_nrow', ctypes.c_int32), ('hfield_ncol', ctypes.c_int32), ('hfield_size', ctypes.c_double*4), ('hfield_data', _D)] +
        [('nmeshvert', ctypes.c_int32), ('mesh_vert', _D), ('geom_vertadr', _I), ('geom_vertnum', _I)] +
        [('npair', ctypes.c_int32), ('pair_geom1', _I), ('pair_geom2', _I), ('pair_friction', _D), ('pair_solref', _D), ('pair_solimp', _D)] +
        [('solver_iterations', ctypes.c_int32), ('max_contacts', ctypes.c_int32),
         ('impratio', ctypes.c_double), ('solver_tolerance', ctypes.c_double), ('meaninertia', ctypes.c_double)] +
        [('solver', ctypes.c_int32), ('cone', ctypes.c_int32), ('ls_iterations', ctypes.c_int32), ('noslip_iterations', ctypes.c_int32),
         ('ls_tolerance', ctypes.c_double), ('noslip_tolerance', ctypes.c_double), ('integrator', ctypes.c_int32)] +
        [('nmeshface', ctypes.c_int32), ('mesh_face', _D), ('geom_faceadr', _I), ('geom_facenum', _I)]
)
SOLVERS = {'pgs': 0, 'cg': 1, 'newton': 2}          # mjtSolver (include/fmj.h FMJ_SOLVER_*)
CONES = {'pyramidal': 0, 'elliptic': 1}            # mjtCone
INTEGRATORS = {'euler': 0, 'rk4': 1, 'implicit': 2, 'implicitfast': 3}      # mjtIntegrator (include/fmj.h FMJ_INT_*: Euler and implicitfast are implemented)


class _CModel(ctypes.Structure):
    """ctypes mirror of ``fmj_model`` (include/fmj.h)."""
    _fields_ = _CMODEL_FIELDS


_INT_FIELDS = ('body_parentid', 'body_rootid', 'body_jntadr', 'body_dofadr', 'body_dofnum', 'jnt_type',
               'jnt_qposadr', 'jnt_dofadr', 'jnt_bodyid', 'jnt_limited', 'dof_bodyid', 'dof_jntid',
               'dof_parentid', 'dof_Madr', 'actuator_jntid', 'actuator_ctrllimited', 'actuator_forcelimited',
               'geom_type', 'geom_bodyid', 'pair_geom1', 'pair_geom2', 'geom_vertadr', 'geom_vertnum', 'geom_faceadr', 'geom_facenum')
_DBL_FIELDS = ('body_pos', 'body_quat', 'body_ipos', 'body_iquat', 'body_mass', 'body_inertia', 'jnt_pos',
               'jnt_axis', 'jnt_stiffness', 'jnt_range', 'jnt_solref', 'jnt_solimp', 'jnt_margin', 'qpos0',
               'dof_armature', 'dof_damping', 'dof_invweight0', 'actuator_gain', 'actuator_bias',
               'actuator_ctrlrange', 'actuator_forcerange', 'geom_size', 'geom_pos', 'geom_quat',
               'geom_friction', 'geom_solref', 'geom_solimp', 'body_invweight0', 'pair_friction', 'pair_solref', 'pair_solimp',
               'mesh_vert', 'mesh_face')


def hull_faces(vertices):
    """Planes of the convex hull of ``vertices`` [n, 3]: rows (nx, ny, nz, d), outward unit normal, n . x <= d inside; coplanar
    triangles of the hull merged into one plane.  A degenerate cloud (flat, collinear) has no faces: [0, 4]."""
    v = np.ascontiguousarray(vertices, float).reshape(-1, 3)
    if len(v) < 4:
        return np.zeros((0, 4))
    try:
        from scipy.spatial import ConvexHull
        eq = ConvexHull(v).equations                      # n . x + off <= 0 inside
    except Exception:
        return np.zeros((0, 4))
    scale = max(float(np.abs(v).max()), 1e-12)
    out, seen = [], set()
    for n0, n1, n2, off in eq:
        key = (round(n0, 6), round(n1, 6), round(n2, 6), round(off/scale, 6))
        if key not in seen:
            seen.add(key)
            out.append((n0, n1, n2, -off))
    return np.array(out, float).reshape(-1, 4)


class Model:
    """Flat, MuJoCo-named model arrays (+ names) for one morphology shared by all envs."""

    def __init__(self):
        self.name = 'animat'
        self.body_names: List[str] = []
        self.joint_names: List[str] = []
        self.actuator_names: List[str] = []
        self.actuator_tags: List[str] = []
        self.swimming: List[dict] = []
        self.key_qpos = None
        self.key_qvel = None

    # ---- construction ------------------------------------------------------------------------
    @classmethod
    def _from_builder(cls, b: ModelBuilder) -> 'Model':
        m = cls()
        m.name = b.name
        nb = len(b.bodies)
        for i, body in enumerate(b.bodies):
            assert body.parent < i, 'bodies must be added parent-first'
        # MuJoCo numbers bodies in depth-first pre-order of the XML nesting (children in the order
        # they were added); a subtree is then a contiguous id range, which the kernels rely on.
        children = [[] for _ in range(nb)]
        for i in range(1, nb):
            children[b.bodies[i].parent].append(i)
        order, stack = [], [0]
        while stack:
            i = stack.pop()
            order.append(i)
            stack.extend(reversed(children[i]))
        new_of_old = {old: new for new, old in enumerate(order)}
        bodies = []
        for old in order:
            body = b.bodies[old]
            bodies.append(_Body(body.name, new_of_old.get(body.parent, -1), body.pos, body.quat, body.mass,
                                body.ipos, body.iquat, body.inertia, body.joint, body.swimming, body.geoms))
        m.timestep = b.timestep
        m.gravity = b.gravity.copy()
        m.nbody = nb
        m.body_names = [x.name for x in bodies]
        m.body_parentid = np.array([max(x.parent, 0) for x in bodies], np.int32)
        m.body_pos = np.array([x.pos for x in bodies], float)
        m.body_quat = np.array([x.quat for x in bodies], float)
        m.body_ipos = np.array([x.ipos for x in bodies], float)
        m.body_iquat = np.array([x.iquat for x in bodies], float)
        m.body_mass = np.array([x.mass for x in bodies], float)
        m.body_inertia = np.array([x.inertia for x in bodies], float)
        rootid = np.zeros(nb, np.int32)
        for i in range(1, nb):
            p = m.body_parentid[i]
            rootid[i] = i if p == 0 else rootid[p]
        m.body_rootid = rootid

        jnt = dict(type=[], qposadr=[], dofadr=[], bodyid=[], pos=[], axis=[], stiffness=[], limited=[],
                   range=[], solref=[], solimp=[], margin=[])
        qpos0, dof_bodyid, dof_jntid, dof_parentid, dof_arm, dof_damp = [], [], [], [], [], []
        body_jntadr = -np.ones(nb, np.int32)
        body_dofadr = -np.ones(nb, np.int32)
        body_dofnum = np.zeros(nb, np.int32)
        m.joint_names = []
        last_dof = -np.ones(nb, np.int32)     # last dof on the path root -> body (inclusive)
        for i in range(1, nb):
            body = bodies[i]
            pdof = last_dof[m.body_parentid[i]]
            if body.joint is None:
                last_dof[i] = pdof
                continue
            jd = body.joint
            j = len(jnt['type'])
            body_jntadr[i] = j
            body_dofadr[i] = len(dof_bodyid)
            m.joint_names.append(jd['name'])
            jnt['type'].append(jd['type']); jnt['qposadr'].append(len(qpos0)); jnt['dofadr'].append(len(dof_bodyid))
            jnt['bodyid'].append(i); jnt['pos'].append(jd['pos']); jnt['axis'].append(jd['axis'])
            jnt['stiffness'].append(jd['stiffness']); jnt['limited'].append(int(jd['limited']))
            jnt['range'].append(jd['range']); jnt['solref'].append(jd['solref']); jnt['solimp'].append(jd['solimp'])
            jnt['margin'].append(jd['margin'])
            if jd['type'] == JNT_FREE:
                assert m.body_parentid[i] == 0, 'free joint only on a root body'
                # reference spawn pose lives in the keyframe (mjcf.py:744-788); qpos0 = body pose
                qpos0 += list(body.pos) + list(body.quat)
                ndof = 6
            else:
                qpos0.append(jd['qpos0'])
                ndof = 1
            body_dofnum[i] = ndof
            for d in range(ndof):
                dof_bodyid.append(i); dof_jntid.append(j)
                dof_parentid.append(pdof if d == 0 else len(dof_bodyid) - 2)
                dof_arm.append(jd['armature']); dof_damp.append(jd['damping'] if jd['type'] != JNT_FREE else 0.0)
            last_dof[i] = len(dof_bodyid) - 1
        m.njnt = len(jnt['type'])
        m.nq = len(qpos0)
        m.nv = len(dof_bodyid)
        m.body_jntadr, m.body_dofadr, m.body_dofnum = body_jntadr, body_dofadr, body_dofnum
        m.jnt_type = np.array(jnt['type'], np.int32)
        m.jnt_qposadr = np.array(jnt['qposadr'], np.int32)
        m.jnt_dofadr = np.array(jnt['dofadr'], np.int32)
        m.jnt_bodyid = np.array(jnt['bodyid'], np.int32)
        m.jnt_pos = np.array(jnt['pos'], float).reshape(-1, 3)
        m.jnt_axis = np.array(jnt['axis'], float).reshape(-1, 3)
        m.jnt_stiffness = np.array(jnt['stiffness'], float)
        m.jnt_limited = np.array(jnt['limited'], np.int32)
        m.jnt_range = np.array(jnt['range'], float).reshape(-1, 2)
        m.jnt_solref = np.array(jnt['solref'], float).reshape(-1, 2)
        m.jnt_solimp = np.array(jnt['solimp'], float).reshape(-1, 5)
        m.jnt_margin = np.array(jnt['margin'], float)
        m.qpos0 = np.array(qpos0, float)
        m.qpos_spring = m.qpos0.copy()
        m.dof_bodyid = np.array(dof_bodyid, np.int32)
        m.dof_jntid = np.array(dof_jntid, np.int32)
        m.dof_parentid = np.array(dof_parentid, np.int32)
        m.dof_armature = np.array(dof_arm, float)
        m.dof_damping = np.array(dof_damp, float)
        madr, n = [], 0
        for i in range(m.nv):
            madr.append(n)
            j = i
            while j >= 0:
                n += 1
                j = m.dof_parentid[j]
        m.dof_Madr = np.array(madr, np.int32)
        m.nM = n

        # actuators
        acts = b.actuators
        m.nu = len(acts)
        m.actuator_names = [a['name'] for a in acts]
        m.actuator_tags = [a['tag'] for a in acts]
        m.actuator_jntid = np.array([m.joint_names.index(a['joint']) for a in acts], np.int32)
        m.actuator_gain = np.array([a['gain'] for a in acts], float)
        m.actuator_bias = np.array([a['bias'] for a in acts], float).reshape(-1, 3)
        m.actuator_ctrllimited = np.array([int(a['ctrllimited']) for a in acts], np.int32)
        m.actuator_ctrlrange = np.array([a['ctrlrange'] for a in acts], float).reshape(-1, 2)
        m.actuator_forcelimited = np.array([int(a['forcelimited']) for a in acts], np.int32)
        m.actuator_forcerange = np.array([a['forcerange'] for a in acts], float).reshape(-1, 2)
        for a in m.actuator_jntid:
            assert m.jnt_type[a] in (JNT_HINGE, JNT_SLIDE)

        # geoms
        geoms = [(i, g) for i, body in enumerate(bodies) for g in body.geoms]
        m.ngeom = len(geoms)
        m.geom_type = np.array([g['type'] for _, g in geoms], np.int32)
        m.geom_bodyid = np.array([i for i, _ in geoms], np.int32)
        m.geom_size = np.array([g['size'] for _, g in geoms], float).reshape(-1, 3)
        m.geom_pos = np.array([g['pos'] for _, g in geoms], float).reshape(-1, 3)
        m.geom_quat = np.array([g['quat'] for _, g in geoms], float).reshape(-1, 4)
        m.geom_friction = np.array([g['friction'] for _, g in geoms], float).reshape(-1, 3)
        m.geom_solref = np.array([g['solref'] for _, g in geoms], float).reshape(-1, 2)
        m.geom_solimp = np.array([g['solimp'] for _, g in geoms], float).reshape(-1, 5)
        # convex meshes: the vertices of all mesh geoms, concatenated in geom order
        m.geom_vertadr = np.full(m.ngeom, -1, np.int32); m.geom_vertnum = np.zeros(m.ngeom, np.int32)
        verts = []
        for gi, (_, g) in enumerate(geoms):
            if g['type'] == GEOM_MESH:
                m.geom_vertadr[gi] = sum(len(v) for v in verts); m.geom_vertnum[gi] = len(g['vertices'])
                verts.append(g['vertices'])
        m.mesh_vert = np.concatenate(verts) if verts else np.zeros((0, 3))
        m.nmeshvert = len(m.mesh_vert)
        # ... and the planes of their hulls (explicit pairs with a mesh: include/fmj.h, ABI 6)
        m.geom_faceadr = np.full(m.ngeom, -1, np.int32); m.geom_facenum = np.zeros(m.ngeom, np.int32)
        faces = []
        for gi, (_, g) in enumerate(geoms):
            if g['type'] == GEOM_MESH:
                f = hull_faces(g['vertices'])
                m.geom_faceadr[gi] = sum(len(x) for x in faces); m.geom_facenum[gi] = len(f)
                faces.append(f)
        m.mesh_face = np.concatenate(faces) if faces else np.zeros((0, 4))
        m.nmeshface = len(m.mesh_face)
        # explicit geom pairs: every geom of body1 x every geom of body2, in the order the pairs were added
        pg1, pg2, pfr, psr, psi = [], [], [], [], []
        for pr in b.pairs:
            b1 = new_of_old[b.body_id(pr['body1'])] if not isinstance(pr['body1'], int) else new_of_old[pr['body1']]
            b2 = new_of_old[b.body_id(pr['body2'])] if not isinstance(pr['body2'], int) else new_of_old[pr['body2']]
            for ga in [gi for gi, (bi, _) in enumerate(geoms) if bi == b1]:
                for gb in [gi for gi, (bi, _) in enumerate(geoms) if bi == b2]:
                    pg1.append(ga); pg2.append(gb); pfr.append(pr['friction']); psr.append(pr['solref']); psi.append(pr['solimp'])
        m.npair = len(pg1)
        m.pair_geom1 = np.array(pg1, np.int32); m.pair_geom2 = np.array(pg2, np.int32)
        m.pair_friction = np.array(pfr, float)
        m.pair_solref = np.array(psr, float).reshape(-1, 2); m.pair_solimp = np.array(psi, float).reshape(-1, 5)
        m.hfield_nrow, m.hfield_ncol = (b.hfield['data'].shape if b.hfield is not None else (0, 0))
        m.hfield_size = b.hfield['size'].copy() if b.hfield is not None else np.zeros(4)
        m.hfield_data = b.hfield['data'].copy() if b.hfield is not None else None
        m.solver_iterations = int(b.options['solver_iterations'])
        m.max_contacts = int(b.options['max_contacts'])
        m.impratio = float(b.options['impratio'])
        m.solver_tolerance = float(b.options['solver_tolerance'])
        m.solver = SOLVERS[str(b.options.get('solver', 'PGS')).lower()]
        m.cone = CONES[str(b.options.get('cone', 'pyramidal')).lower()]
        m.ls_iterations = int(b.options.get('ls_iterations', 50)); m.ls_tolerance = float(b.options.get('ls_tolerance', 0.01))
        m.noslip_iterations = int(b.options.get('noslip_iterations', 0)); m.noslip_tolerance = float(b.options.get('noslip_tolerance', 1e-6))
        m.integrator = INTEGRATORS[str(b.options.get('integrator', 'Euler')).lower()]

        # swimming links, in body order (reference drag.pyx:353-385)
        m.swimming = []
        for i, body in enumerate(bodies):
            if body.swimming is not None:
                m.swimming.append(dict(body=i, name=body.name, **body.swimming))

        # keyframe 0 (reference mjcf.py:744-788, task.py:137)
        m.key_qpos = m.qpos0.copy()
        m.key_qvel = np.zeros(m.nv)

        m._set_const()
        return m

    # ---- mj_setConst: inverse weights at qpos0 ------------------------------------------------
    def _set_const(self):
        M = np_mass_matrix(self, self.qpos0)
        Minv = np.linalg.inv(M) if self.nv else np.zeros((0, 0))
        diw = np.zeros(self.nv)
        for j in range(self.njnt):
            a = self.jnt_dofadr[j]
            if self.jnt_type[j] == JNT_FREE:
                diw[a:a+3] = np.mean(np.diag(Minv)[a:a+3])
                diw[a+3:a+6] = np.mean(np.diag(Minv)[a+3:a+6])
            else:
                diw[a] = Minv[a, a]
        self.dof_invweight0 = diw
        biw = np.zeros((self.nbody, 2))
        kin = np_kinematics(self, self.qpos0)
        for b in range(1, self.nbody):
            jp, jr = np_body_jacobian(self, kin, b, kin['xipos'][b])
            if self.nv:
                biw[b, 0] = np.trace(jp @ Minv @ jp.T)/3
                biw[b, 1] = np.trace(jr @ Minv @ jr.T)/3
        self.body_invweight0 = biw
        self.meaninertia = float(np.mean(np.diag(M))) if self.nv else 1.0

    # ---- sensors layout (reference mjcf.py:950-1002) -------------------------------------------
    @property
    def n_sensor_joints(self):
        return int(np.sum(self.jnt_type != JNT_FREE))

    @property
    def nsensordata(self):
        return 6*(self.nbody - 1) + 3*self.n_sensor_joints + self.nu

    def sensor_names(self) -> List[str]:
        """One name per sensordata scalar group, as mjcf.py names them (prefix matching at
        reference physics.py:86-95 relies on these)."""
        names = []
        for b in range(1, self.nbody):
            names += [f'framelinvel_{self.body_names[b]}', f'frameangvel_{self.body_names[b]}']
        for j in range(self.njnt):
            if self.jnt_type[j] != JNT_FREE:
                names += [f'jointpos_{self.joint_names[j]}', f'jointvel_{self.joint_names[j]}',
                          f'jointlimitfrc_{self.joint_names[j]}']
        for a in range(self.nu):
            names.append(f'actuatorfrc_{self.actuator_tags[a]}_{self.joint_names[self.actuator_jntid[a]]}')
        return names

    # ---- C view -------------------------------------------------------------------------------
    def as_c(self) -> _CModel:
        """ctypes ``fmj_model`` whose pointers alias (and keep alive) contiguous numpy copies."""
        c = _CModel()
        keep = {}
        c.abi_version = ABI_VERSION
        for n in ('nbody', 'njnt', 'nq', 'nv', 'nu', 'ngeom', 'nM', 'solver_iterations', 'max_contacts'):
            setattr(c, n, int(getattr(self, n)))
        c.npair = int(getattr(self, 'npair', 0))
        c.nmeshvert = int(getattr(self, 'nmeshvert', 0))
        c.nmeshface = int(getattr(self, 'nmeshface', 0))
        c.timestep = self.timestep
        c.gravity = (ctypes.c_double*3)(*self.gravity)
        c.impratio = self.impratio
        c.solver_tolerance = self.solver_tolerance
        c.meaninertia = self.meaninertia
        c.solver = int(getattr(self, 'solver', 0)); c.cone = int(getattr(self, 'cone', 0))
        c.ls_iterations = int(getattr(self, 'ls_iterations', 50)); c.ls_tolerance = float(getattr(self, 'ls_tolerance', 0.01))
        c.noslip_iterations = int(getattr(self, 'noslip_iterations', 0)); c.noslip_tolerance = float(getattr(self, 'noslip_tolerance', 1e-6))
        c.integrator = int(getattr(self, 'integrator', 0))
        c.hfield_nrow, c.hfield_ncol = int(getattr(self, 'hfield_nrow', 0)), int(getattr(self, 'hfield_ncol', 0))
        c.hfield_size = (ctypes.c_double*4)(*np.asarray(getattr(self, 'hfield_size', np.zeros(4)), float))
        hd = getattr(self, 'hfield_data', None)
        if hd is not None:
            keep['hfield_data'] = np.ascontiguousarray(hd, np.float64).ravel()
            c.hfield_data = keep['hfield_data'].ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        for n in _INT_FIELDS:
            a = np.ascontiguousarray(getattr(self, n), np.int32)
            if a.size == 0:
                a = np.zeros(1, np.int32)
            keep[n] = a
            setattr(c, n, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        for n in _DBL_FIELDS:
            a = np.ascontiguousarray(getattr(self, n), np.float64)
            if a.size == 0:
                a = np.zeros(1, np.float64)
            keep[n] = a
            setattr(c, n, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        c._keep = keep
        return c

    # ---- convenience --------------------------------------------------------------------------
    def joint_id(self, name): return self.joint_names.index(name)
    def body_id(self, name): return self.body_names.index(name)

    def hinge_joint_names(self) -> List[str]:
        return [n for j, n in enumerate(self.joint_names) if self.jnt_type[j] != JNT_FREE]


# --------------------------------------------------------------------------------------------
# Plain-numpy kinematics / Jacobians / mass matrix.  Host-side, setup-time only (inverse weights
# at qpos0, like MuJoCo's mj_setConst); the tests also use them as an independent check of the
# oracle's CRBA.  Never on the per-step path.

def np_kinematics(m: Model, qpos) -> Dict[str, np.ndarray]:
    qpos = np.asarray(qpos, float)
    nb = m.nbody
    xpos = np.zeros((nb, 3)); xquat = np.zeros((nb, 4)); xquat[0, 0] = 1
    xipos = np.zeros((nb, 3)); xanchor = np.zeros((m.njnt, 3)); xaxis = np.zeros((m.njnt, 3))
    for i in range(1, nb):
        j = m.body_jntadr[i]; p = m.body_parentid[i]
        if j >= 0 and m.jnt_type[j] == JNT_FREE:
            a = m.jnt_qposadr[j]
            xpos[i] = qpos[a:a+3]; q = qpos[a+3:a+7]; xquat[i] = q/np.linalg.norm(q)
            xanchor[j] = xpos[i]; xaxis[j] = m.jnt_axis[j]
        else:
            R = quat2mat(xquat[p])
            xpos[i] = xpos[p] + R @ m.body_pos[i]
            xquat[i] = quat_mul(xquat[p], m.body_quat[i])
            if j >= 0:
                R = quat2mat(xquat[i])
                xaxis[j] = R @ m.jnt_axis[j]
                xanchor[j] = xpos[i] + R @ m.jnt_pos[j]
                dq = qpos[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]]
                if m.jnt_type[j] == JNT_SLIDE:
                    xpos[i] = xpos[i] + xaxis[j]*dq
                else:
                    xquat[i] = quat_mul(xquat[i], axisangle2quat(m.jnt_axis[j], dq))
                    xpos[i] = xanchor[j] - quat2mat(xquat[i]) @ m.jnt_pos[j]
        xquat[i] /= np.linalg.norm(xquat[i])
        xipos[i] = xpos[i] + quat2mat(xquat[i]) @ m.body_ipos[i]
    return dict(xpos=xpos, xquat=xquat, xipos=xipos, xanchor=xanchor, xaxis=xaxis)


def np_body_jacobian(m: Model, kin, body: int, point):
    """World-frame translational / rotational Jacobians (3 x nv) of ``point`` fixed to ``body``."""
    jp = np.zeros((3, m.nv)); jr = np.zeros((3, m.nv))
    b = body
    while b > 0:
        j = m.body_jntadr[b]
        if j >= 0:
            a = m.jnt_dofadr[j]
            if m.jnt_type[j] == JNT_FREE:
                R = quat2mat(kin['xquat'][b])
                jp[:, a:a+3] = np.eye(3)
                for k in range(3):
                    ax = R[:, k]
                    jr[:, a+3+k] = ax
                    jp[:, a+3+k] = np.cross(ax, point - kin['xpos'][b])
            elif m.jnt_type[j] == JNT_SLIDE:
                jp[:, a] = kin['xaxis'][j]
            else:
                jr[:, a] = kin['xaxis'][j]
                jp[:, a] = np.cross(kin['xaxis'][j], point - kin['xanchor'][j])
        b = m.body_parentid[b]
    return jp, jr


def np_mass_matrix(m: Model, qpos) -> np.ndarray:
    """Dense joint-space inertia from body Jacobians: sum_b m Jp'Jp + Jr' I_world Jr (+ armature)."""
    kin = np_kinematics(m, qpos)
    M = np.diag(m.dof_armature.astype(float)) if m.nv else np.zeros((0, 0))
    for b in range(1, m.nbody):
        jp, jr = np_body_jacobian(m, kin, b, kin['xipos'][b])
        R = quat2mat(quat_mul(kin['xquat'][b], m.body_iquat[b]))
        Iw = R @ np.diag(m.body_inertia[b]) @ R.T
        M = M + m.body_mass[b]*jp.T @ jp + jr.T @ Iw @ jr
    return M


# --------------------------------------------------------------------------------------------
# Canonical synthetic morphologies (BASELINE.json configs)

def _capsule_inertia(mass, radius, length):
    """Solid cylinder about its centre, long axis = x (Appendix D)."""
    ixx = 0.5*mass*radius**2
    iyy = mass*(3*radius**2 + length**2)/12.0
    return (ixx, iyy, iyy)


def salamander33(contacts: bool = False, limits: bool = False, full_actuators: bool = True,
                 timestep: float = 1e-3, spawn_z: float = -0.1, self_collisions: bool = False, terrain: str = 'plane',
                 mesh_feet: bool = False) -> Model:
    """salamander-33: free root + 11 spine hinges (z) + 4 legs x 4 hinges.

    nbody=29, njnt=28, nq=34, nv=33; ``full_actuators`` -> nu=81 (mjcf.py:819-854 triple), else nu=27.
    With ``contacts``: ``self_collisions`` adds explicit foot / trunk and foot / foot pairs (morphology.self_collisions,
    reference mjcf.py:1012-1033), ``terrain='hfield'`` replaces the plane by a gently rolling heightfield (reference
    task.py:108-123), ``mesh_feet`` replaces the foot spheres by small convex meshes (reference mjcf.py:270-413).
    """
    b = ModelBuilder('salamander33', timestep=timestep)
    n_spine, L = 12, 0.08
    radii = np.linspace(0.02, 0.008, n_spine)
    rng = (-1.2, 1.2)
    for i in range(n_spine):
        r = radii[i]
        vol = np.pi*r*r*L + 4.0/3.0*np.pi*r**3
        mass = 1000.0*vol
        kw = dict(pos=(0, 0, spawn_z) if i == 0 else (L, 0, 0), mass=mass, ipos=(L/2, 0, 0),
                  inertia=_capsule_inertia(mass, r, L))
        if i == 0:
            b.add_body('body_0', 'world', joint='free', **kw)
        else:
            b.add_body(f'body_{i}', f'body_{i-1}', joint='hinge', jname=f'joint_body_{i}', axis=(0, 0, 1),
                       damping=1e-3, limited=limits, range=rng, **kw)
        area = r/0.02
        b.set_swimming(f'body_{i}', density=1000.0,
                       drag_coefficients=[[-0.01*area, -1.0*area, -1.0*area], [-1e-5, -1e-4, -1e-4]],
                       height=0.5*np.sqrt((L/2 + r)**2))
        if contacts:
            b.add_geom(f'body_{i}', GEOM_CAPSULE, (r, L/2), pos=(L/2, 0, 0),
                       quat=axisangle2quat([0, 1, 0], np.pi/2), friction=(1.0, 0, 0))
    for attach, tag in ((1, 'front'), (5, 'hind')):
        for side, sname in ((+1, 'L'), (-1, 'R')):
            base = f'leg_{tag}_{sname}'
            parent = f'body_{attach}'
            for k, ax in enumerate(((0, 0, 1), (0, 1, 0), (1, 0, 0))):
                name = f'{base}_{k}'
                # SURVEY Appendix D suggests 1e-8 kg m^2 here with kp = 1: that puts a mode at
                # omega*h = 10 (two parallel pitch joints around a near-massless link), unstable for any
                # explicit position actuator at h = 1e-3, MuJoCo included. 1e-6 and kp = 0.1 on the limbs
                # keep every actuator mode below omega*h = 1.4.
                b.add_body(name, parent, pos=(L/2, side*0.025, 0) if k == 0 else (0, 0, 0), mass=1e-3,
                           inertia=(1e-6, 1e-6, 1e-6), joint='hinge', jname=f'joint_{name}', axis=ax,
                           damping=1e-4, limited=limits, range=rng)
                b.set_swimming(name, density=1000.0, drag_coefficients=[[-1e-3]*3, [-1e-6]*3], height=0.5*0.005)
                parent = name
            name = f'{base}_3'
            fl, fr = 0.04, 0.006
            fmass = 1000.0*(np.pi*fr*fr*fl + 4.0/3.0*np.pi*fr**3)
            ixx, iyy, izz = _capsule_inertia(fmass, fr, fl)
            # forearm hangs along -z from the elbow: long axis = z
            b.add_body(name, parent, pos=(0, side*0.02, 0), mass=fmass, ipos=(0, 0, -fl/2),
                       inertia=(iyy, iyy, ixx), joint='hinge', jname=f'joint_{name}', axis=(0, 1, 0),
                       damping=1e-4, limited=limits, range=rng)
            b.set_swimming(name, density=1000.0,
                           drag_coefficients=[[-0.3, -0.3, -0.003], [-1e-5, -1e-5, -1e-6]],
                           height=0.5*(fl/2 + fr))
            if contacts and mesh_feet:      # an icosahedron of the sphere's radius: 12 hull vertices
                g = (1 + 5**0.5)/2
                ico = np.array([(0, s1, s2*g) for s1 in (-1, 1) for s2 in (-1, 1)] + [(s1, s2*g, 0) for s1 in (-1, 1) for s2 in (-1, 1)] +
                               [(s2*g, 0, s1) for s1 in (-1, 1) for s2 in (-1, 1)], float)
                b.add_mesh_geom(name, ico*(fr/np.linalg.norm(ico[0])), pos=(0, 0, -fl), friction=(1.0, 0, 0))
            elif contacts:
                b.add_geom(name, GEOM_SPHERE, (fr,), pos=(0, 0, -fl), friction=(1.0, 0, 0))
    if contacts:
        if terrain == 'hfield':             # rolling ground, +-5 mm over a 4 m x 4 m patch of 65 x 65 samples (elevation = data * 0.01)
            gy, gx = np.meshgrid(np.linspace(-2, 2, 65), np.linspace(-2, 2, 65), indexing='ij')
            b.add_hfield(0.5*np.sin(7.0*gx)*np.cos(5.0*gy), (2.0, 2.0, 0.01, 0.1), friction=(0, 0, 0))
        else:
            b.add_geom('world', GEOM_PLANE, (0, 0, 0), friction=(0, 0, 0))     # arena friction 0 (mjcf.py:1202)
        b.options['max_contacts'] = 40 if mesh_feet else 32     # a mesh foot makes up to 4 contacts (27 limits + 4 x 40 rows <= 192)
        if self_collisions:                 # feet against the trunk segments they can reach and against each other
            for tag, trunk in (('front', (0, 2, 3)), ('hind', (4, 6, 7))):
                for sname in ('L', 'R'):
                    for t in trunk:
                        b.add_contact_pair(f'leg_{tag}_{sname}_3', f'body_{t}')
                b.add_contact_pair(f'leg_{tag}_L_3', f'leg_{tag}_R_3')
            for sname in ('L', 'R'):
                b.add_contact_pair(f'leg_front_{sname}_3', f'leg_hind_{sname}_3')
    for jn in [f'joint_body_{i}' for i in range(1, n_spine)] + [
            f'joint_leg_{t}_{s}_{k}' for t in ('front', 'hind') for s in ('L', 'R') for k in range(4)]:
        kp = 1.0 if jn.startswith('joint_body_') else 0.1
        if full_actuators:
            b.add_joint_actuators(jn, kp=kp, kv=0.0)
        else:
            b.add_position_actuator(jn, kp=kp)
    return b.compile()


def eel(n_joints: int = 20, timestep: float = 1e-3) -> Model:
    """Config-5 eel: free root + ``n_joints`` spine hinges, no limbs (nv = 6 + n_joints)."""
    b = ModelBuilder('eel', timestep=timestep)
    L, n = 0.05, n_joints + 1
    radii = np.linspace(0.015, 0.005, n)
    for i in range(n):
        r = radii[i]
        mass = 1000.0*(np.pi*r*r*L + 4.0/3.0*np.pi*r**3)
        kw = dict(pos=(0, 0, -0.1) if i == 0 else (L, 0, 0), mass=mass, ipos=(L/2, 0, 0),
                  inertia=_capsule_inertia(mass, r, L))
        if i == 0:
            b.add_body('body_0', 'world', joint='free', **kw)
        else:
            b.add_body(f'body_{i}', f'body_{i-1}', joint='hinge', jname=f'joint_body_{i}', axis=(0, 0, 1),
                       damping=5e-4, **kw)
        a = r/0.015
        b.set_swimming(f'body_{i}', drag_coefficients=[[-0.01*a, -0.8*a, -0.8*a], [-1e-7, -2e-6, -2e-6]],
                       height=0.5*(L/2 + r))
    for i in range(1, n):
        b.add_position_actuator(f'joint_body_{i}', kp=0.5)
    return b.compile()


def centipede(n_segments: int = 10, n_spine_joints: int = 15, timestep: float = 1e-3) -> Model:
    """Config-5 centipede: free + 15 spine hinges + 2x2-DoF legs on 10 segments (nv = 61)."""
    b = ModelBuilder('centipede', timestep=timestep)
    L, r = 0.03, 0.006
    n = n_spine_joints + 1
    mass = 1000.0*(np.pi*r*r*L + 4.0/3.0*np.pi*r**3)
    for i in range(n):
        kw = dict(pos=(0, 0, -0.1) if i == 0 else (L, 0, 0), mass=mass, ipos=(L/2, 0, 0),
                  inertia=_capsule_inertia(mass, r, L))
        if i == 0:
            b.add_body('body_0', 'world', joint='free', **kw)
        else:
            b.add_body(f'body_{i}', f'body_{i-1}', joint='hinge', jname=f'joint_body_{i}', axis=(0, 0, 1),
                       damping=2e-4, **kw)
        b.set_swimming(f'body_{i}', drag_coefficients=[[-0.005, -0.3, -0.3], [-1e-6, -1e-5, -1e-5]],
                       height=0.5*(L/2 + r))
        if i < n_segments:
            for side, sname in ((+1, 'L'), (-1, 'R')):
                up = f'leg_{i}_{sname}_0'
                lo = f'leg_{i}_{sname}_1'
                b.add_body(up, f'body_{i}', pos=(L/2, side*r, 0), mass=2e-4, inertia=(5e-7,)*3, joint='hinge',
                           jname=f'joint_{up}', axis=(0, 0, 1), damping=5e-5)
                lm = 1000.0*np.pi*0.0015**2*0.015
                b.add_body(lo, up, pos=(0, side*0.004, 0), mass=lm, ipos=(0, side*0.0075, 0),
                           inertia=(lm*0.015**2/12 + 2e-7, 2e-7, lm*0.015**2/12 + 2e-7), joint='hinge', jname=f'joint_{lo}',
                           axis=(1, 0, 0), damping=5e-5)
                b.set_swimming(up, drag_coefficients=[[-1e-4]*3, [-1e-7]*3], height=0.001)
                b.set_swimming(lo, drag_coefficients=[[-0.02, -0.0002, -0.02], [-1e-7]*3], height=0.004)
    for jn in [x for x in [bd.joint['name'] for bd in b.bodies[1:] if bd.joint] if not x.startswith('root_')]:
        b.add_position_actuator(jn, kp=0.2 if jn.startswith('joint_body_') else 0.05)
    return b.compile()


def wave_controller_params(m: Model, amplitude: float = 0.3, n_wave: float = 1.0):
    """Per-actuator amplitude / phase lag of the travelling-wave position controller used by the
    benchmark configs: A on the axial (``joint_body_*``) position actuators with
    phase lag 2*pi*n_wave*j/n_axial, zero elsewhere (limbs held, velocity/motor actuators idle)."""
    amp = np.zeros(m.nu)
    lag = np.zeros(m.nu)
    axial = [j for j, n in enumerate(m.joint_names) if n.startswith('joint_body_')]
    for a in range(m.nu):
        j = int(m.actuator_jntid[a])
        if m.actuator_tags[a] == 'position' and j in axial:
            amp[a] = amplitude
            lag[a] = 2*np.pi*n_wave*axial.index(j)/len(axial)
    return amp, lag


def synthetic_batch(m: Model, n_envs: int, seed: int = 0, env_offset: int = 0, perturb: float = 0.05):
    """Deterministic per-env initial state + controller phase keyed by GLOBAL env index (so sharding a
    batch over GPUs does not change any env's inputs): hinge qpos perturbation ~U(-p,p),
    phase psi ~U[0,2pi)."""
    qpos = np.tile(m.key_qpos, (n_envs, 1))
    psi = np.zeros(n_envs)
    hinge_q = m.jnt_qposadr[m.jnt_type != JNT_FREE]
    for e in range(n_envs):
        rng = np.random.default_rng([seed, env_offset + e])
        qpos[e, hinge_q] += rng.uniform(-perturb, perturb, len(hinge_q))
        psi[e] = rng.uniform(0, 2*np.pi)
    qvel = np.zeros((n_envs, m.nv))
    return qpos, qvel, psi


def trot_controller_params(m: Model, spine_amplitude: float = 0.2, limb_amplitude: float = 0.3):
    """Per-actuator amplitude / phase lag of a trot-like gait for the walking config (BASELINE configs[3]): axial
    travelling wave plus shoulder-pitch swing with diagonal limb pairs in phase (front-L/hind-R vs front-R/hind-L)."""
    amp, lag = wave_controller_params(m, spine_amplitude, 1.0)
    for a in range(m.nu):
        if m.actuator_tags[a] != 'position':
            continue
        name = m.joint_names[int(m.actuator_jntid[a])]
        if name.startswith('joint_leg_') and name.endswith('_1'):
            amp[a] = limb_amplitude
            lag[a] = 0.0 if ('front_L' in name or 'hind_R' in name) else np.pi
    return amp, lag
