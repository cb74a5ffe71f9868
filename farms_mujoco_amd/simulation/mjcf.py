"""Model compiler: SDF + options -> batched model arrays.

Counterpart of reference farms_mujoco/simulation/mjcf.py (``sdf2mjcf`` :647-1035, ``mjc_add_link`` :132-600,
``setup_mjcf_xml`` :1174-1512) without dm_control / MuJoCo's XML compiler: the SDF tree is walked exactly like
``add_link_recursive`` (:603-644) and every quantity gets the same unit scaling, but the result is a
:class:`~farms_mujoco_amd.model.Model` (flat arrays) instead of an MJCF element tree.  Rendering-only content
(visuals, meshes, textures, cameras, lights) and muscles are out of scope."""
from __future__ import annotations

import os

import numpy as np

from ..io.sdf import ModelSDF, Link
from ..model import (ModelBuilder, Model, euler2quat, quat2mat, GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, DEFAULT_SOLREF,
                     DEFAULT_SOLIMP)
from ..units import SimulationUnitScaling

MIN_MASS = 0
MIN_INERTIA = 0


def euler2mjcquat(euler):
    """Euler (scipy 'xyz': extrinsic x, y, z) to MuJoCo w,x,y,z quaternion (reference mjcf.py:47-53)."""
    return euler2quat(euler)


def euler2mat(euler):
    return quat2mat(euler2quat(euler))


def poseul2mat4d(position, euler):
    t = np.eye(4)
    t[:3, -1] = position
    t[:3, :3] = euler2mat(euler)
    return t


def get_local_transform(parent_pose, child_pose):
    """Link local transform as (position, rotation matrix) (reference mjcf.py:76-96; the reference converts the
    rotation back to Euler angles, which it then turns into a quaternion again)."""
    parent = np.eye(4) if parent_pose is None else poseul2mat4d(parent_pose[:3], parent_pose[3:])
    local = np.linalg.inv(parent) @ poseul2mat4d(child_pose[:3], child_pose[3:])
    return local[:3, -1], local[:3, :3]


def check_supported_options(simulation_options, compile_only=False):
    """What of ``simulation_options.integrator / cone / solver / noslip_iterations`` (forwarded to MuJoCo's option block by the reference,
    mjcf.py:1342-1403) the HIP step implements; anything else would silently run different physics, so it is refused HERE with the
    reason (and again by ``fmj_create``):
      integrator  Euler (with MuJoCo's implicit joint damping), implicitfast, and RK4 (round 5: four forward launches per step through
                  fmj_step - the per-iteration path, no fused launch).  ``implicit`` keeps the Coriolis derivatives, a non-symmetric
                  matrix outside the tree-sparse L'DL of this path.
      solver/cone PGS, CG or Newton with the pyramidal or the elliptic cone (round 5: MuJoCo's elliptic PGS - ray update + friction QCQP per
                  contact - on the device too, also on models with explicit pairs).
      noslip      MuJoCo's post-pass on the friction rows without regularisation: oracle and device (round 5); requested with Newton / CG
                  the device solves the step on the dual problem (fmj_solver_info says so)."""
    if simulation_options is None:
        return
    for name, supported, why in (
            ('integrator', ('euler', 'implicitfast', 'rk4'), 'MuJoCo also has implicit (Coriolis derivatives: a non-symmetric '
                                                            'matrix outside this path\'s tree-sparse factorisation)'),
            ('cone', ('pyramidal', 'elliptic'), ''), ('solver', ('pgs', 'cg', 'newton'), '')):
        value = getattr(simulation_options, name, None)
        if value is not None and str(value).lower() not in supported:
            raise NotImplementedError(f'simulation_options.{name}={value!r}: the HIP step implements {" / ".join(supported)} only' + (f' ({why})' if why else ''))
    if int(getattr(simulation_options, 'noslip_iterations', 0) or 0) < 0:
        raise ValueError('simulation_options.noslip_iterations must not be negative')


def sdf2model(sdf: ModelSDF, **kwargs) -> Model:
    """Compile one animat (reference sdf2mjcf, mjcf.py:647-1035 + the animat part of setup_mjcf_xml :1406-1481)."""
    from ..model import mat2quat
    fixed_base = kwargs.pop('fixed_base', False)
    animat_options = kwargs.pop('animat_options', None)
    simulation_options = kwargs.pop('simulation_options', None)
    units = kwargs.pop('units', simulation_options.units if simulation_options is not None else SimulationUnitScaling())
    use_actuators = kwargs.pop('use_actuators', True)
    use_collisions = kwargs.pop('use_collisions', False)
    plane = kwargs.pop('plane', False)
    hfield = kwargs.pop('hfield', None)
    friction = kwargs.pop('friction', [0, 0, 0])
    solref = kwargs.pop('solref', None)
    solimp = kwargs.pop('solimp', None)
    # actuator ctrl / force limit options (reference mjcf.py:675-684)
    act_pos = dict(ctrllimited=kwargs.pop('act_pos_ctrllimited', False), ctrlrange=kwargs.pop('act_pos_ctrlrange', [-1e6, 1e6]),
                   forcelimited=kwargs.pop('act_pos_forcelimited', False), forcerange=kwargs.pop('act_pos_forcerange', [-1e6, 1e6]))
    act_vel = dict(ctrllimited=kwargs.pop('act_vel_ctrllimited', False), ctrlrange=kwargs.pop('act_vel_ctrlrange', [-1e6, 1e6]),
                   forcelimited=kwargs.pop('act_vel_forcelimited', False), forcerange=kwargs.pop('act_vel_forcerange', [-1e6, 1e6]))
    assert not kwargs, kwargs
    check_supported_options(simulation_options, compile_only=True)

    timestep = 1e-3
    gravity = [0.0, 0.0, -9.81]
    if simulation_options is not None:                      # reference mjcf.py:1187-1192,1329,1336-1341
        timestep = simulation_options.timestep/max(int(simulation_options.num_sub_steps), 1)
        gravity = [g*units.acceleration for g in simulation_options.gravity]
    b = ModelBuilder(sdf.name, timestep=timestep, gravity=gravity)
    if simulation_options is not None:
        b.options['solver_iterations'] = int(simulation_options.n_solver_iters)
        b.options['impratio'] = float(simulation_options.impratio)
        b.options['solver'] = str(getattr(simulation_options, 'solver', 'PGS'))       # mjcf.py:1348-1353
        b.options['cone'] = str(getattr(simulation_options, 'cone', 'pyramidal'))     # mjcf.py:1342-1347
        b.options['noslip_iterations'] = int(getattr(simulation_options, 'noslip_iterations', 0))        # mjcf.py:1392-1397
        b.options['noslip_tolerance'] = float(getattr(simulation_options, 'noslip_tolerance', 1e-6))    # mjcf.py:1398-1403
        b.options['integrator'] = str(getattr(simulation_options, 'integrator', 'Euler') or 'Euler')        # mjcf.py:1360-1365

    link_opts = {l.name: l for l in animat_options.morphology.links} if animat_options is not None else {}
    joint_opts = {j.name: j for j in animat_options.morphology.joints} if animat_options is not None else {}
    motors = {m_.joint_name: m_ for m_ in animat_options.control.motors} if animat_options is not None else {}
    joints_equations = {m_.joint_name: m_.equation for m_ in motors.values() if hasattr(m_, 'equation')}
    muscles = ({mu_.joint_name: mu_ for mu_ in animat_options.control.muscles}
               if animat_options is not None and getattr(animat_options.control, 'muscles', None) is not None else {})

    # wrapper body of the model with the free joint (reference mjcf.py:176-179,716-726); spawn pose: :1409-1413
    spawn = animat_options.spawn.pose if animat_options is not None and getattr(animat_options, 'spawn', None) else sdf.pose
    root_name = sdf.name
    b.add_body(root_name, 'world', pos=[p*units.meters for p in spawn[:3]], quat=euler2mjcquat(spawn[3:]),
               joint=None if fixed_base else 'free')

    def add_link(link: Link, parent_link, parent_name, joint):
        pos, rot = get_local_transform(None if parent_link is None else parent_link.pose, link.pose)
        kw = dict(pos=[p*units.meters for p in pos], quat=mat2quat(rot))
        if link.inertial is not None:                       # reference mjcf.py:536-589
            I = link.inertial.inertias
            mat = np.array([[max(MIN_INERTIA, I[0]), I[1], I[2]], [I[1], max(MIN_INERTIA, I[3]), I[4]],
                            [I[2], I[4], max(MIN_INERTIA, I[5])]])
            assert (np.linalg.eigvalsh(mat) > 0).all(), f'Eigen values <= 0 for link {link.name}'
            R = euler2mat(link.inertial.pose[3:])
            mat = R @ mat @ R.T
            kw.update(mass=link.inertial.mass*units.kilograms, ipos=[p*units.meters for p in link.inertial.pose[:3]],
                      fullinertia=[mat[0, 0]*units.inertia, mat[1, 1]*units.inertia, mat[2, 2]*units.inertia,
                                   mat[0, 1]*units.inertia, mat[0, 2]*units.inertia, mat[1, 2]*units.inertia])
        jkw = {}
        if joint is not None and joint.type in ('revolute', 'continuous', 'prismatic'):    # :181-212
            jo = joint_opts.get(joint.name)
            jkw = dict(joint='slide' if joint.type == 'prismatic' else 'hinge', jname=joint.name, axis=joint.axis.xyz,
                       jpos=[p*units.meters for p in joint.pose[:3]], limited=joint.axis.limits is not None,
                       range=joint.axis.limits[:2] if joint.axis.limits is not None else (0.0, 0.0))
            stiffness = damping = 0.0
            if jo is not None:                              # :1426-1444
                stiffness += jo.stiffness*units.angular_stiffness
                damping += jo.damping*units.angular_damping
                extras = getattr(jo, 'extras', {}) or {}
                if extras.get('solreflimit'):
                    sr = list(extras['solreflimit'])
                    if all(s_ < 0 for s_ in sr):
                        sr[0] *= units.newtons/units.meters; sr[1] *= units.newtons/units.velocity
                    else:
                        sr[0] *= units.seconds
                    jkw['solreflimit'] = sr
                if extras.get('solimplimit'):
                    jkw['solimplimit'] = extras['solimplimit']
                if extras.get('margin'):
                    jkw['margin'] = extras['margin']
                jkw['qpos0'] = 0.0
            mo = motors.get(joint.name)
            if (mo is not None and hasattr(mo, 'equation') and getattr(mo, 'passive', None) is not None
                    and mo.passive.is_passive):                                                    # :1448-1462
                stiffness += mo.passive.stiffness_coefficient*units.angular_stiffness
                damping += mo.passive.damping_coefficient*units.angular_damping
            mu_ = muscles.get(joint.name)
            if mu_ is not None and 'ekeberg' in joints_equations[joint.name]:                      # :1464-1481
                stiffness += mu_.beta*mu_.gamma*units.angular_stiffness
                damping += mu_.delta*units.angular_damping
            jkw.update(stiffness=stiffness, damping=damping)
        b.add_body(link.name, parent_name, **kw, **jkw)
        lo = link_opts.get(link.name)
        mesh_verts = {}                                     # collision index -> vertices of its mesh file (link units)

        def mesh_of(ci):
            if ci not in mesh_verts:
                from ..io.mesh import read_vertices
                g_ = link.collisions[ci].geometry
                uri = g_.uri[len('file://'):] if g_.uri.startswith('file://') else g_.uri
                mesh_verts[ci] = read_vertices(os.path.join(sdf.directory, os.path.expandvars(uri)), scale=g_.size[:3])
            return mesh_verts[ci]
        rbound = 0.0
        if link.collisions:
            g0 = link.collisions[0].geometry
            # a mesh: half the diagonal of its bounding box (MuJoCo's geom_rbound of a mesh is taken about the mesh centre)
            rbound = (0.5*float(np.linalg.norm(np.ptp(mesh_of(0), axis=0))) if g0.kind == 'mesh' else g0.bounding_radius())
        if lo is not None and getattr(lo, 'swimming', False):
            height = getattr(lo, 'height', None)
            b.set_swimming(link.name, density=lo.density, drag_coefficients=lo.drag_coefficients,
                           height=height if height is not None else 0.5*rbound)          # drag.pyx:364-372
        if use_collisions:
            fr = list(lo.friction) if lo is not None and getattr(lo, 'friction', None) is not None else list(friction)
            for ci, col in enumerate(link.collisions):      # :245-267 (margin 0, condim 3)
                g = col.geometry
                gkw = dict(pos=[p*units.meters for p in col.pose[:3]], quat=euler2mjcquat(col.pose[3:]), friction=fr,
                           solref=solref if solref is not None else DEFAULT_SOLREF,
                           solimp=solimp if solimp is not None else DEFAULT_SOLIMP)
                if g.kind == 'sphere':
                    b.add_geom(link.name, GEOM_SPHERE, (g.size[0]*units.meters,), **gkw)
                elif g.kind == 'capsule':
                    b.add_geom(link.name, GEOM_CAPSULE, (g.size[0]*units.meters, 0.5*g.size[1]*units.meters), **gkw)
                elif g.kind == 'cylinder':              # reference mjcf.py:427-440: radius, half length
                    b.add_geom(link.name, GEOM_CYLINDER, (g.size[0]*units.meters, 0.5*g.size[1]*units.meters), **gkw)
                elif g.kind == 'box':                   # SDF box size = full edge lengths, MuJoCo box size = half extents
                    b.add_geom(link.name, GEOM_BOX, tuple(0.5*x*units.meters for x in g.size[:3]), **gkw)
                elif g.kind == 'mesh':                  # :270-413: a MuJoCo mesh geom collides as its convex hull
                    b.add_mesh_geom(link.name, mesh_of(ci)*units.meters, **gkw)     # by position: collisions may share a name
                else:
                    raise NotImplementedError(f'collision shape {g.kind!r} of link {link.name} is outside the HIP subset '
                                              '(sphere / capsule / cylinder / box / convex mesh against planes and a heightfield)')
        for child in sdf.get_children(link):
            add_link(child, link, link.name, sdf.get_parent_joint(child))

    for root in sdf.get_base_links():
        add_link(root, None, root_name, None)
    if plane:
        b.add_geom('world', GEOM_PLANE, (0, 0, 0), friction=(0, 0, 0))            # arena friction 0 (mjcf.py:1202)
        b.options['max_contacts'] = int(plane) if not isinstance(plane, bool) else 32
    if hfield is not None:                                                        # arena heightmap (mjcf.py:486-522)
        b.add_hfield(hfield['data'], hfield['size'], pos=hfield.get('pos', (0, 0, 0)), quat=hfield.get('quat', (1, 0, 0, 0)),
                     friction=(0, 0, 0))
        b.options['max_contacts'] = max(int(b.options['max_contacts']), 32)

    # self-collisions (:1005-1033): one explicit pair per pair of collision shapes of the two links, condim 3, friction 0
    if animat_options is not None and getattr(animat_options.morphology, 'self_collisions', None):
        assert use_collisions, 'morphology.self_collisions needs the collision geoms (use_collisions)'
        for link1, link2 in animat_options.morphology.self_collisions:
            b.add_contact_pair(link1, link2, friction=0.0, solref=solref if solref is not None else DEFAULT_SOLREF)
        b.options['max_contacts'] = max(int(b.options['max_contacts']), 32)
    # actuators: position / velocity / motor per joint (:791-866)
    if use_actuators:
        joint_names = (animat_options.control.joints_names() if animat_options is not None
                       else [j.name for j in sdf.joints if j.type in ('revolute', 'continuous', 'prismatic')])
        for jn in joint_names:
            mo = motors.get(jn)
            gains = getattr(mo, 'gains', None) if mo is not None else None
            lim = getattr(mo, 'limits_torque', None) if mo is not None else None
            b.add_joint_actuators(jn, kp=gains[0]*units.torques if gains else 0.0,
                                  kv=gains[1]*units.angular_damping if gains else 0.0,
                                  forcerange=[t*units.torques for t in lim] if lim is not None else None,
                                  pos_limits=dict(act_pos, forcerange=[v_*units.torques for v_ in act_pos['forcerange']]),
                                  vel_limits=dict(act_vel, ctrlrange=[v_*units.angular_velocity for v_ in act_vel['ctrlrange']],
                                                  forcerange=[v_*units.torques for v_ in act_vel['forcerange']]))
    m = b.compile()
    # keyframe "initial" (:744-788): joint initial positions / velocities, spawn velocity
    for jo in joint_opts.values():
        if jo.name in m.joint_names:
            j = m.joint_id(jo.name)
            m.key_qpos[m.jnt_qposadr[j]] = jo.initial[0]
            m.key_qvel[m.jnt_dofadr[j]] = jo.initial[1]
    if not fixed_base and animat_options is not None and getattr(animat_options.spawn, 'velocity', None) is not None:
        v = animat_options.spawn.velocity
        m.key_qvel[:3] = [x*units.velocity for x in v[:3]]
        m.key_qvel[3:6] = [x*units.angular_velocity for x in v[3:6]]
    return m


def arena_heightfield(arena_options, units=None):
    """The heightmap collision of the arena SDF as a heightfield description (reference mjcf.py:486-522 for the asset,
    :1195-1212 for the arena pose, task.py:108-115 for the data: hfield_data = 2 (image - 0.5), image normalised to
    [0, 1] and flipped to Cartesian rows).  None when the arena SDF holds no heightmap."""
    import os
    from ..io.png import imread
    units = units or SimulationUnitScaling()
    path = getattr(arena_options, 'sdf', None)
    if not path:
        return None
    arena = ModelSDF.read(os.path.expandvars(path))[0]
    for link in arena.links:
        for col in link.collisions:
            g = col.geometry
            if g is None or g.kind != 'heightmap':
                continue
            img = imread(os.path.join(arena.directory, g.uri))
            img = img[:, :, 0] if img.ndim == 3 else img[:, :]                      # RGB vs grey (:492)
            vmin, vmax = np.iinfo(img.dtype).min, np.iinfo(img.dtype).max
            data = np.flip((img.astype(float) - vmin)/(vmax - vmin), axis=0)       # normalise, Cartesian rows (:493-495)
            # the reference OVERWRITES the arena base link's pose with arena_options.spawn.pose (+ ground_height), :1207-1211; the
            # collision keeps its own pose under that body: world pose = spawn pose o collision pose (rotations composed, not
            # Euler angles added; the SDF link pose of the base link plays no part)
            spawn = np.array(getattr(getattr(arena_options, 'spawn', None), 'pose', [0]*6), float)
            base_pos = spawn[:3].copy()
            if getattr(arena_options, 'ground_height', None) is not None:
                base_pos[2] += arena_options.ground_height                        # :1210-1211
            base_quat = euler2mjcquat(spawn[3:])
            from ..model import quat_mul, quat2mat
            pos = base_pos + quat2mat(base_quat) @ np.array(col.pose[:3], float)
            return dict(data=2*(data - 0.5),                                        # task.py:115
                        size=(0.5*g.size[0]*units.meters, 0.5*g.size[1]*units.meters, 0.5*g.size[2]*units.meters,
                              g.size[2]*units.meters),                              # :505-510
                        pos=pos*units.meters, quat=quat_mul(base_quat, euler2mjcquat(col.pose[3:])),
                        image=data)
    return None


def setup_model(simulation_options, animat_options, arena_options=None, **kwargs) -> Model:
    """setup_mjcf_xml counterpart (reference mjcf.py:1174-1512): read the animat SDF named by the options and
    compile it; a flat arena with ``ground_height`` becomes the collision plane, an arena SDF with a heightmap the
    heightfield."""
    sdf = ModelSDF.read(animat_options.sdf)[0]
    units = simulation_options.units if simulation_options is not None else SimulationUnitScaling()
    hfield = arena_heightfield(arena_options, units) if arena_options is not None else None
    plane = hfield is None and arena_options is not None and getattr(arena_options, 'ground_height', None) is not None
    mujoco_kw = dict(getattr(animat_options, 'mujoco', {}) or {})
    return sdf2model(sdf, animat_options=animat_options, simulation_options=simulation_options, hfield=hfield,
                     fixed_base=mujoco_kw.pop('fixed_base', False),
                     use_collisions=plane or hfield is not None or bool(getattr(animat_options.morphology, 'self_collisions', None)),
                     plane=plane,
                     **{k: v for k, v in mujoco_kw.items()
                        if k in ('solref', 'solimp', 'friction') or k.startswith(('act_pos_', 'act_vel_'))}, **kwargs)


def model2mjcf_xml(m: Model, fusestatic: bool = True) -> str:
    """The compiled model as an MJCF document (what ``Simulation.save_mjcf_xml`` writes in the reference,
    simulation.py:215-225 via ``mjcf.export_with_assets``): compiler / option / size blocks as mjcf.py:1244-1403 sets them,
    the body tree with explicit inertials, joints, collision geoms (animat geoms collide with the arena only:
    mjcf.py:251-267,1415-1424), the heightfield asset with its elevation data (task.py:108-115), convex meshes, the explicit
    self-collision pairs (mjcf.py:1012-1033), the actuator triple and the sensors of mjcf.py:950-1002, keyframe 0.
    The text loads in MuJoCo (tests/test_vs_mujoco.py steps it next to the oracle when ``mujoco`` is importable); nothing in
    this package reads it back.  ``fusestatic=False`` keeps jointless bodies as bodies of their own, so that MuJoCo's body
    ids are this model's (the reference compiles with fusestatic=True, mjcf.py:1252: same physics, fewer bodies)."""
    import xml.etree.ElementTree as ET
    from ..model import JNT_FREE, JNT_SLIDE, GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH

    def v(a):
        return ' '.join(repr(float(x)) for x in np.asarray(a).ravel())
    root = ET.Element('mujoco', model=str(getattr(m, 'name', 'animat')))
    ET.SubElement(root, 'compiler', angle='radian', eulerseq='xyz', inertiafromgeom='false', balanceinertia='false',
                  boundmass='0', boundinertia='0', fusestatic='true' if fusestatic else 'false')
    ET.SubElement(root, 'option', timestep=repr(float(m.timestep)), gravity=v(m.gravity), integrator={0: 'Euler', 1: 'RK4', 2: 'implicit', 3: 'implicitfast'}[int(getattr(m, 'integrator', 0))], cone={0: 'pyramidal', 1: 'elliptic'}[int(getattr(m, 'cone', 0))],
                  solver={0: 'PGS', 1: 'CG', 2: 'Newton'}[int(getattr(m, 'solver', 0))], iterations=str(int(m.solver_iterations)), tolerance=repr(float(m.solver_tolerance)),
                  ls_iterations=str(int(getattr(m, 'ls_iterations', 50))), ls_tolerance=repr(float(getattr(m, 'ls_tolerance', 0.01))),
                  impratio=repr(float(m.impratio)), noslip_iterations=str(int(getattr(m, 'noslip_iterations', 0))),
                  noslip_tolerance=repr(float(getattr(m, 'noslip_tolerance', 1e-6))))
    ET.SubElement(root, 'size', nconmax=str(max(int(m.max_contacts), 1)), nkey='1')
    asset = ET.SubElement(root, 'asset')
    world = ET.SubElement(root, 'worldbody')
    elems = {0: world}
    gtypes = {GEOM_PLANE: 'plane', GEOM_HFIELD: 'hfield', GEOM_SPHERE: 'sphere', GEOM_CAPSULE: 'capsule', GEOM_CYLINDER: 'cylinder', GEOM_BOX: 'box'}
    for b in range(1, m.nbody):
        e = ET.SubElement(elems[int(m.body_parentid[b])], 'body', name=m.body_names[b], pos=v(m.body_pos[b]), quat=v(m.body_quat[b]))
        elems[b] = e
        ET.SubElement(e, 'inertial', pos=v(m.body_ipos[b]), quat=v(m.body_iquat[b]), mass=repr(float(m.body_mass[b])),
                      diaginertia=v(m.body_inertia[b]))
        j = int(m.body_jntadr[b])
        if j >= 0:
            if m.jnt_type[j] == JNT_FREE:
                ET.SubElement(e, 'freejoint', name=m.joint_names[j])
            else:
                d = int(m.jnt_dofadr[j])
                at = dict(name=m.joint_names[j], type='slide' if m.jnt_type[j] == JNT_SLIDE else 'hinge', pos=v(m.jnt_pos[j]),
                          axis=v(m.jnt_axis[j]), stiffness=repr(float(m.jnt_stiffness[j])), damping=repr(float(m.dof_damping[d])),
                          armature=repr(float(m.dof_armature[d])), ref=repr(float(m.qpos0[m.jnt_qposadr[j]])),
                          springref=repr(float(m.qpos_spring[m.jnt_qposadr[j]])))
                if m.jnt_limited[j]:
                    at.update(limited='true', range=v(m.jnt_range[j]), margin=repr(float(m.jnt_margin[j])),
                              solreflimit=v(m.jnt_solref[j]), solimplimit=v(m.jnt_solimp[j]))
                ET.SubElement(e, 'joint', **at)
    for g in range(m.ngeom):
        t = int(m.geom_type[g])
        ground = int(m.geom_bodyid[g]) == 0
        # animat geoms: contype 1 / conaffinity 0; arena geoms: 1 / 1 -> animat against arena only (the self-collisions are explicit pairs)
        common = dict(name=f'geom_{g}', pos=v(m.geom_pos[g]), quat=v(m.geom_quat[g]), friction=v(m.geom_friction[g]), solref=v(m.geom_solref[g]),
                      solimp=v(m.geom_solimp[g]), condim='3', margin='0', contype='1', conaffinity='1' if ground else '0')
        if t == GEOM_MESH:                           # convex mesh: its hull vertices as an inline mesh asset
            a0, n = int(m.geom_vertadr[g]), int(m.geom_vertnum[g])
            ET.SubElement(asset, 'mesh', name=f'mesh_{g}', vertex=v(np.asarray(m.mesh_vert[a0:a0 + n]).ravel()))
            ET.SubElement(elems[int(m.geom_bodyid[g])], 'geom', type='mesh', mesh=f'mesh_{g}', **common)
            continue
        if t == GEOM_HFIELD:                         # mjModel.hfield_*: nrow x ncol samples in [0, 1] scaled by size[2] (+ base size[3])
            data = np.asarray(m.hfield_data, float).reshape(int(m.hfield_nrow), int(m.hfield_ncol))
            hs = np.asarray(m.hfield_size, float)
            # MuJoCo normalises the elevation data of an asset to [0, 1] and scales by size[2]: the model's data may be
            # signed (the reference stores 2 (image - 0.5), task.py:115), so the asset is written as (data - min) / range with the
            # elevation range as size[2] and the geom lowered by -min * size[2]: the same surface
            lo, hi = float(data.min()), float(data.max())
            rng_ = hi - lo if hi > lo else 1.0
            ET.SubElement(asset, 'hfield', name='hfield_0', nrow=str(int(m.hfield_nrow)), ncol=str(int(m.hfield_ncol)),
                          size=v([hs[0], hs[1], hs[2]*rng_ if hs[2]*rng_ > 0 else 1e-9, max(hs[3], 1e-9)]),
                          elevation=v(((data - lo)/rng_).ravel()))
            from ..model import quat2mat
            off = quat2mat(np.asarray(m.geom_quat[g], float)) @ np.array([0.0, 0.0, lo*hs[2]])
            hcommon = dict(common, pos=v(np.asarray(m.geom_pos[g], float) + off))
            ET.SubElement(elems[int(m.geom_bodyid[g])], 'geom', type='hfield', hfield='hfield_0', **hcommon)
            continue
        size = {GEOM_PLANE: [0, 0, 0.1], GEOM_SPHERE: m.geom_size[g][:1], GEOM_CAPSULE: m.geom_size[g][:2], GEOM_CYLINDER: m.geom_size[g][:2], GEOM_BOX: m.geom_size[g][:3]}[t]
        ET.SubElement(elems[int(m.geom_bodyid[g])], 'geom', type=gtypes[t], size=v(size), **common)
    if not len(asset):
        root.remove(asset)
    if int(getattr(m, 'npair', 0)):                  # explicit self-collision pairs (mjcf.py:1012-1033)
        con = ET.SubElement(root, 'contact')
        for p_ in range(int(m.npair)):
            mu = float(m.pair_friction[p_])
            ET.SubElement(con, 'pair', name=f'pair_{p_}', geom1=f'geom_{int(m.pair_geom1[p_])}', geom2=f'geom_{int(m.pair_geom2[p_])}', condim='3',
                          friction=v([mu, mu, 0.005, 0.0001, 0.0001]), solref=v(m.pair_solref[p_]), solimp=v(m.pair_solimp[p_]), margin='0', gap='0')
    if m.nu:
        act = ET.SubElement(root, 'actuator')
        for a in range(m.nu):
            at = dict(name=m.actuator_names[a], joint=m.joint_names[int(m.actuator_jntid[a])], gainprm=repr(float(m.actuator_gain[a])),
                      biasprm=v(m.actuator_bias[a]), biastype='affine')
            if m.actuator_ctrllimited[a]:
                at.update(ctrllimited='true', ctrlrange=v(m.actuator_ctrlrange[a]))
            if m.actuator_forcelimited[a]:
                at.update(forcelimited='true', forcerange=v(m.actuator_forcerange[a]))
            ET.SubElement(act, 'general', **at)
    sens = ET.SubElement(root, 'sensor')
    for b in range(1, m.nbody):
        ET.SubElement(sens, 'framelinvel', name=f'framelinvel_{m.body_names[b]}', objtype='body', objname=m.body_names[b])
        ET.SubElement(sens, 'frameangvel', name=f'frameangvel_{m.body_names[b]}', objtype='body', objname=m.body_names[b])
    for j in range(m.njnt):
        if m.jnt_type[j] != JNT_FREE:
            for kind in ('jointpos', 'jointvel', 'jointlimitfrc'):
                ET.SubElement(sens, kind, name=f'{kind}_{m.joint_names[j]}', joint=m.joint_names[j])
    for a in range(m.nu):
        ET.SubElement(sens, 'actuatorfrc', name=f'actuatorfrc_{m.actuator_tags[a]}_{m.joint_names[int(m.actuator_jntid[a])]}',
                      actuator=m.actuator_names[a])
    key = ET.SubElement(root, 'keyframe')
    ET.SubElement(key, 'key', name='initial', qpos=v(m.key_qpos), qvel=v(getattr(m, 'key_qvel', np.zeros(m.nv))))
    ET.indent(root) if hasattr(ET, 'indent') else None
    return ET.tostring(root, encoding='unicode')
