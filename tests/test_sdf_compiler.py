"""SDF + options -> model compiler (counterpart of reference mjcf.py sdf2mjcf / setup_mjcf_xml)."""
import numpy as np
import pytest

from farms_mujoco_amd.io.sdf import ModelSDF
from farms_mujoco_amd.model import ModelBuilder, np_mass_matrix, np_kinematics, euler2quat, quat2mat, JNT_FREE
from farms_mujoco_amd.options import AnimatOptions, SimulationOptions, ArenaOptions
from farms_mujoco_amd.simulation.mjcf import sdf2model, setup_model, get_local_transform
from farms_mujoco_amd.units import SimulationUnitScaling

SDF = """<?xml version="1.0"?>
<sdf version="1.6">
  <model name="swimmer">
    <pose>0 0 0 0 0 0</pose>
    <link name="head">
      <pose>0 0 0 0 0 0</pose>
      <inertial><pose>0.05 0 0 0 0 0</pose><mass>0.10</mass>
        <inertia><ixx>2e-5</ixx><ixy>0</ixy><ixz>0</ixz><iyy>9e-5</iyy><iyz>0</iyz><izz>9e-5</izz></inertia></inertial>
      <collision name="head_col"><pose>0.05 0 0 0 1.5707963267948966 0</pose>
        <geometry><capsule><radius>0.02</radius><length>0.1</length></capsule></geometry></collision>
    </link>
    <link name="trunk">
      <pose>0.1 0 0 0 0 0.3</pose>
      <inertial><pose>0.05 0 0 0 0 0.2</pose><mass>0.08</mass>
        <inertia><ixx>1.5e-5</ixx><ixy>1e-6</ixy><ixz>0</ixz><iyy>7e-5</iyy><iyz>0</iyz><izz>7e-5</izz></inertia></inertial>
      <collision name="trunk_col"><pose>0.05 0 0 0 0 0</pose><geometry><sphere><radius>0.02</radius></sphere></geometry></collision>
    </link>
    <link name="tail">
      <pose>0.19553365 0.02955202 0 0 0 0.3</pose>
      <inertial><pose>0.04 0 0 0 0 0</pose><mass>0.04</mass>
        <inertia><ixx>5e-6</ixx><ixy>0</ixy><ixz>0</ixz><iyy>2e-5</iyy><iyz>0</iyz><izz>2e-5</izz></inertia></inertial>
    </link>
    <link name="fin">
      <pose>0.15 0.03 0 0 0 1.0</pose>
      <inertial><pose>0.01 0 0 0 0 0</pose><mass>0.005</mass>
        <inertia><ixx>1e-6</ixx><ixy>0</ixy><ixz>0</ixz><iyy>1e-6</iyy><iyz>0</iyz><izz>1e-6</izz></inertia></inertial>
    </link>
    <joint name="j_trunk" type="revolute"><parent>head</parent><child>trunk</child><pose>0 0 0 0 0 0</pose>
      <axis><xyz>0 0 1</xyz><limit><lower>-1.0</lower><upper>1.0</upper></limit></axis></joint>
    <joint name="j_tail" type="revolute"><parent>trunk</parent><child>tail</child><pose>0 0 0 0 0 0</pose>
      <axis><xyz>0 0 1</xyz></axis></joint>
    <joint name="j_fin" type="continuous"><parent>trunk</parent><child>fin</child><pose>0.002 0 0 0 0 0</pose>
      <axis><xyz>0 1 0</xyz></axis></joint>
  </model>
</sdf>
"""


@pytest.fixture
def sdf_path(tmp_path):
    p = tmp_path/'swimmer.sdf'
    p.write_text(SDF)
    return str(p)


def _options(sdf_path):
    links = [AnimatOptions.link(n, swimming=True, drag_coefficients=[[-0.01, -0.5, -0.5], [-1e-6, -1e-5, -1e-5]])
             for n in ('head', 'trunk', 'tail', 'fin')]
    joints = [AnimatOptions.joint('j_trunk', initial=(0.1, 0.0), damping=1e-3, stiffness=0.02),
              AnimatOptions.joint('j_tail', initial=(-0.2, 0.5), damping=2e-3), AnimatOptions.joint('j_fin', damping=1e-4)]
    motors = [AnimatOptions.motor('j_trunk', gains=(0.5, 0.01)), AnimatOptions.motor('j_tail', gains=(0.4, 0.0), limits_torque=[-0.3, 0.3]),
              AnimatOptions.motor('j_fin', gains=(0.05, 0.0))]
    return AnimatOptions(name='swimmer', links=links, joints=joints, motors=motors, sdf=sdf_path,
                         spawn_pose=(0.1, -0.2, -0.05, 0.0, 0.0, 0.4), spawn_velocity=(0.1, 0, 0, 0, 0, 0.2))


def test_reader(sdf_path):
    sdf = ModelSDF.read(sdf_path)[0]
    assert sdf.name == 'swimmer' and [l.name for l in sdf.links] == ['head', 'trunk', 'tail', 'fin']
    assert [l.name for l in sdf.get_base_links()] == ['head']
    assert [l.name for l in sdf.get_children(sdf.links[1])] == ['tail', 'fin']
    assert sdf.get_parent_joint(sdf.links[3]).name == 'j_fin' and sdf.joints[0].axis.limits.tolist() == [-1.0, 1.0]
    assert sdf.links[0].collisions[0].geometry.kind == 'capsule'
    assert abs(sdf.links[0].collisions[0].geometry.bounding_radius() - 0.07) < 1e-12


def test_compiled_structure_and_options(sdf_path):
    ao = _options(sdf_path)
    m = setup_model(SimulationOptions(timestep=2e-3, num_sub_steps=2), ao, ArenaOptions())
    assert m.body_names == ['world', 'swimmer', 'head', 'trunk', 'tail', 'fin']           # wrapper + DFS order
    assert m.joint_names == ['root_swimmer', 'j_trunk', 'j_tail', 'j_fin'] and m.jnt_type[0] == JNT_FREE
    assert (m.nq, m.nv, m.nu) == (10, 9, 9) and m.timestep == 1e-3                        # timestep / num_sub_steps
    assert m.actuator_names[:3] == ['actuator_position_j_trunk', 'actuator_velocity_j_trunk', 'actuator_torque_j_trunk']
    j = m.joint_id('j_trunk')
    assert m.jnt_limited[j] == 1 and m.jnt_range[j].tolist() == [-1.0, 1.0] and m.jnt_stiffness[j] == 0.02
    assert m.dof_damping[m.jnt_dofadr[m.joint_id('j_tail')]] == 2e-3
    a = m.actuator_names.index('actuator_position_j_tail')
    assert m.actuator_gain[a] == 0.4 and m.actuator_forcelimited[a] == 1 and m.actuator_forcerange[a].tolist() == [-0.3, 0.3]
    # spawn pose -> root body / keyframe; joint initial state -> keyframe (mjcf.py:744-788)
    assert np.allclose(m.key_qpos[:3], [0.1, -0.2, -0.05]) and np.allclose(m.key_qpos[3:7], euler2quat([0, 0, 0.4]))
    assert m.key_qpos[m.jnt_qposadr[m.joint_id('j_tail')]] == -0.2 and m.key_qvel[m.jnt_dofadr[m.joint_id('j_tail')]] == 0.5
    assert np.allclose(m.key_qvel[:6], [0.1, 0, 0, 0, 0, 0.2])
    # swimming heights default to half the bounding radius of the first collision geom (drag.pyx:364-372)
    sw = {s['name']: s for s in m.swimming}
    assert abs(sw['head']['height'] - 0.035) < 1e-12 and abs(sw['trunk']['height'] - 0.01) < 1e-12


def test_geometry_matches_hand_built_model(sdf_path, oracle):
    """World poses at qpos0 reproduce the SDF link poses; mass matrix and one oracle step equal a hand-built model."""
    ao = _options(sdf_path)
    m = sdf2model(ModelSDF.read(sdf_path)[0], animat_options=ao)
    sdf = ModelSDF.read(sdf_path)[0]
    q0 = m.qpos0.copy(); q0[:3] = 0; q0[3:7] = [1, 0, 0, 0]                                # undo the spawn pose
    kin = np_kinematics(m, q0)
    for link in sdf.links:
        b = m.body_id(link.name)
        assert np.allclose(kin['xpos'][b], link.pose[:3], atol=1e-8)
        assert np.allclose(quat2mat(kin['xquat'][b]), quat2mat(euler2quat(link.pose[3:])), atol=1e-8)
    # joint anchor of the fin = joint pose in the child frame
    R = quat2mat(euler2quat([0, 0, 1.0]))
    assert np.allclose(kin['xanchor'][m.joint_id('j_fin')], np.array([0.15, 0.03, 0]) + R @ [0.002, 0, 0], atol=1e-8)
    # hand-built equivalent
    hb = ModelBuilder('swimmer')
    hb.add_body('swimmer', 'world', pos=(0.1, -0.2, -0.05), quat=euler2quat([0, 0, 0.4]), joint='free')
    hb.add_body('head', 'swimmer', mass=0.10, ipos=(0.05, 0, 0), fullinertia=(2e-5, 9e-5, 9e-5, 0, 0, 0))
    Rt = quat2mat(euler2quat([0, 0, 0.2])); It = Rt @ np.array([[1.5e-5, 1e-6, 0], [1e-6, 7e-5, 0], [0, 0, 7e-5]]) @ Rt.T
    hb.add_body('trunk', 'head', pos=(0.1, 0, 0), quat=euler2quat([0, 0, 0.3]), mass=0.08, ipos=(0.05, 0, 0),
                fullinertia=(It[0, 0], It[1, 1], It[2, 2], It[0, 1], It[0, 2], It[1, 2]), joint='hinge', jname='j_trunk',
                axis=(0, 0, 1), damping=1e-3, stiffness=0.02, limited=True, range=(-1, 1))
    p_tail, _ = get_local_transform([0.1, 0, 0, 0, 0, 0.3], [0.19553365, 0.02955202, 0, 0, 0, 0.3])
    hb.add_body('tail', 'trunk', pos=p_tail, mass=0.04, ipos=(0.04, 0, 0), fullinertia=(5e-6, 2e-5, 2e-5, 0, 0, 0),
                joint='hinge', jname='j_tail', axis=(0, 0, 1), damping=2e-3)
    p_fin, R_fin = get_local_transform([0.1, 0, 0, 0, 0, 0.3], [0.15, 0.03, 0, 0, 0, 1.0])
    from farms_mujoco_amd.model import mat2quat
    hb.add_body('fin', 'trunk', pos=p_fin, quat=mat2quat(R_fin), mass=0.005, ipos=(0.01, 0, 0), fullinertia=(1e-6,)*3 + (0, 0, 0),
                joint='hinge', jname='j_fin', axis=(0, 1, 0), jpos=(0.002, 0, 0), damping=1e-4)
    for jn, kp, kv, fr in (('j_trunk', 0.5, 0.01, None), ('j_tail', 0.4, 0.0, (-0.3, 0.3)), ('j_fin', 0.05, 0.0, None)):
        hb.add_joint_actuators(jn, kp=kp, kv=kv, forcerange=fr)
    h = hb.compile()
    rng = np.random.default_rng(0)
    q = m.qpos0.copy(); q[7:] += rng.uniform(-0.4, 0.4, 3); v = rng.normal(size=m.nv)*0.3; ctrl = rng.normal(size=m.nu)*0.1
    assert np.allclose(np_mass_matrix(m, q), np_mass_matrix(h, q), rtol=1e-12, atol=1e-18)
    a = oracle.step(m, q[None], v[None], ctrl=ctrl[None], n_steps=5); bb = oracle.step(h, q[None], v[None], ctrl=ctrl[None], n_steps=5)
    assert np.allclose(a['qpos'], bb['qpos'], atol=1e-13) and np.allclose(a['sensordata'], bb['sensordata'], atol=1e-11)


def test_unit_scaling(sdf_path):
    """Lengths x meters, masses x kilograms, inertias x inertia, gains x torques (mjcf.py:169,567,582,819-854)."""
    u = SimulationUnitScaling(meters=2.0, seconds=0.5, kilograms=3.0)
    ao = _options(sdf_path)
    m1 = sdf2model(ModelSDF.read(sdf_path)[0], animat_options=ao)
    m2 = sdf2model(ModelSDF.read(sdf_path)[0], animat_options=ao, units=u)
    assert np.allclose(m2.body_pos, m1.body_pos*2.0) and np.allclose(m2.body_mass, m1.body_mass*3.0)
    assert np.allclose(m2.body_inertia, m1.body_inertia*u.inertia) and np.allclose(m2.body_ipos, m1.body_ipos*2.0)
    assert np.allclose(m2.jnt_pos, m1.jnt_pos*2.0)
    a = m1.actuator_names.index('actuator_position_j_trunk')
    assert np.isclose(m2.actuator_gain[a], m1.actuator_gain[a]*u.torques)
    assert np.isclose(m2.dof_damping[6], m1.dof_damping[6]*u.angular_damping)


def test_collision_subset(sdf_path):
    ao = _options(sdf_path)
    m = sdf2model(ModelSDF.read(sdf_path)[0], animat_options=ao, use_collisions=True, plane=True)
    assert m.ngeom == 3 and m.max_contacts == 32                     # capsule + sphere + plane
    g = [i for i in range(m.ngeom) if m.geom_bodyid[i] == m.body_id('head')][0]
    assert np.allclose(m.geom_size[g][:2], [0.02, 0.05])             # capsule: radius, HALF length


def test_box_collision_maps_to_half_extents(tmp_path, sdf_path):
    """SDF <box><size> holds full edge lengths; the compiled geom holds MuJoCo half extents (mjcf.py:486-503)."""
    from farms_mujoco_amd.model import GEOM_BOX
    text = open(sdf_path).read().replace(
        '<geometry><sphere><radius>0.02</radius></sphere></geometry>', '<geometry><box><size>0.1 0.04 0.02</size></box></geometry>')
    p = tmp_path/'box.sdf'
    p.write_text(text)
    ao = _options(str(p))
    m = sdf2model(ModelSDF.read(str(p))[0], animat_options=ao, use_collisions=True, plane=True)
    g = [i for i in range(m.ngeom) if m.geom_type[i] == GEOM_BOX]
    assert len(g) == 1 and np.allclose(m.geom_size[g[0]], [0.05, 0.02, 0.01])


def test_mjcf_export(sdf_path, tmp_path):
    """Simulation.save_mjcf_xml counterpart (reference simulation.py:215-225): the compiled model as an MJCF document
    with the reference's compiler / option settings, explicit inertials, the actuator triple and its sensors."""
    import xml.etree.ElementTree as ET
    from farms_mujoco_amd.simulation.mjcf import model2mjcf_xml
    ao = _options(sdf_path)
    m = sdf2model(ModelSDF.read(sdf_path)[0], animat_options=ao, use_collisions=True, plane=True)
    root = ET.fromstring(model2mjcf_xml(m))
    assert root.find('compiler').get('angle') == 'radian' and root.find('compiler').get('inertiafromgeom') == 'false'
    assert root.find('option').get('cone') == 'pyramidal' and root.find('option').get('solver') == 'PGS'
    bodies = root.findall('.//body')
    assert [b.get('name') for b in bodies] == list(m.body_names[1:])
    assert all(b.find('inertial') is not None for b in bodies)
    assert len(root.findall('.//joint')) + len(root.findall('.//freejoint')) == m.njnt
    assert len(root.findall('.//geom')) == m.ngeom
    assert len(root.findall('./actuator/general')) == m.nu
    names = [s.get('name') for s in root.find('sensor')]
    assert names == m.sensor_names()
    # nesting follows the kinematic tree
    for b in range(2, m.nbody):
        parent = next(p for p in root.iter('body') if any(c is e for c in p for e in [bodies[b - 1]]))
        assert parent.get('name') == m.body_names[m.body_parentid[b]]


def test_cylinder_collision_maps_to_radius_and_half_length(tmp_path, sdf_path):
    """SDF <cylinder> radius / length -> MuJoCo cylinder size (radius, half length), reference mjcf.py:427-440."""
    from farms_mujoco_amd.model import GEOM_CYLINDER
    text = open(sdf_path).read().replace(
        '<geometry><sphere><radius>0.02</radius></sphere></geometry>',
        '<geometry><cylinder><radius>0.02</radius><length>0.1</length></cylinder></geometry>')
    p = tmp_path/'cyl.sdf'
    p.write_text(text)
    m = sdf2model(ModelSDF.read(str(p))[0], animat_options=_options(str(p)), use_collisions=True, plane=True)
    g = [i for i in range(m.ngeom) if m.geom_type[i] == GEOM_CYLINDER]
    assert len(g) == 1 and np.allclose(m.geom_size[g[0]][:2], [0.02, 0.05])


ARENA_SDF = """<?xml version="1.0"?>
<sdf version="1.6">
  <model name="arena">
    <link name="terrain">
      <pose>0 0 0 0 0 0</pose>
      <collision name="terrain_col"><pose>0 0 0 0 0 0</pose>
        <geometry><heightmap><uri>terrain.png</uri><size>2.0 1.0 0.2</size></heightmap></geometry></collision>
    </link>
  </model>
</sdf>
"""


def test_arena_heightmap_compiles_to_heightfield(sdf_path, tmp_path):
    """Arena SDF with a heightmap (reference mjcf.py:486-522, :1195-1212; task.py:108-115): the image is normalised,
    flipped to Cartesian rows and stored as 2 (image - 0.5); the asset size is (x/2, y/2, z/2, z) scaled by units.meters;
    ground_height lifts the arena; the animat gets its collision geoms."""
    from farms_mujoco_amd.io.png import imwrite_gray
    from farms_mujoco_amd.model import GEOM_HFIELD
    rng = np.random.default_rng(1)
    img = rng.integers(0, 65535, (6, 9)).astype(np.uint16)
    imwrite_gray(str(tmp_path/'terrain.png'), img)
    (tmp_path/'arena.sdf').write_text(ARENA_SDF)
    units = SimulationUnitScaling(meters=2.0, seconds=1.0, kilograms=1.0)
    opts = SimulationOptions(timestep=1e-3, units=units)
    arena = ArenaOptions(sdf=str(tmp_path/'arena.sdf'), ground_height=0.05, spawn_pose=[0.1, 0, 0, 0, 0, 0])
    m = setup_model(opts, _options(sdf_path), arena)
    g = int(np.nonzero(m.geom_type == GEOM_HFIELD)[0][0])
    assert m.geom_bodyid[g] == 0 and (m.hfield_nrow, m.hfield_ncol) == (6, 9)
    assert np.allclose(m.hfield_size, [2.0, 1.0, 0.2, 0.4])                     # (x/2, y/2, z/2, z) * meters
    want = 2*(np.flip(img.astype(float)/65535, axis=0) - 0.5)
    assert np.allclose(m.hfield_data, want) and m.hfield_data.min() < -0.9 and m.hfield_data.max() > 0.9
    assert np.allclose(m.geom_pos[g], [0.2, 0, 0.1])                           # (spawn + ground_height) * meters
    assert m.max_contacts >= 32 and (m.geom_type != GEOM_HFIELD).sum() == 2   # head capsule + trunk sphere collide with it
    c = m.as_c()
    assert c.hfield_nrow == 6 and c.hfield_ncol == 9 and abs(c.hfield_data[3] - want.ravel()[3]) < 1e-15


def test_self_collisions_become_contact_pairs(sdf_path):
    """morphology.self_collisions (reference mjcf.py:1005-1033): every pair of collision shapes of the two links becomes an
    explicit pair, friction 0, the global solref if one is given."""
    ao = _options(sdf_path)
    ao.morphology.self_collisions = [['head', 'trunk']]
    ao.mujoco = dict(solref=[0.01, 1.0])
    m = setup_model(SimulationOptions(timestep=1e-3), ao, ArenaOptions(ground_height=0.0))
    assert m.npair == 1
    g1, g2 = int(m.pair_geom1[0]), int(m.pair_geom2[0])
    assert m.body_names[m.geom_bodyid[g1]] == 'head' and m.body_names[m.geom_bodyid[g2]] == 'trunk'
    assert m.pair_friction[0] == 0.0 and np.allclose(m.pair_solref[0], [0.01, 1.0])
    c = m.as_c()
    assert c.npair == 1 and c.pair_geom1[0] == g1 and c.pair_geom2[0] == g2
    # without an arena the pairs alone switch the collision geoms on
    m2 = setup_model(SimulationOptions(timestep=1e-3), ao, ArenaOptions())
    assert m2.npair == 1 and m2.ngeom == 2 and m2.max_contacts >= 32


def test_mesh_collision_from_obj_and_stl(tmp_path):
    """SDF mesh collisions (reference mjcf.py:270-413): the vertices of an .obj / binary .stl / ascii .stl file, scaled by
    the element's <scale>, become a convex mesh geom (hull vertices only); the MJCF export carries them as a mesh asset."""
    import struct
    from farms_mujoco_amd.io.mesh import read_vertices
    from farms_mujoco_amd.io.sdf import ModelSDF
    from farms_mujoco_amd.simulation.mjcf import sdf2model, model2mjcf_xml
    tet = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
    faces = [(0, 2, 1), (0, 1, 3), (0, 3, 2), (1, 2, 3)]
    (tmp_path / 'tet.obj').write_text('# tetrahedron\n' + ''.join(f'v {a} {b} {c}\n' for a, b, c in tet) + 'vn 0 0 1\n'
                                      + ''.join(f'f {i + 1} {j + 1} {k + 1}\n' for i, j, k in faces))
    with open(tmp_path / 'tet.stl', 'wb') as f:
        f.write(b'binary stl'.ljust(80, b' ') + struct.pack('<I', len(faces)))
        for i, j, k in faces:
            f.write(struct.pack('<12fH', 0, 0, 0, *tet[i], *tet[j], *tet[k], 0))
    (tmp_path / 'tet_ascii.stl').write_text('solid t\n' + ''.join(
        'facet normal 0 0 0\n outer loop\n' + ''.join(f'  vertex {tet[v][0]} {tet[v][1]} {tet[v][2]}\n' for v in fc) + ' endloop\nendfacet\n'
        for fc in faces) + 'endsolid t\n')
    for name in ('tet.obj', 'tet.stl', 'tet_ascii.stl'):
        v = read_vertices(str(tmp_path / name), scale=(0.1, 0.2, 0.3))
        assert sorted(map(tuple, np.round(v, 12))) == sorted(map(tuple, tet*[0.1, 0.2, 0.3])), name
    (tmp_path / 'm.sdf').write_text('''<sdf version="1.6"><model name="m">
      <link name="base"><pose>0 0 0.2 0 0 0</pose>
        <inertial><mass>0.3</mass><inertia><ixx>1e-4</ixx><iyy>1e-4</iyy><izz>1e-4</izz></inertia></inertial>
        <collision name="c"><pose>0.01 0 0 0 0 0</pose><geometry><mesh><uri>tet.obj</uri><scale>0.1 0.1 0.1</scale></mesh></geometry></collision>
      </link></model></sdf>''')
    m = sdf2model(ModelSDF.read(str(tmp_path / 'm.sdf'))[0], plane=True, use_collisions=True)
    g = int(np.nonzero(m.geom_type == 7)[0][0])
    assert m.nmeshvert == 4 and m.geom_vertnum[g] == 4 and m.geom_vertadr[g] == 0
    assert sorted(map(tuple, np.round(m.mesh_vert, 12))) == sorted(map(tuple, 0.1*tet))
    assert np.allclose(m.geom_pos[g], [0.01, 0, 0]) and abs(m.geom_size[g][2] - 0.1) < 1e-12      # bounding radius about the geom origin
    xml = model2mjcf_xml(m)
    assert '<mesh name="mesh_%d"' % g in xml and 'type="mesh"' in xml


def test_arena_pose_is_the_spawn_pose_composed_with_the_collision_pose(sdf_path, tmp_path):
    """The reference overwrites the arena base link's pose with arena_options.spawn.pose (+ ground_height) and keeps the
    collision's own pose under that body (mjcf.py:1207-1211): a rotated, offset arena puts the heightfield at
    spawn o collision - rotations composed, the SDF link pose of the base link ignored."""
    from farms_mujoco_amd.io.png import imwrite_gray
    from farms_mujoco_amd.model import GEOM_HFIELD, quat_mul, quat2mat, euler2quat
    imwrite_gray(str(tmp_path/'terrain.png'), np.zeros((4, 4), np.uint16))
    (tmp_path/'arena.sdf').write_text(ARENA_SDF.replace('<pose>0 0 0 0 0 0</pose>\n      <collision', '<pose>5 5 5 1 1 1</pose>\n      <collision')
                                      .replace('<collision name="terrain_col"><pose>0 0 0 0 0 0</pose>', '<collision name="terrain_col"><pose>0.3 0 0.1 0 0 0.4</pose>'))
    spawn = [0.1, -0.2, 0.0, 0.0, 0.3, 1.2]
    arena = ArenaOptions(sdf=str(tmp_path/'arena.sdf'), ground_height=0.05, spawn_pose=spawn)
    m = setup_model(SimulationOptions(timestep=1e-3), _options(sdf_path), arena)
    g = int(np.nonzero(m.geom_type == GEOM_HFIELD)[0][0])
    bq = euler2quat(spawn[3:])
    assert np.allclose(m.geom_pos[g], np.array([0.1, -0.2, 0.05]) + quat2mat(bq) @ np.array([0.3, 0, 0.1]))
    want_q = quat_mul(bq, euler2quat([0, 0, 0.4]))
    assert min(np.abs(m.geom_quat[g] - want_q).max(), np.abs(m.geom_quat[g] + want_q).max()) < 1e-12


def test_two_collisions_with_one_name_sphere_then_mesh(tmp_path):
    """A link whose collisions share a name (the SDF reader's default `<link>_collision`) with the mesh not first: the mesh is
    found by its position in the list, not by comparing Collision objects (which hold arrays)."""
    from farms_mujoco_amd.io.sdf import ModelSDF
    from farms_mujoco_amd.simulation.mjcf import sdf2model
    (tmp_path / 'tet.obj').write_text('v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 3 2\nf 1 2 4\nf 1 4 3\nf 2 3 4\n')
    (tmp_path / 'm.sdf').write_text('''<sdf version="1.6"><model name="m">
      <link name="base"><pose>0 0 0.2 0 0 0</pose>
        <inertial><mass>0.3</mass><inertia><ixx>1e-4</ixx><iyy>1e-4</iyy><izz>1e-4</izz></inertia></inertial>
        <collision name="same"><pose>0 0 0 0 0 0</pose><geometry><sphere><radius>0.02</radius></sphere></geometry></collision>
        <collision name="same"><pose>0.01 0 0 0 0 0</pose><geometry><mesh><uri>tet.obj</uri><scale>0.1 0.1 0.1</scale></mesh></geometry></collision>
      </link></model></sdf>''')
    m = sdf2model(ModelSDF.read(str(tmp_path / 'm.sdf'))[0], plane=True, use_collisions=True)
    assert sorted(m.geom_type.tolist()) == [0, 2, 7] and m.nmeshvert == 4


def test_solver_and_cone_options_reach_the_model_and_the_mjcf(sdf_path):
    """simulation_options.solver / cone / impratio are forwarded like reference mjcf.py:1342-1353 does (round 3: Newton, CG and the
    elliptic cone exist on the device); the exported MJCF names them."""
    import xml.etree.ElementTree as ET
    from farms_mujoco_amd.model import SOLVERS, CONES
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.mjcf import model2mjcf_xml
    ao = _options(sdf_path)
    so = SimulationOptions(solver='Newton', cone='elliptic', n_solver_iters=100, impratio=4.0)
    m = sdf2model(ModelSDF.read(sdf_path)[0], animat_options=ao, simulation_options=so, use_collisions=True, plane=True)
    assert m.solver == SOLVERS['newton'] and m.cone == CONES['elliptic'] and m.solver_iterations == 100 and m.impratio == 4.0
    c = m.as_c()
    assert c.solver == 2 and c.cone == 1
    opt = ET.fromstring(model2mjcf_xml(m)).find('option')
    assert opt.get('solver') == 'Newton' and opt.get('cone') == 'elliptic' and float(opt.get('impratio')) == 4.0
    m = sdf2model(ModelSDF.read(sdf_path)[0], animat_options=ao, simulation_options=SimulationOptions(solver='CG'), use_collisions=True, plane=True)
    assert m.solver == SOLVERS['cg'] and m.cone == CONES['pyramidal']


def test_noslip_options_reach_the_model_and_the_mjcf(sdf_path):
    """simulation_options.noslip_iterations / noslip_tolerance are forwarded like reference mjcf.py:1392-1403 does, so that a run
    configured with the noslip post-pass reaches fmj_create's handling of it instead of silently stepping without it."""
    import xml.etree.ElementTree as ET
    from farms_mujoco_amd.options import SimulationOptions
    from farms_mujoco_amd.simulation.mjcf import model2mjcf_xml
    ao = _options(sdf_path)
    so = SimulationOptions(noslip_iterations=7, noslip_tolerance=1e-5)
    m = sdf2model(ModelSDF.read(sdf_path)[0], animat_options=ao, simulation_options=so, use_collisions=True, plane=True)
    assert m.noslip_iterations == 7 and m.noslip_tolerance == 1e-5
    c = m.as_c()
    assert c.noslip_iterations == 7 and c.noslip_tolerance == 1e-5
    opt = ET.fromstring(model2mjcf_xml(m)).find('option')
    assert int(opt.get('noslip_iterations')) == 7 and float(opt.get('noslip_tolerance')) == 1e-5
    m0 = sdf2model(ModelSDF.read(sdf_path)[0], animat_options=ao, simulation_options=SimulationOptions(), use_collisions=True, plane=True)
    assert m0.noslip_iterations == 0


def test_self_collisions_between_mesh_and_box_links(tmp_path):
    """Round 5: the reference emits a pair for every collision shape of every morphology.self_collisions link
    pair (mjcf.py:1012-1033) and its usual collision shape is a convex mesh (mjcf.py:270-413).  An SDF animat with a mesh link, a box
    link and self_collisions compiles: the pair joins the mesh geom and the box geom, the mesh carries the planes of its hull, the
    C model hands them on, and the oracle finds the contact when the two links overlap."""
    from farms_mujoco_amd.io.sdf import ModelSDF
    from farms_mujoco_amd.options import AnimatOptions, ArenaOptions, SimulationOptions
    from farms_mujoco_amd.simulation.mjcf import setup_model
    tet = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
    (tmp_path / 'tet.obj').write_text(''.join(f'v {a} {b} {c}\n' for a, b, c in tet))
    (tmp_path / 'm.sdf').write_text('''<sdf version="1.6"><model name="m">
      <link name="base"><pose>0 0 0.2 0 0 0</pose>
        <inertial><mass>0.3</mass><inertia><ixx>1e-4</ixx><iyy>1e-4</iyy><izz>1e-4</izz></inertia></inertial>
        <collision name="c"><geometry><mesh><uri>tet.obj</uri><scale>0.1 0.1 0.1</scale></mesh></geometry></collision>
      </link>
      <link name="arm"><pose>0.02 0.02 0.2 0 0 0</pose>
        <inertial><mass>0.1</mass><inertia><ixx>1e-5</ixx><iyy>1e-5</iyy><izz>1e-5</izz></inertia></inertial>
        <collision name="c"><geometry><box><size>0.02 0.02 0.02</size></box></geometry></collision>
      </link>
      <joint name="j" type="revolute"><parent>base</parent><child>arm</child><axis><xyz>0 0 1</xyz><limit><lower>-1</lower><upper>1</upper></limit></axis></joint>
    </model></sdf>''')
    ao = AnimatOptions(name='m', links=[AnimatOptions.link(n) for n in ('base', 'arm')], joints=[AnimatOptions.joint('j', damping=1e-3)],
                       motors=[], sdf=str(tmp_path / 'm.sdf'))
    ao.morphology.self_collisions = [['base', 'arm']]
    m = setup_model(SimulationOptions(timestep=1e-3), ao, ArenaOptions())
    assert m.npair == 1
    g1, g2 = int(m.pair_geom1[0]), int(m.pair_geom2[0])
    assert {int(m.geom_type[g1]), int(m.geom_type[g2])} == {7, 6}                    # a mesh and a box
    gm = g1 if m.geom_type[g1] == 7 else g2
    assert m.nmeshface == 4 and m.geom_facenum[gm] == 4 and m.geom_faceadr[gm] == 0
    f = np.asarray(m.mesh_face)
    assert np.allclose(np.linalg.norm(f[:, :3], axis=1), 1.0)
    v = np.asarray(m.mesh_vert)
    assert (f[:, :3] @ v.T - f[:, 3:4] < 1e-12).all()                                  # every vertex inside every plane
    c = m.as_c()
    assert c.nmeshface == 4 and c.mesh_face[3] == f[0, 3]
    from oracle import oracle as orc
    orc.build()
    o = orc.forward_debug(m, m.qpos0, np.zeros(m.nv))
    assert o['ncon'] >= 1 and (o['contact'][:o['ncon'], 17] < 0).all()               # the box's corner sits inside the tetrahedron
