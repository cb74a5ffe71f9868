"""Explicit pairs beyond sphere / capsule (reference mjcf.py:1012-1033 emits a pair for EVERY collision shape
of every morphology.self_collisions link pair, and its usual collision shape is a convex mesh, mjcf.py:270-413): box, cylinder and
convex-mesh geoms in pairs - the polytope narrow phase of include/fmj.h (ABI 6), HIP against the oracle's collide_pair on identical
fp32 inputs: the contact lists (count, geoms, order, positions, normals, distances), the forces of one step, and a settling run."""
import numpy as np
import pytest

from parity_metrics import group_relerr, qvel_groups
from test_oracle_contacts import _stack

pytestmark = pytest.mark.gpu


def _cube(h):
    return np.array([[(i & 1)*2 - 1, ((i >> 1) & 1)*2 - 1, ((i >> 2) & 1)*2 - 1] for i in range(8)], float)*h


def _ico(r):
    t = (1 + 5**0.5)/2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], float)
    return v*(r/np.linalg.norm(v[0]))


def _cases():
    from farms_mujoco_amd.model import GEOM_BOX, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_MESH
    box = (GEOM_BOX, (0.1, 0.1, 0.1))
    return {
        'box_on_box': (box, (GEOM_BOX, (0.05, 0.05, 0.05)), 0.148, 0.03),
        'sphere_on_box': (box, (GEOM_SPHERE, (0.03, 0, 0)), 0.128, 0.0),
        'capsule_on_box': (box, (GEOM_CAPSULE, (0.02, 0.06, 0)), 0.119, np.pi/2 - 0.02),
        'cylinder_on_box': (box, (GEOM_CYLINDER, (0.04, 0.05, 0)), 0.149, 0.02),
        'box_on_cylinder': ((GEOM_CYLINDER, (0.1, 0.1, 0)), (GEOM_BOX, (0.04, 0.04, 0.04)), 0.139, 0.03),
        'mesh_cube_on_box': (box, (GEOM_MESH, None, _cube(0.05)), 0.148, 0.03),
        'icosahedron_on_mesh_cube': ((GEOM_MESH, None, _cube(0.1)), (GEOM_MESH, None, _ico(0.05)), 0.1 + 0.0425 - 0.002, 0.1),
        'sphere_on_icosahedron': ((GEOM_MESH, None, _ico(0.1)), (GEOM_SPHERE, (0.03, 0, 0)), 0.0795 + 0.03 - 0.002, 0.0),
    }


@pytest.mark.parametrize('case', list(_cases()))
def test_polytope_pair_contacts_and_forces(oracle, case):
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    lower, upper, z, tilt = _cases()[case]
    m, q = _stack(lower, upper, z, tilt=tilt, gravity=(0, 0, -9.81), friction=0.5)
    n = 4
    qpos = np.tile(q, (n, 1)) + np.linspace(0, -1.5e-3, n)[:, None]          # four depths
    phys = BatchedPhysics(m, n)
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = -0.05
    q32 = d.qpos.cpu().numpy().astype(np.float64); v32 = d.qvel.cpu().numpy().astype(np.float64)
    rows, _ = phys.step_debug(want_pgs=False)
    torch.cuda.synchronize()
    o = oracle.step_tf(m, q32, v32, want_AR=False)
    assert int(d.status.abs().sum()) == 0
    assert np.array_equal(d.ncon.cpu().numpy(), o['ncon']) and o['ncon'].max() >= 1, (d.ncon.cpu().numpy(), o['ncon'])
    con = oracle.contacts_from_hip(d.contact.cpu().numpy())
    worst_f = 0.0
    for e in range(n):
        nc = int(o['ncon'][e])
        assert np.array_equal(con[e, :nc, 15:17], o['contact'][e, :nc, 15:17])                          # geom1, geom2
        assert np.abs(con[e, :nc, :3] - o['contact'][e, :nc, :3]).max() < 2e-6, case                    # positions
        assert np.abs(con[e, :nc, 3:12] - o['contact'][e, :nc, 3:12]).max() < 2e-5, case                # frames
        fs = max(np.abs(o['contact'][e, :nc, 12]).max(), 1e-2)
        worst_f = max(worst_f, np.abs(con[e, :nc, 12:15] - o['contact'][e, :nc, 12:15]).max()/fs)
    rows = rows.cpu().numpy()
    for e in range(n):
        ne = int(o['nefc'][e])
        assert np.abs(rows[e, :ne, 0] - o['efc'][e, :ne, 3]*0 - np.repeat(o['contact'][e, :int(o['ncon'][e]), 17], 4)).max() < 2e-6      # row pos = contact dist
    err = np.abs(d.qvel.cpu().numpy() - o['qvel']).max()/np.abs(o['qvel']).max()
    print(case, 'contacts per env', o['ncon'], 'contact-frame forces', worst_f, 'qvel', err)
    assert worst_f < 2e-3 and err < 1e-3
    phys.step(299)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, n_steps=300)
    assert int(d.status.abs().sum()) == 0
    assert np.abs(d.qpos.cpu().numpy() - ref['qpos']).max() < 2e-5
    assert int(d.ncon.min()) >= 1 and np.abs(ref['qvel']).max() < 5e-3                                 # at rest on the lower shape


def test_salamander_with_mesh_feet_and_self_collision_pairs(oracle):
    """The reference's usual case: convex-mesh collision shapes AND morphology.self_collisions.  Mesh feet (12-vertex hulls) in pairs
    with the trunk capsules and with each other; limbs folded under the body so that feet meet trunk and feet meet feet."""
    import torch
    from farms_mujoco_amd.model import salamander33
    from farms_mujoco_amd.physics import BatchedPhysics
    m = salamander33(contacts=True, limits=True, spawn_z=0.045, self_collisions=True, mesh_feet=True)
    assert m.npair > 0 and m.nmeshface >= 20 and (np.asarray(m.geom_type)[np.asarray(m.pair_geom1)] == 7).any()
    legj = [m.jnt_qposadr[m.joint_names.index(f'joint_leg_{t}_{s_}_{k}')] for t in ('front', 'hind') for s_ in ('L', 'R') for k in range(4)]
    rng = np.random.default_rng(7)
    n = 24
    qpos = np.tile(m.qpos0, (n, 1)); qpos[:, 2] = 0.05
    qpos[:, legj] = rng.uniform(-1.2, 1.2, (n, len(legj)))
    phys = BatchedPhysics(m, n)
    d = phys.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(0.02*rng.normal(size=(n, m.nv)), dtype=torch.float32)
    q32 = d.qpos.cpu().numpy().astype(np.float64); v32 = d.qvel.cpu().numpy().astype(np.float64)
    phys.step_debug(want_pgs=False)
    torch.cuda.synchronize()
    o = oracle.step_tf(m, q32, v32, ctrl=np.zeros((n, m.nu)), want_AR=False)
    plane = int(np.nonzero(np.asarray(m.geom_type) == 0)[0][0])
    npair = np.array([int((o['contact'][e, :o['ncon'][e], 15] != plane).sum()) for e in range(n)])
    print('pair contacts per env', npair, 'contacts', o['ncon'])
    assert (npair > 0).sum() >= 3, 'the draw has too few self-contacts'
    assert int(d.status.abs().sum()) == 0 and np.array_equal(d.ncon.cpu().numpy(), o['ncon'])
    con = oracle.contacts_from_hip(d.contact.cpu().numpy())
    worst = 0.0
    for e in range(n):
        nc = int(o['ncon'][e])
        if nc == 0:
            continue
        assert np.array_equal(con[e, :nc, 15:17], o['contact'][e, :nc, 15:17])
        assert np.abs(con[e, :nc, :3] - o['contact'][e, :nc, :3]).max() < 2e-6
        fs = max(np.abs(o['contact'][e, :nc, 12]).max(), 1e-2)
        worst = max(worst, np.abs(con[e, :nc, 12:15] - o['contact'][e, :nc, 12:15]).max()/fs)
    err = group_relerr(d.qvel.cpu().numpy(), o['qvel'], qvel_groups(m))
    print('mesh feet + self-collision pairs: contact-frame forces', worst, 'qvel per component', err)
    assert worst < 5e-3 and err < 3e-2


def test_from_sdf_mesh_collisions_and_self_collisions_end_to_end(oracle, tmp_path):
    """VERDICT round 4 item 4, through the host API: an SDF animat whose links carry MESH collisions (the reference's usual case,
    mjcf.py:270-413) plus a box, `morphology.self_collisions` between them (mjcf.py:1012-1033), on a flat arena: Simulation.from_sdf
    compiles it (hull planes included), the animal folds onto itself under a wave controller and lies on the floor; 300 fused iterations
    against the oracle stepping the same compiled model."""
    import torch
    from farms_mujoco_amd.options import AnimatOptions, ArenaOptions, SimulationOptions, WaterOptions
    from farms_mujoco_amd.simulation.simulation import Simulation
    from test_gpu_fused_parity import _SdfWave, _oracle_initial_state
    oct_ = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float)
    (tmp_path / 'oct.obj').write_text(''.join(f'v {a} {b} {c}\n' for a, b, c in oct_))
    link = lambda name, x, geom: f'''<link name="{name}"><pose>{x} 0 0.06 0 0 0</pose>
        <inertial><mass>0.05</mass><inertia><ixx>2e-5</ixx><iyy>2e-5</iyy><izz>2e-5</izz></inertia></inertial>
        <collision name="c_{name}"><geometry>{geom}</geometry></collision></link>'''
    mesh = '<mesh><uri>oct.obj</uri><scale>0.03 0.02 0.04</scale></mesh>'
    (tmp_path / 'm.sdf').write_text('<sdf version="1.6"><model name="folder">'
        + link('a', 0.0, mesh) + link('b', 0.07, '<box><size>0.04 0.03 0.03</size></box>') + link('c', 0.14, mesh)
        + ''.join(f'<joint name="j_{ch}" type="revolute"><parent>{pa}</parent><child>{ch}</child><pose>-0.035 0 0 0 0 0</pose>'
                  f'<axis><xyz>0 1 0</xyz><limit><lower>-2.6</lower><upper>2.6</upper></limit></axis></joint>' for pa, ch in (('a', 'b'), ('b', 'c')))
        + '</model></sdf>')
    ao = AnimatOptions(name='folder', links=[AnimatOptions.link(n, friction=[0.7, 0, 0]) for n in 'abc'],
                       joints=[AnimatOptions.joint(f'j_{n}', damping=2e-3) for n in 'bc'], motors=[AnimatOptions.motor(f'j_{n}', gains=(0.5, 0.01)) for n in 'bc'],
                       sdf=str(tmp_path / 'm.sdf'), spawn_pose=(0, 0, 0.0, 0, 0, 0))
    ao.morphology.self_collisions = [['a', 'c'], ['a', 'b']]
    n, T = 4, 500
    opts = SimulationOptions(timestep=1e-3, n_iterations=T)
    arena = ArenaOptions(water=WaterOptions(height=None, drag=False), ground_height=0.0)
    from farms_mujoco_amd.simulation.mjcf import setup_model
    m = setup_model(opts, ao, arena)
    assert m.npair == 2 and m.nmeshface == 16 and sorted(set(np.asarray(m.geom_type)[np.asarray(m.pair_geom2)])) == [6, 7]
    from farms_mujoco_amd.data import AnimatData
    ctl = _SdfWave(m, np.zeros(n))
    ctl.amplitude = ctl.amplitude*8.0                       # +-2 rad: the chain folds until link c meets link a (after ~290 steps)
    pairs = [('a', 'c'), ('a', 'b'), ('c', '')]
    data = AnimatData(1e-3, T, n, m.body_names[1:], m.hinge_joint_names(), contacts=pairs)
    sim = Simulation.from_sdf(opts, ao, arena, n_envs=n, buffer_size=T, controller=ctl, data=data)
    sim.reset()
    m = sim.physics.model
    st = _oracle_initial_state(oracle, sim, m)
    sim.run(fused=True)
    torch.cuda.synchronize()
    g2d = sim.task.maps['sensors']['geompair2data']
    ref = oracle.run_fused(m, st, T, buffer_size=T, controller=1, geompair2data=g2d, n_contact_rows=len(pairs),
                           wave=dict(amplitude=ctl.amplitude.cpu().numpy(), phase_lag=ctl.phase_lag.cpu().numpy(),
                                     env_phase=ctl.env_phase.cpu().numpy(), frequency=ctl.frequency))
    d = sim.physics.data
    assert int(d.status.abs().sum()) == 0
    rows = data.sensors.contacts.array.cpu().numpy(); want = ref['contacts']
    self_force = np.abs(want[:, :, 0, 6:9]).max()           # total force of the (a, c) sensor: mesh against mesh
    assert self_force > 1e-3, 'link c never met link a'
    e = np.abs(d.qpos.cpu().numpy() - ref['qpos']).max(1)
    ec = np.abs(rows[..., :9] - want[..., :9]).max()/np.abs(want[..., :9]).max()
    print('from_sdf with mesh collisions + self_collisions: qpos abs err per env', e, 'contact rows', ec, 'peak self-contact force', self_force)
    assert e.max() < 5e-4 and ec < 2e-2
