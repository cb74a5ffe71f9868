"""Generality of the HIP path: seeded random kinematic trees (branching, welded bodies, hinge + slide joints with
off-centre anchors, rotated inertial frames, free or fixed base, spring-dampers, armature, all three actuator kinds
with control / force limits, external forces) against the fp64 oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FMJ_WARN_CONTACTFULL = 8      # include/fmj.h (FMJ_WARN_BADQACC is 4: a frozen env must fail these tests)


def random_tree(seed, contacts=False, meshes=False):
    from farms_mujoco_amd.model import ModelBuilder, euler2quat, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_PLANE
    rng = np.random.default_rng(seed)
    nb = int(rng.integers(3, 22))
    free = bool(rng.integers(0, 2))
    b = ModelBuilder(f'tree{seed}', timestep=1e-3, gravity=(0, 0, -9.81) if rng.integers(0, 2) else (0.5, -0.3, -9.0))
    names = []

    def inertia():
        A = rng.normal(size=(3, 3)); S = A @ A.T*1e-4 + np.eye(3)*2e-4
        return (S[0, 0], S[1, 1], S[2, 2], S[0, 1], S[0, 2], S[1, 2])
    for i in range(nb):
        name = f'b{i}'
        mass = float(rng.uniform(0.05, 0.5))
        kw = dict(mass=mass, ipos=rng.normal(size=3)*0.03, fullinertia=inertia())
        if i == 0:
            if free:
                b.add_body(name, 'world', pos=rng.normal(size=3)*0.2, quat=euler2quat(rng.normal(size=3)), joint='free', **kw)
            else:
                jt = ['hinge', 'slide', None][int(rng.integers(0, 3))]
                jkw = dict(joint=jt, axis=rng.normal(size=3), jpos=rng.normal(size=3)*0.02, damping=0.01) if jt else {}
                b.add_body(name, 'world', pos=rng.normal(size=3)*0.2, quat=euler2quat(rng.normal(size=3)), **jkw, **kw)
        else:
            parent = names[int(rng.integers(max(0, i - 4), i))]
            u = rng.random()
            jt = 'hinge' if u < 0.7 else 'slide' if u < 0.85 else None
            jkw = {}
            if jt:
                jkw = dict(joint=jt, axis=rng.normal(size=3), jpos=rng.normal(size=3)*0.03 if rng.random() < 0.5 else (0, 0, 0),
                           damping=float(rng.choice([0.0, 2e-3, 1e-2])), stiffness=float(rng.choice([0.0, 0.0, 0.05])),
                           armature=float(rng.choice([0.0, 1e-4])), qpos0=float(rng.choice([0.0, 0.2])))
                if contacts and jt == 'hinge' and rng.random() < 0.5:
                    jkw.update(limited=True, range=(-0.3, 0.25))
            b.add_body(name, parent, pos=rng.normal(size=3)*0.08, quat=euler2quat(rng.normal(size=3)*0.5), **jkw, **kw)
        names.append(name)
        if contacts and rng.random() < 0.7:
            kind = int(rng.integers(0, 5 if meshes else 4))
            gk = dict(pos=rng.normal(size=3)*0.02, quat=euler2quat(rng.normal(size=3)), friction=(float(rng.uniform(0.3, 1.0)), 0, 0))
            if kind == 0:
                b.add_geom(name, GEOM_SPHERE, (float(rng.uniform(0.02, 0.05)),), **gk)
            elif kind == 1:
                b.add_geom(name, GEOM_CAPSULE, (float(rng.uniform(0.015, 0.03)), float(rng.uniform(0.02, 0.06))), **gk)
            elif kind == 2:
                b.add_geom(name, GEOM_BOX, tuple(rng.uniform(0.015, 0.05, 3)), **gk)
            elif kind == 3:
                b.add_geom(name, GEOM_CYLINDER, (float(rng.uniform(0.02, 0.05)), float(rng.uniform(0.01, 0.05))), **gk)
            else:                       # convex mesh: a random point cloud (its hull), off-centre in the geom frame
                cloud = rng.normal(size=(int(rng.integers(5, 40)), 3))*rng.uniform(0.01, 0.04, 3) + rng.normal(size=3)*0.01
                b.add_mesh_geom(name, cloud, **gk)
    if contacts:
        b.add_geom('world', GEOM_PLANE, (0, 0, 0), pos=(0, 0, -0.05), friction=(0.2, 0, 0))
        b.options['max_contacts'] = 32
    joints = [bd.joint['name'] for bd in b.bodies[1:] if bd.joint and bd.joint['type'] != 0]
    for jn in joints:
        r = rng.random()
        if r < 0.5:
            b.add_joint_actuators(jn, kp=float(rng.uniform(0.1, 0.5)), kv=float(rng.uniform(0, 0.01)),
                                  forcerange=(-0.2, 0.3) if rng.random() < 0.5 else None)
        elif r < 0.75:
            b.add_position_actuator(jn, kp=0.3)
    if not joints and not free:
        return None
    return b.compile()


@pytest.mark.parametrize('seed,two_per_wave', [(s, True) for s in range(20)] + [(s, False) for s in range(0, 20, 2)])
def test_random_tree_vs_oracle(oracle, seed, two_per_wave, monkeypatch):
    """Every tree here has <= 32 bodies, so it runs in the two-envs-per-wave kernel by default; FMJ_DUAL=0 (read at
    fmj_create) sends the same tree through the one-env-per-wave kernel."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    if not two_per_wave:
        monkeypatch.setenv('FMJ_DUAL', '0')
    m = random_tree(seed)
    if m is None or m.nv == 0:
        pytest.skip('degenerate draw')
    rng = np.random.default_rng(1000 + seed)
    n = 6
    qpos = np.tile(m.qpos0, (n, 1)) + rng.uniform(-0.4, 0.4, (n, m.nq))
    for j in range(m.njnt):
        if m.jnt_type[j] == 0:
            a = m.jnt_qposadr[j]; q = rng.normal(size=(n, 4)); qpos[:, a+3:a+7] = q/np.linalg.norm(q, axis=1, keepdims=True)
    qvel = rng.normal(size=(n, m.nv))*0.5
    ctrl = rng.uniform(-0.6, 0.6, (n, max(m.nu, 1)))[:, :m.nu]
    xf = rng.normal(size=(n, m.nbody, 6))*0.05; xf[:, 0] = 0
    qs = np.tile(m.qpos_spring, (n, 1)) + rng.uniform(-0.1, 0.1, (n, m.nq))
    phys = BatchedPhysics(m, n)
    assert phys.kernel_info()['threads_per_env'] == (32 if two_per_wave else 64)
    d = phys.data
    f32 = lambda a: torch.as_tensor(a, dtype=torch.float32)
    d.qpos[:] = f32(qpos); d.qvel[:] = f32(qvel); d.xfrc_applied[:] = f32(xf); d.qpos_spring[:] = f32(qs)
    if m.nu:
        d.ctrl[:] = f32(ctrl)
    r64 = lambda t: t.cpu().numpy().astype(np.float64)
    q32, v32, c32, x32, s32 = r64(d.qpos), r64(d.qvel), r64(d.ctrl), r64(d.xfrc_applied), r64(d.qpos_spring)
    phys.step(1)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=c32 if m.nu else None, qpos_spring=s32, xfrc_applied=x32)
    assert int(d.status.abs().sum()) == 0

    def err(k):
        a = r64(getattr(d, k)); bb = ref[k]
        return np.abs(a - bb).max()/max(np.abs(bb).max(), 1e-9)
    for k, tol in (('xpos', 5e-6), ('xquat', 5e-6), ('xipos', 5e-6), ('sensordata', 1e-4), ('qpos', 5e-6)):
        assert err(k) < tol, (seed, m.nbody, m.nv, k, err(k))
    # the velocity comes out of the (M + hB) solve: bounded by a small multiple of what fp32 storage of that matrix alone costs
    # on this tree (oracle.fp32_storage), plus the fp32 rounding of a well-conditioned solve
    with oracle.fp32_storage():
        floor = oracle.step(m, q32, v32, ctrl=c32 if m.nu else None, qpos_spring=s32, xfrc_applied=x32)
    fl = np.abs(floor['qvel'] - ref['qvel']).max()/max(np.abs(ref['qvel']).max(), 1e-9)
    assert err('qvel') < 6*fl + 2e-6, (seed, m.nbody, m.nv, 'qvel', err('qvel'), fl)
    phys.step(49)
    torch.cuda.synchronize()
    ref = oracle.step(m, q32, v32, ctrl=c32 if m.nu else None, qpos_spring=s32, xfrc_applied=x32, n_steps=50)
    assert err('qpos') < 2e-3, (seed, 'qpos50', err('qpos'))


@pytest.mark.parametrize('seed,two_per_wave', [(s, True) for s in range(100, 112)] + [(s, False) for s in range(100, 112, 2)])
def test_random_tree_with_limits_and_contacts(oracle, seed, two_per_wave, monkeypatch):
    """The constraint path on random trees: limited hinges, sphere / capsule / box / cylinder geoms over a plane that cuts
    through the tree; contact lists, constraint forces and the state after one and after 30 steps vs the oracle.  Every tree here has
    <= 32 bodies: it runs in the two-env constraint kernel (fmj_cons2.inc) by default, FMJ_DUAL=0 sends it through the one-env kernel."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    if not two_per_wave:
        monkeypatch.setenv('FMJ_DUAL', '0')
    m = random_tree(seed, contacts=True)
    if m is None or m.nv == 0:
        pytest.skip('degenerate draw')
    rng = np.random.default_rng(2000 + seed)
    n = 6
    qpos = np.tile(m.qpos0, (n, 1)) + rng.uniform(-0.4, 0.4, (n, m.nq))
    for j in range(m.njnt):
        if m.jnt_type[j] == 0:
            a = m.jnt_qposadr[j]; q = rng.normal(size=(n, 4)); qpos[:, a+3:a+7] = q/np.linalg.norm(q, axis=1, keepdims=True)
            qpos[:, a+2] = rng.uniform(-0.05, 0.1, n)
    qvel = rng.normal(size=(n, m.nv))*0.2
    ctrl = rng.uniform(-0.4, 0.4, (n, max(m.nu, 1)))[:, :m.nu]
    phys = BatchedPhysics(m, n)
    assert phys.kernel_info()['threads_per_env'] == (32 if two_per_wave else 64)
    d = phys.data
    f32 = lambda a: torch.as_tensor(a, dtype=torch.float32)
    d.qpos[:] = f32(qpos); d.qvel[:] = f32(qvel)
    if m.nu:
        d.ctrl[:] = f32(ctrl)
    r64 = lambda t: t.cpu().numpy().astype(np.float64)
    q32, v32, c32 = r64(d.qpos), r64(d.qvel), r64(d.ctrl)
    phys.step(1)
    torch.cuda.synchronize()
    assert int((d.status & ~FMJ_WARN_CONTACTFULL).abs().sum()) == 0     # a truncated contact list (both sides truncate alike) is the only bit allowed
    fds = [oracle.forward_debug(m, q32[e], v32[e], ctrl=c32[e] if m.nu else None) for e in range(n)]
    ncon_ref = np.array([fd['ncon'] for fd in fds])
    assert np.array_equal(d.ncon.cpu().numpy(), ncon_ref), (d.ncon.cpu().numpy(), ncon_ref)
    ref = oracle.step(m, q32, v32, ctrl=c32 if m.nu else None)

    def err(k):
        a = r64(getattr(d, k)); bb = ref[k]
        return np.abs(a - bb).max()/max(np.abs(bb).max(), 1e-9)
    for k, tol in (('xpos', 5e-6), ('qvel', 3e-3), ('qpos', 2e-5)):
        assert err(k) < tol, (seed, m.nbody, m.nv, ncon_ref, k, err(k))
    for e in range(n):
        fd = fds[e]
        if fd['ncon']:
            f = fd['efc_force'][fd['nefc'] - 4*fd['ncon']:fd['nefc']].reshape(-1, 4).sum(1)
            got = d.contact.cpu().numpy()[e, :fd['ncon'], 12]
            assert np.allclose(got, f, rtol=3e-2, atol=2e-3*max(1.0, np.abs(f).max())), (seed, e, got, f)


MESH_SEEDS = (301, 302, 303, 304, 305, 306, 307, 308, 310, 311)      # draws with a movable tree and at least one mesh geom (300 and 309 have none)


@pytest.mark.parametrize('seed', MESH_SEEDS)
def test_random_tree_with_mesh_geoms(oracle, seed):
    """Random trees whose collision shapes include convex meshes (random point clouds, hulls of 4 to ~30 vertices, rotated
    and off-centre), with limits, over a plane: contact counts, the state after one step and the contact forces vs the
    oracle (models with meshes run the PAIRS instantiation of the constraint kernel)."""
    import torch
    from farms_mujoco_amd.physics import BatchedPhysics
    m = random_tree(seed, contacts=True, meshes=True)
    assert m is not None and m.nv > 0 and m.nmeshvert > 0, f'seed {seed} no longer draws a tree with mesh geoms: pick another for MESH_SEEDS'
    rng = np.random.default_rng(5000 + seed)
    n = 6
    qpos = np.tile(m.qpos0, (n, 1)) + rng.uniform(-0.4, 0.4, (n, m.nq))
    for j in range(m.njnt):
        if m.jnt_type[j] == 0:
            a = m.jnt_qposadr[j]; q = rng.normal(size=(n, 4)); qpos[:, a+3:a+7] = q/np.linalg.norm(q, axis=1, keepdims=True)
            qpos[:, a+2] = rng.uniform(-0.05, 0.1, n)
    qvel = rng.normal(size=(n, m.nv))*0.2
    phys = BatchedPhysics(m, n)
    d = phys.data
    f32 = lambda a: torch.as_tensor(a, dtype=torch.float32)
    d.qpos[:] = f32(qpos); d.qvel[:] = f32(qvel)
    r64 = lambda t: t.cpu().numpy().astype(np.float64)
    q32, v32 = r64(d.qpos), r64(d.qvel)
    phys.step(1)
    torch.cuda.synchronize()
    assert int((d.status & ~FMJ_WARN_CONTACTFULL).abs().sum()) == 0
    fds = [oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu) if m.nu else None) for e in range(n)]
    ncon_ref = np.array([fd['ncon'] for fd in fds])
    assert np.array_equal(d.ncon.cpu().numpy(), ncon_ref), (d.ncon.cpu().numpy(), ncon_ref)
    ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)) if m.nu else None)
    for k, tol in (('xpos', 5e-6), ('qvel', 3e-3), ('qpos', 2e-5)):
        a = r64(getattr(d, k)); bb = ref[k]
        e_ = np.abs(a - bb).max()/max(np.abs(bb).max(), 1e-9)
        assert e_ < tol, (seed, m.nbody, m.nv, ncon_ref, k, e_)
    for e in range(n):
        fd = fds[e]
        if fd['ncon']:
            got = d.contact.cpu().numpy()[e, :fd['ncon']]
            assert np.allclose(got[:, :3], fd['contact'][:fd['ncon'], :3], atol=2e-6), (seed, e)
            f = fd['efc_force'][fd['nefc'] - 4*fd['ncon']:fd['nefc']].reshape(-1, 4).sum(1)
            assert np.allclose(got[:, 12], f, rtol=3e-2, atol=2e-3*max(1.0, np.abs(f).max())), (seed, e, got[:, 12], f)
