"""Known-answer tests that pin the fp64 CPU oracle (the reference has no tests or vectors of its own:
SURVEY §4, §8c).  Analytic recurrences of MuJoCo's semi-implicit Euler, an independent Jacobian-based mass
matrix, an independent Newton-Euler inverse dynamics, and closed forms of the drag law (drag.pyx)."""
import numpy as np
import pytest

from farms_mujoco_amd.model import (ModelBuilder, salamander33, eel, centipede, np_mass_matrix, np_kinematics,
                                    np_body_jacobian, quat2mat, quat_mul, axisangle2quat, JNT_FREE, JNT_HINGE, JNT_SLIDE)


def _free_body(inertia=(0.01, 0.02, 0.03), mass=2.0, gravity=(0, 0, -9.81), h=1e-3):
    b = ModelBuilder('free', timestep=h, gravity=gravity)
    b.add_body('b', 'world', pos=(0, 0, 1.0), mass=mass, inertia=inertia, joint='free')
    return b.compile()


def test_free_fall_recurrence(oracle):
    """Semi-implicit Euler: v_n = -g h n, z_n = z0 - g h^2 n(n+1)/2 exactly."""
    m = _free_body()
    n = 250
    o = oracle.step(m, m.qpos0[None], np.zeros((1, 6)), n_steps=n)
    h, g = m.timestep, 9.81
    assert abs(o['qvel'][0, 2] + g*h*n) < 1e-12
    assert abs(o['qpos'][0, 2] - (1.0 - g*h*h*n*(n + 1)/2)) < 1e-12
    assert np.allclose(o['qpos'][0, 3:7], [1, 0, 0, 0])


def test_hinge_implicit_damping_recurrence(oracle):
    """Single damped hinge, no gravity: omega+ = omega * I/(I + h B) (Euler with implicit joint damping)."""
    h, B = 1e-3, 0.05
    b = ModelBuilder('hinge', timestep=h, gravity=(0, 0, 0))
    b.add_body('l', 'world', mass=0.5, ipos=(0.1, 0, 0), inertia=(1e-3, 2e-3, 3e-3), joint='hinge', axis=(0, 0, 1), damping=B)
    m = b.compile()
    I = 3e-3 + 0.5*0.1**2
    w = 2.0
    o = oracle.step(m, np.zeros((1, 1)), np.array([[w]]), n_steps=5)
    assert abs(o['qvel'][0, 0] - w*(I/(I + h*B))**5) < 1e-13


def test_free_body_momentum(oracle):
    """Torque-free asymmetric body: world-frame angular momentum is conserved up to the O(h) integrator error,
    and the drift halves when h halves."""
    drifts = []
    for h in (2e-4, 1e-4):
        m = _free_body(gravity=(0, 0, 0), h=h)
        q = m.qpos0[None].copy(); v = np.array([[0.1, -0.2, 0.3, 1.0, 2.0, -1.5]])
        I = np.diag(m.body_inertia[1])
        L0 = quat2mat(q[0, 3:7]) @ I @ v[0, 3:]
        o = oracle.step(m, q, v, n_steps=int(round(0.2/h)))
        L1 = quat2mat(o['qpos'][0, 3:7]) @ I @ o['qvel'][0, 3:]
        drifts.append(np.linalg.norm(L1 - L0)/np.linalg.norm(L0))
        assert np.allclose(o['qvel'][0, :3], v[0, :3])
        assert abs(np.linalg.norm(o['qpos'][0, 3:7]) - 1) < 1e-12
    assert drifts[0] < 2e-3 and 1.7 < drifts[0]/drifts[1] < 2.3


def test_position_actuator_steady_state(oracle):
    """Damped pendulum with a position actuator settles where kp (ctrl - q) balances the gravity torque."""
    b = ModelBuilder('pend', timestep=1e-3)
    b.add_body('l', 'world', mass=0.2, ipos=(0.1, 0, 0), inertia=(1e-4, 1e-3, 1e-3), joint='hinge', jname='j',
               axis=(0, 1, 0), damping=0.05)
    b.add_joint_actuators('j', kp=2.0)
    m = b.compile()
    ctrl = np.array([[0.3, 0.0, 0.0]])
    o = oracle.step(m, np.zeros((1, 1)), np.zeros((1, 1)), ctrl=ctrl, n_steps=20000)
    q = o['qpos'][0, 0]
    grav = 0.2*9.81*0.1*np.cos(q)          # about +y at angle q (CoM along x, rotates towards -z for q>0)
    assert abs(o['qvel'][0, 0]) < 1e-9
    assert abs(2.0*(0.3 - q) + grav) < 1e-8


def _inverse_dynamics(m, qpos, qvel, qacc):
    """Independent Newton-Euler inverse dynamics in plain world-frame vectors (no spatial algebra, no common
    reference point): returns tau such that tau = M qacc + bias."""
    kin = np_kinematics(m, qpos)
    nb = m.nbody
    w = np.zeros((nb, 3)); al = np.zeros((nb, 3)); vo = np.zeros((nb, 3)); ao = np.zeros((nb, 3))   # body-origin vel/acc
    ao[0] = -m.gravity
    for b in range(1, nb):
        p = m.body_parentid[b]
        j = m.body_jntadr[b]
        r = kin['xpos'][b] - kin['xpos'][p]
        # rigidly attached to the parent first
        w[b] = w[p]; al[b] = al[p]
        vo[b] = vo[p] + np.cross(w[p], r)
        ao[b] = ao[p] + np.cross(al[p], r) + np.cross(w[p], np.cross(w[p], r))
        if j < 0:
            continue
        a = m.jnt_dofadr[j]
        if m.jnt_type[j] == JNT_FREE:
            Rb = quat2mat(kin['xquat'][b])
            w[b] = Rb @ qvel[a+3:a+6]; vo[b] = qvel[a:a+3]
            al[b] = Rb @ qacc[a+3:a+6] + np.cross(w[b], Rb @ qvel[a+3:a+6])*0
            ao[b] = qacc[a:a+3] - m.gravity
        elif m.jnt_type[j] == JNT_HINGE:
            ax = kin['xaxis'][j]; anchor = kin['xanchor'][j]
            ra = anchor - kin['xpos'][p]
            va = vo[p] + np.cross(w[p], ra)
            aa = ao[p] + np.cross(al[p], ra) + np.cross(w[p], np.cross(w[p], ra))
            wb = w[p] + ax*qvel[a]
            alb = al[p] + ax*qacc[a] + np.cross(w[p], ax*qvel[a])
            rb = kin['xpos'][b] - anchor
            w[b], al[b] = wb, alb
            vo[b] = va + np.cross(wb, rb)
            ao[b] = aa + np.cross(alb, rb) + np.cross(wb, np.cross(wb, rb))
        else:
            ax = kin['xaxis'][j]
            vo[b] = vo[b] + ax*qvel[a]
            ao[b] = ao[b] + ax*qacc[a] + 2*np.cross(w[p], ax*qvel[a])
    F = np.zeros((nb, 3)); N = np.zeros((nb, 3))
    for b in range(1, nb):
        rc = kin['xipos'][b] - kin['xpos'][b]
        ac = ao[b] + np.cross(al[b], rc) + np.cross(w[b], np.cross(w[b], rc))
        Rb = quat2mat(quat_mul(kin['xquat'][b], m.body_iquat[b]))
        Iw = Rb @ np.diag(m.body_inertia[b]) @ Rb.T
        F[b] = m.body_mass[b]*ac
        N[b] = Iw @ al[b] + np.cross(w[b], Iw @ w[b])
    tau = np.zeros(m.nv)
    for b in range(1, nb):
        jp, jr = np_body_jacobian(m, kin, b, kin['xipos'][b])
        tau += jp.T @ F[b] + jr.T @ N[b]
    return tau + m.dof_armature*qacc


@pytest.mark.parametrize('maker', [salamander33, eel, centipede])
def test_forward_inverse_consistency(oracle, maker):
    """M(q) qacc_smooth + bias - (passive + actuator + xfrc) = 0 with the residual from the INDEPENDENT inverse
    dynamics; and M itself equals the Jacobian-sum mass matrix."""
    m = maker()
    rng = np.random.default_rng(7)
    q = m.qpos0.copy(); q[7:] += rng.uniform(-0.4, 0.4, m.nq - 7)
    qq = rng.normal(size=4); q[3:7] = qq/np.linalg.norm(qq)
    v = rng.normal(size=m.nv)*0.7
    ctrl = rng.normal(size=m.nu)*0.1
    xf = rng.normal(size=(m.nbody, 6))*0.05; xf[0] = 0
    o = oracle.forward_debug(m, q, v, ctrl=ctrl, xfrc_applied=xf)
    M2 = np_mass_matrix(m, q)
    assert np.abs(o['M'] - M2).max() < 1e-13*max(1, np.abs(M2).max())
    tau = _inverse_dynamics(m, q, v, o['qacc_smooth'])
    applied = o['qfrc_passive'] + o['qfrc_actuator'] + o['qfrc_xfrc']
    scale = max(1.0, np.abs(applied).max())
    assert np.abs(tau - applied).max() < 1e-9*scale
    # bias alone: inverse dynamics at zero acceleration
    assert np.abs(_inverse_dynamics(m, q, v, np.zeros(m.nv)) - o['qfrc_bias']).max() < 1e-10*scale
    # xfrc: J' f with independent Jacobians
    kin = np_kinematics(m, q)
    qx = np.zeros(m.nv)
    for b in range(1, m.nbody):
        jp, jr = np_body_jacobian(m, kin, b, kin['xipos'][b])
        qx += jp.T @ xf[b, :3] + jr.T @ xf[b, 3:]
    assert np.abs(qx - o['qfrc_xfrc']).max() < 1e-13


def test_body_velocity_sensors(oracle):
    """framelinvel/frameangvel (objtype=body): velocity of the body CoM = J(xipos) qvel."""
    m = salamander33()
    rng = np.random.default_rng(11)
    q = m.qpos0.copy(); q[7:] += rng.uniform(-0.3, 0.3, m.nq - 7)
    v = rng.normal(size=m.nv)
    o = oracle.forward_debug(m, q, v)
    kin = np_kinematics(m, q)
    for b in range(1, m.nbody):
        jp, jr = np_body_jacobian(m, kin, b, kin['xipos'][b])
        assert np.allclose(o['sensordata'][6*(b-1):6*(b-1)+3], jp @ v, atol=1e-12)
        assert np.allclose(o['sensordata'][6*(b-1)+3:6*(b-1)+6], jr @ v, atol=1e-12)


def test_double_pendulum_energy_drift(oracle):
    """Undamped double pendulum: energy drift is O(h)."""
    def build(h):
        b = ModelBuilder('dp', timestep=h)
        b.add_body('a', 'world', mass=1.0, ipos=(0.25, 0, 0), inertia=(1e-3, 0.02, 0.02), joint='hinge', axis=(0, 1, 0))
        b.add_body('b', 'a', pos=(0.5, 0, 0), mass=0.5, ipos=(0.2, 0, 0), inertia=(1e-3, 0.01, 0.01), joint='hinge', axis=(0, 1, 0))
        return b.compile()

    def energy(m, q, v):
        M = np_mass_matrix(m, q)
        kin = np_kinematics(m, q)
        pe = sum(m.body_mass[b]*9.81*kin['xipos'][b][2] for b in range(1, m.nbody))
        return 0.5*v @ M @ v + pe
    drift = []
    for h in (1e-3, 5e-4):
        m = build(h)
        q0 = np.array([0.3, -0.5]); v0 = np.array([0.5, 1.0])
        o = oracle.step(m, q0[None], v0[None], n_steps=int(round(0.5/h)))
        drift.append(abs(energy(m, o['qpos'][0], o['qvel'][0]) - energy(m, q0, v0)))
    assert drift[0] < 5e-2 and 1.6 < drift[0]/drift[1] < 2.4


# ---- drag closed forms (reference drag.pyx) -------------------------------------------------------------

def _one_link_rows(pos, quat_xyzw, lin, ang):
    row = np.zeros((1, 1, 20))
    row[0, 0, 0:3] = pos; row[0, 0, 3:7] = quat_xyzw; row[0, 0, 7:10] = pos; row[0, 0, 10:14] = quat_xyzw
    row[0, 0, 14:17] = lin; row[0, 0, 17:20] = ang
    return row


_SWIM = dict(links_index=[0], xfrc_index=[0], body_index=[1], coefficients=[[[-0.1, -0.2, -0.3], [-0.01, -0.02, -0.03]]],
             masses=[0.5], heights=[0.04], densities=[800.0])


def test_drag_identity_orientation(oracle):
    """F = visc*c*sign(v)*v^2, tau = c_ang*sign(w)*w^2 (drag.pyx:83-88,104-108), fully submerged, no buoyancy."""
    v = np.array([1.0, -2.0, 0.5]); w = np.array([-0.3, 0.2, 1.0])
    rows = _one_link_rows([0, 0, -1.0], [0, 0, 0, 1], v, w)
    water = dict(surface=0.0, velocity=[0, 0, 0], viscosity=1.5, use_buoyancy=False)
    x, xa = oracle.drag(_SWIM, water, rows, np.zeros((1, 1, 6)), 2)
    assert np.allclose(x[0, 0, :3], 1.5*np.array([-0.1, -0.2, -0.3])*np.sign(v)*v*v)
    assert np.allclose(x[0, 0, 3:], np.array([-0.01, -0.02, -0.03])*np.sign(w)*w*w)
    assert np.allclose(xa[0, 1], x[0, 0])        # identity orientation: world == link frame


def test_drag_yaw_rotation_and_current(oracle):
    """Link yawed by 90 deg in a water current: velocity relative to the water, expressed in the link frame."""
    q = [0, 0, np.sin(np.pi/4), np.cos(np.pi/4)]        # x,y,z,w : +90 deg about z
    vw = np.array([0.4, 0.0, 0.0]); cur = np.array([0.1, 0.2, 0.0])
    rows = _one_link_rows([0, 0, -1.0], q, vw, [0, 0, 0])
    water = dict(surface=0.0, velocity=cur, viscosity=1.0, use_buoyancy=False)
    x, xa = oracle.drag(_SWIM, water, rows, np.zeros((1, 1, 6)), 2)
    rel = vw - cur
    vl = np.array([rel[1], -rel[0], rel[2]])            # world -> link frame for +90 deg yaw
    f_link = np.array([-0.1, -0.2, -0.3])*np.sign(vl)*vl*vl
    assert np.allclose(x[0, 0, :3], f_link)
    assert np.allclose(xa[0, 1, :3], [-f_link[1], f_link[0], f_link[2]])     # glue rotates back to world


def test_drag_surface_and_buoyancy_clamp(oracle):
    """Above the surface: row untouched (drag.pyx:192-194); partial submersion: -1000 m g / rho *
    min(max(s - z, 0)/h, 1) with g = -9.81 (drag.pyx:139-146,409)."""
    water = dict(surface=0.0, velocity=[0, 0, 0], viscosity=1.0, use_buoyancy=True)
    sentinel = np.full((1, 1, 6), 3.0)
    x, xa = oracle.drag(_SWIM, water, _one_link_rows([0, 0, 0.01], [0, 0, 0, 1], [1, 1, 1], [0, 0, 0]), sentinel, 2)
    assert np.all(x == 3.0) and np.all(xa[0, 1] == 0.0)
    for z, frac in ((-0.01, 0.25), (-0.04, 1.0), (-1.0, 1.0)):
        x, _ = oracle.drag(_SWIM, water, _one_link_rows([0, 0, z], [0, 0, 0, 1], [0, 0, 0], [0, 0, 0]), np.zeros((1, 1, 6)), 2)
        assert np.allclose(x[0, 0, :3], [0, 0, 1000*0.5*9.81/800.0*frac])


def test_physics2data_layout(oracle):
    """Column layout and unit scaling of the links / joints rows (physics.py:449-524)."""
    m = salamander33()
    rng = np.random.default_rng(2)
    n = 2
    qpos = rng.normal(size=(n, m.nq)); qvel = rng.normal(size=(n, m.nv))
    xpos = rng.normal(size=(n, m.nbody, 3)); xquat = rng.normal(size=(n, m.nbody, 4)); xipos = rng.normal(size=(n, m.nbody, 3))
    sd = rng.normal(size=(n, m.nsensordata))
    lb = np.arange(1, m.nbody); jj = np.nonzero(m.jnt_type != JNT_FREE)[0]
    units = (2.0, 24.0, 48.0, 4.0, 2.0)     # meters, newtons, torques, velocity, angular_velocity
    links, joints = oracle.physics2data(m, qpos, qvel, xpos, xquat, xipos, sd, lb, jj, units=units)
    b = 5
    assert np.allclose(links[1, b-1, 0:3], xipos[1, b]/2.0) and np.allclose(links[1, b-1, 7:10], xpos[1, b]/2.0)
    assert np.allclose(links[1, b-1, 3:7], xquat[1, b][[1, 2, 3, 0]]) and np.allclose(links[1, b-1, 10:14], xquat[1, b][[1, 2, 3, 0]])
    assert np.allclose(links[1, b-1, 14:17], sd[1, 6*(b-1):6*(b-1)+3]/4.0)
    assert np.allclose(links[1, b-1, 17:20], sd[1, 6*(b-1)+3:6*(b-1)+6]/2.0)
    j = 7; jid = jj[j]
    assert joints[0, j, 0] == qpos[0, m.jnt_qposadr[jid]] and np.isclose(joints[0, j, 1], qvel[0, m.jnt_dofadr[jid]]/2.0)
    act = 6*(m.nbody-1) + 3*len(jj)
    acts = [a for a in range(m.nu) if m.actuator_jntid[a] == jid]
    assert len(acts) == 3 and np.isclose(joints[0, j, 8], sum(sd[0, act + a] for a in acts)/48.0)
    assert np.isclose(joints[0, j, 9], sd[0, 6*(m.nbody-1) + 3*j + 2]/48.0)


# ---- closed forms from the textbook, derived independently of the recursive algorithms -------------------

def test_double_pendulum_textbook_accelerations(oracle):
    """Point-mass double pendulum: the angular accelerations of the Lagrangian closed form (absolute angles from the
    downward vertical; e.g. the standard result with 2 m1 + m2 - m2 cos(2 th1 - 2 th2) in the denominator) at random
    states, against qacc_smooth of the oracle (relative joint angles).  Nothing of CRBA / RNE enters the expected value."""
    m1, m2, L1, L2, g = 0.7, 0.4, 0.35, 0.25, 9.81
    b = ModelBuilder('dp_point', timestep=1e-3)
    eps = 1e-12                                   # point masses: body inertia negligible
    b.add_body('a', 'world', mass=m1, ipos=(0, 0, -L1), inertia=(eps, eps, eps), joint='hinge', axis=(0, 1, 0))
    b.add_body('b', 'a', pos=(0, 0, -L1), mass=m2, ipos=(0, 0, -L2), inertia=(eps, eps, eps), joint='hinge', axis=(0, 1, 0))
    m = b.compile()
    rng = np.random.default_rng(0)
    for _ in range(20):
        th1, th2 = rng.uniform(-2.5, 2.5, 2)
        w1, w2 = rng.uniform(-3, 3, 2)
        q = np.array([th1, th2 - th1]); v = np.array([w1, w2 - w1])
        o = oracle.forward_debug(m, q, v)
        # rotation about +y by th moves the hanging mass (0,0,-L) to (-L sin th, 0, -L cos th): the textbook's angle with x -> -x,
        # which leaves the equations unchanged (they are odd in the pair of angles)
        den = 2*m1 + m2 - m2*np.cos(2*th1 - 2*th2)
        a1 = (-g*(2*m1 + m2)*np.sin(th1) - m2*g*np.sin(th1 - 2*th2)
              - 2*np.sin(th1 - th2)*m2*(w2*w2*L2 + w1*w1*L1*np.cos(th1 - th2)))/(L1*den)
        a2 = (2*np.sin(th1 - th2)*(w1*w1*L1*(m1 + m2) + g*(m1 + m2)*np.cos(th1) + w2*w2*L2*m2*np.cos(th1 - th2)))/(L2*den)
        assert np.allclose(o['qacc_smooth'], [a1, a2 - a1], rtol=1e-8, atol=1e-8), (o['qacc_smooth'], a1, a2 - a1)


def test_torsion_spring_frequency_and_damped_decay(oracle):
    """One hinge with joint stiffness k, inertia I about the axis: undamped it oscillates at sqrt(k / I) (period from zero
    crossings); with damping c the envelope decays like exp(-c t / (2 I))."""
    I, k, h = 2e-3, 0.8, 1e-4
    def build(c):
        b = ModelBuilder('spring', timestep=h, gravity=(0, 0, 0))
        b.add_body('a', 'world', mass=0.1, inertia=(I, I, I), joint='hinge', axis=(0, 0, 1), stiffness=k, damping=c)
        return b.compile()
    m = build(0.0)
    n = 20000
    qs = []
    q, v = np.array([[0.3]]), np.array([[0.0]])
    for _ in range(n//100):
        o = oracle.step(m, q, v, n_steps=100); q, v = o['qpos'], o['qvel']; qs.append(q[0, 0])
    qs = np.array(qs); t = (np.arange(len(qs)) + 1)*100*h
    zc = t[:-1][np.sign(qs[:-1]) != np.sign(qs[1:])]
    period = 2*np.mean(np.diff(zc))
    assert abs(period - 2*np.pi/np.sqrt(k/I)) < 2e-2*period
    c = 4e-3
    m = build(c)
    o = oracle.step(m, np.array([[0.3]]), np.array([[0.0]]), n_steps=n)
    amp = np.hypot(o['qpos'][0, 0], o['qvel'][0, 0]/np.sqrt(k/I - (c/(2*I))**2))
    assert abs(amp/0.3 - np.exp(-c*n*h/(2*I))) < 2e-2


def test_sinking_link_terminal_velocity(oracle):
    """A body heavier than water sinks at the speed where the drag sign(v) v^2 c equals weight minus buoyancy
    (drag.pyx:83-88,139-146): v_t = sqrt((m g - buoyancy) / (viscosity c_z))."""
    from farms_mujoco_amd.model import ModelBuilder as MB
    mass, cz, density, height = 0.5, 2.0, 1500.0, 0.05
    b = MB('sinker', timestep=1e-3)
    b.add_body('a', 'world', pos=(0, 0, -1.0), mass=mass, inertia=(1e-3, 1e-3, 1e-3), joint='free')
    b.set_swimming('a', density=density, drag_coefficients=[[-0.5, -0.5, -cz], [-1e-3, -1e-3, -1e-3]], height=height)
    m = b.compile()
    water = dict(surface=0.0, velocity=[0, 0, 0], viscosity=1.0, gravity=-9.81, use_buoyancy=True)
    fd = oracle.forward_debug(m, m.qpos0, np.zeros(6))
    st = dict(qpos=m.qpos0[None], qvel=np.zeros((1, 6)), xpos=fd['xpos'][None], xquat=fd['xquat'][None], xipos=fd['xipos'][None],
              sensordata=fd['sensordata'][None])
    swim = dict(links_index=[0], xfrc_index=[0], body_index=[1], coefficients=[[[-0.5, -0.5, -cz], [-1e-3, -1e-3, -1e-3]]],
                masses=[mass], heights=[height], densities=[density])
    o = oracle.run_fused(m, st, 4000, swim=swim, water=water, buffer_size=1, controller=0, ctrl=np.zeros((1, 0)))
    buoy = 1000.0*9.81*mass/density                      # fully submerged
    vt = np.sqrt((mass*9.81 - buoy)/cz)
    assert abs(-o['qvel'][0, 2] - vt) < 1e-3*vt, (o['qvel'][0, 2], vt)


def _servo_hinge(integrator, h=2e-3, B=0.05, kv=0.4, kp=0.0, forcerange=None):
    b = ModelBuilder('servo', timestep=h, gravity=(0, 0, 0))
    b.options['integrator'] = integrator
    b.add_body('l', 'world', mass=0.5, ipos=(0.1, 0, 0), inertia=(1e-3, 2e-3, 3e-3), joint='hinge', jname='j', axis=(0, 0, 1), damping=B)
    b.add_joint_actuators('j', kp=kp, kv=kv, forcerange=forcerange)
    return b.compile(), 3e-3 + 0.5*0.1**2


def test_implicitfast_velocity_servo_recurrence(oracle):
    """implicitfast (MuJoCo mj_implicit with the Coriolis derivatives dropped): (I + h(B + kv)) (w+ - w) = h (kv (c - w) - B w), i.e. the
    velocity gain of the actuator joins the joint damping on the implicit side.  Euler keeps the actuator explicit: only B is implicit."""
    h, B, kv, c, w0 = 2e-3, 0.05, 0.4, 1.5, -0.7
    for integ in ('implicitfast', 'Euler'):
        m, I = _servo_hinge(integ, h, B, kv)
        assert m.integrator == (3 if integ == 'implicitfast' else 0)
        o = oracle.step(m, np.zeros((1, 1)), np.array([[w0]]), ctrl=np.array([[0.0, c, 0.0]]), n_steps=40)
        w = w0
        for _ in range(40):
            w += h*(kv*(c - w) - B*w)/(I + h*(B + (kv if integ == 'implicitfast' else 0.0)))
        assert abs(o['qvel'][0, 0] - w) < 1e-13, integ


def test_implicitfast_skips_a_clamped_actuator(oracle):
    """mjd_actuator_vel: an actuator whose force sits on its forcerange has no velocity derivative, so a saturated servo steps exactly
    like Euler; the derivative comes back once the force leaves the limit."""
    h, B, kv, c = 2e-3, 0.05, 0.4, 5.0
    fr = (-0.2, 0.2)                                     # kv (c - w) = 2.0 at w = 0: saturated until w > c - 0.5
    out = {}
    for integ in ('implicitfast', 'Euler'):
        m, I = _servo_hinge(integ, h, B, kv, forcerange=fr)
        out[integ] = oracle.step(m, np.zeros((1, 1)), np.zeros((1, 1)), ctrl=np.array([[0.0, c, 0.0]]), n_steps=25)['qvel'][0, 0]
    w = 0.0
    for _ in range(25):
        w += h*(0.2 - B*w)/(I + h*B)
    assert abs(out['Euler'] - w) < 1e-13 and out['implicitfast'] == out['Euler']
    m, I = _servo_hinge('implicitfast', h, B, kv, forcerange=fr)
    w0 = c - 0.1                                          # force = 0.04: inside the range
    o = oracle.step(m, np.zeros((1, 1)), np.array([[w0]]), ctrl=np.array([[0.0, c, 0.0]]), n_steps=1)
    assert abs(o['qvel'][0, 0] - (w0 + h*(kv*(c - w0) - B*w0)/(I + h*(B + kv)))) < 1e-14
