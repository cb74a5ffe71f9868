"""The oracle's constraint solvers cross-checked against each other (no GPU): PGS works on the dual problem
(0.5 f'(A + R) f + f'b over f >= 0), Newton and CG on the primal one (0.5 |qacc - qacc_smooth|^2_M + s(J qacc - aref)).  They
are different algorithms for one convex problem, so a converged PGS, the Newton minimiser and the CG minimiser must agree - the
`forward_inverse_consistency` of the constraint path: a misreading shared by the HIP kernel and the oracle's PGS (both written
from the same notes) would not survive it.  Reference mjcf.py:1348-1359 makes Newton the reference's own fallback solver."""
import copy

import numpy as np
import pytest

from farms_mujoco_amd.model import ModelBuilder, GEOM_PLANE, GEOM_BOX, GEOM_SPHERE, SOLVERS, CONES


def _with(m, solver, iterations, tolerance):
    m2 = copy.copy(m)
    m2.solver = SOLVERS[solver]; m2.solver_iterations = iterations; m2.solver_tolerance = tolerance
    return m2


def _walker():
    from farms_mujoco_amd.model import salamander33
    return salamander33(contacts=True, limits=True, spawn_z=0.045)


def _box_bot():
    """A box on the plane carrying a limited hinge arm with a ball at its end that also touches the plane."""
    b = ModelBuilder('boxbot', timestep=1e-3)
    b.add_body('box', 'world', pos=(0, 0, 0.049), mass=0.8, inertia=(2e-3, 2e-3, 2e-3), joint='free')
    b.add_geom('box', GEOM_BOX, (0.08, 0.05, 0.05), friction=(0.8, 0, 0))
    b.add_body('arm', 'box', pos=(0.08, 0, 0), mass=0.1, ipos=(0.05, 0, 0), inertia=(1e-5, 1e-4, 1e-4), joint='hinge', axis=(0, 1, 0),
               limited=True, range=(-0.2, 0.3), damping=1e-3)
    b.add_geom('arm', GEOM_SPHERE, (0.02,), pos=(0.1, 0, 0), friction=(0.5, 0, 0))
    b.add_geom('world', GEOM_PLANE, (0, 0, 0))
    b.add_position_actuator('joint_arm', kp=0.5)
    b.options['max_contacts'] = 8
    return b.compile()


def _states(m, n, seed, spread=0.1):
    rng = np.random.default_rng(seed)
    qpos = np.tile(m.qpos0, (n, 1))
    free = m.jnt_type[0] == 0
    lo = 7 if free else 0
    qpos[:, lo:] += rng.uniform(-spread, spread, (n, m.nq - lo))
    qvel = rng.normal(size=(n, m.nv))*0.1
    return qpos, qvel


def _agree(oracle, m, qpos, qvel, ctrl, warm, tol=1e-8):
    """PGS to convergence, Newton and CG on the same steps: forces, qacc and the next state."""
    a = oracle.step_tf(_with(m, 'pgs', 200000, 0.0), qpos, qvel, ctrl=ctrl, warmstart=warm)
    b = oracle.step_tf(_with(m, 'newton', 100, 1e-14), qpos, qvel, ctrl=ctrl, warmstart=warm)
    c = oracle.step_tf(_with(m, 'cg', 2000, 1e-16), qpos, qvel, ctrl=ctrl, warmstart=warm)
    assert np.array_equal(a['nefc'], b['nefc']) and np.array_equal(a['nefc'], c['nefc']) and a['nefc'].max() > 0
    worst = 0.0
    for e in range(len(qpos)):
        n = a['nefc'][e]
        if n == 0:
            continue
        fs = max(np.abs(b['efc'][e, :n, 0]).max(), 1e-3)
        for o, wt in ((a, 1.0), (c, 1e-2)):         # CG's line search stops it at ~1e-8 of the forces: held to 1e-6
            worst = max(worst, wt*np.abs(o['efc'][e, :n, 0] - b['efc'][e, :n, 0]).max()/fs,
                        wt*np.abs(o['warmstart'][e] - b['warmstart'][e]).max()/max(np.abs(b['warmstart'][e]).max(), 1.0))
        # KKT of the Newton solution on the dual problem it never saw: f >= 0, A f + b >= 0, complementary
        f = b['efc'][e, :n, 0]; r = b['AR'][e, :n, :n] @ f + b['efc'][e, :n, 1]
        assert f.min() >= 0.0
        assert r.min() > -1e-7*max(np.abs(b['efc'][e, :n, 1]).max(), 1.0), r.min()
        assert np.abs(f*r).max() < 1e-7*fs*max(np.abs(b['efc'][e, :n, 1]).max(), 1.0)
    assert worst < tol, worst
    assert np.abs(a['qvel'] - b['qvel']).max() < 1e-9 and np.abs(c['qvel'] - b['qvel']).max() < 1e-4
    return a, b, c


def test_pgs_newton_cg_agree_on_the_walker(oracle):
    m = _walker()
    qpos, qvel = _states(m, 6, 0)
    qpos[:3, 2] = 0.012                      # three animals pressed into the floor: belly contacts, many rows
    qpos[:, 7 + 3] = 1.25                    # a spine joint past its limit
    a, b, c = _agree(oracle, m, qpos, qvel, np.zeros((6, m.nu)), None)
    assert a['ncon'].max() >= 8 and (b['iterations'] <= 30).all()
    # a second step from the first one's state, warm-started with its qacc
    _agree(oracle, m, b['qpos'], b['qvel'], np.zeros((6, m.nu)), b['warmstart'])


def test_pgs_newton_cg_agree_on_the_box_bot(oracle):
    m = _box_bot()
    qpos, qvel = _states(m, 4, 1, spread=0.05)
    qpos[:, 2] = 0.047                      # the box pressed 3 mm into the floor: four corner contacts
    qpos[:2, 7] = 0.31                      # arm on its upper limit
    qpos[2:, 7] = 0.24                      # arm's ball on the floor
    a, b, _ = _agree(oracle, m, qpos, qvel, np.full((4, m.nu), 0.4), None)
    assert a['ncon'].min() >= 4


@pytest.mark.parametrize('seed', range(100, 110))
def test_pgs_newton_cg_agree_on_random_contact_trees(oracle, seed):
    from test_gpu_random_trees import random_tree
    m = random_tree(seed, contacts=True)
    if m is None or m.nv == 0:
        pytest.skip('degenerate draw')
    rng = np.random.default_rng(2000 + seed)
    n = 4
    qpos = np.tile(m.qpos0, (n, 1)) + rng.uniform(-0.4, 0.4, (n, m.nq))
    for j in range(m.njnt):
        if m.jnt_type[j] == 0:
            a = m.jnt_qposadr[j]; q = rng.normal(size=(n, 4)); qpos[:, a+3:a+7] = q/np.linalg.norm(q, axis=1, keepdims=True)
    qvel = rng.normal(size=(n, m.nv))*0.5
    ctrl = rng.uniform(-0.6, 0.6, (n, max(m.nu, 1)))[:, :m.nu]
    nefc = oracle.step_tf(m, qpos, qvel, ctrl=ctrl if m.nu else None)['nefc']
    if nefc.max() == 0:
        pytest.skip('no active constraint in this draw')
    _agree(oracle, m, qpos, qvel, ctrl if m.nu else None, None)


def test_newton_walk_equals_the_converged_pgs_walk(oracle):
    """200 steps of the trot.  Newton with MuJoCo's default settings (tolerance 1e-8, <= 100 iterations) converges in a handful
    of iterations per step, and a PGS that is allowed to converge (tolerance 1e-12: 150 - 300 sweeps per step) walks the
    same walk to 1e-6: two algorithms, one trajectory - and a measure of how little the walking dynamics amplifies a 1e-9
    difference per step (x100 in 300 steps; it is not chaotic on this horizon).  PGS cut at 50 sweeps (the configuration
    BASELINE configs[3] names) is a different, unconverged map: it drifts from that walk by ~1e-2."""
    from test_gpu_contacts import _trot_tape
    m = _walker()
    n, T = 2, 200
    tape = _trot_tape(m, n, T)
    models = dict(pgs50=m, newton=_with(m, 'newton', 100, 1e-8), pgs=_with(m, 'pgs', 5000, 1e-12))
    st = {k: dict(qpos=np.tile(m.qpos0, (n, 1)), qvel=np.zeros((n, m.nv)), ws=np.zeros((n, m.nv))) for k in models}
    its = {k: [] for k in models}
    for t in range(T):
        for k in models:
            o = oracle.step_tf(models[k], st[k]['qpos'], st[k]['qvel'], ctrl=tape[t], warmstart=st[k]['ws'], want_AR=False)
            st[k] = dict(qpos=o['qpos'], qvel=o['qvel'], ws=o['warmstart'])
            its[k].append(o['iterations'].max())
    assert max(its['newton']) <= 8 and np.median(its['newton']) <= 3, (max(its['newton']), np.median(its['newton']))
    assert max(its['pgs']) < 5000 and min(its['pgs50'][20:]) == 50
    assert np.abs(st['pgs']['qpos'] - st['newton']['qpos']).max() < 1e-6
    assert 1e-4 < np.abs(st['pgs50']['qpos'] - st['newton']['qpos']).max() < 5e-2
    assert st['newton']['qpos'][:, 2].min() > 0.0


# ---- elliptic cone (reference mjcf.py:1342-1347 forwards simulation_options.cone) -------------------------------------------

def _cone_kkt(o, e, mu_of_contact):
    """KKT residuals of the dual cone problem  min 0.5 f'(A + R) f + f'b,  f in K  for env e of a step_tf result: K is f >= 0 for
    a limit row and f_n >= 0, |f_t| <= mu f_n for the three rows of an elliptic contact; the gradient r = (A + R) f + b must lie in
    the dual cone (r >= 0; r_n >= mu |r_t|) and be orthogonal to f.  Written from the optimality conditions, not from a solver."""
    n = int(o['nefc'][e]); f = o['efc'][e, :n, 0]; b = o['efc'][e, :n, 1]; typ = o['efc'][e, :n, 4]; cid = o['efc'][e, :n, 5].astype(int)
    r = o['AR'][e, :n, :n] @ f + b
    fs = max(np.abs(f).max(), 1e-3); bs = max(np.abs(b).max(), 1.0)
    worst = 0.0; i = 0
    while i < n:
        if typ[i] == 2:
            mu = mu_of_contact[cid[i]]
            ft = np.hypot(f[i+1], f[i+2]); rt = np.hypot(r[i+1], r[i+2])
            worst = max(worst, -min(f[i], 0.0)/fs, max(ft - mu*f[i], 0.0)/fs, max(mu*rt - r[i], 0.0)/bs, abs(f[i:i+3] @ r[i:i+3])/(fs*bs))
            i += 3
        else:
            worst = max(worst, -min(f[i], 0.0)/fs, -min(r[i], 0.0)/bs, abs(f[i]*r[i])/(fs*bs))
            i += 1
    return worst


def _elliptic(m, impratio=1.0):
    m2 = copy.copy(m); m2.cone = CONES['elliptic']; m2.impratio = impratio
    return m2


@pytest.mark.parametrize('impratio', [1.0, 4.0])
def test_elliptic_newton_cg_and_the_cone_kkt(oracle, impratio):
    """cone = elliptic: Newton and CG minimise the primal cost with MuJoCo's three cone zones; their forces must satisfy the
    optimality conditions of the DUAL cone problem built from A, R, b - conditions neither solver ever sees.  PGS (ray update +
    friction QCQP, mj_solPGS) reaches the same forces wherever it moves at all: from zero force it cannot leave the apex of a
    cone whose normal residual is positive (its normal-only update clamps to 0 and clears friction) although a friction-bearing
    force lowers the cost - MuJoCo's PGS has that property, the test counts those contacts instead of hiding them."""
    total, stalled = 0, 0
    for mk in (_walker, _box_bot):
        m = _elliptic(mk(), impratio)
        n = 6
        qpos, qvel = _states(m, n, 0, spread=0.1 if mk is _walker else 0.05)
        if mk is _walker:
            qpos[:3, 2] = 0.012; qpos[:, 7 + 3] = 1.25
        else:
            qpos[:, 2] = 0.047; qpos[:3, 7] = 0.31; qpos[3:, 7] = 0.24
        qvel[:, :2] += 0.3                                   # sliding: the friction forces sit on the cone
        ctrl = np.zeros((n, m.nu))
        a = oracle.step_tf(_with(m, 'pgs', 20000, 0.0), qpos, qvel, ctrl=ctrl)
        b = oracle.step_tf(_with(m, 'newton', 100, 1e-14), qpos, qvel, ctrl=ctrl)
        c = oracle.step_tf(_with(m, 'cg', 5000, 1e-16), qpos, qvel, ctrl=ctrl)
        assert np.array_equal(a['nefc'], b['nefc']) and b['ncon'].min() >= 2
        assert (b['iterations'] <= 30).all()
        for e in range(n):
            ne = int(b['nefc'][e])
            mus = {}                                          # friction of a contact = max over its two geoms (MuJoCo's rule), floor mjMINMU
            for k in range(int(b['ncon'][e])):
                g1, g2 = int(b['contact'][e, k, 15]), int(b['contact'][e, k, 16])
                mus[k] = max(float(m.geom_friction[g1][0]), float(m.geom_friction[g2][0]), 1e-5)
            fs = max(np.abs(b['efc'][e, :ne, 0]).max(), 1e-3)
            assert _cone_kkt(b, e, mus) < 1e-7, (mk.__name__, e, _cone_kkt(b, e, mus))
            assert np.abs(c['efc'][e, :ne, 0] - b['efc'][e, :ne, 0]).max()/fs < 1e-6
            total += 1
            if _cone_kkt(a, e, mus) < 1e-6:                   # PGS converged: the same forces
                assert np.abs(a['efc'][e, :ne, 0] - b['efc'][e, :ne, 0]).max()/fs < 1e-5
            else:
                stalled += 1
        assert np.abs(c['qvel'] - b['qvel']).max() < 1e-5          # CG stops at ~1e-8 of the forces (weighted as in _agree)
    print(f'elliptic, impratio {impratio}: PGS stalled at a cone apex in {stalled} of {total} envs')
    assert total - stalled >= 6


def test_elliptic_friction_is_isotropic(oracle):
    """A ball sliding on the plane without spin: with the elliptic cone the friction force opposes the sliding velocity exactly
    and has magnitude mu f_n whatever the direction (the pyramidal cone is a square pyramid: direction-dependent)."""
    b = ModelBuilder('ball', timestep=1e-3)
    b.add_body('ball', 'world', pos=(0, 0, 0.0495), mass=0.5, inertia=(5e-4, 5e-4, 5e-4), joint='free')
    b.add_geom('ball', GEOM_SPHERE, (0.05,), friction=(0.6, 0, 0))
    b.add_geom('world', GEOM_PLANE, (0, 0, 0), friction=(0.6, 0, 0))
    b.options['max_contacts'] = 4
    m = b.compile()
    for cone, tol in (('elliptic', 1e-9), ('pyramidal', None)):
        mm = copy.copy(m); mm.cone = CONES[cone]
        ratios, angles = [], []
        for th in np.linspace(0.1, 1.4, 6):
            qpos = m.qpos0[None].copy(); qvel = np.zeros((1, m.nv)); qvel[0, 0] = 1.0*np.cos(th); qvel[0, 1] = 1.0*np.sin(th)
            o = oracle.step_tf(_with(mm, 'newton', 100, 1e-14), qpos, qvel)
            assert o['ncon'][0] == 1
            fn, f1, f2 = o['contact'][0, 0, 12:15]; fr = o['contact'][0, 0, 3:12].reshape(3, 3)
            ft = f1*fr[1] + f2*fr[2]                              # world-frame friction on the ball
            ratios.append(np.linalg.norm(ft)/fn)
            angles.append(np.arctan2(-ft[1], -ft[0]) - th)
        if cone == 'elliptic':
            assert np.abs(np.array(ratios) - 0.6).max() < tol and np.abs(angles).max() < 1e-9, (ratios, angles)
        else:
            assert np.ptp(ratios) > 0.05                          # the pyramid's friction limit depends on the direction


def test_sliding_slab_obeys_coulomb_momentum_balance(oracle):
    """A slab sliding down an incline from rest (tan(theta) > mu; the incline is a tilted gravity).  Whatever the contacts do in
    between - the four corner contacts chatter - the step is an exact impulse balance, and while sliding every contact obeys
    |f_t| = mu f_n: so the mean acceleration along the slope over a window is  g sin(theta) - mu (g cos(theta) + dv_z / T)  with
    dv_z the change of the normal velocity over the window.  Elliptic cone: exact down the slope whatever its heading in the contact
    frame (1e-6 along a frame axis; 1e-3 at 0.5 rad, where the rocking slab's corner velocities are not exactly parallel to the
    CoM's); pyramidal cone: exact only along a frame axis, 20 % off at 0.5 rad - the square pyramid is not isotropic."""
    th, mu, g = 0.5, 0.3, 9.81
    out = {}
    for yaw in (0.0, 0.5):
        b = ModelBuilder('slab', timestep=1e-3, gravity=(g*np.sin(th)*np.cos(yaw), g*np.sin(th)*np.sin(yaw), -g*np.cos(th)))
        b.add_body('slab', 'world', pos=(0, 0, 0.02), mass=1.0, inertia=(4e-3, 4e-3, 8e-3), joint='free')
        b.add_geom('slab', GEOM_BOX, (0.1, 0.1, 0.02), friction=(mu, 0, 0))
        b.add_geom('world', GEOM_PLANE, (0, 0, 0), friction=(mu, 0, 0))
        b.options['max_contacts'] = 8
        m = b.compile()
        for cone in ('elliptic', 'pyramidal'):
            mm = _with(m, 'newton', 100, 1e-10); mm.cone = CONES[cone]
            q = m.qpos0[None].copy(); v = np.zeros((1, m.nv)); w = np.zeros((1, m.nv))
            vs = []
            for t in range(1000):
                o = oracle.step_tf(mm, q, v, warmstart=w, want_AR=False)
                q, v, w = o['qpos'], o['qvel'], o['warmstart']; vs.append(v[0, :3].copy())
            a = (vs[999] - vs[199])/0.8
            along = a[0]*np.cos(yaw) + a[1]*np.sin(yaw); across = -a[0]*np.sin(yaw) + a[1]*np.cos(yaw)
            out[cone, yaw] = (along, across, g*np.sin(th) - mu*(g*np.cos(th) + a[2]))
    for (cone, yaw), (along, across, want) in out.items():
        print(f'{cone:9s} heading {yaw}: along {along:.6f} expected {want:.6f} across {across:.2e}')
    assert abs(out['elliptic', 0.0][0] - out['elliptic', 0.0][2]) < 1e-6*g and abs(out['pyramidal', 0.0][0] - out['pyramidal', 0.0][2]) < 1e-6*g
    assert abs(out['elliptic', 0.5][0] - out['elliptic', 0.5][2]) < 2e-3*out['elliptic', 0.5][2] and abs(out['elliptic', 0.5][1]) < 1e-3
    assert abs(out['pyramidal', 0.5][0] - out['pyramidal', 0.5][2]) > 0.1*out['pyramidal', 0.5][2]


def _slab_on_incline(th, mu, yaw=0.3):
    g = 9.81
    b = ModelBuilder('slab', timestep=1e-3, gravity=(g*np.sin(th)*np.cos(yaw), g*np.sin(th)*np.sin(yaw), -g*np.cos(th)))
    b.add_body('slab', 'world', pos=(0, 0, 0.02), mass=1.0, inertia=(4e-3, 4e-3, 8e-3), joint='free')
    b.add_geom('slab', GEOM_BOX, (0.1, 0.1, 0.02), friction=(mu, 0, 0))
    b.add_geom('world', GEOM_PLANE, (0, 0, 0), friction=(mu, 0, 0))
    b.options['max_contacts'] = 8
    return b.compile()


@pytest.mark.parametrize('cone', ['pyramidal', 'elliptic'])
@pytest.mark.parametrize('solver', ['pgs', 'newton'])
def test_noslip_stops_the_creep_of_a_sticking_contact(oracle, cone, solver):
    """option.noslip_iterations (reference mjcf.py:1392-1403): MuJoCo's post-pass re-solves the FRICTION forces without the
    regulariser, the normal forces keep the main solver's values.  A slab at rest on an incline it can hold (mu > tan theta) creeps
    under the soft constraint - the regulariser lets the friction rows yield - and stops creeping with noslip: for a contact that
    sticks, the unregularised rows give J_t qacc = aref_t = -B v_t, which is 0 from rest.  Checked after the slab has settled on its
    soft contacts: tangential velocity of the CoM after 300 steps with and without the post-pass, the normal forces the two runs
    carry, and - property of the pass itself - a sweep never raises the unregularised dual cost (improvement >= 0 is what ends it)."""
    th, mu = 0.3, 0.6                       # tan(0.3) = 0.31 < 0.6: sticks
    m = _with(_slab_on_incline(th, mu), solver, 100 if solver == 'newton' else 200, 1e-10)
    m.cone = CONES[cone]
    res = {}
    for ns in (0, 50):
        m.noslip_iterations = ns; m.noslip_tolerance = 1e-12
        q = m.qpos0[None].copy(); v = np.zeros((1, m.nv)); w = np.zeros((1, m.nv))
        for t in range(300):
            o = oracle.step_tf(m, q, v, warmstart=w, want_AR=False)
            q, v, w = o['qpos'], o['qvel'], o['warmstart']
        nc = int(o['ncon'][0])
        assert nc == 4 and int(o['status'][0]) == 0
        res[ns] = dict(vt=np.hypot(v[0, 0], v[0, 1]), fn=o['contact'][0, :nc, 12].sum(), q=q.copy())
    print(f"{cone} {solver}: tangential creep velocity without / with noslip {res[0]['vt']:.3e} / {res[50]['vt']:.3e}; normal force {res[0]['fn']:.6f} / {res[50]['fn']:.6f}")
    assert res[0]['vt'] > 1e-5                                   # the soft contact creeps down the slope ...
    assert res[50]['vt'] < 1e-2*res[0]['vt']                     # ... noslip holds it (measured: 300 times slower)
    assert abs(res[50]['fn'] - 9.81*np.cos(th)) < 2e-3*9.81 and abs(res[0]['fn'] - 9.81*np.cos(th)) < 2e-3*9.81      # both carry the weight's normal part


def test_noslip_keeps_normal_and_limit_forces_and_zero_iterations_is_the_identity(oracle):
    """One step of the walker (feet, bellies, a joint on its limit) with and without the post-pass: every limit-row force and every
    contact's normal force (the sum of its four pyramid rows) is what the main solver left; the friction forces move; with
    noslip_iterations = 0 nothing changes at all."""
    m = _walker()
    qs, vs = _states(m, 6, seed=4)
    base = oracle.step_tf(_with(m, 'pgs', 200, 1e-12), qs, vs, want_AR=False)
    mm = _with(m, 'pgs', 200, 1e-12); mm.noslip_iterations = 0
    same = oracle.step_tf(mm, qs, vs, want_AR=False)
    assert np.array_equal(base['qvel'], same['qvel']) and np.array_equal(base['efc'], same['efc'])
    mm.noslip_iterations = 20; mm.noslip_tolerance = 1e-12
    ns = oracle.step_tf(mm, qs, vs, want_AR=False)
    moved = 0
    for e in range(qs.shape[0]):
        ne, nc = int(base['nefc'][e]), int(base['ncon'][e])
        assert ne == int(ns['nefc'][e]) and nc == int(ns['ncon'][e])
        nlim = ne - 4*nc
        fb, fn = base['efc'][e, :ne, 0], ns['efc'][e, :ne, 0]
        assert np.array_equal(fb[:nlim], fn[:nlim])                                              # limit rows untouched
        for c in range(nc):
            a, b_ = fb[nlim + 4*c:nlim + 4*c + 4], fn[nlim + 4*c:nlim + 4*c + 4]
            assert abs(a[:2].sum() - b_[:2].sum()) < 1e-12*max(1.0, a.sum()) and abs(a[2:].sum() - b_[2:].sum()) < 1e-12*max(1.0, a.sum())
            assert (b_ >= 0).all()
            moved += int(np.abs(a - b_).max() > 1e-9)
    assert moved > 0
    assert not np.array_equal(base['qvel'], ns['qvel'])
