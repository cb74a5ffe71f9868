"""Error metrics of the parity tests.

``relerr`` is the old whole-tensor metric, max|a - b| / max|b|: for qpos its denominator is the root quaternion's 1, so it says
nothing about small components.  ``group_relerr`` is per component: |a - b| / (|b| + floor), with the floor taken per column
group (a group = columns of one physical kind: root position, root quaternion, joint angles, ...), floor = ``rel_floor`` x the
group's RMS magnitude in the reference, never below ``abs_floor``.  A component far below its group's typical size is thus held
to an absolute error of rel_floor x typical size x tol, everything else to a relative one."""
import numpy as np


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max()/max(np.abs(b).max(), 1e-12)


def group_relerr(a, b, groups=None, rel_floor=0.1, abs_floor=1e-9):
    """a, b: [..., ncol]; groups: list of column index arrays / slices (default: one group).  Returns the worst component error."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if groups is None:
        groups = [slice(None)]
    worst = 0.0
    for g in groups:
        ag, bg = a[..., g], b[..., g]
        if bg.size == 0:
            continue
        floor = max(rel_floor*float(np.sqrt(np.mean(bg*bg))), abs_floor)
        worst = max(worst, float((np.abs(ag - bg)/(np.abs(bg) + floor)).max()))
    return worst


def qpos_groups(m):
    """Root position, root quaternion, hinge / slide coordinates."""
    free = m.njnt > 0 and m.jnt_type[0] == 0
    return [slice(0, 3), slice(3, 7), slice(7, m.nq)] if free else [slice(0, m.nq)]


def qvel_groups(m):
    free = m.njnt > 0 and m.jnt_type[0] == 0
    return [slice(0, 3), slice(3, 6), slice(6, m.nv)] if free else [slice(0, m.nv)]


def link_row_groups():
    """AnimatData link rows (include/fmj.h FMJ_LINK_*): CoM position, CoM quaternion, urdf position, urdf quaternion, linear and
    angular velocity."""
    return [slice(0, 3), slice(3, 7), slice(7, 10), slice(10, 14), slice(14, 17), slice(17, 20)]


def fp32_floor_of_solve(H, rhs):
    """What storing the joint-space matrix in fp32 costs, whatever computes it: x solves H x = rhs exactly, x32 solves
    fl32(H) x32 = rhs exactly (every entry of H rounded to the nearest float, nothing else).  Returns |x32 - x| per component.
    The joint-space inertia of a long chain of light links is ill-conditioned (scaled condition numbers of 3e4 .. 2e5 for the
    models here), so this floor is 2e-5 .. 5e-4 of max|x|: no fp32 composite-rigid-body + L'DL step can promise less."""
    H = np.asarray(H, np.float64); rhs = np.asarray(rhs, np.float64)
    x = np.linalg.solve(H, rhs)
    x32 = np.linalg.solve(H.astype(np.float32).astype(np.float64), rhs)
    return np.abs(x32 - x), x


def scaled_condition(H):
    d = 1.0/np.sqrt(np.diag(H))
    return float(np.linalg.cond(H*np.outer(d, d)))


def within_floor(err, floor_err, k=6.0, abs_tol=0.0):
    """A bound stated against the yardstick instead of a fitted number: ``err`` (HIP against the fp64 oracle) may be
    ``k`` times what the fp64 oracle WITH fp32 STORAGE (oracle.fp32_storage(level): the floor of any fp32 engine) differs from the plain
    oracle by, in the same metric on the same run, plus ``abs_tol`` for quantities whose floor is exactly zero."""
    return err <= k*floor_err + abs_tol
