/* fmj_oracle.c — CPU fp64 restatement of the farms_mujoco hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (farms_mujoco_amd/) never does.
 *
 * PARITY UNPINNED versus MuJoCo: the arithmetic of the step lives in the third-party `mujoco`
 * C library, which the reference neither vendors nor pins (reference requirements.txt:1-9;
 * only call sites: simulation.py:53,83-89,156,175, sensors.pyx:70) and which is not installed
 * here; the reference holds no tests or golden vectors.  What follows restates MuJoCo's
 * published forward-dynamics pipeline (mj_step = mj_forward + Euler with implicit joint
 * damping) for the feature subset reference mjcf.py switches on, and is
 * pinned instead by analytic known-answer tests and an independent body-frame RNEA
 * (tests/test_oracle_*.py).
 *
 * The parts the reference OWNS are restated line by line:
 *   drag            <- reference farms_mujoco/swimming/drag.pyx:12-268
 *   physics2data    <- reference farms_mujoco/simulation/physics.py:449-524
 *   contacts2data   <- reference farms_mujoco/sensors/sensors.pyx:20-190 (fmjo_contacts2data; mj_contactForce restated
 *                      for the pyramidal cone)
 *
 * Conventions: spatial vectors are [rot(3); lin(3)]; all c* quantities are expressed in world
 * axes about subtree_com[body_rootid]; quaternions w,x,y,z except AnimatData rows (x,y,z,w).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* diagnostics: sweeps the last PGS solve ran (single-threaded use only) */
static int g_last_pgs_iterations = 0;
static int g_last_noslip_iterations = 0;
int fmjo_last_noslip_iterations(void) { return g_last_noslip_iterations; }
int fmjo_last_pgs_iterations(void) { return g_last_pgs_iterations; }   /* sweeps / iterations of the last solve, any solver */

#include "../include/fmj.h"

/* Test knob: round every stored entry of the joint-space matrices (M after mj_crb, H = M + h B before its factorisation) to
 * fp32, and nothing else - "an fp64 engine whose only flaw is fp32 storage of the inertia matrix".  What such a run differs
 * from the plain fp64 run by is the floor of ANY fp32 composite-rigid-body + L'DL step (the matrices are ill-conditioned:
 * scaled condition numbers 3e4 .. 2e5 for the models here); the GPU parity tests state their velocity / acceleration bounds as
 * small multiples of it instead of fitted numbers.  Read-only while a run is in flight. */
static int g_fp32_storage = 0;
/* Levels (round 5; VERDICT round 4 item 1: "the floor omits fp32 state and fp32 kinematics"):
 *   1  M and H stored in fp32 (the round-3 knob, above);
 *   2  + the STATE an fp32 engine carries from step to step - qpos, qvel, the solver's warm start - rounded to fp32 after every
 *        step, and the kinematic poses every later stage reads (xpos, xquat / xmat, xipos / ximat, joint anchors and axes) rounded
 *        to fp32 after mj_kinematics;
 *   3  + every array handed from one stage to the next: cdof, cvel, qfrc_smooth, qacc_smooth, the constraint rows (J, aref, R).
 * Arithmetic stays fp64 at every level: each level is "an fp64 engine whose only flaw is fp32 storage of those arrays" and what it
 * differs from the plain run by is a floor for any engine that holds them in fp32. */
void fmjo_set_fp32_storage(int on) { g_fp32_storage = on; }
/* fmjo_set_fp32_drop_bits(k): the rounding of the knob above keeps 24 - k mantissa bits instead of fp32's 24, i.e. 2^k times fp32's
 * rounding error in every stored value - "what if the fp32 engine's per-step error were 2^k times the storage floor" (the walking-horizon
 * study of tests/test_gpu_contacts.py::test_thousand_steps_of_walking and scripts/walk_event_diff.py). */
static int g_fp32_drop_bits = 0;
void fmjo_set_fp32_drop_bits(int k) { g_fp32_drop_bits = k < 0 ? 0 : (k > 20 ? 20 : k); }
static void round_to_f32(double* a, int n) {
  for (int i = 0; i < n; i++) {
    float f = (float)a[i];
    if (g_fp32_drop_bits && f == f && f != 0.0f && fabsf(f) < 1e30f) {
      int ex; const double mant = frexp((double)f, &ex);                 /* f = mant 2^ex, 0.5 <= |mant| < 1 */
      const double sc = ldexp(1.0, 24 - g_fp32_drop_bits);
      f = (float)ldexp(nearbyint(mant * sc) / sc, ex);
    }
    a[i] = (double)f;
  }
}

#define MINVAL 1e-15
#define MAXVAL 1e10
#define MINIMP 0.0001
#define MAXIMP 0.9999

/* ------------------------------------------------------------------------------------------ */
/* small vector / quaternion helpers (MuJoCo engine_util_blas / engine_util_spatial semantics)  */

static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double dotn(const double* a, const double* b, int n) {
  double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s;
}
static void mul_quat(double* r, const double* a, const double* b) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                 a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                 a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof t);
}
static void normalize4(double* q) {
  double n = sqrt(dotn(q, q, 4));
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
static void quat2mat(double* m, const double* q) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q[1] * q[2] - q[0] * q[3]); m[2] = 2 * (q[1] * q[3] + q[0] * q[2]);
  m[3] = 2 * (q[1] * q[2] + q[0] * q[3]); m[5] = 2 * (q[2] * q[3] - q[0] * q[1]);
  m[6] = 2 * (q[1] * q[3] - q[0] * q[2]); m[7] = 2 * (q[2] * q[3] + q[0] * q[1]);
}
static void rot_vec_quat(double* r, const double* v, const double* q) {
  double m[9]; quat2mat(m, q);
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void axis_angle2quat(double* q, const double* axis, double angle) {
  if (angle == 0) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* cinert[10] * v[6] */
static void mul_inert_vec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
/* motion cross: vel x v */
static void cross_motion(double* r, const double* vel, const double* v) {
  double t[6];
  cross3(t, vel, v);
  double a[3], b[3];
  cross3(a, vel, v + 3); cross3(b, vel + 3, v);
  t[3] = a[0] + b[0]; t[4] = a[1] + b[1]; t[5] = a[2] + b[2];
  memcpy(r, t, sizeof t);
}
/* force cross: vel x* f */
static void cross_force(double* r, const double* vel, const double* f) {
  double t[6], a[3], b[3];
  cross3(a, vel, f); cross3(b, vel + 3, f + 3);
  t[0] = a[0] + b[0]; t[1] = a[1] + b[1]; t[2] = a[2] + b[2];
  cross3(t + 3, vel, f + 3);
  memcpy(r, t, sizeof t);
}

/* ------------------------------------------------------------------------------------------ */
/* per-env workspace                                                                           */

typedef struct ws_t {
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *subtree_com, *subtree_mass;
  double *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  double *qM, *qLD, *qLDiagInv, *qH, *qHDiagInv;
  double *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_xfrc, *qfrc_smooth, *qacc_smooth;
  double *qfrc_constraint, *qacc, *qacc_warmstart, *actuator_force, *tmpv;
  /* constraints */
  int nefc, ncon, maxefc;
  double *efc_J, *efc_pos, *efc_margin, *efc_R, *efc_aref, *efc_b, *efc_AR, *efc_force, *efc_diagApprox;
  double *efc_MiJT;
  int *efc_type, *efc_id;
  /* contacts */
  int *con_geom, *con_plane, *con_pair; double *con_pos, *con_frame, *con_dist, *con_mu; int *con_efc;   /* geom2, geom1, explicit pair or -1 */
  double *jointlimitfrc;
  double meaninertia;
  int disable_actuation;      /* mj_forward with mjDSBL_ACTUATION (what dm_control runs after physics.reset, task.py:137) */
} ws_t;

enum { EFC_LIMIT = 0, EFC_CONTACT = 1, EFC_ELLIPTIC = 2 };   /* EFC_ELLIPTIC: the rows (normal, tangent 1, tangent 2) of a contact under cone = elliptic */

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

static ws_t* ws_new(const fmj_model* m) {
  ws_t* w = (ws_t*)calloc(1, sizeof(ws_t));
  int nb = m->nbody, nv = m->nv, nj = m->njnt, nM = m->nM;
  w->xpos = dalloc(3 * nb); w->xquat = dalloc(4 * nb); w->xmat = dalloc(9 * nb);
  w->xipos = dalloc(3 * nb); w->ximat = dalloc(9 * nb);
  w->xanchor = dalloc(3 * nj); w->xaxis = dalloc(3 * nj);
  w->subtree_com = dalloc(3 * nb); w->subtree_mass = dalloc(nb);
  w->cinert = dalloc(10 * nb); w->crb = dalloc(10 * nb);
  w->cdof = dalloc(6 * nv); w->cdof_dot = dalloc(6 * nv);
  w->cvel = dalloc(6 * nb); w->cacc = dalloc(6 * nb); w->cfrc = dalloc(6 * nb);
  w->qM = dalloc(nM); w->qLD = dalloc(nM); w->qLDiagInv = dalloc(nv);
  w->qH = dalloc(nM); w->qHDiagInv = dalloc(nv);
  w->qfrc_bias = dalloc(nv); w->qfrc_passive = dalloc(nv); w->qfrc_actuator = dalloc(nv);
  w->qfrc_xfrc = dalloc(nv); w->qfrc_smooth = dalloc(nv); w->qacc_smooth = dalloc(nv);
  w->qfrc_constraint = dalloc(nv); w->qacc = dalloc(nv); w->qacc_warmstart = dalloc(nv);
  w->actuator_force = dalloc(m->nu); w->tmpv = dalloc(nv);
  int maxcon = m->max_contacts > 0 ? m->max_contacts : 0;
  w->maxefc = 2 * nj + 4 * maxcon;
  int me = w->maxefc;
  w->efc_J = dalloc((size_t)me * nv); w->efc_pos = dalloc(me); w->efc_margin = dalloc(me);
  w->efc_R = dalloc(me); w->efc_aref = dalloc(me); w->efc_b = dalloc(me);
  w->efc_AR = dalloc((size_t)me * me); w->efc_force = dalloc(me); w->efc_diagApprox = dalloc(me);
  w->efc_MiJT = dalloc((size_t)me * nv);
  w->efc_type = (int*)calloc(me ? me : 1, sizeof(int)); w->efc_id = (int*)calloc(me ? me : 1, sizeof(int));
  w->con_geom = (int*)calloc(maxcon ? maxcon : 1, sizeof(int));
  w->con_efc = (int*)calloc(maxcon ? maxcon : 1, sizeof(int));
  w->con_plane = (int*)calloc(maxcon ? maxcon : 1, sizeof(int));
  w->con_pair = (int*)calloc(maxcon ? maxcon : 1, sizeof(int));
  w->con_pos = dalloc(3 * maxcon); w->con_frame = dalloc(9 * maxcon);
  w->con_dist = dalloc(maxcon); w->con_mu = dalloc(maxcon);
  w->jointlimitfrc = dalloc(nj);
  return w;
}
static void ws_free(ws_t* w) {
  double** p[] = {&w->xpos, &w->xquat, &w->xmat, &w->xipos, &w->ximat, &w->xanchor, &w->xaxis,
                  &w->subtree_com, &w->subtree_mass, &w->cinert, &w->crb, &w->cdof, &w->cdof_dot,
                  &w->cvel, &w->cacc, &w->cfrc, &w->qM, &w->qLD, &w->qLDiagInv, &w->qH, &w->qHDiagInv,
                  &w->qfrc_bias, &w->qfrc_passive, &w->qfrc_actuator, &w->qfrc_xfrc, &w->qfrc_smooth,
                  &w->qacc_smooth, &w->qfrc_constraint, &w->qacc, &w->qacc_warmstart,
                  &w->actuator_force, &w->tmpv, &w->efc_J, &w->efc_pos, &w->efc_margin, &w->efc_R,
                  &w->efc_aref, &w->efc_b, &w->efc_AR, &w->efc_force, &w->efc_diagApprox,
                  &w->efc_MiJT, &w->con_pos, &w->con_frame, &w->con_dist, &w->con_mu, &w->jointlimitfrc};
  for (size_t i = 0; i < sizeof p / sizeof p[0]; i++) free(*p[i]);
  free(w->efc_type); free(w->efc_id); free(w->con_geom); free(w->con_efc); free(w->con_plane); free(w->con_pair);
  free(w);
}

/* ------------------------------------------------------------------------------------------ */
/* position stage: mj_kinematics, mj_comPos, mj_crb, mj_factorM        */

static void kinematics(const fmj_model* m, ws_t* w, const double* qpos) {
  /* world */
  w->xpos[0] = w->xpos[1] = w->xpos[2] = 0;
  w->xquat[0] = 1; w->xquat[1] = w->xquat[2] = w->xquat[3] = 0;
  quat2mat(w->xmat, w->xquat);
  memcpy(w->xipos, w->xpos, 3 * sizeof(double)); memcpy(w->ximat, w->xmat, 9 * sizeof(double));
  for (int i = 1; i < m->nbody; i++) {
    double* xpos = w->xpos + 3 * i; double* xquat = w->xquat + 4 * i;
    int j = m->body_jntadr[i];
    int pid = m->body_parentid[i];
    if (j >= 0 && m->jnt_type[j] == FMJ_JNT_FREE) {
      int qa = m->jnt_qposadr[j];
      memcpy(xpos, qpos + qa, 3 * sizeof(double));
      memcpy(xquat, qpos + qa + 3, 4 * sizeof(double));
      normalize4(xquat);
      memcpy(w->xanchor + 3 * j, xpos, 3 * sizeof(double));
      memcpy(w->xaxis + 3 * j, m->jnt_axis + 3 * j, 3 * sizeof(double));
    } else {
      /* body frame from parent */
      double v[3];
      rot_vec_quat(v, m->body_pos + 3 * i, w->xquat + 4 * pid);
      for (int k = 0; k < 3; k++) xpos[k] = w->xpos[3 * pid + k] + v[k];
      mul_quat(xquat, w->xquat + 4 * pid, m->body_quat + 4 * i);
      if (j >= 0) {
        double* xaxis = w->xaxis + 3 * j; double* xanchor = w->xanchor + 3 * j;
        rot_vec_quat(xaxis, m->jnt_axis + 3 * j, xquat);
        rot_vec_quat(xanchor, m->jnt_pos + 3 * j, xquat);
        for (int k = 0; k < 3; k++) xanchor[k] += xpos[k];
        int qa = m->jnt_qposadr[j];
        double dq = qpos[qa] - m->qpos0[qa];
        if (m->jnt_type[j] == FMJ_JNT_SLIDE) {
          for (int k = 0; k < 3; k++) xpos[k] += xaxis[k] * dq;
        } else { /* hinge */
          double qloc[4];
          axis_angle2quat(qloc, m->jnt_axis + 3 * j, dq);
          mul_quat(xquat, xquat, qloc);
          rot_vec_quat(v, m->jnt_pos + 3 * j, xquat);
          for (int k = 0; k < 3; k++) xpos[k] = xanchor[k] - v[k];
        }
      }
    }
    normalize4(xquat);
    quat2mat(w->xmat + 9 * i, xquat);
    /* inertial frame */
    double v[3], iq[4];
    rot_vec_quat(v, m->body_ipos + 3 * i, xquat);
    for (int k = 0; k < 3; k++) w->xipos[3 * i + k] = xpos[k] + v[k];
    mul_quat(iq, xquat, m->body_iquat + 4 * i);
    quat2mat(w->ximat + 9 * i, iq);
  }
  if (g_fp32_storage >= 2) {
    round_to_f32(w->xpos, 3 * m->nbody); round_to_f32(w->xquat, 4 * m->nbody); round_to_f32(w->xmat, 9 * m->nbody);
    round_to_f32(w->xipos, 3 * m->nbody); round_to_f32(w->ximat, 9 * m->nbody);
    round_to_f32(w->xanchor, 3 * m->njnt); round_to_f32(w->xaxis, 3 * m->njnt);
  }
}

static void com_pos(const fmj_model* m, ws_t* w) {
  int nb = m->nbody;
  for (int i = 0; i < nb; i++) {
    w->subtree_mass[i] = m->body_mass[i];
    for (int k = 0; k < 3; k++) w->subtree_com[3 * i + k] = m->body_mass[i] * w->xipos[3 * i + k];
  }
  for (int i = nb - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    w->subtree_mass[p] += w->subtree_mass[i];
    for (int k = 0; k < 3; k++) w->subtree_com[3 * p + k] += w->subtree_com[3 * i + k];
  }
  for (int i = 0; i < nb; i++) {
    if (w->subtree_mass[i] < MINVAL) memcpy(w->subtree_com + 3 * i, w->xipos + 3 * i, 3 * sizeof(double));
    else for (int k = 0; k < 3; k++) w->subtree_com[3 * i + k] /= w->subtree_mass[i];
  }
  memset(w->cinert, 0, 10 * sizeof(double));
  for (int i = 1; i < nb; i++) {
    const double* mat = w->ximat + 9 * i; const double* in = m->body_inertia + 3 * i;
    double mass = m->body_mass[i], dif[3];
    for (int k = 0; k < 3; k++) dif[k] = w->xipos[3 * i + k] - w->subtree_com[3 * m->body_rootid[i] + k];
    double* r = w->cinert + 10 * i;
    /* mat * diag(in) * mat' */
    r[0] = mat[0] * mat[0] * in[0] + mat[1] * mat[1] * in[1] + mat[2] * mat[2] * in[2];
    r[1] = mat[3] * mat[3] * in[0] + mat[4] * mat[4] * in[1] + mat[5] * mat[5] * in[2];
    r[2] = mat[6] * mat[6] * in[0] + mat[7] * mat[7] * in[1] + mat[8] * mat[8] * in[2];
    r[3] = mat[0] * mat[3] * in[0] + mat[1] * mat[4] * in[1] + mat[2] * mat[5] * in[2];
    r[4] = mat[0] * mat[6] * in[0] + mat[1] * mat[7] * in[1] + mat[2] * mat[8] * in[2];
    r[5] = mat[3] * mat[6] * in[0] + mat[4] * mat[7] * in[1] + mat[5] * mat[8] * in[2];
    r[0] += mass * (dif[1] * dif[1] + dif[2] * dif[2]);
    r[1] += mass * (dif[0] * dif[0] + dif[2] * dif[2]);
    r[2] += mass * (dif[0] * dif[0] + dif[1] * dif[1]);
    r[3] -= mass * dif[0] * dif[1]; r[4] -= mass * dif[0] * dif[2]; r[5] -= mass * dif[1] * dif[2];
    r[6] = mass * dif[0]; r[7] = mass * dif[1]; r[8] = mass * dif[2]; r[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) {
    int bi = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = w->subtree_com[3 * m->body_rootid[bi] + k] - w->xanchor[3 * j + k];
    const double* axis = w->xaxis + 3 * j;
    double* cd = w->cdof + 6 * da;
    if (m->jnt_type[j] == FMJ_JNT_FREE) {
      memset(cd, 0, 18 * sizeof(double));
      for (int k = 0; k < 3; k++) cd[6 * k + 3 + k] = 1;
      for (int k = 0; k < 3; k++) {
        double* c = cd + 6 * (3 + k);
        const double* mat = w->xmat + 9 * bi;
        c[0] = mat[k]; c[1] = mat[k + 3]; c[2] = mat[k + 6];
        cross3(c + 3, c, off);
      }
    } else if (m->jnt_type[j] == FMJ_JNT_SLIDE) {
      cd[0] = cd[1] = cd[2] = 0; memcpy(cd + 3, axis, 3 * sizeof(double));
    } else {
      memcpy(cd, axis, 3 * sizeof(double)); cross3(cd + 3, axis, off);
    }
  }
}

static void crb(const fmj_model* m, ws_t* w) {
  memcpy(w->crb, w->cinert, 10 * m->nbody * sizeof(double));
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int k = 0; k < 10; k++) w->crb[10 * p + k] += w->crb[10 * i + k];
  }
  memset(w->qM, 0, m->nM * sizeof(double));
  for (int i = 0; i < m->nv; i++) {
    int adr = m->dof_Madr[i];
    double buf[6];
    mul_inert_vec(buf, w->crb + 10 * m->dof_bodyid[i], w->cdof + 6 * i);
    w->qM[adr] = m->dof_armature[i];
    for (int j = i; j >= 0; j = m->dof_parentid[j]) w->qM[adr++] += dotn(w->cdof + 6 * j, buf, 6);
  }
  if (g_fp32_storage) round_to_f32(w->qM, m->nM);
}

/* sparse L'DL in MuJoCo's dof_Madr/dof_parentid storage (mj_factorI) */
static void factor(const fmj_model* m, const double* M, double* LD, double* DiagInv) {
  int nv = m->nv, nM = m->nM;
  memcpy(LD, M, nM * sizeof(double));
  for (int k = nv - 1; k >= 0; k--) {
    int kk = m->dof_Madr[k], ki = kk + 1;
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i], ki++) {
      double t = LD[ki] / LD[kk];
      int cnt = (i < nv - 1 ? m->dof_Madr[i + 1] : nM) - m->dof_Madr[i];
      for (int c = 0; c < cnt; c++) LD[m->dof_Madr[i] + c] -= t * LD[ki + c];
      LD[ki] = t;
    }
    DiagInv[k] = 1.0 / LD[kk];
  }
}
static void solve_ld(const fmj_model* m, double* x, const double* LD, const double* DiagInv) {
  int nv = m->nv;
  for (int i = nv - 1; i >= 0; i--) {
    int a = m->dof_Madr[i] + 1;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[j] -= LD[a++] * x[i];
  }
  for (int i = 0; i < nv; i++) x[i] *= DiagInv[i];
  for (int i = 0; i < nv; i++) {
    int a = m->dof_Madr[i] + 1;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[i] -= LD[a++] * x[j];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* velocity stage: mj_comVel, mj_passive, mj_rne  (Appendix A.4, A.5)                            */

static void com_vel(const fmj_model* m, ws_t* w, const double* qvel) {
  memset(w->cvel, 0, 6 * sizeof(double));
  for (int i = 1; i < m->nbody; i++) {
    double cvel[6];
    memcpy(cvel, w->cvel + 6 * m->body_parentid[i], sizeof cvel);
    int bda = m->body_dofadr[i], j = m->body_jntadr[i];
    if (j >= 0) {
      if (m->jnt_type[j] == FMJ_JNT_FREE) {
        memset(w->cdof_dot + 6 * bda, 0, 18 * sizeof(double));
        for (int k = 0; k < 3; k++) for (int c = 0; c < 6; c++) cvel[c] += w->cdof[6 * (bda + k) + c] * qvel[bda + k];
        for (int k = 3; k < 6; k++) cross_motion(w->cdof_dot + 6 * (bda + k), cvel, w->cdof + 6 * (bda + k));
        for (int k = 3; k < 6; k++) for (int c = 0; c < 6; c++) cvel[c] += w->cdof[6 * (bda + k) + c] * qvel[bda + k];
      } else {
        cross_motion(w->cdof_dot + 6 * bda, cvel, w->cdof + 6 * bda);
        for (int c = 0; c < 6; c++) cvel[c] += w->cdof[6 * bda + c] * qvel[bda];
      }
    }
    memcpy(w->cvel + 6 * i, cvel, sizeof cvel);
  }
}

static void passive(const fmj_model* m, ws_t* w, const double* qpos, const double* qvel, const double* qpos_spring) {
  for (int i = 0; i < m->nv; i++) w->qfrc_passive[i] = -m->dof_damping[i] * qvel[i];
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_type[j] == FMJ_JNT_FREE || m->jnt_stiffness[j] == 0) continue;
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    w->qfrc_passive[da] += -m->jnt_stiffness[j] * (qpos[qa] - qpos_spring[qa]);
  }
}

static void rne(const fmj_model* m, ws_t* w, const double* qvel) {
  memset(w->cacc, 0, 6 * sizeof(double));
  for (int k = 0; k < 3; k++) w->cacc[3 + k] = -m->gravity[k];
  memset(w->cfrc, 0, 6 * sizeof(double));
  for (int i = 1; i < m->nbody; i++) {
    double* cacc = w->cacc + 6 * i;
    memcpy(cacc, w->cacc + 6 * m->body_parentid[i], 6 * sizeof(double));
    int bda = m->body_dofadr[i];
    for (int d = 0; d < m->body_dofnum[i]; d++)
      for (int c = 0; c < 6; c++) cacc[c] += w->cdof_dot[6 * (bda + d) + c] * qvel[bda + d];
    double t[6], t1[6];
    mul_inert_vec(w->cfrc + 6 * i, w->cinert + 10 * i, cacc);
    mul_inert_vec(t, w->cinert + 10 * i, w->cvel + 6 * i);
    cross_force(t1, w->cvel + 6 * i, t);
    for (int c = 0; c < 6; c++) w->cfrc[6 * i + c] += t1[c];
  }
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int c = 0; c < 6; c++) w->cfrc[6 * p + c] += w->cfrc[6 * i + c];
  }
  for (int i = 0; i < m->nv; i++) w->qfrc_bias[i] = dotn(w->cdof + 6 * i, w->cfrc + 6 * m->dof_bodyid[i], 6);
}

/* ------------------------------------------------------------------------------------------ */
/* actuation (A.6) and external force accumulation (A.7)                                        */

static void actuation(const fmj_model* m, ws_t* w, const double* qpos, const double* qvel, const double* ctrl) {
  memset(w->qfrc_actuator, 0, m->nv * sizeof(double));
  for (int a = 0; a < m->nu; a++) {
    int j = m->actuator_jntid[a];
    double c = ctrl ? ctrl[a] : 0.0;
    if (m->actuator_ctrllimited[a]) c = fmin(fmax(c, m->actuator_ctrlrange[2 * a]), m->actuator_ctrlrange[2 * a + 1]);
    double len = qpos[m->jnt_qposadr[j]], vel = qvel[m->jnt_dofadr[j]];
    const double* b = m->actuator_bias + 3 * a;
    double f = m->actuator_gain[a] * c + b[0] + b[1] * len + b[2] * vel;
    if (m->actuator_forcelimited[a]) f = fmin(fmax(f, m->actuator_forcerange[2 * a]), m->actuator_forcerange[2 * a + 1]);
    if (w->disable_actuation) f = 0.0;
    w->actuator_force[a] = f;
    w->qfrc_actuator[m->jnt_dofadr[j]] += f;
  }
}

/* J' * [force; torque] applied at `point` on `body` (mj_applyFT via mj_jac) */
static void apply_ft(const fmj_model* m, const ws_t* w, const double* force, const double* torque,
                     const double* point, int body, double* qfrc) {
  double off[3];
  for (int k = 0; k < 3; k++) off[k] = point[k] - w->subtree_com[3 * m->body_rootid[body] + k];
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double* cd = w->cdof + 6 * i;
    double jp[3];
    cross3(jp, cd, off);
    for (int k = 0; k < 3; k++) jp[k] += cd[3 + k];
    qfrc[i] += dotn(jp, force, 3) + dotn(cd, torque, 3);
  }
}
/* dense Jacobian rows (translational jacp[3,nv], rotational jacr[3,nv]) of a world point on body */
static void jac_point(const fmj_model* m, const ws_t* w, double* jacp, double* jacr, const double* point, int body) {
  int nv = m->nv;
  if (jacp) memset(jacp, 0, 3 * nv * sizeof(double));
  if (jacr) memset(jacr, 0, 3 * nv * sizeof(double));
  double off[3];
  for (int k = 0; k < 3; k++) off[k] = point[k] - w->subtree_com[3 * m->body_rootid[body] + k];
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double* cd = w->cdof + 6 * i;
    double jp[3];
    cross3(jp, cd, off);
    for (int k = 0; k < 3; k++) {
      if (jacp) jacp[k * nv + i] = jp[k] + cd[3 + k];
      if (jacr) jacr[k * nv + i] = cd[k];
    }
  }
}

static void xfrc_accumulate(const fmj_model* m, ws_t* w, const double* xfrc) {
  memset(w->qfrc_xfrc, 0, m->nv * sizeof(double));
  if (!xfrc) return;
  for (int i = 1; i < m->nbody; i++) {
    const double* f = xfrc + 6 * i;
    int nz = 0; for (int k = 0; k < 6; k++) nz |= (f[k] != 0);
    if (nz) apply_ft(m, w, f, f + 3, w->xipos + 3 * i, i, w->qfrc_xfrc);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* constraints: joint limits + plane contacts, pyramidal cone, PGS (Appendix A.9, A.10, E)       */

static void get_impedance(const double* solimp_in, double pos, double margin, double* imp) {
  double dmin = fmin(fmax(solimp_in[0], MINIMP), MAXIMP), dmax = fmin(fmax(solimp_in[1], MINIMP), MAXIMP);
  double width = fmax(0.0, solimp_in[2]);
  double mid = fmin(fmax(solimp_in[3], MINIMP), MAXIMP), power = fmax(1.0, solimp_in[4]);
  if (dmin == dmax || width <= MINVAL) { *imp = 0.5 * (dmin + dmax); return; }
  double x = fabs((pos - margin) / width), y;
  if (x >= 1) y = 1;
  else if (x <= 0) y = 0;
  else if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  *imp = dmin + y * (dmax - dmin);
}

static int add_efc(const fmj_model* m, ws_t* w, const double* jrow, double pos, double margin, int type, int id) {
  if (w->nefc >= w->maxefc) return -1;
  int e = w->nefc++;
  memcpy(w->efc_J + (size_t)e * m->nv, jrow, m->nv * sizeof(double));
  w->efc_pos[e] = pos; w->efc_margin[e] = margin; w->efc_type[e] = type; w->efc_id[e] = id;
  return e;
}

/* plane narrow phase: geom 'g' (sphere/capsule/box) vs the plane geom 'p' */
static void add_contact(const fmj_model* m, ws_t* w, int p, int g, const double* pos, const double* n, double dist, double mu, int* warn) {
  if (w->ncon >= m->max_contacts) { *warn |= FMJ_WARN_CONTACTFULL; return; }
  int c = w->ncon++;
  w->con_geom[c] = g; w->con_plane[c] = p; w->con_pair[c] = -1; w->con_dist[c] = dist; w->con_mu[c] = mu;
  memcpy(w->con_pos + 3 * c, pos, 3 * sizeof(double));
  /* frame: x = normal, y/z = tangents (mju_makeFrame) */
  double* f = w->con_frame + 9 * c;
  memcpy(f, n, 3 * sizeof(double));
  double t[3] = {0, 0, 0};
  if (f[1] < -0.5 || f[1] > 0.5) t[2] = 1; else t[1] = 1;
  double d = dotn(t, f, 3);
  for (int k = 0; k < 3; k++) t[k] -= d * f[k];
  double nn = sqrt(dotn(t, t, 3));
  for (int k = 0; k < 3; k++) f[3 + k] = t[k] / nn;
  cross3(f + 6, f, f + 3);
}

/* Height of world point pt above ground geom p (plane or heightfield) along the local surface normal n.
 * Heightfield (MuJoCo hfield semantics, see include/fmj.h): the plane of the grid triangle under the point, cells split
 * along the diagonal (c, r) - (c + 1, r + 1); nothing outside the grid. */
static double ground_dist(const fmj_model* m, const ws_t* w, int p, const double* pt, double* n) {
  int pb = m->geom_bodyid[p];
  double pq[4], ppos[3], v[3], pm[9];
  mul_quat(pq, w->xquat + 4 * pb, m->geom_quat + 4 * p);
  rot_vec_quat(v, m->geom_pos + 3 * p, w->xquat + 4 * pb);
  for (int k = 0; k < 3; k++) ppos[k] = w->xpos[3 * pb + k] + v[k];
  quat2mat(pm, pq);
  double dif[3]; for (int k = 0; k < 3; k++) dif[k] = pt[k] - ppos[k];
  if (m->geom_type[p] == FMJ_GEOM_PLANE) {
    n[0] = pm[2]; n[1] = pm[5]; n[2] = pm[8];
    return dotn(dif, n, 3);
  }
  /* heightfield: local coordinates of the point */
  double lx = pm[0] * dif[0] + pm[3] * dif[1] + pm[6] * dif[2], ly = pm[1] * dif[0] + pm[4] * dif[1] + pm[7] * dif[2],
         lz = pm[2] * dif[0] + pm[5] * dif[1] + pm[8] * dif[2];
  int nc = m->hfield_ncol, nr = m->hfield_nrow;
  double rx = m->hfield_size[0], ry = m->hfield_size[1], zt = m->hfield_size[2];
  double sx = (nc - 1) / (2 * rx), sy = (nr - 1) / (2 * ry);
  double gx = (lx + rx) * sx, gy = (ly + ry) * sy;
  n[0] = pm[2]; n[1] = pm[5]; n[2] = pm[8];
  if (!(gx >= 0 && gx <= nc - 1 && gy >= 0 && gy <= nr - 1)) return 1e30;
  int c = (int)gx, r = (int)gy; if (c > nc - 2) c = nc - 2; if (r > nr - 2) r = nr - 2;
  double fx = gx - c, fy = gy - r;
  const double* D = m->hfield_data + (size_t)r * nc + c;
  double z00 = D[0] * zt, z10 = D[1] * zt, z01 = D[nc] * zt, z11 = D[nc + 1] * zt, gxs, gys;
  if (fx >= fy) { gxs = z10 - z00; gys = z11 - z10; } else { gxs = z11 - z01; gys = z01 - z00; }
  double zs = z00 + gxs * fx + gys * fy;
  double nl[3] = {-gxs * sx, -gys * sy, 1.0}, inv = 1.0 / sqrt(dotn(nl, nl, 3));
  for (int k = 0; k < 3; k++) nl[k] *= inv;
  for (int k = 0; k < 3; k++) n[k] = pm[3 * k] * nl[0] + pm[3 * k + 1] * nl[1] + pm[3 * k + 2] * nl[2];
  return (lz - zs) * inv;
}

/* ground narrow phase: geom 'g' (sphere / capsule / cylinder / box) vs the ground geom 'p' (plane or heightfield) */
static void collide_plane(const fmj_model* m, ws_t* w, int p, int g, int* warn) {
  double v[3], n[3];
  /* geom pose */
  int gb = m->geom_bodyid[g];
  double gq[4], gpos[3], gm[9];
  mul_quat(gq, w->xquat + 4 * gb, m->geom_quat + 4 * g);
  rot_vec_quat(v, m->geom_pos + 3 * g, w->xquat + 4 * gb);
  for (int k = 0; k < 3; k++) gpos[k] = w->xpos[3 * gb + k] + v[k];
  quat2mat(gm, gq);
  double mu = fmax(fmax(m->geom_friction[3 * p], m->geom_friction[3 * g]), 1e-5);      /* max over the two geoms (mj_contactParam), floor mjMINMU */
  const double* size = m->geom_size + 3 * g;
  int type = m->geom_type[g];
  if (type == FMJ_GEOM_SPHERE || type == FMJ_GEOM_CAPSULE) {
    int nseg = type == FMJ_GEOM_CAPSULE ? 2 : 1;
    for (int s = 0; s < nseg; s++) {
      double c[3];
      double sgn = nseg == 2 ? (s == 0 ? 1.0 : -1.0) : 0.0;
      for (int k = 0; k < 3; k++) c[k] = gpos[k] + sgn * size[1] * gm[3 * k + 2];
      double dist = ground_dist(m, w, p, c, n) - size[0];
      if (dist < 0) {  /* margin = 0 (mjcf.py:253) */
        double pos[3];
        for (int k = 0; k < 3; k++) pos[k] = c[k] - n[k] * (size[0] + 0.5 * dist);
        add_contact(m, w, p, g, pos, n, dist, mu, warn);
      }
    }
  } else if (type == FMJ_GEOM_CYLINDER) {
    /* plane - cylinder, MuJoCo's published rim-point construction (mjc_PlaneCylinder; restated, not compiled from
     * MuJoCo): the rim point deepest into the plane on the near disk, the same point on the far disk, then two points
     * at +-120 degrees on the near rim.  margin = 0 (reference mjcf.py:253).  A heightfield is taken as the plane under
     * the cylinder's centre. */
    double axis[3] = {gm[2], gm[5], gm[8]}, vec[3], vec1[3];
    double dist = ground_dist(m, w, p, gpos, n);
    double prjaxis = dotn(n, axis, 3);
    if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }
    for (int k = 0; k < 3; k++) vec[k] = axis[k] * prjaxis - n[k];
    double len2 = dotn(vec, vec, 3);
    if (len2 >= 1e-30) { double sc = size[0] / sqrt(len2); for (int k = 0; k < 3; k++) vec[k] *= sc; }
    else { for (int k = 0; k < 3; k++) vec[k] = gm[3 * k] * size[0]; }
    double prjvec = dotn(vec, n, 3);
    for (int k = 0; k < 3; k++) axis[k] *= size[1];
    prjaxis *= size[1];
    double d0 = dist + prjaxis + prjvec;
    if (d0 < 0) {
      double pos[3];
      for (int k = 0; k < 3; k++) pos[k] = gpos[k] + vec[k] + axis[k] - n[k] * 0.5 * d0;
      add_contact(m, w, p, g, pos, n, d0, mu, warn);
      double d1 = dist - prjaxis + prjvec;
      if (d1 < 0) {
        for (int k = 0; k < 3; k++) pos[k] = gpos[k] + vec[k] - axis[k] - n[k] * 0.5 * d1;
        add_contact(m, w, p, g, pos, n, d1, mu, warn);
      }
      vec1[0] = vec[1] * axis[2] - vec[2] * axis[1]; vec1[1] = vec[2] * axis[0] - vec[0] * axis[2]; vec1[2] = vec[0] * axis[1] - vec[1] * axis[0];
      double l1 = sqrt(dotn(vec1, vec1, 3));
      if (l1 > 1e-15) for (int k = 0; k < 3; k++) vec1[k] *= size[0] * 0.8660254037844386 / l1;
      double prjvec1 = dotn(vec1, n, 3);
      for (int sgn = 1; sgn >= -1; sgn -= 2) {
        double d2 = dist + prjaxis - 0.5 * prjvec + sgn * prjvec1;
        if (d2 < 0) {
          for (int k = 0; k < 3; k++) pos[k] = gpos[k] + sgn * vec1[k] + axis[k] - 0.5 * vec[k] - n[k] * 0.5 * d2;
          add_contact(m, w, p, g, pos, n, d2, mu, warn);
        }
      }
    }
  } else if (type == FMJ_GEOM_BOX) {
    int cnt = 0;
    for (int corner = 0; corner < 8 && cnt < 4; corner++) {
      double loc[3] = {(corner & 1 ? 1 : -1) * size[0], (corner & 2 ? 1 : -1) * size[1], (corner & 4 ? 1 : -1) * size[2]};
      double c[3];
      for (int k = 0; k < 3; k++) c[k] = gpos[k] + gm[3 * k] * loc[0] + gm[3 * k + 1] * loc[1] + gm[3 * k + 2] * loc[2];
      double dist = ground_dist(m, w, p, c, n);
      if (dist < 0) {
        double pos[3];
        for (int k = 0; k < 3; k++) pos[k] = c[k] - n[k] * 0.5 * dist;
        add_contact(m, w, p, g, pos, n, dist, mu, warn);
        cnt++;
      }
    }
  } else if (type == FMJ_GEOM_MESH) {
    /* convex mesh (include/fmj.h): up to 4 contacts at the deepest penetrating vertices, deepest first, equal depths in
     * vertex order; nothing when the ground under the geom's origin is farther than the bounding radius (size[2]) */
    double dc = ground_dist(m, w, p, gpos, n);
    if (dc < size[2]) {
      int cnt = 0; double dq[4], cq[4][3], nq[4][3];
      const double* V = m->mesh_vert + 3 * (size_t)m->geom_vertadr[g];
      for (int vtx = 0; vtx < m->geom_vertnum[g]; vtx++) {
        double c[3], nn[3];
        for (int k = 0; k < 3; k++) c[k] = gpos[k] + gm[3 * k] * V[3 * vtx] + gm[3 * k + 1] * V[3 * vtx + 1] + gm[3 * k + 2] * V[3 * vtx + 2];
        double td = ground_dist(m, w, p, c, nn);
        if (!(td < 0)) continue;
        int have = 1;
        for (int k = 0; k < 4 && have; k++) {
          int empty = k >= cnt;
          if (empty || td < dq[k]) {            /* carry the displaced entry down; a filled empty slot ends the walk */
            double sd = dq[k], sc[3], sn[3];
            memcpy(sc, cq[k], sizeof sc); memcpy(sn, nq[k], sizeof sn);
            dq[k] = td; memcpy(cq[k], c, sizeof sc); memcpy(nq[k], nn, sizeof sn);
            td = sd; memcpy(c, sc, sizeof sc); memcpy(nn, sn, sizeof sn);
            if (empty) have = 0;
          }
        }
        if (cnt < 4) cnt++;
      }
      for (int k = 0; k < cnt; k++) {
        double pos[3];
        for (int j = 0; j < 3; j++) pos[j] = cq[k][j] - nq[k][j] * 0.5 * dq[k];
        add_contact(m, w, p, g, pos, nq[k], dq[k], mu, warn);
      }
    }
  }
}

/* closest points of two segments c1 +- h1 a1 and c2 +- h2 a2 (unit axes): parameters s, t along the axes (the classic
 * clamped solution; parallel segments take the midpoint of their overlap) */
static void segment_closest(const double* c1, const double* a1, double h1, const double* c2, const double* a2, double h2,
                            double* s_out, double* t_out) {
  double d[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
  double b = dotn(a1, a2, 3), da1 = dotn(d, a1, 3), da2 = dotn(d, a2, 3);
  double det = 1.0 - b * b, s, t;
  if (det > 1e-9) {
    s = (b * da2 - da1) / det;
    s = fmin(fmax(s, -h1), h1);
  } else {                                  /* parallel: centre of the overlap of the two parameter intervals */
    double lo = fmax(-h1, -da1 - h2), hi = fmin(h1, -da1 + h2);       /* c2's interval seen on axis 1: centre -da1 */
    s = lo <= hi ? 0.5 * (lo + hi) : (fabs(lo - h1) < fabs(hi + h1) ? h1 : -h1);
    s = fmin(fmax(s, -h1), h1);
  }
  t = b * s + da2; t = fmin(fmax(t, -h2), h2);
  s = b * t - da1; s = fmin(fmax(s, -h1), h1);
  *s_out = s; *t_out = t;
}

/* ---- polytopes (box, cylinder, convex mesh) in explicit pairs: include/fmj.h, ABI 6 ------------------------------------- */
static int geom_is_round(int t) { return t == FMJ_GEOM_SPHERE || t == FMJ_GEOM_CAPSULE; }
static int poly_nvert(const fmj_model* m, int g) {
  int t = m->geom_type[g];
  return t == FMJ_GEOM_BOX ? 8 : t == FMJ_GEOM_CYLINDER ? 24 : t == FMJ_GEOM_MESH ? m->geom_vertnum[g] : 0;
}
/* vertex k of polytope g in the geom frame */
static void poly_vertex(const fmj_model* m, int g, int k, double* v) {
  const double* sz = m->geom_size + 3 * g;
  int t = m->geom_type[g];
  if (t == FMJ_GEOM_BOX) { v[0] = (k & 1 ? 1 : -1) * sz[0]; v[1] = (k & 2 ? 1 : -1) * sz[1]; v[2] = (k & 4 ? 1 : -1) * sz[2]; }
  else if (t == FMJ_GEOM_CYLINDER) {            /* 12 points on each rim in steps of 150 degrees (0, 150, 300, 90, ...: equal-depth ties keep
                                                   spread-out points), the first on +x; k < 12: the +z rim */
    double a = (((k % 12) * 5) % 12) * (M_PI / 6.0);
    v[0] = sz[0] * cos(a); v[1] = sz[0] * sin(a); v[2] = k < 12 ? sz[1] : -sz[1];
  } else memcpy(v, m->mesh_vert + 3 * ((size_t)m->geom_vertadr[g] + k), 3 * sizeof(double));
}
/* signed distance of the point x (geom frame) to polytope g = max over its faces of (n . x - d), and that face's outward normal */
static double poly_signed(const fmj_model* m, int g, const double* x, double* n) {
  const double* sz = m->geom_size + 3 * g;
  int t = m->geom_type[g];
  double s = -1e300;
  n[0] = n[1] = 0; n[2] = 1;
  if (t == FMJ_GEOM_BOX) {
    for (int k = 0; k < 3; k++) {
      double sk = fabs(x[k]) - sz[k];
      if (sk > s) { s = sk; n[0] = n[1] = n[2] = 0; n[k] = x[k] < 0 ? -1 : 1; }
    }
  } else if (t == FMJ_GEOM_CYLINDER) {
    double r = sqrt(x[0] * x[0] + x[1] * x[1]);
    s = fabs(x[2]) - sz[1]; n[0] = n[1] = 0; n[2] = x[2] < 0 ? -1 : 1;
    if (r - sz[0] > s) { s = r - sz[0]; if (r > MINVAL) { n[0] = x[0] / r; n[1] = x[1] / r; } else { n[0] = 1; n[1] = 0; } n[2] = 0; }
  } else {
    const double* F = m->mesh_face + 4 * (size_t)m->geom_faceadr[g];
    for (int f = 0; f < m->geom_facenum[g]; f++) {
      double sf = F[4 * f] * x[0] + F[4 * f + 1] * x[1] + F[4 * f + 2] * x[2] - F[4 * f + 3];
      if (sf > s) { s = sf; n[0] = F[4 * f]; n[1] = F[4 * f + 1]; n[2] = F[4 * f + 2]; }
    }
  }
  return s;
}
static double geom_rbound(const fmj_model* m, int g) {
  const double* sz = m->geom_size + 3 * g;
  switch (m->geom_type[g]) {
    case FMJ_GEOM_SPHERE: return sz[0];
    case FMJ_GEOM_CAPSULE: return sz[0] + sz[1];
    case FMJ_GEOM_CYLINDER: return sqrt(sz[0] * sz[0] + sz[1] * sz[1]);
    case FMJ_GEOM_BOX: return sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]);
    default: return sz[2];          /* mesh: its bounding radius about the geom origin */
  }
}
static void geom_pose(const fmj_model* m, const ws_t* w, int g, double* pos, double* mat) {
  int gb = m->geom_bodyid[g];
  double gq[4], v[3];
  mul_quat(gq, w->xquat + 4 * gb, m->geom_quat + 4 * g);
  rot_vec_quat(v, m->geom_pos + 3 * g, w->xquat + 4 * gb);
  for (int i = 0; i < 3; i++) pos[i] = w->xpos[3 * gb + i] + v[i];
  quat2mat(mat, gq);
}
static void to_local(const double* pos, const double* mat, const double* xw, double* xl) {
  double d[3] = {xw[0] - pos[0], xw[1] - pos[1], xw[2] - pos[2]};
  for (int k = 0; k < 3; k++) xl[k] = mat[k] * d[0] + mat[3 + k] * d[1] + mat[6 + k] * d[2];
}
static void to_world_dir(const double* mat, const double* vl, double* vw) {
  for (int k = 0; k < 3; k++) vw[k] = mat[3 * k] * vl[0] + mat[3 * k + 1] * vl[1] + mat[3 * k + 2] * vl[2];
}
static void pair_emit(const fmj_model* m, ws_t* w, int pr, int g1, int g2, const double* pos, const double* n, double dist, int* warn) {
  double mu = fmax(m->pair_friction[pr], 1e-5);        /* mjMINMU */
  if (w->ncon >= m->max_contacts) { *warn |= FMJ_WARN_CONTACTFULL; return; }
  add_contact(m, w, g1, g2, pos, n, dist, mu, warn);
  w->con_pair[w->ncon - 1] = pr;
}

/* explicit pair 'pr' (geom1, geom2), normal from geom1 to geom2 (mjContact convention), margin 0.  Sphere / capsule pairs: one
 * contact at the closest points of their segments, position midway between the surfaces.  Pairs with a box, a cylinder or a convex
 * mesh: include/fmj.h (ABI 6). */
static void collide_pair(const fmj_model* m, ws_t* w, int pr, int* warn) {
  int g[2] = {m->pair_geom1[pr], m->pair_geom2[pr]};
  const int round0 = geom_is_round(m->geom_type[g[0]]), round1 = geom_is_round(m->geom_type[g[1]]);
  if (!(round0 && round1)) {
    double pos[2][3], mat[2][9];
    geom_pose(m, w, g[0], pos[0], mat[0]); geom_pose(m, w, g[1], pos[1], mat[1]);
    double dc[3] = {pos[1][0] - pos[0][0], pos[1][1] - pos[0][1], pos[1][2] - pos[0][2]};
    if (sqrt(dotn(dc, dc, 3)) > geom_rbound(m, g[0]) + geom_rbound(m, g[1])) return;
    if (round0 != round1) {                       /* polytope against sphere / capsule: the round geom's centres against the faces */
      const int rk = round0 ? 0 : 1, pk = 1 - rk;
      const double rad = m->geom_size[3 * g[rk]];
      const int ncen = m->geom_type[g[rk]] == FMJ_GEOM_CAPSULE ? 2 : 1;
      for (int c = 0; c < ncen; c++) {
        const double sgn = ncen == 2 ? (c == 0 ? 1.0 : -1.0) : 0.0, half = m->geom_size[3 * g[rk] + 1];
        double cw[3], cl[3], nl[3], nw[3];
        for (int k = 0; k < 3; k++) cw[k] = pos[rk][k] + sgn * half * mat[rk][3 * k + 2];
        to_local(pos[pk], mat[pk], cw, cl);
        const double dist = poly_signed(m, g[pk], cl, nl) - rad;
        if (!(dist < 0)) continue;
        to_world_dir(mat[pk], nl, nw);            /* out of the polytope, towards the round geom */
        double cp[3], n12[3];
        for (int k = 0; k < 3; k++) { cp[k] = cw[k] - nw[k] * (rad + 0.5 * dist); n12[k] = pk == 0 ? nw[k] : -nw[k]; }
        pair_emit(m, w, pr, g[0], g[1], cp, n12, dist, warn);
      }
      return;
    }
    /* polytope against polytope: vertices of one inside the other; the four deepest, deepest first */
    int cnt = 0; double dq[4], cq[4][3], nq[4][3];
    for (int side = 0; side < 2; side++) {        /* side 0: geom1's vertices in geom2; side 1: geom2's vertices in geom1 */
      const int va = side, fb = 1 - side;
      const int nvert = poly_nvert(m, g[va]);
      for (int k = 0; k < nvert; k++) {
        double vl[3], vw[3], xl[3], nl[3], nw[3];
        poly_vertex(m, g[va], k, vl);
        to_world_dir(mat[va], vl, vw);
        for (int i = 0; i < 3; i++) vw[i] += pos[va][i];
        to_local(pos[fb], mat[fb], vw, xl);
        double td = poly_signed(m, g[fb], xl, nl);
        if (!(td < 0)) continue;
        to_world_dir(mat[fb], nl, nw);            /* out of the polytope the vertex is inside of */
        double c[3], nn[3];
        for (int i = 0; i < 3; i++) { c[i] = vw[i] - nw[i] * 0.5 * td; nn[i] = fb == 0 ? nw[i] : -nw[i]; }
        int have = 1;
        for (int q = 0; q < 4 && have; q++) {
          int empty = q >= cnt;
          if (empty || td < dq[q]) {
            double sd = dq[q], sc[3], sn[3];
            memcpy(sc, cq[q], sizeof sc); memcpy(sn, nq[q], sizeof sn);
            dq[q] = td; memcpy(cq[q], c, sizeof sc); memcpy(nq[q], nn, sizeof sn);
            td = sd; memcpy(c, sc, sizeof sc); memcpy(nn, sn, sizeof sn);
            if (empty) have = 0;
          }
        }
        if (cnt < 4) cnt++;
      }
    }
    for (int q = 0; q < cnt; q++) pair_emit(m, w, pr, g[0], g[1], cq[q], nq[q], dq[q], warn);
    return;
  }
  double cen[2][3], ax[2][3], half[2], rad[2];
  for (int k = 0; k < 2; k++) {
    int gb = m->geom_bodyid[g[k]];
    double gq[4], gm[9], v[3];
    mul_quat(gq, w->xquat + 4 * gb, m->geom_quat + 4 * g[k]);
    rot_vec_quat(v, m->geom_pos + 3 * g[k], w->xquat + 4 * gb);
    for (int i = 0; i < 3; i++) cen[k][i] = w->xpos[3 * gb + i] + v[i];
    quat2mat(gm, gq);
    for (int i = 0; i < 3; i++) ax[k][i] = gm[3 * i + 2];
    rad[k] = m->geom_size[3 * g[k]];
    half[k] = m->geom_type[g[k]] == FMJ_GEOM_CAPSULE ? m->geom_size[3 * g[k] + 1] : 0.0;
  }
  double s, t;
  segment_closest(cen[0], ax[0], half[0], cen[1], ax[1], half[1], &s, &t);
  double p1[3], p2[3], n[3];
  for (int i = 0; i < 3; i++) { p1[i] = cen[0][i] + s * ax[0][i]; p2[i] = cen[1][i] + t * ax[1][i]; n[i] = p2[i] - p1[i]; }
  double len = sqrt(dotn(n, n, 3));
  double dist = len - rad[0] - rad[1];
  if (!(dist < 0)) return;                 /* margin = 0 */
  if (len < MINVAL) { n[0] = 0; n[1] = 0; n[2] = 1; } else for (int i = 0; i < 3; i++) n[i] /= len;
  double pos[3];
  for (int i = 0; i < 3; i++) pos[i] = p1[i] + n[i] * (rad[0] + 0.5 * dist);
  pair_emit(m, w, pr, g[0], g[1], pos, n, dist, warn);
}

static void make_constraints(const fmj_model* m, ws_t* w, const double* qpos, const double* qvel, int* warn) {
  int nv = m->nv;
  w->nefc = 0; w->ncon = 0;
  double* jrow = w->tmpv;
  /* joint limits (mj_instantiateLimit) */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] == FMJ_JNT_FREE) continue;
    double value = qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - value);
      if (dist < margin) {
        memset(jrow, 0, nv * sizeof(double));
        jrow[m->jnt_dofadr[j]] = -side;
        int e = add_efc(m, w, jrow, dist, margin, EFC_LIMIT, j);
        if (e >= 0) w->efc_diagApprox[e] = m->dof_invweight0[m->jnt_dofadr[j]];
      }
    }
  }
  /* contacts: every animat geom against every ground geom, plane or heightfield (arena vs animat, mjcf.py:251-267) */
  for (int p = 0; p < m->ngeom; p++) {
    if (m->geom_type[p] != FMJ_GEOM_PLANE && m->geom_type[p] != FMJ_GEOM_HFIELD) continue;
    for (int g = 0; g < m->ngeom; g++) if (m->geom_type[g] != FMJ_GEOM_PLANE && m->geom_type[g] != FMJ_GEOM_HFIELD) collide_plane(m, w, p, g, warn);
  }
  for (int pr = 0; pr < m->npair; pr++) collide_pair(m, w, pr, warn);      /* after the ground contacts */
  double* jacp = dalloc(3 * nv);
  double* jacp1 = dalloc(3 * nv);
  for (int c = 0; c < w->ncon; c++) {
    int g = w->con_geom[c], b = m->geom_bodyid[g], b1 = m->geom_bodyid[w->con_plane[c]];
    jac_point(m, w, jacp, NULL, w->con_pos + 3 * c, b);
    if (b1 > 0) {                          /* jacdif = J(body2) - J(body1): both bodies move in a self-collision */
      jac_point(m, w, jacp1, NULL, w->con_pos + 3 * c, b1);
      for (int i = 0; i < 3 * nv; i++) jacp[i] -= jacp1[i];
    }
    /* contact-frame Jacobian of (geom body - plane body); plane is static => J = -(-J_b)?
       MuJoCo: jacdif = J(body2) - J(body1), geom1 = plane, geom2 = animat geom. */
    const double* f = w->con_frame + 9 * c;
    double jn[64 * 4], *jt1, *jt2;  /* nv <= 256 guard below */
    double* buf = nv <= 64 ? jn : dalloc(3 * nv);
    jt1 = buf + nv; jt2 = buf + 2 * nv;
    for (int i = 0; i < nv; i++) {
      buf[i] = f[0] * jacp[i] + f[1] * jacp[nv + i] + f[2] * jacp[2 * nv + i];
      jt1[i] = f[3] * jacp[i] + f[4] * jacp[nv + i] + f[5] * jacp[2 * nv + i];
      jt2[i] = f[6] * jacp[i] + f[7] * jacp[nv + i] + f[8] * jacp[2 * nv + i];
    }
    double mu = w->con_mu[c];
    double tran = m->body_invweight0[2 * b] + m->body_invweight0[2 * m->geom_bodyid[w->con_plane[c]]];
    w->con_efc[c] = w->nefc;
    if (m->cone == FMJ_CONE_ELLIPTIC) {
      /* mj_instantiateContact, elliptic cone, condim 3: one row per axis of the contact frame; only the normal row has a
       * position (dist, margin 0), the friction rows have pos = margin = 0 */
      for (int r = 0; r < 3; r++) {
        int e = add_efc(m, w, r == 0 ? buf : (r == 1 ? jt1 : jt2), r == 0 ? w->con_dist[c] : 0.0, 0.0, EFC_ELLIPTIC, c);
        if (e >= 0) w->efc_diagApprox[e] = tran;
      }
    } else
    for (int r = 0; r < 4; r++) {
      const double* jt = r < 2 ? jt1 : jt2;
      double sgn = (r & 1) ? -1.0 : 1.0;
      for (int i = 0; i < nv; i++) jrow[i] = buf[i] + sgn * mu * jt[i];
      int e = add_efc(m, w, jrow, w->con_dist[c], 0.0, EFC_CONTACT, c);
      if (e >= 0) w->efc_diagApprox[e] = tran + mu * mu * tran;
    }
    if (buf != jn) free(buf);
  }
  free(jacp); free(jacp1);
  /* impedance, R, aref (mj_makeImpedance, mj_referenceConstraint) */
  for (int e = 0; e < w->nefc; e++) {
    const double *solref, *solimp;
    if (w->efc_type[e] == EFC_LIMIT) { solref = m->jnt_solref + 2 * w->efc_id[e]; solimp = m->jnt_solimp + 5 * w->efc_id[e]; }
    else if (w->con_pair[w->efc_id[e]] >= 0) { int pr = w->con_pair[w->efc_id[e]]; solref = m->pair_solref + 2 * pr; solimp = m->pair_solimp + 5 * pr; }
    else { int g = w->con_geom[w->efc_id[e]]; solref = m->geom_solref + 2 * g; solimp = m->geom_solimp + 5 * g; }
    double imp; get_impedance(solimp, w->efc_pos[e], w->efc_margin[e], &imp);
    double dmax = fmin(fmax(solimp[1], MINIMP), MAXIMP);
    double K, B;
    if (solref[0] > 0) {
      double tc = fmax(solref[0], 2 * m->timestep), dr = solref[1];
      K = 1.0 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
      B = 2.0 / fmax(MINVAL, dmax * tc);
    } else { K = -solref[0] / fmax(MINVAL, dmax * dmax); B = -solref[1] / fmax(MINVAL, dmax); }
    if (w->efc_type[e] == EFC_ELLIPTIC && e > w->con_efc[w->efc_id[e]]) K = 0;   /* friction rows have no position term (mj_makeImpedance) */
    w->efc_R[e] = fmax(MINVAL, (1 - imp) * w->efc_diagApprox[e] / imp);
    double vel = dotn(w->efc_J + (size_t)e * nv, qvel, nv);
    w->efc_aref[e] = -B * vel - K * imp * (w->efc_pos[e] - w->efc_margin[e]);
  }
  /* pyramidal: all rows of a contact share R = 2 mu^2 R_first, mu scaled by 1/sqrt(impratio) */
  for (int c = 0; c < w->ncon; c++) {
    int e0 = w->con_efc[c];
    if (m->cone == FMJ_CONE_ELLIPTIC) {
      /* elliptic: R of friction dimension j = R_normal mu^2 / friction_j^2 with mu = friction_0 / sqrt(impratio): in the
       * coordinates U_0 = mu jar_0, U_j = friction_j jar_j the regulariser is isotropic and the friction cone is circular */
      if (e0 + 3 > w->nefc) continue;
      double mu = w->con_mu[c] / sqrt(m->impratio > 0 ? m->impratio : 1.0), fr = w->con_mu[c];
      for (int r = 1; r < 3; r++) w->efc_R[e0 + r] = fmax(MINVAL, w->efc_R[e0] * mu * mu / (fr * fr));
      continue;
    }
    if (e0 + 4 > w->nefc) continue;
    double mu = w->con_mu[c] / sqrt(m->impratio > 0 ? m->impratio : 1.0);
    double Rpy = 2 * mu * mu * w->efc_R[e0];
    for (int r = 0; r < 4; r++) w->efc_R[e0 + r] = fmax(MINVAL, Rpy);
  }
}

/* ---- elliptic cone (condim 3), primal side: mj_constraintUpdate's three zones -------------------------------------------
 * jar = (normal, tangent 1, tangent 2) residuals of one contact; D0 = 1 / R of the normal row, mu = friction / sqrt(impratio),
 * fr = the friction coefficient of the tangents.  With U_0 = mu jar_0, U_j = fr jar_j, N = U_0, T = |U_1..2| the cost is
 * D0 / mu^2 times half the squared distance of U to the cone N >= mu T:
 *   top zone    (N >= mu T):        0, no force
 *   bottom zone (mu N + T <= 0):    0.5 sum D_j jar_j^2, force_j = -D_j jar_j            (D_j = D0 fr^2 / mu^2 for the tangents)
 *   middle zone:                    0.5 Dm (N - mu T)^2, Dm = D0 / (mu^2 (1 + mu^2)), force_0 = -Dm (N - mu T) mu,
 *                                   force_j = -force_0 U_j fr / T
 * Returns the zone (0 top, 1 bottom, 2 middle); force / hess (3 x 3, d2 cost / d jar2) may be NULL. */
static int cone_zone(const double* jar, double D0, double mu, double fr, double* cost, double* force, double* hess) {
  double U[3] = {jar[0] * mu, jar[1] * fr, jar[2] * fr};
  double N = U[0], T = sqrt(U[1] * U[1] + U[2] * U[2]);
  if (force) force[0] = force[1] = force[2] = 0;
  if (hess) memset(hess, 0, 9 * sizeof(double));
  *cost = 0;
  if (N >= mu * T || (T <= 0 && N >= 0)) return 0;
  if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
    double Dj[3] = {D0, D0 * fr * fr / (mu * mu), D0 * fr * fr / (mu * mu)};
    for (int j = 0; j < 3; j++) { *cost += 0.5 * Dj[j] * jar[j] * jar[j]; if (force) force[j] = -Dj[j] * jar[j]; if (hess) hess[4 * j] = Dj[j]; }
    return 1;
  }
  double Dm = D0 / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
  *cost = 0.5 * Dm * NmT * NmT;
  if (force) { force[0] = -Dm * NmT * mu; force[1] = -force[0] / T * U[1] * fr; force[2] = -force[0] / T * U[2] * fr; }
  if (hess) {
    /* a = d(N - mu T)/d jar; d2 T / d jar_a d jar_b = fr^2 delta_ab / T - fr^4 jar_a jar_b / T^3 on the tangents */
    double a[3] = {mu, -mu * fr * U[1] / T, -mu * fr * U[2] / T};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) hess[3 * i + j] = Dm * a[i] * a[j];
    double k = -Dm * NmT * mu;                      /* > 0 in the middle zone */
    for (int i = 1; i < 3; i++) for (int j = 1; j < 3; j++)
      hess[3 * i + j] += k * ((i == j ? fr * fr / T : 0.0) - fr * fr * U[i] * U[j] / (T * T * T));
  }
  return 2;
}
/* first row of an elliptic contact whose three rows all fit the row budget? */
static int ell_block(const ws_t* w, int e) {
  return w->efc_type[e] == EFC_ELLIPTIC && e == w->con_efc[w->efc_id[e]] && e + 3 <= w->nefc;
}
/* forces, active states and cost of the constraint part for residuals jar (mj_constraintUpdate): unilateral rows + cones */
static double rows_update(const fmj_model* m, const ws_t* w, const double* jar, double* force, int* state) {
  double cost = 0;
  const double isq = 1.0 / sqrt(m->impratio > 0 ? m->impratio : 1.0);
  for (int e = 0; e < w->nefc; e++) {
    if (ell_block(w, e)) {
      double c, f[3]; int c_ = w->efc_id[e];
      int z = cone_zone(jar + e, 1.0 / w->efc_R[e], w->con_mu[c_] * isq, w->con_mu[c_], &c, f, NULL);
      cost += c;
      for (int j = 0; j < 3; j++) { if (force) force[e + j] = f[j]; if (state) state[e + j] = z; }
      e += 2;
      continue;
    }
    double D = 1.0 / w->efc_R[e];
    if (jar[e] < 0) { if (force) force[e] = -D * jar[e]; cost += 0.5 * D * jar[e] * jar[e]; if (state) state[e] = 1; }
    else { if (force) force[e] = 0; if (state) state[e] = 0; }
  }
  return cost;
}

/* mju_QCQP2: min 0.5 x'A x + x'b  s.t.  sum (x_i / d_i)^2 <= r^2, by Newton's method on the multiplier in scaled coordinates.
 * Returns 1 when the constraint is active. */
static int qcqp2(double* res, const double A_[4], const double b_[2], const double d[2], double r) {
  double b1 = b_[0] * d[0], b2 = b_[1] * d[1];
  double A11 = A_[0] * d[0] * d[0], A22 = A_[3] * d[1] * d[1], A12 = A_[1] * d[0] * d[1];
  double la = 0, r2 = r * r, v1 = 0, v2 = 0;
  for (int it = 0; it < 20; it++) {
    double det = (A11 + la) * (A22 + la) - A12 * A12;
    if (det < 1e-10) { v1 = v2 = 0; break; }
    double di = 1 / det, P11 = (A22 + la) * di, P22 = (A11 + la) * di, P12 = -A12 * di;
    v1 = -P11 * b1 - P12 * b2; v2 = -P12 * b1 - P22 * b2;
    double val = v1 * v1 + v2 * v2 - r2;
    if (val < 1e-10) break;
    double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
    double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * d[0]; res[1] = v2 * d[1];
  return la != 0;
}

/* one PGS pass over the three rows of an elliptic contact starting at row i (mj_solPGS, elliptic branch): first the whole force
 * vector of the contact moves along its own ray (or, with no normal force yet, the normal alone moves and friction is cleared),
 * then the friction forces solve their QCQP inside the cone of the new normal force.  Returns the cost change (<= 0). */
static double pgs_elliptic_block(const fmj_model* m, ws_t* w, int i) {
  int n = w->nefc, c = w->efc_id[i];
  double res[3], A[9], old[3], *f = w->efc_force + i;
  for (int j = 0; j < 3; j++) {
    res[j] = w->efc_b[i + j];
    for (int k = 0; k < n; k++) res[j] += w->efc_AR[(size_t)(i + j) * n + k] * w->efc_force[k];
    for (int k = 0; k < 3; k++) A[3 * j + k] = w->efc_AR[(size_t)(i + j) * n + i + k];
    old[j] = f[j];
  }
  if (f[0] < MINVAL) {
    f[0] -= res[0] / A[0]; if (f[0] < 0) f[0] = 0;
    f[1] = f[2] = 0;
  } else {
    double v[3] = {f[0], f[1], f[2]}, v1[3], denom = 0, num = 0;
    for (int j = 0; j < 3; j++) { v1[j] = A[3 * j] * v[0] + A[3 * j + 1] * v[1] + A[3 * j + 2] * v[2]; }
    for (int j = 0; j < 3; j++) { denom += v[j] * v1[j]; num += v[j] * res[j]; }
    if (denom >= MINVAL) {
      double x = -num / denom;
      if (f[0] + x * v[0] < 0) x = -f[0] / v[0];
      for (int j = 0; j < 3; j++) f[j] += x * v[j];
    }
  }
  if (f[0] < MINVAL) f[1] = f[2] = 0;
  else {
    double Ac[4] = {A[4], A[5], A[7], A[8]}, bc[2], d[2] = {w->con_mu[c], w->con_mu[c]}, v[2];
    for (int j = 1; j < 3; j++) bc[j - 1] = res[j] - A[3 * j + 1] * old[1] - A[3 * j + 2] * old[2] + A[3 * j] * (f[0] - old[0]);
    /* res is the residual at the OLD forces: remove the old friction's share, put the normal's change in */
    if (qcqp2(v, Ac, bc, d, f[0])) {
      double s = (v[0] / d[0]) * (v[0] / d[0]) + (v[1] / d[1]) * (v[1] / d[1]);
      s = sqrt(f[0] * f[0] / fmax(MINVAL, s));
      v[0] *= s; v[1] *= s;
    }
    f[1] = v[0]; f[2] = v[1];
  }
  double dl[3] = {f[0] - old[0], f[1] - old[1], f[2] - old[2]}, change = 0;
  for (int j = 0; j < 3; j++) { change += dl[j] * res[j]; for (int k = 0; k < 3; k++) change += 0.5 * dl[j] * A[3 * j + k] * dl[k]; }
  if (change > 1e-10) { f[0] = old[0]; f[1] = old[1]; f[2] = old[2]; change = 0; }
  (void)m;
  return change;
}

static double dual_cost(const ws_t* w, const double* f) {
  int n = w->nefc; double c = 0;
  for (int i = 0; i < n; i++) {
    double s = 0; for (int j = 0; j < n; j++) s += w->efc_AR[(size_t)i * n + j] * f[j];
    c += f[i] * (0.5 * s + w->efc_b[i]);
  }
  return c;
}

/* AR = J M^-1 J' + R ; b = J qacc_smooth - aref (mj_projectConstraint + the dual's linear term) */
static void build_dual(const fmj_model* m, ws_t* w) {
  int nv = m->nv, n = w->nefc;
  for (int e = 0; e < n; e++) {
    double* x = w->efc_MiJT + (size_t)e * nv;
    memcpy(x, w->efc_J + (size_t)e * nv, nv * sizeof(double));
    solve_ld(m, x, w->qLD, w->qLDiagInv);
  }
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) w->efc_AR[(size_t)i * n + j] = dotn(w->efc_J + (size_t)i * nv, w->efc_MiJT + (size_t)j * nv, nv);
    w->efc_AR[(size_t)i * n + i] += w->efc_R[i];
    w->efc_b[i] = dotn(w->efc_J + (size_t)i * nv, w->qacc_smooth, nv) - w->efc_aref[i];
  }
}

/* qfrc_constraint = J' f, the joint-limit sensor forces, qacc = qacc_smooth + M^-1 qfrc_constraint (MuJoCo's dualFinish) */
static void dual_finish(const fmj_model* m, ws_t* w) {
  int nv = m->nv, n = w->nefc;
  memset(w->qfrc_constraint, 0, nv * sizeof(double));
  memset(w->jointlimitfrc, 0, m->njnt * sizeof(double));
  for (int e = 0; e < n; e++) {
    for (int i = 0; i < nv; i++) w->qfrc_constraint[i] += w->efc_J[(size_t)e * nv + i] * w->efc_force[e];
    if (w->efc_type[e] == EFC_LIMIT) w->jointlimitfrc[w->efc_id[e]] += w->efc_force[e];
  }
  memcpy(w->qacc, w->qfrc_constraint, nv * sizeof(double));
  solve_ld(m, w->qacc, w->qLD, w->qLDiagInv);
  for (int i = 0; i < nv; i++) w->qacc[i] += w->qacc_smooth[i];
}

static int qcqp2(double* res, const double A_[4], const double b_[2], const double d[2], double r);

/* mj_solNoSlip (option.noslip_iterations, reference mjcf.py:1392-1403), restated from MuJoCo's documentation and solver source
 * as recalled [MJ-knowledge]; parity unpinned like the rest.  A post-pass after the main solver, whatever it was: Gauss-Seidel on the
 * dual problem WITHOUT the regulariser R, over the friction dimensions of the contacts only - the normal forces (and the limit
 * rows) keep the values the main solver gave them, so the soft contact still yields along its normal, but the friction forces are
 * re-solved as hard constraints: a sticking contact ends up with zero tangential acceleration instead of the slow creep the
 * regulariser allows.
 *   pyramidal contact: its rows come in opposing pairs (j, j + 1) = normal +- mu tangent.  The pair's sum (its share of the normal
 *     force) stays fixed, mid = (f_j + f_j+1) / 2; the difference y is minimised exactly: with the 2 x 2 block Ac of A (R removed)
 *     and bc = residual - Ac f_old,  K1 = Ac00 + Ac11 - 2 Ac01,  K0 = mid (Ac00 - Ac11) + bc0 - bc1,  y = -K0 / K1 clamped to
 *     [-mid, mid], f = (mid + y, mid - y);
 *   elliptic contact: with the normal force fixed, the friction forces solve the QCQP  min 0.5 v'Ac v + v'bc,  |v / mu| <= f_n
 *     (mju_QCQP2 for condim 3), rescaled onto the cone when the constraint is active.
 * A sweep's improvement (decrease of the unregularised dual cost; the first sweep adds the regulariser's share 0.5 R f^2 of the
 * main solver's forces) times 1 / (meaninertia max(1, nv)) is tested against noslip_tolerance. */
static void noslip(const fmj_model* m, ws_t* w) {
  const int n = w->nefc, nv = m->nv;
  if (n == 0 || m->noslip_iterations <= 0 || w->ncon == 0) return;
  if (m->solver == FMJ_SOLVER_NEWTON || m->solver == FMJ_SOLVER_CG) build_dual(m, w);      /* the primal solvers never form A */
  double* f = w->efc_force;
  const double scale = 1.0 / (w->meaninertia * (nv > 1 ? nv : 1));
#define A_(i_, j_) (w->efc_AR[(size_t)(i_) * n + (j_)] - ((i_) == (j_) ? w->efc_R[i_] : 0.0))
  int iter = 0;
  while (iter < m->noslip_iterations) {
    double improvement = 0;
    if (iter == 0) for (int i = 0; i < n; i++) improvement += 0.5 * f[i] * f[i] * w->efc_R[i];
    for (int c = 0; c < w->ncon; c++) {
      const int i = w->con_efc[c];
      if (m->cone == FMJ_CONE_ELLIPTIC) {
        if (i + 3 > n) continue;
        double res[3], old[3] = {f[i], f[i + 1], f[i + 2]};
        for (int j = 0; j < 3; j++) { res[j] = w->efc_b[i + j]; for (int k = 0; k < n; k++) res[j] += A_(i + j, k) * f[k]; }
        if (f[i] < MINVAL) f[i + 1] = f[i + 2] = 0;
        else {
          double Ac[4] = {A_(i + 1, i + 1), A_(i + 1, i + 2), A_(i + 2, i + 1), A_(i + 2, i + 2)}, bc[2], d[2] = {w->con_mu[c], w->con_mu[c]}, v[2];
          for (int j = 0; j < 2; j++) bc[j] = res[1 + j] - Ac[2 * j] * old[1] - Ac[2 * j + 1] * old[2];
          if (qcqp2(v, Ac, bc, d, f[i])) {
            double sv = (v[0] / d[0]) * (v[0] / d[0]) + (v[1] / d[1]) * (v[1] / d[1]);
            sv = sqrt(f[i] * f[i] / fmax(MINVAL, sv));
            v[0] *= sv; v[1] *= sv;
          }
          f[i + 1] = v[0]; f[i + 2] = v[1];
        }
        double dl[3] = {0, f[i + 1] - old[1], f[i + 2] - old[2]}, change = 0;
        for (int j = 1; j < 3; j++) { change += dl[j] * res[j]; for (int k = 1; k < 3; k++) change += 0.5 * dl[j] * A_(i + j, i + k) * dl[k]; }
        if (change > 1e-10) { f[i + 1] = old[1]; f[i + 2] = old[2]; change = 0; }
        improvement -= change;
        continue;
      }
      if (i + 4 > n) continue;
      for (int j = i; j < i + 4; j += 2) {
        double res[2], old[2] = {f[j], f[j + 1]};
        for (int r = 0; r < 2; r++) { res[r] = w->efc_b[j + r]; for (int k = 0; k < n; k++) res[r] += A_(j + r, k) * f[k]; }
        const double Ac[4] = {A_(j, j), A_(j, j + 1), A_(j + 1, j), A_(j + 1, j + 1)};
        const double bc0 = res[0] - Ac[0] * old[0] - Ac[1] * old[1], bc1 = res[1] - Ac[2] * old[0] - Ac[3] * old[1];
        const double mid = 0.5 * (old[0] + old[1]);
        const double K1 = Ac[0] + Ac[3] - Ac[1] - Ac[2], K0 = mid * (Ac[0] - Ac[3]) + bc0 - bc1;
        if (K1 < MINVAL) f[j] = f[j + 1] = mid;
        else {
          const double y = -K0 / K1;
          if (y < -mid) { f[j] = 0; f[j + 1] = 2 * mid; }
          else if (y > mid) { f[j] = 2 * mid; f[j + 1] = 0; }
          else { f[j] = mid + y; f[j + 1] = mid - y; }
        }
        const double dl[2] = {f[j] - old[0], f[j + 1] - old[1]};
        double change = dl[0] * res[0] + dl[1] * res[1] + 0.5 * (dl[0] * (Ac[0] * dl[0] + Ac[1] * dl[1]) + dl[1] * (Ac[2] * dl[0] + Ac[3] * dl[1]));
        if (change > 1e-10) { f[j] = old[0]; f[j + 1] = old[1]; change = 0; }
        improvement -= change;
      }
    }
#undef A_
    iter++;
    if (improvement * scale < m->noslip_tolerance) break;
#define A_(i_, j_) (w->efc_AR[(size_t)(i_) * n + (j_)] - ((i_) == (j_) ? w->efc_R[i_] : 0.0))
  }
#undef A_
  g_last_noslip_iterations = iter;
  dual_finish(m, w);
}

static void solve_primal(const fmj_model* m, ws_t* w, int newton);
static void solve_constraints(const fmj_model* m, ws_t* w) {
  int nv = m->nv, n = w->nefc;
  memset(w->qfrc_constraint, 0, nv * sizeof(double));
  memset(w->jointlimitfrc, 0, m->njnt * sizeof(double));
  if (n == 0) { memcpy(w->qacc, w->qacc_smooth, nv * sizeof(double)); g_last_pgs_iterations = 0; return; }
  if (m->solver == FMJ_SOLVER_NEWTON || m->solver == FMJ_SOLVER_CG) { solve_primal(m, w, m->solver == FMJ_SOLVER_NEWTON); noslip(m, w); return; }
  build_dual(m, w);
  /* warm start from previous qacc (mj_fwdConstraint) */
  {
    double* jar = dalloc(n);
    for (int i = 0; i < n; i++) jar[i] = dotn(w->efc_J + (size_t)i * nv, w->qacc_warmstart, nv) - w->efc_aref[i];
    rows_update(m, w, jar, w->efc_force, NULL);      /* mj_constraintUpdate at the warm start: cone zones for elliptic contacts */
    free(jar);
  }
  if (dual_cost(w, w->efc_force) > 0) memset(w->efc_force, 0, n * sizeof(double));
  /* PGS (mj_solPGS) */
  double scale = 1.0 / (w->meaninertia * (nv > 1 ? nv : 1));
  for (int it = 0; it < m->solver_iterations; it++) {
    double improvement = 0;
    for (int i = 0; i < n; i++) {
      if (ell_block(w, i)) { improvement -= pgs_elliptic_block(m, w, i); i += 2; continue; }
      double res = w->efc_b[i];
      for (int j = 0; j < n; j++) res += w->efc_AR[(size_t)i * n + j] * w->efc_force[j];
      double old = w->efc_force[i];
      double f = old - res / w->efc_AR[(size_t)i * n + i];
      if (f < 0) f = 0;
      double delta = f - old;
      double change = 0.5 * delta * delta * w->efc_AR[(size_t)i * n + i] + delta * res;
      if (change > 1e-10) { f = old; change = 0; }
      w->efc_force[i] = f;
      improvement -= change;
    }
    g_last_pgs_iterations = it + 1;
    if (improvement * scale < m->solver_tolerance) break;
  }
  dual_finish(m, w);
  noslip(m, w);
}

/* ------------------------------------------------------------------------------------------ */
/* primal solvers: Newton and CG (mj_solNewton / mj_solCG = mj_solPrimal; MuJoCo's published algorithm, restated from
 * its documentation [MJ-knowledge], not compiled from it).  Reference mjcf.py:1348-1359 forwards
 * simulation_options.solver / n_solver_iters; its own fallback is 'Newton' with 1000 iterations.
 *
 * The problem (pyramidal cone, limits: every row unilateral; an elliptic contact replaces its rows' terms by the cone cost of
 * cone_zone above and its share of the Hessian by J_c' Hc J_c):
 *     minimise over qacc   0.5 (qacc - qacc_smooth)' M (qacc - qacc_smooth) + sum_e s_e(J_e qacc - aref_e),
 *     s_e(x) = 0.5 x^2 / R_e for x < 0, else 0;        force_e = -x / R_e for x < 0, else 0.
 * Its dual is the problem PGS solves (0.5 f'(A + R) f + f'b over f >= 0), so a converged PGS and the Newton minimiser
 * give the same forces and the same qacc = qacc_smooth + M^-1 J' f: tests/test_oracle_solvers.py holds the two to 1e-8.
 * Newton: search = -H^-1 grad with H = M + J_active' D J_active (dense Cholesky); CG: Polak-Ribiere on the M^-1
 * preconditioned gradient.  Line search: exact derivatives of the piecewise-quadratic cost along the search line, Newton
 * steps on phi' from alpha = 0 until phi' changes sign, then Newton safeguarded by the bracket, to
 * |phi'| < tolerance * ls_tolerance * |search| / scale - the ingredients of MuJoCo's PrimalLineSearch; the iterates inside
 * the bracket may differ from MuJoCo's, the minimiser they close in on cannot. */

static void mul_M_sparse(const fmj_model* m, const double* M, const double* v, double* res) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) res[i] = 0;
  for (int i = 0; i < nv; i++) {
    int a = m->dof_Madr[i];
    res[i] += M[a++] * v[i];
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j], a++) { res[i] += M[a] * v[j]; res[j] += M[a] * v[i]; }
  }
}

typedef struct primal_t {
  const fmj_model* m; ws_t* w; int nv, n;
  double *qacc, *Ma, *jar, *grad, *Mgrad, *search, *Mv, *Jv, *H, *D, *force, *qfc;
  int* state;
  double cost;
} primal_t;

/* mj_constraintUpdate for unilateral rows + the Gauss term: cost, forces, qfrc_constraint, active set */
static void primal_update(primal_t* P) {
  const fmj_model* m = P->m; ws_t* w = P->w; int nv = P->nv, n = P->n;
  double cost = rows_update(m, w, P->jar, P->force, P->state);
  for (int i = 0; i < nv; i++) P->qfc[i] = 0;
  for (int e = 0; e < n; e++) if (P->state[e]) for (int i = 0; i < nv; i++) P->qfc[i] += w->efc_J[(size_t)e * nv + i] * P->force[e];
  double g = 0;
  for (int i = 0; i < nv; i++) g += (P->Ma[i] - w->qfrc_smooth[i]) * (P->qacc[i] - w->qacc_smooth[i]);
  P->cost = cost + 0.5 * g;
  for (int i = 0; i < nv; i++) P->grad[i] = P->Ma[i] - w->qfrc_smooth[i] - P->qfc[i];
  (void)m;
}

/* Mgrad = H^-1 grad, H = M + J_active' D J_active, dense Cholesky (mj_solNewton's MakeHessian / FactorizeHessian) */
static void newton_direction(primal_t* P) {
  const fmj_model* m = P->m; ws_t* w = P->w; int nv = P->nv, n = P->n;
  double* H = P->H;
  memset(H, 0, (size_t)nv * nv * sizeof(double));
  for (int i = 0; i < nv; i++) { int a = m->dof_Madr[i]; for (int j = i; j >= 0; j = m->dof_parentid[j]) { H[i * nv + j] = H[j * nv + i] = w->qM[a++]; } }
  const double isq = 1.0 / sqrt(m->impratio > 0 ? m->impratio : 1.0);
  for (int e = 0; e < n; e++) if (P->state[e]) {
    if (ell_block(w, e)) {
      /* bottom zone: three quadratic rows (their own D); middle zone: J_c' Hc J_c with the 3 x 3 Hessian of the cone */
      double c, hc[9]; int c_ = w->efc_id[e];
      cone_zone(P->jar + e, P->D[e], w->con_mu[c_] * isq, w->con_mu[c_], &c, NULL, hc);
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        if (hc[3 * a + b] == 0) continue;
        const double *Ja = w->efc_J + (size_t)(e + a) * nv, *Jb = w->efc_J + (size_t)(e + b) * nv;
        for (int i = 0; i < nv; i++) if (Ja[i] != 0) for (int j = 0; j < nv; j++) H[i * nv + j] += hc[3 * a + b] * Ja[i] * Jb[j];
      }
      e += 2;
      continue;
    }
    const double* J = w->efc_J + (size_t)e * nv; double d = P->D[e];
    for (int i = 0; i < nv; i++) if (J[i] != 0) for (int j = 0; j < nv; j++) H[i * nv + j] += d * J[i] * J[j];
  }
  for (int k = 0; k < nv; k++) {            /* in-place lower Cholesky */
    double s = H[k * nv + k];
    for (int c = 0; c < k; c++) s -= H[k * nv + c] * H[k * nv + c];
    s = sqrt(s > MINVAL ? s : MINVAL);
    H[k * nv + k] = s;
    for (int i = k + 1; i < nv; i++) {
      double t = H[i * nv + k];
      for (int c = 0; c < k; c++) t -= H[i * nv + c] * H[k * nv + c];
      H[i * nv + k] = t / s;
    }
  }
  double* x = P->Mgrad;
  for (int i = 0; i < nv; i++) { double t = P->grad[i]; for (int c = 0; c < i; c++) t -= H[i * nv + c] * x[c]; x[i] = t / H[i * nv + i]; }
  for (int i = nv - 1; i >= 0; i--) { double t = x[i]; for (int c = i + 1; c < nv; c++) t -= H[c * nv + i] * x[c]; x[i] = t / H[i * nv + i]; }
}

typedef struct ls_pt { double alpha, cost, d0, d1; } ls_pt;
static ls_pt ls_eval(const primal_t* P, const double* qg, double alpha) {
  ls_pt p; p.alpha = alpha;
  double c = qg[0] + alpha * (qg[1] + alpha * qg[2]), d0 = qg[1] + 2 * alpha * qg[2], d1 = 2 * qg[2];
  const double isq = 1.0 / sqrt(P->m->impratio > 0 ? P->m->impratio : 1.0);
  for (int e = 0; e < P->n; e++) {
    if (ell_block(P->w, e)) {
      double x3[3], cc, f3[3], hc[9]; int c_ = P->w->efc_id[e];
      for (int j = 0; j < 3; j++) x3[j] = P->jar[e + j] + alpha * P->Jv[e + j];
      cone_zone(x3, P->D[e], P->w->con_mu[c_] * isq, P->w->con_mu[c_], &cc, f3, hc);
      c += cc;
      for (int a = 0; a < 3; a++) { d0 -= f3[a] * P->Jv[e + a]; for (int b = 0; b < 3; b++) d1 += hc[3 * a + b] * P->Jv[e + a] * P->Jv[e + b]; }
      e += 2;
      continue;
    }
    double x = P->jar[e] + alpha * P->Jv[e];
    if (x < 0) { c += 0.5 * P->D[e] * x * x; d0 += P->D[e] * x * P->Jv[e]; d1 += P->D[e] * P->Jv[e] * P->Jv[e]; }
  }
  p.cost = c; p.d0 = d0; p.d1 = d1 > MINVAL ? d1 : MINVAL;
  return p;
}
static double primal_linesearch(primal_t* P, double scale) {
  const fmj_model* m = P->m; ws_t* w = P->w; int nv = P->nv, n = P->n;
  mul_M_sparse(m, w->qM, P->search, P->Mv);
  for (int e = 0; e < n; e++) P->Jv[e] = dotn(w->efc_J + (size_t)e * nv, P->search, nv);
  double snorm = sqrt(dotn(P->search, P->search, nv));
  if (snorm < MINVAL) return 0;
  double qg[3] = {0, 0, 0.5 * dotn(P->search, P->Mv, nv)};
  for (int i = 0; i < nv; i++) { qg[0] += 0.5 * (P->Ma[i] - w->qfrc_smooth[i]) * (P->qacc[i] - w->qacc_smooth[i]); qg[1] += P->search[i] * (P->Ma[i] - w->qfrc_smooth[i]); }
  const int lsmax = m->ls_iterations > 0 ? m->ls_iterations : 50;
  const double gtol = m->solver_tolerance * (m->ls_tolerance > 0 ? m->ls_tolerance : 0.01) * snorm / scale;
  ls_pt p0 = ls_eval(P, qg, 0.0);
  ls_pt p1 = ls_eval(P, qg, -p0.d0 / p0.d1);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d0) < gtol) return p1.alpha;
  int it = 0;
  const double dir = p1.d0 < 0 ? 1.0 : -1.0;
  ls_pt p2 = p1;
  while (p1.d0 * dir <= -gtol && it < lsmax) {       /* one-sided Newton until phi' changes sign */
    p2 = p1; p1 = ls_eval(P, qg, p1.alpha - p1.d0 / p1.d1); it++;
    if (fabs(p1.d0) < gtol) return p1.alpha;
  }
  if (it >= lsmax || p1.d0 * dir <= 0) return p1.alpha;
  ls_pt lo = dir > 0 ? p2 : p1, hi = dir > 0 ? p1 : p2;   /* phi'(lo) < 0 < phi'(hi): the minimiser is bracketed */
  ls_pt best = fabs(lo.d0) < fabs(hi.d0) ? lo : hi;
  while (it < lsmax) {
    double a = best.alpha - best.d0 / best.d1;
    if (!(a > lo.alpha && a < hi.alpha)) a = 0.5 * (lo.alpha + hi.alpha);
    ls_pt p = ls_eval(P, qg, a); it++;
    if (fabs(p.d0) < gtol) return p.alpha;
    if (p.d0 < 0) lo = p; else hi = p;
    best = p;
    if (hi.alpha - lo.alpha <= 1e-16 * fabs(hi.alpha)) break;
  }
  return best.cost < p0.cost ? best.alpha : 0.0;
}

/* mj_solPrimal: Newton (newton = 1) or CG.  Starts from w->qacc (set by the warm start) */
static void solve_primal(const fmj_model* m, ws_t* w, int newton) {
  int nv = m->nv, n = w->nefc;
  primal_t P; memset(&P, 0, sizeof P);
  P.m = m; P.w = w; P.nv = nv; P.n = n;
  double* buf = dalloc((size_t)9 * nv + 4 * (size_t)n + (size_t)nv * nv);
  P.qacc = buf; P.Ma = buf + nv; P.grad = buf + 2 * nv; P.Mgrad = buf + 3 * nv; P.search = buf + 4 * nv; P.Mv = buf + 5 * nv; P.qfc = buf + 6 * nv;
  double* oldgrad = buf + 7 * nv; double* oldMgrad = buf + 8 * nv;
  P.jar = buf + 9 * nv; P.Jv = P.jar + n; P.D = P.Jv + n; P.force = P.D + n; P.H = P.force + n;
  P.state = (int*)calloc(n ? n : 1, sizeof(int));
  const double scale = 1.0 / (w->meaninertia * (nv > 1 ? nv : 1));
  for (int e = 0; e < n; e++) P.D[e] = 1.0 / w->efc_R[e];
  /* warm start (mj_fwdConstraint): the previous qacc unless qacc_smooth costs less */
  double cost_ws;
  memcpy(P.qacc, w->qacc_warmstart, nv * sizeof(double));
  mul_M_sparse(m, w->qM, P.qacc, P.Ma);
  for (int e = 0; e < n; e++) P.jar[e] = dotn(w->efc_J + (size_t)e * nv, P.qacc, nv) - w->efc_aref[e];
  primal_update(&P); cost_ws = P.cost;
  {
    double* js = dalloc(n);
    for (int e = 0; e < n; e++) js[e] = dotn(w->efc_J + (size_t)e * nv, w->qacc_smooth, nv) - w->efc_aref[e];
    double cs = rows_update(m, w, js, NULL, NULL);
    free(js);
    if (cost_ws > cs) {
      memcpy(P.qacc, w->qacc_smooth, nv * sizeof(double));
      mul_M_sparse(m, w->qM, P.qacc, P.Ma);
      for (int e = 0; e < n; e++) P.jar[e] = dotn(w->efc_J + (size_t)e * nv, P.qacc, nv) - w->efc_aref[e];
      primal_update(&P);
    }
  }
  if (newton) newton_direction(&P);
  else { memcpy(P.Mgrad, P.grad, nv * sizeof(double)); solve_ld(m, P.Mgrad, w->qLD, w->qLDiagInv); }
  for (int i = 0; i < nv; i++) P.search[i] = -P.Mgrad[i];
  int iter = 0;
  while (iter < m->solver_iterations) {
    double alpha = primal_linesearch(&P, scale);
    if (alpha == 0) break;
    for (int i = 0; i < nv; i++) { P.qacc[i] += alpha * P.search[i]; P.Ma[i] += alpha * P.Mv[i]; }
    for (int e = 0; e < n; e++) P.jar[e] += alpha * P.Jv[e];
    double oldcost = P.cost;
    if (!newton) { memcpy(oldgrad, P.grad, nv * sizeof(double)); memcpy(oldMgrad, P.Mgrad, nv * sizeof(double)); }
    primal_update(&P);
    if (newton) { newton_direction(&P); for (int i = 0; i < nv; i++) P.search[i] = -P.Mgrad[i]; }
    else {
      memcpy(P.Mgrad, P.grad, nv * sizeof(double)); solve_ld(m, P.Mgrad, w->qLD, w->qLDiagInv);
      double num = 0, den = dotn(oldgrad, oldMgrad, nv);
      for (int i = 0; i < nv; i++) num += P.grad[i] * (P.Mgrad[i] - oldMgrad[i]);
      double beta = num / fmax(MINVAL, den); if (beta < 0) beta = 0;                 /* Polak-Ribiere */
      for (int i = 0; i < nv; i++) P.search[i] = -P.Mgrad[i] + beta * P.search[i];
    }
    double improvement = scale * (oldcost - P.cost), gradient = scale * sqrt(dotn(P.grad, P.grad, nv));
    iter++;
    if (improvement < m->solver_tolerance || gradient < m->solver_tolerance) break;
  }
  g_last_pgs_iterations = iter;
  memcpy(w->qacc, P.qacc, nv * sizeof(double));
  memcpy(w->efc_force, P.force, n * sizeof(double));
  memcpy(w->qfrc_constraint, P.qfc, nv * sizeof(double));
  for (int e = 0; e < n; e++) if (w->efc_type[e] == EFC_LIMIT) w->jointlimitfrc[w->efc_id[e]] += w->efc_force[e];
  free(P.state); free(buf);
}

/* ------------------------------------------------------------------------------------------ */
/* sensors (Appendix A.11) in the order of reference mjcf.py:950-1002                            */

static void sensors(const fmj_model* m, const ws_t* w, const double* qpos, const double* qvel, double* sd) {
  if (!sd) return;
  int adr = 0;
  for (int b = 1; b < m->nbody; b++) {   /* framelinvel, frameangvel (objtype=body => inertial frame) */
    const double* cv = w->cvel + 6 * b;
    double dif[3], c[3];
    for (int k = 0; k < 3; k++) dif[k] = w->xipos[3 * b + k] - w->subtree_com[3 * m->body_rootid[b] + k];
    cross3(c, dif, cv);
    for (int k = 0; k < 3; k++) sd[adr + k] = cv[3 + k] - c[k];
    for (int k = 0; k < 3; k++) sd[adr + 3 + k] = cv[k];
    adr += 6;
  }
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_type[j] == FMJ_JNT_FREE) continue;
    sd[adr++] = qpos[m->jnt_qposadr[j]];
    sd[adr++] = qvel[m->jnt_dofadr[j]];
    sd[adr++] = w->jointlimitfrc[j];
  }
  for (int a = 0; a < m->nu; a++) sd[adr++] = w->actuator_force[a];
}

/* ------------------------------------------------------------------------------------------ */
/* mj_forward + mj_Euler                                                                        */

static int forward(const fmj_model* m, ws_t* w, const double* qpos, const double* qvel, const double* ctrl,
                   const double* qpos_spring, const double* xfrc, double* sensordata) {
  int nv = m->nv, warn = 0;
  kinematics(m, w, qpos);
  com_pos(m, w);
  crb(m, w);
  factor(m, w->qM, w->qLD, w->qLDiagInv);
  com_vel(m, w, qvel);
  passive(m, w, qpos, qvel, qpos_spring);
  rne(m, w, qvel);
  actuation(m, w, qpos, qvel, ctrl);
  xfrc_accumulate(m, w, xfrc);
  for (int i = 0; i < nv; i++)
    w->qfrc_smooth[i] = w->qfrc_passive[i] - w->qfrc_bias[i] + w->qfrc_actuator[i] + w->qfrc_xfrc[i];
  if (g_fp32_storage >= 3) { round_to_f32(w->cdof, 6 * nv); round_to_f32(w->cvel, 6 * m->nbody); round_to_f32(w->qfrc_smooth, nv); }
  memcpy(w->qacc_smooth, w->qfrc_smooth, nv * sizeof(double));
  solve_ld(m, w->qacc_smooth, w->qLD, w->qLDiagInv);
  if (g_fp32_storage >= 3) round_to_f32(w->qacc_smooth, nv);
  make_constraints(m, w, qpos, qvel, &warn);
  if (g_fp32_storage >= 3) { round_to_f32(w->efc_J, w->nefc * nv); round_to_f32(w->efc_aref, w->nefc); round_to_f32(w->efc_R, w->nefc); }
  solve_constraints(m, w);
  sensors(m, w, qpos, qvel, sensordata);
  return warn;
}

/* mj_integratePos: qpos advanced by h * vel (quaternion of a free joint: rotated by the angle h |w| about w) */
static void integrate_pos(const fmj_model* m, double* qpos, const double* vel, double h) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == FMJ_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * vel[da + k];
      double ax[3] = {vel[da + 3], vel[da + 4], vel[da + 5]};
      double nrm = sqrt(dotn(ax, ax, 3));
      if (nrm < MINVAL) { ax[0] = 1; ax[1] = ax[2] = 0; nrm = 0; } else for (int k = 0; k < 3; k++) ax[k] /= nrm;
      double qr[4];
      axis_angle2quat(qr, ax, h * nrm);
      normalize4(qpos + qa + 3);
      mul_quat(qpos + qa + 3, qpos + qa + 3, qr);
    } else qpos[qa] += h * vel[da];
  }
}

static void euler(const fmj_model* m, ws_t* w, double* qpos, double* qvel) {
  int nv = m->nv; double h = m->timestep;
  int damped = 0;
  /* mj_Euler: implicit in the joint damping (eulerdamp).  mj_implicit with mjINT_IMPLICITFAST: M - h D with D = d(qfrc_passive +
   * qfrc_actuator)/d qvel, Coriolis terms dropped and D symmetrised; with joint-transmission actuators and joint dampers only, D is
   * diagonal: -damping - sum over the joint's actuators of (-biasprm[2]) (mjd_passive_vel, mjd_actuator_vel; an actuator whose force
   * sits on its forcerange contributes nothing).  Recalled from MuJoCo's documentation like everything else here: parity unpinned. */
  double* bd = (double*)calloc((size_t)(nv > 0 ? nv : 1), sizeof(double));
  for (int i = 0; i < nv; i++) bd[i] = m->dof_damping[i];
  if (m->integrator == FMJ_INT_IMPLICITFAST && !w->disable_actuation) {
    for (int a = 0; a < m->nu; a++) {
      const double f = w->actuator_force[a];
      if (m->actuator_forcelimited[a] && (f <= m->actuator_forcerange[2 * a] || f >= m->actuator_forcerange[2 * a + 1])) continue;
      bd[m->jnt_dofadr[m->actuator_jntid[a]]] += -m->actuator_bias[3 * a + 2];
    }
  }
  for (int i = 0; i < nv; i++) damped |= (bd[i] != 0);
  double* qacc = w->tmpv;
  if (!damped) memcpy(qacc, w->qacc, nv * sizeof(double));
  else {
    memcpy(w->qH, w->qM, m->nM * sizeof(double));
    for (int i = 0; i < nv; i++) w->qH[m->dof_Madr[i]] += h * bd[i];
    if (g_fp32_storage) round_to_f32(w->qH, m->nM);
    factor(m, w->qH, w->qH, w->qHDiagInv);
    for (int i = 0; i < nv; i++) qacc[i] = w->qfrc_smooth[i] + w->qfrc_constraint[i];
    solve_ld(m, qacc, w->qH, w->qHDiagInv);
  }
  free(bd);
  for (int i = 0; i < nv; i++) qvel[i] += h * qacc[i];
  integrate_pos(m, qpos, qvel, h);
}

/* mj_RungeKutta(m, d, 4) after the step's mj_forward (recalled from MuJoCo's engine_forward.c, parity unpinned like the rest): the classical
 * tableau A = [[1/2], [0, 1/2], [0, 0, 1]], B = [1/6, 1/3, 1/3, 1/6].  Stage i forwards the state X[i] = X[0] (+) h A[i-1] F[i-1] with
 * mj_forwardSkip(mjSTAGE_NONE, skipsensor = 1): positions, velocities, constraints and qacc are recomputed, sensordata keeps the values of the
 * step's first forward pass; positions are advanced with mj_integratePos from X[0] by the stage's VELOCITY, velocities by its acceleration.
 * No implicit damping: qacc is M^-1 (...) (the Euler-only eulerdamp does not apply).  The warm start of all four passes is the previous step's
 * (mj_advance saves qacc - the fourth pass's - at the end).  ctrl and xfrc_applied are held over the step. */
static int rk4(const fmj_model* m, ws_t* w, double* qpos, double* qvel, const double* ctrl, const double* qpos_spring, const double* xfrc) {
  static const double RA[3] = {0.5, 0.5, 1.0}, RB[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
  int nq = m->nq, nv = m->nv, warn = 0;
  double* q0 = (double*)malloc(sizeof(double) * (size_t)(nq + 3 * nv + 1));
  double* v0 = q0 + nq; double* sv = v0 + nv; double* sa = sv + nv;
  memcpy(q0, qpos, nq * sizeof(double)); memcpy(v0, qvel, nv * sizeof(double));
  for (int i = 0; i < nv; i++) { sv[i] = RB[0] * qvel[i]; sa[i] = RB[0] * w->qacc[i]; }
  for (int s = 1; s < 4; s++) {
    /* X[s]: from X[0] by the derivative of stage s - 1 (its velocity is still in qvel, its acceleration in w->qacc) */
    memcpy(qpos, q0, nq * sizeof(double));
    integrate_pos(m, qpos, qvel, m->timestep * RA[s - 1]);
    for (int i = 0; i < nv; i++) qvel[i] = v0[i] + m->timestep * RA[s - 1] * w->qacc[i];
    if (g_fp32_storage >= 2) { round_to_f32(qpos, nq); round_to_f32(qvel, nv); }
    warn |= forward(m, w, qpos, qvel, ctrl, qpos_spring, xfrc, NULL);      /* skipsensor: sensordata stays the first pass's */
    for (int i = 0; i < nv; i++) { sv[i] += RB[s] * qvel[i]; sa[i] += RB[s] * w->qacc[i]; }
  }
  memcpy(qpos, q0, nq * sizeof(double));
  integrate_pos(m, qpos, sv, m->timestep);
  for (int i = 0; i < nv; i++) qvel[i] = v0[i] + m->timestep * sa[i];
  free(q0);
  return warn;
}

static int bad(const double* x, int n) {
  for (int i = 0; i < n; i++) if (!(fabs(x[i]) <= MAXVAL)) return 1;
  return 0;
}

static double mean_inertia(const fmj_model* m) {
  /* stat.meaninertia: mean diagonal of M at qpos0 */
  ws_t* w = ws_new(m);
  kinematics(m, w, m->qpos0); com_pos(m, w); crb(m, w);
  double s = 0;
  for (int i = 0; i < m->nv; i++) s += w->qM[m->dof_Madr[i]];
  ws_free(w);
  return m->nv ? s / m->nv : 1.0;
}

/* one mj_step for one env. Derived outputs are for the PRE-integration state. */
static int step_one(const fmj_model* m, ws_t* w, double* qpos, double* qvel, const double* ctrl,
                    const double* qpos_spring, const double* xfrc, double* sensordata) {
  int warn = 0;
  if (bad(qpos, m->nq)) warn |= FMJ_WARN_BADQPOS;
  if (bad(qvel, m->nv)) warn |= FMJ_WARN_BADQVEL;
  warn |= forward(m, w, qpos, qvel, ctrl, qpos_spring, xfrc, sensordata);
  if (bad(w->qacc, m->nv)) warn |= FMJ_WARN_BADQACC;
  if (m->integrator == FMJ_INT_RK4) { warn |= rk4(m, w, qpos, qvel, ctrl, qpos_spring, xfrc); memcpy(w->tmpv, w->qacc, m->nv * sizeof(double)); }   /* the reported qacc: the last pass's */
  else euler(m, w, qpos, qvel);
  memcpy(w->qacc_warmstart, w->qacc, m->nv * sizeof(double));
  if (g_fp32_storage >= 2) { round_to_f32(qpos, m->nq); round_to_f32(qvel, m->nv); round_to_f32(w->qacc_warmstart, m->nv); }
  return warn;
}

/* ------------------------------------------------------------------------------------------ */
/* drag: reference farms_mujoco/swimming/drag.pyx (x,y,z,w quaternions, farms_core transform)    */

static void fq_conj(const double* q, double* o) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }
static void fq_mult(const double* a, const double* b, double* o, int full) {
  double t[4];
  t[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  t[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  t[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  t[3] = full ? a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2] : 0.0;
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; if (full) o[3] = t[3];
}
/* out = quat * (vector,0) * conj(quat)  (drag.pyx:50-63 usage) */
static void fq_rot(const double* v, const double* q, double* out) {
  double qc[4], v4[4] = {v[0], v[1], v[2], 0.0}, t4[4];
  fq_conj(q, qc);
  fq_mult(q, v4, t4, 1);
  fq_mult(t4, qc, out, 0);
}

/* drag_forces (drag.pyx:152-268) for one link; returns 1 if a force was written */
static int drag_link(const double* link_row, double* xfrc_row, const double* coeff /*[2,3]*/,
                     double surface, const double* water_vel, double viscosity,
                     double mass, double height, double density, double gravity, int use_buoyancy) {
  const double* pos = link_row + FMJ_LINK_COM_POS;           /* drag.pyx:189-191 */
  if (pos[2] > surface) return 0;                            /* drag.pyx:193-194 */
  const double* urdf2global = link_row + FMJ_LINK_URDF_QUAT; /* link_swimming_info, drag.pyx:43-47 */
  const double* com2global = link_row + FMJ_LINK_COM_QUAT;
  double global2urdf[4], com2urdf[4], urdf2com[4], lin[3], ang[3];
  fq_conj(urdf2global, global2urdf);
  fq_mult(global2urdf, com2global, com2urdf, 1);
  fq_conj(com2urdf, urdf2com);
  fq_rot(link_row + FMJ_LINK_COM_LINVEL, global2urdf, lin);  /* drag.pyx:50-63 */
  fq_rot(link_row + FMJ_LINK_COM_ANGVEL, global2urdf, ang);
  double buoy[3] = {0, 0, 0};
  if (use_buoyancy) {                                        /* compute_buoyancy, drag.pyx:139-149 */
    if (mass > 0 && pos[2] < surface) {
      double t[3] = {0, 0, -1000 * mass * gravity / density * fmin(fmax(surface - pos[2], 0) / height, 1)};
      fq_rot(t, global2urdf, buoy);
    }
  }
  double fluid[3];
  fq_rot(water_vel, global2urdf, fluid);                     /* drag.pyx:235-244 */
  for (int i = 0; i < 3; i++) lin[i] -= fluid[i];
  double force[3], torque[3];
  for (int i = 0; i < 3; i++) {                              /* compute_force, drag.pyx:83-88 */
    force[i] = lin[i] * lin[i];
    if (lin[i] < 0) force[i] *= -1;
    force[i] *= viscosity * coeff[i];
    force[i] += buoy[i];
  }
  for (int i = 0; i < 3; i++) {                              /* compute_torque, drag.pyx:104-108 */
    torque[i] = ang[i] * ang[i];
    if (ang[i] < 0) torque[i] *= -1;
    torque[i] *= coeff[3 + i];
  }
  fq_rot(force, urdf2com, force);                            /* drag.pyx:261-262 */
  fq_rot(torque, urdf2com, torque);
  for (int i = 0; i < 3; i++) { xfrc_row[i] = force[i]; xfrc_row[3 + i] = torque[i]; }  /* :265-267 */
  return 1;
}

/* SwimmingHandler.step (drag.pyx:389-411) for n_envs; rows batch-first fp64.
 * If xfrc_applied != NULL also writes the world-frame glue. */
int fmjo_drag(int n_envs, int n_links_rows, int n_xfrc_rows, int nbody, int ns,
              const int32_t* links_index, const int32_t* xfrc_index, const int32_t* body_index,
              const double* coefficients, const double* masses, const double* heights, const double* densities,
              double surface, const double* water_vel, double viscosity, double gravity, int use_buoyancy,
              double newtons, double torques,
              const double* links, double* xfrc, double* xfrc_applied) {
  for (int e = 0; e < n_envs; e++) {
    const double* L = links + (size_t)e * n_links_rows * FMJ_LINK_SIZE;
    double* X = xfrc + (size_t)e * n_xfrc_rows * FMJ_XFRC_SIZE;
    for (int i = 0; i < ns; i++) {
      const double* lrow = L + (size_t)links_index[i] * FMJ_LINK_SIZE;
      double* xrow = X + (size_t)xfrc_index[i] * FMJ_XFRC_SIZE;
      int applied = drag_link(lrow, xrow, coefficients + 6 * i, surface, water_vel, viscosity,
                              masses[i], heights[i], densities[i], gravity, use_buoyancy);
      if (xfrc_applied) {
        double* xa = xfrc_applied + ((size_t)e * nbody + body_index[i]) * 6;
        if (!applied) { memset(xa, 0, 6 * sizeof(double)); continue; }
        /* CoM frame == body frame (physics.py:463-466); rotate to world with com2global */
        double f[3], t[3];
        fq_rot(xrow, lrow + FMJ_LINK_COM_QUAT, f);
        fq_rot(xrow + 3, lrow + FMJ_LINK_COM_QUAT, t);
        for (int k = 0; k < 3; k++) { xa[k] = f[k] * newtons; xa[3 + k] = t[k] * torques; }
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* physics2data (reference physics.py:527-545; readers :449-466, :435-446, :481-524)             */

static void physics2data_one(const fmj_model* m, const double* qpos, const double* qvel,
                             const double* xpos, const double* xquat, const double* xipos, const double* sd,
                             int n_links, const int32_t* links_body, int n_joints, const int32_t* joints_jnt,
                             const double* units /*meters,newtons,torques,velocity,angular_velocity*/,
                             int links_only, double* links, double* joints) {
  double meters = units[0], torques = units[2], velocity = units[3], angvel = units[4];
  int lin_adr = 0;                         /* framelinvel/frameangvel interleaved per body from body 1 */
  int njs = 0; for (int j = 0; j < m->njnt; j++) njs += m->jnt_type[j] != FMJ_JNT_FREE;
  int jnt_adr = 6 * (m->nbody - 1), act_adr = jnt_adr + 3 * njs;
  for (int i = 0; i < n_links; i++) {
    int b = links_body[i]; double* r = links + (size_t)i * FMJ_LINK_SIZE;
    for (int k = 0; k < 3; k++) r[FMJ_LINK_URDF_POS + k] = xpos[3 * b + k] / meters;       /* :451-454 */
    const double* q = xquat + 4 * b;                                                           /* :455-458 wxyz->xyzw */
    r[FMJ_LINK_URDF_QUAT + 0] = q[1]; r[FMJ_LINK_URDF_QUAT + 1] = q[2]; r[FMJ_LINK_URDF_QUAT + 2] = q[3]; r[FMJ_LINK_URDF_QUAT + 3] = q[0];
    for (int k = 0; k < 3; k++) r[FMJ_LINK_COM_POS + k] = xipos[3 * b + k] / meters;        /* :459-462 */
    r[FMJ_LINK_COM_QUAT + 0] = q[1]; r[FMJ_LINK_COM_QUAT + 1] = q[2]; r[FMJ_LINK_COM_QUAT + 2] = q[3]; r[FMJ_LINK_COM_QUAT + 3] = q[0];  /* :463-466 */
    const double* s = sd + lin_adr + 6 * (b - 1);
    for (int k = 0; k < 3; k++) r[FMJ_LINK_COM_LINVEL + k] = s[k] / velocity;               /* :437-440 */
    for (int k = 0; k < 3; k++) r[FMJ_LINK_COM_ANGVEL + k] = s[3 + k] / angvel;             /* :441-446 */
  }
  if (links_only) return;
  /* sensor index of each non-free joint */
  for (int i = 0; i < n_joints; i++) {
    int j = joints_jnt[i]; double* r = joints + (size_t)i * FMJ_JOINT_SIZE;
    int sj = 0; for (int jj = 0; jj < j; jj++) sj += m->jnt_type[jj] != FMJ_JNT_FREE;
    r[FMJ_JOINT_LIMIT_FORCE] = sd[jnt_adr + 3 * sj + 2] / torques;                           /* :484-487 */
    r[FMJ_JOINT_POSITION] = qpos[m->jnt_qposadr[j]];                                         /* :502-504 */
    r[FMJ_JOINT_VELOCITY] = qvel[m->jnt_dofadr[j]] / angvel;                                 /* :505-507 */
    /* motor torque: sum of the joint's actuatorfrc sensors (:510-524); written with '=' on a
       fresh row */
    double t = 0;
    for (int a = 0; a < m->nu; a++) if (m->actuator_jntid[a] == j) t += sd[act_adr + a] * (1.0 / torques);
    r[FMJ_JOINT_TORQUE] = t;
  }
}

int fmjo_physics2data(const fmj_model* m, int n_envs, const double* qpos, const double* qvel,
                      const double* xpos, const double* xquat, const double* xipos, const double* sensordata,
                      int nsensordata, int n_links, const int32_t* links_body, int n_joints, const int32_t* joints_jnt,
                      const double* units, int links_only, double* links, double* joints) {
  for (int e = 0; e < n_envs; e++)
    physics2data_one(m, qpos + (size_t)e * m->nq, qvel + (size_t)e * m->nv, xpos + (size_t)e * m->nbody * 3,
                     xquat + (size_t)e * m->nbody * 4, xipos + (size_t)e * m->nbody * 3,
                     sensordata + (size_t)e * nsensordata, n_links, links_body, n_joints, joints_jnt, units,
                     links_only, links + (size_t)e * n_links * FMJ_LINK_SIZE,
                     joints ? joints + (size_t)e * n_joints * FMJ_JOINT_SIZE : NULL);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* contacts: mj_contactForce (pyramidal, condim 3) + cycontacts2data (reference sensors.pyx:20-190) */

#define FMJO_CONTACT_W 18   /* pos(3) frame(9) force(3: normal,t1,t2) geom1 geom2 dist */

/* the contact list of the last forward pass as records of FMJO_CONTACT_W doubles (mjContact + mj_contactForce).
 * geom1 = the plane, geom2 = the animat geom: MuJoCo orders a pair by geom type and the plane has the lowest. */
static void export_contacts(const ws_t* w, double* out) {
  for (int c = 0; c < w->ncon; c++) {
    double* o = out + (size_t)FMJO_CONTACT_W * c;
    memcpy(o, w->con_pos + 3 * c, 3 * sizeof(double)); memcpy(o + 3, w->con_frame + 9 * c, 9 * sizeof(double));
    const double* f = w->efc_force + w->con_efc[c];      /* pyramid edge forces -> contact-frame force (mju_decodePyramid) */
    double mu = w->con_mu[c];
    if (w->efc_type[w->con_efc[c]] == EFC_ELLIPTIC) { o[12] = f[0]; o[13] = f[1]; o[14] = f[2]; }      /* elliptic: the rows are the frame axes */
    else { o[12] = f[0] + f[1] + f[2] + f[3]; o[13] = mu * (f[0] - f[1]); o[14] = mu * (f[2] - f[3]); }
    o[15] = w->con_plane[c]; o[16] = w->con_geom[c]; o[17] = w->con_dist[c];
  }
}

/* geompair2data lookup: keys = n_keys x (geom_a, geom_b or -1, row); -1 if the key is absent */
static int key_row(int n_keys, const int32_t* keys, int a, int b) {
  for (int k = 0; k < n_keys; k++) if (keys[3 * k] == a && keys[3 * k + 1] == b) return keys[3 * k + 2];
  return -1;
}

/* store_forces (sensors.pyx:20-52) */
static double store_forces(double* row, const double* forcetorque, const double* frame, const double* pos, int sign) {
  double reaction[3], friction[3], total[3];
  for (int i = 0; i < 3; i++) {
    reaction[i] = sign * forcetorque[0] * frame[0 + i];                                       /* :34 */
    double friction1 = sign * forcetorque[1] * frame[3 + i], friction2 = sign * forcetorque[2] * frame[6 + i];   /* :35-36 */
    friction[i] = friction1 + friction2;                                                       /* :37 */
    total[i] = reaction[i] + friction[i];                                                      /* :38 */
  }
  for (int i = 0; i < 3; i++) {                                                                /* :39-47 */
    row[FMJ_CONTACT_REACTION + i] += reaction[i]; row[FMJ_CONTACT_FRICTION + i] += friction[i]; row[FMJ_CONTACT_TOTAL + i] += total[i];
  }
  double norm = sqrt(total[0] * total[0] + total[1] * total[1] + total[2] * total[2]);         /* :48 */
  for (int i = 0; i < 3; i++) row[FMJ_CONTACT_POSITION + i] += norm * pos[i];                  /* :49-51 */
  return norm;
}

/* cycontacts2data (sensors.pyx:140-190) for one env; rows [n_rows][12] are zeroed first (the reference adds into a
 * fresh ring-buffer row) */
static void contacts2data_one(int ncon, const double* con, int n_rows, int n_keys, const int32_t* keys,
                              double meters, double newtons, double* rows) {
  double* norm_sum = dalloc(n_rows);
  memset(rows, 0, (size_t)n_rows * FMJ_CONTACT_SIZE * sizeof(double));
  for (int c = 0; c < ncon; c++) {
    const double* ct = con + (size_t)FMJO_CONTACT_W * c;
    int geom1 = (int)ct[15], geom2 = (int)ct[16];
    const int pair[4][2] = {{geom1, geom2}, {geom2, geom1}, {geom1, -1}, {geom2, -1}};       /* :163-168 */
    const int sign[4] = {-1, +1, -1, +1};
    for (int k = 0; k < 4; k++) {
      int index = key_row(n_keys, keys, pair[k][0], pair[k][1]);                              /* :169-170 */
      if (index < 0) continue;
      norm_sum[index] += store_forces(rows + (size_t)index * FMJ_CONTACT_SIZE, ct + 12, ct + 3, ct, sign[k]);   /* :55-75 */
    }
  }
  double imeters = 1.0 / meters, inewtons = 1.0 / newtons;                                     /* :122-123 */
  for (int index = 0; index < n_rows; index++) {
    double* r = rows + (size_t)index * FMJ_CONTACT_SIZE;
    if (norm_sum[index] > 0) for (int i = 0; i < 3; i++) r[FMJ_CONTACT_POSITION + i] /= norm_sum[index];   /* :85-88 */
    for (int i = 0; i < 9; i++) r[i] *= inewtons;                                              /* :99-107 */
    for (int i = 0; i < 3; i++) r[FMJ_CONTACT_POSITION + i] *= imeters;                        /* :108-110 */
  }
  free(norm_sum);
}

int fmjo_contacts2data(int n_envs, int max_contacts, const double* contact, const int32_t* ncon, int n_rows, int n_keys,
                       const int32_t* keys, double meters, double newtons, double* rows) {
  for (int e = 0; e < n_envs; e++)
    contacts2data_one(ncon[e], contact + (size_t)e * max_contacts * FMJO_CONTACT_W, n_rows, n_keys, keys, meters, newtons,
                      rows + (size_t)e * n_rows * FMJ_CONTACT_SIZE);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* public: batched step                                                                         */

typedef struct step_job {
  const fmj_model* m; int e0, e1, n_steps; int64_t ctrl_step_stride;
  double *qpos, *qvel; const double *ctrl, *qpos_spring, *xfrc;
  double *xpos, *xquat, *xipos, *sensordata, *qacc; int32_t* status; int nsd; double meaninertia;
} step_job;

static int nsensordata(const fmj_model* m) {
  int njs = 0; for (int j = 0; j < m->njnt; j++) njs += m->jnt_type[j] != FMJ_JNT_FREE;
  return 6 * (m->nbody - 1) + 3 * njs + m->nu;
}

static void* step_worker(void* arg) {
  step_job* J = (step_job*)arg; const fmj_model* m = J->m;
  ws_t* w = ws_new(m); w->meaninertia = J->meaninertia;
  double* sd = dalloc(J->nsd);
  for (int e = J->e0; e < J->e1; e++) {
    memset(w->qacc_warmstart, 0, m->nv * sizeof(double));
    int warn = 0;
    for (int s = 0; s < J->n_steps; s++) {
      const double* ctrl = J->ctrl ? J->ctrl + (size_t)s * J->ctrl_step_stride + (size_t)e * m->nu : NULL;
      warn |= step_one(m, w, J->qpos + (size_t)e * m->nq, J->qvel + (size_t)e * m->nv, ctrl,
                       J->qpos_spring + (size_t)e * m->nq, J->xfrc ? J->xfrc + (size_t)e * m->nbody * 6 : NULL, sd);
    }
    if (J->xpos) memcpy(J->xpos + (size_t)e * m->nbody * 3, w->xpos, 3 * m->nbody * sizeof(double));
    if (J->xquat) memcpy(J->xquat + (size_t)e * m->nbody * 4, w->xquat, 4 * m->nbody * sizeof(double));
    if (J->xipos) memcpy(J->xipos + (size_t)e * m->nbody * 3, w->xipos, 3 * m->nbody * sizeof(double));
    if (J->sensordata) memcpy(J->sensordata + (size_t)e * J->nsd, sd, J->nsd * sizeof(double));
    /* the acceleration Euler actually integrated ((M+hB)^-1 f when any damping > 0), NOT mjData.qacc */
    if (J->qacc) memcpy(J->qacc + (size_t)e * m->nv, w->tmpv, m->nv * sizeof(double));
    if (J->status) J->status[e] |= warn;
  }
  free(sd); ws_free(w);
  return NULL;
}

int fmjo_nsensordata(const fmj_model* m) { return nsensordata(m); }

/* mj_step x n_steps for n_envs (batch-first fp64 arrays), n_threads pthreads over env ranges. */
int fmjo_step(const fmj_model* m, int n_envs, int n_steps, int64_t ctrl_step_stride,
              double* qpos, double* qvel, const double* ctrl, const double* qpos_spring, const double* xfrc_applied,
              double* xpos, double* xquat, double* xipos, double* sensordata, double* qacc, int32_t* status,
              int n_threads) {
  if (!m || m->abi_version != FMJ_ABI_VERSION) return FMJ_ERR_ARG;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_envs) n_threads = n_envs;
  double mi = mean_inertia(m);
  pthread_t* th = (pthread_t*)malloc(n_threads * sizeof(pthread_t));
  step_job* jobs = (step_job*)malloc(n_threads * sizeof(step_job));
  for (int t = 0; t < n_threads; t++) {
    step_job J = {m, (int)((int64_t)n_envs * t / n_threads), (int)((int64_t)n_envs * (t + 1) / n_threads), n_steps,
                  ctrl_step_stride, qpos, qvel, ctrl, qpos_spring, xfrc_applied, xpos, xquat, xipos, sensordata, qacc,
                  status, nsensordata(m), mi};
    jobs[t] = J;
    if (n_threads == 1) step_worker(&jobs[t]); else pthread_create(&th[t], NULL, step_worker, &jobs[t]);
  }
  if (n_threads > 1) for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
  return FMJ_OK;
}

/* One mj_step per env with the solver's warm start handed in and out (teacher-forced parity runs: the caller owns the whole
 * state, including qacc_warmstart) and the constraint problem of that step laid open.  Arrays are batch-first fp64; any
 * output pointer may be NULL.  warmstart [n, nv]: in = qacc of the previous step, out = this step's qacc (mjData.qacc,
 * before the implicit-damping re-solve).  counts [n, 3] = ncon, nefc, solver iterations.  efc [n, maxefc, 6] = force, b =
 * J qacc_smooth - aref, R, aref, type (0 limit / 1 contact), id (joint / contact).  efc_AR [n, maxefc, maxefc] = J M^-1 J' +
 * diag(R) (row stride maxefc), efc_J [n, maxefc, nv].  contact [n, max_contacts, FMJO_CONTACT_W].  maxefc =
 * fmjo_maxefc(m).  Single-threaded. */
int fmjo_maxefc(const fmj_model* m) { return 2 * m->njnt + 4 * (m->max_contacts > 0 ? m->max_contacts : 0); }
int fmjo_step_tf(const fmj_model* m, int n_envs, double* qpos, double* qvel, const double* ctrl, const double* qpos_spring,
                 const double* xfrc_applied, double* warmstart, double* sensordata, double* qacc_integrated, int32_t* counts,
                 double* efc, double* efc_AR, double* efc_J, double* contact, int32_t* status) {
  if (!m || m->abi_version != FMJ_ABI_VERSION) return FMJ_ERR_ARG;
  ws_t* w = ws_new(m); w->meaninertia = mean_inertia(m);
  const int nv = m->nv, nsd = nsensordata(m), me = w->maxefc, mc = m->max_contacts > 0 ? m->max_contacts : 0;
  double* sd = dalloc(nsd);
  for (int e = 0; e < n_envs; e++) {
    if (warmstart) memcpy(w->qacc_warmstart, warmstart + (size_t)e * nv, nv * sizeof(double));
    else memset(w->qacc_warmstart, 0, nv * sizeof(double));
    int warn = step_one(m, w, qpos + (size_t)e * m->nq, qvel + (size_t)e * nv, ctrl ? ctrl + (size_t)e * m->nu : NULL,
                        qpos_spring + (size_t)e * m->nq, xfrc_applied ? xfrc_applied + (size_t)e * m->nbody * 6 : NULL, sd);
    if (warmstart) memcpy(warmstart + (size_t)e * nv, w->qacc, nv * sizeof(double));
    if (sensordata) memcpy(sensordata + (size_t)e * nsd, sd, nsd * sizeof(double));
    if (qacc_integrated) memcpy(qacc_integrated + (size_t)e * nv, w->tmpv, nv * sizeof(double));
    if (counts) { counts[3 * e] = w->ncon; counts[3 * e + 1] = w->nefc; counts[3 * e + 2] = w->nefc ? g_last_pgs_iterations : 0; }
    const int n = w->nefc;
    if (efc) for (int r = 0; r < n; r++) {
      double* o = efc + ((size_t)e * me + r) * 6;
      o[0] = w->efc_force[r]; o[1] = dotn(w->efc_J + (size_t)r * nv, w->qacc_smooth, nv) - w->efc_aref[r]; o[2] = w->efc_R[r]; o[3] = w->efc_aref[r];
      o[4] = w->efc_type[r]; o[5] = w->efc_id[r];
    }
    if (efc_J) memcpy(efc_J + (size_t)e * me * nv, w->efc_J, (size_t)n * nv * sizeof(double));
    if (efc_AR) {
      /* for every solver: A = J M^-1 J' + diag(R), from the factor of M of this step */
      for (int r = 0; r < n; r++) { double* x = w->efc_MiJT + (size_t)r * nv; memcpy(x, w->efc_J + (size_t)r * nv, nv * sizeof(double)); solve_ld(m, x, w->qLD, w->qLDiagInv); }
      for (int r = 0; r < n; r++) for (int c = 0; c < n; c++)
        efc_AR[((size_t)e * me + r) * me + c] = dotn(w->efc_J + (size_t)r * nv, w->efc_MiJT + (size_t)c * nv, nv) + (r == c ? w->efc_R[r] : 0.0);
    }
    if (contact && mc) export_contacts(w, contact + (size_t)e * mc * FMJO_CONTACT_W);
    if (status) status[e] |= warn;
  }
  free(sd); ws_free(w);
  return FMJ_OK;
}

/* mj_forward internals for ONE env, for KATs: any output pointer may be NULL.
 * Mdense [nv,nv] is the full symmetric joint-space inertia. */
int fmjo_forward_debug(const fmj_model* m, const double* qpos, const double* qvel, const double* ctrl,
                       const double* qpos_spring, const double* xfrc_applied,
                       double* Mdense, double* qfrc_bias, double* qfrc_passive, double* qfrc_actuator,
                       double* qfrc_xfrc, double* qfrc_smooth, double* qacc_smooth, double* qfrc_constraint,
                       double* qacc, double* xpos, double* xquat, double* xipos, double* subtree_com,
                       double* cvel, double* sensordata, int32_t* ncon_nefc, double* efc_force, double* contact_out) {
  ws_t* w = ws_new(m); w->meaninertia = mean_inertia(m);
  double* sd = dalloc(nsensordata(m));
  forward(m, w, qpos, qvel, ctrl, qpos_spring, xfrc_applied, sd);
  int nv = m->nv;
  if (Mdense) {
    memset(Mdense, 0, (size_t)nv * nv * sizeof(double));
    for (int i = 0; i < nv; i++) { int a = m->dof_Madr[i]; for (int j = i; j >= 0; j = m->dof_parentid[j]) { Mdense[i * nv + j] = Mdense[j * nv + i] = w->qM[a++]; } }
  }
#define CP(dst, src, n) if (dst) memcpy(dst, src, (n) * sizeof(double))
  CP(qfrc_bias, w->qfrc_bias, nv); CP(qfrc_passive, w->qfrc_passive, nv); CP(qfrc_actuator, w->qfrc_actuator, nv);
  CP(qfrc_xfrc, w->qfrc_xfrc, nv); CP(qfrc_smooth, w->qfrc_smooth, nv); CP(qacc_smooth, w->qacc_smooth, nv);
  CP(qfrc_constraint, w->qfrc_constraint, nv); CP(qacc, w->qacc, nv);
  CP(xpos, w->xpos, 3 * m->nbody); CP(xquat, w->xquat, 4 * m->nbody); CP(xipos, w->xipos, 3 * m->nbody);
  CP(subtree_com, w->subtree_com, 3 * m->nbody); CP(cvel, w->cvel, 6 * m->nbody);
  CP(sensordata, sd, nsensordata(m));
  if (ncon_nefc) { ncon_nefc[0] = w->ncon; ncon_nefc[1] = w->nefc; }
  CP(efc_force, w->efc_force, w->nefc);
  if (contact_out) export_contacts(w, contact_out);   /* [ncon, FMJO_CONTACT_W] */
#undef CP
  free(sd); ws_free(w);
  return FMJ_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* fused loop: ExperimentTask.before_step + Environment.step (reference task.py:168-186,         */
/* simulation.py:155-156): physics2data(row) -> drag -> xfrc glue -> ctrl -> mj_step             */

typedef struct fused_job {
  const fmj_model* m; int e0, e1, n_envs;
  int n_steps, iteration0, buffer_size, do_readout, do_drag, controller; int64_t ctrl_step_stride;
  double *qpos, *qvel; const double *ctrl, *qpos_spring;
  double *xpos, *xquat, *xipos, *sensordata; int32_t* status;
  double *links, *joints, *xfrc;   /* [buffer, n_envs, n, width] */
  int n_links; const int32_t* links_body; int n_joints; const int32_t* joints_jnt;
  int ns; const int32_t *sw_links_index, *sw_xfrc_index, *sw_body_index;
  const double *coefficients, *masses, *heights, *densities;
  double surface, water_vel[3], viscosity, gravity; int use_buoyancy;
  const double* units;  /* meters,newtons,torques,velocity,angular_velocity */
  const double *wave_amplitude, *wave_phase_lag, *wave_env_phase; double wave_frequency;
  double meaninertia;
  int n_xfrc;                       /* rows per env of the xfrc array */
  double* contacts; int n_contact_rows, n_keys; const int32_t* keys;   /* contact sensor rows [buffer, n_envs, n_rows, 12] or NULL */
  int substeps, substep_links;      /* ExperimentTask.substeps (task.py:61) and substeps_links (task.py:63): see the loop below */
  int n_iterations_total;           /* the run's n_iterations or 0 */
} fused_job;

static void* fused_worker(void* arg) {
  fused_job* J = (fused_job*)arg; const fmj_model* m = J->m;
  int nsd = nsensordata(m), nb = m->nbody;
  ws_t* w = ws_new(m); w->meaninertia = J->meaninertia;
  double* ctrl = dalloc(m->nu);
  double* xa = dalloc(6 * nb);
  double* tmp_links = dalloc((size_t)J->n_links * FMJ_LINK_SIZE);
  double* con = dalloc((size_t)(m->max_contacts > 0 ? m->max_contacts : 1) * FMJO_CONTACT_W);
  for (int e = J->e0; e < J->e1; e++) {
    double* qpos = J->qpos + (size_t)e * m->nq; double* qvel = J->qvel + (size_t)e * m->nv;
    double* xpos = J->xpos + (size_t)e * nb * 3; double* xquat = J->xquat + (size_t)e * nb * 4;
    double* xipos = J->xipos + (size_t)e * nb * 3; double* sd = J->sensordata + (size_t)e * nsd;
    memset(w->qacc_warmstart, 0, m->nv * sizeof(double));
    memset(xa, 0, 6 * nb * sizeof(double));
    int warn = 0;
    if (J->contacts) {     /* contact list of the state the loop starts from: physics.reset's mj_forward, actuation off */
      w->disable_actuation = 1;
      forward(m, w, qpos, qvel, NULL, J->qpos_spring + (size_t)e * m->nq, NULL, NULL);
      w->disable_actuation = 0;
    }
    /* ExperimentTask's own counters, restated literally (task.py:49,64-66,168-186,348-369): sim_iteration counts environment
     * steps, `iteration` is advanced by after_step when (sim_iteration + 1) % substeps == 0 AFTER the increment - i.e. after
     * sub-step substeps - 2 of every group - and every write of before_step goes to the ring index of
     * `iteration` as it stands at that moment.  n_steps counts iterations: n_steps * substeps environment steps. */
    const int S = J->substeps > 1 ? J->substeps : 1;
    int sim_iteration = 0, iteration = J->iteration0;
    for (int s = 0; s < J->n_steps * S; s++) {
      const int full_step = (sim_iteration % S) == 0;                       /* task.py:175 */
      /* task.py:176; a sub-step whose task.iteration has reached the run's n_iterations writes no rows and keeps its drag force
       * (include/fmj.h, fmj_fused_args::n_iterations: the reference never executes it) */
      const int sensors_now = full_step || (J->substep_links && !(J->n_iterations_total > 0 && iteration >= J->n_iterations_total));
      const int links_only = !full_step;
      int it = iteration, index = it % J->buffer_size;                     /* task.py:158 */
      double* lrow = tmp_links;
      if (J->contacts && J->do_readout && sensors_now && !links_only) {   /* cycontacts2data on the contacts of the last forward pass (physics.py:533-545) */
        export_contacts(w, con);
        contacts2data_one(w->ncon, con, J->n_contact_rows, J->n_keys, J->keys, J->units[0], J->units[1],
                          J->contacts + ((size_t)index * J->n_envs + e) * J->n_contact_rows * FMJ_CONTACT_SIZE);
      }
      if (J->do_readout && sensors_now) {
        lrow = J->links + ((size_t)index * J->n_envs + e) * J->n_links * FMJ_LINK_SIZE;
        double* jrow = J->joints + ((size_t)index * J->n_envs + e) * J->n_joints * FMJ_JOINT_SIZE;
        physics2data_one(m, qpos, qvel, xpos, xquat, xipos, sd, J->n_links, J->links_body, J->n_joints,
                         J->joints_jnt, J->units, links_only, lrow, jrow);
      } else if (J->do_drag && sensors_now) {
        physics2data_one(m, qpos, qvel, xpos, xquat, xipos, sd, J->n_links, J->links_body, 0, NULL, J->units, 1, lrow, NULL);
      }
      if (J->do_drag && sensors_now) {                                     /* the swimming callback: task.py:180-182 (full step, or its substep flag) */
        double* xrow = J->xfrc + ((size_t)index * J->n_envs + e) * J->n_xfrc * FMJ_XFRC_SIZE;
        fmjo_drag(1, J->n_links, J->n_xfrc, nb, J->ns, J->sw_links_index, J->sw_xfrc_index, J->sw_body_index,
                  J->coefficients, J->masses, J->heights, J->densities, J->surface, J->water_vel, J->viscosity,
                  J->gravity, J->use_buoyancy, J->units[1], J->units[2], lrow, xrow, xa);
      }
      if (full_step) {                                                     /* step_control: task.py:184-186,288-346 */
        if (J->controller == 1) {
          double t = it * (m->timestep * S);   /* task.py:290: iteration * timestep, the timestep of an iteration (the model's is timestep / substeps: mjcf.py:1187-1192) */
          for (int a = 0; a < m->nu; a++)
            ctrl[a] = J->wave_amplitude[a] * sin(2 * M_PI * J->wave_frequency * t - J->wave_phase_lag[a] + J->wave_env_phase[e]);
        } else if (J->ctrl) {
          memcpy(ctrl, J->ctrl + (size_t)(s / S) * J->ctrl_step_stride + (size_t)e * m->nu, m->nu * sizeof(double));
        }
      }
      warn |= step_one(m, w, qpos, qvel, ctrl, J->qpos_spring + (size_t)e * m->nq, J->do_drag ? xa : NULL, sd);
      memcpy(xpos, w->xpos, 3 * nb * sizeof(double)); memcpy(xquat, w->xquat, 4 * nb * sizeof(double));
      memcpy(xipos, w->xipos, 3 * nb * sizeof(double));
      sim_iteration++;                                                     /* after_step: task.py:351-355 */
      if (((sim_iteration + 1) % S) == 0) iteration++;
    }
    if (J->status) J->status[e] |= warn;
  }
  free(ctrl); free(xa); free(tmp_links); free(con); ws_free(w);
  return NULL;
}

int fmjo_run_fused(const fmj_model* m, int n_envs, int n_steps, int iteration0, int buffer_size,
                   int do_readout, int do_drag, int controller, int64_t ctrl_step_stride,
                   double* qpos, double* qvel, const double* ctrl, const double* qpos_spring,
                   double* xpos, double* xquat, double* xipos, double* sensordata, int32_t* status,
                   double* links, double* joints, double* xfrc,
                   int n_links, const int32_t* links_body, int n_joints, const int32_t* joints_jnt,
                   int ns, const int32_t* sw_links_index, const int32_t* sw_xfrc_index, const int32_t* sw_body_index,
                   const double* coefficients, const double* masses, const double* heights, const double* densities,
                   double surface, const double* water_vel, double viscosity, double gravity, int use_buoyancy,
                   const double* units, const double* wave_amplitude, const double* wave_phase_lag,
                   const double* wave_env_phase, double wave_frequency, int n_threads,
                   int n_xfrc, double* contacts, int n_contact_rows, int n_keys, const int32_t* keys, int substeps, int substep_links,
                   int n_iterations_total) {
  if (!m || m->abi_version != FMJ_ABI_VERSION) return FMJ_ERR_ARG;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_envs) n_threads = n_envs;
  double mi = mean_inertia(m);
  pthread_t* th = (pthread_t*)malloc(n_threads * sizeof(pthread_t));
  fused_job* jobs = (fused_job*)calloc(n_threads, sizeof(fused_job));
  for (int t = 0; t < n_threads; t++) {
    fused_job* J = &jobs[t];
    J->m = m; J->e0 = (int)((int64_t)n_envs * t / n_threads); J->e1 = (int)((int64_t)n_envs * (t + 1) / n_threads);
    J->n_envs = n_envs; J->n_steps = n_steps; J->iteration0 = iteration0; J->buffer_size = buffer_size;
    J->do_readout = do_readout; J->do_drag = do_drag; J->controller = controller; J->ctrl_step_stride = ctrl_step_stride;
    J->qpos = qpos; J->qvel = qvel; J->ctrl = ctrl; J->qpos_spring = qpos_spring;
    J->xpos = xpos; J->xquat = xquat; J->xipos = xipos; J->sensordata = sensordata; J->status = status;
    J->links = links; J->joints = joints; J->xfrc = xfrc;
    J->n_links = n_links; J->links_body = links_body; J->n_joints = n_joints; J->joints_jnt = joints_jnt;
    J->ns = ns; J->sw_links_index = sw_links_index; J->sw_xfrc_index = sw_xfrc_index; J->sw_body_index = sw_body_index;
    J->coefficients = coefficients; J->masses = masses; J->heights = heights; J->densities = densities;
    J->surface = surface; if (water_vel) memcpy(J->water_vel, water_vel, sizeof J->water_vel);
    J->viscosity = viscosity; J->gravity = gravity; J->use_buoyancy = use_buoyancy; J->units = units;
    J->wave_amplitude = wave_amplitude; J->wave_phase_lag = wave_phase_lag; J->wave_env_phase = wave_env_phase;
    J->wave_frequency = wave_frequency; J->meaninertia = mi;
    J->n_xfrc = n_xfrc > 0 ? n_xfrc : n_links; J->contacts = contacts; J->n_contact_rows = n_contact_rows; J->n_keys = n_keys; J->keys = keys;
    J->substeps = substeps; J->substep_links = substeps > 1 ? substep_links : 0; J->n_iterations_total = n_iterations_total;
    if (n_threads == 1) fused_worker(J); else pthread_create(&th[t], NULL, fused_worker, J);
  }
  if (n_threads > 1) for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
  return FMJ_OK;
}

/* ---- oscillator-network controller (include/fmj.h: fmj_cpg_desc / fmj_cpg_tape), fp64, explicit Euler.
 * The reference holds only the AnimatController interface (task.py:292-346); this restates the build's own
 * device controller so that the HIP kernel has a checker. */
int fmjo_cpg_tape(const fmj_cpg_desc* d, int n_envs, int n_steps, double h, double* phase, double* amp, double* damp,
                  const double* drive, double* tape) {
  if (!d || d->n_osc < 1 || d->n_osc > 64) return FMJ_ERR_ARG;
  const int no = d->n_osc, nu = d->nu;
  const double PI = 3.14159265358979323846;
  for (int e = 0; e < n_envs; e++) {
    double* th = phase + (size_t)e * no; double* r = amp + (size_t)e * no; double* rd = damp + (size_t)e * no;
    const double dr = drive ? drive[e] : 1.0;
    for (int s = 0; s < n_steps; s++) {
      double* t = tape + ((size_t)s * n_envs + e) * nu;
      for (int u = 0; u < nu; u++) {
        double v = d->out_offset[u];
        if (d->out_a[u] >= 0) v += d->out_gain[u] * r[d->out_a[u]] * (1.0 + cos(th[d->out_a[u]]));
        if (d->out_b[u] >= 0) v -= d->out_gain[u] * r[d->out_b[u]] * (1.0 + cos(th[d->out_b[u]]));
        t[u] = v;
      }
      double dth[64];
      for (int i = 0; i < no; i++) dth[i] = 2.0 * PI * d->frequency[i] * dr;
      for (int k = 0; k < d->n_conn; k++) {
        const int i = d->conn_to[k], j = d->conn_from[k];
        dth[i] += r[j] * d->conn_weight[k] * sin(th[j] - th[i] - d->conn_bias[k]);
      }
      for (int i = 0; i < no; i++) {
        const double a = d->rate[i];
        const double rdd = a * (0.25 * a * (d->amplitude[i] - r[i]) - rd[i]);
        const double nth = th[i] + h * dth[i];
        r[i] += h * rd[i];
        rd[i] += h * rdd;
        th[i] = nth > PI ? nth - 2.0 * PI : nth;
      }
    }
  }
  return FMJ_OK;
}
