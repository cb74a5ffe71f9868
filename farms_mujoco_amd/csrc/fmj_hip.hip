// fmj_hip.hip — MI355X (gfx950) implementation of include/fmj.h.
//
// One wavefront (64 lanes) owns one environment here (models without constraints that fit half a wave run two per wave:
// fmj_dual2.inc).  Lanes play two roles, "lane = body" and "lane = dof"; cross-lane data goes through ds_bpermute, DPP,
// v_readlane and LDS, everything else stays in registers.  There is no MFMA: the largest "matrix" is the nv x nv (<= 64)
// tree-sparse joint-space inertia.  Kernel arguments are read through the kernarg segment where they are used.
//
// Per step (mj_step semantics, MuJoCo's documented pipeline restricted to what reference
// farms_mujoco/simulation/mjcf.py emits):
//   K  local joint transforms (lane=body), composed along the chains by pointer jumping (ds_bpermute)
//   C  subtree CoM by wave reduction; cinert (lane=body), cdof (lane=body -> dof slots in LDS)
//   V  joint velocity vJ; chain sums by pointer jumping: cvel, cacc (no cdof_dot array:
//      cdof_dot*qvel = cvel_parent x vJ by linearity of the motion cross product)
//   F  body force f = I a + v x* I v - F_ext (lane=body); next iteration's links row and drag
//   S  subtree sums over the contiguous DFS id range (fp64 DPP prefix-sum differences): composite inertia, force
//   Q  qfrc_smooth (lane=dof): passive + actuation - cdof . f_subtree
//   M  row i of M born in the registers of lane = dof i (one entry per depth of its chain)
//   L  L'DL by rounds of unrelated dofs, rows in registers, pivot rows through LDS; with constraints M and
//      H = M + diag(armature + h*damping) are factored together, then limits / contacts / PGS (fmj_cons_rows.inc)
//   X  two triangular sweeps (v_readlane, ds_bpermute); semi-implicit Euler; sensors
//
// Fused mode wraps this with the reference's before_step work (task.py:168-186):
// physics2data row write (physics.py:527-545), SwimmingHandler.step (drag.pyx:389-411) and the
// xfrc_applied glue, and the ctrl write (task.py:288-346), looping n_steps inside one launch.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/fmj.h"

#define FMJ_MAX_LANES 64
#define FMJ_MAXD 32         // max dof-chain length (register row length) of models with limits / contacts and of the two-env kernel
#define FMJ_MAXD_DEEP 64    // ... of the one-env kernel without constraints (long swimmers: an eel of 58 joints)
#define FMJ_MAXBD 32        // max body-chain length

static thread_local std::string g_err;
static int set_err(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_err(FMJ_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// ---------------------------------------------------------------------------------------------
// device model (fp32 tables shared by all envs)

// One elimination round of the two-env kernel: up to three lane dofs of the same depth (so none is an ancestor of
// another), deepest level first.  Masks name lanes, the same 32-bit pattern in both halves.  Read through the scalar
// cache (constant address space), one round ahead of its use.
struct DualRound {
  int p0, p1, p2, depth;              // lane dofs, their absolute depth; p1 = -1: single-member round; an absent third
                                      // member repeats p0 under empty masks
  unsigned long long anc[3];          // lanes whose dof is a proper ancestor of p_c
  int ro[3];                          // two-env kernel: byte offset of p_c's published row, p_c * dual_row_stride(rs) * 4; ro[1] = -1 in a single-member round
  int pad_[3];
};

// The same for ldl_factor of the one-env kernel: up to SIX dofs of a level per round (a centipede has five per level: one round
// trip per level instead of two); members beyond np repeat p[0] under empty masks and are skipped three at a time.
struct WideRound {
  int p[6], depth, np;
  unsigned long long anc[6];
};

struct DevModel {
  int nbody, nv, nq, nu, njnt, nM;
  int max_bdepth;     // pointer-jumping rounds = ceil(log2(longest root->body chain))
  int max_subsize;    // largest subtree (bodies)
  int rs;             // row stride of Hrow (multiple of 4, >= max dof depth + 1)
  int root_free;      // 1 if body 1 carries a free joint
  int any_stiffness, any_box, nsensordata, njs;
  int any_jpos, any_bquat, any_iquat;   // some hinge anchor off the body origin / some body frame rotated in its parent / some inertial frame rotated in its body
  int n_links, n_joints, n_xfrc, ns;
  int anc_stride;     // bytes per chain row (multiple of 4)
  float h, gx, gy, gz, mtot_inv;
  const float4* btab;         // [64][BT_STRIDE]
  const float4* dtab;         // [64][DT_STRIDE]
  const float4* atab;         // [nu][AT_STRIDE]
  const float4* stab;         // [ns][ST_STRIDE]
  const float4* gtab;         // [ngeom][GT_STRIDE]
  const float4* ptab;         // [nplane][PT_STRIDE]
  const uint8_t* b_anc;       // [nbody][anc_stride] ancestor at distance 2^r (0 = world)
  const float* hf_data;       // [hf_nrow][hf_ncol] heightfield samples (one heightfield per model)
  const float4* mesh_vert;    // [nmeshvert] hull vertices of the mesh geoms, geom frame (gtab size.x / .y: first vertex, count)
  const float4* mesh_face;    // [nmeshface] planes of their hulls (n, d: n . x <= d inside), geom frame (gtab info.w: first face, sol1.w: count): explicit pairs
  int any_polypair;           // some explicit pair involves a box, a cylinder or a mesh (include/fmj.h, ABI 6)
  int any_mesh;
  int hf_nrow, hf_ncol;
  int npair, nfl;             // explicit geom pairs; fork rows (4 per pair contact) the LDS path provides
  const float4* qtab;         // [npair][QT_STRIDE]: (geom1, geom2 bits, friction, -), (solref, solimp 0..1), (solimp 2..4, -)
  float* cons_zf;             // [n_envs][maxefc][rs] fork parts of the compact rows (HBM path)
  float* cons_rows;           // [n_envs][maxefc][8] row parameters of envs with more rows than LDS holds
  float* cons_a;              // [n_envs][maxefc][AG_LD] their PGS matrix
  float* cons_z;              // [n_envs][maxefc][rs] their compact constraint rows
  const int8_t* lcad;         // [nv][nv] depth of the deepest dof the chains of two dofs share (-1: none), padded to 4 bytes
  // ---- constraint path (joint limits, plane contacts, pyramidal cone, PGS) ----
  int cons;                   // 1 if the model has limits or collision geoms
  int ngeom, nplane, max_contacts, maxefc, solver_iterations, nvs;   // nvs = odd row stride of the Jacobian rows
  float solver_tolerance, pgs_scale, impratio_isqrt;
  int noslip_iterations; float noslip_tolerance;      // option.noslip_iterations / noslip_tolerance (mjcf.py:1392-1403): the post-pass of fmj_cons_rows.inc
  int cone;                   // FMJ_CONE_PYRAMIDAL, or FMJ_CONE_ELLIPTIC (Newton / CG only: three rows per contact, cone cost in fmj_newton.inc)
  int solver, ls_iterations;  // FMJ_SOLVER_PGS, or FMJ_SOLVER_NEWTON / FMJ_SOLVER_CG (both the NEWTON instantiation of the constraint kernel); line search
  float ls_tolerance;
  // ---- two-envs-per-wave instantiation (fmj_dual2.inc)
  int dual_ok, dual_t0;                // eligible, translational dofs carried as scalars (3 with a free root)
  int implicitfast;                    // integrator = implicitfast: the velocity gains of unclamped actuators join the damping on the diagonal of H (include/fmj.h)
  float hdamp;                         // the step by which joint damping enters H = M + diag(armature + hdamp * damping): h (mj_Euler's eulerdamp, implicitfast), 0 with RK4 (qacc = M^-1 ...)
  int cons2_ok;                        // constraints + two envs per wave (fmj_cons2.inc): limits / ground contacts, pyramidal cone, PGS
  float dual_tadd[3];                  // m_total + armature + h*damping of the translational dofs
  float dual_taddm[3];                 // m_total + armature: the same block of M itself (fmj_cons2.inc factors both)
  const struct DualRound* dual_rounds; // [dual_nround] elimination rounds of the two-env kernel (fmj_dual2.inc)
  int dual_nround;
  const struct DualRound* rounds1;     // [nround1] the same for the one-env kernel (lane = dof, all dofs)
  int nround1;
  const struct WideRound* rounds6;     // [nround6] up to six dofs per round (ldl_factor)
  int nround6;
  const uint32_t* ancl1;               // [64][rs / 4] the same table for the one-env kernel (lane = dof)
  int maxdep1;                         // deepest dof depth
  const uint32_t* dual_ancl;           // [32][rs / 4] per lane dof: 4 * (lane of its ancestor at each absolute depth), own lane elsewhere
  int dual_maxdep;                     // deepest absolute depth of a lane dof
};


// packed model tables (one pointer per family keeps kernel-argument SGPR pressure down)
#define BT_STRIDE 9    // per body: pos_mass, quat, ipos, iquat, inertia, axis_q0, jpos_k, info(int4), info2(int4)
#define DT_STRIDE 6    // per dof: info(int4), prm, act(int4: first, count, joint-sensor slot, dof parent), lim, sol0, sol1
#define AT_STRIDE 3    // per actuator (sorted by dof): prm, lim, (source index bits, -, -, -)
#define ST_STRIDE 4    // per swimming link: c0 (force coefficients, mass), c1 (torque coefficients, height), c2 (density, rows, body), c3 (mass / density, 1 / height)
#define GT_STRIDE 6    // per geom: info(int4), size, pos, quat, sol0, sol1
#define QT_STRIDE 3    // per explicit geom pair
#define PT_STRIDE 4    // per ground geom: plane (n, offset) or heightfield position; prm (friction, heightfield flag, geom id); heightfield quat; heightfield rx, ry, size z
__device__ __forceinline__ unsigned __float_as_uint_(float f) { return (unsigned)__float_as_int(f); }
__device__ __forceinline__ int4 as_int4(float4 v) { return make_int4(__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)); }
// Global-memory accesses are typed (address space 1) at the access: a pointer that reaches a kernel through the kernarg
// segment (scalar loads where it is used, see the step kernels) is a generic pointer to the compiler, and a generic
// access is a flat instruction.
#define AS1 __attribute__((address_space(1)))
#define AS4 __attribute__((address_space(4)))
template <class T> __device__ __forceinline__ const T AS1* gptr(const T* p) { return (const T AS1*)p; }
template <class T> __device__ __forceinline__ T AS1* gptr(T* p) { return (T AS1*)p; }
// float4 / float2 are classes in HIP: global accesses go through the native vector types
typedef float vf4_t __attribute__((ext_vector_type(4)));
typedef float vf2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ldg4(const float4* p, unsigned i) { const vf4_t v = ((const vf4_t AS1*)p)[i]; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 ldg4f(const float* p) { const vf4_t v = *(const vf4_t AS1*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void stg4(float AS1* p, float a, float b, float c, float d) { *(vf4_t AS1*)p = vf4_t{a, b, c, d}; }
__device__ __forceinline__ void stg2(float AS1* p, float a, float b) { *(vf2_t AS1*)p = vf2_t{a, b}; }
// generic -> global -> generic: the address-space inference then types every access made through the result as global
template <class T> __device__ __forceinline__ T* glob(T* p) { return (T*)(T AS1*)p; }
__device__ __forceinline__ float2 ldg2f(const float* p) { const vf2_t v = *(const vf2_t AS1*)p; return make_float2(v.x, v.y); }
#define BTAB(b, k) ldg4(M.btab, (unsigned)(b) * BT_STRIDE + (k))
#define BTABI(b, k) as_int4(BTAB(b, k))
#define DTAB(d, k) ldg4(M.dtab, (unsigned)(d) * DT_STRIDE + (k))
#define DTABI(d, k) as_int4(DTAB(d, k))
#define ATAB(a, k) ldg4(M.atab, (unsigned)(a) * AT_STRIDE + (k))
#define STAB(i, k) ldg4(M.stab, (unsigned)(i) * ST_STRIDE + (k))
#define GTAB(g, k) ldg4(M.gtab, (unsigned)(g) * GT_STRIDE + (k))
#define GTABI(g, k) as_int4(GTAB(g, k))
#define PTAB(p, k) ldg4(M.ptab, (unsigned)(p) * PT_STRIDE + (k))
#define QTAB(p, k) ldg4(M.qtab, (unsigned)(p) * QT_STRIDE + (k))

struct StepArgs {
  float* qpos; float* qvel; const float* ctrl; const float* qpos_spring; const float* xfrc_applied;
  float* xpos; float* xquat; float* xipos; float* sensordata; float* qacc; float* time; int* status;
  float* qacc_warmstart; float* contact; int* ncon; float* contacts_rows; float inv_newtons;
  int n_envs, n_steps, iteration0, buffer_size, do_readout, do_drag, controller, integrate, disable_actuation;
  int rows_ahead;             // fused: fmj_fused_args::rows_ahead (the two-env unconstrained kernel only)
  int n_it_total;             // fused: fmj_fused_args::n_iterations (0 = unknown): a sub-step whose task.iteration reached it writes no rows
  int substeps, sub_links;    // fused: physics steps per iteration (>= 1); sub-steps write links-only rows + drag (include/fmj.h)
  long long ctrl_step_stride, row_stride_links, row_stride_joints, row_stride_xfrc, row_stride_contacts;
  int n_contact_rows, n_pairs; const int* geom_sensor; const int* pairs;
  float* links; float* joints; float* xfrc; float* xfrc_applied_out;
  // water / units
  float surface, viscosity, wvx, wvy, wvz, wgravity; int use_buoyancy;
  float inv_meters, inv_velocity, inv_angvel, inv_torques, newtons, torques;
  // wave controller
  const float* w_amp; const float* w_lag; const float* w_env; float w_freq;
  float* ctrl_out;            // fused + wave controller: ctrl of the launch's last step (physics.data.ctrl)
  const int* env_order;       // one-env kernel: env of workgroup b (NULL: b); heavier envs first shortens a launch's tail
  int* resume;                // [n_envs] or NULL: steps of this launch already completed per env (written by fmj_cons2.inc, read by the one-env kernel)
  float* dbg_H; float* dbg_qfrc;   // fmj_forward_debug: rows of H = M + diag(armature + h damping) [n_envs][nv][rs], qfrc_smooth [n_envs][nv]
  float* dbg_efc; float* dbg_pgs;  // fmj_step_debug: constraint rows after the solve [n_envs][maxefc][8], dual-cost improvement per PGS sweep [n_envs][solver_iterations]
};

static_assert(alignof(DevModel) == 8 && alignof(StepArgs) == 8, "kernarg layout of (DevModel, StepArgs)");
#define FMJ_KARG_A_OFF ((sizeof(DevModel) + 7) & ~(size_t)7)

struct fmj_ctx {
  int device, n_envs;
  DevModel dm;
  std::vector<void*> allocs;
  size_t lds_bytes, lds_bytes_dual2, lds_bytes_cons2;
  int* d_resume;              // [n_envs] hand-over of the two-env constraint kernel to the one-env kernel
  int rk4;                    // integrator = RK4: fmj_step runs four forward launches per step (fmj_rk4_stage_kernel between them)
  float *rk_q0, *rk_v0, *rk_sv, *rk_sa, *rk_sd;      // [n_envs][nq | nv | nv | nv | nsensordata] X[0], sum B F, the sensordata the later passes may scribble on
  int solver_requested;       // fmj_model.solver as handed in (fmj_create may run the dual solver instead: fmj_solver_info)
  int dual_wps;               // waves per SIMD the dual2 build is registered for: 4, or 3 when the batch cannot fill more (FMJ_WPS overrides)
  fmj_sensor_layout_t layout;
  // host copies needed later
  std::vector<int> body_link_row, dof_joint_row, body_swim;
  std::vector<int> h_b_info2;     // mutable table mirror
  std::vector<int> h_d_info;
  float4* d_btab; float4* d_dtab; std::vector<float4> h_btab, h_dtab;
  int nbody, nv, nu, njnt;
  std::vector<int> jnt_dofadr, jnt_type;
  int ngeom, n_contact_rows, n_pairs; std::vector<int> geom_sensor, geom_is_plane; int* d_geom_sensor; int* d_pairs;
  int* d_links_body; int* d_joints_dof;      // row -> body / dof maps of the standalone readout operator
  std::vector<float4> h_atab; std::vector<int> a_src;   // actuator table mirror (fmj_set_actuator_forcerange)
};

// ---------------------------------------------------------------------------------------------
// device math

struct v3 { float x, y, z; };
struct q4 { float w, x, y, z; };

__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
__device__ __forceinline__ v3 add3(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 sub3(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 scl3(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ q4 qmul(q4 a, q4 b) {
  q4 r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return r;
}
// v' = q v q*  via t = 2 qv x v ; v' = v + w t + qv x t
__device__ __forceinline__ v3 qrot(q4 q, v3 v) {
  v3 qv = mk3(q.x, q.y, q.z);
  v3 t = scl3(cross(qv, v), 2.0f);
  return add3(add3(v, scl3(t, q.w)), cross(qv, t));
}
// 1 / sqrt(x): v_rsq_f32 (1 ulp) + one Newton step, ~6 VALU; the IEEE sqrt and divide sequences cost ~35
__device__ __forceinline__ float rsqrt_nr(float x) { const float r = __builtin_amdgcn_rsqf(x); return r * fmaf(-0.5f * x * r, r, 1.5f); }
// 1 / x in double from the fp32 reciprocal and two Newton steps (the IEEE fp64 divide is ~20 instructions)
__device__ __forceinline__ double rcp_f64_nr(double x) {
  double r = (double)__builtin_amdgcn_rcpf((float)x);
  r = r * (2.0 - x * r);
  return r * (2.0 - x * r);
}
// sin(2 pi t), t in revolutions: exact reduction to [-1/2, 1/2] (v_rndne), fold to [0, 1/4], odd polynomial of degree 11 on
// [0, pi/2] (truncation 6e-8): ~15 VALU against ~45 for the library sinf with its radian argument reduction
__device__ __forceinline__ float sin_turns(float t) {
  t -= rintf(t);
  const float a = fabsf(t);
  const float x = 6.2831853071795865f * (a > 0.25f ? 0.5f - a : a), x2 = x * x;
  const float s = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, -1.0f / 39916800.0f, 1.0f / 362880.0f), -1.0f / 5040.0f), 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
  return copysignf(s, t);
}
__device__ __forceinline__ q4 qnormalize(q4 q) {
  float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  if (n2 < 1e-30f) { q4 r = {1.f, 0.f, 0.f, 0.f}; return r; }
  const float inv = rsqrt_nr(n2);
  q4 r = {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
  return r;
}
__device__ __forceinline__ q4 axisangle(v3 ax, float ang) {
  float s, c;
  sincosf(0.5f * ang, &s, &c);
  q4 r = {c, ax.x * s, ax.y * s, ax.z * s};
  return r;
}
// the free joint turns by h*|omega| per step: for half-angles below 0.25 rad the Taylor polynomials are exact to
// fp32 (sin: x^9/9! < 1e-11, cos: x^10/10! < 3e-13) and cost a tenth of the range-reduced sincosf
__device__ __forceinline__ q4 axisangle_small(v3 ax, float ang) {
  const float x = 0.5f * ang, x2 = x * x;
  if (!(fabsf(x) <= 0.25f)) return axisangle(ax, ang);
  const float s = x * fmaf(x2, fmaf(x2, fmaf(x2, -1.0f / 5040.0f, 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
  const float c = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 1.0f / 40320.0f, -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
  q4 r = {c, ax.x * s, ax.y * s, ax.z * s};
  return r;
}
// hinge angles: for half-angles up to pi/2 (|q| <= pi) the Taylor polynomials of degree 11 / 12 are exact to fp32 (sin: x^13/13! <
// 6e-8, cos: x^14/14! < 7e-9) and cost a third of the range-reduced sincosf, which remains the path for larger angles
__device__ __forceinline__ q4 axisangle_mid(v3 ax, float ang) {
  const float x = 0.5f * ang, x2 = x * x;
  if (!(fabsf(x) <= 1.5707964f)) return axisangle(ax, ang);
  const float s = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, -1.0f / 39916800.0f, 1.0f / 362880.0f), -1.0f / 5040.0f), 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
  const float c = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 1.0f / 479001600.0f, -1.0f / 3628800.0f), 1.0f / 40320.0f), -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
  q4 r = {c, ax.x * s, ax.y * s, ax.z * s};
  return r;
}
struct m33 { float a[9]; };
__device__ __forceinline__ m33 q2m(q4 q) {
  m33 m;
  float q00 = q.w * q.w, q11 = q.x * q.x, q22 = q.y * q.y, q33 = q.z * q.z;
  m.a[0] = q00 + q11 - q22 - q33; m.a[4] = q00 - q11 + q22 - q33; m.a[8] = q00 - q11 - q22 + q33;
  m.a[1] = 2.f * (q.x * q.y - q.w * q.z); m.a[2] = 2.f * (q.x * q.z + q.w * q.y);
  m.a[3] = 2.f * (q.x * q.y + q.w * q.z); m.a[5] = 2.f * (q.y * q.z - q.w * q.x);
  m.a[6] = 2.f * (q.x * q.z - q.w * q.y); m.a[7] = 2.f * (q.y * q.z + q.w * q.x);
  return m;
}
__device__ __forceinline__ v3 mrot(const m33& m, v3 v) {
  return mk3(fmaf(m.a[0], v.x, fmaf(m.a[1], v.y, m.a[2] * v.z)),
             fmaf(m.a[3], v.x, fmaf(m.a[4], v.y, m.a[5] * v.z)),
             fmaf(m.a[6], v.x, fmaf(m.a[7], v.y, m.a[8] * v.z)));
}
// spatial vectors: [rot; lin]
struct s6 { v3 r, l; };
__device__ __forceinline__ s6 s6add(s6 a, s6 b) { s6 o = {add3(a.r, b.r), add3(a.l, b.l)}; return o; }
__device__ __forceinline__ s6 s6scl(s6 a, float s) { s6 o = {scl3(a.r, s), scl3(a.l, s)}; return o; }
__device__ __forceinline__ float s6dot(s6 a, s6 b) { return dot3(a.r, b.r) + dot3(a.l, b.l); }
// motion cross  vel x v
__device__ __forceinline__ s6 cross_motion(s6 vel, s6 v) {
  s6 o = {cross(vel.r, v.r), add3(cross(vel.r, v.l), cross(vel.l, v.r))};
  return o;
}
// force cross  vel x* f
__device__ __forceinline__ s6 cross_force(s6 vel, s6 f) {
  s6 o = {add3(cross(vel.r, f.r), cross(vel.l, f.l)), cross(vel.r, f.l)};
  return o;
}
// cinert(10) * v : i = [Ixx Iyy Izz Ixy Ixz Iyz mdx mdy mdz m]
__device__ __forceinline__ s6 inert_mul(const float* i, s6 v) {
  s6 o;
  o.r.x = i[0] * v.r.x + i[3] * v.r.y + i[4] * v.r.z - i[8] * v.l.y + i[7] * v.l.z;
  o.r.y = i[3] * v.r.x + i[1] * v.r.y + i[5] * v.r.z + i[8] * v.l.x - i[6] * v.l.z;
  o.r.z = i[4] * v.r.x + i[5] * v.r.y + i[2] * v.r.z - i[7] * v.l.x + i[6] * v.l.y;
  o.l.x = i[8] * v.r.y - i[7] * v.r.z + i[9] * v.l.x;
  o.l.y = i[6] * v.r.z - i[8] * v.r.x + i[9] * v.l.y;
  o.l.z = i[7] * v.r.x - i[6] * v.r.y + i[9] * v.l.z;
  return o;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float bcast(float v, int lane) {   // lane must be wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// Hides a value from loop-invariant code motion: per-body model constants are re-read from the
// (L1/L2-resident) tables inside every step instead of pinning ~40 VGPRs across the whole loop.
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float pinf(float x) { asm volatile("" : "+v"(x)); return x; }
// the same for a wave-uniform value: conditions made from it (f < nefc for 60 columns, lvl < maxdep for 20 levels) are otherwise
// computed once, kept as SGPR-pair masks for every later use and spilled
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }
// LDS ordering between lanes of the one wave that forms the workgroup
// (every kernel here runs 64-thread workgroups = one wave: LDS operations of one wave execute in order, so ordering
// between lanes needs no s_barrier and no s_waitcnt, only that the compiler keeps the program order)
#ifdef FMJ_FULL_BARRIERS
#define WSYNC() __syncthreads()
#else
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                     __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
#define WSYNC_LOCAL() __builtin_amdgcn_wave_barrier()

__device__ __forceinline__ void lds_put6(float* p, s6 v) {
  *(float4*)p = make_float4(v.r.x, v.r.y, v.r.z, v.l.x);
  *(float2*)(p + 4) = make_float2(v.l.y, v.l.z);
}
__device__ __forceinline__ s6 lds_get6(const float* p) {
  float4 a = *(const float4*)p; float2 b = *(const float2*)(p + 4);
  s6 v = {mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y)};
  return v;
}

// farms_core transform helpers on x,y,z,w quaternions (reference drag.pyx:9 cimports)
struct fq { float x, y, z, w; };
__device__ __forceinline__ fq fq_conj(fq q) { fq r = {-q.x, -q.y, -q.z, q.w}; return r; }
__device__ __forceinline__ fq fq_mult(fq a, fq b) {
  fq o;
  o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return o;
}
__device__ __forceinline__ v3 fq_rot(v3 v, fq q) {   // q * (v,0) * conj(q)
  fq v4 = {v.x, v.y, v.z, 0.f};
  fq t = fq_mult(fq_mult(q, v4), fq_conj(q));
  return mk3(t.x, t.y, t.z);
}

// drag_forces for one link (reference drag.pyx:152-268). Row values are already unit-scaled.
// Returns false when the link is above the surface (no write, drag.pyx:192-194).
__device__ __forceinline__ bool drag_link(v3 com_pos, fq urdf2global, fq com2global, v3 lin_w, v3 ang_w,
                                          float4 c0, float4 c1, float density, const StepArgs& A,
                                          v3* force_out, v3* torque_out) {
  if (com_pos.z > A.surface) return false;
  fq global2urdf = fq_conj(urdf2global);                      // drag.pyx:45
  fq com2urdf = fq_mult(global2urdf, com2global);             // :46
  fq urdf2com = fq_conj(com2urdf);                            // :47
  v3 lin = fq_rot(lin_w, global2urdf);                        // :50-56
  v3 ang = fq_rot(ang_w, global2urdf);                        // :57-63
  v3 buoy = mk3(0.f, 0.f, 0.f);
  float mass = c0.w, height = c1.w;
  if (A.use_buoyancy && mass > 0.f && com_pos.z < A.surface) {   // :139-146
    float fz = -1000.f * mass * A.wgravity / density * fminf(fmaxf(A.surface - com_pos.z, 0.f) / height, 1.f);
    buoy = fq_rot(mk3(0.f, 0.f, fz), global2urdf);
  }
  v3 fluid = fq_rot(mk3(A.wvx, A.wvy, A.wvz), global2urdf);   // :235-244
  lin = sub3(lin, fluid);
  v3 f, t;
  f.x = lin.x * lin.x; if (lin.x < 0.f) f.x = -f.x; f.x = f.x * (A.viscosity * c0.x) + buoy.x;   // :83-88
  f.y = lin.y * lin.y; if (lin.y < 0.f) f.y = -f.y; f.y = f.y * (A.viscosity * c0.y) + buoy.y;
  f.z = lin.z * lin.z; if (lin.z < 0.f) f.z = -f.z; f.z = f.z * (A.viscosity * c0.z) + buoy.z;
  t.x = ang.x * ang.x; if (ang.x < 0.f) t.x = -t.x; t.x *= c1.x;                                   // :104-108
  t.y = ang.y * ang.y; if (ang.y < 0.f) t.y = -t.y; t.y *= c1.y;
  t.z = ang.z * ang.z; if (ang.z < 0.f) t.z = -t.z; t.z *= c1.z;
  *force_out = fq_rot(f, urdf2com);                           // :261-262
  *torque_out = fq_rot(t, urdf2com);
  return true;
}

// Same law for the fused loop, where the link's CoM orientation IS its body orientation (reference
// physics.py:455-466 fills both from xquat), so com2urdf is the identity (drag.pyx:46-47,261-262 become no-ops)
// and every quaternion sandwich q v q* collapses to one rotation matrix: ~80 VALU instead of ~400.
template <class AT>
__device__ __forceinline__ bool drag_link_same_frames(v3 com_pos, q4 q_wxyz, v3 lin_w, v3 ang_w, float4 c0, float4 c1,
                                                      float density, AT& A, v3* force_link, v3* force_w, v3* torque_link, v3* torque_w) {
  if (com_pos.z > A.surface) return false;                     // drag.pyx:192-194
  const m33 R = q2m(q_wxyz);                                   // link -> world
  // world -> link = R^T  (drag.pyx:50-63, 235-244)
  v3 lin = mk3(R.a[0] * lin_w.x + R.a[3] * lin_w.y + R.a[6] * lin_w.z, R.a[1] * lin_w.x + R.a[4] * lin_w.y + R.a[7] * lin_w.z,
               R.a[2] * lin_w.x + R.a[5] * lin_w.y + R.a[8] * lin_w.z);
  const v3 ang = mk3(R.a[0] * ang_w.x + R.a[3] * ang_w.y + R.a[6] * ang_w.z, R.a[1] * ang_w.x + R.a[4] * ang_w.y + R.a[7] * ang_w.z,
                     R.a[2] * ang_w.x + R.a[5] * ang_w.y + R.a[8] * ang_w.z);
  v3 buoy = mk3(0.f, 0.f, 0.f);
  const float mass = c0.w, height = c1.w;
  if (A.use_buoyancy && mass > 0.f && com_pos.z < A.surface) {  // drag.pyx:139-146
    const float fz = -1000.f * mass * A.wgravity / density * fminf(fmaxf(A.surface - com_pos.z, 0.f) / height, 1.f);
    buoy = mk3(R.a[6] * fz, R.a[7] * fz, R.a[8] * fz);
  }
  if (A.wvx != 0.f || A.wvy != 0.f || A.wvz != 0.f) {
    lin.x -= R.a[0] * A.wvx + R.a[3] * A.wvy + R.a[6] * A.wvz;
    lin.y -= R.a[1] * A.wvx + R.a[4] * A.wvy + R.a[7] * A.wvz;
    lin.z -= R.a[2] * A.wvx + R.a[5] * A.wvy + R.a[8] * A.wvz;
  }
  v3 f, t;
  f.x = fabsf(lin.x) * lin.x * (A.viscosity * c0.x) + buoy.x;   // sign(v) v^2 c visc + buoyancy (drag.pyx:83-88)
  f.y = fabsf(lin.y) * lin.y * (A.viscosity * c0.y) + buoy.y;
  f.z = fabsf(lin.z) * lin.z * (A.viscosity * c0.z) + buoy.z;
  t.x = fabsf(ang.x) * ang.x * c1.x; t.y = fabsf(ang.y) * ang.y * c1.y; t.z = fabsf(ang.z) * ang.z * c1.z;   // :104-108
  *force_link = f; *torque_link = t;
  *force_w = mrot(R, f); *torque_w = mrot(R, t);
  return true;
}

// ---------------------------------------------------------------------------------------------
// the step kernel

struct Carry {          // mjData derived fields owned by lane=body (state before the last integration)
  v3 xpos; q4 xquat; v3 xipos; v3 linvel, angvel;
};

__host__ __device__ inline int r4(int x) { return (x + 3) & ~3; }

// LDS layout in floats; shared by host (size) and device (carve)
struct LdsLayout {
  int P1, CI, CD, HR, QP, QV, XV, VT, ANC, total;
  int HM, YJ, EP, CT, XS, WW, QW, DI, SD, PO, AT, LC, CH, na;      // constraint path only
  int YF, CF, nfl;                                                  // explicit pairs: fork parts of their rows, fork chain per row
};
#define FMJ_NFL 16     // fork rows (4 pair contacts) kept on chip: with 32 the walker with pairs took 21.3 KB of LDS, 7 workgroups per CU instead of 8
#define AG_LD 192      // row length of the global PGS matrix (three 64-lane slots)
#define FMJ_NA 64      // constraint rows handled with one row per lane and A in registers (every lane a row)
__host__ __device__ inline LdsLayout lds_layout(int nb, int nv, int nq, int rs, int anc_stride, int cons = 0, int maxefc = 0,
                                                int maxcon = 0, int nvs = 0, int npair = 0) {
  LdsLayout L;
  const int nmax = nb > nv ? nb : nv;
  int o = 0;
  // every region starts on a 16-byte boundary (float4 LDS accesses; a misaligned ds_read_b128 is split and stalls)
  L.HR = o; if (cons) o += nv * rs;   // depth-indexed rows of the L'DL of H = M + h B (without constraints they overlay CD / F / CI, see below)
  L.QP = o; o += r4(nq);
  L.QV = o; o += r4(nv);
  L.XV = o; o += r4(nv);
  L.VT = o; o += 8;
  L.ANC = o; o += r4(r4(nb * anc_stride) / 4);
  L.HM = L.YJ = L.EP = L.CT = L.XS = L.WW = L.QW = L.DI = L.SD = L.PO = L.AT = L.LC = L.CH = o; L.na = 0;
  L.YF = L.CF = o; L.nfl = 0;
  const int dead = nmax * 8 + r4(nb * 12);       // F, CI: not live between the M phase and the next step
  if (cons) {
    L.HM = o; o += nv * rs;           // rows of M, then its L'DL
    L.CT = o; o += maxcon * 16;       // contacts: pos(3) normal(3) t1(3) t2(3) dist mu geom plane
    L.XS = o; L.WW = o; o += r4(nv);  // qacc_smooth
    L.QW = o; o += r4(nv);            // qacc_warmstart
    L.DI = o; o += r4(nv);            // 1/D of the M factor
    L.SD = o; o += r4(nv);            // 1/sqrt(D)
    L.PO = o; o += nb * 8;            // body poses: xpos(3) -, xquat(4)
    L.LC = o; o += r4((nv * nv + 3) / 4);        // int8 [nv][nv]: depth of the deepest dof two chains share (-1: none)
    L.CH = o; o += r4((maxefc + 3) / 4);         // uint8 per row: last dof of the row's chain + 1
    L.na = maxefc < FMJ_NA ? maxefc : FMJ_NA;    // rows kept on chip ("small"); larger row sets live in HBM, staged through YJ
    L.EP = o; o += L.na * 8;                      // per row: -, aref, R, b, force, R0, type|id, mu
    if (npair > 0) {
      int nf = 4 * (npair < maxcon ? npair : maxcon);
      L.nfl = nf < FMJ_NFL ? nf : FMJ_NFL;
      L.YF = o; o += L.nfl * rs;                  // fork part of the rows of pair contacts (the second body's branch)
      L.CF = o; o += r4((maxefc + 3) / 4);        // uint8 per row: last dof of the fork's chain + 1 (0: no fork)
    }
  }
  const int r1 = o;
  L.CD = o; o += nv * 8;              // cdof
  L.P1 = o; o += nmax * 8;            // F (body force -> subtree force)
  L.CI = o; o += r4(nb * 12);         // subtree inertia about its own CoM (6), subtree CoM (3), subtree mass
  if (!cons) {
    // the rows of H are born in registers (M phase) and only reach LDS when the factorisation publishes them: CD / F / CI
    // are dead by then
    L.HR = r1;
    if (r1 + nv * rs > o) o = r1 + nv * rs;
  }
  if (cons) {
    // the compact constraint rows (na x rs) overlay F and CI, which are dead from the Jacobian rows on
    L.YJ = L.P1;
    const int extra = L.na * rs - dead;
    L.AT = o; if (extra > 0) o += r4(extra);
  }
  o = r4(o);
  L.total = o;
  return L;
}

// ---- wave-wide inclusive prefix sum in fp64 with DPP (no LDS): row_shr 1/2/4/8 inside 16-lane rows, then
// row_bcast:15 / row_bcast:31 across rows (GFX9 DPP controls; same sequence LLVM's atomic optimizer emits).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  // full row mask: lanes without a source read 0 through bound_ctrl and the destination needs no initialisation (2 v_mov per
  // step saved); a partial row mask keeps `old` = 0 in the rows it leaves out
  constexpr bool BC = ROWMASK == 0xF;
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xF, BC);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xF, BC);
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double wave_prefix_f64(double v) {
  v += dpp_f64<0x111, 0xF>(v);
  v += dpp_f64<0x112, 0xF>(v);
  v += dpp_f64<0x114, 0xF>(v);
  v += dpp_f64<0x118, 0xF>(v);
  v += dpp_f64<0x142, 0xA>(v);
  v += dpp_f64<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ double lane_gather_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// Sum of x over the contiguous DFS subtree [lane, last] from the inclusive prefix P: P[last] - P[lane] + x
__device__ __forceinline__ double subtree_sum_f64(double x, int last) {
  const double P = wave_prefix_f64(x);
  return lane_gather_f64(P, last) - P + x;
}

// ---- sparse L'DL on depth-indexed rows (MuJoCo's mj_factorI / mj_solveLD on the dof tree) --------------------
// Lane i owns row i in registers (UNSCALED: entries are L*D). At pivot k lane k publishes its final row in HR,
// every lane reads it back in one batch of LDS reads and the ancestors of k update their rows.  The diagonal is
// tracked in its own register so D_k comes from a v_readlane before the LDS round trip.  Slots past a lane's
// depth only ever hold finite garbage that is never read as a matrix entry (the caller zero-fills HR before
// assembling the matrix).  On return HR holds the final rows (L*D) and dinv = 1/D_lane.
// 1/x to fp32 accuracy: hardware reciprocal (1 ulp) refined by one Newton step
__device__ __forceinline__ float rcp_nr(float x) { const float r = __builtin_amdgcn_rcpf(x); return r * (2.0f - x * r); }
// v where the lane's bit in the (uniform) mask is set, else 0
__device__ __forceinline__ float mask_select(const float v, const unsigned long long m) {
  float r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
  return r;
}

typedef const DualRound __attribute__((address_space(4)))* cround_p;   // constant address space: uniform index -> s_load

typedef float f2_t __attribute__((ext_vector_type(2)));
// acc -= t * v on a float pair, in place.  Tied operands: the rows of the factorisation are updated inside the branches of a
// uniform switch (groups per round); left to the compiler each branch puts its results where it likes and every join costs a
// copy of the whole register row (~36 v_mov per round measured), with a tied accumulator the row never moves.
__device__ __forceinline__ void pk_fnma(f2_t& acc, const f2_t tt, const f2_t v) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(acc) : "v"(tt), "v"(v));
}
template <int MAXD>
__device__ __forceinline__ void ldl_factor(float* HR, float* DV, const WideRound* rounds, int nround, int maxdep, int lane, bool isd, int ddepth,
                                           f2_t (&r)[MAXD / 2], float diag, float& dinv_mine, float& x) {
  constexpr int RS = MAXD;
  // r = the lane's row in registers as float pairs (the update is v_pk_fma_f32), diag its diagonal entry (M phase)
  // Rounds of unrelated dofs (same depth, deepest level first, <= 6 per round; records built at fmj_create and read
  // through the scalar cache one round ahead): every lane publishes its working row and 1/diag (DV) - only the
  // members' are read, theirs are final - and every proper ancestor i of a member k (uniform lane mask) does
  // row_i -= (row_k[depth_i] / D_k) row_k.  One LDS round trip per tree level instead of one per dof.
  // x: a right-hand side whose leaves-first sweep rides in the rounds (when dof k is a pivot its x_k is complete: every proper
  // ancestor i does x_i -= L[k][i] x_k next to its row update); on return x holds L^-T-swept rhs, to be scaled by 1 / D and pulled.
  // A pivot at depth d has entries in slots 0 .. d - 1 only, so a round publishes, reads and applies ceil(d / 4) float4 groups:
  // one loop per group count, run one after the other (rounds come deepest level first).
  (void)maxdep;
  typedef const WideRound __attribute__((address_space(4)))* wround_p;
  const wround_p RND = (wround_p)rounds;
#define APPLY_PIVOT(NG_, p_, am_) do { \
    const float tk_ = HR[(p_) * RS + ddepth]; const float dki_ = DV[p_]; \
    float4 rk_[NG_]; \
    _Pragma("unroll") for (int g = 0; g < NG_; g++) rk_[g] = *(const float4*)(HR + (p_) * RS + 4 * g); \
    const float t_ = mask_select(tk_ * dki_, am_); \
    const f2_t tt_ = f2_t{t_, t_}; \
    _Pragma("unroll") for (int g = 0; g < NG_; g++) { \
      pk_fnma(r[2 * g], tt_, f2_t{rk_[g].x, rk_[g].y}); \
      pk_fnma(r[2 * g + 1], tt_, f2_t{rk_[g].z, rk_[g].w}); } \
    diag = fmaf(-t_, tk_, diag); \
    x = fmaf(-t_, bcast(x, p_), x); } while (0)
#define ROUND_BODY(NG_) do { \
    if (isd) { \
      _Pragma("unroll") for (int d = 0; d < 4 * NG_; d += 4) *(float4*)(HR + lane * RS + d) = make_float4(r[d / 2].x, r[d / 2].y, r[d / 2 + 1].x, r[d / 2 + 1].y); \
      DV[lane] = __builtin_amdgcn_rcpf(diag); \
    } \
    WSYNC(); \
    APPLY_PIVOT(NG_, p0, a0); \
    if (np > 1) { APPLY_PIVOT(NG_, p1, a1); APPLY_PIVOT(NG_, p2, a2); } \
    if (np > 3) { APPLY_PIVOT(NG_, p3, a3); APPLY_PIVOT(NG_, p4, a4); APPLY_PIVOT(NG_, p5, a5); } \
    WSYNC(); } while (0)
  {
    int p0 = RND[0].p[0], p1 = RND[0].p[1], p2 = RND[0].p[2], p3 = RND[0].p[3], p4 = RND[0].p[4], p5 = RND[0].p[5], dep = RND[0].depth, np = RND[0].np;
    unsigned long long a0 = RND[0].anc[0], a1 = RND[0].anc[1], a2 = RND[0].anc[2], a3 = RND[0].anc[3], a4 = RND[0].anc[4], a5 = RND[0].anc[5];
    int rd = 0;
#define ROUNDS_AT(NG_, COND_) \
    _Pragma("unroll 1") while (rd < nround && (COND_)) { \
      const int rn = rd + 1 < nround ? rd + 1 : rd; \
      const int np0 = RND[rn].p[0], np1 = RND[rn].p[1], np2 = RND[rn].p[2], np3 = RND[rn].p[3], np4 = RND[rn].p[4], np5 = RND[rn].p[5], ndep = RND[rn].depth, nnp = RND[rn].np; \
      const unsigned long long na0 = RND[rn].anc[0], na1 = RND[rn].anc[1], na2 = RND[rn].anc[2], na3 = RND[rn].anc[3], na4 = RND[rn].anc[4], na5 = RND[rn].anc[5]; \
      ROUND_BODY(NG_); \
      p0 = np0; p1 = np1; p2 = np2; p3 = np3; p4 = np4; p5 = np5; a0 = na0; a1 = na1; a2 = na2; a3 = na3; a4 = na4; a5 = na5; dep = ndep; np = nnp; rd++; \
    }
    if (MAXD >= 64) ROUNDS_AT((MAXD >= 64 ? 16 : 1), dep > 60)
    if (MAXD >= 60) ROUNDS_AT((MAXD >= 60 ? 15 : 1), dep > 56)
    if (MAXD >= 56) ROUNDS_AT((MAXD >= 56 ? 14 : 1), dep > 52)
    if (MAXD >= 52) ROUNDS_AT((MAXD >= 52 ? 13 : 1), dep > 48)
    if (MAXD >= 48) ROUNDS_AT((MAXD >= 48 ? 12 : 1), dep > 44)
    if (MAXD >= 44) ROUNDS_AT((MAXD >= 44 ? 11 : 1), dep > 40)
    if (MAXD >= 40) ROUNDS_AT((MAXD >= 40 ? 10 : 1), dep > 36)
    if (MAXD >= 36) ROUNDS_AT((MAXD >= 36 ? 9 : 1), dep > 32)
    if (MAXD >= 32) ROUNDS_AT((MAXD >= 32 ? 8 : 1), dep > 28)
    if (MAXD >= 28) ROUNDS_AT((MAXD >= 28 ? 7 : 1), dep > 24)
    if (MAXD >= 24) ROUNDS_AT((MAXD >= 24 ? 6 : 1), dep > 20)
    if (MAXD >= 20) ROUNDS_AT((MAXD >= 20 ? 5 : 1), dep > 16)
    if (MAXD >= 16) ROUNDS_AT((MAXD >= 16 ? 4 : 1), dep > 12)
    if (MAXD >= 12) ROUNDS_AT((MAXD >= 12 ? 3 : 1), dep > 8)
    if (MAXD >= 8) ROUNDS_AT((MAXD >= 8 ? 2 : 1), dep > 4)
    ROUNDS_AT(1, true)
#undef ROUNDS_AT
  }
#undef ROUND_BODY
#undef APPLY_PIVOT
  dinv_mine = isd ? __builtin_amdgcn_rcpf(diag) : 0.f;
  // every lane's register row is final since its own round: publish it once more, scaled by 1/D, so that
  // HR holds the unit-triangular factor L itself (diagonal slot = 1) and the solves need no per-entry scaling
  if (isd) {
#pragma unroll
    for (int d = 0; d < MAXD; d += 4)
      *(float4*)(HR + lane * RS + d) = make_float4(r[d / 2].x * dinv_mine, r[d / 2].y * dinv_mine, r[d / 2 + 1].x * dinv_mine, r[d / 2 + 1].y * dinv_mine);
  }
  WSYNC();
}

// The constraint instantiation needs two factorisations per step with the same sparsity: M (constraint rows, qacc_smooth)
// and H = M + diag(armature + h damping) (the implicit-damping solve).  They are eliminated together, round by round:
// one LDS round trip per tree level serves both, and the leaves-first sweep of the solve M x = rhs rides in the rounds
// (when dof k is a pivot its x_k is complete: every proper ancestor i does x_i -= L[k][i] x_k).  A pivot at depth d has
// entries in slots 0 .. d - 1 only, so a round moves ceil(d / 4) float4 groups, picked by a uniform switch (static
// register indices).  On return HM / HR hold the unit-triangular factors, x the swept right-hand side of the M system.
template <int MAXD>
__device__ __forceinline__ void ldl_factor2(float* HM, float* HR, float* DVM, float* DVH, const DualRound* rounds, int nround, int lane,
                                            bool isd, int ddepth, f2_t (&rh)[MAXD / 2], float dgm, float dgh, float& dinv_m, float& dinv_h, float& x) {
  constexpr int RS = MAXD;
  // rh = the lane's row of M (M phase; M and H differ on the diagonal only: dgm, dgh), worked on in place for H
  f2_t rm[MAXD / 2];
#pragma unroll
  for (int d = 0; d < MAXD / 2; d++) rm[d] = rh[d];
  const cround_p RND = (cround_p)rounds;
#define APPLY_PIVOT2(NG_, p_, am_) do { \
    const float tkm_ = HM[(p_) * RS + ddepth], tkh_ = HR[(p_) * RS + ddepth]; const float dkm_ = DVM[p_], dkh_ = DVH[p_]; \
    float4 km_[NG_], kh_[NG_]; \
    _Pragma("unroll") for (int g = 0; g < NG_; g++) { km_[g] = *(const float4*)(HM + (p_) * RS + 4 * g); kh_[g] = *(const float4*)(HR + (p_) * RS + 4 * g); } \
    const float tm_ = mask_select(tkm_ * dkm_, am_), th_ = mask_select(tkh_ * dkh_, am_); \
    const f2_t ttm_ = f2_t{tm_, tm_}, tth_ = f2_t{th_, th_}; \
    _Pragma("unroll") for (int g = 0; g < NG_; g++) { \
      pk_fnma(rm[2 * g], ttm_, f2_t{km_[g].x, km_[g].y}); \
      pk_fnma(rm[2 * g + 1], ttm_, f2_t{km_[g].z, km_[g].w}); \
      pk_fnma(rh[2 * g], tth_, f2_t{kh_[g].x, kh_[g].y}); \
      pk_fnma(rh[2 * g + 1], tth_, f2_t{kh_[g].z, kh_[g].w}); } \
    dgm = fmaf(-tm_, tkm_, dgm); dgh = fmaf(-th_, tkh_, dgh); \
    x = fmaf(-tm_, bcast(x, p_), x); } while (0)
#define ROUND_BODY2(NG_) do { \
    if (isd) { \
      _Pragma("unroll") for (int d = 0; d < 4 * NG_; d += 4) { \
        *(float4*)(HM + lane * RS + d) = make_float4(rm[d / 2].x, rm[d / 2].y, rm[d / 2 + 1].x, rm[d / 2 + 1].y); \
        *(float4*)(HR + lane * RS + d) = make_float4(rh[d / 2].x, rh[d / 2].y, rh[d / 2 + 1].x, rh[d / 2 + 1].y); } \
      DVM[lane] = __builtin_amdgcn_rcpf(dgm); DVH[lane] = __builtin_amdgcn_rcpf(dgh); \
    } \
    WSYNC(); \
    APPLY_PIVOT2(NG_, p0, a0); \
    if (p1 >= 0) { APPLY_PIVOT2(NG_, p1, a1); APPLY_PIVOT2(NG_, p2, a2); } \
    WSYNC(); } while (0)
  {
    // Rounds come deepest level first, so the group count never grows from one round to the next: one loop per group count,
    // run one after the other (a switch inside one loop makes every join copy the whole register rows).
    int p0 = RND[0].p0, p1 = RND[0].p1, p2 = RND[0].p2, dep = RND[0].depth;
    unsigned long long a0 = RND[0].anc[0], a1 = RND[0].anc[1], a2 = RND[0].anc[2];
    int rd = 0;
#define ROUNDS_AT(NG_, COND_) \
    _Pragma("unroll 1") while (rd < nround && (COND_)) { \
      const int rn = rd + 1 < nround ? rd + 1 : rd; \
      const int np0 = RND[rn].p0, np1 = RND[rn].p1, np2 = RND[rn].p2, ndep = RND[rn].depth; \
      const unsigned long long na0 = RND[rn].anc[0], na1 = RND[rn].anc[1], na2 = RND[rn].anc[2]; \
      ROUND_BODY2(NG_); \
      p0 = np0; p1 = np1; p2 = np2; a0 = na0; a1 = na1; a2 = na2; dep = ndep; rd++; \
    }
    if (MAXD >= 32) ROUNDS_AT((MAXD >= 32 ? 8 : 1), dep > 28)
    if (MAXD >= 28) ROUNDS_AT((MAXD >= 28 ? 7 : 1), dep > 24)
    if (MAXD >= 24) ROUNDS_AT((MAXD >= 24 ? 6 : 1), dep > 20)
    if (MAXD >= 20) ROUNDS_AT((MAXD >= 20 ? 5 : 1), dep > 16)
    if (MAXD >= 16) ROUNDS_AT((MAXD >= 16 ? 4 : 1), dep > 12)
    if (MAXD >= 12) ROUNDS_AT((MAXD >= 12 ? 3 : 1), dep > 8)
    if (MAXD >= 8) ROUNDS_AT((MAXD >= 8 ? 2 : 1), dep > 4)
    ROUNDS_AT(1, true)
#undef ROUNDS_AT
  }
#undef ROUND_BODY2
#undef APPLY_PIVOT2
  dinv_m = isd ? __builtin_amdgcn_rcpf(dgm) : 0.f;
  dinv_h = isd ? __builtin_amdgcn_rcpf(dgh) : 0.f;
  if (isd) {
#pragma unroll
    for (int d = 0; d < MAXD; d += 4) {
      *(float4*)(HM + lane * RS + d) = make_float4(rm[d / 2].x * dinv_m, rm[d / 2].y * dinv_m, rm[d / 2 + 1].x * dinv_m, rm[d / 2 + 1].y * dinv_m);
      *(float4*)(HR + lane * RS + d) = make_float4(rh[d / 2].x * dinv_h, rh[d / 2].y * dinv_h, rh[d / 2 + 1].x * dinv_h, rh[d / 2 + 1].y * dinv_h);
    }
  }
  WSYNC();
}

// Root-first sweep x_d -= L[d][a] x_a over the proper ancestors a of d, a tree level at a time: every lane pulls x of
// its ancestor at depth lvl with one ds_bpermute (lane table from the model: byte = 4 * lane) and applies its own L
// entry for that depth (its row of HR, read once up front).  maxdep dependent steps instead of one per dof.
template <int MAXD>
__device__ __forceinline__ float ldl_pull_sweep(const float* HR, float x, int dli, bool isd, int ddepth, const uint32_t* ancl, int maxdep) {
  constexpr int RS = MAXD;
  maxdep = opaque_s(maxdep);
  float4 row[MAXD / 4];
#pragma unroll
  for (int g = 0; g < MAXD / 4; g++) row[g] = *(const float4*)(HR + dli * RS + 4 * g);
#pragma unroll
  for (int g = 0; g < MAXD / 4; g++) {
    const uint32_t ab = gptr(ancl)[(unsigned)dli * (MAXD / 4) + g];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int lvl = 4 * g + k;
      if (lvl < maxdep) {
        const float xs = __int_as_float(__builtin_amdgcn_ds_bpermute((int)((ab >> (8 * k)) & 0xffu), __float_as_int(x)));
        const float rl = k == 0 ? row[g].x : (k == 1 ? row[g].y : (k == 2 ? row[g].z : row[g].w));
        x = fmaf((isd && lvl < ddepth) ? -rl : 0.f, xs, x);
      }
    }
  }
  return x;
}

// x = (L' D L)^-1 rhs; HR rows hold L (see ldl_factor), dinv = 1/D_lane.
template <int MAXD>
__device__ __forceinline__ float ldl_solve(const float* HR, float rhs, int lane, bool isd, int ddepth, int dsub, int nv, float dinv_mine,
                                           const uint32_t* ancl, int maxdep) {
  constexpr int RS = MAXD;
  float x = rhs;
  const int dli = isd ? lane : 0;                 // in-bounds row for idle lanes; their result is discarded
  {
    int i = nv - 1;
    for (; i >= 8; i -= 8) {                       // batches of 8 pivots: the 8 LDS reads do not depend on x
      float l[8];
#pragma unroll
      for (int u = 0; u < 8; u++) l[u] = HR[(i - u) * RS + ddepth];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const bool anc = lane < i - u && i - u < lane + dsub;
        x = fmaf(anc ? -l[u] : 0.f, bcast(x, i - u), x);
      }
    }
#pragma unroll 1
    for (; i >= 1; i--) {
      const float l = (lane < i && i < lane + dsub) ? HR[i * RS + ddepth] : 0.f;
      x = fmaf(-l, bcast(x, i), x);
    }
  }
  x *= dinv_mine;
  x = ldl_pull_sweep<MAXD>(HR, x, dli, isd, ddepth, ancl, maxdep);
  return x;
}

// ---- constraint path: joint limits + plane contacts, pyramidal cone, PGS ----------
// A row of J touches only the dofs on the chain from its body to the root: YJ[e][RS] holds it indexed by dof depth
// and is transformed (lane = row, in registers) to Z_e = J_e L^-1 D^-1/2 (M = L'DL), so that A = J M^-1 J' = Z Z'.
// A + diag(R) is formed explicitly (one row per lane in registers up to LL.na rows, full rows in an HBM scratch beyond) and
// PGS runs with one row per lane.  EP[e][8] = {-, aref, R, b, force, R0, type|id bits, mu}; CH[e] = chain's last dof + 1.

__device__ __forceinline__ float wave_sum_fast(float v) {   // DPP reduction, total broadcast from lane 63
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));  x = __float_as_int(v);   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));  x = __float_as_int(v);   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true)); x = __float_as_int(v);   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true)); x = __float_as_int(v);   // row_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true)); x = __float_as_int(v);   // row_bcast:15
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true));                          // row_bcast:31
  return bcast(v, 63);
}

__device__ __forceinline__ float impedance(float d0, float d1, float width, float mid, float power, float pos, float margin) {
  d0 = fminf(fmaxf(d0, 1e-4f), 0.9999f); d1 = fminf(fmaxf(d1, 1e-4f), 0.9999f);
  width = fmaxf(0.f, width); mid = fminf(fmaxf(mid, 1e-4f), 0.9999f); power = fmaxf(1.f, power);
  if (d0 == d1 || width <= 1e-15f) return 0.5f * (d0 + d1);
  const float x = fabsf((pos - margin) / width);
  float y;
  if (x >= 1.f) y = 1.f;
  else if (x <= 0.f) y = 0.f;
  else if (power == 1.f) y = x;
  else if (power == 2.f) y = x <= mid ? x * x / mid : 1.f - (1.f - x) * (1.f - x) / (1.f - mid);   // MuJoCo's default solimp: no powf (~300 VALU a pair)
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1.f);
  else y = 1.f - powf(1.f - x, power) / powf(1.f - mid, power - 1.f);
  return d0 + y * (d1 - d0);
}

// R and the (K*imp, B) pair of one row from solref / solimp (mj_makeImpedance)
__device__ __forceinline__ void row_params(float sr0, float sr1, float si0, float si1, float si2, float si3, float si4,
                                           float pos, float margin, float diag_approx, float h, float* R, float* kimp, float* bb) {
  const float imp = impedance(si0, si1, si2, si3, si4, pos, margin);
  const float dmax = fminf(fmaxf(si1, 1e-4f), 0.9999f);
  float K, B;
  if (sr0 > 0.f) {
    const float tc = fmaxf(sr0, 2.f * h);
    K = 1.f / fmaxf(1e-15f, dmax * dmax * tc * tc * sr1 * sr1);
    B = 2.f / fmaxf(1e-15f, dmax * tc);
  } else { K = -sr0 / fmaxf(1e-15f, dmax * dmax); B = -sr1 / fmaxf(1e-15f, dmax); }
  *R = fmaxf(1e-15f, (1.f - imp) * diag_approx / imp);
  *kimp = K * imp; *bb = B;
}

// ---- polytopes (box, cylinder, convex mesh) in explicit pairs: include/fmj.h (ABI 6), oracle poly_vertex / poly_signed / collide_pair
__device__ __forceinline__ bool geom_is_round(int t) { return t == FMJ_GEOM_SPHERE || t == FMJ_GEOM_CAPSULE; }
// vertex k of polytope (type, size row gs) in the geom frame
template <class MT>
__device__ __forceinline__ v3 poly_vertex(MT& M, int type, float4 gs, int k) {
  if (type == FMJ_GEOM_BOX) return mk3((k & 1) ? gs.x : -gs.x, (k & 2) ? gs.y : -gs.y, (k & 4) ? gs.z : -gs.z);
  if (type == FMJ_GEOM_CYLINDER) {          // 12 points on each rim in steps of 150 degrees, the first on +x; k < 12: the +z rim
    float sn, cs;
    sincospif((float)(((k % 12) * 5) % 12) * (1.0f / 6.0f), &sn, &cs);
    return mk3(gs.x * cs, gs.x * sn, k < 12 ? gs.y : -gs.y);
  }
  const float4 v = ldg4(M.mesh_vert, (unsigned)(__float_as_int(gs.x) + k));
  return mk3(v.x, v.y, v.z);
}
// signed distance of x (geom frame) to the polytope = max over its faces of (n . x - d), and that face's outward normal
template <class MT>
__device__ __forceinline__ float poly_signed(MT& M, int type, float4 gs, int face0, int nface, v3 x, v3* n) {
  if (type == FMJ_GEOM_BOX) {
    const float sx = fabsf(x.x) - gs.x, sy = fabsf(x.y) - gs.y, sz = fabsf(x.z) - gs.z;
    float s = sx; *n = mk3(x.x < 0.f ? -1.f : 1.f, 0.f, 0.f);
    if (sy > s) { s = sy; *n = mk3(0.f, x.y < 0.f ? -1.f : 1.f, 0.f); }
    if (sz > s) { s = sz; *n = mk3(0.f, 0.f, x.z < 0.f ? -1.f : 1.f); }
    return s;
  }
  if (type == FMJ_GEOM_CYLINDER) {
    const float r = sqrtf(x.x * x.x + x.y * x.y);
    float s = fabsf(x.z) - gs.y; *n = mk3(0.f, 0.f, x.z < 0.f ? -1.f : 1.f);
    if (r - gs.x > s) { s = r - gs.x; *n = r > 1e-15f ? mk3(x.x / r, x.y / r, 0.f) : mk3(1.f, 0.f, 0.f); }
    return s;
  }
  float s = -1e30f; *n = mk3(0.f, 0.f, 1.f);
  for (int f = 0; f < nface; f++) {
    const float4 pl = ldg4(M.mesh_face, (unsigned)(face0 + f));
    const float sf = pl.x * x.x + pl.y * x.y + pl.z * x.z - pl.w;
    if (sf > s) { s = sf; *n = mk3(pl.x, pl.y, pl.z); }
  }
  return s;
}
__device__ __forceinline__ float geom_rbound(int type, float4 gs) {
  return type == FMJ_GEOM_SPHERE ? gs.x : type == FMJ_GEOM_CAPSULE ? gs.x + gs.y : type == FMJ_GEOM_CYLINDER ? sqrtf(gs.x * gs.x + gs.y * gs.y)
       : type == FMJ_GEOM_BOX ? sqrtf(gs.x * gs.x + gs.y * gs.y + gs.z * gs.z) : gs.z;
}
// explicit pair (g1, g2) with at least one polytope: up to 4 contacts (position, normal from geom1 to geom2, distance), PO = body poses in LDS
template <class MT>
__device__ __forceinline__ int pair_polytope(MT& M, const float* PO, int g1, int g2, v3* cq, v3* nq, float* dq) {
  int type[2], face0[2], nface[2], nvert[2]; float4 gs[2]; v3 pos[2]; q4 wq[2], wqc[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int g = k ? g2 : g1;
    const int4 gi = GTABI(g, 0);
    const float4 gp = GTAB(g, 2), gq = GTAB(g, 3);
    gs[k] = GTAB(g, 1);
    type[k] = gi.x; face0[k] = gi.w; nface[k] = __float_as_int(GTAB(g, 5).w);
    nvert[k] = gi.x == FMJ_GEOM_BOX ? 8 : gi.x == FMJ_GEOM_CYLINDER ? 24 : gi.x == FMJ_GEOM_MESH ? __float_as_int(gs[k].y) : 0;
    const float4 bp = *(const float4*)(PO + gi.y * 8), bq = *(const float4*)(PO + gi.y * 8 + 4);
    const q4 bqq = {bq.x, bq.y, bq.z, bq.w}, gqq = {gq.x, gq.y, gq.z, gq.w};
    wq[k] = qmul(bqq, gqq); wqc[k] = q4{wq[k].w, -wq[k].x, -wq[k].y, -wq[k].z};
    pos[k] = add3(mk3(bp.x, bp.y, bp.z), qrot(bqq, mk3(gp.x, gp.y, gp.z)));
  }
  const v3 dc = sub3(pos[1], pos[0]);
  if (sqrtf(dot3(dc, dc)) > geom_rbound(type[0], gs[0]) + geom_rbound(type[1], gs[1])) return 0;
  const bool r0 = geom_is_round(type[0]), r1 = geom_is_round(type[1]);
  int cnt = 0;
  if (r0 != r1) {                              // polytope against sphere / capsule: the round geom's centres against the faces
    const int rk = r0 ? 0 : 1, pk = 1 - rk;
    const float rad = gs[rk].x, half = gs[rk].y;
    const int ncen = type[rk] == FMJ_GEOM_CAPSULE ? 2 : 1;
    const v3 ax = qrot(wq[rk], mk3(0.f, 0.f, 1.f));
    for (int c = 0; c < ncen; c++) {
      const float sgn = ncen == 2 ? (c == 0 ? 1.f : -1.f) : 0.f;
      const v3 cw = add3(pos[rk], scl3(ax, sgn * half));
      v3 nl;
      const float dist = poly_signed(M, type[pk], gs[pk], face0[pk], nface[pk], qrot(wqc[pk], sub3(cw, pos[pk])), &nl) - rad;
      if (dist < 0.f) {
        const v3 nw = qrot(wq[pk], nl);
        const v3 cp = sub3(cw, scl3(nw, rad + 0.5f * dist)), n12 = pk == 0 ? nw : scl3(nw, -1.f);
#pragma unroll
        for (int q = 0; q < 2; q++) if (cnt == q) { cq[q] = cp; nq[q] = n12; dq[q] = dist; }
        cnt++;
      }
    }
    return cnt;
  }
  for (int side = 0; side < 2; side++) {       // side 0: geom1's vertices in geom2; side 1: geom2's vertices in geom1
    const int va = side, fb = 1 - side;
    for (int k = 0; k < nvert[va]; k++) {
      const v3 vw = add3(pos[va], qrot(wq[va], poly_vertex(M, type[va], gs[va], k)));
      v3 nl;
      float td = poly_signed(M, type[fb], gs[fb], face0[fb], nface[fb], qrot(wqc[fb], sub3(vw, pos[fb])), &nl);
      if (!(td < 0.f)) continue;
      const v3 nw = qrot(wq[fb], nl);
      v3 tc = sub3(vw, scl3(nw, 0.5f * td)), tn = fb == 0 ? nw : scl3(nw, -1.f);
      bool have = true;                          // insertion by depth, ties keep the earlier candidate (the rule of mesh against ground)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool empty = q >= cnt;
        if (have && (empty || td < dq[q])) {
          const float sd = dq[q]; const v3 sc = cq[q], sn = nq[q];
          dq[q] = td; cq[q] = tc; nq[q] = tn;
          td = sd; tc = sc; tn = sn;
          have = !empty;
        }
      }
      if (cnt < 4) cnt++;
    }
  }
  return cnt;
}

// ---- elliptic cone on the dual side (mj_solPGS's block update, the noslip pass): oracle cone_zone / qcqp2 / pgs_elliptic_block in fp32
// forces of one elliptic contact at the residuals jar = (normal, tangent 1, tangent 2): mj_constraintUpdate's three zones (the warm start)
__device__ __forceinline__ void ell_zone_forces(float j0, float j1, float j2, float D0, float fr, float isq, float* f0, float* f1, float* f2) {
  const float mus = fr * isq;
  const float U1 = j1 * fr, U2 = j2 * fr, N = j0 * mus, T = sqrtf(U1 * U1 + U2 * U2);
  if (N >= mus * T || (T <= 0.f && N >= 0.f)) { *f0 = 0.f; *f1 = 0.f; *f2 = 0.f; }
  else if (mus * N + T <= 0.f || (T <= 0.f && N < 0.f)) { const float Dt = D0 * fr * fr / (mus * mus); *f0 = -D0 * j0; *f1 = -Dt * j1; *f2 = -Dt * j2; }
  else { const float Dm = D0 / (mus * mus * (1.f + mus * mus)); const float NmT = N - mus * T; *f0 = -Dm * NmT * mus; const float sc = -(*f0) / T * fr; *f1 = sc * U1; *f2 = sc * U2; }
}
// mju_QCQP2: min 0.5 x'A x + x'b  s.t.  sum (x_i / d)^2 <= r^2 (both friction coefficients equal: condim 3), Newton on the multiplier
__device__ __forceinline__ bool qcqp2_dev(float a11, float a12, float a22, float b1, float b2, float d, float r, float* x1, float* x2) {
  b1 *= d; b2 *= d; a11 *= d * d; a22 *= d * d; a12 *= d * d;
  float la = 0.f, v1 = 0.f, v2 = 0.f;
  const float r2 = r * r;
  for (int it = 0; it < 20; it++) {
    const float det = (a11 + la) * (a22 + la) - a12 * a12;
    if (det < 1e-10f) { v1 = v2 = 0.f; break; }
    const float di = 1.f / det, P11 = (a22 + la) * di, P22 = (a11 + la) * di, P12 = -a12 * di;
    v1 = -P11 * b1 - P12 * b2; v2 = -P12 * b1 - P22 * b2;
    const float val = v1 * v1 + v2 * v2 - r2;
    if (val < 1e-10f) break;
    const float deriv = -2.f * (P11 * v1 * v1 + 2.f * P12 * v1 * v2 + P22 * v2 * v2);
    const float delta = -val / deriv;
    if (delta < 1e-10f) break;
    la += delta;
  }
  *x1 = v1 * d; *x2 = v2 * d;
  return la != 0.f;
}

// Height of world point p above ground entry pl along the local surface normal, and that normal.  Plane: n . p - offset.
// Heightfield (MuJoCo hfield semantics: nrow x ncol samples over [-rx, rx] x [-ry, ry] of the geom frame, elevation = data *
// size z): the plane of the grid triangle under p (cells split along the diagonal (c, r) - (c + 1, r + 1)); nothing outside
// the grid.  PTAB(pl, 1).y != 0 marks a heightfield, PTAB(pl, 0) then holds its position, (pl, 2) its quaternion, (pl, 3)
// rx, ry, size z.
template <class MT>
__device__ __forceinline__ float ground_dist(MT& M, int pl, float4 pn, float4 pp, v3 p, v3* n) {
  if (pp.y == 0.f) { *n = mk3(pn.x, pn.y, pn.z); return dot3(p, *n) - pn.w; }
  const float4 hq = PTAB(pl, 2), hs = PTAB(pl, 3);
  const q4 q = {hq.x, hq.y, hq.z, hq.w}, qc = {hq.x, -hq.y, -hq.z, -hq.w};
  const v3 pl_ = qrot(qc, sub3(p, mk3(pn.x, pn.y, pn.z)));
  const int nc = M.hf_ncol, nr = M.hf_nrow;
  const float sx = (float)(nc - 1) / (2.f * hs.x), sy = (float)(nr - 1) / (2.f * hs.y);
  const float gx = (pl_.x + hs.x) * sx, gy = (pl_.y + hs.y) * sy;
  *n = qrot(q, mk3(0.f, 0.f, 1.f));
  if (!(gx >= 0.f && gx <= (float)(nc - 1) && gy >= 0.f && gy <= (float)(nr - 1))) return 1e30f;
  const int c = min((int)gx, nc - 2), r = min((int)gy, nr - 2);
  const float fx = gx - (float)c, fy = gy - (float)r;
  const float AS1* D = gptr(M.hf_data) + (size_t)r * nc + c;
  const float z00 = D[0] * hs.z, z10 = D[1] * hs.z, z01 = D[nc] * hs.z, z11 = D[nc + 1] * hs.z;
  float zs, gxs, gys;                                          // surface height under p and its slopes per cell
  if (fx >= fy) { gxs = z10 - z00; gys = z11 - z10; } else { gxs = z11 - z01; gys = z01 - z00; }
  zs = z00 + gxs * fx + gys * fy;
  const v3 nl = mk3(-gxs * sx, -gys * sy, 1.f);
  const float inv = 1.0f / sqrtf(dot3(nl, nl));
  *n = qrot(q, scl3(nl, inv));
  return (pl_.z - zs) * inv;                                   // n_z (p_z - z_surface)
}

// One contact-sensor row (reference sensors.pyx:20-137,158-182): accumulate every contact whose keys hit `row`.
// Contact records: pos(3) frame(9) force(3: normal,t1,t2) int32 geom1 << 16 | geom2.  The four keys of
// sensors.pyx:163-169 in the reference's order: (g1,g2) -1, (g2,g1) +1, (g1,-1) -1, (g2,-1) +1; a key that maps to
// `row` adds the contact once (a contact can hit one row through several keys, as in the reference).
__device__ __forceinline__ void contact_row(const float* C, int nc, int row, const int* geom_sensor_, int n_pairs, const int* pairs_,
                                            float inv_newtons, float inv_meters, float* out_) {
  float acc[12]; float norm_sum = 0.f;
  const int AS1* const geom_sensor = gptr(geom_sensor_);
  const int AS1* const pairs = gptr(pairs_);
  float AS1* const out = gptr(out_);
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.f;
  for (int c = 0; c < nc; c++) {
    const float* ct = C + c * 16;
    const int gg = __float_as_int(ct[15]);
    const int g1 = gg >> 16, g2 = gg & 0xffff;
    int nneg = 0, npos = 0;                       // matching keys of sign -1 / +1
    for (int p = 0; p < n_pairs; p++) {
      if (pairs[3 * p + 2] != row) continue;
      if (pairs[3 * p] == g1 && pairs[3 * p + 1] == g2) nneg++;
      if (pairs[3 * p] == g2 && pairs[3 * p + 1] == g1) npos++;
    }
    if (geom_sensor[g1] == row) nneg++;
    if (geom_sensor[g2] == row) npos++;
    if (nneg + npos == 0) continue;
    float tot[3];
    const float sgn = (float)(npos - nneg), cnt = (float)(npos + nneg);
#pragma unroll
    for (int i = 0; i < 3; i++) {                                            // store_forces, sensors.pyx:33-52
      const float reaction = ct[12] * ct[3 + i];
      const float friction = ct[13] * ct[6 + i] + ct[14] * ct[9 + i];
      tot[i] = reaction + friction;
      acc[FMJ_CONTACT_REACTION + i] += sgn * reaction; acc[FMJ_CONTACT_FRICTION + i] += sgn * friction; acc[FMJ_CONTACT_TOTAL + i] += sgn * tot[i];
    }
    const float nrm = cnt * sqrtf(tot[0] * tot[0] + tot[1] * tot[1] + tot[2] * tot[2]);   // |+-total| once per matching key
#pragma unroll
    for (int i = 0; i < 3; i++) acc[FMJ_CONTACT_POSITION + i] += nrm * ct[i];
    norm_sum += nrm;
  }
  if (norm_sum > 0.f) { for (int i = 0; i < 3; i++) acc[FMJ_CONTACT_POSITION + i] /= norm_sum; }     // sensors.pyx:85-88
#pragma unroll
  for (int k = 0; k < 9; k++) out[k] = acc[k] * inv_newtons;                                          // :99-107
#pragma unroll
  for (int k = 9; k < 12; k++) out[k] = acc[k] * inv_meters;                                          // :108-110
}

// Emits, for iteration `it`, what ExperimentTask.before_step does with the link data of the last
// forward pass (reference task.py:168-186): the links row (physics.py:449-466,435-446), the drag of
// every swimming link (drag.pyx:389-411 -> xfrc row) and the world-frame xfrc_applied of this body.
template <class MT, class AT>
__device__ __forceinline__ void emit_links_and_drag(MT& M, AT& A, int env, int it, bool isb, bool frozen,
                                                    int link_row, int swim_slot, v3 xpos, q4 xquat, v3 xipos, v3 linvel,
                                                    v3 angvel, float* xf) {
  const int index = it % A.buffer_size;
  const v3 r_com = scl3(xipos, A.inv_meters), r_urdf = scl3(xpos, A.inv_meters);
  const v3 r_lin = scl3(linvel, A.inv_velocity), r_ang = scl3(angvel, A.inv_angvel);
  const fq r_q = {xquat.x, xquat.y, xquat.z, xquat.w};   // wxyz -> xyzw (physics.py:458)
  if (A.do_readout && isb && !frozen && link_row >= 0) {
    float AS1* row = gptr(A.links) + ((size_t)index * A.row_stride_links + (size_t)env * M.n_links * FMJ_LINK_SIZE) + link_row * FMJ_LINK_SIZE;
    stg4(row + 0, r_com.x, r_com.y, r_com.z, r_q.x);
    stg4(row + 4, r_q.y, r_q.z, r_q.w, r_urdf.x);
    stg4(row + 8, r_urdf.y, r_urdf.z, r_q.x, r_q.y);
    stg4(row + 12, r_q.z, r_q.w, r_lin.x, r_lin.y);
    stg4(row + 16, r_lin.z, r_ang.x, r_ang.y, r_ang.z);
  }
  if (A.do_drag) {
#pragma unroll
    for (int k = 0; k < 6; k++) xf[k] = 0.f;
    if (isb && swim_slot >= 0) {
      const float4 s0 = STAB(swim_slot, 0), s1 = STAB(swim_slot, 1), s2 = STAB(swim_slot, 2);
      v3 fo, to, fw, tw;
      if (drag_link_same_frames(r_com, xquat, r_lin, r_ang, s0, s1, s2.x, A, &fo, &fw, &to, &tw)) {
        if (!frozen) {
          float AS1* xr = gptr(A.xfrc) + ((size_t)index * A.row_stride_xfrc + (size_t)env * M.n_xfrc * FMJ_XFRC_SIZE) + __float_as_int(s2.z) * FMJ_XFRC_SIZE;
          stg2(xr + 0, fo.x, fo.y);
          stg2(xr + 2, fo.z, to.x);
          stg2(xr + 4, to.y, to.z);
        }
        xf[0] = fw.x * A.newtons; xf[1] = fw.y * A.newtons; xf[2] = fw.z * A.newtons;
        xf[3] = tw.x * A.torques; xf[4] = tw.y * A.torques; xf[5] = tw.z * A.torques;
      }
    }
  }
}

// Diagnostic build only (-DFMJ_STAMPS): per-phase s_memtime deltas of env 0, summed over the steps of a launch,
// written over qacc[0, :]. Never enabled in the shipped library; its numbers are shares, not run times.
#ifdef FMJ_STAMPS
#define NSTAMP 24
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
    stamp_acc[i] += (float)(t_ - stamp_prev); stamp_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i)
#endif

// PAIRS: the model has explicit geom pairs (rows over two branches of the tree) or mesh geoms; a separate instantiation
// because the fork handling costs ~150 VGPRs that every constraint model would otherwise pay for in spills (and the mesh
// vertex loop another 20).
// NEWTON: the constraint forces come from MuJoCo's Newton solver on the primal problem instead of PGS on the dual one (see the
// Newton block of fmj_cons_rows.inc); models without explicit pairs.
// MESH: the narrow phase has the convex-mesh vertex loop (~20 VGPRs); on by itself for models with mesh geoms but no explicit pairs,
// which then do not pay for the fork code of PAIRS (its spills cost the mesh-foot walker 52 KB of scratch traffic per env-step).
template <bool FUSED, int MAXD, bool CONS, bool PAIRS = false, bool NEWTON = false, bool MESH = PAIRS, bool ELL = false>
__global__ void __launch_bounds__(64, (CONS || MAXD > 32) ? 2 : 4) fmj_step_kernel(const DevModel M_by_value, const StepArgs A_by_value) {
  extern __shared__ __align__(16) float lds[];
  // the two arguments are read where they are used, through the kernarg segment (scalar loads), instead of being held
  // in SGPRs - and spilled - for the whole launch (see fmj_dual2.inc); the pointers are laundered once per step
  const char AS4* const karg = (const char AS4*)__builtin_amdgcn_kernarg_segment_ptr();
  const DevModel AS4* Mp = (const DevModel AS4*)karg;
  const StepArgs AS4* Ap = (const StepArgs AS4*)(karg + FMJ_KARG_A_OFF);
#define M (*Mp)
#define A (*Ap)
  const int env = A.env_order ? gptr(A.env_order)[blockIdx.x] : blockIdx.x;
  // Hand-over from the two-env constraint kernel (fmj_cons2.inc): resume[env] steps of this launch are done; this kernel runs the
  // rest from the launch-boundary state that kernel stored (the row of the first remaining iteration is already written).
  const int step0 = (CONS && A.resume) ? gptr(A.resume)[env] : 0;
  if (step0 >= A.n_steps && CONS && A.resume) return;
  const int lane = threadIdx.x;
  const int nb = M.nbody, nv = M.nv, nq = M.nq, nu = M.nu;
  constexpr int RS = MAXD;                     // row stride of H == register row length (dispatch guarantees M.rs == MAXD)
  const LdsLayout LL = lds_layout(nb, nv, nq, RS, M.anc_stride, CONS ? 1 : 0, M.maxefc, M.max_contacts, M.nvs, M.npair);
  float* F = lds + LL.P1;                      // body force -> subtree force
  float* CI = lds + LL.CI;
  float* CD = lds + LL.CD;
  float* HR = lds + LL.HR;
  float* QP = lds + LL.QP;
  float* QV = lds + LL.QV;
  float* XV = lds + LL.XV;
  float* VT = lds + LL.VT;
  const uint8_t* JMP = (const uint8_t*)(lds + LL.ANC);   // [nb][anc_stride]: ancestor at distance 2^r
  float* HM = lds + LL.HM;  float* YJ = lds + LL.YJ;  float* EPL = lds + LL.EP;  float* CT = lds + LL.CT;
  float* XS = lds + LL.XS;  float* QW = lds + LL.QW;  float* DI = lds + LL.DI;
  float* PO = lds + LL.PO;  float* LC = lds + LL.LC;  float* SD = lds + LL.SD;  float* CH = lds + LL.CH;
  float* YFL = lds + LL.YF; float* CF = lds + LL.CF;

  const bool isb = lane > 0 && lane < nb;
  const int bl = isb ? lane : 0;
  const bool isd = lane < nv;
  const int dl = isd ? lane : 0;
  // dof-role constants that index loops stay resident (2 VGPRs); everything else is re-read per step
  const int4 d_info0 = DTABI(dl, 0);
  const int ddepth_o = isd ? d_info0.y : 0;
  const int dsub_o = isd ? d_info0.z : 0;

  // ---- load tables + state -------------------------------------------------------------------------
  int warn = 0;
  // The step reads LDS words it never wrote (record pads; row slots of the factorisation no round has published yet, under
  // masks that select them away but still multiply them by zero): they must hold finite values, not what the previous
  // workgroup left there.
  for (int i = lane * 4; i < LL.total; i += 256) *(float4*)(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  WSYNC();
  {
    uint32_t* jw = (uint32_t*)(lds + LL.ANC);
    const int nw = r4(nb * M.anc_stride) / 4;
    for (int i = lane; i < nw; i += 64) jw[i] = ((const uint32_t*)M.b_anc)[i];
  }
  {
    const float* gq = glob(A.qpos) + (size_t)env * nq;
    const float* gv = glob(A.qvel) + (size_t)env * nv;
    for (int i = lane; i < nq; i += 64) { const float v = gq[i]; QP[i] = v; if (!(fabsf(v) <= 1e10f)) warn |= FMJ_WARN_BADQPOS; }   // mj_checkPos
    for (int i = lane; i < nv; i += 64) { const float v = gv[i]; QV[i] = v; if (!(fabsf(v) <= 1e10f)) warn |= FMJ_WARN_BADQVEL; }   // mj_checkVel
    if (lane < 8) VT[lane] = 0.f;
    if (CONS) for (int i = lane; i < nv; i += 64) QW[i] = gptr(A.qacc_warmstart)[(size_t)env * nv + i];
    if (CONS) for (int i = lane; i < (nv * nv + 3) / 4; i += 64) ((uint32_t*)LC)[i] = ((const uint32_t*)M.lcad)[i];
  }
  int cy_ncon = 0;                                  // contacts of the last forward pass (records incl. forces stay in CT)
  if (CONS && FUSED && A.contacts_rows) {
    cy_ncon = gptr(A.ncon)[env];
    for (int i = lane; i < cy_ncon * 16; i += 64) CT[i] = gptr(A.contact)[(size_t)env * M.max_contacts * 16 + i];
  }
  float cy_limfrc = 0.f;                            // carried joint-limit force of this dof's joint (physics.py:484-487)
  if (CONS && FUSED && isd) { const int4 da0 = DTABI(dl, 2); if (DTAB(dl, 1).w != 0.f) cy_limfrc = gptr(A.sensordata)[(size_t)env * M.nsensordata + 6 * (nb - 1) + 3 * da0.z + 2] * A.inv_torques; }
  float xf[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // world-frame external force / torque on this body
  float cy_actsum = 0.f;                            // carried motor torque (physics.py:510-524)
  if (FUSED) {
    const float4 dp = DTAB(dl, 1);
    if (isd && dp.w != 0.f) {
      const int4 da = DTABI(dl, 2);
      const float* sa = glob(A.sensordata) + (size_t)env * M.nsensordata + 6 * (nb - 1) + 3 * M.njs;
#pragma unroll
      for (int a = 0; a < 4; a++) if (a < da.y) cy_actsum += sa[__float_as_int(ATAB(da.x + a, 2).x)] * A.inv_torques;
    }
  }
  if (!(FUSED && A.do_drag) && A.xfrc_applied && isb) {
    const float* x = glob(A.xfrc_applied) + (size_t)env * nb * 6 + bl * 6;
#pragma unroll
    for (int k = 0; k < 6; k++) xf[k] = x[k];
  }
  // An env with a bad-state bit is frozen (include/fmj.h): it is not integrated and writes no rows, from the step that
  // finds the bad value on and in later launches until the caller clears its status word.
  bool frozen = (gptr(A.status)[env] & FMJ_WARN_FREEZE) != 0 || __any((warn & FMJ_WARN_FREEZE) != 0);
  int steps_done = 0;
  // The fields of the last forward pass are not carried through LDS: the links row of iteration it + 1 and its drag
  // (ExperimentTask.before_step, reference task.py:168-186) are emitted by the step that computes them (step it); the row of
  // the launch's first iteration comes from the fields the caller hands over.
  if (FUSED && !frozen && A.n_steps > 0 && !(CONS && A.resume)) {
    const int cl = lane < nb ? lane : 0;
    const float* p = glob(A.xpos) + (size_t)env * nb * 3 + cl * 3;
    const float4 q = *(const float4*)(glob(A.xquat) + (size_t)env * nb * 4 + cl * 4);
    const float* ip = glob(A.xipos) + (size_t)env * nb * 3 + cl * 3;
    const float* sd = glob(A.sensordata) + (size_t)env * M.nsensordata + 6 * (isb ? lane - 1 : 0);
    const int4 ci2 = BTABI(bl, 8);
    const q4 cq = {q.x, q.y, q.z, q.w};
    emit_links_and_drag(M, A, env, A.iteration0, isb, false, ci2.z, ci2.w, mk3(p[0], p[1], p[2]), cq, mk3(ip[0], ip[1], ip[2]),
                        mk3(sd[0], sd[1], sd[2]), mk3(sd[3], sd[4], sd[5]), xf);
  }
  WSYNC();
#ifdef FMJ_STAMPS
  float stamp_acc[NSTAMP];
#pragma unroll
  for (int i = 0; i < NSTAMP; i++) stamp_acc[i] = 0.f;
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif

  const int lane_outer = lane;
  // sub-steps (include/fmj.h, reference task.py:168-186,348-369): `sub` = index of this physics step inside its iteration
  // (0 = the full step), `itm` = iterations completed in this launch
  int sub = 0, itm = 0;
  if (step0 > 0) { const int S0 = A.substeps > 1 ? A.substeps : 1; itm = step0 / S0; sub = step0 - itm * S0; }
#pragma unroll 1
  for (int step = step0; step < A.n_steps; step++) {
    if (frozen) break;
    asm volatile("" : "+s"(Mp), "+s"(Ap));       // arguments are re-read from the kernarg segment in every step
    // per-lane LDS/global addresses are recomputed every step instead of being hoisted and spilled
    const int lane = opaque(lane_outer);
    const int it = A.iteration0 + itm;
    const bool last = step == A.n_steps - 1;
    const int S_sub = A.substeps;
    const bool full = sub == 0;
    const int nsub = sub + 1 >= S_sub ? 0 : sub + 1;           // the next physics step: its sub index, whether it is a full step,
    const bool nfull = nsub == 0;                               // and task.iteration as its before_step will see it: the reference
    const int nit = (nfull ? it + 1 : it) + ((nsub >= 1 && nsub >= S_sub - 1) ? 1 : 0);   // advances it after sub-step S - 2 (task.py:352-355)
    const int blo = opaque(bl), dlo = opaque(dl);
    // the lane's depth / subtree size are laundered too: every per-lane predicate made from them (lvl < ddepth for 20
    // levels, the ancestor tests of the sweeps) is loop-invariant and was hoisted into SGPR pairs - ~100 of them, all spilled
    const int ddepth = opaque(ddepth_o), dsub = opaque(dsub_o);
    // ============ before_step (reference task.py:168-186) ============
    STAMP(0);   // emit links + drag
    if (CONS && FUSED && A.do_readout && A.contacts_rows && full) {     // cycontacts2data from the carried contact list
      const int index = it % A.buffer_size;
      for (int row = lane; row < A.n_contact_rows; row += 64)
        contact_row(CT, cy_ncon, row, glob(A.geom_sensor), A.n_pairs, glob(A.pairs), A.inv_newtons, A.inv_meters,
                    glob(A.contacts_rows) + ((size_t)index * A.row_stride_contacts + ((size_t)env * A.n_contact_rows + row) * FMJ_CONTACT_SIZE));
      WSYNC();
    }
    // joint part (physics.py:500-524): needs the CURRENT qpos/qvel
    if (FUSED && A.do_readout && full) {
      const int4 di = DTABI(dlo, 0);
      const float4 dp = DTAB(dlo, 1);
      if (isd && dp.w != 0.f && di.w >= 0) {
        const int index = it % A.buffer_size;
        float AS1* row = gptr(A.joints) + ((size_t)index * A.row_stride_joints + (size_t)env * M.n_joints * FMJ_JOINT_SIZE) + di.w * FMJ_JOINT_SIZE;
        stg4(row + 0, QP[__float_as_int(dp.z)], QV[lane] * A.inv_angvel, 0.f, 0.f);      // the whole row, as in fmj_dual2.inc
        stg4(row + 4, 0.f, 0.f, 0.f, 0.f);
        stg4(row + 8, cy_actsum, cy_limfrc, 0.f, 0.f);
      }
    }

    STAMP(1);   // joints row
    // ============ mj_step ============
    const int4 c_info = BTABI(blo, 7);        // parent, jtype, qadr, dadr
    const int jtype = isb ? c_info.y : -1;
    const int qadr = c_info.z, dadr = c_info.w;
    const float4 c_axis_q0 = BTAB(blo, 5);
    const float4 c_jpos_k = BTAB(blo, 6);
    // ---- K: local transforms, composed along the chains by pointer jumping (log2(depth) rounds)
    const bool any_jpos = M.any_jpos != 0, any_bquat = M.any_bquat != 0, any_iquat = M.any_iquat != 0;   // model-wide (uniform) shortcuts
    v3 xp; q4 xq;
    {
      const float4 c_pos_mass = BTAB(blo, 0);
      const float4 c_quat = BTAB(blo, 1);
      xp = mk3(c_pos_mass.x, c_pos_mass.y, c_pos_mass.z);
      xq.w = c_quat.x; xq.x = c_quat.y; xq.y = c_quat.z; xq.z = c_quat.w;
      if (jtype == FMJ_JNT_FREE) {
        xp = mk3(QP[qadr], QP[qadr + 1], QP[qadr + 2]);
        q4 rq = {QP[qadr + 3], QP[qadr + 4], QP[qadr + 5], QP[qadr + 6]};
        xq = qnormalize(rq);
      } else if (jtype == FMJ_JNT_HINGE) {
        const float q = QP[qadr] - c_axis_q0.w;
        const v3 ax = mk3(c_axis_q0.x, c_axis_q0.y, c_axis_q0.z);
        const q4 ql = axisangle_mid(ax, q);       // half-angle sin / cos by polynomial for |q| <= pi
        if (any_jpos) {                           // anchor off the body origin: the body turns about the anchor
          const v3 jp = mk3(c_jpos_k.x, c_jpos_k.y, c_jpos_k.z);
          xp = add3(xp, qrot(xq, sub3(jp, qrot(ql, jp))));
        }
        xq = any_bquat ? qmul(xq, ql) : ql;       // body frames aligned with their parents': the local rotation is the joint's
      } else if (jtype == FMJ_JNT_SLIDE) {
        const float q = QP[qadr] - c_axis_q0.w;
        xp = add3(xp, qrot(xq, scl3(mk3(c_axis_q0.x, c_axis_q0.y, c_axis_q0.z), q)));
      }
      if (!isb) { xp = mk3(0.f, 0.f, 0.f); xq.w = 1.f; xq.x = xq.y = xq.z = 0.f; }
      const uint32_t jm = lane < nb ? *(const uint32_t*)(JMP + lane * M.anc_stride) : 0u;   // rounds 0..3
      // the partner's pose comes out of its registers with ds_bpermute (lane 0 is the world: identity)
      for (int r = 0; r < M.max_bdepth; r++) {      // max_bdepth = number of jumping rounds
        const int a = r < 4 ? (int)((jm >> (8 * r)) & 0xff) : (lane < nb ? (int)JMP[lane * M.anc_stride + r] : 0);
        const int src = a << 2;
#define PULL(v_) __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v_)))
        const v3 ap = mk3(PULL(xp.x), PULL(xp.y), PULL(xp.z));
        const q4 aqq = {PULL(xq.w), PULL(xq.x), PULL(xq.y), PULL(xq.z)};
#undef PULL
        xp = add3(ap, qrot(aqq, xp));
        xq = qmul(aqq, xq);
      }
      xq = qnormalize(xq);
      if (CONS && lane < nb) {
        *(float4*)(PO + lane * 8) = make_float4(xp.x, xp.y, xp.z, 0.f);
        *(float4*)(PO + lane * 8 + 4) = make_float4(xq.w, xq.x, xq.y, xq.z);
      }
    }
    STAMP(2);   // K
    v3 xi;
    {
      const float4 c_ipos = BTAB(blo, 2);
      xi = add3(xp, qrot(xq, mk3(c_ipos.x, c_ipos.y, c_ipos.z)));
    }
    // ---- C: tree CoM (wave reduction, result is wave-uniform), cinert, cdof
    const float mass = isb ? BTAB(blo, 0).w : 0.f;
    v3 com;
    com.x = wave_sum_fast(mass * xi.x) * M.mtot_inv;     // DPP reductions, no LDS traffic
    com.y = wave_sum_fast(mass * xi.y) * M.mtot_inv;
    com.z = wave_sum_fast(mass * xi.z) * M.mtot_inv;
    float iw[6];     // world-frame inertia about the body's own CoM
    {
      const float4 c_iquat = BTAB(blo, 3);
      const float4 c_inertia = BTAB(blo, 4);
      q4 iq = {c_iquat.x, c_iquat.y, c_iquat.z, c_iquat.w};
      const m33 Ri = q2m(any_iquat ? qmul(xq, iq) : xq);
      const float i0 = c_inertia.x, i1 = c_inertia.y, i2 = c_inertia.z;
      iw[0] = Ri.a[0] * Ri.a[0] * i0 + Ri.a[1] * Ri.a[1] * i1 + Ri.a[2] * Ri.a[2] * i2;
      iw[1] = Ri.a[3] * Ri.a[3] * i0 + Ri.a[4] * Ri.a[4] * i1 + Ri.a[5] * Ri.a[5] * i2;
      iw[2] = Ri.a[6] * Ri.a[6] * i0 + Ri.a[7] * Ri.a[7] * i1 + Ri.a[8] * Ri.a[8] * i2;
      iw[3] = Ri.a[0] * Ri.a[3] * i0 + Ri.a[1] * Ri.a[4] * i1 + Ri.a[2] * Ri.a[5] * i2;
      iw[4] = Ri.a[0] * Ri.a[6] * i0 + Ri.a[1] * Ri.a[7] * i1 + Ri.a[2] * Ri.a[8] * i2;
      iw[5] = Ri.a[3] * Ri.a[6] * i0 + Ri.a[4] * Ri.a[7] * i1 + Ri.a[5] * Ri.a[8] * i2;
      if (!isb) {
#pragma unroll
        for (int k = 0; k < 6; k++) iw[k] = 0.f;
      }
    }
    STAMP(3);   // C
    // ---- V: joint velocity vJ, then cvel = chain sum of vJ, cacc = a0 + chain sum of cvel_parent x vJ
    s6 cv, ca;
    {
      s6 vJ = {mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 0.f)};
      s6 vt = {mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 0.f)};       // translational part of a free root
      if (jtype == FMJ_JNT_HINGE || jtype == FMJ_JNT_SLIDE) {
        const v3 axw = qrot(xq, mk3(c_axis_q0.x, c_axis_q0.y, c_axis_q0.z));
        s6 cd;
        if (jtype == FMJ_JNT_HINGE) {
          const v3 anchor = any_jpos ? add3(xp, qrot(xq, mk3(c_jpos_k.x, c_jpos_k.y, c_jpos_k.z))) : xp;
          cd.r = axw; cd.l = cross(axw, sub3(com, anchor));
        } else { cd.r = mk3(0.f, 0.f, 0.f); cd.l = axw; }
        lds_put6(CD + dadr * 8, cd);
        vJ = s6scl(cd, QV[dadr]);
      } else if (jtype == FMJ_JNT_FREE) {
        const v3 off = sub3(com, xp);
        const m33 R = q2m(xq);
        vt.l = mk3(QV[dadr], QV[dadr + 1], QV[dadr + 2]);
#pragma unroll
        for (int k = 0; k < 3; k++) {
          s6 ct = {mk3(0.f, 0.f, 0.f), mk3(k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f)};
          lds_put6(CD + (dadr + k) * 8, ct);
          const v3 col = mk3(R.a[k], R.a[k + 3], R.a[k + 6]);
          s6 cr = {col, cross(col, off)};
          lds_put6(CD + (dadr + 3 + k) * 8, cr);
          vJ = s6add(vJ, s6scl(cr, QV[dadr + 3 + k]));
        }
      }
      const uint32_t jm = lane < nb ? *(const uint32_t*)(JMP + lane * M.anc_stride) : 0u;
      cv = s6add(vJ, vt);
#define PULL6(dst_, src_, v_) do { \
        dst_.r = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(src_, __float_as_int(v_.r.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(src_, __float_as_int(v_.r.y))), \
                     __int_as_float(__builtin_amdgcn_ds_bpermute(src_, __float_as_int(v_.r.z)))); \
        dst_.l = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(src_, __float_as_int(v_.l.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(src_, __float_as_int(v_.l.y))), \
                     __int_as_float(__builtin_amdgcn_ds_bpermute(src_, __float_as_int(v_.l.z)))); } while (0)
      for (int r = 0; r < M.max_bdepth; r++) {
        const int a = r < 4 ? (int)((jm >> (8 * r)) & 0xff) : (lane < nb ? (int)JMP[lane * M.anc_stride + r] : 0);
        s6 o; PULL6(o, a << 2, cv);
        cv = s6add(cv, o);
      }
      // w = cdof_dot * qvel of this body's joint = cvel(before the joint's rotary part) x vJ
      s6 cpar; PULL6(cpar, (isb ? c_info.x : 0) << 2, cv);
      cpar = s6add(cpar, vt);
      ca = cross_motion(cpar, vJ);
      if (!isb) { ca.r = ca.l = mk3(0.f, 0.f, 0.f); }
      for (int r = 0; r < M.max_bdepth; r++) {
        const int a = r < 4 ? (int)((jm >> (8 * r)) & 0xff) : (lane < nb ? (int)JMP[lane * M.anc_stride + r] : 0);
        s6 o; PULL6(o, a << 2, ca);
        ca = s6add(ca, o);
      }
#undef PULL6
      ca.l = sub3(ca.l, mk3(M.gx, M.gy, M.gz));
      if (!isb) { cv.r = cv.l = mk3(0.f, 0.f, 0.f); }
    }
    STAMP(4);   // V
    // ---- F: body force (inertial minus external), about the common point
    s6 fbody;
    {
      // cinert * v with cinert = {Iw, d = xi - com, m} (MuJoCo's cinert about the tree CoM, never materialised):
      // lin = p = m (u + w x d),  rot = Iw w + d x p
      const v3 d = sub3(xi, com);
      s6 ia, iv;
      ia.l = scl3(add3(ca.l, cross(ca.r, d)), mass);
      ia.r = add3(mk3(iw[0] * ca.r.x + iw[3] * ca.r.y + iw[4] * ca.r.z, iw[3] * ca.r.x + iw[1] * ca.r.y + iw[5] * ca.r.z,
                      iw[4] * ca.r.x + iw[5] * ca.r.y + iw[2] * ca.r.z), cross(d, ia.l));
      iv.l = scl3(add3(cv.l, cross(cv.r, d)), mass);
      iv.r = add3(mk3(iw[0] * cv.r.x + iw[3] * cv.r.y + iw[4] * cv.r.z, iw[3] * cv.r.x + iw[1] * cv.r.y + iw[5] * cv.r.z,
                      iw[4] * cv.r.x + iw[5] * cv.r.y + iw[2] * cv.r.z), cross(d, iv.l));
      s6 f = s6add(ia, cross_force(cv, iv));
      const v3 fw = mk3(xf[0], xf[1], xf[2]), tw = mk3(xf[3], xf[4], xf[5]);
      f.r = sub3(f.r, add3(tw, cross(sub3(xi, com), fw)));
      f.l = sub3(f.l, fw);
      if (!isb) { f.r = f.l = mk3(0.f, 0.f, 0.f); }
      fbody = f;
    }
    // ---- sensors of this (pre-integration) state; they are next iteration's link data (mj_step lag)
    {
      const v3 linvel = add3(cv.l, cross(cv.r, sub3(xi, com)));
      if (FUSED && !last && (nfull || (A.sub_links && !(A.n_it_total > 0 && nit >= A.n_it_total)))) {      // the next before_step's links row and drag (xf is consumed above, in this step's F);
        const int4 ci2 = BTABI(blo, 8);                     // a sub-step that writes no row keeps the drag force it has
        emit_links_and_drag(M, A, env, nit, isb, false, ci2.z, ci2.w, xp, xq, xi, linvel, cv.r, xf);
      }
      if (last && lane < nb) {
        float* p = glob(A.xpos) + (size_t)env * nb * 3 + lane * 3; p[0] = xp.x; p[1] = xp.y; p[2] = xp.z;
        *(float4*)(glob(A.xquat) + (size_t)env * nb * 4 + lane * 4) = make_float4(xq.w, xq.x, xq.y, xq.z);
        float* ip = glob(A.xipos) + (size_t)env * nb * 3 + lane * 3; ip[0] = xi.x; ip[1] = xi.y; ip[2] = xi.z;
        if (isb) {
          float* sp = glob(A.sensordata) + (size_t)env * M.nsensordata + 6 * (lane - 1);
          *(float2*)(sp) = make_float2(linvel.x, linvel.y);
          *(float2*)(sp + 2) = make_float2(linvel.z, cv.r.x);
          *(float2*)(sp + 4) = make_float2(cv.r.y, cv.r.z);
        }
      }
    }
    WSYNC();
    STAMP(5);   // F + carry
    // ---- S: subtree sums over the contiguous DFS id range [lane, lane + subsize) as prefix-sum differences
    // (Euler-tour trick).  The sums are taken about the tree CoM in fp64 - a subtree total is the difference
    // of two prefixes that can be 1e4 times larger, which fp32 cannot carry - and the composite inertia is
    // then moved to the subtree's own CoM s (parallel-axis shift, still fp64), so everything M sees is local.
    {
      const int last = isb ? lane + BTABI(blo, 8).y - 1 : lane;
      const double dm = (double)mass;
      const double dx = (double)xi.x - (double)com.x, dy = (double)xi.y - (double)com.y, dz = (double)xi.z - (double)com.z;
      const double ms = (double)BTAB(blo, 2).w;                              // subtree mass: a model constant
      const double px = subtree_sum_f64(dm * dx, last), py = subtree_sum_f64(dm * dy, last), pz = subtree_sum_f64(dm * dz, last);
      const double minv = ms > 0.0 ? rcp_f64_nr(ms) : 0.0;
      const double ex = px * minv, ey = py * minv, ez = pz * minv;           // subtree CoM relative to the tree CoM
      // one scan at a time: pinf() (asm volatile) fences keep the compiler from forming all sixteen fp64 inputs
      // up front, which would cost ~50 VGPRs; each result is converted to fp32 at once
#define SCAN(expr) pinf((float)(expr))
      const float m0 = pinf(mass), d0 = pinf((float)dx), d1 = pinf((float)dy), d2 = pinf((float)dz);
      (void)m0; (void)d0; (void)d1; (void)d2;
      const float i0 = SCAN(subtree_sum_f64((double)pinf(iw[0]) + dm * (dy * dy + dz * dz), last) - ms * (ey * ey + ez * ez));
      const float i1 = SCAN(subtree_sum_f64((double)pinf(iw[1]) + dm * (dx * dx + dz * dz), last) - ms * (ex * ex + ez * ez));
      const float i2 = SCAN(subtree_sum_f64((double)pinf(iw[2]) + dm * (dx * dx + dy * dy), last) - ms * (ex * ex + ey * ey));
      const float i3 = SCAN(subtree_sum_f64((double)pinf(iw[3]) - dm * dx * dy, last) + ms * ex * ey);
      const float i4 = SCAN(subtree_sum_f64((double)pinf(iw[4]) - dm * dx * dz, last) + ms * ex * ez);
      const float i5 = SCAN(subtree_sum_f64((double)pinf(iw[5]) - dm * dy * dz, last) + ms * ey * ez);
      s6 fs;
      fs.r.x = SCAN(subtree_sum_f64((double)pinf(fbody.r.x), last));
      fs.r.y = SCAN(subtree_sum_f64((double)pinf(fbody.r.y), last));
      fs.r.z = SCAN(subtree_sum_f64((double)pinf(fbody.r.z), last));
      fs.l.x = SCAN(subtree_sum_f64((double)pinf(fbody.l.x), last));
      fs.l.y = SCAN(subtree_sum_f64((double)pinf(fbody.l.y), last));
      fs.l.z = SCAN(subtree_sum_f64((double)pinf(fbody.l.z), last));
#undef SCAN
      if (lane < nb) {
        *(float4*)(CI + lane * 12) = make_float4(i0, i1, i2, i3);
        *(float4*)(CI + lane * 12 + 4) = make_float4(i4, i5, (float)((double)com.x + ex), (float)((double)com.y + ey));
        *(float2*)(CI + lane * 12 + 8) = make_float2((float)((double)com.z + ez), (float)ms);
        lds_put6(F + lane * 8, fs);
      }
    }
    WSYNC();
    STAMP(6);   // S
    // ---- Q: qfrc_smooth, buf = (I_s w, m v(s))  (lane = dof)
    float qfrc = 0.f;
    float dvel = 0.f;                               // implicitfast: velocity gains of this dof's unclamped actuators
    float af0 = 0.f, af1 = 0.f, af2 = 0.f, af3 = 0.f;
    const float4 d_prm = DTAB(dlo, 1);             // armature, damping, qposadr bits, hinge/slide flag
    const int4 d_act = DTABI(dlo, 2);               // first actuator, count, joint sensor slot
    const int d_qadr = __float_as_int(d_prm.z);
    const bool d_scalar = isd && d_prm.w != 0.f;
    s6 cd = {mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 0.f)}, bf = cd;     // this dof's cdof; (I_s w_i, m v_i(s)), s = CoM of the subtree it moves
    v3 sc = mk3(0.f, 0.f, 0.f);                                    // s relative to the tree CoM
    if (isd) {
      const int body = DTABI(dlo, 0).x;
      cd = lds_get6(CD + lane * 8);
      {
        const float4 a = *(const float4*)(CI + body * 12), b = *(const float4*)(CI + body * 12 + 4);
        const float2 c = *(const float2*)(CI + body * 12 + 8);
        sc = sub3(mk3(b.z, b.w, c.x), com);
        const v3 vs = add3(cd.l, cross(cd.r, sc));                 // velocity of the subtree CoM per unit dof rate
        bf.r = mk3(a.x * cd.r.x + a.w * cd.r.y + b.x * cd.r.z, a.w * cd.r.x + a.y * cd.r.y + b.y * cd.r.z, b.x * cd.r.x + b.y * cd.r.y + a.z * cd.r.z);
        bf.l = scl3(vs, c.y);
      }
      const float qd = QV[lane];
      qfrc = -d_prm.y * qd - s6dot(cd, lds_get6(F + body * 8));
      if (d_scalar) {
        const float qj = QP[d_qadr];
        if (M.any_stiffness) {
          const float kst = BTAB(body, 6).w;
          if (kst != 0.f) qfrc -= kst * (qj - gptr(A.qpos_spring)[(size_t)env * nq + d_qadr]);
        }
        float asum = 0.f;
        float cbase = 0.f;
        if (FUSED && A.controller == 1) {
          // phase in cycles kept in fp64 so long runs keep the argument exact (task.py:290: time = iteration*timestep)
          double cyc = (double)A.w_freq * ((double)it * ((double)M.h * (double)S_sub));   // task.py:290: time = iteration * timestep (of an iteration)
          cyc -= floor(cyc);
          cbase = 6.283185307179586f * (float)cyc + gptr(A.w_env)[env];
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {                 // mj_fwdActuation, joint transmission
          if (a < d_act.y) {
            const int ai = d_act.x + a, src = __float_as_int(ATAB(ai, 2).x);
            const float4 p = ATAB(ai, 0), lim = ATAB(ai, 1);
            float c;
            if (FUSED && A.controller == 1) { const float amp = gptr(A.w_amp)[src]; c = amp != 0.f ? amp * sinf(cbase - gptr(A.w_lag)[src]) : 0.f; }
            else c = A.ctrl ? gptr(A.ctrl)[(size_t)itm * A.ctrl_step_stride + (size_t)env * nu + src] : 0.f;      // one ctrl row per iteration
            c = fminf(fmaxf(c, lim.x), lim.y);
            float f = p.x * c + p.y + p.z * qj + p.w * qd;
            if (M.implicitfast && !A.disable_actuation && f > lim.z && f < lim.w) dvel -= p.w;      // d force / d qvel of an unclamped actuator (mjd_actuator_vel)
            f = fminf(fmaxf(f, lim.z), lim.w);
            if (A.disable_actuation) f = 0.f;
            if (a == 0) af0 = f; else if (a == 1) af1 = f; else if (a == 2) af2 = f; else af3 = f;
            asum += f;
          }
        }
        qfrc += asum;
        cy_actsum = asum * A.inv_torques;
        if (FUSED && A.controller == 1 && last && A.ctrl_out) {     // what task.py:288-346 leaves in physics.data.ctrl
#pragma unroll
          for (int a = 0; a < 4; a++) if (a < d_act.y) {
            const int src = __float_as_int(ATAB(d_act.x + a, 2).x);
            const float amp = gptr(A.w_amp)[src];
            gptr(A.ctrl_out)[(size_t)env * nu + src] = amp != 0.f ? amp * sinf(cbase - gptr(A.w_lag)[src]) : 0.f;
          }
        }
        if (last) {
          float* sa = glob(A.sensordata) + (size_t)env * M.nsensordata + 6 * (nb - 1) + 3 * M.njs;   // actuatorfrc
          if (0 < d_act.y) sa[__float_as_int(ATAB(d_act.x + 0, 2).x)] = af0;
          if (1 < d_act.y) sa[__float_as_int(ATAB(d_act.x + 1, 2).x)] = af1;
          if (2 < d_act.y) sa[__float_as_int(ATAB(d_act.x + 2, 2).x)] = af2;
          if (3 < d_act.y) sa[__float_as_int(ATAB(d_act.x + 3, 2).x)] = af3;
        }
      }
    }
    WSYNC();
    STAMP(7);   // Q
    // ---- M: row i of M (lane = dof i), one entry per depth of the chain root -> i, born in registers:
    //      M[i][j] = w_j . (I_s w_i) + v_j(s) . (m v_i(s)), s = CoM of the subtree dof i moves, j = ancestor of i at that depth;
    //      v_j(s) = v_j + w_j x sc, so M[i][j] = w_j . (I_s w_i + sc x p_i) + v_j . p_i with p_i = m v_i(s): the bracket gi is
    //      the lane's own and an entry costs two dot products.  Slots at and past the lane's own depth hold finite values
    //      that are never read as matrix entries (beyond the chain the table names the lane itself); the diagonals live in
    //      hdg_m (+ armature) and hdg_h (+ armature + h damping).  Lanes without a dof have cd = 0: their rows are zero.
    f2_t hrow[MAXD / 2];
    float hdg_m, hdg_h;
    {
      const v3 gi = add3(bf.r, cross(sc, bf.l));
      const float mii = dot3(cd.r, gi) + dot3(cd.l, bf.l);
      hdg_m = isd ? mii + d_prm.x : 1.f;
      hdg_h = isd ? mii + (d_prm.x + M.hdamp * (d_prm.y + dvel)) : 1.f;
      const int maxdep = M.maxdep1;
#pragma unroll
      for (int g = 0; g < MAXD / 4; g++) {
        const uint32_t ab = gptr(M.ancl1)[(unsigned)dlo * (MAXD / 4) + g];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int d = 4 * g + k;
          float mij = 0.f;
          if (d <= maxdep) {                        // uniform test, static register index
            const int al = (int)((ab >> (8 * k)) & 0xffu) >> 2;            // lane = dof of the ancestor at depth d
            const s6 cdj = lds_get6(CD + al * 8);
            mij = dot3(cdj.r, gi) + dot3(cdj.l, bf.l);
          }
          if (k & 1) hrow[d / 2].y = mij; else hrow[d / 2].x = mij;
        }
      }
    }
    WSYNC();                                        // without constraints the published rows overlay CD / F / CI from here
    STAMP(8);   // M
    if (!FUSED && A.dbg_H) {      // fmj_forward_debug: the assembled rows of H and the right-hand side, before any factorisation
      if (isd) {
#pragma unroll
        for (int d = 0; d < MAXD; d++) {
          const float v = (d & 1) ? hrow[d / 2].y : hrow[d / 2].x;
          gptr(A.dbg_H)[((size_t)env * nv + lane) * RS + d] = d < ddepth ? v : (d == ddepth ? hdg_h : 0.f);
        }
        gptr(A.dbg_qfrc)[(size_t)env * nv + lane] = qfrc;
      }
    }
    // ---- constraints (CONS instantiation only): qfrc_constraint from limits + plane contacts via PGS
    float qfrc_c = 0.f;
    float dinv_h = 0.f;               // 1 / D of the factor of H
    if (CONS) {
      const float h = M.h;
      // (1) factor M and H (together, see ldl_factor2), qacc_smooth = M^-1 qfrc_smooth
      float dinv_m;
      float xs = isd ? qfrc : 0.f;
      f2_t mrow[MAXD / 2];              // NEWTON: the rows of M itself stay in registers (the Hessian M + J'DJ is rebuilt from them)
      if (NEWTON) {
#pragma unroll
        for (int d = 0; d < MAXD / 2; d++) mrow[d] = hrow[d];
      }
      ldl_factor2<MAXD>(HM, HR, DI, XV, M.rounds1, M.nround1, lane, isd, ddepth, hrow, hdg_m, hdg_h, dinv_m, dinv_h, xs);
      xs = ldl_pull_sweep<MAXD>(HM, xs * dinv_m, isd ? lane : 0, isd, ddepth, M.ancl1, M.maxdep1);
      if (isd) { XS[lane] = xs; DI[lane] = dinv_m; SD[lane] = sqrtf(dinv_m); XV[lane] = dinv_h; }   // 1 / D of H waits in XV
      STAMP(12);  // factor M + qacc_smooth
      // (2) joint limit rows (mj_instantiateLimit): lane = dof, rows ordered by joint then side (-1, +1)
      const float4 lim = DTAB(dlo, 3);
      float dist_lo = 0.f, dist_hi = 0.f; bool act_lo = false, act_hi = false;
      if (d_scalar && lim.x != 0.f) {
        const float qj = QP[d_qadr];
        dist_lo = qj - lim.y; dist_hi = lim.z - qj;
        act_lo = dist_lo < lim.w; act_hi = dist_hi < lim.w;
      }
      const unsigned long long m_lo = __ballot(act_lo), m_hi = __ballot(act_hi);
      const unsigned long long lt = (1ull << lane) - 1ull;
      const int e_lo = __popcll(m_lo & lt) + __popcll(m_hi & lt);
      const int e_hi = e_lo + (act_lo ? 1 : 0);
      const int nlim = __popcll(m_lo) + __popcll(m_hi);
      // (3) ground contacts: lane = geom; contacts ordered by (ground geom, geom, point) like the oracle.  A ground entry is a
      //     world-attached plane or heightfield; the heightfield is met as the plane of the grid triangle under each
      //     candidate point (ground_dist), so every contact carries its own normal.
      int ncon = 0;
      for (int pl = 0; pl < M.nplane; pl++) {
        const float4 pn = PTAB(pl, 0), pp = PTAB(pl, 1);
        for (int g0 = 0; g0 < M.ngeom; g0 += 64) {
          const int g = g0 + lane;
          // up to 4 contacts per geom: sphere 1, capsule 2 (segment ends), box the first 4 penetrating corners in
          // corner order (what the oracle's collide_ground does), cylinder its rim points
          int cnt = 0; v3 cq[4], nq[4]; float dq[4]; float rad = 0.f, mu = 0.f;
#pragma unroll
          for (int k = 0; k < 4; k++) { cq[k] = mk3(0.f, 0.f, 0.f); nq[k] = mk3(0.f, 0.f, 1.f); dq[k] = 0.f; }
          if (g < M.ngeom) {
            const int4 gi = GTABI(g, 0);
            if (gi.x == FMJ_GEOM_SPHERE || gi.x == FMJ_GEOM_CAPSULE) {
              const float4 gs = GTAB(g, 1), gp = GTAB(g, 2), gq = GTAB(g, 3);
              const float4 bp = *(const float4*)(PO + gi.y * 8), bq = *(const float4*)(PO + gi.y * 8 + 4);
              const q4 bqq = {bq.x, bq.y, bq.z, bq.w};
              const v3 cen = add3(mk3(bp.x, bp.y, bp.z), qrot(bqq, mk3(gp.x, gp.y, gp.z)));
              rad = gs.x; mu = fmaxf(fmaxf(pp.x, gs.w), 1e-5f);
              v3 ax = mk3(0.f, 0.f, 0.f);
              if (gi.x == FMJ_GEOM_CAPSULE) { const q4 gqq = {gq.x, gq.y, gq.z, gq.w}; ax = scl3(qrot(qmul(bqq, gqq), mk3(0.f, 0.f, 1.f)), gs.y); }
              const v3 c0 = add3(cen, ax), c1 = sub3(cen, ax);
              v3 n0, n1;
              const float d0 = ground_dist(M, pl, pn, pp, c0, &n0) - rad;
              const float d1 = ground_dist(M, pl, pn, pp, c1, &n1) - rad;
              const bool a0 = d0 < 0.f, a1 = gi.x == FMJ_GEOM_CAPSULE && d1 < 0.f;
              if (a0) { cq[0] = c0; dq[0] = d0; nq[0] = n0; cnt = 1; }
              if (a1) { if (cnt == 0) { cq[0] = c1; dq[0] = d1; nq[0] = n1; } else { cq[1] = c1; dq[1] = d1; nq[1] = n1; } cnt++; }
            }
          }
          if (M.any_box) {
            const int4 gi = g < M.ngeom ? GTABI(g, 0) : make_int4(-1, 0, 0, 0);
            if (gi.x == FMJ_GEOM_BOX) {
              const float4 gs = GTAB(g, 1), gp = GTAB(g, 2), gq = GTAB(g, 3);
              const float4 bp = *(const float4*)(PO + gi.y * 8), bq = *(const float4*)(PO + gi.y * 8 + 4);
              const q4 bqq = {bq.x, bq.y, bq.z, bq.w}, gqq = {gq.x, gq.y, gq.z, gq.w};
              const q4 wq = qmul(bqq, gqq);
              const v3 cen = add3(mk3(bp.x, bp.y, bp.z), qrot(bqq, mk3(gp.x, gp.y, gp.z)));
              const v3 ex = scl3(qrot(wq, mk3(1.f, 0.f, 0.f)), gs.x), ey = scl3(qrot(wq, mk3(0.f, 1.f, 0.f)), gs.y), ez = scl3(qrot(wq, mk3(0.f, 0.f, 1.f)), gs.z);
              mu = fmaxf(fmaxf(pp.x, gs.w), 1e-5f);
#pragma unroll
              for (int corner = 0; corner < 8; corner++) {
                const v3 c = add3(add3(cen, (corner & 1) ? ex : scl3(ex, -1.f)), add3((corner & 2) ? ey : scl3(ey, -1.f), (corner & 4) ? ez : scl3(ez, -1.f)));
                v3 nc;
                const float d = ground_dist(M, pl, pn, pp, c, &nc);
                const bool pen = d < 0.f && cnt < 4;
#pragma unroll
                for (int k = 0; k < 4; k++) if (pen && cnt == k) { cq[k] = c; dq[k] = d; nq[k] = nc; }
                cnt += pen ? 1 : 0;
              }
            }
            if (gi.x == FMJ_GEOM_CYLINDER) {      // rim points (the oracle's collide_ground, MuJoCo's mjc_PlaneCylinder construction)
              const float4 gs = GTAB(g, 1), gp = GTAB(g, 2), gq = GTAB(g, 3);
              const float4 bp = *(const float4*)(PO + gi.y * 8), bq = *(const float4*)(PO + gi.y * 8 + 4);
              const q4 bqq = {bq.x, bq.y, bq.z, bq.w}, gqq = {gq.x, gq.y, gq.z, gq.w};
              const q4 wq = qmul(bqq, gqq);
              const v3 cen = add3(mk3(bp.x, bp.y, bp.z), qrot(bqq, mk3(gp.x, gp.y, gp.z)));
              v3 nrm_;                                         // a heightfield is taken as the plane under the cylinder's centre
              const float dist = ground_dist(M, pl, pn, pp, cen, &nrm_);
              v3 axis = qrot(wq, mk3(0.f, 0.f, 1.f));
              float prjaxis = dot3(nrm_, axis);
              if (prjaxis > 0.f) { axis = scl3(axis, -1.f); prjaxis = -prjaxis; }
              v3 vec = sub3(scl3(axis, prjaxis), nrm_);
              const float len2 = dot3(vec, vec);
              if (len2 >= 1e-30f) vec = scl3(vec, gs.x / sqrtf(len2)); else vec = scl3(qrot(wq, mk3(1.f, 0.f, 0.f)), gs.x);
              const float prjvec = dot3(vec, nrm_);
              axis = scl3(axis, gs.y); prjaxis *= gs.y;
              mu = fmaxf(fmaxf(pp.x, gs.w), 1e-5f);
              const float d0 = dist + prjaxis + prjvec;
              if (d0 < 0.f) {
                cq[0] = add3(cen, add3(vec, axis)); dq[0] = d0; cnt = 1;
                const float d1 = dist - prjaxis + prjvec;
                if (d1 < 0.f) { cq[1] = add3(cen, sub3(vec, axis)); dq[1] = d1; cnt = 2; }
                v3 vec1 = cross(vec, axis);
                const float l1 = sqrtf(dot3(vec1, vec1));
                if (l1 > 1e-15f) vec1 = scl3(vec1, gs.x * 0.8660254037844386f / l1);
                const float prjvec1 = dot3(vec1, nrm_);
#pragma unroll
                for (int sg = 0; sg < 2; sg++) {
                  const float sgn = sg ? -1.f : 1.f;
                  const float d2 = dist + prjaxis - 0.5f * prjvec + sgn * prjvec1;
                  const v3 c2 = add3(cen, add3(scl3(vec1, sgn), sub3(axis, scl3(vec, 0.5f))));
                  const bool pen = d2 < 0.f;
#pragma unroll
                  for (int k = 1; k < 4; k++) if (pen && cnt == k) { cq[k] = c2; dq[k] = d2; }
                  cnt += pen ? 1 : 0;
                }
              }
#pragma unroll
              for (int k = 0; k < 4; k++) nq[k] = nrm_;
            }
          }
          if (MESH && M.any_mesh) {      // convex mesh: its deepest penetrating vertices, deepest first (include/fmj.h)
            const int4 gi = g < M.ngeom ? GTABI(g, 0) : make_int4(-1, 0, 0, 0);
            if (gi.x == FMJ_GEOM_MESH) {
              const float4 gs = GTAB(g, 1), gp = GTAB(g, 2), gq = GTAB(g, 3);
              const float4 bp = *(const float4*)(PO + gi.y * 8), bq = *(const float4*)(PO + gi.y * 8 + 4);
              const q4 bqq = {bq.x, bq.y, bq.z, bq.w}, gqq = {gq.x, gq.y, gq.z, gq.w};
              const v3 cen = add3(mk3(bp.x, bp.y, bp.z), qrot(bqq, mk3(gp.x, gp.y, gp.z)));
              mu = fmaxf(fmaxf(pp.x, gs.w), 1e-5f);
              v3 nc0;
              if (ground_dist(M, pl, pn, pp, cen, &nc0) < gs.z) {       // the ground is within the bounding radius
                const m33 R = q2m(qmul(bqq, gqq));
                const int v0 = __float_as_int(gs.x), nvert = __float_as_int(gs.y);
                for (int vtx = 0; vtx < nvert; vtx++) {
                  const float4 vv = ldg4(M.mesh_vert, (unsigned)(v0 + vtx));
                  v3 tc = add3(cen, mrot(R, mk3(vv.x, vv.y, vv.z))), tn;
                  float td = ground_dist(M, pl, pn, pp, tc, &tn);
                  if (td < 0.f) {
                    bool have = true;
#pragma unroll
                    for (int k = 0; k < 4; k++) {                       // carry the displaced entry down; a filled empty slot ends the walk
                      const bool empty = k >= cnt;
                      if (have && (empty || td < dq[k])) {
                        const float sd = dq[k]; const v3 sc = cq[k], sn = nq[k];
                        dq[k] = td; cq[k] = tc; nq[k] = tn;
                        td = sd; tc = sc; tn = sn;
                        have = !empty;
                      }
                    }
                    if (cnt < 4) cnt++;
                  }
                }
              }
            }
          }
          int before = 0, total = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) { const unsigned long long bk = __ballot(cnt > k); before += __popcll(bk & lt); total += __popcll(bk); }
          const int s0 = ncon + before;
          ncon += total;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int slot = s0 + k;
            if (k < cnt && slot < M.max_contacts) {
              // frame: x = normal, t1 from (0,1,0) or (0,0,1) made orthogonal, t2 = n x t1 (mju_makeFrame)
              const v3 nrm = nq[k];
              v3 t1 = (nrm.y < -0.5f || nrm.y > 0.5f) ? mk3(0.f, 0.f, 1.f) : mk3(0.f, 1.f, 0.f);
              t1 = sub3(t1, scl3(nrm, dot3(t1, nrm)));
              t1 = scl3(t1, 1.0f / sqrtf(dot3(t1, t1)));
              const v3 t2 = cross(nrm, t1);
              const v3 pos = sub3(cq[k], scl3(nrm, rad + 0.5f * dq[k]));
              float* ct = CT + slot * 16;
              *(float4*)(ct) = make_float4(pos.x, pos.y, pos.z, nrm.x);
              *(float4*)(ct + 4) = make_float4(nrm.y, nrm.z, t1.x, t1.y);
              *(float4*)(ct + 8) = make_float4(t1.z, t2.x, t2.y, t2.z);
              // geom | (last dof of its body's chain + 1) << 16: the Jacobian rows need the chain without a table read per contact
              *(float4*)(ct + 12) = make_float4(dq[k], mu, __int_as_float(g | ((GTABI(g, 0).z + 1) << 16)), pp.z);
            }
          }
        }
      }
      int ncg = ncon;                               // ground contacts; explicit pairs follow, in pair order (like the oracle)
      if (PAIRS && M.npair) {
        for (int p0 = 0; p0 < M.npair; p0 += 64) {
          const int pr = p0 + lane;
          int pcnt = 0; v3 pcq[4], pnq[4]; float pdq[4]; float mu = 0.f; int g1 = 0, g2 = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) { pcq[k] = mk3(0.f, 0.f, 0.f); pnq[k] = mk3(0.f, 0.f, 1.f); pdq[k] = 0.f; }
          if (pr < M.npair) {
            const float4 q0 = QTAB(pr, 0);
            g1 = __float_as_int(q0.x); g2 = __float_as_int(q0.y); mu = q0.z;
            const int t1 = GTABI(g1, 0).x, t2 = GTABI(g2, 0).x;
            if (M.any_polypair && !(geom_is_round(t1) && geom_is_round(t2))) pcnt = pair_polytope(M, PO, g1, g2, pcq, pnq, pdq);
            else {
            v3 cen[2], ax[2]; float half[2], rad[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
              const int g = k ? g2 : g1;
              const int4 gi = GTABI(g, 0);
              const float4 gs = GTAB(g, 1), gp = GTAB(g, 2), gq = GTAB(g, 3);
              const float4 bp = *(const float4*)(PO + gi.y * 8), bq = *(const float4*)(PO + gi.y * 8 + 4);
              const q4 bqq = {bq.x, bq.y, bq.z, bq.w}, gqq = {gq.x, gq.y, gq.z, gq.w};
              cen[k] = add3(mk3(bp.x, bp.y, bp.z), qrot(bqq, mk3(gp.x, gp.y, gp.z)));
              ax[k] = qrot(qmul(bqq, gqq), mk3(0.f, 0.f, 1.f));
              rad[k] = gs.x; half[k] = gi.x == FMJ_GEOM_CAPSULE ? gs.y : 0.f;
            }
            // closest points of the two segments (the oracle's segment_closest)
            const v3 d = sub3(cen[0], cen[1]);
            const float b = dot3(ax[0], ax[1]), da1 = dot3(d, ax[0]), da2 = dot3(d, ax[1]);
            const float det = 1.f - b * b;
            float sp, tp;
            if (det > 1e-9f) sp = fminf(fmaxf((b * da2 - da1) / det, -half[0]), half[0]);
            else {
              const float lo = fmaxf(-half[0], -da1 - half[1]), hi = fminf(half[0], -da1 + half[1]);
              sp = lo <= hi ? 0.5f * (lo + hi) : (fabsf(lo - half[0]) < fabsf(hi + half[0]) ? half[0] : -half[0]);
              sp = fminf(fmaxf(sp, -half[0]), half[0]);
            }
            tp = fminf(fmaxf(b * sp + da2, -half[1]), half[1]);
            sp = fminf(fmaxf(b * tp - da1, -half[0]), half[0]);
            const v3 p1 = add3(cen[0], scl3(ax[0], sp)), p2 = add3(cen[1], scl3(ax[1], tp));
            v3 n = sub3(p2, p1);
            const float len = sqrtf(dot3(n, n));
            const float dist = len - rad[0] - rad[1];
            n = len < 1e-15f ? mk3(0.f, 0.f, 1.f) : scl3(n, 1.0f / len);
            if (dist < 0.f) { pcnt = 1; pdq[0] = dist; pnq[0] = n; pcq[0] = add3(p1, scl3(n, rad[0] + 0.5f * dist)); }
            }
          }
          int before = 0, total = 0;                  // contacts in pair order, a pair's own in its own order (like the oracle)
#pragma unroll
          for (int k = 0; k < 4; k++) { const unsigned long long bk = __ballot(pcnt > k); before += __popcll(bk & lt); total += __popcll(bk); }
          const int s0 = ncon + before;
          ncon += total;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int slot = s0 + k;
            if (k < pcnt && slot < M.max_contacts) {
              const v3 nrm = pnq[k], pos = pcq[k];
              v3 t1 = (nrm.y < -0.5f || nrm.y > 0.5f) ? mk3(0.f, 0.f, 1.f) : mk3(0.f, 1.f, 0.f);
              t1 = sub3(t1, scl3(nrm, dot3(t1, nrm)));
              t1 = scl3(t1, 1.0f / sqrtf(dot3(t1, t1)));
              const v3 t2 = cross(nrm, t1);
              float* ct = CT + slot * 16;
              *(float4*)(ct) = make_float4(pos.x, pos.y, pos.z, nrm.x);
              *(float4*)(ct + 4) = make_float4(nrm.y, nrm.z, t1.x, t1.y);
              *(float4*)(ct + 8) = make_float4(t1.z, t2.x, t2.y, t2.z);
              *(float4*)(ct + 12) = make_float4(pdq[k], mu, __int_as_float(g2 | ((GTABI(g2, 0).z + 1) << 16)), __int_as_float(g1 | ((pr + 1) << 16)));
            }
          }
        }
      }
      if (ncon > M.max_contacts) { ncon = M.max_contacts; warn |= FMJ_WARN_CONTACTFULL; }
      if (ncg > ncon) ncg = ncon;
      // rows per contact: the four edges of the friction pyramid, or - elliptic cone, Newton / CG only - the three axes of the contact frame
      constexpr int crs = ELL ? 3 : 4;          // ELL: its own instantiation (with NEWTON and MESH), so that the pyramidal kernels carry none of it
      const int nefc = nlim + crs * ncon;
      WSYNC();
      STAMP(13);  // limits + contacts
      // (4) Jacobian rows, stored compactly: a row touches only the dofs on the chain from its body to the root, so
      //     YC[e][dd] is the entry at the chain's dof of depth dd (RS floats per row) and CHN[e] names the chain's last
      //     dof (+1).  Limit rows: +-1 at the dof.  Contact rows: n.Jp +- mu t.Jp, Jp column of dof d = cdof_lin + cdof_rot x (p - com).
      //     Up to LL.na rows everything stays on chip ("small": the rows overlay T/F, V/BUF and CI, dead by now); with
      //     more rows the row vectors, the per-row parameters and A live in a per-env HBM scratch.
      const int e_p0 = nlim + crs * ncg;            // first row of the pair contacts: their fork parts are rows e - e_p0 of YF
      const bool hasp = PAIRS && M.npair != 0 && ncon > ncg; // some pair contact is active in this env (uniform)
      // an env with an active pair contact (rare) takes the HBM path: the register path then carries no fork code at all
      constexpr bool ELLPGS = ELL && !NEWTON;          // PGS with elliptic cones: block updates on the explicit matrix, in the HBM copy of the row code only
      if (!ELLPGS && nefc <= LL.na && !hasp && M.noslip_iterations == 0) {
        constexpr bool small = true;
        float* const YC = YJ;
        float* const EP = EPL;
        float* const YF = YFL;
#include "fmj_cons_rows.inc"
      } else {
        constexpr bool small = false;
        float* const YC = glob(M.cons_z) + (size_t)env * M.maxefc * RS;
        float* const EP = glob(M.cons_rows) + (size_t)env * M.maxefc * 8;
        float* const YF = M.cons_zf ? glob(M.cons_zf) + (size_t)env * M.maxefc * RS : nullptr;
#include "fmj_cons_rows.inc"
      }
    }
    // ---- L + X: sparse L'DL of H = M + diag(armature + h*damping) and the solve H qacc = qfrc_smooth
    float my_qacc;
    {
      if (CONS) dinv_h = isd ? XV[lane] : 0.f;          // factored together with M
      float xr = isd ? qfrc : 0.f;                        // non-CONS: the right-hand side rides in the rounds of the factorisation
      if (!CONS) ldl_factor<MAXD>(HR, XV, M.rounds6, M.nround6, M.maxdep1, lane, isd, ddepth, hrow, hdg_h, dinv_h, xr);
      STAMP(9);   // L
      if (CONS) my_qacc = ldl_solve<MAXD>(HR, qfrc + qfrc_c, lane, isd, ddepth, dsub, nv, dinv_h, M.ancl1, M.maxdep1);
      else my_qacc = ldl_pull_sweep<MAXD>(HR, xr * dinv_h, isd ? lane : 0, isd, ddepth, M.ancl1, M.maxdep1);
    }
    STAMP(10);  // X
    // ---- semi-implicit Euler (mj_Euler with implicit joint damping)
    const float hstep = A.integrate ? M.h : 0.f;     // fmj_forward: mj_forward only
    const float pre_qd = isd ? QV[lane] : 0.f;
    const float nvel = pre_qd + hstep * my_qacc;
    if (isd) {
      if (!(fabsf(my_qacc) <= 1e10f)) warn |= FMJ_WARN_BADQACC;      // mj_checkAcc
      if (!(fabsf(nvel) <= 1e10f)) warn |= FMJ_WARN_BADQVEL;
      // the root position this step would commit is tested BEFORE anything is committed (the translational dofs of a free root are
      // dofs 0..2 at qpos 0..2): a step that raises BADQPOS leaves qpos, qvel and time exactly at the previous step's values
      if (M.root_free && lane < 3 && A.integrate && !(fabsf(QP[lane] + M.h * nvel) <= 1e10f)) warn |= FMJ_WARN_BADQPOS;
    }
    if (__any((warn & FMJ_WARN_FREEZE) != 0)) frozen = true;         // the state stays at its last finite values
    if (isd && !frozen) {
      XV[lane] = my_qacc;
      QV[lane] = nvel;
      if (d_scalar) {
        const float pre_q = QP[d_qadr];
        QP[d_qadr] = pre_q + hstep * nvel;
        if (last) {
          float* s = glob(A.sensordata) + (size_t)env * M.nsensordata + 6 * (nb - 1) + 3 * d_act.z;   // jointpos, jointvel, jointlimitfrc
          s[0] = pre_q; s[1] = pre_qd; if (!CONS) s[2] = 0.f;
        }
      }
    }
    if (!frozen) steps_done++;
    WSYNC();
    if (jtype == FMJ_JNT_FREE && A.integrate && !frozen) {     // free joint position update (lane = root body)
      // the new root position is tested before it is committed: a frozen env keeps its last finite state (include/fmj.h)
      const float nx = QP[qadr] + M.h * QV[dadr], ny = QP[qadr + 1] + M.h * QV[dadr + 1], nz = QP[qadr + 2] + M.h * QV[dadr + 2];
      if (!(fabsf(nx) <= 1e10f) || !(fabsf(ny) <= 1e10f) || !(fabsf(nz) <= 1e10f)) warn |= FMJ_WARN_BADQPOS;
      else {
        QP[qadr] = nx; QP[qadr + 1] = ny; QP[qadr + 2] = nz;
        const v3 w = mk3(QV[dadr + 3], QV[dadr + 4], QV[dadr + 5]);
        const float n2 = dot3(w, w), rn = rsqrt_nr(n2), n = n2 * rn;
        q4 qo = {QP[qadr + 3], QP[qadr + 4], QP[qadr + 5], QP[qadr + 6]};
        qo = qnormalize(qo);
        if (n2 >= 1e-30f) qo = qmul(qo, axisangle_small(scl3(w, rn), M.h * n));
        QP[qadr + 3] = qo.w; QP[qadr + 4] = qo.x; QP[qadr + 5] = qo.y; QP[qadr + 6] = qo.z;
      }
    }
    if (__any((warn & FMJ_WARN_BADQPOS) != 0)) frozen = true;   // no row of the next iteration is emitted from a bad position
    sub = nsub; itm += nfull ? 1 : 0;
    WSYNC();
    STAMP(11);  // Euler
  }

  // ---- store state ---------------------------------------------------------------------------------------
  float* oq = glob(A.qpos) + (size_t)env * nq;
  float* ov = glob(A.qvel) + (size_t)env * nv;
  for (int i = lane; i < nq; i += 64) oq[i] = QP[i];
  for (int i = lane; i < nv; i += 64) ov[i] = QV[i];
  if (A.qacc && steps_done > 0) for (int i = lane; i < nv; i += 64) gptr(A.qacc)[(size_t)env * nv + i] = XV[i];
  // mj_step saves qacc as the next warm start when it advances the state; mj_forward alone does not
  if (CONS && !frozen && A.integrate) for (int i = lane; i < nv; i += 64) gptr(A.qacc_warmstart)[(size_t)env * nv + i] = QW[i];
#ifdef FMJ_STAMPS
  if (env == 0 && lane == 0 && A.qacc) {
#pragma unroll
    for (int i = 0; i < NSTAMP; i++) gptr(A.qacc)[i] = stamp_acc[i];
  }
#endif
  if (A.time && lane == 0 && A.integrate) gptr(A.time)[env] += M.h * steps_done;
  if (__ballot(warn != 0)) {
    int w = warn;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w |= __shfl_xor(w, o, 64);
    if (lane == 0) gptr(A.status)[env] |= w;
  }
}

#undef M
#undef A
static_assert(FMJ_JOINT_POSITION == 0 && FMJ_JOINT_VELOCITY == 1 && FMJ_JOINT_TORQUE == 8 && FMJ_JOINT_LIMIT_FORCE == 9 && FMJ_JOINT_SIZE == 12, "the fused kernels store a joints row as three float4");
#include "fmj_dual2.inc"
#include "fmj_cons2.inc"

// ---------------------------------------------------------------------------------------------
// Build layout: this file is compiled once per register row length with -DFMJ_TU_MAXD=<4..32> (only the step-kernel
// instantiations of that MAXD and a getter for their host stubs) and once without it (standalone operators and all
// host code), in parallel, and the objects are linked into one libfmj_hip.so (farms_mujoco_amd/_lib.py).
#ifdef FMJ_TU_MAXD
#define FMJ_CAT2(a, b) a##b
#define FMJ_CAT(a, b) FMJ_CAT2(a, b)
extern "C" __attribute__((visibility("hidden"))) void* FMJ_CAT(fmj_tu_kernel_, FMJ_TU_MAXD)(int fused, int cons, int dual) {
#if FMJ_TU_MAXD > 32        // rows longer than 32: the unconstrained one-env kernel only (FMJ_MAXD_DEEP)
  (void)cons; (void)dual;
  return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, false> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, false>;
#elif defined(FMJ_DEV_DUAL2_ONLY)      // development build: only the fused two-env kernels (compile time)
  (void)cons;
  if (dual == 2) return fused ? (void*)fmj_step_dual2_kernel<true, FMJ_TU_MAXD, 4> : nullptr;
  if (dual == 4) return fused ? (void*)fmj_step_dual2_kernel<true, FMJ_TU_MAXD, 2> : nullptr;
  if (dual == 3) return fused ? (void*)fmj_step_dual2_kernel<true, FMJ_TU_MAXD, 3> : nullptr;
  return nullptr;
#elif defined(FMJ_DEV_CONS2_FUSED_ONLY)      // development build: the fused two-env constraint kernel alone (resource remarks, ISA listings)
  (void)cons;
  if (dual == 5) return fused ? (void*)fmj_step_cons2_kernel<true, FMJ_TU_MAXD> : nullptr;
  return nullptr;
#elif defined(FMJ_DEV_PGSOPT_ONLY)      // development build: the one-env constraint kernels of the PGS options (elliptic cone, noslip)
  if (dual) return nullptr;
  if (cons == 9) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, true, false, true, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, true, false, true, true>;
  if (cons == 8) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, false, false, true, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, false, false, true, true>;
  if (cons == 1) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true>;
  return nullptr;
#elif defined(FMJ_DEV_CONS2_ONLY)      // development build: only the two-env constraint kernel and its one-env fallback (compile time)
  if (dual == 5) return fused ? (void*)fmj_step_cons2_kernel<true, FMJ_TU_MAXD> : (void*)fmj_step_cons2_kernel<false, FMJ_TU_MAXD>;
  if (cons == 1 && dual == 0) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true>;
  return nullptr;
#else
  if (dual == 5) return fused ? (void*)fmj_step_cons2_kernel<true, FMJ_TU_MAXD> : (void*)fmj_step_cons2_kernel<false, FMJ_TU_MAXD>;
  if (dual >= 16 && !fused) return (void*)fmj_step_dual2_kernel<false, FMJ_TU_MAXD, 4, true>;      // single steps with the rare options (implicitfast): one register tier is enough, they are launch-bound
  if (dual == 16 + 2) return (void*)fmj_step_dual2_kernel<true, FMJ_TU_MAXD, 4, true>;      // fused launches with sub-steps / implicitfast
  if (dual == 16 + 4) return (void*)fmj_step_dual2_kernel<true, FMJ_TU_MAXD, 2, true>;
  if (dual == 16 + 3) return (void*)fmj_step_dual2_kernel<true, FMJ_TU_MAXD, 3, true>;
  if (dual == 2) return fused ? (void*)fmj_step_dual2_kernel<true, FMJ_TU_MAXD, 4> : (void*)fmj_step_dual2_kernel<false, FMJ_TU_MAXD, 4>;
  if (dual == 4) return fused ? (void*)fmj_step_dual2_kernel<true, FMJ_TU_MAXD, 2> : (void*)fmj_step_dual2_kernel<false, FMJ_TU_MAXD, 2>;
  if (dual == 3) return fused ? (void*)fmj_step_dual2_kernel<true, FMJ_TU_MAXD, 3> : (void*)fmj_step_dual2_kernel<false, FMJ_TU_MAXD, 3>;
  if (cons == 9) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, true, false, true, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, true, false, true, true>;
  if (cons == 8) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, false, false, true, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, false, false, true, true>;
  if (cons == 7) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, true, true, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, true, true, true>;
  if (cons == 6) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, false, true, true, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, false, true, true, true>;
  if (cons == 5) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, false, true, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, false, true, true>;
  if (cons == 4) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, false, false, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, false, false, true>;
  if (cons == 3) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, false, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, false, true>;
  if (cons == 2) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true, true>;
  if (cons) return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, true> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, true>;
  return fused ? (void*)fmj_step_kernel<true, FMJ_TU_MAXD, false> : (void*)fmj_step_kernel<false, FMJ_TU_MAXD, false>;
#endif
}
#else

// ---------------------------------------------------------------------------------------------
// standalone operators (same arithmetic as the fused loop; one wave per env)

// SwimmingHandler.step (reference drag.pyx:389-411): lane = swimming link
__device__ __forceinline__ void drag_rows_of_env(const DevModel& M, const StepArgs& A, const int env) {
  for (int s = threadIdx.x; s < M.ns; s += 64) {
    const float4 s0 = STAB(s, 0), s1 = STAB(s, 1), s2 = STAB(s, 2);
    const int li = __float_as_int(s2.y), xi = __float_as_int(s2.z), body = __float_as_int(s2.w);
    const float* row = A.links + ((size_t)env * M.n_links + li) * FMJ_LINK_SIZE;
    const float4 a = *(const float4*)(row), b = *(const float4*)(row + 4), c = *(const float4*)(row + 8),
                 d = *(const float4*)(row + 12), e = *(const float4*)(row + 16);
    v3 com = mk3(a.x, a.y, a.z);
    fq comq = {a.w, b.x, b.y, b.z};
    fq urdfq = {c.z, c.w, d.x, d.y};
    v3 lin = mk3(d.z, d.w, e.x), ang = mk3(e.y, e.z, e.w);
    v3 fo, to;
    const bool applied = drag_link(com, urdfq, comq, lin, ang, s0, s1, s2.x, A, &fo, &to);
    if (applied) {
      float* xr = A.xfrc + ((size_t)env * M.n_xfrc + xi) * FMJ_XFRC_SIZE;
      *(float2*)(xr + 0) = make_float2(fo.x, fo.y);
      *(float2*)(xr + 2) = make_float2(fo.z, to.x);
      *(float2*)(xr + 4) = make_float2(to.y, to.z);
    }
    if (A.xfrc_applied_out) {
      float* xa = A.xfrc_applied_out + ((size_t)env * M.nbody + body) * 6;
      if (applied) {
        v3 fw = fq_rot(fo, comq), tw = fq_rot(to, comq);
        xa[0] = fw.x * A.newtons; xa[1] = fw.y * A.newtons; xa[2] = fw.z * A.newtons;
        xa[3] = tw.x * A.torques; xa[4] = tw.y * A.torques; xa[5] = tw.z * A.torques;
      } else { for (int k = 0; k < 6; k++) xa[k] = 0.f; }
    }
  }
}
__global__ void __launch_bounds__(64) fmj_drag_kernel(const DevModel M, const StepArgs A) { drag_rows_of_env(M, A, blockIdx.x); }

// mj_RungeKutta(m, d, 4) (oracle rk4): the state update between the four forward passes of a step.  The passes themselves are launches of the
// step kernel with integrate = 0 (fmj_step's RK4 branch); this kernel advances qpos / qvel to the next stage's state X[s + 1] = X[0] (+) h A[s] F[s]
// (F[s] = the velocity the pass ran at and the acceleration it returned) and accumulates sum B[s] F[s]; after the fourth pass it commits
// X[0] (+) h sum B F, saves that pass's qacc as the warm start and advances time (mj_advance).  Lane = body (its joint), one wave per env.
struct RkArgs { float* q0; float* v0; float* sv; float* sa; int stage; };
__global__ void __launch_bounds__(64) fmj_rk4_stage_kernel(const DevModel M, const StepArgs A, const RkArgs R) {
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= A.n_envs) return;
  if ((A.status[env] & FMJ_WARN_FREEZE) != 0) return;            // a frozen env keeps its last finite state (include/fmj.h)
  const int nq = M.nq, nv = M.nv;
  float* q = A.qpos + (size_t)env * nq; float* v = A.qvel + (size_t)env * nv;
  const float* a = A.qacc + (size_t)env * nv;
  float* q0 = R.q0 + (size_t)env * nq; float* v0 = R.v0 + (size_t)env * nv;
  float* sv = R.sv + (size_t)env * nv; float* sa = R.sa + (size_t)env * nv;
  const int s = R.stage;
  const float rb = (s == 0 || s == 3) ? (1.0f / 6.0f) : (1.0f / 3.0f);
  const float ra = s < 2 ? 0.5f : 1.0f;
  const bool isb = lane >= 1 && lane < M.nbody;
  const int4 ci = BTABI(lane < M.nbody ? lane : 0, 7);           // parent, jtype, qadr, dadr
  const int jtype = isb ? ci.y : -1, qadr = ci.z, dadr = ci.w;
  const int nd = jtype == FMJ_JNT_FREE ? 6 : ((jtype == FMJ_JNT_HINGE || jtype == FMJ_JNT_SLIDE) ? 1 : 0);
  const int nqj = jtype == FMJ_JNT_FREE ? 7 : nd;
  float vs[6], as[6], svn[6], san[6], q0j[7];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const bool on = k < nd;
    vs[k] = on ? v[dadr + k] : 0.f; as[k] = on ? a[dadr + k] : 0.f;
    svn[k] = fmaf(rb, vs[k], (on && s > 0) ? sv[dadr + k] : 0.f); san[k] = fmaf(rb, as[k], (on && s > 0) ? sa[dadr + k] : 0.f);
  }
#pragma unroll
  for (int k = 0; k < 7; k++) q0j[k] = k < nqj ? (s == 0 ? q[qadr + k] : q0[qadr + k]) : 0.f;
  float v0j[6];
#pragma unroll
  for (int k = 0; k < 6; k++) v0j[k] = k < nd ? (s == 0 ? vs[k] : v0[dadr + k]) : 0.f;
  // the step the positions take from X[0], and the velocity they take it with: the stage's own (s < 3), the weighted sum (s == 3)
  const float hs = s < 3 ? M.h * ra : M.h;
  float vv[6], vn[6];
#pragma unroll
  for (int k = 0; k < 6; k++) { vv[k] = s < 3 ? vs[k] : svn[k]; vn[k] = fmaf(hs, s < 3 ? as[k] : san[k], v0j[k]); }
  if (nd > 0) {
    int warn = 0;                                                  // mj_checkAcc / mj_checkVel on what this stage commits
#pragma unroll
    for (int k = 0; k < 6; k++) if (k < nd) { if (!(fabsf(as[k]) <= 1e10f)) warn |= FMJ_WARN_BADQACC; if (!(fabsf(vn[k]) <= 1e10f)) warn |= FMJ_WARN_BADQVEL; }
    if (warn) { atomicOr(&A.status[env], warn); return; }
    if (s == 0) {
#pragma unroll
      for (int k = 0; k < 7; k++) if (k < nqj) q0[qadr + k] = q0j[k];
#pragma unroll
      for (int k = 0; k < 6; k++) if (k < nd) v0[dadr + k] = v0j[k];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) if (k < nd) { sv[dadr + k] = svn[k]; sa[dadr + k] = san[k]; v[dadr + k] = vn[k]; }
    if (jtype == FMJ_JNT_FREE) {
      q[qadr] = fmaf(hs, vv[0], q0j[0]); q[qadr + 1] = fmaf(hs, vv[1], q0j[1]); q[qadr + 2] = fmaf(hs, vv[2], q0j[2]);
      const v3 w = mk3(vv[3], vv[4], vv[5]);
      const float n2 = dot3(w, w), rn = rsqrt_nr(n2), n = n2 * rn;
      q4 qo = {q0j[3], q0j[4], q0j[5], q0j[6]};
      qo = qnormalize(qo);
      if (n2 >= 1e-30f) qo = qmul(qo, axisangle_small(scl3(w, rn), hs * n));
      q[qadr + 3] = qo.w; q[qadr + 4] = qo.x; q[qadr + 5] = qo.y; q[qadr + 6] = qo.z;
    } else q[qadr] = fmaf(hs, vv[0], q0j[0]);
    if (s == 3 && A.qacc_warmstart) {
#pragma unroll
      for (int k = 0; k < 6; k++) if (k < nd) A.qacc_warmstart[(size_t)env * nv + dadr + k] = as[k];
    }
  }
  if (s == 3 && lane == 0 && A.time) A.time[env] += M.h;
}

// drag_forces (reference drag.pyx:152-268) of one link in every env: thread = env, rows addressed by an env stride
__global__ void __launch_bounds__(256) fmj_drag_link_kernel(const int n_envs, const float* links_row, const long long links_stride,
                                                            float* xfrc_row, const long long xfrc_stride, const float4 c0,
                                                            const float4 c1, const float density, const StepArgs A, int* hydro) {
  const int env = blockIdx.x * 256 + threadIdx.x;
  if (env >= n_envs) return;
  const float* row = links_row + (size_t)env * links_stride;
  v3 com = mk3(row[FMJ_LINK_COM_POS], row[FMJ_LINK_COM_POS + 1], row[FMJ_LINK_COM_POS + 2]);
  fq comq = {row[FMJ_LINK_COM_QUAT], row[FMJ_LINK_COM_QUAT + 1], row[FMJ_LINK_COM_QUAT + 2], row[FMJ_LINK_COM_QUAT + 3]};
  fq urdfq = {row[FMJ_LINK_URDF_QUAT], row[FMJ_LINK_URDF_QUAT + 1], row[FMJ_LINK_URDF_QUAT + 2], row[FMJ_LINK_URDF_QUAT + 3]};
  v3 lin = mk3(row[FMJ_LINK_COM_LINVEL], row[FMJ_LINK_COM_LINVEL + 1], row[FMJ_LINK_COM_LINVEL + 2]);
  v3 ang = mk3(row[FMJ_LINK_COM_ANGVEL], row[FMJ_LINK_COM_ANGVEL + 1], row[FMJ_LINK_COM_ANGVEL + 2]);
  v3 fo, to;
  const bool applied = drag_link(com, urdfq, comq, lin, ang, c0, c1, density, A, &fo, &to);
  if (applied) {
    float* xr = xfrc_row + (size_t)env * xfrc_stride;
    xr[0] = fo.x; xr[1] = fo.y; xr[2] = fo.z; xr[3] = to.x; xr[4] = to.y; xr[5] = to.z;
  }
  if (hydro) hydro[env] = applied ? 1 : 0;
}

// physics2data (reference physics.py:527-545): lane = link row, then lane = joint row
__device__ __forceinline__ void physics2data_rows_of_env(const DevModel& M, const StepArgs& A, const int env, const int links_only,
                                                         const int* links_body, const int* joints_dof) {
  const int nb = M.nbody;
  const float* sd = A.sensordata + (size_t)env * M.nsensordata;
  for (int i = threadIdx.x; i < M.n_links; i += 64) {
    const int b = links_body[i];
    const float* p = A.xpos + ((size_t)env * nb + b) * 3;
    const float* q = A.xquat + ((size_t)env * nb + b) * 4;
    const float* ip = A.xipos + ((size_t)env * nb + b) * 3;
    const float* s = sd + 6 * (b - 1);
    float* row = A.links + ((size_t)env * M.n_links + i) * FMJ_LINK_SIZE;
    const float im = A.inv_meters, iv = A.inv_velocity, ia = A.inv_angvel;
    *(float4*)(row + 0) = make_float4(ip[0] * im, ip[1] * im, ip[2] * im, q[1]);
    *(float4*)(row + 4) = make_float4(q[2], q[3], q[0], p[0] * im);
    *(float4*)(row + 8) = make_float4(p[1] * im, p[2] * im, q[1], q[2]);
    *(float4*)(row + 12) = make_float4(q[3], q[0], s[0] * iv, s[1] * iv);
    *(float4*)(row + 16) = make_float4(s[2] * iv, s[3] * ia, s[4] * ia, s[5] * ia);
  }
  if (links_only) return;
  const float* sa = sd + 6 * (nb - 1) + 3 * M.njs;
  for (int i = threadIdx.x; i < M.n_joints; i += 64) {
    const int d = joints_dof[i];
    const float4 prm = DTAB(d, 1);
    const int4 act = DTABI(d, 2);
    float* row = A.joints + ((size_t)env * M.n_joints + i) * FMJ_JOINT_SIZE;
    row[FMJ_JOINT_POSITION] = A.qpos[(size_t)env * M.nq + __float_as_int(prm.z)];
    row[FMJ_JOINT_VELOCITY] = A.qvel[(size_t)env * M.nv + d] * A.inv_angvel;
    float t = 0.f;
    for (int a = 0; a < act.y; a++) t += sa[__float_as_int(ATAB(act.x + a, 2).x)] * A.inv_torques;
    row[FMJ_JOINT_TORQUE] = t;
    row[FMJ_JOINT_LIMIT_FORCE] = sd[6 * (nb - 1) + 3 * act.z + 2] * A.inv_torques;
  }
}
__global__ void __launch_bounds__(64) fmj_physics2data_kernel(const DevModel M, const StepArgs A, const int links_only,
                                                               const int* links_body, const int* joints_dof) {
  if (A.status[blockIdx.x] & FMJ_WARN_FREEZE) return;         // a frozen env writes no rows (include/fmj.h)
  physics2data_rows_of_env(M, A, blockIdx.x, links_only, links_body, joints_dof);
}

// ExperimentTask.before_step up to the host callbacks, in ONE launch (fmj_before_step): physics2data rows, cycontacts2data rows, then the
// swimming callback's drag from the links row just written (same wave: a barrier orders the stores and the loads) - the operators
// above, called one after the other, bit for bit.  flags: FMJ_BEFORE_*.
__global__ void __launch_bounds__(64) fmj_before_step_kernel(const DevModel M, const StepArgs A, const int flags, const int* links_body,
                                                              const int* joints_dof, const int n_rows, const int* geom_sensor,
                                                              const int n_pairs, const int* pairs) {
  const int env = blockIdx.x;
  if (A.status[env] & FMJ_WARN_FREEZE) return;
  if (flags & FMJ_BEFORE_ROWS) physics2data_rows_of_env(M, A, env, (flags & FMJ_BEFORE_LINKS_ONLY) ? 1 : 0, links_body, joints_dof);
  if ((flags & FMJ_BEFORE_CONTACTS) && !(flags & FMJ_BEFORE_LINKS_ONLY)) {
    const int nc = A.ncon[env];
    const float* C = A.contact + (size_t)env * M.max_contacts * 16;
    for (int row = threadIdx.x; row < n_rows; row += 64)
      contact_row(C, nc, row, geom_sensor, n_pairs, pairs, A.inv_newtons, A.inv_meters, A.contacts_rows + ((size_t)env * n_rows + row) * FMJ_CONTACT_SIZE);
  }
  if (flags & FMJ_BEFORE_DRAG) {
    __threadfence_block();
    __syncthreads();
    drag_rows_of_env(M, A, env);
  }
}


// cycontacts2data (reference sensors.pyx:140-190): lane = contact sensor row; every contact of the env is
// tested against the 4 keys (g1,g2,-1) (g2,g1,+1) (g1,-1,-1) (g2,-1,+1) of geompair2data (sensors.pyx:163-169).
__global__ void __launch_bounds__(64) fmj_contacts2data_kernel(const DevModel M, const StepArgs A, const int n_rows,
                                                                const int* geom_sensor, const int n_pairs, const int* pairs) {
  const int env = blockIdx.x;
  if (A.status && (A.status[env] & FMJ_WARN_FREEZE)) return;
  const int nc = A.ncon[env];
  const float* C = A.contact + (size_t)env * M.max_contacts * 16;
  for (int row = threadIdx.x; row < n_rows; row += 64)
    contact_row(C, nc, row, geom_sensor, n_pairs, pairs, A.inv_newtons, A.inv_meters,
                A.contacts_rows + ((size_t)env * n_rows + row) * FMJ_CONTACT_SIZE);
}

// ---------------------------------------------------------------------------------------------
// host side

template <class T>
static int upload(fmj_ctx* c, const std::vector<T>& h, const T** dptr) {
  void* d = nullptr;
  size_t bytes = (h.size() ? h.size() : 1) * sizeof(T);
  HIP_TRY(hipMalloc(&d, bytes));
  c->allocs.push_back(d);
  if (h.size()) HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *dptr = (const T*)d;
  return FMJ_OK;
}
#define UP(vec, field) do { int rc_ = upload(c, vec, &c->dm.field); if (rc_) { fmj_destroy(c); return rc_; } } while (0)

static float4 f4(double a, double b, double c, double d) { return make_float4((float)a, (float)b, (float)c, (float)d); }
static float ibits(int i) { float f; memcpy(&f, &i, 4); return f; }

// pick the instantiation whose register row length matches the model's dof-chain length
typedef void (*step_kernel_t)(const DevModel, const StepArgs);
extern "C" {
void* fmj_tu_kernel_4(int, int, int);  void* fmj_tu_kernel_8(int, int, int);  void* fmj_tu_kernel_12(int, int, int); void* fmj_tu_kernel_16(int, int, int);
void* fmj_tu_kernel_20(int, int, int); void* fmj_tu_kernel_24(int, int, int); void* fmj_tu_kernel_28(int, int, int); void* fmj_tu_kernel_32(int, int, int);
void* fmj_tu_kernel_36(int, int, int); void* fmj_tu_kernel_40(int, int, int); void* fmj_tu_kernel_44(int, int, int); void* fmj_tu_kernel_48(int, int, int);
void* fmj_tu_kernel_52(int, int, int); void* fmj_tu_kernel_56(int, int, int); void* fmj_tu_kernel_60(int, int, int); void* fmj_tu_kernel_64(int, int, int);
}
static step_kernel_t tu_kernel(int rs, bool fused, int cons, int dual) {      // cons: 0 none, 1 limits / ground contacts, 2 + explicit pairs (and meshes), 3 Newton / CG solver, 4 + meshes only, 5 Newton / CG + meshes, 6 Newton / CG + elliptic cone (+ meshes), 7 Newton / CG + explicit pairs (+ meshes), 8 PGS + elliptic cone (+ meshes), 9 PGS + elliptic cone + explicit pairs
  void* k;
  switch (rs) {
    case 4: k = fmj_tu_kernel_4(fused, cons, dual); break;
    case 8: k = fmj_tu_kernel_8(fused, cons, dual); break;
    case 12: k = fmj_tu_kernel_12(fused, cons, dual); break;
    case 16: k = fmj_tu_kernel_16(fused, cons, dual); break;
    case 20: k = fmj_tu_kernel_20(fused, cons, dual); break;
    case 24: k = fmj_tu_kernel_24(fused, cons, dual); break;
    case 28: k = fmj_tu_kernel_28(fused, cons, dual); break;
    case 32: k = fmj_tu_kernel_32(fused, cons, dual); break;
    case 36: k = fmj_tu_kernel_36(fused, cons, dual); break;
    case 40: k = fmj_tu_kernel_40(fused, cons, dual); break;
    case 44: k = fmj_tu_kernel_44(fused, cons, dual); break;
    case 48: k = fmj_tu_kernel_48(fused, cons, dual); break;
    case 52: k = fmj_tu_kernel_52(fused, cons, dual); break;
    case 56: k = fmj_tu_kernel_56(fused, cons, dual); break;
    case 60: k = fmj_tu_kernel_60(fused, cons, dual); break;
    default: k = fmj_tu_kernel_64(fused, cons, dual); break;
  }
  return (step_kernel_t)k;
}
static step_kernel_t pick_kernel(const fmj_ctx* c, bool fused) {
  const int cons = !c->dm.cons ? 0 : (c->dm.solver != FMJ_SOLVER_PGS ? (c->dm.cone == FMJ_CONE_ELLIPTIC ? 6 : (c->dm.npair > 0 ? 7 : (c->dm.any_mesh ? 5 : 3)))
                                      : (c->dm.cone == FMJ_CONE_ELLIPTIC ? (c->dm.npair > 0 ? 9 : 8) : (c->dm.npair > 0 ? 2 : (c->dm.any_mesh ? 4 : 1))));
  return tu_kernel(c->dm.rs, fused, cons, 0);
}
static int launch_step(fmj_ctx* c, bool fused, const StepArgs& A, void* stream) {
  if (c->dm.dual_ok && A.integrate) {      // two envs per wave (fmj_dual2.inc); fmj_forward keeps the single-env kernel
    step_kernel_t k = tu_kernel(c->dm.rs, fused, 0, (c->dual_wps == 2 ? 4 : (c->dual_wps == 3 ? 3 : 2)) + (((fused && A.substeps > 1) || c->dm.implicitfast) ? 16 : 0));      // + 16: the instantiation with the rare options (sub-steps, implicitfast)
    hipLaunchKernelGGL(k, dim3((c->n_envs + 1) / 2), dim3(64), c->lds_bytes_dual2, (hipStream_t)stream, c->dm, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(FMJ_ERR_HIP, std::string("dual step kernel launch: ") + hipGetErrorString(e));
    return FMJ_OK;
  }
  if (c->dm.cons && (!A.qacc_warmstart || !A.contact || !A.ncon))
    return set_err(FMJ_ERR_ARG, "fmj_data: qacc_warmstart, contact and ncon are required for models with limits / contacts");
  if (c->dm.cons2_ok && A.integrate && !(fused && A.do_drag)) {
    // two envs per wave (fmj_cons2.inc), then the one-env kernel for whatever that kernel handed over (resume[env] < n_steps: an
    // env with more rows than a wave holds on chip); envs it completed cost the second launch one early exit each
    StepArgs A2 = A;
    A2.resume = c->d_resume;
    hipLaunchKernelGGL(tu_kernel(c->dm.rs, fused, 0, 5), dim3((c->n_envs + 1) / 2), dim3(64), c->lds_bytes_cons2, (hipStream_t)stream, c->dm, A2);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return set_err(FMJ_ERR_HIP, std::string("two-env constraint kernel launch: ") + hipGetErrorString(e2));
    hipLaunchKernelGGL(pick_kernel(c, fused), dim3(c->n_envs), dim3(64), c->lds_bytes, (hipStream_t)stream, c->dm, A2);
    e2 = hipGetLastError();
    if (e2 != hipSuccess) return set_err(FMJ_ERR_HIP, std::string("step kernel launch: ") + hipGetErrorString(e2));
    return FMJ_OK;
  }
  hipLaunchKernelGGL(pick_kernel(c, fused), dim3(c->n_envs), dim3(64), c->lds_bytes, (hipStream_t)stream, c->dm, A);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err(FMJ_ERR_HIP, std::string("step kernel launch: ") + hipGetErrorString(e));
  return FMJ_OK;
}

// device copies of the row -> body / dof maps used by fmj_physics2data (rebuilt from the host mirrors; tiny)
static int sync_readout_maps(fmj_ctx* c) {
  std::vector<int> lb(c->dm.n_links ? c->dm.n_links : 1, 1), jd(c->dm.n_joints ? c->dm.n_joints : 1, 0);
  for (int b = 1; b < c->nbody; b++) if (c->h_b_info2[4 * b + 2] >= 0) lb[c->h_b_info2[4 * b + 2]] = b;
  for (int dd = 0; dd < c->nv; dd++) if (c->h_d_info[4 * dd + 3] >= 0) jd[c->h_d_info[4 * dd + 3]] = dd;
  const int* a = nullptr; const int* b2 = nullptr;
  int rc;
  if ((rc = upload(c, lb, &a)) || (rc = upload(c, jd, &b2))) return rc;
  c->d_links_body = (int*)a; c->d_joints_dof = (int*)b2;
  return FMJ_OK;
}

// re-pack the host-mutable int fields (link row / swim slot / joint row) and refresh the device tables
static int sync_tables(fmj_ctx* c) {
  for (int b = 0; b < 64; b++) c->h_btab[b * BT_STRIDE + 8] = make_float4(ibits(c->h_b_info2[4 * b]), ibits(c->h_b_info2[4 * b + 1]), ibits(c->h_b_info2[4 * b + 2]), ibits(c->h_b_info2[4 * b + 3]));
  for (int d = 0; d < 64; d++) c->h_dtab[d * DT_STRIDE] = make_float4(ibits(c->h_d_info[4 * d]), ibits(c->h_d_info[4 * d + 1]), ibits(c->h_d_info[4 * d + 2]), ibits(c->h_d_info[4 * d + 3]));
  HIP_TRY(hipMemcpy(c->d_btab, c->h_btab.data(), c->h_btab.size() * sizeof(float4), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_dtab, c->h_dtab.data(), c->h_dtab.size() * sizeof(float4), hipMemcpyHostToDevice));
  return FMJ_OK;
}

extern "C" {

const char* fmj_last_error(void) { return g_err.c_str(); }
int fmj_abi_version(void) { return FMJ_ABI_VERSION; }

void fmj_destroy(fmj_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (void* p : c->allocs) (void)hipFree(p);
  delete c;
}

int fmj_create(const fmj_model* m, int32_t n_envs, int32_t device, fmj_ctx** out) {
  if (!m || !out || n_envs <= 0) return set_err(FMJ_ERR_ARG, "fmj_create: NULL model/out or n_envs <= 0");
  *out = nullptr;
  if (m->abi_version != FMJ_ABI_VERSION) return set_err(FMJ_ERR_ARG, "fmj_create: abi_version mismatch");
  const int nb = m->nbody, nv = m->nv, nq = m->nq, nu = m->nu, nj = m->njnt;
  if (nb < 2 || nb > 64 || nv < 1 || nv > 64) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: need 2 <= nbody <= 64 and 1 <= nv <= 64 (one wavefront per environment)");
  int any_limit = 0, nplane = 0, any_box = 0, n_hfield = 0, any_mesh = 0, any_polypair = 0;     // nplane counts the ground geoms: planes and the heightfield
  for (int j = 0; j < nj; j++) if (m->jnt_limited[j] && m->jnt_type[j] != FMJ_JNT_FREE) any_limit = 1;
  for (int g = 0; g < m->ngeom; g++) {
    int t = m->geom_type[g];
    if (t == FMJ_GEOM_PLANE) { if (m->geom_bodyid[g] != 0) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: planes must be attached to the world body"); nplane++; }
    else if (t == FMJ_GEOM_HFIELD) {
      if (m->geom_bodyid[g] != 0) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: heightfields must be attached to the world body");
      if (n_hfield++) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: one heightfield geom per model");
      if (m->hfield_nrow < 2 || m->hfield_ncol < 2 || !m->hfield_data || !(m->hfield_size[0] > 0) || !(m->hfield_size[1] > 0))
        return set_err(FMJ_ERR_ARG, "fmj_create: heightfield needs nrow, ncol >= 2, data and positive x / y radii");
      nplane++;
    }
    else if (t == FMJ_GEOM_MESH) {
      if (m->nmeshvert < 1 || !m->mesh_vert || !m->geom_vertadr || !m->geom_vertnum) return set_err(FMJ_ERR_ARG, "fmj_create: mesh geom without mesh_vert / geom_vertadr / geom_vertnum");
      if (m->geom_vertnum[g] < 1 || m->geom_vertadr[g] < 0 || m->geom_vertadr[g] + m->geom_vertnum[g] > m->nmeshvert)
        return set_err(FMJ_ERR_ARG, "fmj_create: mesh vertex range out of bounds");
      if (m->geom_bodyid[g] < 1 || m->geom_bodyid[g] >= nb) return set_err(FMJ_ERR_ARG, "fmj_create: geom_bodyid out of range");
      any_mesh = 1;
    }
    else if (t != FMJ_GEOM_SPHERE && t != FMJ_GEOM_CAPSULE && t != FMJ_GEOM_BOX && t != FMJ_GEOM_CYLINDER) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: only plane / heightfield / sphere / capsule / cylinder / box / convex mesh geoms are in the HIP path");
    else if (m->geom_bodyid[g] < 1 || m->geom_bodyid[g] >= nb) return set_err(FMJ_ERR_ARG, "fmj_create: geom_bodyid out of range");
    if (t == FMJ_GEOM_BOX || t == FMJ_GEOM_CYLINDER) any_box = 1;      // geoms with up to 4 contacts
  }
  if (m->npair < 0 || (m->npair > 0 && (!m->pair_geom1 || !m->pair_geom2 || !m->pair_friction || !m->pair_solref || !m->pair_solimp)))
    return set_err(FMJ_ERR_ARG, "fmj_create: pair arrays missing");
  if (m->ngeom >= 65536 || m->npair >= 32767) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: too many geoms / pairs");
  for (int p = 0; p < m->npair; p++) {
    const int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    if (g1 < 0 || g1 >= m->ngeom || g2 < 0 || g2 >= m->ngeom) return set_err(FMJ_ERR_ARG, "fmj_create: pair geom out of range");
    const int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    for (int k = 0; k < 2; k++) {
      const int tt = k ? t2 : t1, gg = k ? g2 : g1;
      if (tt == FMJ_GEOM_PLANE || tt == FMJ_GEOM_HFIELD) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: explicit contact pairs join two animat geoms (sphere / capsule / box / cylinder / convex mesh); the ground meets every geom already");
      if (tt == FMJ_GEOM_BOX || tt == FMJ_GEOM_CYLINDER || tt == FMJ_GEOM_MESH) any_polypair = 1;
      if (tt == FMJ_GEOM_MESH && (m->nmeshface < 4 || !m->mesh_face || !m->geom_faceadr || !m->geom_facenum || m->geom_facenum[gg] < 4 ||
                                  m->geom_faceadr[gg] < 0 || m->geom_faceadr[gg] + m->geom_facenum[gg] > m->nmeshface))
        return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: an explicit pair with a mesh geom needs the planes of its convex hull (mesh_face / geom_faceadr / geom_facenum, at least 4: a flat or collinear vertex cloud has no hull)");
    }
    if (m->geom_bodyid[g1] == m->geom_bodyid[g2]) return set_err(FMJ_ERR_ARG, "fmj_create: a contact pair joins geoms of two bodies");
  }
  const int cons = any_limit || (nplane > 0 && m->ngeom > nplane) || m->npair > 0;
  if (cons && m->solver != FMJ_SOLVER_PGS && m->solver != FMJ_SOLVER_NEWTON && m->solver != FMJ_SOLVER_CG) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: solver must be FMJ_SOLVER_PGS, FMJ_SOLVER_CG or FMJ_SOLVER_NEWTON");
  if (cons && m->cone != FMJ_CONE_PYRAMIDAL && m->cone != FMJ_CONE_ELLIPTIC) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: cone must be FMJ_CONE_PYRAMIDAL or FMJ_CONE_ELLIPTIC");
  // Pyramid rows carry R = 2 mu^2 R0: below mu ~ 1e-3 (the reference's arena has friction 0, mjcf.py:1202, so a contact's friction is
  // its link's - 1e-5 after MuJoCo's clamp when the link has none) a contact force is a residual too small for an fp32 primal
  // iteration (see fmj_cons_rows.inc on the friction-0 pairs).  Such a model is solved on the dual problem throughout: the PGS
  // kernels, run to the solver's tolerance (up to 10 x solver_iterations sweeps) - the same convex problem, the same minimiser.
  bool dual_instead = false;
  if (cons && m->solver != FMJ_SOLVER_PGS && m->cone == FMJ_CONE_PYRAMIDAL && nplane > 0) {
    double gmu = 0;
    const double isq = 1.0 / sqrt(m->impratio > 0 ? m->impratio : 1.0);      // the rule uses mu = friction / sqrt(impratio)
    for (int g = 0; g < m->ngeom; g++) if (m->geom_type[g] == FMJ_GEOM_PLANE || m->geom_type[g] == FMJ_GEOM_HFIELD) gmu = std::max(gmu, m->geom_friction[3 * g]);
    for (int g = 0; g < m->ngeom; g++)
      if (m->geom_type[g] != FMJ_GEOM_PLANE && m->geom_type[g] != FMJ_GEOM_HFIELD && std::max(gmu, m->geom_friction[3 * g]) * isq < 1e-3) dual_instead = true;
  }
  // The noslip post-pass works on the dual matrices, which the primal solvers never form; explicit pairs under the elliptic cone have the
  // dual block update only (fmj_cons_rows.inc (8c)).  Such a model requested with Newton / CG is solved on the dual problem as well.
  if (cons && m->solver != FMJ_SOLVER_PGS && (m->noslip_iterations > 0 || (m->cone == FMJ_CONE_ELLIPTIC && m->npair > 0))) dual_instead = true;
  if (m->noslip_iterations < 0 || !(m->noslip_tolerance >= 0)) return set_err(FMJ_ERR_ARG, "fmj_create: noslip_iterations / noslip_tolerance must not be negative");
  if (m->integrator != FMJ_INT_EULER && m->integrator != FMJ_INT_IMPLICITFAST && m->integrator != FMJ_INT_RK4)
    return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: integrator must be FMJ_INT_EULER, FMJ_INT_IMPLICITFAST or FMJ_INT_RK4 (implicit keeps the Coriolis derivatives, a non-symmetric matrix outside this path's tree-sparse factorisation)");
  if (cons && (m->ngeom > nplane || m->npair > 0) && !any_limit && m->max_contacts < 1) return set_err(FMJ_ERR_ARG, "fmj_create: max_contacts must be >= 1 with collision geoms");
  if ((m->npair > 0 || (nplane > 0 && m->ngeom > nplane)) && m->max_contacts < 1) return set_err(FMJ_ERR_ARG, "fmj_create: max_contacts must be >= 1 with collision geoms");
  // structure checks: single tree rooted at body 1, DFS pre-order, <= 1 joint per body
  if (m->body_parentid[1] != 0) return set_err(FMJ_ERR_ARG, "fmj_create: body 1 must be the root (parent = world)");
  std::vector<int> bdepth(nb, 0), subsize(nb, 1);
  for (int i = 2; i < nb; i++) {
    int p = m->body_parentid[i];
    if (p < 1 || p >= i) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: single kinematic tree with parent-first numbering required");
    bool ok = false;                      // DFS pre-order: parent(i) is an ancestor-or-self of i-1
    for (int a = i - 1; a >= 1; a = m->body_parentid[a]) if (a == p) { ok = true; break; }
    if (!ok) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: bodies must be numbered in depth-first pre-order");
    bdepth[i] = bdepth[p] + 1;
  }
  for (int i = nb - 1; i >= 2; i--) subsize[m->body_parentid[i]] += subsize[i];
  int max_bdepth = 0, max_sub = 0;
  for (int i = 1; i < nb; i++) { if (bdepth[i] + 1 > max_bdepth) max_bdepth = bdepth[i] + 1; if (subsize[i] > max_sub) max_sub = subsize[i]; }
  if (max_bdepth > 255) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: body chain too long");
  int expect_dof = 0, expect_q = 0;
  for (int i = 1; i < nb; i++) {
    int j = m->body_jntadr[i];
    if (j < 0) { if (m->body_dofnum[i] != 0) return set_err(FMJ_ERR_ARG, "fmj_create: body_dofnum without joint"); continue; }
    if (m->jnt_bodyid[j] != i) return set_err(FMJ_ERR_ARG, "fmj_create: jnt_bodyid inconsistent");
    int t = m->jnt_type[j];
    int nd = t == FMJ_JNT_FREE ? 6 : 1, nqj = t == FMJ_JNT_FREE ? 7 : 1;
    if (t == FMJ_JNT_BALL) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: ball joints unsupported (reference mjcf.py emits hinge/slide/free only)");
    if (t == FMJ_JNT_FREE && i != 1) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: free joint only on the root body");
    if (m->body_dofnum[i] != nd || m->body_dofadr[i] != expect_dof || m->jnt_dofadr[j] != expect_dof || m->jnt_qposadr[j] != expect_q)
      return set_err(FMJ_ERR_ARG, "fmj_create: dof/qpos addresses must follow body order with one joint per body");
    expect_dof += nd; expect_q += nqj;
  }
  if (expect_dof != nv || expect_q != nq) return set_err(FMJ_ERR_ARG, "fmj_create: nv/nq do not match the joints");
  std::vector<int> ddepth(nv, 0), dsub(nv, 1);
  int max_ddepth = 0;
  for (int d = 0; d < nv; d++) {
    int p = m->dof_parentid[d];
    if (p >= d) return set_err(FMJ_ERR_ARG, "fmj_create: dof_parentid must precede the dof");
    ddepth[d] = p < 0 ? 0 : ddepth[p] + 1;
    if (ddepth[d] > max_ddepth) max_ddepth = ddepth[d];
  }
  for (int d = nv - 1; d >= 0; d--) if (m->dof_parentid[d] >= 0) dsub[m->dof_parentid[d]] += dsub[d];
  if (max_ddepth + 1 > (cons ? FMJ_MAXD : FMJ_MAXD_DEEP))
    return set_err(FMJ_ERR_UNSUPPORTED, cons ? "fmj_create: dof chain longer than 32 in a model with limits / contacts (64 without)" : "fmj_create: dof chain longer than 64");
  std::vector<int> nact(nj, 0);
  for (int a = 0; a < nu; a++) {
    int j = m->actuator_jntid[a];
    if (j < 0 || j >= nj || m->jnt_type[j] == FMJ_JNT_FREE) return set_err(FMJ_ERR_ARG, "fmj_create: actuator must act on a hinge/slide joint");
    if (++nact[j] > 4) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: more than 4 actuators on one joint");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_err(FMJ_ERR_NODEVICE, "fmj_create: no HIP device visible");
  if (device < 0 || device >= ndev) return set_err(FMJ_ERR_ARG, "fmj_create: bad device ordinal");
  HIP_TRY(hipSetDevice(device));

  fmj_ctx* c = new fmj_ctx();
  c->solver_requested = cons ? m->solver : FMJ_SOLVER_PGS;
  c->device = device; c->n_envs = n_envs; c->nbody = nb; c->nv = nv; c->nu = nu; c->njnt = nj;
  c->d_btab = nullptr; c->d_dtab = nullptr; c->d_links_body = nullptr; c->d_joints_dof = nullptr;
  DevModel& D = c->dm;
  memset(&D, 0, sizeof D);
  D.nbody = nb; D.nv = nv; D.nq = nq; D.nu = nu; D.njnt = nj; D.nM = m->nM;
  { int rounds = 0; while ((1 << rounds) < max_bdepth) rounds++; D.max_bdepth = rounds; }   // pointer-jumping rounds
  D.max_subsize = max_sub;
  D.rs = r4(max_ddepth + 1);
  D.root_free = m->body_jntadr[1] >= 0 && m->jnt_type[m->body_jntadr[1]] == FMJ_JNT_FREE;
  D.h = (float)m->timestep; D.gx = (float)m->gravity[0]; D.gy = (float)m->gravity[1]; D.gz = (float)m->gravity[2];
  double mtot = 0; for (int i = 1; i < nb; i++) mtot += m->body_mass[i];
  D.mtot_inv = (float)(1.0 / mtot);
  D.anc_stride = r4(D.max_bdepth > 4 ? D.max_bdepth : 4);
  int njs = 0; for (int j = 0; j < nj; j++) njs += m->jnt_type[j] != FMJ_JNT_FREE;
  D.njs = njs; D.nsensordata = 6 * (nb - 1) + 3 * njs + nu;
  c->layout.nsensordata = D.nsensordata; c->layout.framelinvel_adr = 0; c->layout.jointpos_adr = 6 * (nb - 1);
  c->layout.actuatorfrc_adr = 6 * (nb - 1) + 3 * njs; c->layout.first_link_body = 1;
  c->layout.first_sensor_jnt = D.root_free ? 1 : 0;

  std::vector<float4> b_pos_mass(64), b_quat(64), b_ipos(64), b_iquat(64), b_inertia(64), j_axis_q0(64), j_pos_k(64);
  std::vector<int4> b_info(64), b_info2(64), d_info(64), d_act(64);
  std::vector<float4> d_prm(64);
  std::vector<uint8_t> b_anc((size_t)r4(nb * D.anc_stride), 0);
  c->body_link_row.assign(nb, -1); c->dof_joint_row.assign(nv, -1); c->body_swim.assign(nb, -1);
  c->jnt_dofadr.assign(m->jnt_dofadr, m->jnt_dofadr + nj); c->jnt_type.assign(m->jnt_type, m->jnt_type + nj);
  int any_k = 0;
  for (int i = 0; i < 64; i++) {
    b_pos_mass[i] = f4(0, 0, 0, 0); b_quat[i] = f4(1, 0, 0, 0); b_ipos[i] = f4(0, 0, 0, 0); b_iquat[i] = f4(1, 0, 0, 0);
    b_inertia[i] = f4(0, 0, 0, 0); j_axis_q0[i] = f4(0, 0, 1, 0); j_pos_k[i] = f4(0, 0, 0, 0);
    b_info[i] = make_int4(0, -1, 0, 0); b_info2[i] = make_int4(-1, 0, -1, -1);
    d_info[i] = make_int4(0, 0, 0, -1); d_prm[i] = f4(0, 0, 0, 0); d_act[i] = make_int4(0, 0, 0, 0);
  }
  for (int i = 1; i < nb; i++) {
    b_pos_mass[i] = f4(m->body_pos[3 * i], m->body_pos[3 * i + 1], m->body_pos[3 * i + 2], m->body_mass[i]);
    b_quat[i] = f4(m->body_quat[4 * i], m->body_quat[4 * i + 1], m->body_quat[4 * i + 2], m->body_quat[4 * i + 3]);
    b_ipos[i] = f4(m->body_ipos[3 * i], m->body_ipos[3 * i + 1], m->body_ipos[3 * i + 2], 0);
    b_iquat[i] = f4(m->body_iquat[4 * i], m->body_iquat[4 * i + 1], m->body_iquat[4 * i + 2], m->body_iquat[4 * i + 3]);
    b_inertia[i] = f4(m->body_inertia[3 * i], m->body_inertia[3 * i + 1], m->body_inertia[3 * i + 2], 0);
    int j = m->body_jntadr[i];
    if (j >= 0) {
      double q0 = m->jnt_type[j] == FMJ_JNT_FREE ? 0.0 : m->qpos0[m->jnt_qposadr[j]];
      j_axis_q0[i] = f4(m->jnt_axis[3 * j], m->jnt_axis[3 * j + 1], m->jnt_axis[3 * j + 2], q0);
      j_pos_k[i] = f4(m->jnt_pos[3 * j], m->jnt_pos[3 * j + 1], m->jnt_pos[3 * j + 2], m->jnt_stiffness[j]);
      if (m->jnt_type[j] != FMJ_JNT_FREE && m->jnt_stiffness[j] != 0) any_k = 1;
      b_info[i] = make_int4(m->body_parentid[i], m->jnt_type[j], m->jnt_qposadr[j], m->jnt_dofadr[j]);
    } else b_info[i] = make_int4(m->body_parentid[i], -1, 0, 0);
    c->body_link_row[i] = i - 1;
    b_info2[i] = make_int4(bdepth[i], subsize[i], i - 1, -1);
    {   // pointer-jumping table: ancestor at distance 2^r (0 = world, which holds the identity)
      for (int r = 0; r < D.max_bdepth; r++) {
        int a = i, hops = 1 << r;
        while (hops-- > 0 && a > 0) a = m->body_parentid[a];
        b_anc[(size_t)i * D.anc_stride + r] = (uint8_t)a;
      }
    }
  }
  D.any_stiffness = any_k; D.any_box = any_box;
  D.any_jpos = D.any_bquat = D.any_iquat = 0;
  for (int i = 1; i < nb; i++) {
    const int j = m->body_jntadr[i];
    if (j >= 0 && m->jnt_type[j] != FMJ_JNT_FREE && (m->jnt_pos[3 * j] != 0 || m->jnt_pos[3 * j + 1] != 0 || m->jnt_pos[3 * j + 2] != 0)) D.any_jpos = 1;
    const bool freeb = j >= 0 && m->jnt_type[j] == FMJ_JNT_FREE;
    if (!freeb && !(fabs(m->body_quat[4 * i]) == 1.0 && m->body_quat[4 * i + 1] == 0 && m->body_quat[4 * i + 2] == 0 && m->body_quat[4 * i + 3] == 0)) D.any_bquat = 1;
    if (!(m->body_iquat[4 * i] == 1.0 && m->body_iquat[4 * i + 1] == 0 && m->body_iquat[4 * i + 2] == 0 && m->body_iquat[4 * i + 3] == 0)) D.any_iquat = 1;
  }
  D.n_links = nb - 1; D.n_joints = njs; D.n_xfrc = nb - 1; D.ns = 0;
  // actuators sorted by dof
  std::vector<float4> a_prm, a_lim; std::vector<int> a_src;
  int sj = 0;
  for (int j = 0; j < nj; j++) {
    int d0 = m->jnt_dofadr[j];
    if (m->jnt_type[j] == FMJ_JNT_FREE) {
      for (int k = 0; k < 6; k++) { d_info[d0 + k] = make_int4(m->jnt_bodyid[j], ddepth[d0 + k], dsub[d0 + k], -1);
        d_prm[d0 + k] = f4(m->dof_armature[d0 + k], m->dof_damping[d0 + k], 0, 0); d_act[d0 + k] = make_int4(0, 0, 0, 0); }
      continue;
    }
    d_info[d0] = make_int4(m->jnt_bodyid[j], ddepth[d0], dsub[d0], sj);
    c->dof_joint_row[d0] = sj;
    d_prm[d0] = make_float4((float)m->dof_armature[d0], (float)m->dof_damping[d0], ibits(m->jnt_qposadr[j]), 1.0f);
    int first = (int)a_src.size(), cnt = 0;
    for (int a = 0; a < nu; a++) if (m->actuator_jntid[a] == j) {
      a_prm.push_back(f4(m->actuator_gain[a], m->actuator_bias[3 * a], m->actuator_bias[3 * a + 1], m->actuator_bias[3 * a + 2]));
      double cl = m->actuator_ctrllimited[a] ? m->actuator_ctrlrange[2 * a] : -3.0e38, ch = m->actuator_ctrllimited[a] ? m->actuator_ctrlrange[2 * a + 1] : 3.0e38;
      double fl = m->actuator_forcelimited[a] ? m->actuator_forcerange[2 * a] : -3.0e38, fh = m->actuator_forcelimited[a] ? m->actuator_forcerange[2 * a + 1] : 3.0e38;
      a_lim.push_back(f4(cl, ch, fl, fh)); a_src.push_back(a); cnt++;
    }
    d_act[d0] = make_int4(first, cnt, sj, 0);
    sj++;
  }
  // the caller's sparse layout of M (mjModel dof_Madr / nM) must describe the same tree
  int e = 0;
  for (int i = 0; i < nv; i++) {
    if (m->dof_Madr[i] != e) { fmj_destroy(c); return set_err(FMJ_ERR_ARG, "fmj_create: dof_Madr inconsistent"); }
    for (int j = i; j >= 0; j = m->dof_parentid[j]) e++;
  }
  if (e != m->nM) { fmj_destroy(c); return set_err(FMJ_ERR_ARG, "fmj_create: nM inconsistent"); }
  c->h_b_info2.assign((int*)b_info2.data(), (int*)b_info2.data() + 64 * 4);
  c->h_d_info.assign((int*)d_info.data(), (int*)d_info.data() + 64 * 4);

  {   // subtree mass (a model constant) rides in ipos.w
    std::vector<double> smass(nb, 0.0);
    for (int i = 1; i < nb; i++) smass[i] = m->body_mass[i];
    for (int i = nb - 1; i >= 2; i--) smass[m->body_parentid[i]] += smass[i];
    for (int i = 1; i < nb; i++) b_ipos[i].w = (float)smass[i];
  }
  UP(b_anc, b_anc);
  {   // depth of the deepest dof shared by the root chains of two dofs (constraint rows couple only through it)
    std::vector<int8_t> lcad((size_t)r4(nv * nv), (int8_t)-1);
    for (int a = 0; a < nv; a++)
      for (int b = 0; b < nv; b++) {
        int x = a, y = b;
        while (x >= 0 && y >= 0 && x != y) { if (ddepth[x] > ddepth[y]) x = m->dof_parentid[x]; else if (ddepth[y] > ddepth[x]) y = m->dof_parentid[y]; else { x = m->dof_parentid[x]; y = m->dof_parentid[y]; } }
        lcad[(size_t)a * nv + b] = (x >= 0 && x == y) ? (int8_t)ddepth[x] : (int8_t)-1;
      }
    UP(lcad, lcad);
  }
  // ---- constraint tables
  D.cons = cons; D.ngeom = m->ngeom; D.nplane = nplane; D.nvs = nv | 1;
  D.max_contacts = cons ? (m->max_contacts > 0 ? m->max_contacts : 1) : 0;
  {
    int nlimj = 0; for (int j = 0; j < nj; j++) nlimj += (m->jnt_limited[j] && m->jnt_type[j] != FMJ_JNT_FREE);
    D.maxefc = cons ? nlimj + 4 * D.max_contacts : 0;
    // the HBM constraint path keeps A in rows of AG_LD floats and three 64-row slots per lane (fmj_cons_rows.inc)
    if (D.maxefc > AG_LD) { fmj_destroy(c); return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: more than 192 constraint rows possible (limited joints + 4 * max_contacts): lower max_contacts"); }
    // the block solvers (elliptic PGS, noslip) keep forces, b, R and mu of every row in the LDS staging area of the row code: min(maxefc, 64) * rs floats
    if (cons && (m->noslip_iterations > 0 || (m->cone == FMJ_CONE_ELLIPTIC && (m->solver == FMJ_SOLVER_PGS || dual_instead))) && std::min(D.maxefc, FMJ_NA) * D.rs < 4 * D.maxefc) {
      fmj_destroy(c); return set_err(FMJ_ERR_UNSUPPORTED, "fmj_create: PGS with the elliptic cone / noslip: too many constraint rows for this model's row length (needs min(maxefc, 64) * rs >= 4 * maxefc): lower max_contacts");
    }
  }
  D.implicitfast = m->integrator == FMJ_INT_IMPLICITFAST;
  D.hdamp = m->integrator == FMJ_INT_RK4 ? 0.0f : (float)m->timestep;
  D.solver_iterations = dual_instead ? 10 * m->solver_iterations : m->solver_iterations; D.solver_tolerance = (float)m->solver_tolerance;
  D.cone = cons ? m->cone : FMJ_CONE_PYRAMIDAL;
  D.noslip_iterations = cons ? m->noslip_iterations : 0; D.noslip_tolerance = (float)m->noslip_tolerance;
  D.solver = (cons && !dual_instead) ? m->solver : FMJ_SOLVER_PGS; D.ls_iterations = m->ls_iterations > 0 ? m->ls_iterations : 50;
  D.ls_tolerance = (float)(m->ls_tolerance > 0 ? m->ls_tolerance : 0.01);
  D.impratio_isqrt = (float)(1.0 / sqrt(m->impratio > 0 ? m->impratio : 1.0));
  D.pgs_scale = (float)(1.0 / ((m->meaninertia > 0 ? m->meaninertia : 1.0) * (nv > 1 ? nv : 1)));
  std::vector<int> d_parent(64, -1);
  for (int d = 0; d < nv; d++) d_parent[d] = m->dof_parentid[d];
  std::vector<int4> g_info(m->ngeom ? m->ngeom : 1); std::vector<float4> g_size(g_info.size()), g_pos(g_info.size()), g_quat(g_info.size()), g_sol0(g_info.size()), g_sol1(g_info.size());
  std::vector<float4> p_plane(nplane ? nplane : 1), p_prm(nplane ? nplane : 1), p_hq(nplane ? nplane : 1, f4(1, 0, 0, 0)), p_hs(nplane ? nplane : 1, f4(1, 1, 0, 0));
  std::vector<float4> d_lim(64, f4(0, 0, 0, 0)), d_sol0(64, f4(0.02, 1, 0.9, 0.95)), d_sol1(64, f4(0.001, 0.5, 2, 0));
  if (cons) {
    int ip = 0;
    for (int g = 0; g < m->ngeom; g++) {
      int b = m->geom_bodyid[g];
      int last = -1;
      for (int a = b; a >= 1 && last < 0; a = m->body_parentid[a]) if (m->body_dofnum[a] > 0) last = m->body_dofadr[a] + m->body_dofnum[a] - 1;
      g_info[g] = make_int4(m->geom_type[g], b, last, (m->geom_type[g] == FMJ_GEOM_MESH && m->geom_faceadr && m->nmeshface > 0) ? m->geom_faceadr[g] : 0);
      g_size[g] = f4(m->geom_size[3 * g], m->geom_size[3 * g + 1], m->geom_size[3 * g + 2], m->geom_friction[3 * g]);
      if (m->geom_type[g] == FMJ_GEOM_MESH) {          // first vertex, vertex count, bounding radius about the geom origin
        double r2 = 0;
        for (int i = m->geom_vertadr[g]; i < m->geom_vertadr[g] + m->geom_vertnum[g]; i++) {
          const double* v = m->mesh_vert + 3 * (size_t)i;
          r2 = fmax(r2, v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        }
        g_size[g] = make_float4(ibits(m->geom_vertadr[g]), ibits(m->geom_vertnum[g]), (float)(sqrt(r2) * (1.0 + 1e-6)), (float)m->geom_friction[3 * g]);
      }
      g_pos[g] = f4(m->geom_pos[3 * g], m->geom_pos[3 * g + 1], m->geom_pos[3 * g + 2], m->body_invweight0[2 * b]);
      g_quat[g] = f4(m->geom_quat[4 * g], m->geom_quat[4 * g + 1], m->geom_quat[4 * g + 2], m->geom_quat[4 * g + 3]);
      g_sol0[g] = f4(m->geom_solref[2 * g], m->geom_solref[2 * g + 1], m->geom_solimp[5 * g], m->geom_solimp[5 * g + 1]);
      g_sol1[g] = f4(m->geom_solimp[5 * g + 2], m->geom_solimp[5 * g + 3], m->geom_solimp[5 * g + 4], 0);
      g_sol1[g].w = ibits((m->geom_type[g] == FMJ_GEOM_MESH && m->geom_facenum && m->nmeshface > 0) ? m->geom_facenum[g] : 0);      // hull planes (explicit pairs)
      if (m->geom_type[g] == FMJ_GEOM_PLANE) {
        // world-attached plane: normal = z axis of the geom frame, point = geom_pos
        const double* q = m->geom_quat + 4 * g; const double* p = m->geom_pos + 3 * g;
        double nx = 2 * (q[1] * q[3] + q[0] * q[2]), ny = 2 * (q[2] * q[3] - q[0] * q[1]), nz = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
        double nn = sqrt(nx * nx + ny * ny + nz * nz); nx /= nn; ny /= nn; nz /= nn;
        p_plane[ip] = f4(nx, ny, nz, nx * p[0] + ny * p[1] + nz * p[2]);
        p_prm[ip] = make_float4((float)m->geom_friction[3 * g], 0.f, ibits(g), 0.f);
        ip++;
      }
      if (m->geom_type[g] == FMJ_GEOM_HFIELD) {     // position, flag, frame, extent: see ground_dist
        const double* q = m->geom_quat + 4 * g; const double* p = m->geom_pos + 3 * g;
        p_plane[ip] = f4(p[0], p[1], p[2], 0);
        p_prm[ip] = make_float4((float)m->geom_friction[3 * g], 1.f, ibits(g), 0.f);
        p_hq[ip] = f4(q[0], q[1], q[2], q[3]);
        p_hs[ip] = f4(m->hfield_size[0], m->hfield_size[1], m->hfield_size[2], m->hfield_size[3]);
        ip++;
      }
    }
    for (int j = 0; j < nj; j++) {
      if (m->jnt_type[j] == FMJ_JNT_FREE) continue;
      int d = m->jnt_dofadr[j];
      d_lim[d] = f4(m->jnt_limited[j] ? 1 : 0, m->jnt_range[2 * j], m->jnt_range[2 * j + 1], m->jnt_margin[j]);
      d_sol0[d] = f4(m->jnt_solref[2 * j], m->jnt_solref[2 * j + 1], m->jnt_solimp[5 * j], m->jnt_solimp[5 * j + 1]);
      d_sol1[d] = f4(m->jnt_solimp[5 * j + 2], m->jnt_solimp[5 * j + 3], m->jnt_solimp[5 * j + 4], m->dof_invweight0[d]);
    }
  }
  // ---- pack the tables (one device array per family)
  auto i4f = [](int4 v) { return make_float4(ibits(v.x), ibits(v.y), ibits(v.z), ibits(v.w)); };
  c->h_btab.assign(64 * BT_STRIDE, f4(0, 0, 0, 0)); c->h_dtab.assign(64 * DT_STRIDE, f4(0, 0, 0, 0));
  for (int b = 0; b < 64; b++) {
    float4* t = &c->h_btab[b * BT_STRIDE];
    t[0] = b_pos_mass[b]; t[1] = b_quat[b]; t[2] = b_ipos[b]; t[3] = b_iquat[b]; t[4] = b_inertia[b]; t[5] = j_axis_q0[b]; t[6] = j_pos_k[b];
    t[7] = i4f(b_info[b]); t[8] = i4f(b_info2[b]);
  }
  for (int d = 0; d < 64; d++) {
    float4* t = &c->h_dtab[d * DT_STRIDE];
    int4 act = d_act[d]; act.w = d_parent[d];
    t[0] = i4f(d_info[d]); t[1] = d_prm[d]; t[2] = i4f(act); t[3] = d_lim[d]; t[4] = d_sol0[d]; t[5] = d_sol1[d];
  }
  std::vector<float4> atab((a_src.size() ? a_src.size() : 1) * AT_STRIDE, f4(0, 0, 0, 0));
  for (size_t a = 0; a < a_src.size(); a++) { atab[a * AT_STRIDE] = a_prm[a]; atab[a * AT_STRIDE + 1] = a_lim[a]; atab[a * AT_STRIDE + 2] = make_float4(ibits(a_src[a]), 0.f, 0.f, 0.f); }
  std::vector<float4> gtab(g_info.size() * GT_STRIDE), ptab(p_plane.size() * PT_STRIDE);
  for (size_t g = 0; g < g_info.size(); g++) { float4* t = &gtab[g * GT_STRIDE]; t[0] = i4f(g_info[g]); t[1] = g_size[g]; t[2] = g_pos[g]; t[3] = g_quat[g]; t[4] = g_sol0[g]; t[5] = g_sol1[g]; }
  for (size_t p = 0; p < p_plane.size(); p++) { ptab[p * PT_STRIDE] = p_plane[p]; ptab[p * PT_STRIDE + 1] = p_prm[p]; ptab[p * PT_STRIDE + 2] = p_hq[p]; ptab[p * PT_STRIDE + 3] = p_hs[p]; }
  D.npair = cons ? m->npair : 0; D.nfl = 0; D.qtab = nullptr;
  {
    std::vector<float4> qtab((m->npair ? m->npair : 1) * QT_STRIDE, f4(0, 0, 0, 0));
    for (int p = 0; p < m->npair; p++) {
      const double mu = m->pair_friction[p] > 1e-5 ? m->pair_friction[p] : 1e-5;       // mjMINMU
      qtab[p * QT_STRIDE] = make_float4(ibits(m->pair_geom1[p]), ibits(m->pair_geom2[p]), (float)mu, 0.f);
      qtab[p * QT_STRIDE + 1] = f4(m->pair_solref[2 * p], m->pair_solref[2 * p + 1], m->pair_solimp[5 * p], m->pair_solimp[5 * p + 1]);
      qtab[p * QT_STRIDE + 2] = f4(m->pair_solimp[5 * p + 2], m->pair_solimp[5 * p + 3], m->pair_solimp[5 * p + 4], 0);
    }
    UP(qtab, qtab);
  }
  D.any_polypair = cons ? any_polypair : 0; D.mesh_face = nullptr;
  if (cons && any_polypair) {
    std::vector<float4> mf(m->nmeshface > 0 ? m->nmeshface : 1, f4(0, 0, 1, 0));
    for (int i = 0; i < m->nmeshface; i++) mf[i] = f4(m->mesh_face[4 * i], m->mesh_face[4 * i + 1], m->mesh_face[4 * i + 2], m->mesh_face[4 * i + 3]);
    UP(mf, mesh_face);
  }
  D.any_mesh = cons ? any_mesh : 0; D.mesh_vert = nullptr;
  if (any_mesh) {
    std::vector<float4> mv((size_t)m->nmeshvert);
    for (int i = 0; i < m->nmeshvert; i++) mv[i] = f4(m->mesh_vert[3 * i], m->mesh_vert[3 * i + 1], m->mesh_vert[3 * i + 2], 0);
    UP(mv, mesh_vert);
  }
  D.hf_nrow = D.hf_ncol = 0; D.hf_data = nullptr;
  if (n_hfield) {
    std::vector<float> hf((size_t)m->hfield_nrow * m->hfield_ncol);
    for (size_t i = 0; i < hf.size(); i++) hf[i] = (float)m->hfield_data[i];
    D.hf_nrow = m->hfield_nrow; D.hf_ncol = m->hfield_ncol;
    UP(hf, hf_data);
  }
  c->h_atab = atab; c->a_src = a_src;
  UP(c->h_btab, btab); UP(c->h_dtab, dtab); UP(atab, atab); UP(gtab, gtab); UP(ptab, ptab);
  {   // two envs per wave: bodies and the dofs minus a free root's translational dofs must fit 32 lanes
    const int t0 = D.root_free ? 3 : 0;
    const char* envv = getenv("FMJ_DUAL");
    D.dual_t0 = t0;
    const bool halves_ok = nb <= 32 && nv - t0 <= 32 && !(envv && envv[0] == '0');      // bodies / lane dofs of an env fit half a wave
    const bool rk4 = m->integrator == FMJ_INT_RK4;      // four forward launches of the one-env kernel per step: no two-env kernel, no fused launch
    D.dual_ok = !cons && halves_ok && !rk4;
    // the two-env constraint kernel covers what BASELINE configs[3] needs: limits + ground contacts of sphere / capsule / box / cylinder
    // geoms on ONE ground geom, pyramidal cone, PGS; everything else (pairs, meshes, Newton / CG, the elliptic cone) keeps the one-env kernel
    D.cons2_ok = cons && halves_ok && !rk4 && D.rs <= FMJ_MAXD && m->solver == FMJ_SOLVER_PGS && !dual_instead && m->cone == FMJ_CONE_PYRAMIDAL && m->noslip_iterations == 0 && m->npair == 0 &&
                 !any_mesh && nplane <= 1 && m->ngeom <= 32;
    for (int t = 0; t < 3; t++) D.dual_tadd[t] = t < t0 ? (float)(mtot + m->dof_armature[t] + m->timestep * m->dof_damping[t]) : 1.0f;
    for (int t = 0; t < 3; t++) D.dual_taddm[t] = t < t0 ? (float)(mtot + m->dof_armature[t]) : 1.0f;
    {   // elimination rounds of the one-env kernel (lane = dof): dofs grouped by depth, deepest first, <= 3 per round
      std::vector<DualRound> rounds;
      std::vector<unsigned long long> ancm(nv, 0ull), descm(nv, 0ull);
      int maxdep = 0;
      for (int i = 0; i < nv; i++) {
        if (ddepth[i] > maxdep) maxdep = ddepth[i];
        for (int a = m->dof_parentid[i]; a >= 0; a = m->dof_parentid[a]) { ancm[i] |= 1ull << a; descm[a] |= 1ull << i; }
      }
      for (int dep = maxdep; dep >= 0; dep--) {
        std::vector<int> lvl;
        for (int i = 0; i < nv; i++) if (ddepth[i] == dep) lvl.push_back(i);
        for (size_t q = 0; q < lvl.size(); q += 3) {
          DualRound R; memset(&R, 0, sizeof R);
          int* pp[3] = {&R.p0, &R.p1, &R.p2};
          R.depth = dep;
          for (int c = 0; c < 3; c++) {
            if (q + c < lvl.size()) { const int pv = lvl[q + c]; *pp[c] = pv; R.anc[c] = ancm[pv]; }
            else *pp[c] = c == 1 ? -1 : R.p0;
          }
          rounds.push_back(R);
        }
      }
      D.nround1 = (int)rounds.size();
      UP(rounds, rounds1);
      std::vector<WideRound> wide;                  // the same levels, up to six dofs per round
      for (int dep = maxdep; dep >= 0; dep--) {
        std::vector<int> lvl;
        for (int i = 0; i < nv; i++) if (ddepth[i] == dep) lvl.push_back(i);
        for (size_t q = 0; q < lvl.size(); q += 6) {
          WideRound R; memset(&R, 0, sizeof R);
          R.depth = dep; R.np = (int)std::min<size_t>(6, lvl.size() - q);
          for (int c = 0; c < 6; c++) {
            if (q + c < lvl.size()) { R.p[c] = lvl[q + c]; R.anc[c] = ancm[lvl[q + c]]; }
            else R.p[c] = lvl[q];                   // under an empty mask
          }
          wide.push_back(R);
        }
      }
      D.nround6 = (int)wide.size();
      UP(wide, rounds6);
      {   // per dof, per depth: byte = 4 * lane of the ancestor at that depth (own lane where there is none)
        std::vector<uint32_t> ancl((size_t)64 * (D.rs / 4), 0u);
        for (int i = 0; i < 64; i++) {
          uint8_t* row = (uint8_t*)&ancl[(size_t)i * (D.rs / 4)];
          for (int l = 0; l < D.rs; l++) row[l] = (uint8_t)(4 * i);
          if (i < nv) for (int a = m->dof_parentid[i]; a >= 0; a = m->dof_parentid[a]) row[ddepth[a]] = (uint8_t)(4 * a);
        }
        D.maxdep1 = maxdep;
        UP(ancl, ancl1);
      }
    }
    {   // elimination rounds: lane dofs grouped by depth, deepest first, at most three per round
      const int nd = nv - t0 > 0 ? nv - t0 : 0;
      std::vector<DualRound> rounds;
      if (D.dual_ok || D.cons2_ok) {
        std::vector<unsigned long long> ancm(nd, 0ull), descm(nd, 0ull);
        int maxdep = 0;
        for (int i = 0; i < nd; i++) {
          if (ddepth[i + t0] > maxdep) maxdep = ddepth[i + t0];
          for (int a = m->dof_parentid[i + t0]; a >= t0; a = m->dof_parentid[a]) {
            ancm[i] |= (1ull << (a - t0)) | (1ull << (a - t0 + 32));
            descm[a - t0] |= (1ull << i) | (1ull << (i + 32));
          }
        }
        for (int dep = maxdep; dep >= 0; dep--) {
          std::vector<int> lvl;
          for (int i = 0; i < nd; i++) if (ddepth[i + t0] == dep) lvl.push_back(i);
          for (size_t q = 0; q < lvl.size(); q += 3) {
            DualRound R; memset(&R, 0, sizeof R);
            int* pp[3] = {&R.p0, &R.p1, &R.p2};
            R.depth = dep;
            for (int c = 0; c < 3; c++) {
              if (q + c < lvl.size()) { const int pv = lvl[q + c]; *pp[c] = pv; R.anc[c] = ancm[pv]; }
              else *pp[c] = c == 1 ? -1 : R.p0;
            }
            for (int c = 0; c < 3; c++) R.ro[c] = *pp[c] < 0 ? -1 : *pp[c] * dual_row_stride(D.rs) * 4;
            rounds.push_back(R);
          }
        }
      }
      D.dual_nround = (int)rounds.size();
      { DualRound R; memset(&R, 0, sizeof R); R.p1 = -1; R.ro[1] = -1; R.depth = -1; rounds.push_back(R); }     // terminator: the two-env kernel's loops stop at it and read it as 'the round after the last'
      UP(rounds, dual_rounds);
      {   // per lane dof, per absolute depth: byte = 4 * lane of the ancestor at that depth (the solve pulls x from there
          // with ds_bpermute); own lane where there is none.  The kernel adds the half's offset.
        std::vector<uint32_t> ancl((size_t)32 * (D.rs / 4), 0u);
        int md = 0;
        for (int i = 0; i < 32; i++) {
          uint8_t* row = (uint8_t*)&ancl[(size_t)i * (D.rs / 4)];
          for (int l = 0; l < D.rs; l++) row[l] = (uint8_t)(4 * i);
          if ((D.dual_ok || D.cons2_ok) && i < nd) {
            if (ddepth[i + t0] > md) md = ddepth[i + t0];
            for (int a = m->dof_parentid[i + t0]; a >= t0; a = m->dof_parentid[a]) row[ddepth[a]] = (uint8_t)(4 * (a - t0));
          }
        }
        D.dual_maxdep = md;
        UP(ancl, dual_ancl);
      }
    }
  }
  c->d_btab = (float4*)D.btab; c->d_dtab = (float4*)D.dtab;
  { std::vector<float4> empty4(ST_STRIDE, f4(0, 0, 0, 0)); UP(empty4, stab); }
  c->ngeom = m->ngeom; c->geom_sensor.assign(m->ngeom ? m->ngeom : 1, -1); c->n_contact_rows = 0; c->d_geom_sensor = nullptr; c->d_pairs = nullptr; c->n_pairs = 0;
  c->geom_is_plane.assign(m->ngeom ? m->ngeom : 1, 0);
  for (int g = 0; g < m->ngeom; g++) c->geom_is_plane[g] = m->geom_type[g] == FMJ_GEOM_PLANE || m->geom_type[g] == FMJ_GEOM_HFIELD;
  D.cons_rows = nullptr; D.cons_a = nullptr; D.cons_z = nullptr; D.cons_zf = nullptr;
  if (D.cons && D.maxefc > 0) {   // HBM scratch of envs whose constraint rows outgrow LDS (fmj_step_kernel<.., CONS = true>, "big" path)
    void* p1 = nullptr; void* p2 = nullptr;
    void* p3 = nullptr;
    if (hipMalloc(&p1, (size_t)n_envs * D.maxefc * 8 * sizeof(float)) != hipSuccess ||
        hipMalloc(&p2, (size_t)n_envs * D.maxefc * AG_LD * sizeof(float)) != hipSuccess ||
        hipMalloc(&p3, (size_t)n_envs * D.maxefc * D.rs * sizeof(float)) != hipSuccess) {
      if (p1) (void)hipFree(p1);
      if (p2) (void)hipFree(p2);
      fmj_destroy(c);
      return set_err(FMJ_ERR_HIP, "fmj_create: out of device memory for the constraint scratch");
    }
    c->allocs.push_back(p1); c->allocs.push_back(p2); c->allocs.push_back(p3);
    D.cons_rows = (float*)p1; D.cons_a = (float*)p2; D.cons_z = (float*)p3;
    if (D.npair > 0) {
      void* p4 = nullptr;
      if (hipMalloc(&p4, (size_t)n_envs * D.maxefc * D.rs * sizeof(float)) != hipSuccess) { fmj_destroy(c); return set_err(FMJ_ERR_HIP, "fmj_create: out of device memory for the constraint scratch"); }
      c->allocs.push_back(p4);
      D.cons_zf = (float*)p4;
    }
  }
  LdsLayout L = lds_layout(nb, nv, nq, D.rs, D.anc_stride, D.cons, D.maxefc, D.max_contacts, D.nvs, D.npair);
  D.nfl = L.nfl;
  c->lds_bytes = (size_t)L.total * sizeof(float);
  c->lds_bytes_dual2 = D.dual_ok ? (size_t)(2 * lds2_layout(nb, nv, nq, D.rs, D.dual_t0).total + r4(nb * D.anc_stride) / 4) * sizeof(float) : 0;
  c->lds_bytes_cons2 = 0; c->d_resume = nullptr;
  if (D.cons2_ok) {
    c->lds_bytes_cons2 = (size_t)lds3_layout(nb, nv, nq, D.rs, D.dual_t0, D.max_contacts, D.anc_stride).total * sizeof(float);
    void* pr = nullptr;
    if (c->lds_bytes_cons2 > 64 * 1024 || hipMalloc(&pr, (size_t)n_envs * sizeof(int)) != hipSuccess) D.cons2_ok = 0;      // (cannot happen within the size limits above)
    else { c->allocs.push_back(pr); c->d_resume = (int*)pr; (void)hipMemset(pr, 0, (size_t)n_envs * sizeof(int)); }
  }
  c->rk4 = m->integrator == FMJ_INT_RK4; c->rk_q0 = c->rk_v0 = c->rk_sv = c->rk_sa = c->rk_sd = nullptr;
  if (c->rk4) {
    const size_t per_env = (size_t)nq + 3 * (size_t)nv + (size_t)(D.nsensordata > 0 ? D.nsensordata : 1);
    void* pk = nullptr;
    if (hipMalloc(&pk, per_env * (size_t)n_envs * sizeof(float)) != hipSuccess) { fmj_destroy(c); return set_err(FMJ_ERR_HIP, "fmj_create: RK4 stage buffers"); }
    c->allocs.push_back(pk);
    c->rk_q0 = (float*)pk; c->rk_v0 = c->rk_q0 + (size_t)n_envs * nq; c->rk_sv = c->rk_v0 + (size_t)n_envs * nv; c->rk_sa = c->rk_sv + (size_t)n_envs * nv;
    c->rk_sd = c->rk_sa + (size_t)n_envs * nv;
  }
  {
    hipDeviceProp_t prop;
    int n_cu = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    const int waves = (n_envs + 1) / 2;
    // the build registered for the waves per SIMD the batch can fill: 2 (256 registers: model constants resident), 3 (168) or 4 (128)
    c->dual_wps = waves <= 2 * 4 * n_cu ? 2 : (waves <= 3 * 4 * n_cu ? 3 : 4);
    const char* w = getenv("FMJ_WPS");
    if (w && (w[0] == '2' || w[0] == '3' || w[0] == '4')) c->dual_wps = w[0] - '0';
  }
  if (c->lds_bytes > 64 * 1024) {
    hipError_t e1 = hipFuncSetAttribute((const void*)pick_kernel(c, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
    hipError_t e2 = hipFuncSetAttribute((const void*)pick_kernel(c, false), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes);
    if (e1 != hipSuccess || e2 != hipSuccess) { fmj_destroy(c); return set_err(FMJ_ERR_HIP, "fmj_create: LDS request too large"); }
  }
  *out = c;
  return FMJ_OK;
}

int fmj_get_sensor_layout(const fmj_ctx* c, fmj_sensor_layout_t* out) {
  if (!c || !out) return set_err(FMJ_ERR_ARG, "fmj_get_sensor_layout: NULL");
  *out = c->layout; return FMJ_OK;
}

int fmj_kernel_info(const fmj_ctx* c, int32_t* lds_bytes_per_env, int32_t* threads_per_env) {
  if (!c) return set_err(FMJ_ERR_ARG, "fmj_kernel_info: NULL ctx");
  if (lds_bytes_per_env) *lds_bytes_per_env = (int32_t)(c->dm.dual_ok ? c->lds_bytes_dual2 / 2 : (c->dm.cons2_ok ? c->lds_bytes_cons2 / 2 : c->lds_bytes));
  if (threads_per_env) *threads_per_env = (c->dm.dual_ok || c->dm.cons2_ok) ? 32 : 64;   // the integrating step packs two envs per wave when it can
  return FMJ_OK;
}

int fmj_set_swimming(fmj_ctx* c, int32_t ns, int32_t n_xfrc_rows, const int32_t* links_index, const int32_t* xfrc_index,
                     const int32_t* body_index, const double* coefficients, const double* masses,
                     const double* heights, const double* densities) {
  if (!c || ns < 0 || n_xfrc_rows < 0 || (ns > 0 && (!links_index || !xfrc_index || !body_index || !coefficients || !masses || !heights || !densities)))
    return set_err(FMJ_ERR_ARG, "fmj_set_swimming: NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  std::vector<float4> c0(ns ? ns : 1), c1(ns ? ns : 1), c2(ns ? ns : 1);
  for (int b = 0; b < 64; b++) c->h_b_info2[4 * b + 3] = -1;
  for (int s = 0; s < ns; s++) {
    int b = body_index[s];
    if (b < 1 || b >= c->nbody) return set_err(FMJ_ERR_ARG, "fmj_set_swimming: body index out of range");
    if (links_index[s] < 0 || links_index[s] >= c->dm.n_links || xfrc_index[s] < 0 || xfrc_index[s] >= n_xfrc_rows)
      return set_err(FMJ_ERR_ARG, "fmj_set_swimming: row index out of range");
    if (c->h_b_info2[4 * b + 2] != links_index[s]) return set_err(FMJ_ERR_ARG, "fmj_set_swimming: links_index must be the readout row of the same body (call fmj_set_readout_maps first)");
    const double* k = coefficients + 6 * s;
    c0[s] = f4(k[0], k[1], k[2], masses[s]); c1[s] = f4(k[3], k[4], k[5], heights[s]);
    c2[s] = make_float4((float)densities[s], ibits(links_index[s]), ibits(xfrc_index[s]), ibits(b));
    c->h_b_info2[4 * b + 3] = s;
  }
  std::vector<float4> stab((ns ? ns : 1) * ST_STRIDE, f4(0, 0, 0, 0));
  for (int s2 = 0; s2 < ns; s2++) {
    stab[s2 * ST_STRIDE] = c0[s2]; stab[s2 * ST_STRIDE + 1] = c1[s2]; stab[s2 * ST_STRIDE + 2] = c2[s2];
    stab[s2 * ST_STRIDE + 3] = f4(densities[s2] != 0 ? masses[s2] / densities[s2] : 0.0, heights[s2] != 0 ? 1.0 / heights[s2] : 0.0, 0, 0);
  }
  int rc;
  if ((rc = upload(c, stab, &c->dm.stab))) return rc;
  if ((rc = sync_tables(c))) return rc;
  c->dm.ns = ns;
  c->dm.n_xfrc = n_xfrc_rows;       // env stride of the xfrc rows = rows the caller's tensor holds per env
  return FMJ_OK;
}

int fmj_set_actuator_forcerange(fmj_ctx* c, int32_t nu, const int32_t* forcelimited, const double* forcerange) {
  if (!c || !forcelimited || !forcerange) return set_err(FMJ_ERR_ARG, "fmj_set_actuator_forcerange: NULL argument");
  if (nu != c->nu) return set_err(FMJ_ERR_ARG, "fmj_set_actuator_forcerange: nu does not match the model");
  HIP_TRY(hipSetDevice(c->device));
  for (size_t a = 0; a < c->a_src.size(); a++) {      // table order = sorted by dof; a_src names the model's actuator
    const int src = c->a_src[a];
    float4& lim = c->h_atab[a * AT_STRIDE + 1];
    lim.z = forcelimited[src] ? (float)forcerange[2 * src] : -3.0e38f;
    lim.w = forcelimited[src] ? (float)forcerange[2 * src + 1] : 3.0e38f;
  }
  // the table is live: a step kernel of this context may still be running on a non-blocking stream, which the null-stream
  // copy below would not wait for.  Drain the device first (this is a set-up call, task.py:253-286 runs it once).
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy((void*)c->dm.atab, c->h_atab.data(), c->h_atab.size() * sizeof(float4), hipMemcpyHostToDevice));
  return FMJ_OK;
}

int fmj_set_readout_maps(fmj_ctx* c, int32_t n_links, const int32_t* links_body, int32_t n_joints, const int32_t* joints_jnt) {
  if (!c || n_links < 0 || n_joints < 0 || (n_links && !links_body) || (n_joints && !joints_jnt)) return set_err(FMJ_ERR_ARG, "fmj_set_readout_maps: NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  for (int b = 0; b < 64; b++) c->h_b_info2[4 * b + 2] = -1;
  for (int d = 0; d < 64; d++) c->h_d_info[4 * d + 3] = -1;
  for (int i = 0; i < n_links; i++) {
    int b = links_body[i];
    if (b < 1 || b >= c->nbody) return set_err(FMJ_ERR_ARG, "fmj_set_readout_maps: link body out of range");
    c->h_b_info2[4 * b + 2] = i;
  }
  for (int i = 0; i < n_joints; i++) {
    int j = joints_jnt[i];
    if (j < 0 || j >= c->njnt || c->jnt_type[j] == FMJ_JNT_FREE) return set_err(FMJ_ERR_ARG, "fmj_set_readout_maps: joint rows must be hinge/slide joints");
    c->h_d_info[4 * c->jnt_dofadr[j] + 3] = i;
  }
  { int rc2 = sync_tables(c); if (rc2) return rc2; }
  c->dm.n_links = n_links; c->dm.n_joints = n_joints;
  { int rc3 = sync_readout_maps(c); if (rc3) return rc3; }
  return FMJ_OK;
}

static int fill_data(const fmj_ctx* c, const fmj_data* d, StepArgs* A, bool need_state) {
  memset(A, 0, sizeof *A);
  if (!d) return set_err(FMJ_ERR_ARG, "NULL fmj_data");
  if (need_state && (!d->qpos || !d->qvel || !d->xpos || !d->xquat || !d->xipos || !d->sensordata || !d->status))
    return set_err(FMJ_ERR_ARG, "fmj_data: qpos, qvel, xpos, xquat, xipos, sensordata, status are required");
  A->qpos = d->qpos; A->qvel = d->qvel; A->ctrl = d->ctrl; A->qpos_spring = d->qpos_spring; A->xfrc_applied = d->xfrc_applied;
  A->xpos = d->xpos; A->xquat = d->xquat; A->xipos = d->xipos; A->sensordata = d->sensordata; A->qacc = d->qacc;
  A->time = d->time; A->status = d->status; A->n_envs = c->n_envs;
  A->qacc_warmstart = d->qacc_warmstart; A->contact = d->contact; A->ncon = d->ncon;
  A->inv_meters = A->inv_velocity = A->inv_angvel = A->inv_torques = A->newtons = A->torques = 1.0f;
  A->buffer_size = 1; A->substeps = 1;
  return FMJ_OK;
}

static void fill_units(StepArgs* A, const fmj_units* u) {
  A->inv_meters = 1.0f / u->meters; A->inv_velocity = 1.0f / u->velocity; A->inv_angvel = 1.0f / u->angular_velocity;
  A->inv_torques = 1.0f / u->torques; A->newtons = u->newtons; A->torques = u->torques; A->inv_newtons = 1.0f / u->newtons;
}
static void fill_water(StepArgs* A, const fmj_water* w) {
  A->surface = w->surface; A->viscosity = w->viscosity; A->wvx = w->velocity[0]; A->wvy = w->velocity[1]; A->wvz = w->velocity[2];
  A->wgravity = w->gravity; A->use_buoyancy = w->use_buoyancy;
}

int fmj_step(fmj_ctx* c, const fmj_data* d, int32_t n_steps, int64_t ctrl_step_stride, void* stream) {
  if (!c) return set_err(FMJ_ERR_ARG, "fmj_step: NULL ctx");
  if (n_steps < 0) return set_err(FMJ_ERR_ARG, "fmj_step: n_steps < 0");
  StepArgs A; int rc = fill_data(c, d, &A, true); if (rc) return rc;
  if (c->dm.any_stiffness && !d->qpos_spring) return set_err(FMJ_ERR_ARG, "fmj_step: qpos_spring required (model has joint stiffness)");
  A.n_steps = n_steps; A.ctrl_step_stride = ctrl_step_stride; A.integrate = 1;
  HIP_TRY(hipSetDevice(c->device));
  if (c->rk4) {
    // mj_step with mjINT_RK4 (oracle rk4): per step four forward launches (mj_forward, then three mj_forwardSkip(skipsensor): their
    // sensordata goes to a scratch array, so the caller's keeps the first pass's - poses, contacts and contact forces are the last pass's,
    // as MuJoCo leaves them) with the state update of fmj_rk4_stage_kernel after each.  ctrl and xfrc_applied are held over a step.
    if (!A.qacc) return set_err(FMJ_ERR_ARG, "fmj_step: the RK4 integrator needs fmj_data.qacc (the stages read the accelerations there)");
    for (int st = 0; st < n_steps; st++) {
      StepArgs F = A;
      F.n_steps = 1; F.integrate = 0; F.ctrl_step_stride = 0;
      if (A.ctrl && ctrl_step_stride) F.ctrl = A.ctrl + (size_t)st * (size_t)ctrl_step_stride;
      for (int sg = 0; sg < 4; sg++) {
        StepArgs G = F;
        if (sg > 0) G.sensordata = c->rk_sd;
        int rc2 = launch_step(c, false, G, stream);
        if (rc2) return rc2;
        RkArgs R; R.q0 = c->rk_q0; R.v0 = c->rk_v0; R.sv = c->rk_sv; R.sa = c->rk_sa; R.stage = sg;
        hipLaunchKernelGGL(fmj_rk4_stage_kernel, dim3(c->n_envs), dim3(64), 0, (hipStream_t)stream, c->dm, F, R);
        HIP_TRY(hipGetLastError());
      }
    }
    return FMJ_OK;
  }
  return launch_step(c, false, A, stream);
}

int fmj_forward(fmj_ctx* c, const fmj_data* d, int32_t disable_actuation, void* stream) {
  if (!c) return set_err(FMJ_ERR_ARG, "fmj_forward: NULL ctx");
  StepArgs A; int rc = fill_data(c, d, &A, true); if (rc) return rc;
  if (c->dm.any_stiffness && !d->qpos_spring) return set_err(FMJ_ERR_ARG, "fmj_forward: qpos_spring required (model has joint stiffness)");
  A.n_steps = 1; A.integrate = 0; A.disable_actuation = disable_actuation;
  HIP_TRY(hipSetDevice(c->device));
  return launch_step(c, false, A, stream);
}

int fmj_forward_debug(fmj_ctx* c, const fmj_data* d, int32_t disable_actuation, float* H_rows, int32_t* row_stride,
                      float* qfrc_smooth, void* stream) {
  if (!c || !H_rows || !qfrc_smooth) return set_err(FMJ_ERR_ARG, "fmj_forward_debug: NULL argument");
  StepArgs A; int rc = fill_data(c, d, &A, true); if (rc) return rc;
  if (c->dm.any_stiffness && !d->qpos_spring) return set_err(FMJ_ERR_ARG, "fmj_forward_debug: qpos_spring required (model has joint stiffness)");
  A.n_steps = 1; A.integrate = 0; A.disable_actuation = disable_actuation; A.dbg_H = H_rows; A.dbg_qfrc = qfrc_smooth;
  if (row_stride) *row_stride = c->dm.rs;
  HIP_TRY(hipSetDevice(c->device));
  return launch_step(c, false, A, stream);
}

int fmj_constraint_info(const fmj_ctx* c, int32_t* maxefc, int32_t* max_contacts, int32_t* solver_iterations) {
  if (!c) return set_err(FMJ_ERR_ARG, "fmj_constraint_info: NULL ctx");
  if (maxefc) *maxefc = c->dm.maxefc;
  if (max_contacts) *max_contacts = c->dm.max_contacts;
  if (solver_iterations) *solver_iterations = c->dm.solver_iterations;
  return FMJ_OK;
}

int fmj_solver_info(const fmj_ctx* c, int32_t* requested, int32_t* effective, int32_t* iterations) {
  if (!c) return set_err(FMJ_ERR_ARG, "fmj_solver_info: NULL ctx");
  if (requested) *requested = c->solver_requested;
  if (effective) *effective = c->dm.solver;
  if (iterations) *iterations = c->dm.solver_iterations;
  return FMJ_OK;
}

int fmj_step_debug(fmj_ctx* c, const fmj_data* d, float* efc_rows, float* pgs_improvement, void* stream) {
  if (!c || !efc_rows) return set_err(FMJ_ERR_ARG, "fmj_step_debug: NULL argument");
  if (!c->dm.cons) return set_err(FMJ_ERR_ARG, "fmj_step_debug: the model has no constraints");
  StepArgs A; int rc = fill_data(c, d, &A, true); if (rc) return rc;
  if (c->dm.any_stiffness && !d->qpos_spring) return set_err(FMJ_ERR_ARG, "fmj_step_debug: qpos_spring required (model has joint stiffness)");
  A.n_steps = 1; A.integrate = 1; A.dbg_efc = efc_rows; A.dbg_pgs = pgs_improvement;
  HIP_TRY(hipSetDevice(c->device));
  return launch_step(c, false, A, stream);
}

int fmj_step_fused(fmj_ctx* c, const fmj_data* d, const fmj_fused_args* a, void* stream) {
  if (!c || !a) return set_err(FMJ_ERR_ARG, "fmj_step_fused: NULL argument");
  StepArgs A; int rc = fill_data(c, d, &A, true); if (rc) return rc;
  if (a->n_steps < 0 || a->buffer_size < 1) return set_err(FMJ_ERR_ARG, "fmj_step_fused: bad n_steps/buffer_size");
  if (c->rk4) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_step_fused: the RK4 integrator steps through fmj_step (four forward launches per step); write the rows with fmj_before_step");
  if (c->dm.any_stiffness && !d->qpos_spring) return set_err(FMJ_ERR_ARG, "fmj_step_fused: qpos_spring required");
  if (a->do_readout && (!a->rows_base.links || !a->rows_base.joints)) return set_err(FMJ_ERR_ARG, "fmj_step_fused: readout needs links and joints rows");
  if (a->do_drag && (!a->rows_base.xfrc || c->dm.ns == 0)) return set_err(FMJ_ERR_ARG, "fmj_step_fused: drag needs xfrc rows and fmj_set_swimming");
  if (a->controller == 1 && (!a->wave.amplitude || !a->wave.phase_lag || !a->wave.env_phase)) return set_err(FMJ_ERR_ARG, "fmj_step_fused: wave controller arrays missing");
  if (a->controller != 0 && a->controller != 1) return set_err(FMJ_ERR_ARG, "fmj_step_fused: unknown controller");
  A.integrate = 1;
  A.substeps = a->substeps > 1 ? a->substeps : 1; A.sub_links = (A.substeps > 1 && a->substep_links) ? 1 : 0; A.n_it_total = a->n_iterations > 0 ? a->n_iterations : 0;
  if ((long long)a->n_steps * A.substeps > 0x7fffffffLL) return set_err(FMJ_ERR_ARG, "fmj_step_fused: n_steps * substeps overflows");
  A.n_steps = a->n_steps * A.substeps; A.iteration0 = a->iteration0; A.buffer_size = a->buffer_size; A.do_readout = a->do_readout;
  A.do_drag = a->do_drag; A.controller = a->controller; A.ctrl_step_stride = a->ctrl_step_stride;
  A.row_stride_links = a->row_stride_links; A.row_stride_joints = a->row_stride_joints; A.row_stride_xfrc = a->row_stride_xfrc;
  A.links = a->rows_base.links; A.joints = a->rows_base.joints; A.xfrc = a->rows_base.xfrc;
  if (a->rows_base.contacts && a->do_readout) {
    if (!c->dm.cons || !c->d_geom_sensor) return set_err(FMJ_ERR_ARG, "fmj_step_fused: contact rows need a model with collision geoms and fmj_set_contact_maps");
    A.contacts_rows = a->rows_base.contacts; A.row_stride_contacts = a->row_stride_contacts;
    A.n_contact_rows = c->n_contact_rows; A.n_pairs = c->n_pairs; A.geom_sensor = c->d_geom_sensor; A.pairs = c->d_pairs;
  }
  fill_units(&A, &a->units); fill_water(&A, &a->water);
  A.w_amp = a->wave.amplitude; A.w_lag = a->wave.phase_lag; A.w_env = a->wave.env_phase; A.w_freq = a->wave.frequency;
  A.ctrl_out = a->controller == 1 ? a->ctrl_out : nullptr;
  if (a->rows_ahead) {
    if (!c->dm.dual_ok || A.substeps != 1) return set_err(FMJ_ERR_UNSUPPORTED, "fmj_step_fused: rows_ahead needs the two-env unconstrained step kernel and substeps = 1 (write the rows with fmj_before_step instead)");
    A.rows_ahead = 1; A.xfrc_applied_out = (float*)d->xfrc_applied;
  }
  A.env_order = c->dm.dual_ok ? nullptr : a->env_order;      // the two-env constraint kernel pairs neighbours of the order: similar row counts
  HIP_TRY(hipSetDevice(c->device));
  return launch_step(c, true, A, stream);
}

int fmj_drag(fmj_ctx* c, const fmj_rows* rows, const fmj_water* water, const fmj_units* units, float* xfrc_applied, void* stream) {
  if (!c || !rows || !water || !units || !rows->links || !rows->xfrc) return set_err(FMJ_ERR_ARG, "fmj_drag: NULL argument");
  if (c->dm.ns == 0) return set_err(FMJ_ERR_ARG, "fmj_drag: call fmj_set_swimming first");
  StepArgs A; memset(&A, 0, sizeof A);
  A.n_envs = c->n_envs; A.links = rows->links; A.xfrc = rows->xfrc; A.xfrc_applied_out = xfrc_applied;
  fill_units(&A, units); fill_water(&A, water);
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(fmj_drag_kernel, dim3(c->n_envs), dim3(64), 0, (hipStream_t)stream, c->dm, A);
  HIP_TRY(hipGetLastError());
  return FMJ_OK;
}

int fmj_drag_link(int32_t n_envs, int32_t device, const float* links_row, int64_t links_env_stride, float* xfrc_row,
                  int64_t xfrc_env_stride, const double* coefficients, double mass, double height, double density,
                  const fmj_water* water, int32_t* hydro, void* stream) {
  if (n_envs <= 0 || !links_row || !xfrc_row || !coefficients || !water) return set_err(FMJ_ERR_ARG, "fmj_drag_link: NULL argument or n_envs <= 0");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return set_err(FMJ_ERR_NODEVICE, "fmj_drag_link: no HIP device visible");
  if (device < 0 || device >= ndev) return set_err(FMJ_ERR_ARG, "fmj_drag_link: bad device ordinal");
  HIP_TRY(hipSetDevice(device));
  StepArgs A; memset(&A, 0, sizeof A);
  fill_water(&A, water);
  const float4 c0 = f4(coefficients[0], coefficients[1], coefficients[2], mass), c1 = f4(coefficients[3], coefficients[4], coefficients[5], height);
  hipLaunchKernelGGL(fmj_drag_link_kernel, dim3((n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int)n_envs, links_row,
                     (long long)links_env_stride, xfrc_row, (long long)xfrc_env_stride, c0, c1, (float)density, A, (int*)hydro);
  HIP_TRY(hipGetLastError());
  return FMJ_OK;
}

int fmj_physics2data(fmj_ctx* c, const fmj_data* d, const fmj_rows* rows, const fmj_units* units, int32_t links_only, void* stream) {
  if (!c || !d || !rows || !units || !rows->links || (!links_only && !rows->joints)) return set_err(FMJ_ERR_ARG, "fmj_physics2data: NULL argument");
  StepArgs A; int rc = fill_data(c, d, &A, true); if (rc) return rc;
  A.links = rows->links; A.joints = rows->joints; fill_units(&A, units);
  HIP_TRY(hipSetDevice(c->device));
  if (!c->d_links_body || !c->d_joints_dof) { int rc2 = sync_readout_maps(c); if (rc2) return rc2; }
  hipLaunchKernelGGL(fmj_physics2data_kernel, dim3(c->n_envs), dim3(64), 0, (hipStream_t)stream, c->dm, A, (int)links_only, (const int*)c->d_links_body, (const int*)c->d_joints_dof);
  HIP_TRY(hipGetLastError());
  return FMJ_OK;
}

int fmj_before_step(fmj_ctx* c, const fmj_data* d, const fmj_rows* rows, const fmj_water* water, const fmj_units* units, int32_t flags,
                    float* xfrc_applied, void* stream) {
  if (!c || !d || !rows || !units) return set_err(FMJ_ERR_ARG, "fmj_before_step: NULL argument");
  const bool want_rows = flags & FMJ_BEFORE_ROWS, links_only = flags & FMJ_BEFORE_LINKS_ONLY, want_con = (flags & FMJ_BEFORE_CONTACTS) && !links_only, want_drag = flags & FMJ_BEFORE_DRAG;
  if (want_rows && (!rows->links || (!links_only && !rows->joints))) return set_err(FMJ_ERR_ARG, "fmj_before_step: links / joints rows missing");
  if (want_con && (!rows->contacts || !d->contact || !d->ncon || !c->d_geom_sensor)) return set_err(FMJ_ERR_ARG, "fmj_before_step: contact rows need rows->contacts, the contact list and fmj_set_contact_maps");
  if (want_drag && (!water || !rows->links || !rows->xfrc || c->dm.ns == 0)) return set_err(FMJ_ERR_ARG, "fmj_before_step: drag needs water, links and xfrc rows and fmj_set_swimming");
  StepArgs A; int rc = fill_data(c, d, &A, true); if (rc) return rc;
  A.links = rows->links; A.joints = rows->joints; A.xfrc = rows->xfrc; A.contacts_rows = rows->contacts; A.xfrc_applied_out = xfrc_applied;
  fill_units(&A, units);
  if (water) fill_water(&A, water);
  HIP_TRY(hipSetDevice(c->device));
  if (!c->d_links_body || !c->d_joints_dof) { int rc2 = sync_readout_maps(c); if (rc2) return rc2; }
  hipLaunchKernelGGL(fmj_before_step_kernel, dim3(c->n_envs), dim3(64), 0, (hipStream_t)stream, c->dm, A, (int)flags, (const int*)c->d_links_body,
                     (const int*)c->d_joints_dof, c->n_contact_rows, (const int*)c->d_geom_sensor, c->n_pairs, (const int*)c->d_pairs);
  HIP_TRY(hipGetLastError());
  return FMJ_OK;
}

int fmj_set_contact_maps(fmj_ctx* c, int32_t n_rows, const int32_t* geom_sensor, int32_t n_pairs, const int32_t* pairs) {
  if (!c || n_rows < 0 || n_pairs < 0 || (c->ngeom && !geom_sensor) || (n_pairs && !pairs)) return set_err(FMJ_ERR_ARG, "fmj_set_contact_maps: NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  for (int g = 0; g < c->ngeom; g++) {
    if (geom_sensor[g] >= n_rows) return set_err(FMJ_ERR_ARG, "fmj_set_contact_maps: row out of range");
    c->geom_sensor[g] = geom_sensor[g];
  }
  for (int p = 0; p < n_pairs; p++)
    if (pairs[3 * p] < 0 || pairs[3 * p] >= c->ngeom || pairs[3 * p + 1] < 0 || pairs[3 * p + 1] >= c->ngeom || pairs[3 * p + 2] < 0 || pairs[3 * p + 2] >= n_rows)
      return set_err(FMJ_ERR_ARG, "fmj_set_contact_maps: pair out of range");
  std::vector<int> gs(c->geom_sensor), pr(pairs, pairs + 3 * n_pairs);
  const int* dg = nullptr; const int* dp = nullptr;
  int rc;
  if ((rc = upload(c, gs, &dg)) || (rc = upload(c, pr, &dp))) return rc;
  c->d_geom_sensor = (int*)dg; c->d_pairs = (int*)dp; c->n_pairs = n_pairs; c->n_contact_rows = n_rows;
  return FMJ_OK;
}

int fmj_contacts2data(fmj_ctx* c, const fmj_data* d, const fmj_rows* rows, const fmj_units* units, void* stream) {
  if (!c || !d || !rows || !units || !rows->contacts || !d->contact || !d->ncon) return set_err(FMJ_ERR_ARG, "fmj_contacts2data: NULL argument");
  if (!c->d_geom_sensor) return set_err(FMJ_ERR_ARG, "fmj_contacts2data: call fmj_set_contact_maps first");
  StepArgs A; memset(&A, 0, sizeof A);
  A.n_envs = c->n_envs; A.contact = d->contact; A.ncon = d->ncon; A.contacts_rows = rows->contacts; A.status = d->status;
  fill_units(&A, units);
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(fmj_contacts2data_kernel, dim3(c->n_envs), dim3(64), 0, (hipStream_t)stream, c->dm, A, c->n_contact_rows,
                     (const int*)c->d_geom_sensor, c->n_pairs, (const int*)c->d_pairs);
  HIP_TRY(hipGetLastError());
  return FMJ_OK;
}

// ---- oscillator-network controller (include/fmj.h: fmj_cpg_*) ---------------------------------------------------
struct fmj_cpg {
  int device, n_osc, n_conn, nu, max_deg;
  float4* osc;      // [n_osc] 2 pi f, a, R, -
  int* row;         // [n_osc + 1] CSR by target oscillator
  float4* conn;     // [n_conn] from (bits), w, phi, -
  float4* out;      // [nu] a (bits), b (bits), gain, offset
};

__global__ void __launch_bounds__(64) fmj_cpg_kernel(const int n_osc, const int nu, const float4* __restrict__ osc,
                                                     const int* __restrict__ row, const float4* __restrict__ conn,
                                                     const float4* __restrict__ outp, const int n_envs, const int n_steps,
                                                     const float h, float* phase, float* amp, float* damp,
                                                     const float* drive, float* tape) {
  __shared__ float TH[64], RR[64];
  const int env = blockIdx.x, lane = threadIdx.x;
  const bool iso = lane < n_osc;
  const int ol = iso ? lane : 0;
  const float4 o = osc[ol];
  const float dr = drive ? drive[env] : 1.f;
  const float omega = o.x * dr, a = o.y, R = o.z;
  const int k0 = row[ol], k1 = iso ? row[ol + 1] : k0;
  float th = iso ? phase[(size_t)env * n_osc + lane] : 0.f;
  float r = iso ? amp[(size_t)env * n_osc + lane] : 0.f;
  float rd = iso ? damp[(size_t)env * n_osc + lane] : 0.f;
  for (int s = 0; s < n_steps; s++) {
    TH[lane] = th; RR[lane] = r;
    __syncthreads();
    float* t = tape + ((size_t)s * n_envs + env) * nu;
    for (int u = lane; u < nu; u += 64) {
      const float4 q = outp[u];
      const int ia = __float_as_int(q.x), ib = __float_as_int(q.y);
      float v = q.w;
      if (ia >= 0) v += q.z * RR[ia] * (1.f + cosf(TH[ia]));
      if (ib >= 0) v -= q.z * RR[ib] * (1.f + cosf(TH[ib]));
      t[u] = v;
    }
    float dth = omega;
    for (int k = k0; k < k1; k++) {
      const float4 c = conn[k];
      const int j = __float_as_int(c.x);
      dth = fmaf(RR[j] * c.y, sinf(TH[j] - th - c.z), dth);
    }
    const float rdd = a * (0.25f * a * (R - r) - rd);
    __syncthreads();
    th = fmaf(h, dth, th);
    r = fmaf(h, rd, r);
    rd = fmaf(h, rdd, rd);
    if (th > 3.14159265358979f) th -= 6.28318530717959f;        // only differences and cosines of phases are used
  }
  if (iso) { phase[(size_t)env * n_osc + lane] = th; amp[(size_t)env * n_osc + lane] = r; damp[(size_t)env * n_osc + lane] = rd; }
}

int fmj_cpg_create(const fmj_cpg_desc* d, int32_t device, fmj_cpg** out) {
  if (!d || !out || d->n_osc < 1 || d->n_osc > 64 || d->n_conn < 0 || d->nu < 1) return set_err(FMJ_ERR_ARG, "fmj_cpg_create: need 1..64 oscillators, nu >= 1");
  if (!d->frequency || !d->rate || !d->amplitude || !d->out_a || !d->out_b || !d->out_gain || !d->out_offset ||
      (d->n_conn && (!d->conn_to || !d->conn_from || !d->conn_weight || !d->conn_bias))) return set_err(FMJ_ERR_ARG, "fmj_cpg_create: NULL array");
  for (int k = 0; k < d->n_conn; k++)
    if (d->conn_to[k] < 0 || d->conn_to[k] >= d->n_osc || d->conn_from[k] < 0 || d->conn_from[k] >= d->n_osc) return set_err(FMJ_ERR_ARG, "fmj_cpg_create: connection index out of range");
  for (int u = 0; u < d->nu; u++)
    if (d->out_a[u] >= d->n_osc || d->out_b[u] >= d->n_osc) return set_err(FMJ_ERR_ARG, "fmj_cpg_create: output oscillator out of range");
  HIP_TRY(hipSetDevice(device));
  std::vector<float4> osc(d->n_osc), conn(d->n_conn ? d->n_conn : 1), outp(d->nu);
  std::vector<int> row(d->n_osc + 1, 0);
  for (int i = 0; i < d->n_osc; i++) osc[i] = make_float4((float)(6.283185307179586 * d->frequency[i]), (float)d->rate[i], (float)d->amplitude[i], 0.f);
  for (int k = 0; k < d->n_conn; k++) row[d->conn_to[k] + 1]++;
  for (int i = 0; i < d->n_osc; i++) row[i + 1] += row[i];
  { std::vector<int> fill(row.begin(), row.end() - 1);
    for (int k = 0; k < d->n_conn; k++) conn[fill[d->conn_to[k]]++] = make_float4(ibits(d->conn_from[k]), (float)d->conn_weight[k], (float)d->conn_bias[k], 0.f); }
  for (int u = 0; u < d->nu; u++) outp[u] = make_float4(ibits(d->out_a[u]), ibits(d->out_b[u]), (float)d->out_gain[u], (float)d->out_offset[u]);
  fmj_cpg* c = new fmj_cpg();
  c->device = device; c->n_osc = d->n_osc; c->n_conn = d->n_conn; c->nu = d->nu;
  c->osc = nullptr; c->row = nullptr; c->conn = nullptr; c->out = nullptr;
  bool ok = hipMalloc((void**)&c->osc, osc.size() * sizeof(float4)) == hipSuccess && hipMalloc((void**)&c->row, row.size() * sizeof(int)) == hipSuccess &&
            hipMalloc((void**)&c->conn, conn.size() * sizeof(float4)) == hipSuccess && hipMalloc((void**)&c->out, outp.size() * sizeof(float4)) == hipSuccess;
  ok = ok && hipMemcpy(c->osc, osc.data(), osc.size() * sizeof(float4), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(c->row, row.data(), row.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(c->conn, conn.data(), conn.size() * sizeof(float4), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(c->out, outp.data(), outp.size() * sizeof(float4), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { fmj_cpg_destroy(c); return set_err(FMJ_ERR_HIP, "fmj_cpg_create: device allocation failed"); }
  *out = c;
  return FMJ_OK;
}

void fmj_cpg_destroy(fmj_cpg* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->osc) (void)hipFree(c->osc);
  if (c->row) (void)hipFree(c->row);
  if (c->conn) (void)hipFree(c->conn);
  if (c->out) (void)hipFree(c->out);
  delete c;
}

int fmj_cpg_tape(fmj_cpg* c, int32_t n_envs, int32_t n_steps, double timestep, float* phase, float* amp, float* damp,
                 const float* drive, float* ctrl_tape, void* stream) {
  if (!c || n_envs <= 0 || n_steps < 0 || !phase || !amp || !damp || !ctrl_tape) return set_err(FMJ_ERR_ARG, "fmj_cpg_tape: bad argument");
  if (n_steps == 0) return FMJ_OK;
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(fmj_cpg_kernel, dim3(n_envs), dim3(64), 0, (hipStream_t)stream, c->n_osc, c->nu, c->osc, c->row, c->conn, c->out,
                     n_envs, n_steps, (float)timestep, phase, amp, damp, drive, ctrl_tape);
  HIP_TRY(hipGetLastError());
  return FMJ_OK;
}

int fmj_sc(const char* name) {
  static const struct { const char* n; int v; } tab[] = {
      {"LINK_COM_POS", FMJ_LINK_COM_POS}, {"LINK_COM_QUAT", FMJ_LINK_COM_QUAT}, {"LINK_URDF_POS", FMJ_LINK_URDF_POS},
      {"LINK_URDF_QUAT", FMJ_LINK_URDF_QUAT}, {"LINK_COM_LINVEL", FMJ_LINK_COM_LINVEL}, {"LINK_COM_ANGVEL", FMJ_LINK_COM_ANGVEL},
      {"LINK_SIZE", FMJ_LINK_SIZE}, {"JOINT_POSITION", FMJ_JOINT_POSITION}, {"JOINT_VELOCITY", FMJ_JOINT_VELOCITY},
      {"JOINT_FORCE", FMJ_JOINT_FORCE}, {"JOINT_TORQUE3", FMJ_JOINT_TORQUE3}, {"JOINT_TORQUE", FMJ_JOINT_TORQUE},
      {"JOINT_LIMIT_FORCE", FMJ_JOINT_LIMIT_FORCE}, {"JOINT_SIZE", FMJ_JOINT_SIZE}, {"CONTACT_REACTION", FMJ_CONTACT_REACTION},
      {"CONTACT_FRICTION", FMJ_CONTACT_FRICTION}, {"CONTACT_TOTAL", FMJ_CONTACT_TOTAL}, {"CONTACT_POSITION", FMJ_CONTACT_POSITION},
      {"CONTACT_SIZE", FMJ_CONTACT_SIZE}, {"XFRC_FORCE", FMJ_XFRC_FORCE}, {"XFRC_TORQUE", FMJ_XFRC_TORQUE}, {"XFRC_SIZE", FMJ_XFRC_SIZE}};
  if (!name) return -1;
  for (auto& t : tab) if (!strcmp(t.n, name)) return t.v;
  return -1;
}

}  // extern "C"
#endif  // FMJ_TU_MAXD
