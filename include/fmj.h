/* fmj.h — C-ABI of the MI355X-native batched locomotion step ("fmj" = farms-mujoco).
 *
 * This is the drop-in boundary for ONE hot path of farmsim/farms_mujoco: the per-step
 * physics + hydrodynamic drag + sensor readout.  Every entry point below names the
 * reference interface (file:line under the reference tree) it replaces.
 *
 *   - plain C, `extern "C"`, plain pointers and sizes, no torch / HIP types in signatures
 *     (a HIP stream is passed as `void*`; NULL = the null stream);
 *   - the library borrows caller-owned DEVICE buffers for the duration of a call and owns
 *     only its immutable device copy of the model (fp32) and small index tables;
 *   - every function returns an int status (0 = FMJ_OK), never throws, and records a
 *     message retrievable with fmj_last_error();
 *   - all per-environment arrays are fp32, batch-first, C-contiguous: `[n_envs, n, ...]`,
 *     one wavefront owns one environment row, so a row is one coalesced access.
 *
 * Quaternions: `xquat`/`qpos[3:7]` are w,x,y,z (MuJoCo convention); AnimatData link rows are
 * x,y,z,w (farms_core convention, permutation at reference physics.py:458,466).
 */
#ifndef FMJ_H_
#define FMJ_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMJ_ABI_VERSION 6

/* ---- status codes ------------------------------------------------------------------------ */
enum {
  FMJ_OK = 0,
  FMJ_ERR_ARG = 1,          /* NULL pointer / bad size / inconsistent model             */
  FMJ_ERR_UNSUPPORTED = 2,  /* model feature outside the supported subset               */
  FMJ_ERR_HIP = 3,          /* HIP runtime error (message has hipGetErrorString)        */
  FMJ_ERR_NODEVICE = 4      /* no GPU visible                                           */
};

/* ---- joint / geom enums (values follow MuJoCo's mjtJoint / mjtGeom) ---------------------- */
enum { FMJ_JNT_FREE = 0, FMJ_JNT_BALL = 1 /* unsupported */, FMJ_JNT_SLIDE = 2, FMJ_JNT_HINGE = 3 };
/* constraint solver and friction cone (values follow MuJoCo's mjtSolver / mjtCone; reference mjcf.py:1342-1353 forwards
 * simulation_options.solver / .cone to the option block, its own fallbacks being 'Newton' and 'pyramidal') */
enum { FMJ_SOLVER_PGS = 0, FMJ_SOLVER_CG = 1, FMJ_SOLVER_NEWTON = 2 };
enum { FMJ_CONE_PYRAMIDAL = 0, FMJ_CONE_ELLIPTIC = 1 };
/* integrator (values follow MuJoCo's mjtIntegrator; reference mjcf.py:1360-1365 forwards simulation_options.integrator, its own fallback
 * being 'Euler').  Implemented: Euler (with MuJoCo's implicit joint damping) and implicitfast.  For the models this path covers -
 * joint-transmission actuators, joint spring-dampers, no fluid or tendon forces - the velocity derivative of the smooth forces that
 * implicitfast keeps (passive + actuation, Coriolis terms dropped) is DIAGONAL: joint damping plus the velocity gain (-biasprm[2]) of
 * every actuator whose force is not clamped by its forcerange, so implicitfast is the Euler step with that sum in place of the damping.
 * RK4 (round 5): mj_RungeKutta with the classical tableau - per step four forward passes (launches of the step kernel with the integration
 * off) and a small kernel that advances the state between them; qacc is M^-1 (...) without the implicit damping, sensordata keeps the first
 * pass's values, poses / contacts / contact forces the last pass's (what mj_step leaves in mjData).  RK4 steps through fmj_step only
 * (fmj_data.qacc required); fmj_step_fused refuses it.  implicit (the full, non-symmetric derivative incl. Coriolis terms, LU-factored)
 * is refused. */
enum { FMJ_INT_EULER = 0, FMJ_INT_RK4 = 1, FMJ_INT_IMPLICIT = 2, FMJ_INT_IMPLICITFAST = 3 };
enum { FMJ_GEOM_PLANE = 0, FMJ_GEOM_HFIELD = 1, FMJ_GEOM_SPHERE = 2, FMJ_GEOM_CAPSULE = 3, FMJ_GEOM_CYLINDER = 5, FMJ_GEOM_BOX = 6, FMJ_GEOM_MESH = 7 };   /* mjtGeom values */

/* per-env warning bits written to fmj_data.status (dm_control raises PhysicsError on these;
 * reference simulation.py:157-161,176-179).  An env whose status carries one of the BAD* bits is FROZEN: from the
 * step that found the bad value on (and in every later launch until the caller clears the status word) its state is
 * left at the last finite values, it is no longer integrated and none of its ring-buffer rows are written
 * (the fused loop writes the links / xfrc row of iteration it + 1 in step it, so the row after the last good
 * step may already be there, computed from that step's finite state).  FMJ_WARN_CONTACTFULL only reports a truncated contact list and does not freeze. */
enum { FMJ_WARN_BADQPOS = 1, FMJ_WARN_BADQVEL = 2, FMJ_WARN_BADQACC = 4, FMJ_WARN_CONTACTFULL = 8 };
#define FMJ_WARN_FREEZE (FMJ_WARN_BADQPOS | FMJ_WARN_BADQVEL | FMJ_WARN_BADQACC)

/* ---- AnimatData column convention ("sc" in farms_core; reference physics.py:427-523) ------
 * farms_core is not vendored in the reference, so the integers are defined HERE and nowhere
 * else; host code queries them through fmj_sc(). Link CoM position at columns 0..2 and xfrc
 * force/torque at 0..2 / 3..5 are fixed by reference drag.pyx:189-191,265-267. */
enum {
  FMJ_LINK_COM_POS = 0,    /* 3 */
  FMJ_LINK_COM_QUAT = 3,   /* 4, xyzw */
  FMJ_LINK_URDF_POS = 7,   /* 3 */
  FMJ_LINK_URDF_QUAT = 10, /* 4, xyzw */
  FMJ_LINK_COM_LINVEL = 14,/* 3 */
  FMJ_LINK_COM_ANGVEL = 17,/* 3 */
  FMJ_LINK_SIZE = 20,
  FMJ_JOINT_POSITION = 0,
  FMJ_JOINT_VELOCITY = 1,
  FMJ_JOINT_FORCE = 2,     /* 3 (force sensors, unused unless use_frc_trq_sensors: fmj_step_fused stores 0 in columns 2..7 and 10..11) */
  FMJ_JOINT_TORQUE3 = 5,   /* 3 */
  FMJ_JOINT_TORQUE = 8,    /* motor torque = sum of the 3 actuatorfrc (physics.py:510-524) */
  FMJ_JOINT_LIMIT_FORCE = 9,
  FMJ_JOINT_SIZE = 12,
  FMJ_CONTACT_REACTION = 0, FMJ_CONTACT_FRICTION = 3, FMJ_CONTACT_TOTAL = 6,
  FMJ_CONTACT_POSITION = 9, FMJ_CONTACT_SIZE = 12,
  FMJ_XFRC_FORCE = 0, FMJ_XFRC_TORQUE = 3, FMJ_XFRC_SIZE = 6
};

/* ---- model: HOST pointers, fp64, MuJoCo mjModel naming ------------------------------------
 * Replaces the `const mjModel*` argument of mujoco.mj_step (reached through
 * reference simulation.py:53,83-89,156).  Supported subset = what reference mjcf.py emits
 * for an animat: one kinematic forest, each body carries 0 or 1 joint
 * (free / hinge / slide), explicit inertials, joint spring-dampers, position / velocity /
 * motor actuators on joints, Euler integrator with implicit joint damping. */
typedef struct fmj_model {
  int32_t abi_version;      /* = FMJ_ABI_VERSION */
  int32_t nbody, njnt, nq, nv, nu, ngeom;
  int32_t nM;               /* number of non-zeros of the sparse joint-space inertia    */
  double timestep;          /* option.timestep (reference mjcf.py:1187-1192,1329)      */
  double gravity[3];        /* option.gravity  (mjcf.py:1336-1341)                     */

  /* bodies [nbody]; body 0 = world */
  const int32_t* body_parentid;
  const int32_t* body_rootid;
  const int32_t* body_jntadr;   /* -1 if the body has no joint */
  const int32_t* body_dofadr;   /* -1 if no dof */
  const int32_t* body_dofnum;
  const double* body_pos;       /* [nbody,3] */
  const double* body_quat;      /* [nbody,4] wxyz */
  const double* body_ipos;      /* [nbody,3] */
  const double* body_iquat;     /* [nbody,4] */
  const double* body_mass;      /* [nbody]   */
  const double* body_inertia;   /* [nbody,3] principal moments */

  /* joints [njnt] */
  const int32_t* jnt_type;
  const int32_t* jnt_qposadr;
  const int32_t* jnt_dofadr;
  const int32_t* jnt_bodyid;
  const double* jnt_pos;        /* [njnt,3] */
  const double* jnt_axis;       /* [njnt,3] */
  const double* jnt_stiffness;  /* [njnt]   */
  const int32_t* jnt_limited;   /* [njnt]   */
  const double* jnt_range;      /* [njnt,2] */
  const double* jnt_solref;     /* [njnt,2] limit solref */
  const double* jnt_solimp;     /* [njnt,5] limit solimp */
  const double* jnt_margin;     /* [njnt]   */
  const double* qpos0;          /* [nq] */

  /* dofs [nv] */
  const int32_t* dof_bodyid;
  const int32_t* dof_jntid;
  const int32_t* dof_parentid;  /* -1 at a tree root */
  const int32_t* dof_Madr;      /* start of row i in the sparse M (row = i, parent(i), ...) */
  const double* dof_armature;
  const double* dof_damping;
  const double* dof_invweight0; /* [nv] (M^-1)_ii at qpos0; limit/contact regulariser     */

  /* actuators [nu]: force = gain*ctrl + bias0 + bias1*q + bias2*qdot (joint transmission) */
  const int32_t* actuator_jntid;
  const double* actuator_gain;       /* gainprm[0] */
  const double* actuator_bias;       /* [nu,3] biasprm[0:3] */
  const int32_t* actuator_ctrllimited;
  const double* actuator_ctrlrange;  /* [nu,2] */
  const int32_t* actuator_forcelimited;
  const double* actuator_forcerange; /* [nu,2] (task.py:279-286 rewrites this at run time) */

  /* collision geoms [ngeom] (config 4: animat geoms vs arena; reference mjcf.py:251-527).  Supported pairs: a ground geom
   * (world-attached plane, or ONE world-attached heightfield, reference mjcf.py:486-522 / task.py:108-123) against sphere
   * (1 contact), capsule (2: segment ends), cylinder (rim points: up to 4), box (first 4 penetrating corners) and convex
   * mesh (its 4 deepest penetrating vertices, see mesh_vert below).  The
   * heightfield is met as the plane of the grid triangle under each candidate point ("plane per cell").  At most 192
   * constraint rows per env: limited joints + 4 * max_contacts <= 192. */
  const int32_t* geom_type;     /* FMJ_GEOM_* */
  const int32_t* geom_bodyid;
  const double* geom_size;      /* [ngeom,3] MuJoCo sizes: sphere r; capsule / cylinder r, half length; box half extents */
  const double* geom_pos;       /* [ngeom,3] body frame */
  const double* geom_quat;      /* [ngeom,4] */
  const double* geom_friction;  /* [ngeom,3] */
  const double* geom_solref;    /* [ngeom,2] */
  const double* geom_solimp;    /* [ngeom,5] */
  const double* body_invweight0;/* [nbody,2] translational, rotational */
  /* the heightfield asset of the FMJ_GEOM_HFIELD geom (mjModel.hfield_*): nrow x ncol samples, row-major with the row
   * along +y and the column along +x of the geom frame, over [-size[0], size[0]] x [-size[1], size[1]]; elevation =
   * data * size[2] (the reference stores 2 * (image - 0.5) in hfield_data, task.py:108-115); size[3] = base depth, unused */
  int32_t hfield_nrow, hfield_ncol;
  double hfield_size[4];
  const double* hfield_data;    /* [nrow*ncol] or NULL */
  /* convex meshes (mjModel.mesh_vert of the geom's mesh asset; the reference turns every SDF mesh collision into one,
   * mjcf.py:270-413): geom g of type FMJ_GEOM_MESH owns vertices [geom_vertadr[g], + geom_vertnum[g]) of mesh_vert, in
   * the geom frame; they stand for their convex hull.  Against the ground a mesh gives up to 4 contacts, at its deepest
   * penetrating vertices (deepest first, equal depths in vertex order).  Not MuJoCo's mjc_PlaneConvex point selection
   * (support point + tilted directions), which is not restated here.  Explicit pairs: see mesh_face below. */
  int32_t nmeshvert;
  const double* mesh_vert;      /* [nmeshvert,3] or NULL */
  const int32_t* geom_vertadr;  /* [ngeom] (-1: not a mesh) or NULL when nmeshvert == 0 */
  const int32_t* geom_vertnum;  /* [ngeom] */
  /* explicit contact pairs between animat geoms (MJCF contact/pair, condim 3; the reference emits one per pair of
   * collision shapes of every morphology.self_collisions link pair, friction 0: mjcf.py:1012-1033).  Supported shapes:
   * sphere and capsule; one contact per pair, at the closest points of the two segments.  A sliding friction below
   * MuJoCo's mjMINMU = 1e-5 is raised to it. */
  int32_t npair;
  const int32_t* pair_geom1;    /* [npair] */
  const int32_t* pair_geom2;    /* [npair] */
  const double* pair_friction;  /* [npair] */
  const double* pair_solref;    /* [npair,2] */
  const double* pair_solimp;    /* [npair,5] */

  /* constraint solver options (mjcf.py:1330-1403) */
  int32_t solver_iterations;
  int32_t max_contacts;         /* per environment */
  double impratio;
  double solver_tolerance;
  double meaninertia;           /* mjModel.stat.meaninertia: mean diagonal of M at qpos0 (solver termination scale) */
  /* ABI 4 */
  int32_t solver;               /* FMJ_SOLVER_*: option.solver (mjcf.py:1348-1353).  HIP path: PGS, Newton or CG on any model of the subset; under
                                   Newton / CG an env with an ACTIVE explicit-pair contact is solved on the dual problem (PGS to tolerance) */
  int32_t cone;                 /* FMJ_CONE_*:   option.cone   (mjcf.py:1342-1347).  HIP path: pyramidal with any solver, elliptic with
                                   Newton / CG (three rows per contact in fmj_step_debug's rows; maxefc stays the 4-per-contact bound) */
  int32_t ls_iterations;        /* Newton / CG line-search iterations (MuJoCo option.ls_iterations, default 50); <= 0: 50 */
  int32_t noslip_iterations;    /* option.noslip_iterations (mjcf.py:1392-1397); 0 = off */
  double ls_tolerance;          /* option.ls_tolerance (default 0.01); <= 0: 0.01 */
  double noslip_tolerance;      /* option.noslip_tolerance (mjcf.py:1398-1403) */
  /* ABI 5 */
  int32_t integrator;           /* FMJ_INT_*: option.integrator (mjcf.py:1360-1365) */
  /* ABI 6: explicit pairs between ANY two collision shapes (round 5; the reference emits a pair for every collision shape of every
   * morphology.self_collisions link pair, and its usual collision shape is a convex mesh: mjcf.py:1012-1033,270-413).
   * A box, a cylinder and a mesh are POLYTOPES: vertices (box: 8 corners; cylinder: 12 points on each rim, the first on +x, enumerated in steps of 150 degrees so that equal-depth ties keep spread-out points;
   * mesh: mesh_vert) and outward faces (box, cylinder: analytic - the cylinder's true side surface; mesh: the planes of its hull,
   * mesh_face[k] = unit normal n and offset d in the geom frame, n . x <= d inside).  Narrow phase of a pair (geom1, geom2), normal
   * from geom1 to geom2, margin 0, restated by oracle collide_pair:
   *   sphere / capsule against sphere / capsule: one contact at the closest points of the two segments (rounds 2-4);
   *   polytope against sphere / capsule: each centre (a capsule's two end centres) against the polytope's faces: signed distance
   *     s = max over faces (n . c - d), contact when s - radius < 0 with the face of that maximum as the normal - exact where the
   *     closest feature is a face, an under-estimate of the distance (<= true distance) near edges and corners;
   *   polytope against polytope: every vertex of one inside the other (s < 0) is a candidate with the other's nearest face as normal;
   *     the (<= 4) deepest candidates are kept, deepest first, equal depths in candidate order (geom1's vertices, then geom2's):
   *     the rule of mesh against ground.  Edge-edge crossings without a vertex inside are not found.
   * Not MuJoCo's convex-convex pipeline (MPR / GJK + multi-contact heuristics): like the plane-mesh rule, a documented stand-in. */
  int32_t nmeshface;
  const double* mesh_face;      /* [nmeshface,4] or NULL */
  const int32_t* geom_faceadr;  /* [ngeom] (-1: not a mesh) or NULL when nmeshface == 0 */
  const int32_t* geom_facenum;  /* [ngeom] */
} fmj_model;

/* ---- per-env device buffers for the physics step -------------------------------------------
 * The mjData fields the reference reads or writes around mj_step
 * (reference task.py:317,332,343-346; physics.py:449-524).  DEVICE pointers, fp32. */
typedef struct fmj_data {
  float* qpos;               /* [n_envs,nq]      in/out */
  float* qvel;               /* [n_envs,nv]      in/out */
  const float* ctrl;         /* [n_envs,nu]      in     */
  const float* qpos_spring;  /* [n_envs,nq]      in  (per-env: task.py:343-346)          */
  const float* xfrc_applied; /* [n_envs,nbody,6] in, world frame force(3),torque(3); may be NULL */
  /* derived fields, valid for the state BEFORE integration (mj_step semantics) */
  float* xpos;               /* [n_envs,nbody,3] */
  float* xquat;              /* [n_envs,nbody,4] wxyz */
  float* xipos;              /* [n_envs,nbody,3] */
  float* sensordata;         /* [n_envs,nsensordata] layout: fmj_get_sensor_layout()      */
  float* qacc;               /* [n_envs,nv] may be NULL */
  float* time;               /* [n_envs]   may be NULL */
  int32_t* status;           /* [n_envs]   warning bits, OR-accumulated                   */
  /* constraint path: required when the model has joint limits or collision geoms, else may be NULL */
  float* qacc_warmstart;     /* [n_envs,nv] in/out: qacc of the previous step (PGS warm start)  */
  float* contact;            /* [n_envs,max_contacts,16] out: pos(3) frame(9: normal,t1,t2) force(3: normal,t1,t2
                                in the contact frame, what mj_contactForce returns) and one int32 (bit pattern in
                                the float slot) = geom1 << 16 | geom2 (mjContact.geom1 / geom2)            */
  int32_t* ncon;             /* [n_envs] out: active contacts of the last forward pass              */
} fmj_data;

/* sensordata layout, in the order reference mjcf.py:950-1002 adds the sensors */
typedef struct fmj_sensor_layout_t {
  int32_t nsensordata;
  int32_t framelinvel_adr;   /* + 6*(body-1)     : framelinvel(3), frameangvel(3) interleaved per link */
  int32_t jointpos_adr;      /* + 3*j            : jointpos, jointvel, jointlimitfrc per non-free joint */
  int32_t actuatorfrc_adr;   /* + a              : actuatorfrc per actuator */
  int32_t first_link_body;   /* bodies [first_link_body, nbody) carry link sensors */
  int32_t first_sensor_jnt;  /* joints [first_sensor_jnt, njnt) carry joint sensors */
} fmj_sensor_layout_t;

/* ---- AnimatData ring-buffer rows (reference task.py:62,158,208-216) --------------------------
 * `[buffer_size, n_envs, n, width]` fp32 device tensors; `*_row` point at ONE ring index. */
typedef struct fmj_rows {
  float* links;     /* [n_envs,n_links,FMJ_LINK_SIZE]   */
  float* joints;    /* [n_envs,n_joints,FMJ_JOINT_SIZE] */
  float* xfrc;      /* [n_envs,n_links,FMJ_XFRC_SIZE]   */
  float* contacts;  /* [n_envs,n_contact_sensors,FMJ_CONTACT_SIZE] or NULL */
} fmj_rows;

/* unit scaling (farms_core SimulationUnitScaling; reference physics.py:428-524) */
typedef struct fmj_units {
  float meters, newtons, torques, velocity, angular_velocity, kilograms;
} fmj_units;

/* water + per-link swimming constants (reference drag.pyx:271-306,333-387) */
typedef struct fmj_water {
  float surface;       /* WaterProperties._surface (1e8 when sph, drag.pyx:386-387) */
  float density;       /* stored, unused by the reference arithmetic (drag.pyx:142) */
  float viscosity;
  float velocity[3];   /* global frame */
  float gravity;       /* -9.81 hard-coded at drag.pyx:409 */
  int32_t use_buoyancy;
} fmj_water;

typedef struct fmj_ctx fmj_ctx;   /* opaque */

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* Replaces mjcf.Physics.from_mjcf_model (reference simulation.py:53): validates the model,
 * builds the level / ancestor tables, uploads the fp32 device copy. `device` = HIP ordinal. */
int fmj_create(const fmj_model* model, int32_t n_envs, int32_t device, fmj_ctx** out);
void fmj_destroy(fmj_ctx* ctx);
const char* fmj_last_error(void);
int fmj_abi_version(void);
int fmj_get_sensor_layout(const fmj_ctx* ctx, fmj_sensor_layout_t* out);
/* LDS bytes / VGPR-independent facts the host needs for reporting */
int fmj_kernel_info(const fmj_ctx* ctx, int32_t* lds_bytes_per_env, int32_t* threads_per_env);

/* ---- swimming links (SwimmingHandler.__init__, reference drag.pyx:333-387) ------------------
 * n_xfrc_rows: rows per env of the xfrc array (len(data.sensors.xfrc.names); the env stride of every xfrc row
 * address); links_index / xfrc_index: row of each swimming link in links / xfrc arrays (xfrc_index < n_xfrc_rows);
 * body_index: MuJoCo body id of each swimming link (datalinks2xfrc, physics.py:385-393);
 * coefficients [ns,2,3]; masses, heights, densities [ns]. HOST pointers, copied. */
int fmj_set_swimming(fmj_ctx* ctx, int32_t ns, int32_t n_xfrc_rows, const int32_t* links_index,
                     const int32_t* xfrc_index, const int32_t* body_index,
                     const double* coefficients, const double* masses,
                     const double* heights, const double* densities);

/* Run-time rewrite of the actuator force limits: what ExperimentTask.initialize_control does to
 * physics.named.model.actuator_forcelimited / actuator_forcerange (reference task.py:253-286: position and velocity
 * actuators of motors that are not position-controlled get forcerange = [0, 0]).  forcelimited [nu], forcerange
 * [nu,2]: HOST pointers, copied; the call drains the device (hipDeviceSynchronize) and then refreshes the device table, so it is
 * ordered against launches of this context on any stream. */
int fmj_set_actuator_forcerange(fmj_ctx* ctx, int32_t nu, const int32_t* forcelimited, const double* forcerange);

/* link / joint readout maps (get_physics2data_maps, reference physics.py:188-393):
 * link row i <- body links_body[i]; joint row j <- joint joints_jnt[j]. HOST pointers. */
int fmj_set_readout_maps(fmj_ctx* ctx, int32_t n_links, const int32_t* links_body,
                         int32_t n_joints, const int32_t* joints_jnt);

/* ---- the hot path ------------------------------------------------------------------------- */
/* mujoco.mj_step(model, data) x n_steps for every env (reference simulation.py:156,175 via
 * dm_control Environment.step -> Physics.step; legacy_step=False, simulation.py:36-37,45).
 * ctrl_step_stride: element stride between successive steps' ctrl rows (0 = same ctrl). */
int fmj_step(fmj_ctx* ctx, const fmj_data* d, int32_t n_steps, int64_t ctrl_step_stride,
             void* hip_stream);

/* mj_forward without integration: fills xpos/xquat/xipos/sensordata/qacc for the current
 * qpos/qvel.  disable_actuation=1 reproduces what dm_control runs after
 * physics.reset(keyframe_id=0) (reference task.py:137): mj_forward with actuation disabled. */
int fmj_forward(fmj_ctx* ctx, const fmj_data* d, int32_t disable_actuation, void* hip_stream);

/* Diagnostic twin of fmj_forward (parity tests of the intermediate stages): additionally stores, per env, the rows of
 * H = M + diag(armature + timestep * damping) as the step assembles them - H_rows [n_envs, nv, *row_stride] DEVICE,
 * row i holds H[i][j] at column depth(j) for the dofs j on the chain root -> i - and qfrc_smooth [n_envs, nv] =
 * passive - bias + actuation + J' xfrc_applied.  Not on the product path. */
int fmj_forward_debug(fmj_ctx* ctx, const fmj_data* d, int32_t disable_actuation, float* H_rows, int32_t* row_stride,
                      float* qfrc_smooth, void* hip_stream);

/* Diagnostic twin of fmj_step for ONE step of a model with constraints (parity tests of the constraint solve; the
 * reference reads the same quantities from mjData.efc_* after mj_step).  Besides stepping, it stores per env the
 * constraint rows as the solver left them - efc_rows [n_envs, maxefc, 8] DEVICE = {pos, aref, R (after the pyramidal rule),
 * b = J qacc_smooth - aref, force, R before the pyramidal rule, int32 bits: row kind (bit 30 = contact) | joint dof or
 * contact index, mu}, rows ordered as MuJoCo orders them (limits by joint and side, then 4 per contact); the number of
 * rows of an env is (limited sides active) + 4 * ncon - and, if pgs_improvement != NULL [n_envs, solver_iterations] DEVICE,
 * the decrease of the dual cost achieved by every PGS sweep that ran (never negative for a correct sweep; entries of sweeps
 * that did not run are left untouched).  maxefc / solver_iterations: fmj_constraint_info.  Not on the product path. */
/* Under Newton / CG with the ELLIPTIC cone only slots 2 (R), 3, 4 (force), 6 (kind | id) and 7 (mu) of a contact's rows are meaningful:
 * the solver keeps the contact's 3 x 3 Hessian block in slots 0 / 1 and the cone exchange in slot 5 (fmj_newton.inc); under Newton /
 * CG in general slot 3 holds J qacc_smooth - aref. */
int fmj_step_debug(fmj_ctx* ctx, const fmj_data* d, float* efc_rows, float* pgs_improvement, void* hip_stream);
int fmj_constraint_info(const fmj_ctx* ctx, int32_t* maxefc, int32_t* max_contacts, int32_t* solver_iterations);
/* The solver that actually runs (ABI 5).  fmj_create replaces a requested Newton / CG by the dual solver (PGS run to the solver's
 * tolerance, up to 10 x solver_iterations sweeps: the same convex problem, the same minimiser) when some ground contact's friction
 * after impratio is below 1e-3 - rows an fp32 primal iteration cannot resolve (the reference's arena has friction 0, mjcf.py:1202) -;
 * *requested / *effective receive FMJ_SOLVER_* values, *iterations the sweep / iteration budget in force (fmj_step_debug's
 * pgs_improvement rows are per sweep of the EFFECTIVE solver).  Any pointer may be NULL. */
int fmj_solver_info(const fmj_ctx* ctx, int32_t* requested, int32_t* effective, int32_t* iterations);

/* SwimmingHandler.step(iteration) (reference drag.pyx:389-411 -> drag_forces :152-268) for
 * every env: reads rows->links, writes rows->xfrc (rows of links above the surface are left
 * untouched, drag.pyx:192-194).  If xfrc_applied != NULL also performs the glue the reference
 * leaves to an external callback: xfrc_applied[body] = R_body * (F,T) * units,
 * zero for links above the surface. */
int fmj_drag(fmj_ctx* ctx, const fmj_rows* rows, const fmj_water* water,
             const fmj_units* units, float* xfrc_applied, void* hip_stream);

/* drag_forces(iteration, data_links, links_index, data_xfrc, xfrc_index, coefficients, z3, z4, water, mass, height,
 * density, gravity, use_buoyancy) (reference drag.pyx:152-268) for ONE link in every env, no context needed:
 * links_row / xfrc_row point at that link's row of env 0, *_env_stride = elements between envs.  Returns through
 * hydrodynamics_out [n_envs] (DEVICE int32, may be NULL) what the reference returns (1 = the link is in the water). */
int fmj_drag_link(int32_t n_envs, int32_t device, const float* links_row, int64_t links_env_stride, float* xfrc_row,
                  int64_t xfrc_env_stride, const double* coefficients /* [2,3] */, double mass, double height,
                  double density, const fmj_water* water, int32_t* hydrodynamics_out, void* hip_stream);

/* physics2data(physics, iteration, data, maps, units, links_only) (reference
 * physics.py:527-545): mjData fields -> AnimatData rows with unit scaling. */
int fmj_physics2data(fmj_ctx* ctx, const fmj_data* d, const fmj_rows* rows,
                     const fmj_units* units, int32_t links_only, void* hip_stream);

/* Contact sensors: map of collision geoms / geom pairs to AnimatData contact rows
 * (geompair2data, reference physics.py:360-382).  geom_sensor[g] = row of key (g, -1) or -1;
 * pairs = n_pairs x (geom_a, geom_b, row) = the keys (geom_a, geom_b) of geompair2data.  A contact (geom1, geom2)
 * is added to the rows of the keys (geom1, geom2) and (geom1, -1) with sign -1 and to those of (geom2, geom1) and
 * (geom2, -1) with sign +1 (reference sensors.pyx:163-169).  HOST pointers, copied. */
int fmj_set_contact_maps(fmj_ctx* ctx, int32_t n_contact_sensors, const int32_t* geom_sensor,
                         int32_t n_pairs, const int32_t* pairs);

/* cycontacts2data(physics, iteration, data, geompair2data, meters, newtons) (reference
 * sensors.pyx:140-190): contact list of the last forward pass -> rows->contacts (row is
 * overwritten: reaction, friction, total force, force-weighted position). */
int fmj_contacts2data(fmj_ctx* ctx, const fmj_data* d, const fmj_rows* rows, const fmj_units* units,
                      void* hip_stream);

/* ExperimentTask.before_step up to the host callbacks (reference task.py:168-182) in ONE launch (ABI 6): the rows of
 * fmj_physics2data (FMJ_BEFORE_ROWS; FMJ_BEFORE_LINKS_ONLY: the links rows alone, a sub-step), of fmj_contacts2data
 * (FMJ_BEFORE_CONTACTS) and the swimming callback's fmj_drag on the links row just written (FMJ_BEFORE_DRAG; xfrc_applied as in
 * fmj_drag, may be NULL) - the same device code as the three operators, the same bits, one launch instead of three: what an
 * iteration with HOST callbacks costs is this launch plus the step (Simulation.run(fused=False)).  water may be NULL without
 * FMJ_BEFORE_DRAG.  Frozen envs are skipped. */
#define FMJ_BEFORE_ROWS 1
#define FMJ_BEFORE_LINKS_ONLY 2
#define FMJ_BEFORE_CONTACTS 4
#define FMJ_BEFORE_DRAG 8
int fmj_before_step(fmj_ctx* ctx, const fmj_data* d, const fmj_rows* rows, const fmj_water* water, const fmj_units* units,
                    int32_t flags, float* xfrc_applied, void* hip_stream);

/* Fused loop: for s in [0,n_steps): physics2data(row (it0+s)%buffer) -> drag -> xfrc_applied
 * -> ctrl -> mj_step, state and derived fields resident in LDS between steps
 * (ExperimentTask.before_step + Environment.step, reference task.py:168-186, simulation.py:155-156).
 * rows_base point at ring index 0; row_stride_* = elements between ring indices.
 * controller: 0 = ctrl tape (ctrl + s*ctrl_step_stride), 1 = built-in travelling-wave
 * position controller (see fmj_wave_controller).
 * Sub-steps (ABI 5; reference mjcf.py:1187-1192,1329: the model's timestep is options.timestep / num_sub_steps, and
 * simulation.py:148-156 runs n_iterations * substeps environment steps): with substeps = S > 1 an iteration is S calls of
 * mj_step; n_steps counts ITERATIONS (the launch starts on a full step and runs n_steps * S physics steps).  Only the first
 * sub-step of an iteration is a "full step" (task.py:175): it writes the links / joints / contact rows, runs the drag and the
 * controller (time = iteration * S * model timestep; a ctrl tape holds one row per iteration); the ctrl and - unless
 * substep_links - the drag force then stay as they are for the remaining sub-steps.  With substep_links != 0 (some callback
 * asked for sub-steps, task.py:63,176,181: here the swimming callback) every sub-step also writes a links-only row and
 * recomputes the drag from it, at the ring index of task.iteration AS THE REFERENCE COUNTS IT: after_step advances the
 * iteration when (sim_iteration + 1) % S == 0 (task.py:352-355), i.e. after sub-step S - 2 of each group, so the sub-steps
 * j >= S - 1 of iteration m write row m + 1 (overwritten by the next full step) and, for S >= 3, sub-steps 1 .. S - 2 overwrite
 * the links part of row m itself (reproduced, not fixed). */
typedef struct fmj_wave_controller {
  const float* amplitude;   /* [nu] DEVICE, per actuator (0 for non-position actuators) */
  const float* phase_lag;   /* [nu] DEVICE */
  const float* env_phase;   /* [n_envs] DEVICE */
  float frequency;          /* Hz */
} fmj_wave_controller;

typedef struct fmj_fused_args {
  int32_t n_steps;
  int32_t iteration0;       /* task.iteration of the first step */
  int32_t buffer_size;      /* ring length (task.py:62) */
  int32_t do_readout;       /* write links/joints rows each step */
  int32_t do_drag;          /* SwimmingHandler.step + xfrc glue each step */
  int32_t controller;       /* 0 tape, 1 wave */
  int64_t ctrl_step_stride;
  int64_t row_stride_links, row_stride_joints, row_stride_xfrc;
  int64_t row_stride_contacts;  /* used when rows_base.contacts != NULL: contact rows are written too
                                   (needs fmj_set_contact_maps) */
  fmj_rows rows_base;
  fmj_water water;
  fmj_units units;
  fmj_wave_controller wave;
  float* ctrl_out;          /* [n_envs,nu] DEVICE or NULL: with controller 1, receives the ctrl of the LAST step of the
                               launch (what physics.data.ctrl holds after task.py:288-346 ran for that iteration) */
  const int32_t* env_order; /* [n_envs] DEVICE or NULL: a permutation of the envs; workgroup b of the one-env kernel steps
                               env_order[b].  Results do not depend on it; listing the envs with the most contacts first
                               keeps them from starting last and setting the launch time (constraint models) */
  /* ABI 5 */
  int32_t substeps;         /* physics steps per iteration (simulation_options.num_sub_steps; task.py:61,64-66); <= 1: one */
  int32_t substep_links;    /* != 0: sub-steps write links-only rows and recompute the drag (a callback with substep=True) */
  /* ABI 6 */
  int32_t n_iterations;     /* the run's n_iterations (task.py:49), 0 = unknown.  With sub-steps task.iteration reaches n_iterations one
                               sub-step before the run ends (task.py:352-355); the reference never executes that sub-step's before_step
                               (its assert at task.py:170; dm_control's first step only resets), here it runs and writes NO rows and
                               keeps the drag force it has - the ring index n_iterations % buffer_size would be row 0 of a full log */
  int32_t rows_ahead;       /* != 0 (the per-iteration host path, Simulation.run(fused=False)): everything before_step writes for the launch's FIRST
                               iteration is already there (fmj_before_step, or the launch before) and xfrc_applied may have been edited by host
                               callbacks since: nothing of it is written, the drag force is read from fmj_data::xfrc_applied; the launch
                               writes instead, after its last step, the rows, the drag and xfrc_applied of the iteration that FOLLOWS it - an
                               iteration with host callbacks is then ONE launch.  Models without constraints that run two envs per wave only
                               (FMJ_ERR_UNSUPPORTED otherwise), substeps <= 1, xfrc_applied not NULL when do_drag. */
} fmj_fused_args;

int fmj_step_fused(fmj_ctx* ctx, const fmj_data* d, const fmj_fused_args* args, void* hip_stream);

/* ---- on-device controller: a network of amplitude-controlled phase oscillators --------------------
 * The reference only defines the AnimatController interface (task.py:292-346: step/positions/torques/springrefs);
 * the oscillator networks themselves live in its callers (farms_amphibious). This is the batched device counterpart:
 *   theta_i' = 2 pi f_i + sum_k r_j(k) w_k sin(theta_j(k) - theta_i - phi_k)      (connections k into i)
 *   r_i''    = a_i (a_i / 4 (R_i - r_i) - r_i')
 * integrated with explicit Euler at the physics timestep, one wave per env, lane = oscillator (n_osc <= 64).
 * Output u of nu:  ctrl_u = gain_u (r_a (1 + cos theta_a) - r_b (1 + cos theta_b)) + offset_u   (b < 0: single-sided)
 * fmj_cpg_tape advances the network n_steps and writes ctrl_tape[s][env][u] of the state BEFORE each step, which is
 * exactly what fmj_step / fmj_step_fused consume as a ctrl tape (controller 0, ctrl_step_stride = n_envs * nu). */
typedef struct fmj_cpg_desc {
  int32_t n_osc, n_conn, nu;
  const double* frequency;    /* [n_osc] Hz */
  const double* rate;         /* [n_osc] a_i */
  const double* amplitude;    /* [n_osc] R_i */
  const int32_t* conn_to;     /* [n_conn] i */
  const int32_t* conn_from;   /* [n_conn] j */
  const double* conn_weight;  /* [n_conn] w */
  const double* conn_bias;    /* [n_conn] phi */
  const int32_t* out_a;       /* [nu] oscillator a of output u (-1: output is offset only) */
  const int32_t* out_b;       /* [nu] oscillator b or -1 */
  const double* out_gain;     /* [nu] */
  const double* out_offset;   /* [nu] */
} fmj_cpg_desc;
typedef struct fmj_cpg fmj_cpg;
int fmj_cpg_create(const fmj_cpg_desc* desc, int32_t device, fmj_cpg** out);
void fmj_cpg_destroy(fmj_cpg* cpg);
/* phase, amp, damp: [n_envs][n_osc] DEVICE fp32 (updated in place); drive: [n_envs] DEVICE fp32 or NULL, scales the
 * intrinsic frequencies per env; ctrl_tape: [n_steps][n_envs][nu] DEVICE fp32. */
int fmj_cpg_tape(fmj_cpg* cpg, int32_t n_envs, int32_t n_steps, double timestep, float* phase, float* amp, float* damp,
                 const float* drive, float* ctrl_tape, void* hip_stream);

/* enum query so host code never hard-codes column integers: name is e.g. "LINK_COM_POS" */
int fmj_sc(const char* name);

#ifdef __cplusplus
}
#endif
#endif /* FMJ_H_ */
