// hb_device.hpp — device-side model tables and batch buffers (fp32), shared by the host
// runtime (hb_api.cpp) and the kernel translation units (hb_step.hip, hb_step_duo.hip, hb_narrow.hip, hb_env.hip).
//
// The model is replicated read-only per device as two flat arrays (int, float); DevModel holds
// typed pointers into them plus the per-env LDS layout.  All tables are small (a few KB) and
// identical for every env, so lane-indexed reads hit L1/L2 and uniform reads go through the
// scalar cache.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// Model tables are immutable for the lifetime of a batch: device code addresses them through the
// constant address space so that wave-uniform reads become scalar (s_load) instructions served by
// the scalar cache, even though the kernel also writes global memory.
#if defined(__HIP_DEVICE_COMPILE__)
#define HB_CONST __attribute__((address_space(4)))
#else
#define HB_CONST
#endif

namespace hb {

constexpr int kGroup = 64;      // lanes cooperating on one env (one wavefront)
constexpr int kNconMax = 24;    // contact capacity per env (overflow -> HB_WARN_CONTACTFULL)
constexpr int kNefcMax = 63;    // constraint-row capacity per env (overflow -> HB_WARN_CNSTRFULL); lane 63 / row 63 of C carries the extra right-hand side
// the general instantiations (mesh hulls, height-field prisms, condim 4 / 6: the reference's own robot) with the Newton solver hold
// their rows in kBigGroups groups of 64 (lane l owns rows l, l + 64, ...): 256 rows, 48 contacts
constexpr int kBigGroups = 4;
constexpr int kBigNefcMax = 64 * kBigGroups;
constexpr int kBigNconMax = 48;
// ... and with PGS in kPgsGroups groups with the row-by-row matrix AR in LDS (a condim 4 / 6 model solved by PGS: variant 3): 128 rows
constexpr int kPgsGroups = 2;
constexpr int kPgsNefcMax = 64 * kPgsGroups;
constexpr int kListMax = 128;   // general collision: pairs that survive the broadphase per step (more: HB_WARN_CONTACTFULL)
constexpr int kWorkMax = 256;   // general collision: narrowphase work items (pair or pair x prism) per step
constexpr int kConStride = 20;  // floats per contact record in LDS
constexpr int kDiagConStride = 16;
constexpr int kMetaStride = 4;  // floats per row in the general variants' row meta
constexpr int kMeshChunk = 8;   // neighbour records per climb round's load batch (DevModel::mesh_nbr)
constexpr int kMeshStart = 96;  // start records per mesh: cube map, 6 faces x 4 x 4 (DevModel::mesh_start)
constexpr int kCountStride = 8;  // ints per env in BatchPtrs::counts: ncon, nefc, niter, cost, self-collision flag, spare
constexpr int kBrecQuads = 18;  // float4s per level-ordered body record (see build_device_model)

// contact record layout in LDS (floats)
enum { C_DIST = 0, C_POS = 1, C_FRAME = 4, C_PAIR = 13, C_ROW = 14, C_DIM = 15, C_FRIC = 16 };

struct DevModel {
  // sizes
  int nq, nv, nu, nbody, njnt, ngeom, ntendon, nM, npair, nlevel, ntree, nlimcand, nhfielddata;
  int nstate;   // floats per env in the global state record: time, qpos, qvel, qacc_warmstart
  int nobs;
  // which instantiation of the step kernel runs this model: 0 classic (plane / sphere / capsule, condim 1 / 3), 1 general collision
  // (mesh hulls and height-field prisms through MPR, condim 1 / 3 / 4 / 6) with the 63-row solvers, 2 general collision + Newton on
  // kBigNefcMax rows, 3 general collision + PGS on kPgsNefcMax rows (a PGS model with condim 4 / 6 pairs)
  int variant, ncon_max, nefc_max;
  int mpr_iterations;
  float mpr_tolerance;
  // options
  float timestep, gravity[3], inv_sqrt_impratio, tolerance, pgs_scale;
  int iterations, disableflags;
  int solver, ls_iterations;  // mjtSolver (0 PGS, 2 Newton); Newton line-search evaluation cap
  float ls_tolerance;
  // bodies.  Level-ordered records, kBrecQuads float4 each (slot 0 = world, slot s = lane s-1 of the tree passes):
  // [0] b,parent,jntnum,jntadr  [1] depth,treeid,mass,childnum  [2] pos  [3] quat  [4] ipos  [5] iquat  [6] inertia
  // [7..8] children[8]  [9+3j] joint j: (type,qposadr,dofadr,qpos0) (axis) (pos)
  const float4 HB_CONST* brec;
  const int HB_CONST* body_treeid;
  const float HB_CONST *body_invweight0, *tree_invmass;
  const unsigned long long HB_CONST* body_dofmask;  // bit d set: dof d moves this body
  // joints
  const int HB_CONST *jnt_type, *jnt_qposadr, *jnt_dofadr;
  const float HB_CONST* qpos0;
  // dofs
  const float4 HB_CONST* drec;   // per dof: (jntid,bodyid,type,k) (treeid,armature,damping,stiffness) (qposadr,qpos_spring,-,-)
  const int HB_CONST *dof_jntid, *dof_Madr;
  const float HB_CONST* dof_damping;
  // sparse mass matrix (ancestor-chain layout of mjModel.dof_Madr)
  const int HB_CONST* mrec;      // per entry: i | j << 8 | body(i) << 16
  const float2 HB_CONST* mdiag;  // per entry: (armature, damping) on the diagonal, 0 elsewhere
  // dense view of the sparse mass matrix (both solvers eliminate it on the matrix cores): [32 columns j][32 rows i] -> index of the {M, H} pair holding
  // M(i, j) (nM: the zero pad pair, nM + 1: the one pad pair = identity beyond nv)
  const int HB_CONST* mdense;
  const int HB_CONST* mdense_c;  // the same view in the MFMA accumulator layout: [16 registers][64 lanes]
  // geoms
  const int HB_CONST *geom_type, *geom_bodyid;
  const float HB_CONST *geom_size, *geom_pos, *geom_quat, *geom_rbound;
  const float HB_CONST* geom_half;  // [ngeom][3] half extents of the geom's bounding box in its own frame
  int box_cull;                     // 1: oriented-box test behind the bounding spheres of a portal-search pair (HB_BOX_CULL=0 switches it off: tests)
  // meshes: hull vertices as 16-byte records (x, y, z, link) and per geom the first record / the count (0 for other geom types);
  // link = first neighbour record << 8 | number of kMeshChunk-record chunks; mesh_nbr: one record per (vertex, neighbour): (x, y, z
  // of the neighbour, the neighbour's own link): the hull's edge graph with the coordinates inlined, every vertex's list padded to
  // whole chunks with copies of the vertex itself (never an improvement), so that a climb round is one batch of independent
  // 16-byte loads (hb_mpr.hpp: ccd_support).  mesh_start: per mesh a cube map of kMeshStart records (6 faces x 4 x 4 cells): the
  // hull's support vertex for the cell's centre direction, where the first climb of a test starts.
  const float4 HB_CONST* mesh_vert;
  const float4 HB_CONST* mesh_nbr;
  const float4 HB_CONST* mesh_start;
  const int HB_CONST *geom_meshadr, *geom_meshnum;
  // height fields (static terrain on the world body): per geom the field id (-1 otherwise)
  const int HB_CONST *geom_dataid, *hfield_nrow, *hfield_ncol, *hfield_adr;
  const float HB_CONST *hfield_size, *hfield_data;
  // collision candidates with pre-mixed contact parameters
  const int HB_CONST *pair_geom1, *pair_geom2, *pair_dim, *pair_self;
  // per pair, 5 float4: [0] body1,body2,tree1,tree2 [1] dofmask1 (lo,hi), dofmask2 (lo,hi) [2] margin-gap, solref[2], invweight0 sum
  // [3] solimp[0..3] [4] solimp[4], condim
  const float4 HB_CONST* prec;
  // per pair, 3 float4 for mj_collision: [0] geom1, geom2, type1 | type2 << 8, margin [1] rbound1, rbound2, size1[0..1] [2] size2[0..1]
  const float4 HB_CONST* crec;
  const float4 HB_CONST* arec;  // per actuator: 4 quads (hb_api.cpp)
  const float4 HB_CONST* lrec;  // per limit candidate: 4 quads (hb_api.cpp)
  const float HB_CONST* pair_fricab;  // per pair: (sliding friction of the floor geom if it is in the pair, else 0; the other geom's / the mixed one)
  const float HB_CONST *pair_friction, *pair_solref, *pair_solimp, *pair_margin, *pair_gap;
  // limit candidates: 2 per limited joint/tendon in constraint order (lower, upper)
  const int HB_CONST *lim_kind, *lim_id, *lim_side;  // kind 0 = joint, 1 = tendon
  const float HB_CONST *lim_range, *lim_margin, *lim_solref, *lim_solimp, *lim_invweight;
  // tendons (fixed)
  const int HB_CONST *tendon_adr, *tendon_num, *wrap_dofadr, *wrap_qposadr;
  const float HB_CONST* wrap_prm;
  const float4 HB_CONST* trec;  // per tendon, 3 float4: coefficients, qpos addresses, dof addresses of its first four wraps
  // actuators
  const int HB_CONST *act_qposadr, *act_dofadr, *act_ctrllimited, *act_forcelimited;
  const float HB_CONST *act_gear, *act_ctrlrange, *act_forcerange, *act_gain, *act_bias;
  // env adapter
  int obs_root_body, obs_root_dofadr, obs_root_qadr;
  const int HB_CONST* obs_src;  // [nobs - 3]: state-record offset each copied observation entry comes from (-1: zero)
  const int HB_CONST* obs_jnt;  // [(nobs - 6) / 2]: the scalar joints in observation order (joint order, or actuator order: hb_env_config)
  // LDS layout (float offsets per env) — persistent region
  int o_gquat;  // general collision only: world orientation of every geom (4 floats each)
  int o_meta;   // general variants: per-row (R, K imp (pos - margin), B, -) written by makeConstraint
  int o_AR;     // variant 3 (PGS on kPgsNefcMax rows): the matrix AR, [kPgsNefcMax][kPgsNefcMax]
  int o_qpos, o_qvel, o_warm, o_ctrl, o_gpos, o_gaxis, o_scom, o_cdof, o_qLD, o_smooth, o_vec0, o_vec1, o_vec2, o_tenlen;
  // region A (dynamics scratch)
  int o_xpos, o_xmat, o_xipos, o_xanchor, o_xaxis, o_cinert, o_crb, o_cvel;
  // region B (constraints), aliases region A: contacts, C rows, row meta (later W), forces
  int o_con, o_C, o_efc, o_force;
  int lds_floats;  // total floats per env
  int cstride;     // row stride of C (odd: conflict-free lane-strided access; the last column is zero padding)
};

typedef const DevModel HB_CONST& DevModelRef;

// device copy of hb_env_config (include/hb.h), same field order
constexpr int kEnvMaxPairs = 16;
struct EnvConfig {
  float target_velocity[2];
  float target_z, min_z, max_time, safe_torque, control_frequency, action_scale;
  float w_hvel, w_upright, w_height, w_torque, w_ctrl_change, w_ctrl_reg, w_symmetry;
  float self_collision_penalty, terminal_reward, upright_tol;
  int n_equal, n_opposite;
  int equal_pairs[kEnvMaxPairs][2];
  int opposite_pairs[kEnvMaxPairs][2];
  int auto_reset, reset_keyframe;
  float reset_perturb;
  int reward_kind;
  float w_vvel, min_z_grounded;
  int reset_collision_mode;
  float reset_quat_perturb;
  int obs_actuator_order;
};

// hb_env_randomization (include/hb.h), same layout
struct EnvRand {
  float factor;
  unsigned seed;
  float control_timestep;
  float joint_angle_noise, joint_velocity_noise, gyro_noise, imu_noise, action_noise;
  float min_delay, max_delay;
  int frozen_noise, push_enabled;
  float push_min_interval, push_max_interval, push_min_duration, push_max_duration, push_min_force, push_max_force;
};
// hb_domain_randomization (include/hb.h), same layout
struct DomainRand {
  float factor;
  unsigned seed;
  float friction_min_mult, friction_max_mult, max_mass_change, max_external_mass;
  float armature_max_change, stiffness_max_change, margin_max_change, range_max_change;
  float kp_nominal, kp_max_change, force_limit_max_change;
  float floor_bump_min, floor_bump_max;  // height-field elevations in [0, min + factor (max - min)] (cpu_env.py:267-280); max 0: the model's own data
};
// per-env model parameters [n_env][stride]: mass[nbody] | armature[nv] | stiffness[nv] | lim_margin[nlimcand] |
// lim_range[nlimcand] | act_gain[nu] | act_bias1[nu] | act_forcerange[2 nu] | floor friction scale | hfield_data[nhfielddata]
struct DomainLayout {
  int o_mass, o_arm, o_stiff, o_lmargin, o_lrange, o_gain, o_bias1, o_frc, o_fric, o_hfield, stride;
};
__host__ __device__ inline DomainLayout domain_layout(int nbody, int nv, int nlimcand, int nu, int nhfielddata) {
  DomainLayout L;
  L.o_mass = 0; L.o_arm = nbody; L.o_stiff = L.o_arm + nv; L.o_lmargin = L.o_stiff + nv; L.o_lrange = L.o_lmargin + nlimcand;
  L.o_gain = L.o_lrange + nlimcand; L.o_bias1 = L.o_gain + nu; L.o_frc = L.o_bias1 + nu; L.o_fric = L.o_frc + 2 * nu; L.o_hfield = L.o_fric + 1; L.stride = L.o_hfield + nhfielddata;
  return L;
}
constexpr int kDelaySlots = 64;  // ring size of the delay FIFOs (delays <= 63 control steps)
// per-env state of the realism layer, all [n_env]-major device arrays (null when hb_env_randomize is off)
struct EnvRandState {
  int* k_act;      // actions pushed this episode
  int* k_obs;      // observations pushed this episode
  int* delay;      // [n][4]: action, joints, gyro, gravity (control steps)
  float* fifo_act;   // [n][kDelaySlots][nu]
  float* fifo_joint; // [n][kDelaySlots][2 nscalar]
  float* fifo_gyro;  // [n][kDelaySlots][3]
  float* fifo_grav;  // [n][kDelaySlots][3]
  float* push;     // [n][8]: start, duration, magnitude, dir x, dir y, body, event count, -
  float* xfrc;     // [n][nbody][6] (the batch's xfrc_applied)
};

// fused policy kernel (hb_policy_kernel): layer sizes, host-packed weights (MFMA B-operand order) and biases
// MJPC "Humanoid Stand" cost (tasks/humanoid/stand/stand.cc:41-104): layout of the read-out row and the cost terms
struct StandTask {
  int n_feet, o_head, o_feet, o_com, o_vel, o_qvel, o_ctrl, nv, nu, stride;
  float height_goal, risk;
  int norm[5];
  float weight[5], p[5], q[5];
};

// MJPC "Humanoid Walk" cost (tasks/humanoid/walk/walk.cc:44-163): read-out row offsets and the cost terms (dims as the
// task's user sensors declare them, applied to the residual vector in order)
struct WalkTask {
  int o_torso, o_foot_r, o_foot_l, o_pelvis, o_com, o_vel, o_axes, o_linvel, o_sub, o_qpos, o_ctrl, nq, nu, stride;
  float height_goal, speed_goal, risk;
  int nterm, dim[8], norm[8];
  float weight[8], p[8], q[8];
};

// hb_task_cost: cost terms over consecutive slices of a residual vector (task.cc:71-110)
struct CostSpec {
  int nterm, dim[8], norm[8];
  float weight[8], p[8], q[8], risk;
};

struct PolicyDesc {
  int nl;
  int sizes[5];
  const float* w[4];
  const float* b[4];
  int ldx;  // LDS row stride of an activation tile: widest layer + 4 (K is swept four columns at a time: zero pad columns)
};

// Buffers of the STAGED step of the general variants (launch_step): hb_pose_kernel writes every geom's world pose and the env's
// narrowphase work items, hb_narrow_kernel evaluates the items (one per lane, at the occupancy of a small kernel: the MPR climbs are
// chains of dependent loads), the step kernel appends the results in item order.  All null: the step kernel does all of it itself.
struct StageBufs {
  float* geom;     // [n_env][10 ngeom]: world positions[3 ngeom] | z axes[3 ngeom] | orientation quaternions[4 ngeom]
  int4* item;      // [n_env][kWorkMax]: the env's items that need a portal search: prisms of height-field pairs from the front, pairs
                   // of geoms from the back, each in item order:
                   // env, pair | item index << 16, sub-item | ncols << 16, rmin | cmin << 16 (sub-grid of a height-field pair)
  int* nsearch;    // [n_env][2] how many of each kind
  int* nwork;      // [n_env] work items of the env (clamped to kWorkMax)
  float4* result;  // [n_env][kWorkMax][4]: dist0, pos0 | normal0, n | dist1, pos1 | normal1, pair
  int nq, nv, nu, pose_lds;  // host-side copies for the per-step pointer arithmetic and the pose kernel's LDS bytes
  // Variant 2 (Newton, 256 rows in four register groups, one wave per SIMD): the step kernel proper is first the ONE-group
  // instantiation on the variant-1 layout (dm_fast; 63 rows, 24 contacts, two waves per SIMD); an env whose step needs more
  // raises defer[env] and leaves its state alone, and the four-group kernel then steps exactly those envs (rerun = 1).
  const DevModel* dm_fast;  // null: no fast pass (not variant 2, or diagnostics on: their buffers have the big kernel's strides)
  int fast_lds;             // its dynamic LDS bytes
  int* defer;               // [n_env]
  // the deferred envs of a launch as a list, so that the second pass is a handful of waves that loop over it instead of one wave per env
  // that looks at its flag and leaves (4096 waves of the four-group kernel, one per SIMD: 18 us): defer_list[blk0 + k], k < defer_count[blk0]
  // (blk0: the launch's first slot - every env segment has its own stretch and counter).  hb_pose_kernel zeroes the counter.
  int* defer_list;
  int* defer_count;
  int rerun;                // set by launch_step for the second pass
  int no_mesh;              // the model has no mesh geoms: the narrowphase launch is hb_narrow_prim_kernel
};

// ---- Size-specialised instantiation of the classic PGS step kernel (step_body's SIZED) ---------------------------------------------
// Every loop bound and every LDS offset of the step kernel is a function of the model's sizes.  For the size signature of the
// reference's 27-dof humanoid (simulation/mujoco/model/humanoid/humanoid.xml) they are compile-time constants in one
// more instantiation: addresses fold into the instructions' offset fields and 40 SGPRs spill instead of 84.  Any model with this
// signature takes it (the host compares sizes AND the layout it computed itself, field by field: build_device_model); every other
// model takes the generic kernels.
struct SizedModel {
  int nq, nv, nu, nbody, njnt, ngeom, ntendon, nM, ntree, npair, nlevel, nlimcand, nstate, cstride;
  int o_qpos, o_qvel, o_warm, o_ctrl, o_gpos, o_gaxis, o_scom, o_cdof, o_qLD, o_smooth, o_vec0, o_vec1, o_vec2, o_tenlen;
  int o_xpos, o_xmat, o_xipos, o_xanchor, o_xaxis, o_cinert, o_crb, o_cvel;
  int o_con, o_C, o_efc, o_force, o_gquat, o_meta, lds_floats;
};
// the classic layout (variant 0, full capacity) as build_device_model's lay() computes it
constexpr SizedModel sized_model(int nq, int nv, int nu, int nbody, int njnt, int ngeom, int ntendon, int nM, int ntree, int npair, int nlevel, int nlimcand, int variant = 0, int cstride = 33) {
  SizedModel z{};
  z.nq = nq; z.nv = nv; z.nu = nu; z.nbody = nbody; z.njnt = njnt; z.ngeom = ngeom; z.ntendon = ntendon; z.nM = nM; z.ntree = ntree; z.npair = npair; z.nlevel = nlevel; z.nlimcand = nlimcand;
  z.nstate = 1 + nq + 2 * nv; z.cstride = cstride;
  int off = 0;
  auto up = [](int n) { return (n + 3) & ~3; };
  z.o_gquat = 0; z.o_meta = 0;
  if (variant) { z.o_gquat = off; off += up(4 * ngeom); }  // (general collision: world orientation of every geom)
  z.o_qpos = off; off += up(nq); z.o_qvel = off; off += up(nv); z.o_warm = off; off += up(nv); z.o_ctrl = off; off += up(nu > 1 ? nu : 1);
  z.o_gpos = off; off += up(3 * ngeom); z.o_gaxis = off; off += up(3 * ngeom); z.o_scom = off; off += up(3 * (ntree > 1 ? ntree : 1)); z.o_cdof = off; off += up(12 * nv);
  z.o_qLD = off; off += up(2 * nM + 4); z.o_smooth = off; off += up(nv);
  z.o_vec0 = off; off += 32; z.o_vec1 = off; off += 32; z.o_vec2 = off; off += 32; z.o_tenlen = off; off += up(ntendon > 1 ? ntendon : 1);
  const int region = off;
  z.o_xpos = off; off += up(12 * nbody); z.o_xmat = off; off += up(9 * nbody); z.o_xipos = off; off += up(3 * nbody);
  z.o_xanchor = off; off += up(3 * njnt); z.o_xaxis = off; off += up(3 * njnt); z.o_cinert = off; off += up(10 * nbody); z.o_crb = off; off += up(20 * nbody);
  z.o_cvel = off; off += up(12 * nbody);
  const int endA = off;
  off = region;
  z.o_con = off; off += up(kNconMax * kConStride); z.o_C = off; off += up((kNefcMax + 1) * cstride);
  z.o_efc = off; off += up(13 * kNefcMax > 32 * 36 ? 13 * kNefcMax : 32 * 36);
  if (variant == 1) { z.o_meta = off; off += up(64 * kMetaStride); }
  z.o_force = off; off += up(kGroup > kNefcMax ? kGroup : kNefcMax);
  z.lds_floats = endA > off ? endA : off;
  return z;
}
constexpr SizedModel kSizedHumanoid27 = sized_model(28, 27, 21, 17, 22, 20, 2, 243, 1, 159, /*tree levels*/ 7, /*limit candidates*/ 46);
// the same humanoid on a height field (configs[4]; general collision, PGS: the variant-1 layout of the staged step's fast kernel)
// the reference's own robot (simulation/assets/world.xml + humanoid.xml: 18 dofs, 13 geoms) on the variant-1 layout of its fast Newton kernel
constexpr SizedModel kSizedTeamV1 = sized_model(19, 18, 12, 15, 13, 13, 0, 117, 1, 37, /*tree levels*/ 5, /*limit candidates*/ 24, /*variant*/ 1, /*J row stride of a Newton kernel of dense order 20*/ 21);
constexpr SizedModel kSizedHumanoid27V1 = sized_model(28, 27, 21, 17, 22, 20, 2, 243, 1, 159, 7, 46, /*variant*/ 1);

// LDS of hb_pose_kernel in floats: qpos | body poses (12 floats each, kXpqStride) | geom position, z axis, quaternion | the work lists
__host__ __device__ inline int pose_lds_floats(int nq, int nb, int ngeom) {
  return ((nq + 3) & ~3) + 12 * nb + 2 * ((3 * ngeom + 3) & ~3) + 4 * ngeom + kListMax * 5 + kWorkMax;
}

// most step calls one launch executes (the calls the host enqueued back to back, hb_api.cpp fold_steps)
constexpr int kFoldMax = 256;
struct BatchPtrs {
  float* state;        // [n_env][nstate]
  const float* ctrl;   // [n_env][nu] or [T][n_env][nu]
  float* qpos_out;     // nullable, [T][n_env][nq]
  float* qvel_out;     // nullable, [T][n_env][nv]: with qpos_out the recorded states of a trajectory (trajectory.cc:175-190)
  float* xfrc;         // nullable, [n_env][nbody][6]
  float xfrc_rate, xfrc_scale;  // rollout noise (hb_rollout_noise): xfrc <- rate * xfrc + scale * N(0, 1) before every step; scale 0: off
  unsigned xfrc_seed, xfrc_call;
  int* status;         // [n_env] accumulated HB_WARN_* bits
  int* counts;         // [n_env][kCountStride]
  float* qfrc_out;     // nullable [n_env][nv]: qfrc_smooth + qfrc_constraint of the last step (env adapter's joint torques)
  float* diag_qacc;    // nullable [n_env][nv]
  float* diag_force;   // nullable [n_env][kNefcMax]
  float* diag_contact; // nullable [n_env][kNconMax][kDiagConStride]
  int n_env;
  int blk0, nblk;      // this launch covers dispatch slots blk0 .. blk0+nblk-1 (one block each) of the batch
  int ctrl_mode;       // 0: ctrl[e][nu] held for all steps; 1: ctrl[t][e][nu]; 2: on-device Halton; 3: ctrl_tab[t][e][nu]
  int t0, env_offset;  // Halton indexing
  int integrate;       // 1: mj_step, 0: mj_forward only
  // heavy-first block scheduling (nullable): slot s takes env = order[s], a permutation sorted by
  // the cost of each env's previous step (counts[4*e+3]) so the most expensive envs are dispatched first
  const int* order;    // [n_env]
  const int* order2;   // [n_env] nullable: the same for the narrowphase launch of a staged step, by the cost of its waves (counts[..][7])
  const float* dr;     // nullable [n_env][dr_stride]: per-env model parameters (DomainLayout)
  int dr_stride;
  // sensor read-out (hb_rollout_sensors), nullable: [T][n_env][sensor_stride] = framepos of sensor_body[0..n) | subtreecom | subtreelinvel of tree sensor_tree
  float* sensor_out;
  int sensor_stride, sensor_nframe, sensor_tree;  // sensor_tree < 0: no subtree sensors
  int sensor_body[16];
  float sensor_off[16][3];  // framepos of a site: offset in the body frame (zero: the body frame itself)
  int sensor_flags;         // bit 0: append qvel[nv], bit 1: append ctrl[nu], bit 2: append qpos[nq] (the inputs MJPC residuals read beside sensors)
  // further read-outs, appended in this order behind framepos / subtreecom / subtreelinvel:
  int sensor_naxis, sensor_axis_body[8], sensor_axis_which[8];  // framexaxis (0) / framezaxis (2) of a body frame (objtype xbody)
  int sensor_nlinvel, sensor_linvel_body[8];                     // framelinvel, objtype body: velocity of the inertial frame origin, world axes
  int sensor_nsub;                                               // subtreelinvel of further bodies (any body, not only a tree root)
  unsigned long long sensor_submask[4];                          //   bit b: body b belongs to that subtree
  float sensor_subinv[4];                                        //   1 / subtree mass
  const unsigned char* env_mask;  // nullable [n_env]: envs with a zero byte are skipped by this launch
  unsigned long long* stamps;  // diagnostic builds (-DHB_STAMPS) only: [n_env][16] s_memtime stamps of the last step
  int lean_ok;                // bit 0: the model's options allow the lean instantiations (mjOption.disableflags == 0); bit 1: its sizes and LDS
                              // layout are kSizedHumanoid27's (the size-specialised instantiations); bit 2: its fast layout is kSizedTeamV1's
  int stop_phase;             // diagnostic builds only: 0 = off (HB_STOP_PHASE in the environment, read at every launch)
  const float* ctrl_tab[kFoldMax];  // ctrl_mode 3: step t of this launch is the step call whose [n_env][nu] controls these are (hb_api.cpp: fold_steps)
  int duo;                    // host side only (launch_step): two envs per wave 0 never, 1 where it pays, 2 always (hb_batch_duo)
  StageBufs stage;
};

}  // namespace hb
