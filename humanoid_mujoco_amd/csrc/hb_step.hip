// hb_step.hip — the fused physics-step kernel for gfx950 (MI355X).
// (one of four kernel translation units: hb_step.hip - this file, the fused step kernels; hb_narrow.hip - the staged step's pose and
// narrowphase kernels; hb_env.hip - env adapter, policy, task costs, reset; hb_step_duo.hip - two envs per wave)
//
// One 64-lane wavefront advances ONE environment through the whole mj_step pipeline
// (reference API: simulation/mujoco/include/mujoco/mujoco.h:120 mj_step; stage list
// mujoco.h:247-372; oracle: oracle/mjstep_oracle.c) with every intermediate resident in LDS
// or registers.  HBM traffic per env-step is the state record in and out (plus ctrl), i.e. the
// algorithmic bytes of SURVEY.md §8(d).  A multi-step rollout keeps the state on chip between
// steps.  See DESIGN.md for the lane mappings of each stage and the LDS map.
//
// Lane mappings (G = 64 lanes per env):
//   tree passes (kinematics, comVel/RNE forward)  lanes = bodies of one depth level
//   backward passes (crb, cfrc)                   lanes = (body, component), pull from children
//   qM                                            lanes = sparse mass-matrix entries
//   L^T D L                                       lanes = update triples of one pivot dof
//   collision                                     lanes = candidate geom pairs
//   constraint rows, half-solve, AR, PGS          lane  = constraint row
//   dof vectors                                   lanes = dofs
#include "hb_kcommon.hpp"
#include "hb_collide.hpp"
#include "hb_launch.hpp"

namespace hb {

// SOLVER: mjtSolver of the instantiation (0 = PGS, 2 = Newton); everything outside the constraint solve, the mass-matrix
// factorisation and the integrator's damped solve is shared.
// COLL: 0 = the classic narrowphase (plane / sphere / capsule pairs, condim 1 / 3, inline), 1 = the general one (adds mesh hulls and
// height-field prisms through MPR, condim 4 / 6).  NG: constraint rows live in NG groups of 64 (lane l owns rows l + 64 g); NG > 1
// only with the Newton solver (kBigGroups: 256 rows, for the reference's own robot: ten pyramid rows per condim-6 contact).
// DEFER: the fast pass of a staged step.  1 (variant 2, StageBufs::dm_fast): an env-step that overflows this instantiation's rows or
// contacts is not stepped here but flagged for the four-group kernel.  2 (variant 1): same capacities as the full kernel, nothing to
// overflow into.  Both: an env-step whose qacc comes out bad (mj_checkAcc: reset, second forward pass with a narrowphase of its own) is
// flagged as well, so that a fast instantiation carries no portal-search code at all (it costs the step kernel ~ 280 spilled registers).
// (sizes and LDS offsets: the model's, or - SIZED - the constants of kSizedHumanoid27, hb_device.hpp)
#define HB_SZ(f) (SIZED ? (NDENSE == 20 ? kSizedTeamV1.f : COLL ? kSizedHumanoid27V1.f : kSizedHumanoid27.f) : M.f)
// the fast pass of a staged step hands an env to the second pass: its flag, and a place in the launch's list (StageBufs::defer_list)
__device__ __forceinline__ void defer_env(const BatchPtrs& P, int env) {
  P.stage.defer[env] = 1;
  if (P.stage.defer_list) P.stage.defer_list[P.blk0 + atomicAdd(&P.stage.defer_count[P.blk0], 1)] = env;
}
// the second pass: a few waves that walk the list (any other launch: one wave per slot, as ever)
#define HB_STEP_OR_RERUN(...)                                                                               \
  do {                                                                                                      \
    const bool rr_ = P.stage.rerun && P.stage.defer_list;                                                   \
    const int n_ = rr_ ? P.stage.defer_count[P.blk0] : (int)gridDim.x;                                      \
    for (int i_ = (int)blockIdx.x; i_ < n_; i_ += (int)gridDim.x) step_body<__VA_ARGS__>(Mp, P, nsteps, rr_ ? P.stage.defer_list[P.blk0 + i_] : -1); \
  } while (0)
template <int SOLVER, int NDENSE, int COLL = 0, int NG = 1, int DEFER = 0, int LEAN = 0, int SIZED = 0>
__device__ __forceinline__ void step_body(const DevModel* Mp, const BatchPtrs& P, int nsteps_in, int env_in = -1) {
  const int nsteps = LEAN == 1 ? 1 : nsteps_in;  // (LEAN == 1 is launched for single steps only: the step API; rollouts take LEAN == 2)
  // LEAN (1 = a single step without the constraint-force read-out; 2 = any number of steps, read-out optional): a launch without the optional inputs and outputs (applied forces and their noise, constraint-force / sensor / trajectory
  // read-outs, diagnostics, per-env model parameters, an env mask; mj_step, not mj_forward) - known at compile time, so their tests,
  // pointers and code are not in the kernel at all
  float* const P_xfrc = LEAN ? nullptr : P.xfrc;
  const float P_xfrc_scale = LEAN ? 0.f : P.xfrc_scale, P_xfrc_rate = LEAN ? 0.f : P.xfrc_rate;
  const auto P_xfrc_seed = P.xfrc_seed; const auto P_xfrc_call = P.xfrc_call;
  float* const P_qfrc_out = LEAN == 1 ? nullptr : P.qfrc_out;  // (LEAN == 2: the lean kernel of the env adapter, whose reward reads the constraint forces)
  float* const P_sensor_out = LEAN ? nullptr : P.sensor_out;
  float* const P_qpos_out = LEAN ? nullptr : P.qpos_out;
  float* const P_qvel_out = LEAN ? nullptr : P.qvel_out;
  float* const P_diag_qacc = LEAN ? nullptr : P.diag_qacc;
  float* const P_diag_force = LEAN ? nullptr : P.diag_force;
  float* const P_diag_contact = LEAN ? nullptr : P.diag_contact;
  const float* const P_dr = LEAN ? nullptr : P.dr;
  const int P_dr_stride = LEAN ? 0 : P.dr_stride;
  const unsigned char* const P_env_mask = LEAN ? nullptr : P.env_mask;
  const int P_integrate = LEAN ? 1 : P.integrate;
  static_assert(SIZED == 0 || (NG == 1 && ((NDENSE == 28 && (COLL == 0 || SOLVER == 0)) || (NDENSE == 20 && COLL == 1 && SOLVER == 2))), "the size-specialised instantiations: the humanoid (classic or variant-1 layout) and the robot (Newton, variant-1 layout)");
  static_assert(NG == 1 || SOLVER == 2 || (COLL == 1 && NG == kPgsGroups && DEFER == 0), "PGS on more than one row group: the general variant's kPgsGroups instantiation");
  constexpr int kNR = NG == 1 ? kNefcMax : 64 * NG;  // row capacity of this instantiation
  constexpr int kNC = NG == 1 ? kNconMax : kBigNconMax;  // contact capacity
  // the model tables are read through a constant-address-space pointer (not by-value kernel
  // arguments): the ~100 table pointers and every wave-uniform table entry are fetched on demand by
  // scalar loads through the scalar cache instead of living in (and spilling from) SGPRs
  DevModelRef M = *(const DevModel HB_CONST*)(uintptr_t)Mp;
  const int M_disableflags = LEAN ? 0 : M.disableflags;  // (a lean launch: the host has checked that the model disables nothing, BatchPtrs::lean_ok)
  // PGS instantiation of dense order <= 28: M^-1 = W W' from an elimination on the matrix cores instead of the sparse
  // L'DL schedule (which stays for 29..32 dofs)
  extern __shared__ float lds[];
  const int lane0 = threadIdx.x;
  int lane = lane0;
  if (env_in < 0 && (int)blockIdx.x >= P.nblk) return;
  const int slot = P.blk0 + (int)blockIdx.x;
  const int env = env_in >= 0 ? env_in : (P.order ? P.order[slot] : slot);  // (env_in: the second pass of a staged step walks the list of deferred envs)
  if (P_env_mask && !P_env_mask[env]) return;  // masked stepping (hb_env_reset's settle step)
  if constexpr (COLL != 0 && DEFER == 0) {
    if (P.stage.rerun) {  // second pass of a staged step: only the envs the fast pass deferred
      const int d = P.stage.defer[env];
      if (!d) return;
      if (lane0 == 0) P.stage.defer[env] = 0;
    }
  }
  // per-env model parameters (domain randomisation), nullable; offsets per DomainLayout
  const float* dr = P_dr ? P_dr + (size_t)env * P_dr_stride : nullptr;
  const DomainLayout DL = domain_layout(HB_SZ(nbody), HB_SZ(nv), HB_SZ(nlimcand), HB_SZ(nu), M.nhfielddata);
  const int nv = HB_SZ(nv), nq = HB_SZ(nq), nb = HB_SZ(nbody), cs = HB_SZ(cstride);

  float* s_qpos = lds + HB_SZ(o_qpos);
  float* s_qvel = lds + HB_SZ(o_qvel);
  float* s_warm = lds + HB_SZ(o_warm);
  float* s_ctrl = lds + HB_SZ(o_ctrl);
  float* s_gpos = lds + HB_SZ(o_gpos);
  float* s_gaxis = lds + HB_SZ(o_gaxis);
  float* s_scom = lds + HB_SZ(o_scom);
  float* s_cdof = lds + HB_SZ(o_cdof);
  // {M, H = M + h*diag(damping)} interleaved, in the sparse dof-ancestor layout: assembled once per step, read as dense views
  f32x2* s_qLD = reinterpret_cast<f32x2*>(lds + HB_SZ(o_qLD));
  float* s_smooth = lds + HB_SZ(o_smooth);  // qfrc_smooth
  float* s_v0 = lds + HB_SZ(o_vec0);        // scratch dof vectors
  float* s_v1 = lds + HB_SZ(o_vec1);
  float* s_v2 = lds + HB_SZ(o_vec2);
  float* s_tenlen = lds + HB_SZ(o_tenlen);
  float* s_xpq = lds + HB_SZ(o_xpos);  // per body: xpos[3], -, xquat[4] (two ds_read/write_b128)
  float* s_xmat = lds + HB_SZ(o_xmat);
  float* s_xipos = lds + HB_SZ(o_xipos);
  float* s_xanchor = lds + HB_SZ(o_xanchor);
  float* s_xaxis = lds + HB_SZ(o_xaxis);
  float* s_cinert = lds + HB_SZ(o_cinert);
  float* s_if = lds + HB_SZ(o_crb);   // per body: composite inertia[10] | cfrc[6]
  float* s_va = lds + HB_SZ(o_cvel);  // per body: cvel[6] | cacc[6]
  float* s_con = lds + HB_SZ(o_con);
  float* s_C = lds + HB_SZ(o_C);
  float* s_efc = lds + HB_SZ(o_efc);  // per-row meta, stride kNR; dead once the row quantities are in registers
  float* s_W = lds + HB_SZ(o_efc);    // W = L^-1 D^-1/2, [32][33], aliases the row meta
  float* s_force = lds + HB_SZ(o_force);
  const float* s_gquat = lds + HB_SZ(o_gquat);  // general collision only: world orientation of every geom
  float* s_meta = lds + HB_SZ(o_meta);          // general variants: per-row (R, K imp (pos - margin), B, -)
  float* s_AR = lds + M.o_AR;              // PGS on several row groups only: the matrix AR, [kNR][kNR]
  (void)s_AR;
  (void)s_gquat; (void)s_meta;
  constexpr int kCs = 33;            // row stride of C (odd: conflict-free lane-strided access; column 32 is zero padding)
  static_assert(kNefcMax == kGroup - 1, "C holds kNefcMax constraint rows plus the qfrc_smooth row");
  // per-row meta slots
  enum { E_POS = 0, E_MARGIN, E_SOLREF0, E_SOLREF1, E_IMP0, E_IMP1, E_IMP2, E_IMP3, E_IMP4, E_DA, E_DAFIRST, E_MU2, E_FORCE, E_NSLOT };

  // the controls of the first step are requested before the state, those of step t+1 at the top of step t:
  // an HBM round trip each (streamed, never cached) that would otherwise open every step
  float ctrl_pf = 0.f;
  // (ctrl_mode 3: step calls the host folded into this launch, BatchPtrs::ctrl_tab - multi-step instantiations only)
  const int ctrl_mode = P.ctrl_mode, ctrl_t0 = P.t0;
  const float* ctrl_src = (LEAN != 1 && ctrl_mode == 3) ? P.ctrl_tab[0] : P.ctrl;
  if (ctrl_mode != 2 && lane < HB_SZ(nu)) ctrl_pf = ctrl_src[(size_t)env * HB_SZ(nu) + lane];
  float* gstate = P.state + (size_t)env * HB_SZ(nstate);
  auto ld_state = [&](int i) -> float { return gstate[i]; };
  auto st_state = [&](int i, float v) { gstate[i] = v; };
  float time = ld_state(0);
  for (int i = lane; i < nq; i += kGroup) s_qpos[i] = ld_state(1 + i);
  for (int i = lane; i < nv; i += kGroup) { s_qvel[i] = ld_state(1 + nq + i); s_warm[i] = ld_state(1 + nq + nv + i); }
  int status = 0;
  bool eulerdamp = false;
  if (!(M_disableflags & (1 << 14))) {
    bool d = false;
    for (int i = lane; i < nv; i += kGroup) d |= M.dof_damping[i] > 0.f;
    eulerdamp = __any(d);
  }
#ifdef HB_STAMPS
  unsigned long long stamps_[16] = {0};
#endif
  // pad pairs behind the matrix (zero, one: what the dense views read outside the sparsity pattern / beyond nv); never written again
  if (lane < 4) s_qLD[HB_SZ(nM) + (lane >> 1)][lane & 1] = (lane >> 1) == 1 ? 1.f : 0.f;
  // the dof vectors are read 32 wide (dot32): their tails beyond nv stay zero for the whole launch
  if (lane < 32) { s_v0[lane] = 0.f; s_v1[lane] = 0.f; s_v2[lane] = 0.f; }
  gsync();

  // mj_resetData inside a step (mj_check*, mujoco.h:301-307) also zeroes ctrl and xfrc_applied: the rest of that step runs without controls
  bool ctrl_zeroed = false;  // this pass of the step runs on reset data
  bool redo = false;         // this pass is the second mj_forward of a step whose first one gave a bad qacc (mj_checkAcc)
  for (int step = 0; step < nsteps; step++) {
    if constexpr (LEAN != 1) {
      // a launch of several steps whose waves are all resident (hb_api.cpp folds step calls into such launches): the two waves of a SIMD are
      // blocks half a round apart and the SIMD favours the older one, so they take the higher priority in turns (hb_step_duo.hip has the numbers)
      if (nsteps > 1) {
        if (((step & 1) == 0) == ((int)blockIdx.x < ((int)gridDim.x >> 1))) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
    }
    // re-materialise the lane id every step: keeps per-lane table addresses and loads inside the step
    // instead of hoisted out of the rollout loop into long-lived (spilled) registers
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane0) : "memory");
    // Lane l owns body slot l+1 (level order) for the whole step: its record is fetched once, here,
    // and stays in registers through the tree passes (no table loads inside the level loops).
    // (requested first thing in the step: in a one-step launch the records then travel together with the state)
    const bool bl = lane + 1 < nb;
    float4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, bp = q0, bq = q0, ip = q0, ch0 = q0, ch1 = q0;
    float4 JA[3], JB[3], JC[3];
#pragma unroll
    for (int jj = 0; jj < 3; jj++) { JA[jj] = q0; JB[jj] = q0; JC[jj] = q0; }
    if (bl) {
      const float4 HB_CONST* R = M.brec + (size_t)(lane + 1) * kBrecQuads;
      q0 = R[0]; q1 = R[1]; bp = R[2]; bq = R[3]; ip = R[4]; ch0 = R[7]; ch1 = R[8];
#pragma unroll
      for (int jj = 0; jj < 3; jj++) { JA[jj] = R[9 + 3 * jj]; JB[jj] = R[10 + 3 * jj]; JC[jj] = R[11 + 3 * jj]; }
    }
    // likewise what the passes right behind the kinematics read per geom / per dof (at most 64 geoms, 32 dofs: one each per lane)
    int pf_gbody = 0;
    V3 pf_gpos = {0.f, 0.f, 0.f};
    Q4 pf_gquat = {1.f, 0.f, 0.f, 0.f};
    if (lane < HB_SZ(ngeom)) { pf_gbody = M.geom_bodyid[lane]; pf_gpos = ld3(M.geom_pos + 3 * lane); pf_gquat = ldq(M.geom_quat + 4 * lane); }
    float4 pf_dA = {0.f, 0.f, 0.f, 0.f}, pf_dB = pf_dA;
    if (lane < nv) { pf_dA = M.drec[3 * lane]; pf_dB = M.drec[3 * lane + 1]; }
    HB_STAMP(0);
    // ---------------------------------------------------------------- controls
    if (ctrl_mode == 2) {
      int idx = 1 + ctrl_t0 + step + 1000 * (P.env_offset + env);
      for (int i = lane; i < HB_SZ(nu); i += kGroup) s_ctrl[i] = 2.f * halton(idx, i + 2) - 1.f;
    } else {
      const float* c = ((LEAN != 1 && ctrl_mode == 3) ? P.ctrl_tab[step] : ctrl_src + (ctrl_mode == 1 ? (size_t)step * P.n_env * HB_SZ(nu) : 0)) + (size_t)env * HB_SZ(nu);
      if (lane < HB_SZ(nu)) s_ctrl[lane] = ctrl_pf;
      for (int i = lane + kGroup; i < HB_SZ(nu); i += kGroup) s_ctrl[i] = c[i];
      if (ctrl_mode == 1 && step + 1 < nsteps && lane < HB_SZ(nu)) ctrl_pf = c[(size_t)P.n_env * HB_SZ(nu) + lane];
      if (LEAN != 1 && ctrl_mode == 3 && step + 1 < nsteps && lane < HB_SZ(nu)) ctrl_pf = P.ctrl_tab[step + 1][(size_t)env * HB_SZ(nu) + lane];
    }
    // ---------------------------------------------------------------- mj_checkPos / mj_checkVel
    {
      bool badp = false, badv = false;
      for (int i = lane; i < nq; i += kGroup) { float v = s_qpos[i]; badp |= !(fabsf(v) <= HB_MAXVAL); }
      for (int i = lane; i < nv; i += kGroup) { float v = s_qvel[i]; badv |= !(fabsf(v) <= HB_MAXVAL); }
      bool anyp = __any(badp), anyv = __any(badv);
      if (anyp || anyv) {
        status |= anyp ? (1 << 4) : (1 << 5);
        for (int i = lane; i < nq; i += kGroup) s_qpos[i] = M.qpos0[i];
        for (int i = lane; i < nv; i += kGroup) { s_qvel[i] = 0.f; s_warm[i] = 0.f; }
        time = 0.f;
        ctrl_zeroed = true;
        if (P_xfrc) for (int i = lane; i < 6 * nb; i += kGroup) P_xfrc[(size_t)env * nb * 6 + i] = 0.f;
      }
      if (ctrl_zeroed) for (int i = lane; i < HB_SZ(nu); i += kGroup) s_ctrl[i] = 0.f;
    }
    gsync();

    HB_STAMP(1);
    // ---------------------------------------------------------------- mj_kinematics
    // Tree passes read one level-ordered record per body (M.brec: parent, joints, frames, children
    // in kBrecQuads float4s) so that all table loads of a lane are independent of each other.
    if (lane == 0) {
      st3(s_xpq, {0.f, 0.f, 0.f}); stq(s_xpq + 4, {1.f, 0.f, 0.f, 0.f}); st3(s_xipos, {0.f, 0.f, 0.f});
      for (int i = 0; i < 9; i++) s_xmat[i] = (i % 4 == 0) ? 1.f : 0.f;
    }
    gsync();
    const int myb = __float_as_int(q0.x), myp = __float_as_int(q0.y), myjn = __float_as_int(q0.z), myja = __float_as_int(q0.w);
    const int mylevel = bl ? (__float_as_int(q1.x) & 255) : -1, mycn = __float_as_int(q1.w);
    const int myanc2 = (__float_as_int(q1.x) >> 8) & 255, myanc4 = (__float_as_int(q1.x) >> 16) & 255, myanc8 = (__float_as_int(q1.x) >> 24) & 255;
    const float mymass = (dr && bl) ? dr[DL.o_mass + myb] : q1.z;
    const int mych[8] = {__float_as_int(ch0.x), __float_as_int(ch0.y), __float_as_int(ch0.z), __float_as_int(ch0.w),
                         __float_as_int(ch1.x), __float_as_int(ch1.y), __float_as_int(ch1.z), __float_as_int(ch1.w)};
    // The pose of a body relative to its parent (body frame, then its joints in order) does not depend on
    // the parent's world pose, so all of it - the sin/cos of every joint included - is done here for every
    // body at once; the level loop that follows only composes parent and local pose (mj_kinematics does the
    // same products in world coordinates, body by body).
    const bool isfree = bl && myjn == 1 && __float_as_int(JA[0].x) == 0;
    V3 posl = {bp.x, bp.y, bp.z};
    Q4 quatl = {bq.x, bq.y, bq.z, bq.w};
    V3 axl[3], ancl[3];  // joint axes and anchors in the parent frame
#pragma unroll
    for (int jj = 0; jj < 3; jj++) { axl[jj] = {0.f, 0.f, 0.f}; ancl[jj] = {0.f, 0.f, 0.f}; }
    if (isfree) {
      const int qa = __float_as_int(JA[0].y);
      posl = ld3(s_qpos + qa);
      quatl = qnormalize(ldq(s_qpos + qa + 3));
    } else if (bl) {
#pragma unroll
      for (int jj = 0; jj < 3; jj++) {
        if (jj < myjn) {
          const int qa = __float_as_int(JA[jj].y);
          const V3 laxis = {JB[jj].x, JB[jj].y, JB[jj].z}, lpos = {JC[jj].x, JC[jj].y, JC[jj].z};
          axl[jj] = qrot(quatl, laxis);
          ancl[jj] = qrot(quatl, lpos) + posl;
          const float dq = s_qpos[qa] - JA[jj].w;
          if (__float_as_int(JA[jj].x) == 2) posl = posl + axl[jj] * dq;
          else {
            quatl = qmul(quatl, axisangle(laxis, dq));
            posl = ancl[jj] - qrot(quatl, lpos);
          }
        }
      }
    }
    // World poses by pointer jumping: every body starts with its pose relative to its parent and, in round r, composes
    // it with the (partially composed) pose of its ancestor 2^r links up; after ceil(log2(depth)) rounds it is the world
    // pose.  The world body holds the identity and is every short chain's fixed point.  (mj_kinematics composes the same
    // transforms root to leaf; the association differs, the product does not.)
    V3 mypos = posl;
    Q4 myquat = quatl;
    if (bl) {
      reinterpret_cast<float4*>(s_xpq + kXpqStride * myb)[0] = {mypos.x, mypos.y, mypos.z, 0.f};
      reinterpret_cast<float4*>(s_xpq + kXpqStride * myb)[1] = {myquat.w, myquat.x, myquat.y, myquat.z};
    }
    gsync();
    for (int r = 0, span = 1; span < HB_SZ(nlevel) - 1 || r == 0; r++, span <<= 1) {
      const int anc = r == 0 ? myp : (r == 1 ? myanc2 : (r == 2 ? myanc4 : myanc8));
      float4 pp4 = {0.f, 0.f, 0.f, 0.f}, pq4 = {1.f, 0.f, 0.f, 0.f};
      if (bl) { const float4* Pp = reinterpret_cast<const float4*>(s_xpq + kXpqStride * anc); pp4 = Pp[0]; pq4 = Pp[1]; }
      gsync();  // every lane has read its ancestor before anyone overwrites a pose
      if (bl && anc != 0) {
        const Q4 pq = {pq4.x, pq4.y, pq4.z, pq4.w};
        mypos = V3{pp4.x, pp4.y, pp4.z} + qrot(pq, mypos);
        myquat = qnormalize(qmul(pq, myquat));
        reinterpret_cast<float4*>(s_xpq + kXpqStride * myb)[0] = {mypos.x, mypos.y, mypos.z, 0.f};
        reinterpret_cast<float4*>(s_xpq + kXpqStride * myb)[1] = {myquat.w, myquat.x, myquat.y, myquat.z};
      }
      gsync();
    }
    if (bl) {  // everything that hangs off the world poses, all bodies at once
      if (isfree) {
        st3(s_xanchor + 3 * myja, mypos);
        st3(s_xaxis + 3 * myja, {JB[0].x, JB[0].y, JB[0].z});
      } else {
        const Q4 pq = ldq(s_xpq + kXpqStride * myp + 4);
        const V3 pp = ld3(s_xpq + kXpqStride * myp);
#pragma unroll
        for (int jj = 0; jj < 3; jj++) {
          if (jj < myjn) {
            st3(s_xaxis + 3 * (myja + jj), qrot(pq, axl[jj]));
            st3(s_xanchor + 3 * (myja + jj), qrot(pq, ancl[jj]) + pp);
          }
        }
      }
      float mat[9];
      q2mat(mat, myquat);
      for (int i = 0; i < 9; i++) s_xmat[9 * myb + i] = mat[i];
      st3(s_xipos + 3 * myb, mypos + mrot(mat, {ip.x, ip.y, ip.z}));
    }
    gsync();
    HB_STAMP(2);
    // geoms: world position and z axis
    if (lane < HB_SZ(ngeom)) {
      const int g = lane, b = pf_gbody;
      st3(s_gpos + 3 * g, ld3(s_xpq + kXpqStride * b) + mrot(s_xmat + 9 * b, pf_gpos));
      Q4 q = qmul(ldq(s_xpq + kXpqStride * b + 4), pf_gquat);
      st3(s_gaxis + 3 * g, {2.f * (q.x * q.z + q.w * q.y), 2.f * (q.y * q.z - q.w * q.x), q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z});
      if constexpr (COLL != 0) stq(lds + HB_SZ(o_gquat) + 4 * g, q);
    }
    // ---------------------------------------------------------------- mj_comPos
    for (int t = 0; t < HB_SZ(ntree); t++) {
      V3 acc = {0.f, 0.f, 0.f};
      if (bl && __float_as_int(q1.y) == t) acc = ld3(s_xipos + 3 * myb) * mymass;
      float im = M.tree_invmass[t];
      float sx = wave_sum(acc.x) * im, sy = wave_sum(acc.y) * im, sz = wave_sum(acc.z) * im;
      if (lane == 0) st3(s_scom + 3 * t, {sx, sy, sz});
    }
    gsync();
    if (bl) {
      const float4 HB_CONST* R = M.brec + (size_t)(lane + 1) * kBrecQuads;
      const float4 iq = R[5], in4 = R[6];
      const int b = myb;
      V3 com = ld3(s_scom + 3 * __float_as_int(q1.y));
      V3 dif = ld3(s_xipos + 3 * b) - com;
      float mat[9];
      q2mat(mat, qmul(ldq(s_xpq + kXpqStride * b + 4), {iq.x, iq.y, iq.z, iq.w}));
      const float in0 = in4.x, in1 = in4.y, in2 = in4.z, mass = mymass;
      float t[9];
      for (int r = 0; r < 3; r++) { t[3 * r] = mat[3 * r] * in0; t[3 * r + 1] = mat[3 * r + 1] * in1; t[3 * r + 2] = mat[3 * r + 2] * in2; }
      float* res = s_cinert + 10 * b;
      res[0] = t[0] * mat[0] + t[1] * mat[1] + t[2] * mat[2] + mass * (dif.y * dif.y + dif.z * dif.z);
      res[1] = t[3] * mat[3] + t[4] * mat[4] + t[5] * mat[5] + mass * (dif.x * dif.x + dif.z * dif.z);
      res[2] = t[6] * mat[6] + t[7] * mat[7] + t[8] * mat[8] + mass * (dif.x * dif.x + dif.y * dif.y);
      res[3] = t[0] * mat[3] + t[1] * mat[4] + t[2] * mat[5] - mass * dif.x * dif.y;
      res[4] = t[0] * mat[6] + t[1] * mat[7] + t[2] * mat[8] - mass * dif.x * dif.z;
      res[5] = t[3] * mat[6] + t[4] * mat[7] + t[5] * mat[8] - mass * dif.y * dif.z;
      res[6] = mass * dif.x; res[7] = mass * dif.y; res[8] = mass * dif.z; res[9] = mass;
    }
    if (lane < 10) s_cinert[lane] = 0.f;
    if (lane < nv) {
      const int d = lane;
      const float4 dA = pf_dA, dB = pf_dB;
      const int j = __float_as_int(dA.x), b = __float_as_int(dA.y), type = __float_as_int(dA.z), k = __float_as_int(dA.w);
      V3 off = ld3(s_scom + 3 * __float_as_int(dB.x)) - ld3(s_xanchor + 3 * j);
      V3 ang = {0.f, 0.f, 0.f}, lin = {0.f, 0.f, 0.f};
      if (type == 0) {
        if (k < 3) { lin = {k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f}; }
        else {
          int c = k - 3;
          ang = {s_xmat[9 * b + c], s_xmat[9 * b + 3 + c], s_xmat[9 * b + 6 + c]};
          lin = cross(ang, off);
        }
      } else if (type == 2) {
        lin = ld3(s_xaxis + 3 * j);
      } else {
        ang = ld3(s_xaxis + 3 * j);
        lin = cross(ang, off);
      }
      reinterpret_cast<float4*>(s_cdof + kCdofStride * d)[0] = {ang.x, ang.y, ang.z, 0.f};
      reinterpret_cast<float4*>(s_cdof + kCdofStride * d)[1] = {lin.x, lin.y, lin.z, 0.f};
    }
    // fixed tendon lengths
    for (int t = lane; t < HB_SZ(ntendon); t += kGroup) {
      float len = 0.f;
      // the first four wraps come in one record (no dependent table walk); longer tendons finish from the wrap tables
      const float4 tc = M.trec[3 * t], tq = M.trec[3 * t + 1];
      len = tc.x * s_qpos[__float_as_int(tq.x)] + tc.y * s_qpos[__float_as_int(tq.y)] + tc.z * s_qpos[__float_as_int(tq.z)] + tc.w * s_qpos[__float_as_int(tq.w)];
      const int nw = M.tendon_num[t];
      for (int w = 4; w < nw; w++) len += M.wrap_prm[M.tendon_adr[t] + w] * s_qpos[M.wrap_qposadr[M.tendon_adr[t] + w]];
      s_tenlen[t] = len;
    }
    gsync();
    HB_STAMP(3);
    // ---------------------------------------------------------------- mj_comVel + mj_rne forward pass
    // A body's velocity is its parent's plus a local term lv = sum_j cdof_j qvel_j, and its bias
    // acceleration is the parent's plus sum_j (cvel before dof j) x cdof_j qvel_j.  The cross product is
    // linear in both arguments, so that sum splits into  cvel_parent x lv  plus a purely local part la
    // (the partial sums of this body's own dofs); lv and la are formed for all bodies at once and the
    // level loop is reduced to two 6-vector updates per body.
    float lv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, la[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bl) {
      float cd[6], t[6];
      if (isfree) {
        const int da = __float_as_int(JA[0].z);
        for (int k = 0; k < 3; k++) {
          const float qv = s_qvel[da + k];
          ld_cdof(s_cdof, da + k, cd);
          for (int i = 0; i < 6; i++) lv[i] += cd[i] * qv;
        }
        // the three rotational dofs all see the velocity after the translational ones (mj_comVel, free joint)
        float rot[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 3; k++) {
          const float qv = s_qvel[da + 3 + k];
          ld_cdof(s_cdof, da + 3 + k, cd);
          cross_motion(t, lv, cd);
          for (int i = 0; i < 6; i++) { la[i] += t[i] * qv; rot[i] += cd[i] * qv; }
        }
        for (int i = 0; i < 6; i++) lv[i] += rot[i];
      } else {
#pragma unroll
        for (int jj = 0; jj < 3; jj++) {
          if (jj < myjn) {
            const int da = __float_as_int(JA[jj].z);
            const float qv = s_qvel[da];
            ld_cdof(s_cdof, da, cd);
            cross_motion(t, lv, cd);
            for (int i = 0; i < 6; i++) { la[i] += t[i] * qv; lv[i] += cd[i] * qv; }
          }
        }
      }
    }
    // Tree-pass storage is one record per body, 16-byte aligned, moved as ds_read/write_b128:
    //   s_va[12 b ..] = cvel[6] | cacc[6]          s_if[16 b ..] = composite inertia[10] | force[6]
    // Velocities and bias accelerations along each chain, also by pointer jumping.  A chain segment carries
    //   V = sum of its bodies' lv,   A = sum of their la + sum over ordered pairs (a' above a) of lv_a' x lv_a,
    // and an upper segment U composes with the lower one L as  V = V_U + V_L,  A = A_U + A_L + V_U x V_L  (the motion
    // cross product is bilinear, so the cross terms of all pairs (a' in U, a in L) collapse into one product).  This is
    // associative: after ceil(log2(depth)) rounds V is the body's cvel and A its cacc minus the world's (-gravity).
    float mycvel[6], mycacc[6];
    for (int i = 0; i < 6; i++) { mycvel[i] = lv[i]; mycacc[i] = la[i]; }
    if (bl) {
      float4* Op = reinterpret_cast<float4*>(s_va + 12 * myb);
      Op[0] = {mycvel[0], mycvel[1], mycvel[2], mycvel[3]};
      Op[1] = {mycvel[4], mycvel[5], mycacc[0], mycacc[1]};
      Op[2] = {mycacc[2], mycacc[3], mycacc[4], mycacc[5]};
    }
    gsync();
    for (int r = 0, span = 1; span < HB_SZ(nlevel) - 1 || r == 0; r++, span <<= 1) {
      const int anc = r == 0 ? myp : (r == 1 ? myanc2 : (r == 2 ? myanc4 : myanc8));
      float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
      if (bl && anc != 0) { const float4* Pp = reinterpret_cast<const float4*>(s_va + 12 * anc); a0 = Pp[0]; a1 = Pp[1]; a2 = Pp[2]; }
      gsync();  // every lane has read its ancestor's segment before anyone overwrites one
      if (bl && anc != 0) {
        const float uv[6] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y};
        const float ua[6] = {a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
        float t[6];
        cross_motion(t, uv, mycvel);
        for (int i = 0; i < 6; i++) { mycacc[i] += ua[i] + t[i]; mycvel[i] += uv[i]; }
        float4* Op = reinterpret_cast<float4*>(s_va + 12 * myb);
        Op[0] = {mycvel[0], mycvel[1], mycvel[2], mycvel[3]};
        Op[1] = {mycvel[4], mycvel[5], mycacc[0], mycacc[1]};
        Op[2] = {mycacc[2], mycacc[3], mycacc[4], mycacc[5]};
      }
      gsync();
    }
    if (!(M_disableflags & (1 << 6))) for (int i = 0; i < 3; i++) mycacc[3 + i] -= M.gravity[i];  // the world's cacc
    // sensor read-out for planner residuals (mj_sensorPos/Vel of framepos, subtreecom, subtreelinvel)
    if (P_sensor_out) {
      float* so = P_sensor_out + ((size_t)step * P.n_env + env) * P.sensor_stride;
      for (int k = 0; k < P.sensor_nframe; k++) {
        const int sb = P.sensor_body[k];
        const V3 w = ld3(s_xpq + kXpqStride * sb) + qrot(ldq(s_xpq + kXpqStride * sb + 4), {P.sensor_off[k][0], P.sensor_off[k][1], P.sensor_off[k][2]});  // site = body frame + offset
        if (lane < 3) so[3 * k + lane] = lane == 0 ? w.x : (lane == 1 ? w.y : w.z);
      }
      {
        int o = 3 * P.sensor_nframe + (P.sensor_tree >= 0 ? 6 : 0);
        for (int k = 0; k < P.sensor_naxis; k++, o += 3) {  // framexaxis / framezaxis: a column of the body's rotation
          const V3 e = P.sensor_axis_which[k] == 0 ? V3{1.f, 0.f, 0.f} : V3{0.f, 0.f, 1.f};
          const V3 a = qrot(ldq(s_xpq + kXpqStride * P.sensor_axis_body[k] + 4), e);
          if (lane < 3) so[o + lane] = lane == 0 ? a.x : (lane == 1 ? a.y : a.z);
        }
        for (int k = 0; k < P.sensor_nlinvel; k++, o += 3) {  // mj_objectVelocity at the inertial frame origin, world axes
          const int sb = P.sensor_linvel_body[k];
          const float* cv = s_va + 12 * sb;
          const V3 v = V3{cv[3], cv[4], cv[5]} + cross(V3{cv[0], cv[1], cv[2]}, ld3(s_xipos + 3 * sb) - ld3(s_scom + 3 * M.body_treeid[sb]));
          if (lane < 3) so[o + lane] = lane == 0 ? v.x : (lane == 1 ? v.y : v.z);
        }
        for (int k = 0; k < P.sensor_nsub; k++, o += 3) {  // mj_subtreeVel of a body's subtree: its linear momentum over its mass
          V3 mom = {0.f, 0.f, 0.f};
          if (bl && ((P.sensor_submask[k] >> myb) & 1ull)) {
            const V3 ang = {mycvel[0], mycvel[1], mycvel[2]}, lin = {mycvel[3], mycvel[4], mycvel[5]};
            mom = (lin + cross(ang, ld3(s_xipos + 3 * myb) - ld3(s_scom + 3 * __float_as_int(q1.y)))) * mymass;
          }
          const float im = P.sensor_subinv[k];
          const float vx = wave_sum(mom.x) * im, vy = wave_sum(mom.y) * im, vz = wave_sum(mom.z) * im;
          if (lane < 3) so[o + lane] = lane == 0 ? vx : (lane == 1 ? vy : vz);
        }
        if (P.sensor_flags & 4) { for (int i = lane; i < nq; i += kGroup) so[o + i] = s_qpos[i]; o += nq; }
        if (P.sensor_flags & 1) { for (int i = lane; i < nv; i += kGroup) so[o + i] = s_qvel[i]; o += nv; }
        if (P.sensor_flags & 2) for (int i = lane; i < HB_SZ(nu); i += kGroup) so[o + i] = s_ctrl[i];
      }
      if (P.sensor_tree >= 0) {
        const int t = P.sensor_tree;
        const V3 com = ld3(s_scom + 3 * t);
        // mj_subtreeVel for a whole tree: linear momentum over mass; a body's com moves with lin + ang x (xipos - com)
        V3 mom = {0.f, 0.f, 0.f};
        if (bl && __float_as_int(q1.y) == t) {
          const V3 ang = {mycvel[0], mycvel[1], mycvel[2]}, lin = {mycvel[3], mycvel[4], mycvel[5]};
          mom = (lin + cross(ang, ld3(s_xipos + 3 * myb) - com)) * mymass;
        }
        const float im = M.tree_invmass[t];
        const float vx = wave_sum(mom.x) * im, vy = wave_sum(mom.y) * im, vz = wave_sum(mom.z) * im;
        if (lane < 3) {
          so[3 * P.sensor_nframe + lane] = lane == 0 ? com.x : (lane == 1 ? com.y : com.z);
          so[3 * P.sensor_nframe + 3 + lane] = lane == 0 ? vx : (lane == 1 ? vy : vz);
        }
      }
    }
    // body-local force cinert cacc + cvel x* (cinert cvel), and the composite inertia seeds, all bodies at once
    if (bl) {
      float in[10], f0[6], f1[6], f2[6];
      for (int i = 0; i < 10; i++) in[i] = s_cinert[10 * myb + i];
      mul_inert_vec(f0, in, mycacc);
      mul_inert_vec(f1, in, mycvel);
      cross_force(f2, mycvel, f1);
      float4* Op = reinterpret_cast<float4*>(s_if + kIfStride * myb);
      Op[0] = {in[0], in[1], in[2], in[3]};
      Op[1] = {in[4], in[5], in[6], in[7]};
      Op[2] = {in[8], in[9], f0[0] + f2[0], f0[1] + f2[1]};
      Op[3] = {f0[2] + f2[2], f0[3] + f2[3], f0[4] + f2[4], f0[5] + f2[5]};
    }
    if (lane < 16) s_if[lane] = lane < 10 ? s_cinert[lane] : 0.f;  // world body
    // first round of mass-matrix entry words, requested ahead of the sweep that produces what they index
    int pf_pk = 0;
    float2 pf_ad = {0.f, 0.f};
    if (lane < HB_SZ(nM)) { pf_pk = M.mrec[lane]; pf_ad = M.mdiag[lane]; }
    gsync();
    // mj_crb and the mj_rne backward pass share one sweep up the tree: children into parents (pull form)
    for (int L = HB_SZ(nlevel) - 2; L >= 1; L--) {
      if (mylevel == L && mycn > 0) {
        float4* Op = reinterpret_cast<float4*>(s_if + kIfStride * myb);
        float4 acc[4] = {Op[0], Op[1], Op[2], Op[3]};
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (k < mycn) {
            const float4* Cp = reinterpret_cast<const float4*>(s_if + kIfStride * mych[k]);
#pragma unroll
            for (int q = 0; q < 4; q++) { const float4 c = Cp[q]; acc[q].x += c.x; acc[q].y += c.y; acc[q].z += c.z; acc[q].w += c.w; }
          }
#pragma unroll
        for (int q = 0; q < 4; q++) Op[q] = acc[q];
      }
      gsync();
    }
    HB_STAMP(4);
    // ---------------------------------------------------------------- qM from the composite inertias
    // dof records of the bias pass and the actuator records: requested here, used behind the mass matrix
    float4 pf_bA = {0.f, 0.f, 0.f, 0.f}, pf_bB = pf_bA, pf_bC = pf_bA, pf_a0 = pf_bA, pf_a1 = pf_bA, pf_a2 = pf_bA, pf_a3 = pf_bA;
    if (lane < nv) { pf_bA = M.drec[3 * lane]; pf_bB = M.drec[3 * lane + 1]; pf_bC = M.drec[3 * lane + 2]; }
    if (lane < HB_SZ(nu)) { const float4 HB_CONST* AR4 = M.arec + (size_t)lane * 4; pf_a0 = AR4[0]; pf_a1 = AR4[1]; pf_a2 = AR4[2]; pf_a3 = AR4[3]; }
    for (int e = lane; e < HB_SZ(nM); e += kGroup) {
      const int pk = pf_pk;  // i | j << 8 | body(i) << 16
      const float2 ad = pf_ad;  // (armature, damping) on diagonal entries, 0 elsewhere
      if (e + kGroup < HB_SZ(nM)) { pf_pk = M.mrec[e + kGroup]; pf_ad = M.mdiag[e + kGroup]; }  // the next round's, one ahead
      const int i = pk & 255, j = (pk >> 8) & 255, bi = pk >> 16;
      float buf[6], cd[6];
      ld_cdof(s_cdof, i, cd);
      float in[10];
      {
        const float4* Ip = reinterpret_cast<const float4*>(s_if + kIfStride * bi);
        const float4 i0 = Ip[0], i1 = Ip[1], i2 = Ip[2];
        in[0] = i0.x; in[1] = i0.y; in[2] = i0.z; in[3] = i0.w; in[4] = i1.x; in[5] = i1.y; in[6] = i1.z; in[7] = i1.w; in[8] = i2.x; in[9] = i2.y;
      }
      mul_inert_vec(buf, in, cd);
      float sacc = 0.f;
      float cj[6];
      ld_cdof(s_cdof, j, cj);
      for (int t = 0; t < 6; t++) sacc += cj[t] * buf[t];
      sacc += (dr && i == j) ? dr[DL.o_arm + i] : ad.x;
      // H = M + h diag(damping): matrix of the implicit-damping Euler solve (mj_Euler), kept beside M
      s_qLD[e] = {sacc, sacc + (eulerdamp ? M.timestep * ad.y : 0.f)};
    }
    gsync();
    HB_STAMP(5);
    // (mj_factorM: the assembled sparse matrix stays as it is; both solvers eliminate a dense view of it on the matrix cores later)
    HB_STAMP(6);
    // ---------------------------------------------------------------- qfrc_bias, mj_passive, mj_fwdActuation -> qfrc_smooth
    if (lane < nv) {
      const int d = lane;
      const float4 dA = pf_bA, dB = pf_bB, dC = pf_bC;
      float bias = 0.f;
      const int b = __float_as_int(dA.y);
      float cdd[6];
      ld_cdof(s_cdof, d, cdd);
      for (int t = 0; t < 6; t++) bias += cdd[t] * s_if[kIfStride * b + 10 + t];
      float passive = 0.f;
      if (!(M_disableflags & (1 << 5))) {
        if (__float_as_int(dA.z) >= 2) passive -= (dr ? dr[DL.o_stiff + d] : dB.w) * (s_qpos[__float_as_int(dC.x)] - dC.y);
        passive -= dB.z * s_qvel[d];
      }
      s_smooth[d] = passive - bias;
    }
    gsync();
    if (!(M_disableflags & (1 << 10))) {
      for (int a = lane; a < HB_SZ(nu); a += kGroup) {
        if (a != lane) { const float4 HB_CONST* AR4 = M.arec + (size_t)a * 4; pf_a0 = AR4[0]; pf_a1 = AR4[1]; pf_a2 = AR4[2]; pf_a3 = AR4[3]; }  // (more than 64 actuators)
        const int qa = __float_as_int(pf_a0.z), da = __float_as_int(pf_a0.w);
        float ctrl = s_ctrl[a];
        if (__float_as_int(pf_a0.x) && !(M_disableflags & (1 << 7))) ctrl = clampf(ctrl, pf_a1.x, pf_a1.y);
        float gear = pf_a1.z;
        const float gain = dr ? dr[DL.o_gain + a] : pf_a1.w, bias1 = dr ? dr[DL.o_bias1 + a] : pf_a2.y;
        float force = gain * ctrl + pf_a2.x + bias1 * gear * s_qpos[qa] + pf_a2.z * gear * s_qvel[da];
        if (__float_as_int(pf_a0.y)) force = clampf(force, dr ? dr[DL.o_frc + 2 * a] : pf_a3.x, dr ? dr[DL.o_frc + 2 * a + 1] : pf_a3.y);
        atomicAdd(&s_smooth[da], gear * force);
      }
    }
    gsync();
    // xfrc_applied: Cartesian wrench at each body com (mj_xfrcAccumulate)
    if (P_xfrc) {
      if (P_xfrc_scale > 0.f && P_integrate && !P.stage.rerun) {  // (a plain mj_forward - hb_forward, the terminal read-out - leaves the process where it is;
        // the rerun of a deferred env-step finds the process already advanced by the fast pass)
        // Trajectory::NoisyRollout's perturbation (trajectory.cc:147-156): Ornstein-Uhlenbeck noise on every xfrc_applied entry
        float* xw = P_xfrc + (size_t)env * nb * 6;
        for (int i = lane; i < 6 * nb; i += kGroup)
          xw[i] = P_xfrc_rate * xw[i] + P_xfrc_scale * rng_normal(P_xfrc_seed, P.env_offset + env, P_xfrc_call, P.t0 + step, RS_XFRC, i);
        gsync();
      }
      const float* xf = P_xfrc + (size_t)env * nb * 6;
      for (int b = 1; b < nb; b++) {
        float f[6];
        bool nz = false;
        for (int i = 0; i < 6; i++) { f[i] = xf[6 * b + i]; nz |= (f[i] != 0.f); }
        if (!nz) continue;  // uniform: every lane reads the same values
        V3 off = ld3(s_xipos + 3 * b) - ld3(s_scom + 3 * M.body_treeid[b]);
        unsigned long long mask = M.body_dofmask[b];
        for (int d = lane; d < nv; d += kGroup) {
          if (!((mask >> d) & 1ull)) continue;
          float cdd[6];
          ld_cdof(s_cdof, d, cdd);
          V3 ang = {cdd[0], cdd[1], cdd[2]}, lin = {cdd[3], cdd[4], cdd[5]};
          V3 jp = lin + cross(ang, off);
          s_smooth[d] += jp.x * f[0] + jp.y * f[1] + jp.z * f[2] + ang.x * f[3] + ang.y * f[4] + ang.z * f[5];
        }
      }
      gsync();
    }

    // ================================================================ region B from here on (aliases the dynamics scratch)
    HB_STAMP(7);
    // ---------------------------------------------------------------- mj_collision
    int ncon = 0;
    const bool contacts_on = !(M_disableflags & ((1 << 0) | (1 << 4)));
    if constexpr (COLL != 0) {
      if (contacts_on) {
        // staged step: the narrowphase ran in its own kernel on this step's poses (launch_step); the second forward pass of a step
        // whose first one was reset (mj_checkAcc) runs on other poses and does its own
        if (DEFER != 0 || (P.stage.result && !redo)) ncon = collide_gather<kNC>(M, lane, env, P.stage, s_gaxis, s_con, status);  // (DEFER: always staged, never a second pass)
        else ncon = collide_general<kNC>(M, dr ? dr + DL.o_hfield : (const float*)M.hfield_data, lane, s_gpos, s_gaxis, s_gquat, s_con, reinterpret_cast<int*>(s_C), status);
      }
    } else if (contacts_on) {
      // Two passes.  (1) Broadphase over every candidate pair - bounding spheres, or distance to the plane - with the
      // survivors compacted, IN PAIR ORDER, into a list (ballot + popcount; the list borrows the head of C, which is not
      // written before makeConstraint).  (2) Narrowphase over the list, 64 survivors per round: for the humanoid that is
      // one round instead of three, and the contact order (= pair order) is what it was.  One 3-quad record per pair;
      // pass 1 reads two of them and has the next round's in flight.
      int* s_list = reinterpret_cast<int*>(s_C);
      int nlist = 0;
      {
        float4 n0 = M.crec[3 * (size_t)lane], n1 = M.crec[3 * (size_t)lane + 1];
        for (int p0 = 0; p0 < HB_SZ(npair); p0 += kGroup) {
          const int p = p0 + lane;
          const float4 c0 = n0, c1 = n1;
          if (p0 + kGroup < HB_SZ(npair)) { const float4 HB_CONST* N = M.crec + 3 * (size_t)(p + kGroup); n0 = N[0]; n1 = N[1]; }
          bool pass = false;
          if (p < HB_SZ(npair)) {
            const int g1 = __float_as_int(c0.x), g2 = __float_as_int(c0.y), t1 = __float_as_int(c0.z) & 255;
            const V3 dp = ld3(s_gpos + 3 * g2) - ld3(s_gpos + 3 * g1);
            if (t1 == 0) pass = dot(dp, ld3(s_gaxis + 3 * g1)) <= c0.w + c1.y;
            else { const float bound = c1.x + c1.y + c0.w; pass = dot(dp, dp) <= bound * bound; }
          }
          const unsigned long long bal = __ballot(pass);
          if (pass) s_list[nlist + __popcll(bal & ((1ull << lane) - 1ull))] = p;
          nlist += __popcll(bal);
        }
      }
      nlist = uniform(nlist);
      gsync();
      for (int i0 = 0; i0 < nlist; i0 += kGroup) {
        const bool have = i0 + lane < nlist;
        const int p = have ? s_list[i0 + lane] : 0;
        float4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
        if (have) { const float4 HB_CONST* N = M.crec + 3 * (size_t)p; c0 = N[0]; c1 = N[1]; c2 = N[2]; }
        ConOut co0, co1;
        int n = 0;
        V3 hint = {0.f, 0.f, 0.f};
        float margin = 0.f;
        if (have) {
          const int g1 = __float_as_int(c0.x), g2 = __float_as_int(c0.y);
          const int t1 = __float_as_int(c0.z) & 255, t2 = __float_as_int(c0.z) >> 8;
          margin = c0.w;
          const float rb1 = c1.x, rb2 = c1.y;
          V3 pos1 = ld3(s_gpos + 3 * g1), pos2 = ld3(s_gpos + 3 * g2), ax2 = ld3(s_gaxis + 3 * g2);
          float r2 = c2.x, l2 = c2.y;
          if (t1 == 0) {
            V3 normal = ld3(s_gaxis + 3 * g1);
            if (dot(pos2 - pos1, normal) <= margin + rb2) {
              if (t2 == 2) n = plane_sphere(co0, margin, pos1, normal, pos2, r2) ? 1 : 0;
              else {
                ConOut ca, cb;
                const bool h1 = plane_sphere(ca, margin, pos1, normal, pos2 + ax2 * l2, r2);
                const bool h2 = plane_sphere(cb, margin, pos1, normal, pos2 - ax2 * l2, r2);
                co0 = h1 ? ca : cb;
                co1 = cb;
                n = (h1 ? 1 : 0) + (h2 ? 1 : 0);
                hint = ax2;
              }
            }
          } else if (t1 >= 2) {
            V3 dp = pos2 - pos1;
            float bound = rb1 + rb2 + margin;
            if (dot(dp, dp) <= bound * bound) {
              float r1 = c1.z, l1 = c1.w;
              if (t1 == 2 && t2 == 2) n = sphere_sphere(co0, margin, pos1, r1, pos2, r2) ? 1 : 0;
              else if (t1 == 2) {
                float x = clampf(dot(ax2, pos1 - pos2), -l2, l2);
                n = sphere_sphere(co0, margin, pos1, r1, pos2 + ax2 * x, r2) ? 1 : 0;
              } else {
                n = capsule_capsule(co0, co1, margin, pos1, ld3(s_gaxis + 3 * g1), r1, l1, pos2, ax2, r2, l2);
              }
            }
          }
        }
        // ordered append: slot = ncon + (# contacts of lower lanes)
        unsigned long long b1 = __ballot(n >= 1), b2 = __ballot(n >= 2);
        unsigned long long lt = (1ull << lane) - 1ull;
        int slot = ncon + __popcll(b1 & lt) + __popcll(b2 & lt);
        if (n >= 1 && slot < kNC) {
          float* c = s_con + slot * kConStride;
          c[C_DIST] = co0.dist;
          st3(c + C_POS, co0.pos);
          make_frame(c + C_FRAME, co0.n, hint);
          c[C_PAIR] = __int_as_float(p);
        }
        if (n >= 2 && slot + 1 < kNC) {
          float* c = s_con + (slot + 1) * kConStride;
          c[C_DIST] = co1.dist;
          st3(c + C_POS, co1.pos);
          make_frame(c + C_FRAME, co1.n, hint);
          c[C_PAIR] = __int_as_float(p);
        }
        ncon += __popcll(b1) + __popcll(b2);
      }
      if (ncon > kNC) {
        status |= (1 << 1); ncon = kNC;
      }
    }
    ncon = uniform(ncon);
    // self collision (CPUEnv._check_self_collision, cpu_env.py:576-584): a contact whose geoms both belong to the robot
    // (one table fetch, requested here and collected behind the limit rows)
    bool selfc = false;
    if (lane < ncon) selfc = M.pair_self[__float_as_int(s_con[lane * kConStride + C_PAIR])] != 0;
    gsync();

    HB_STAMP(8);
    // ---------------------------------------------------------------- mj_makeConstraint
    int nefc = 0;
    const bool constraints_on = !(M_disableflags & (1 << 0));
    const int selfcol = __any(selfc) ? 1 : 0;
    if constexpr (COLL != 0) {
      // ---- general form: limits, then contacts of dimension 1 / 3 / 4 / 6 (one row, or 2 (dim - 1) pyramid rows); what the solver
      // needs of a row besides its Jacobian is written once, here: s_meta[row] = (R, K imp (pos - margin), B, -)
      if (constraints_on && !(M_disableflags & (1 << 3))) {
        for (int c0 = 0; c0 < HB_SZ(nlimcand); c0 += kGroup) {
          const int c = c0 + lane;
          bool active = false;
          float dist = 0.f, margin = 0.f;
          int side = 0, kind = 0, id = 0;
          float4 l0 = {0.f, 0.f, 0.f, 0.f}, l1 = l0, l2 = l0, l3 = l0;
          if (c < HB_SZ(nlimcand)) {
            const float4 HB_CONST* LR = M.lrec + (size_t)c * 4;
            l0 = LR[0]; l1 = LR[1]; l2 = LR[2]; l3 = LR[3];
            kind = __float_as_int(l0.x); id = __float_as_int(l0.y); side = __float_as_int(l0.z);
            margin = dr ? dr[DL.o_lmargin + c] : l1.x;
            const float value = kind == 0 ? s_qpos[__float_as_int(l0.w)] : s_tenlen[id];
            dist = (float)side * ((dr ? dr[DL.o_lrange + c] : l1.y) - value);
            active = dist < margin;
          }
          const unsigned long long bal = __ballot(active);
          const int row = nefc + __popcll(bal & ((1ull << lane) - 1ull));
          if (active && row < kNR) {
            float* Jr = s_C + row * cs;
            for (int k = 0; k < cs; k++) Jr[k] = 0.f;
            if (kind == 0) Jr[__float_as_int(l3.z)] = (float)(-side);
            else for (int w = 0; w < M.tendon_num[id]; w++) Jr[M.wrap_dofadr[M.tendon_adr[id] + w]] = (float)(-side) * M.wrap_prm[M.tendon_adr[id] + w];
            const float solimp[5] = {l2.x, l2.y, l2.z, l2.w, l3.x};
            const float imp = clampf(impedance(solimp, dist, margin), HB_MINIMP, HB_MAXIMP);
            float K, Bc;
            kb_from_solref(l1.z, l1.w, solimp[1], M.timestep, !(M_disableflags & (1 << 11)), K, Bc);
            float* e = s_meta + kMetaStride * row;
            e[0] = fmaxf(HB_MINVAL, (1.f - imp) * l3.y / imp); e[1] = K * imp * (dist - margin); e[2] = Bc;
          }
          nefc += __popcll(bal);
        }
        if (nefc > kNR) { status |= (1 << 2); nefc = kNR; }
      }
      if (constraints_on && contacts_on) {
        int myrows = 0, pairid = 0, dim = 1;
        bool incl = false;
        if (lane < ncon) {
          const float* c = s_con + lane * kConStride;
          pairid = __float_as_int(c[C_PAIR]);
          incl = c[C_DIST] < M.pair_margin[pairid] - M.pair_gap[pairid];
          dim = M.pair_dim[pairid];
          myrows = incl ? (dim == 1 ? 1 : 2 * (dim - 1)) : 0;
        }
        int total_rows;
        const int base = nefc + wave_excl_scan(myrows, lane, total_rows);
        const bool fits = base + myrows <= kNR;
        if (lane < ncon) {
          float* c = s_con + lane * kConStride;
          c[C_ROW] = __int_as_float((incl && fits) ? base : -1);
          c[C_DIM] = __int_as_float(dim);
          c[C_FRIC] = fmaxf(1e-5f, dr ? fmaxf(M.pair_fricab[2 * pairid] * dr[DL.o_fric], M.pair_fricab[2 * pairid + 1]) : M.pair_friction[3 * pairid]);
        }
        if (__ballot(lane < ncon && incl && !fits)) status |= (1 << 2);
        if constexpr (DEFER == 1) {
          if (status & ((1 << 1) | (1 << 2))) {  // more contacts or rows than this instantiation holds: the four-group kernel steps this env
            if (lane == 0) defer_env(P, env);
            return;
          }
        }
        const unsigned long long placed = __ballot(lane < ncon && incl && fits);
        const int nefc_after = placed ? __builtin_amdgcn_readlane(base + myrows, 63 - __builtin_clzll(placed)) : nefc;
        gsync();
        const int rowv = (lane < ncon && incl && fits) ? base : -1;
        for (int ci = 0; ci < ncon; ci++) {
          const int row = __builtin_amdgcn_readlane(rowv, ci);
          if (row < 0) continue;
          const int pid = __builtin_amdgcn_readlane(pairid, ci);
          const float4 HB_CONST* PR = M.prec + (size_t)pid * 5;
          const float4 p0 = PR[0], p1 = PR[1], p2 = PR[2], p3 = PR[3], p4 = PR[4];
          const float* c = s_con + ci * kConStride;
          const int cdim = __float_as_int(p4.y);
          const unsigned long long m1 = ((unsigned long long)__float_as_uint(p1.y) << 32) | __float_as_uint(p1.x);
          const unsigned long long m2 = ((unsigned long long)__float_as_uint(p1.w) << 32) | __float_as_uint(p1.z);
          const V3 cpos = ld3(c + C_POS);
          const V3 off1 = cpos - ld3(s_scom + 3 * __float_as_int(p0.z)), off2 = cpos - ld3(s_scom + 3 * __float_as_int(p0.w));
          const V3 fn = ld3(c + C_FRAME), ft1 = ld3(c + C_FRAME + 3), ft2 = ld3(c + C_FRAME + 6);
          // friction per direction: sliding (2), torsional, rolling (2): mjContact.friction (mjdata.h:113)
          const float mu = c[C_FRIC], mu_t = fmaxf(1e-5f, M.pair_friction[3 * pid + 1]), mu_r = fmaxf(1e-5f, M.pair_friction[3 * pid + 2]);
          for (int d = lane; d < cs; d += kGroup) {
            V3 jd = {0.f, 0.f, 0.f}, jr = {0.f, 0.f, 0.f};  // relative linear velocity at the contact point and relative angular velocity, per unit qvel[d]
            if (d < nv) {
              float cdd[6];
              ld_cdof(s_cdof, d, cdd);
              const V3 ang = {cdd[0], cdd[1], cdd[2]}, lin = {cdd[3], cdd[4], cdd[5]};
              if ((m2 >> d) & 1ull) { jd = jd + lin + cross(ang, off2); jr = jr + ang; }
              if ((m1 >> d) & 1ull) { jd = jd - (lin + cross(ang, off1)); jr = jr - ang; }
            }
            const float j0 = dot(fn, jd);
            float* Jr = s_C + row * cs + d;
            if (cdim == 1) Jr[0] = j0;
            else {
              const float j1 = mu * dot(ft1, jd), j2 = mu * dot(ft2, jd);
              Jr[0] = j0 + j1; Jr[cs] = j0 - j1; Jr[2 * cs] = j0 + j2; Jr[3 * cs] = j0 - j2;
              if (cdim > 3) {
                const float j3 = mu_t * dot(fn, jr);
                Jr[4 * cs] = j0 + j3; Jr[5 * cs] = j0 - j3;
                if (cdim > 4) {
                  const float j4 = mu_r * dot(ft1, jr), j5 = mu_r * dot(ft2, jr);
                  Jr[6 * cs] = j0 + j4; Jr[7 * cs] = j0 - j4; Jr[8 * cs] = j0 + j5; Jr[9 * cs] = j0 - j5;
                }
              }
            }
          }
          const int nr = cdim == 1 ? 1 : 2 * (cdim - 1);
          if (lane < nr) {
            const float tran = p2.w, dist = c[C_DIST], margin = p2.x;
            const float solimp[5] = {p3.x, p3.y, p3.z, p3.w, p4.x};
            const float imp = clampf(impedance(solimp, dist, margin), HB_MINIMP, HB_MAXIMP);
            float K, Bc;
            kb_from_solref(p2.y, p2.z, solimp[1], M.timestep, !(M_disableflags & (1 << 11)), K, Bc);
            // pyramidal rows share 2 mu^2 R(first row), mu = friction[0] / sqrt(impratio); the first row's diagApprox is tran + friction[0]^2 tran
            const float mus = mu * M.inv_sqrt_impratio;
            const float Rown = fmaxf(HB_MINVAL, (1.f - imp) * (cdim == 1 ? tran : tran + mu * mu * tran) / imp);
            float* e = s_meta + kMetaStride * (row + lane);
            e[0] = cdim == 1 ? Rown : 2.f * mus * mus * Rown; e[1] = K * imp * (dist - margin); e[2] = Bc;
          }
        }
        nefc = nefc_after;
      }
    } else {
    // (a) limits: 2 candidates (lower, upper) per limited joint / tendon, in constraint order
    if (constraints_on && !(M_disableflags & (1 << 3))) {
      for (int c0 = 0; c0 < HB_SZ(nlimcand); c0 += kGroup) {
        int c = c0 + lane;
        bool active = false;
        float dist = 0.f, margin = 0.f;
        int side = 0, kind = 0, id = 0;
        float4 l0 = {0.f, 0.f, 0.f, 0.f}, l1 = l0, l2 = l0, l3 = l0;
        if (c < HB_SZ(nlimcand)) {
          const float4 HB_CONST* LR = M.lrec + (size_t)c * 4;  // the candidate's whole record in one round trip
          l0 = LR[0]; l1 = LR[1]; l2 = LR[2]; l3 = LR[3];
          kind = __float_as_int(l0.x); id = __float_as_int(l0.y); side = __float_as_int(l0.z);
          margin = dr ? dr[DL.o_lmargin + c] : l1.x;
          float value = kind == 0 ? s_qpos[__float_as_int(l0.w)] : s_tenlen[id];
          dist = (float)side * ((dr ? dr[DL.o_lrange + c] : l1.y) - value);
          active = dist < margin;
        }
        unsigned long long bal = __ballot(active);
        int row = nefc + __popcll(bal & ((1ull << lane) - 1ull));
        if (active && row < kNR) {
          float* Jr = s_C + row * cs;
          for (int k = 0; k < cs; k++) Jr[k] = 0.f;
          if (kind == 0) Jr[__float_as_int(l3.z)] = (float)(-side);
          else for (int w = 0; w < M.tendon_num[id]; w++) Jr[M.wrap_dofadr[M.tendon_adr[id] + w]] = (float)(-side) * M.wrap_prm[M.tendon_adr[id] + w];
          float* e = s_efc + row;
          e[E_POS * kNR] = dist; e[E_MARGIN * kNR] = margin;
          e[E_SOLREF0 * kNR] = l1.z; e[E_SOLREF1 * kNR] = l1.w;
          e[(E_IMP0 + 0) * kNR] = l2.x; e[(E_IMP0 + 1) * kNR] = l2.y; e[(E_IMP0 + 2) * kNR] = l2.z; e[(E_IMP0 + 3) * kNR] = l2.w;
          e[(E_IMP0 + 4) * kNR] = l3.x;
          e[E_DA * kNR] = l3.y; e[E_DAFIRST * kNR] = l3.y; e[E_MU2 * kNR] = 0.f;
        }
        nefc += __popcll(bal);
      }
      if (nefc > kNR) {
        status |= (1 << 2); nefc = kNR;
      }
    }
    // (b) contacts: row base by prefix sum over contacts (1 row for condim 1, 4 for condim 3)
    if (constraints_on && contacts_on) {
      int myrows = 0, pairid = 0;
      bool incl = false;
      if (lane < ncon) {
        const float* c = s_con + lane * kConStride;
        pairid = __float_as_int(c[C_PAIR]);
        float includemargin = M.pair_margin[pairid] - M.pair_gap[pairid];
        incl = c[C_DIST] < includemargin;
        myrows = incl ? (M.pair_dim[pairid] == 1 ? 1 : 4) : 0;
      }
      // rows before this contact: a contact has 0, 1 or 4 rows, so the exclusive prefix sum is two ballots and two
      // population counts (no cross-lane shuffles)
      const unsigned long long lower = (1ull << lane) - 1ull;
      const unsigned long long one_row = __ballot(myrows == 1), four_rows = __ballot(myrows == 4);
      int base = nefc + __popcll(one_row & lower) + 4 * __popcll(four_rows & lower);
      bool fits = base + myrows <= kNR;
      if (lane < ncon) {
        float* c = s_con + lane * kConStride;
        c[C_ROW] = __int_as_float((incl && fits) ? base : -1);
        c[C_DIM] = __int_as_float(M.pair_dim[pairid] == 1 ? 1 : 3);
        c[C_FRIC] = fmaxf(1e-5f, dr ? fmaxf(M.pair_fricab[2 * pairid] * dr[DL.o_fric], M.pair_fricab[2 * pairid + 1]) : M.pair_friction[3 * pairid]);
      }
      if (__ballot(lane < ncon && incl && !fits)) {
        status |= (1 << 2);
      }
      // contacts are materialised in order; once one does not fit, none of the later ones does
      // (bases grow with the lane: the last contact that fits ends the rows)
      const unsigned long long placed = __ballot(lane < ncon && incl && fits);
      const int nefc_after = placed ? __builtin_amdgcn_readlane(base + myrows, 63 - __builtin_clzll(placed)) : nefc;
      gsync();
      // Jacobian rows: uniform loop over contacts, lanes over dofs.  The pair id and first row of contact ci come
      // out of the lanes that own them (v_readlane); everything the pair contributes is one 5-quad record, fetched
      // with scalar loads.
      const int rowv = (lane < ncon && incl && fits) ? base : -1;
      for (int ci = 0; ci < ncon; ci++) {
        const int row = __builtin_amdgcn_readlane(rowv, ci);
        if (row < 0) continue;
        const int pid = __builtin_amdgcn_readlane(pairid, ci);
        const float4 HB_CONST* PR = M.prec + (size_t)pid * 5;
        const float4 p0 = PR[0], p1 = PR[1], p2 = PR[2], p3 = PR[3], p4 = PR[4];
        const float* c = s_con + ci * kConStride;
        const int dim = __float_as_int(p4.y) == 1 ? 1 : 3;
        const int b1 = __float_as_int(p0.x), b2 = __float_as_int(p0.y);
        const unsigned long long m1 = ((unsigned long long)__float_as_uint(p1.y) << 32) | __float_as_uint(p1.x);
        const unsigned long long m2 = ((unsigned long long)__float_as_uint(p1.w) << 32) | __float_as_uint(p1.z);
        V3 cpos = ld3(c + C_POS);
        V3 off1 = cpos - ld3(s_scom + 3 * __float_as_int(p0.z)), off2 = cpos - ld3(s_scom + 3 * __float_as_int(p0.w));
        V3 fn = ld3(c + C_FRAME), ft1 = ld3(c + C_FRAME + 3), ft2 = ld3(c + C_FRAME + 6);
        float mu = c[C_FRIC];
        (void)b1; (void)b2;
        for (int d = lane; d < cs; d += kGroup) {
          V3 jd = {0.f, 0.f, 0.f};
          if (d < nv) {
            float cdd[6];
          ld_cdof(s_cdof, d, cdd);
          V3 ang = {cdd[0], cdd[1], cdd[2]}, lin = {cdd[3], cdd[4], cdd[5]};
            if ((m2 >> d) & 1ull) jd = jd + lin + cross(ang, off2);
            if ((m1 >> d) & 1ull) jd = jd - (lin + cross(ang, off1));
          }
          float j0 = dot(fn, jd);
          if (dim == 1) s_C[row * cs + d] = j0;
          else {
            float j1 = mu * dot(ft1, jd), j2 = mu * dot(ft2, jd);
            s_C[row * cs + d] = j0 + j1;
            s_C[(row + 1) * cs + d] = j0 - j1;
            s_C[(row + 2) * cs + d] = j0 + j2;
            s_C[(row + 3) * cs + d] = j0 - j2;
          }
        }
        int nr = dim == 1 ? 1 : 4;
        if (lane < nr) {
          const float tran = p2.w;
          float* e = s_efc + row + lane;
          e[E_POS * kNR] = c[C_DIST];
          e[E_MARGIN * kNR] = p2.x;
          e[E_SOLREF0 * kNR] = p2.y; e[E_SOLREF1 * kNR] = p2.z;
          e[(E_IMP0 + 0) * kNR] = p3.x; e[(E_IMP0 + 1) * kNR] = p3.y; e[(E_IMP0 + 2) * kNR] = p3.z; e[(E_IMP0 + 3) * kNR] = p3.w;
          e[(E_IMP0 + 4) * kNR] = p4.x;
          float da = dim == 1 ? tran : tran + mu * mu * tran;
          e[E_DA * kNR] = da; e[E_DAFIRST * kNR] = da;
          float mus = mu * M.inv_sqrt_impratio;
          e[E_MU2 * kNR] = dim == 1 ? 0.f : 2.f * mus * mus;
        }
      }
      nefc = nefc_after;
    }
    }  // classic makeConstraint
    nefc = uniform(nefc);
    // extra right-hand side: row nefc of C holds qfrc_smooth (transformed below, with the rows, into y)
    if constexpr (SOLVER == 0) for (int d = lane; d < cs; d += kGroup) s_C[nefc * cs + d] = d < nv ? s_smooth[d] : 0.f;
    gsync();

    HB_STAMP(9);
    // ---------------------------------------------------------------- per-row quantities (lane = row; NG rows per lane: l + 64 g)
    bool actg[NG];
    float Rg[NG], Ddg[NG], arefg[NG], jwg[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
      const int row = lane + 64 * g;
      actg[g] = row < nefc;
      Rg[g] = 1.f; Ddg[g] = 1.f; arefg[g] = 0.f; jwg[g] = 0.f;
      if (64 * g < nefc && actg[g]) {
        const float* Jr = s_C + row * cs;
        float vel = 0.f, jw_ = 0.f;
        {  // four columns in flight: two independent accumulation chains per product
          float vel2 = 0.f, jw2 = 0.f;
          int k = 0;
          for (; k + 4 <= nv; k += 4) {
            const float j0 = Jr[k], j1 = Jr[k + 1], j2 = Jr[k + 2], j3 = Jr[k + 3];
            vel += j0 * s_qvel[k] + j2 * s_qvel[k + 2]; vel2 += j1 * s_qvel[k + 1] + j3 * s_qvel[k + 3];
            jw_ += j0 * s_warm[k] + j2 * s_warm[k + 2]; jw2 += j1 * s_warm[k + 1] + j3 * s_warm[k + 3];
          }
          for (; k < nv; k++) { const float j = Jr[k]; vel += j * s_qvel[k]; jw_ += j * s_warm[k]; }
          vel += vel2; jw_ += jw2;
        }
        jwg[g] = jw_;
        if constexpr (COLL != 0) {
          const float* e = s_meta + kMetaStride * row;
          Rg[g] = e[0];
          Ddg[g] = 1.f / Rg[g];
          arefg[g] = -e[2] * vel - e[1];
        } else {
          const float* e = s_efc + row;
          float pos = e[E_POS * kNR], margin = e[E_MARGIN * kNR];
          float solref0 = e[E_SOLREF0 * kNR], solref1 = e[E_SOLREF1 * kNR];
          float solimp[5];
          for (int i = 0; i < 5; i++) solimp[i] = e[(E_IMP0 + i) * kNR];
          float imp = clampf(impedance(solimp, pos, margin), HB_MINIMP, HB_MAXIMP);
          float mu2 = e[E_MU2 * kNR];
          float Rown = fmaxf(HB_MINVAL, (1.f - imp) * e[E_DA * kNR] / imp);
          Rg[g] = mu2 > 0.f ? mu2 * Rown : Rown;  // pyramidal: all rows share 2 mu^2 R(first); first row's diagApprox == own
          Ddg[g] = 1.f / Rg[g];
          float K, Bc;
          kb_from_solref(solref0, solref1, solimp[1], M.timestep, !(M_disableflags & (1 << 11)), K, Bc);
          arefg[g] = -Bc * vel - K * imp * (pos - margin);
        }
      }
    }
    const bool rowact = actg[0];
    float R = Rg[0], Dd = Ddg[0], aref = arefg[0], jw = jwg[0], force = 0.f, bvec = 0.f;
    (void)R; (void)bvec; (void)jw;
    gsync();
    HB_STAMP(10);
    int niter = 0;
    float newton_grad = 0.f;  // Newton: gradient left at the solution (lane = dof); enters the damped Euler solve
    (void)newton_grad;
    if constexpr (SOLVER == 0) {
    // ---------------------------------------------------------------- C = J W, W = L^-1 D^-1/2 (the half solve of mj_solveM2 as one GEMM)
    // rows 0..nefc-1 are constraint rows, row nefc is qfrc_smooth (-> y = D^-1/2 L^-T qfrc_smooth).
    // 32-row tiles x 32 dof columns x K = 32 on the matrix cores; A operands are preloaded so the
    // product can be written back over J in place.
    {
      // M^-1 = W W', W = T' D^-1/2 out of the elimination of the dense view of M on the matrix cores
      // (lane id re-materialised: keeps the masks and addresses of this stage from being computed, and held, phases early)
      int lw;
      asm volatile("v_mov_b32 %0, %1" : "=v"(lw) : "v"(lane0));
      f32x16 T, S;
      sym_factor_mfma<NDENSE / 2>(load_sym_pairs<0>(M, s_qLD, lw), T, S, lw);
      store_w_rows(s_W, kWs, T, S, lw);
      gsync();
    }
    if constexpr (NG == 1) {
    {
      const int col = lane & 31, half = lane >> 5;
#pragma unroll
      for (int I = 0; I < 2; I++) {
        if (I == 0 || nefc >= 32) {
          const int arow = 32 * I + col;
          const float* Ap = s_C + arow * cs + half;
          const bool av = arow <= nefc;
          // K runs over the dense order only: the W rows beyond it are identity padding and meet zero J columns (an exact + 0)
          constexpr int kKP = NDENSE / 2;
          float a[kKP];
#pragma unroll
          for (int kk = 0; kk < kKP; kk++) a[kk] = av ? Ap[2 * kk] : 0.f;
          f32x16 D;
#pragma unroll
          for (int r = 0; r < 16; r++) D[r] = 0.f;
#pragma unroll
          for (int kk = 0; kk < kKP; kk++) D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], s_W[(2 * kk + half) * kWs + col], D, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = 32 * I + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row <= nefc) s_C[row * cs + col] = D[r];
          }
        }
      }
    }
    gsync();
    HB_STAMP(11);
    // ---------------------------------------------------------------- efc_b, AR = C C^T + diag(R) (mj_projectConstraint)
    const float* yv = s_C + nefc * cs;
    // AR lives in registers: lane j holds ar[i] = AR[i][j] (= AR[j][i]) for every row i.
    // AR = C C^T is formed on the matrix cores: v_mfma_f32_32x32x2_f32 (exact f32) accumulates 32x32
    // tiles over K = 32 dof columns, operands read straight from the LDS rows of C, the regulariser R
    // injected through the accumulator input of the diagonal tiles.  A 32x32 result has its column on
    // the lane and half of its rows in each 32-lane half; v_permlane32_swap pairs tile (I,0) with
    // tile (I,1) so that every lane ends up with the full column it owns, with no LDS round trip.
    float ar[kNR];
    float Aii = 1.f;
    {
      const float* Cr = s_C + lane * cs;
      float jas = 0.f, diag = 0.f;
#pragma unroll
      for (int k = 0; k < kCs; k++) {
        const float c = rowact ? Cr[k] : 0.f;
        jas += c * yv[k];
        diag += c * c;
      }
      bvec = jas - aref;
      Aii = rowact ? diag + R : 1.f;
      const int col = lane & 31, half = lane >> 5;
      const bool two = nefc > 32;  // rows 32..62 in use: all four tiles, else only tile (0,0)
      const bool v0 = col < nefc, v1 = 32 + col < nefc;
      const float* A0p = s_C + col * cs + half;
      const float* A1p = s_C + (32 + col) * cs + half;
      // R of rows col and 32 + col in every lane: one v_permlane32_swap of R with itself (x: the lower half copied
      // up, y: the upper half copied down)
      const u32x2 rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(R), __float_as_uint(R), false, false);
      const float R0 = __uint_as_float(rr.x), R1 = __uint_as_float(rr.y);
      f32x16 X0, Y0, X1, Y1;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;  // C/D layout: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
        X0[r] = (row == col && v0) ? R0 : 0.f;
        Y1[r] = (row == col && v1) ? R1 : 0.f;
        Y0[r] = 0.f;
        X1[r] = 0.f;
      }
      if (!two) {
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
          const float a0 = v0 ? A0p[2 * kk] : 0.f;
          X0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, X0, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
          const float a0 = v0 ? A0p[2 * kk] : 0.f;
          const float a1 = v1 ? A1p[2 * kk] : 0.f;
          X0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, X0, 0, 0, 0);  // AR[0:32, 0:32]
          Y0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, Y0, 0, 0, 0);  // AR[0:32, 32:64]
          X1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a0, X1, 0, 0, 0);  // AR[32:64, 0:32]
          Y1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, Y1, 0, 0, 0);  // AR[32:64, 32:64]
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ra = (r & 3) + 8 * (r >> 2);
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(X0[r]), __float_as_uint(Y0[r]), false, false);
        ar[ra] = __uint_as_float(s0.x);
        if (ra + 4 < kNR) ar[ra + 4] = __uint_as_float(s0.y);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(X1[r]), __float_as_uint(Y1[r]), false, false);
        if (32 + ra < kNR) ar[32 + ra] = __uint_as_float(s1.x);
        if (32 + ra + 4 < kNR) ar[32 + ra + 4] = __uint_as_float(s1.y);
      }
    }
    HB_STAMP(12);
    // ---------------------------------------------------------------- mj_fwdConstraint: warm start + PGS
    if (nefc > 0) {
      // Rows >= nefc are inert by construction (zero AR entries, zero residual and force), so the
      // unrolled row loops below run in unguarded 4-row chunks up to nefc rounded up.
      const float nAinv = -1.f / Aii;
      float arf = 0.f;  // (AR force)_lane
      if (!(M_disableflags & (1 << 8))) {
        const float jar = jw - aref;
        force = (rowact && jar < 0.f) ? -Dd * jar : 0.f;
#pragma unroll
        for (int c = 0; c < (kNR + 3) / 4; c++) {
          if (c * 4 < nefc) {
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (c * 4 + r < kNR) arf += ar[c * 4 + r] * rdlane(force, c * 4 + r);
          }
        }
        // cost(f) = 0.5 f'AR f + f'b; keep the warm start only if it beats zero
        const float cost = wave_sum(rowact ? force * (0.5f * arf + bvec) : 0.f);
        if (cost > 0.f) { force = 0.f; arf = 0.f; }
      }
      float res = rowact ? bvec + arf : 0.f;
      // Gauss-Seidel sweeps in column form.  Every lane proposes the step of its own row from its
      // current residual, delta = max(-res/AR_ii, -force) (= max(0, force - res/AR_ii) - force); when
      // row i's turn comes the proposal of lane i is the valid one: it is broadcast (v_readlane), all
      // residuals follow (one fma with the lane's AR column entry) and v_writelane records it in the
      // owner's lane.  The dependency chain per row is mul - max - readlane - fma.
      //
      // Cost change of the sweep: sum_i delta_i (0.5 delta_i AR_ii + res_i at its turn) telescopes to
      // 0.5 delta . (res_before + res_after), one wave reduction per sweep instead of per-row terms.
      //
      // The reference reverts a row whose cost change is > 1e-10.  For scalar rows that is unreachable:
      // the unclamped step gives -0.5 res^2 / AR_ii, the clamped one -f (res - 0.5 f AR_ii) with
      // res > f AR_ii, both <= 0; the oracle counts its reverts and the tests assert zero.
      // (the solver options are read once: a scalar load inside the sweep loop is a memory round trip per sweep)
      const int max_sweeps = M.iterations;
      const float pgs_tol = M.tolerance, pgs_scale = M.pgs_scale;
      while (niter < max_sweeps) {
        int ne;
        asm volatile("s_mov_b32 %0, %1" : "=s"(ne) : "s"(nefc));
        const float nforce = -force, res0 = res;
        int dl = 0;
        // hand-unrolled (ar[i] needs a compile-time register index) with one scalar exit test per 4 rows
#define HB_PGS_ROW(i)                                                              \
  if ((i) < kNR) {                                                            \
    const float d_ = fmaxf(res * nAinv, nforce);                                   \
    const int di_ = __builtin_amdgcn_readlane(__float_as_int(d_), (i));            \
    res = __builtin_fmaf(ar[(i) < kNR ? (i) : 0], __int_as_float(di_), res);  \
    dl = hb_writelane(di_, (i), dl);                                               \
  }
#define HB_PGS_CHUNK(c) \
  if ((c) * 4 >= ne) break; \
  HB_PGS_ROW((c) * 4) HB_PGS_ROW((c) * 4 + 1) HB_PGS_ROW((c) * 4 + 2) HB_PGS_ROW((c) * 4 + 3)
        do {
          HB_PGS_CHUNK(0) HB_PGS_CHUNK(1) HB_PGS_CHUNK(2) HB_PGS_CHUNK(3) HB_PGS_CHUNK(4) HB_PGS_CHUNK(5) HB_PGS_CHUNK(6) HB_PGS_CHUNK(7)
          HB_PGS_CHUNK(8) HB_PGS_CHUNK(9) HB_PGS_CHUNK(10) HB_PGS_CHUNK(11) HB_PGS_CHUNK(12) HB_PGS_CHUNK(13) HB_PGS_CHUNK(14) HB_PGS_CHUNK(15)
        } while (0);
#undef HB_PGS_CHUNK
#undef HB_PGS_ROW
        static_assert(kNefcMax <= 64, "PGS sweep is unrolled for at most 64 rows");
        const float delta = __int_as_float(dl);
        force += delta;  // a clamped row lands on exactly 0
        const float improvement = -0.5f * wave_sum(delta * (res0 + res));
        niter++;
        if (improvement * pgs_scale < pgs_tol) break;
      }
    }
    if (lane < kNR) s_force[lane] = rowact ? force : 0.f;
    gsync();
    HB_STAMP(13);
    // ---------------------------------------------------------------- dual finish: s = sum_i f_i C_i ; qacc = W (y + s)
    const bool want_qfrc = P_qfrc_out != nullptr;
    for (int k = lane; k < nv; k += kGroup) {
      float sacc = 0.f;
      for (int i = 0; i < nefc; i++) sacc += s_force[i] * s_C[i * cs + k];
      s_v2[k] = yv[k] + sacc;  // y + s
    }
    gsync();
    if (lane < nv) s_v0[lane] = dot32(s_W + lane * kWs, s_v2);
    gsync();
    // qfrc_smooth + qfrc_constraint = M qacc (the dual finish defines qacc that way; M is intact in the sparse pairs): formed
    // only for callers that read the joint torques (the env adapter's reward); the integrator does not need it (mj_Euler below)
    if (want_qfrc && lane < nv) {
      float acc = 0.f;
#pragma unroll 8
      for (int j = 0; j < 32; j++) acc = __builtin_fmaf(s_qLD[M.mdense[j * 32 + lane]].x, j < nv ? s_v0[j] : 0.f, acc);
      P_qfrc_out[(size_t)env * nv + lane] = acc;
    }
    } else {
      // ================================================================ PGS on NG row groups (kPgsNefcMax rows): a condim 4 / 6 model solved
      // by PGS (ten rows per contact) does not fit the 63 rows whose AR one lane-column each keeps in registers.  Same algorithm, same
      // row order, same arithmetic per row - with AR in LDS ([kNR][kNR], 64 KB: one env per CU) and lane l owning rows l, l + 64.  This is
      // the fallback the one-group kernel defers to (staged step) and slow: Newton is the solver for such models (the reference's default).
      {  // C = J W for every 32-row tile up to the qfrc_smooth row
        const int col = lane & 31, half = lane >> 5;
        for (int I = 0; 32 * I <= nefc; I++) {
          const int arow = 32 * I + col;
          const float* Ap = s_C + arow * cs + half;
          const bool av = arow <= nefc;
          constexpr int kKP = NDENSE / 2;
          float a[kKP];
#pragma unroll
          for (int kk = 0; kk < kKP; kk++) a[kk] = av ? Ap[2 * kk] : 0.f;
          f32x16 D;
#pragma unroll
          for (int r = 0; r < 16; r++) D[r] = 0.f;
#pragma unroll
          for (int kk = 0; kk < kKP; kk++) D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], s_W[(2 * kk + half) * kWs + col], D, 0, 0, 0);
          gsync();  // (every lane's A operands of this tile are in registers before the tile's rows are overwritten)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = 32 * I + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row <= nefc) s_C[row * cs + col] = D[r];
          }
        }
      }
      gsync();
      HB_STAMP(11);
      const float* yv = s_C + nefc * cs;
      float bg[NG], nAinvg[NG], forceg[NG], resg[NG];
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const float* Cr = s_C + (lane + 64 * g) * cs;
        float jas = 0.f, diag = 0.f;
        for (int k = 0; k < kCs; k++) {
          const float c = actg[g] ? Cr[k] : 0.f;
          jas += c * yv[k];
          diag += c * c;
        }
        bg[g] = jas - arefg[g];
        nAinvg[g] = -1.f / (actg[g] ? diag + Rg[g] : 1.f);
        if (lane + 64 * g < kNR) s_force[lane + 64 * g] = Rg[g];  // (R of every row, for the diagonal of AR below; the forces later)
      }
      gsync();
      {  // AR = C C' + diag(R), tile by tile on the matrix cores, into LDS
        const int col = lane & 31, half = lane >> 5;
        const int nt = (nefc + 31) >> 5;
        for (int I = 0; I < nt; I++)
          for (int J = 0; J < nt; J++) {
            const int ra = 32 * I + col, rb = 32 * J + col;
            const float* Ap = s_C + ra * cs + half;
            const float* Bp = s_C + rb * cs + half;
            const bool va = ra < nefc, vb = rb < nefc;
            f32x16 X;
#pragma unroll
            for (int r = 0; r < 16; r++) {
              const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
              X[r] = (I == J && row == col && vb) ? s_force[rb] : 0.f;
            }
#pragma unroll
            for (int kk = 0; kk < 16; kk++) X = __builtin_amdgcn_mfma_f32_32x32x2f32(va ? Ap[2 * kk] : 0.f, vb ? Bp[2 * kk] : 0.f, X, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) s_AR[(32 * I + (r & 3) + 8 * (r >> 2) + 4 * half) * kNR + rb] = X[r];
          }
      }
      gsync();
      HB_STAMP(12);
      if (nefc > 0) {
        float arfg[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) { forceg[g] = 0.f; arfg[g] = 0.f; }
        if (!(M_disableflags & (1 << 8))) {
#pragma unroll
          for (int g = 0; g < NG; g++) {
            const float jar = jwg[g] - arefg[g];
            forceg[g] = (actg[g] && jar < 0.f) ? -Ddg[g] * jar : 0.f;
          }
          for (int i = 0; i < nefc; i++) {
            const float fi = (i >> 6) == 0 ? rdlane(forceg[0], i & 63) : rdlane(forceg[NG - 1], i & 63);
#pragma unroll
            for (int g = 0; g < NG; g++) arfg[g] += (actg[g] ? s_AR[i * kNR + lane + 64 * g] : 0.f) * fi;  // (columns beyond the last tile were never written)
          }
          float cl = 0.f;
#pragma unroll
          for (int g = 0; g < NG; g++) cl += actg[g] ? forceg[g] * (0.5f * arfg[g] + bg[g]) : 0.f;
          if (wave_sum(cl) > 0.f) {
#pragma unroll
            for (int g = 0; g < NG; g++) { forceg[g] = 0.f; arfg[g] = 0.f; }
          }
        }
#pragma unroll
        for (int g = 0; g < NG; g++) resg[g] = actg[g] ? bg[g] + arfg[g] : 0.f;
        const int max_sweeps = M.iterations;
        const float pgs_tol = M.tolerance, pgs_scale = M.pgs_scale;
        while (niter < max_sweeps) {
          float res0[NG], dlt[NG], nf[NG];
#pragma unroll
          for (int g = 0; g < NG; g++) { res0[g] = resg[g]; dlt[g] = 0.f; nf[g] = -forceg[g]; }
          for (int i = 0; i < nefc; i++) {
            const int gi = i >> 6, li = i & 63;
            // the step of row i from its current residual: max(-res / AR_ii, -force); its lane proposes, everybody follows
            const float dprop = gi == 0 ? fmaxf(resg[0] * nAinvg[0], nf[0]) : fmaxf(resg[NG - 1] * nAinvg[NG - 1], nf[NG - 1]);
            const float di = rdlane(dprop, li);
#pragma unroll
            for (int g = 0; g < NG; g++) resg[g] = __builtin_fmaf(actg[g] ? s_AR[i * kNR + lane + 64 * g] : 0.f, di, resg[g]);
            if (lane == li) { if (gi == 0) dlt[0] = di; else dlt[NG - 1] = di; }
          }
          float imp = 0.f;
#pragma unroll
          for (int g = 0; g < NG; g++) { forceg[g] += dlt[g]; imp += dlt[g] * (res0[g] + resg[g]); }
          const float improvement = -0.5f * wave_sum(imp);
          niter++;
          if (improvement * pgs_scale < pgs_tol) break;
        }
      } else {
#pragma unroll
        for (int g = 0; g < NG; g++) forceg[g] = 0.f;
      }
      gsync();
#pragma unroll
      for (int g = 0; g < NG; g++) if (lane + 64 * g < kNR) s_force[lane + 64 * g] = actg[g] ? forceg[g] : 0.f;
      force = forceg[0];
      if (P_diag_force) {
#pragma unroll
        for (int g = 1; g < NG; g++) P_diag_force[(size_t)env * kNR + lane + 64 * g] = actg[g] ? forceg[g] : 0.f;
      }
      gsync();
      HB_STAMP(13);
      // dual finish: s = sum_i f_i C_i ; qacc = W (y + s)
      for (int k = lane; k < nv; k += kGroup) {
        float sacc = 0.f;
        for (int i = 0; i < nefc; i++) sacc += s_force[i] * s_C[i * cs + k];
        s_v2[k] = yv[k] + sacc;
      }
      gsync();
      if (lane < nv) s_v0[lane] = dot32(s_W + lane * kWs, s_v2);
      gsync();
      if (P_qfrc_out && lane < nv) {
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < 32; j++) acc = __builtin_fmaf(s_qLD[M.mdense[j * 32 + lane]].x, j < nv ? s_v0[j] : 0.f, acc);
        P_qfrc_out[(size_t)env * nv + lane] = acc;
      }
    }
    } else {
      // ---------------------------------------------------------------- mj_fwdConstraint, Newton solver (mj_solNewton)
      // Primal problem (oracle/mjstep_oracle.c: sol_newton): minimise over qacc
      //   1/2 (qacc - qacc_smooth)' M (qacc - qacc_smooth) + sum_rows 1/2 D min(0, J qacc - aref)^2
      // by Newton steps with an exact line search.  One wave owns the env, so the whole iteration is uniform:
      // lane = constraint row (rows l + 64 g, g < NG) for jar / force / J rows, lane = dof for qacc / gradient / M rows; the Hessian
      // H = M + J' diag(D active) J is formed on the matrix cores and factorised in registers.
      const int li = lane & 31;
      const bool dofl = lane < nv;
      const int lic = li < nv ? li : 0;  // column of J this lane reads (rows are cs wide: NDENSE + 1 in the big layout, 33 otherwise)
      const bool colv = li < nv;
      // dense M (identity beyond nv) one row per lane, [32][33] over the dead row meta; J rows stay in C
      float* s_Md = s_efc;
      const float* Mrow = s_Md + li * kCs;
      const float* Jrowg[NG];
#pragma unroll
      for (int g = 0; g < NG; g++) Jrowg[g] = s_C + (actg[g] ? lane + 64 * g : 0) * cs;
      constexpr bool kMfma = NDENSE <= 28;  // order <= 28: elimination on the matrix cores; else Cholesky in registers
      f32x2 H[16];
#pragma unroll
      for (int j = 0; j < 32; j++) H[j >> 1][j & 1] = s_qLD[M.mdense[j * 32 + li]].x;
      if (lane < 32) {
#pragma unroll
        for (int j = 0; j < 32; j++) s_Md[lane * kCs + j] = H[j >> 1][j & 1];
      }
      // dof vectors are mirrored in both halves of the wave (lane l and l + 32 hold dof l & 31)
      const float smooth = li < nv ? s_smooth[li] : 0.f;
      const float warm = li < nv ? s_warm[li] : 0.f;
      // qacc_smooth = M^-1 qfrc_smooth
      float dv = 1.f;
      float qs;
      if constexpr (kMfma) {
        gsync();
        qs = sym_solve_mfma<NDENSE / 2>(load_sym(s_Md, kCs, lane), smooth, lane);
      } else {
        dv = chol_rows<NDENSE>(H, s_v1, li, lane);
        qs = chol_solve_rows<NDENSE>(H, dv, smooth);
        qs = rdlane_mirror(qs, lane);
        gsync();
      }
      float qacc = qs, qfc = 0.f;
      float forceg[NG];
#pragma unroll
      for (int g = 0; g < NG; g++) forceg[g] = 0.f;
      HB_STAMP(11);
      if (nefc > 0) {
        // J x for every row group (and M x through the shared broadcasts): rows beyond nefc give 0
#define HB_JDOT(x, Mout, Jout)                                                        \
  {                                                                                   \
    float j0_;                                                                        \
    rowdot2<NDENSE>(Mrow, Jrowg[0], (x), (Mout), j0_);                                \
    (Jout)[0] = actg[0] ? j0_ : 0.f;                                                  \
    _Pragma("unroll") for (int g_ = 1; g_ < NG; g_++) {                               \
      float t_ = 0.f;                                                                 \
      if (64 * g_ < nefc) t_ = rowdot<NDENSE>(Jrowg[g_], (x));                        \
      (Jout)[g_] = actg[g_] ? t_ : 0.f;                                               \
    }                                                                                 \
  }
        // starting point (warmstart() of mj_fwdConstraint): qacc_warmstart unless qacc_smooth costs less
        float Ma, jqs[NG], jar[NG];
        HB_JDOT(qs, Ma, jqs)
#pragma unroll
        for (int g = 0; g < NG; g++) jar[g] = actg[g] ? jqs[g] - arefg[g] : 1.f;  // rows beyond nefc: never active
        if (!(M_disableflags & (1 << 8))) {
          const float Mw = rowdot<NDENSE>(Mrow, warm);
          float cw = dofl ? 0.5f * (Mw - smooth) * (warm - qs) : 0.f, cq = 0.f;
          float jarw[NG];
#pragma unroll
          for (int g = 0; g < NG; g++) {
            jarw[g] = actg[g] ? jwg[g] - arefg[g] : 1.f;
            cw += jarw[g] < 0.f ? 0.5f * Ddg[g] * jarw[g] * jarw[g] : 0.f;
            cq += jar[g] < 0.f ? 0.5f * Ddg[g] * jar[g] * jar[g] : 0.f;
          }
          if (wave_sum(cw - cq) <= 0.f) {
            qacc = warm; Ma = Mw;
#pragma unroll
            for (int g = 0; g < NG; g++) jar[g] = jarw[g];
          }
        }
        const float scale = M.pgs_scale, tol = M.tolerance, lstol = M.ls_tolerance;
        const int maxiter = M.iterations, lsmax = M.ls_iterations;
        float cost = 0.f;
        unsigned long long act_prev[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) act_prev[g] = 0ull;
        bool have_factor = false;
        HB_STAMP(12);
#if defined(HB_STAMPS) && defined(HB_PROBE_NEWTON)
        unsigned long long np_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, np_t = __builtin_amdgcn_s_memtime();
#endif
        for (;;) {
          // PrimalUpdateConstraint: state, force, qfrc_constraint = J' force, cost
          bool act[NG];
          unsigned long long actmask[NG];
          float rowcost = 0.f;
#pragma unroll
          for (int g = 0; g < NG; g++) {
            act[g] = jar[g] < 0.f;
            forceg[g] = act[g] ? -Ddg[g] * jar[g] : 0.f;
            actmask[g] = __ballot(act[g]);
            rowcost += act[g] ? 0.5f * Ddg[g] * jar[g] * jar[g] : 0.f;
          }
          qfc = 0.f;
          {
            // eight rows in flight; rows beyond nefc are read from the last row with a zero force
            float q1 = 0.f;
            const int last = nefc - 1;
#pragma unroll
            for (int g = 0; g < NG; g++) {
              if (64 * g >= nefc) break;
              for (int i0 = 0; i0 < 64 && 64 * g + i0 < nefc; i0 += 8) {
                if (!((actmask[g] >> i0) & 0xffull)) continue;  // eight inactive rows: zero force
                float c[8];
#pragma unroll
                for (int u = 0; u < 8; u++) c[u] = s_C[min(64 * g + i0 + u, last) * cs + lic];
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                  qfc = __builtin_fmaf(c[u], rdlane(forceg[g], i0 + u), qfc);
                  q1 = __builtin_fmaf(c[u + 1], rdlane(forceg[g], i0 + u + 1), q1);
                }
              }
            }
            qfc += q1;
            if (!colv) qfc = 0.f;
          }
          const float oldcost = cost;
          newton_grad = li < nv ? Ma - smooth - qfc : 0.f;
          cost = wave_sum(rowcost + (dofl ? 0.5f * (Ma - smooth) * (qacc - qs) : 0.f));
          // |grad| and the fp32 resolution of its own terms: the reference's gradient test (scale |grad| < tolerance) cannot
          // be met by a sum of O(1e2) terms in fp32, so the test is floored at that sum's rounding level
          const float g2 = wave_sum(dofl ? newton_grad * newton_grad : 0.f);
          const float gm = fabsf(Ma) + fabsf(smooth) + fabsf(qfc);
          const float gm2 = wave_sum(dofl ? gm * gm : 0.f);
          const float gradtol = fmaxf(tol / scale, 1e-6f * sqrtf(gm2));
          if (niter > 0 && (scale * (oldcost - cost) < tol || sqrtf(g2) < gradtol)) break;
          if (niter >= maxiter) break;
          HB_NP(0);
          float search;
          if constexpr (kMfma) {
            // Hessian of the active set in the accumulator layout: M, plus J' diag(D active) J on the matrix cores
            // (MakeHessian), eliminated together with the gradient: search = -H^-1 grad
#pragma unroll
            for (int g = 0; g < NG; g++) if (64 * g < nefc || g == 0) s_force[lane + 64 * g] = act[g] ? Ddg[g] : 0.f;
            gsync();
            const int half = lane >> 5;
            f32x16 X = load_sym(s_Md, kCs, lane);
#pragma unroll
            for (int g = 0; g < NG; g++) {
              if (64 * g >= nefc) break;
              for (int kk = 0; kk < 32 && 64 * g + 2 * kk < nefc; kk++) {
                if (!((actmask[g] >> (2 * kk)) & 3ull)) continue;  // both rows inactive: nothing to add
                const int row = 64 * g + 2 * kk + half;
                const bool v = row < nefc && colv;
                const float a = v ? s_C[row * cs + lic] : 0.f;
                const float b = v ? a * s_force[row] : 0.f;
                X = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, X, 0, 0, 0);  // += J_row' (D J_row)
              }
            }
            HB_NP(1);
            search = -sym_solve_mfma<NDENSE / 2>(X, newton_grad, lane);
            HB_NP(2);
            gsync();
          } else {
            // Hessian of the active set (MakeHessian; rebuilt only when the active set changed) and its Cholesky factor
            bool changed = !have_factor;
#pragma unroll
            for (int g = 0; g < NG; g++) changed |= actmask[g] != act_prev[g];
            if (changed) {
#pragma unroll
              for (int g = 0; g < NG; g++) if (64 * g < nefc || g == 0) s_force[lane + 64 * g] = act[g] ? Ddg[g] : 0.f;
              gsync();
              const int half = lane >> 5;
              f32x16 X;
#pragma unroll
              for (int r = 0; r < 16; r++) X[r] = 0.f;
              for (int kk = 0; 2 * kk < nefc; kk++) {
                const int row = 2 * kk + half;
                const bool v = row < nefc && colv;
                const float a = v ? s_C[row * cs + lic] : 0.f;
                const float b = v ? a * s_force[row] : 0.f;
                X = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, X, 0, 0, 0);  // += J_row' (D J_row)
              }
#pragma unroll
              for (int r = 0; r < 16; r++) {
                const int ra = crow(r);  // C/D layout: row = crow(reg) + 4*(lane>>5), col = lane&31
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(X[r]), __float_as_uint(X[r]), false, false);
                H[ra >> 1][ra & 1] = Mrow[ra] + __uint_as_float(sw.x);
                H[(ra + 4) >> 1][ra & 1] = Mrow[ra + 4] + __uint_as_float(sw.y);
              }
              HB_NP(1);
              dv = chol_rows<NDENSE>(H, s_v1, li, lane);
              HB_NP(2);
#pragma unroll
              for (int g = 0; g < NG; g++) act_prev[g] = actmask[g];
              have_factor = true;
              gsync();
            }
            // PrimalUpdateGradient: search = -H^-1 grad
            search = -rdlane_mirror(chol_solve_rows<NDENSE>(H, dv, newton_grad), lane);
          }
          HB_NP(3);
          float Mv, Jv[NG];
          HB_JDOT(search, Mv, Jv)
          HB_NP(4);
          // PrimalSearch: exact line search on the piecewise quadratic phi(alpha) = cost(qacc + alpha search):
          // Newton iterations in alpha, kept inside the bracket of the sign change once there is one.  The slope
          // tolerance is the reference's (tolerance * ls_tolerance * |search| / scale) floored at fp32 resolution
          // of the slope's own terms.
          const float gq = dofl ? Ma - smooth : 0.f;
          const float qg1 = wave_sum(search * gq), qg2 = 0.5f * wave_sum(dofl ? search * Mv : 0.f), sn2 = wave_sum(dofl ? search * search : 0.f);
          float DJv[NG], DJv2[NG], magr = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int g = 0; g < NG; g++) {
            DJv[g] = Ddg[g] * Jv[g]; DJv2[g] = DJv[g] * Jv[g];
            magr += act[g] ? fabsf(DJv[g] * jar[g]) : 0.f;
            s0 += act[g] ? DJv[g] * jar[g] : 0.f;
            s1 += act[g] ? DJv2[g] : 0.f;
          }
          const float mag = wave_sum(fabsf(search * gq) + magr);
          const float gtol = fmaxf(tol * lstol * sqrtf(sn2) / scale, 2e-6f * mag);
          float alpha = 0.f;
          {
            const float d0 = qg1 + wave_sum(s0), d1 = 2.f * qg2 + wave_sum(s1);
            if (d0 < -gtol) {
              float lo = 0.f, hi = -1.f, a = -d0 / d1;
              for (int it = 0; it < lsmax; it++) {
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                  const float x = jar[g] + a * Jv[g];
                  const bool on = x < 0.f;
                  t0 += on ? DJv[g] * x : 0.f;
                  t1 += on ? DJv2[g] : 0.f;
                }
                const float e0 = qg1 + 2.f * a * qg2 + wave_sum(t0), e1 = 2.f * qg2 + wave_sum(t1);
                alpha = a;
                if (fabsf(e0) < gtol) break;
                if (e0 < 0.f) lo = a; else hi = a;
                float an = a - e0 / e1;
                if (hi >= 0.f && !(an > lo && an < hi)) an = 0.5f * (lo + hi);
                a = an;
              }
            }
          }
          HB_NP(5);
          if (alpha == 0.f) break;
          qacc = __builtin_fmaf(alpha, search, qacc);
          Ma = __builtin_fmaf(alpha, Mv, Ma);
#pragma unroll
          for (int g = 0; g < NG; g++) jar[g] = actg[g] ? __builtin_fmaf(alpha, Jv[g], jar[g]) : 1.f;
          niter++;
        }
#undef HB_JDOT
#if defined(HB_STAMPS) && defined(HB_PROBE_NEWTON)
        if (P.stamps) for (int i = 0; i < 8; i++) stamps_[i] = np_acc[i];
#endif
      } else {
        HB_STAMP(12);
      }
      HB_STAMP(13);
      force = forceg[0];
      if (P_diag_force) {
#pragma unroll
        for (int g = 1; g < NG; g++) P_diag_force[(size_t)env * kNR + lane + 64 * g] = actg[g] ? forceg[g] : 0.f;
      }
      if (dofl) {
        s_v0[lane] = qacc;
        if (P_qfrc_out) P_qfrc_out[(size_t)env * nv + lane] = smooth + qfc;  // qfrc_smooth + qfrc_constraint
      }
      gsync();
    }
    // mj_checkAcc (mujoco.h:307): a bad qacc resets the data and runs mj_forward again; the step then integrates that result
    {
      bool bad = false;
      for (int i = lane; i < nv; i += kGroup) bad |= !(fabsf(s_v0[i]) <= HB_MAXVAL);
      if (__any(bad)) {
        if constexpr (DEFER != 0) {  // the reset and the second forward pass (with a narrowphase of its own) are the four-group kernel's
          if (lane == 0) defer_env(P, env);
          return;
        }
        status |= (1 << 6);
        for (int i = lane; i < nq; i += kGroup) s_qpos[i] = M.qpos0[i];
        for (int i = lane; i < nv; i += kGroup) { s_qvel[i] = 0.f; s_warm[i] = 0.f; s_v0[i] = 0.f; }
        time = 0.f;
        newton_grad = 0.f;
        if (P_xfrc) for (int i = lane; i < 6 * nb; i += kGroup) P_xfrc[(size_t)env * nb * 6 + i] = 0.f;
        gsync();
        if (!redo) {  // second pass of this step from the reset state (the loop increment undoes the decrement)
          redo = true;
          ctrl_zeroed = true;
          step--;
          continue;
        }
        // (the reset state itself gives a bad qacc: nothing sane is left to do; integrate with qacc = 0)
      }
    }
    redo = false;
    ctrl_zeroed = false;
    // diagnostics of this step (parity tests)
    if (P_diag_qacc) for (int i = lane; i < nv; i += kGroup) P_diag_qacc[(size_t)env * nv + i] = s_v0[i];
    if (P_diag_force && lane < kNR) P_diag_force[(size_t)env * kNR + lane] = rowact ? force : 0.f;
    if (P_diag_contact) {
      for (int idx = lane; idx < kNC * kDiagConStride; idx += kGroup) {
        int ci = idx / kDiagConStride, f = idx % kDiagConStride;
        float v = 0.f;
        if (ci < ncon) {
          const float* c = s_con + ci * kConStride;
          if (f < 13) v = c[f];
          else {
            int pid = __float_as_int(c[C_PAIR]);
            v = f == 13 ? (float)__float_as_int(c[C_DIM]) : (f == 14 ? (float)M.pair_geom1[pid] : (float)M.pair_geom2[pid]);
          }
        }
        P_diag_contact[((size_t)env * kNC) * kDiagConStride + idx] = v;
      }
    }
    if (lane == 0) { int* c = P.counts + kCountStride * (size_t)env; c[0] = ncon; c[1] = nefc; c[2] = niter; c[3] = nefc * (niter + 4); c[4] = selfcol; }

    HB_STAMP(14);
    if (P_integrate) {
      // ---------------------------------------------------------------- mj_Euler: (M + h diag(damping)) qacc' = qfrc_smooth + qfrc_constraint
      // With H = M + h B and M qacc = qfrc_smooth + qfrc_constraint the solve is the same as
      //   qacc' = qacc - H^-1 (h B qacc),
      // which needs neither the right-hand side nor qfrc_constraint: only qacc and the damping vector.
      for (int i = lane; i < nv; i += kGroup) s_warm[i] = s_v0[i];  // qacc_warmstart <- qacc
      if (eulerdamp) {
        if constexpr (SOLVER == 0) {
          // H = M + h B from the H halves of the assembled pairs, eliminated on the matrix cores with h B qacc as the
          // right-hand side column
          int le;
          asm volatile("v_mov_b32 %0, %1" : "=v"(le) : "v"(lane0));
          if constexpr (NDENSE <= 28) {
            const int li = le & 31;
            const float rhs = li < nv ? M.timestep * M.dof_damping[li] * s_v0[li] : 0.f;
            const float x = sym_solve_mfma<NDENSE / 2>(load_sym_pairs<1>(M, s_qLD, le), rhs, le);
            if (le < nv) s_v2[le] = s_v0[le] - x;
            gsync();
          } else {
            // order 32 has no spare column for the right-hand side: H^-1 = W_H W_H' with W_H and its transpose written
            // out of the elimination into the (now dead) C rows, then two row-times-vector passes
            float* WH = s_C;
            float* WHT = s_C + 32 * kWs;  // runs on into the (dead) row-meta / W area behind C: the host checks the room
            if (le < nv) s_v2[le] = M.timestep * M.dof_damping[le] * s_v0[le];  // h B qacc
            f32x16 T, S;
            sym_factor_mfma<NDENSE / 2>(load_sym_pairs<1>(M, s_qLD, le), T, S, le);
            store_w_rows(WH, kWs, T, S, le);
            store_w_cols(WHT, kWs, T, S, le);
            gsync();
            float p = 0.f;
            if (le < nv) p = dot32(WHT + le * kWs, s_v2);  // W_H' (h B qacc)
            gsync();
            if (le < nv) s_v1[le] = p;
            gsync();
            float q = 0.f;
            if (le < nv) q = dot32(WH + le * kWs, s_v1);
            gsync();
            if (le < nv) s_v2[le] = s_v0[le] - q;
            gsync();
          }
        } else {
          // dense: H = M + h B from the H halves of the assembled pairs, Cholesky in registers; the right-hand side
          // h B qacc + grad (grad = M qacc - qfrc_smooth - qfrc_constraint, the Newton residual: qfrc_smooth + qfrc_constraint
          // = M qacc - grad, so H^-1 (qfrc_smooth + qfrc_constraint) = qacc - H^-1 (h B qacc + grad))
          const int li = lane & 31;
          const float rhs = li < nv ? M.timestep * M.dof_damping[li] * s_v0[li] + newton_grad : 0.f;
          float x;
          if constexpr (NDENSE <= 28) {
            const float* s_Md = s_efc;  // dense M, still in place
            f32x16 X = load_sym(s_Md, kCs, lane);
            const float hd = li < nv ? M.timestep * M.dof_damping[li] : 0.f;
            const int q = li - 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; r++) X[r] += q == crow(r) ? hd : 0.f;
            x = sym_solve_mfma<NDENSE / 2>(X, rhs, lane);
          } else {
            f32x2 He[16];
#pragma unroll
            for (int j = 0; j < 32; j++) He[j >> 1][j & 1] = s_qLD[M.mdense[j * 32 + li]].y;
            const float dve = chol_rows<NDENSE>(He, s_v1, li, lane);
            x = chol_solve_rows<NDENSE>(He, dve, rhs);
          }
          if (lane < nv) s_v2[lane] = s_v0[lane] - x;
          gsync();
        }
      } else {
        for (int i = lane; i < nv; i += kGroup) s_v2[i] = s_v0[i];
        gsync();
      }
      // mj_advance
      const float h = M.timestep;
      for (int i = lane; i < nv; i += kGroup) s_qvel[i] += h * s_v2[i];
      gsync();
      for (int j = lane; j < HB_SZ(njnt); j += kGroup) {
        int qa = M.jnt_qposadr[j], da = M.jnt_dofadr[j];
        if (M.jnt_type[j] == 0) {
          for (int i = 0; i < 3; i++) s_qpos[qa + i] += h * s_qvel[da + i];
          float n;
          V3 w = normalized(ld3(s_qvel + da + 3), &n);
          Q4 q = qnormalize(ldq(s_qpos + qa + 3));
          stq(s_qpos + qa + 3, qmul(q, axisangle(w, h * n)));
        } else s_qpos[qa] += h * s_qvel[da];
      }
      time += h;
      gsync();
      if (P_qpos_out) {
        float* o = P_qpos_out + ((size_t)step * P.n_env + env) * nq;
        for (int i = lane; i < nq; i += kGroup) o[i] = s_qpos[i];
      }
      if (P_qvel_out) {
        float* o = P_qvel_out + ((size_t)step * P.n_env + env) * nv;
        for (int i = lane; i < nv; i += kGroup) o[i] = s_qvel[i];
      }
    }
  }

#ifdef HB_STAMPS
  HB_STAMP(15);
  if (lane == 0 && P.stamps) for (int i = 0; i < 16; i++) P.stamps[(size_t)env * 16 + i] = stamps_[i];
#endif
  if (P_integrate) {
    if (lane == 0) st_state(0, time);
    for (int i = lane; i < nq; i += kGroup) st_state(1 + i, s_qpos[i]);
    for (int i = lane; i < nv; i += kGroup) { st_state(1 + nq + i, s_qvel[i]); st_state(1 + nq + nv + i, s_warm[i]); }
  }
  if (status && lane == 0) atomicOr(P.status + env, status);
}

#undef HB_SZ
// PGS instantiations: dense order 28 (nv <= 28: the 27-dof humanoid; M^-1 by elimination on the matrix cores) and 32 (sparse L'DL)
// (224 registers instead of the 229 the allocator would take - two values spilled - so that beside two of its waves a SIMD has 64
// registers left: what the closed loop's policy kernel runs in, hb_policy_lean_kernel; amdgpu_num_vgpr counts per half of the file)
__attribute__((amdgpu_num_vgpr(112))) __global__ __launch_bounds__(kGroup, 2) void hb_step_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28>(Mp, P, nsteps); }
__attribute__((amdgpu_num_vgpr(112))) __global__ __launch_bounds__(kGroup, 2) void hb_step_lean_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 0, 1, 0, 1>(Mp, P, nsteps); }
// (the single-step lean kernel with the sizes and the LDS layout of the reference's 27-dof humanoid as constants)
__attribute__((amdgpu_num_vgpr(112))) __global__ __launch_bounds__(kGroup, 2) void hb_step_h27_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 0, 1, 0, 1, 1>(Mp, P, nsteps); }
__attribute__((amdgpu_num_vgpr(112))) __global__ __launch_bounds__(kGroup, 2) void hb_step_h27_q_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 0, 1, 0, 2, 1>(Mp, P, nsteps); }
__attribute__((amdgpu_num_vgpr(112))) __global__ __launch_bounds__(kGroup, 2) void hb_step_lean_q_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 0, 1, 0, 2>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step32_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 32>(Mp, P, nsteps); }
// General instantiations (mesh hulls, height-field prisms, condim 4 / 6): PGS on 63 rows (configs[4]: the 27-dof humanoid on terrain),
// Newton on 256 rows (the reference's own robot, simulation/assets/world.xml: 18 dofs -> dense order 20; up to 28 dofs)
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { HB_STEP_OR_RERUN(0, 28, 1, 1); }
// PGS on kPgsNefcMax rows (AR in LDS: one env per CU) for condim 4 / 6 models, and the one-group fast pass that defers to it (variant 3)
__global__ __launch_bounds__(kGroup, 1) void hb_step_gen_big_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { HB_STEP_OR_RERUN(0, 28, 1, kPgsGroups); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_fast1_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_fast_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, 1, 2>(Mp, P, nsteps); }  // staged step, fast pass
__global__ __launch_bounds__(kGroup, 1) void hb_step_newton_big20_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { HB_STEP_OR_RERUN(2, 20, 1, kBigGroups); }
__global__ __launch_bounds__(kGroup, 1) void hb_step_newton_big28_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { HB_STEP_OR_RERUN(2, 28, 1, kBigGroups); }
// fast pass of a variant-2 model's staged step: Newton on one row group, general collision results, deferring what does not fit
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton_gen20_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 20, 1, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton_gen28_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28, 1, 1, 1>(Mp, P, nsteps); }
// Newton instantiations: dense order 28 (nv <= 28: the 27-dof humanoid) and 32
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton28_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28>(Mp, P, nsteps); }
// lean instantiations (step_body's LEAN: no optional inputs / outputs in the launch) of the kernels the plain step API spends its time in
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton28_lean_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28, 0, 1, 0, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton28_h27_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28, 0, 1, 0, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton28_lean_q_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28, 0, 1, 0, 2>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_fast_lean_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, 1, 2, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton_gen20_team_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 20, 1, 1, 1, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_fast_h27_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, 1, 2, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton_gen20_lean_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 20, 1, 1, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton32_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 32>(Mp, P, nsteps); }

// every step-kernel launch leaves its kernel's name behind (hb_last_kernel: tests and bench.py name the kernel they measured by what the
// library says it launched, not by a literal)
static thread_local const char* g_last_step_kernel = "";
const char* last_step_kernel() { return g_last_step_kernel; }
#define HB_STEP_LAUNCH(kernel, ...) do { g_last_step_kernel = #kernel; hipLaunchKernelGGL(kernel, __VA_ARGS__); } while (0)
// Two envs per wave (hb_step_duo.hip) for the lean launches of the 27-dof humanoid's PGS kernel - where it pays.  A duo wave takes ~ 1.6 x as
// long as a one-env wave and holds two envs: 25 % more env-steps per wave-cycle.  But a single-step launch lasts as long as its slowest
// wave, and the batch-wide barrier between step calls is hidden only while the launches in flight hold more waves than the chip has
// slots (8 per CU for both kernels): 4096 envs are two rounds of one-env waves (the second hides the first one's tail) but exactly ONE
// round of duo waves - 92 against 78 us per step on MI355X; from 8192 envs on the duo kernel wins, 124 against 148 us
// (profiles/r04_duo_sizes.txt).  A rollout launch has no barrier between steps: duo as soon as the batch fills the chip twice over.
// BatchPtrs::duo (hb_batch_duo; HB_DUO in the environment is a new batch's default): 0 never, 1 where it pays, 2 always.
static int wave_slots() {
  static const int slots = [] { int dev = 0, cus = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256; return 8 * cus; }();
  return slots;
}
static bool duo_pays(const BatchPtrs& P, int nsteps) {
  if (P.duo == 0) return false;
  if (P.duo == 2) return true;
  // a launch of several steps: as soon as one-env waves would need a second round (3072 envs: 65 us per step against 76; 2048 envs are
  // one round of one-env waves: 42 against 56 - profiles/r04_fold_sizes_by_batch.txt)
  if (nsteps > 1) return P.n_env > wave_slots();
  // (a launch that covers the whole batch is an unpipelined step call: nothing overlaps its tail anyway, and from 1.5 x the slots on one round
  // of duo waves beats two rounds of one-env waves - 103 against 112 us at 4096 envs)
  return P.nblk == P.n_env ? 2 * P.n_env >= 3 * wave_slots() : 2 * P.n_env >= 5 * wave_slots();
}

// the lean instantiations apply when the launch has none of the optional inputs / outputs (BatchPtrs::lean_ok bit 0, HB_TUNE_LEAN)
static bool lean_launch(const BatchPtrs& P, bool with_qfrc = false) {
  return (P.lean_ok & 1) && !P.xfrc && (with_qfrc || !P.qfrc_out) && !P.sensor_out && !P.qpos_out && !P.qvel_out && !P.diag_qacc && !P.diag_force && !P.diag_contact && !P.dr && !P.env_mask &&
         P.integrate;
}
// Does it pay to run step calls the host has enqueued back to back as ONE launch of several steps (hb_api.cpp: fold_steps)?  When all the
// launch's waves are on the chip at once: then no wave waits for a slot while others run through their steps, and no env waits for the
// batch's slowest one between steps.  One-env waves: up to 8 per CU (2048 envs on MI355X: 42 us per step against 64 for pipelined single
// steps); two-envs-per-wave waves, for the models that have that kernel: up to twice as many envs (4096: 67 against 78).  Beyond one
// round a multi-step launch is no faster than pipelined single steps, and slower when its last round is part empty (4608 envs: 104
// against 85) - profiles/r04_fold_sizes_by_batch.txt.
bool fold_pays(int variant, int solver, int nv, const BatchPtrs& P) {
  if (variant != 0) return false;
  if (P.n_env <= wave_slots()) return true;
  const bool duo_kernel = solver != 2 && nv <= 28 && lean_launch(P) && (P.lean_ok & 2) && duo_pays(P, 2);
  return duo_kernel && (P.n_env + 1) / 2 <= wave_slots();
}
static hipError_t launch_step_kernel(const DevModel* M_dev, int variant, int solver, int nv, size_t shmem, const BatchPtrs& P, int nsteps, hipStream_t stream) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  // (the second pass of a staged step: kRerunWaves waves that walk the list of deferred envs - HB_STEP_OR_RERUN)
  constexpr int kRerunWaves = 1024;  // (one per SIMD: the four-group kernels hold one wave per SIMD)
  const int grid = (P.stage.rerun && P.stage.defer_list) ? (P.nblk < kRerunWaves ? P.nblk : kRerunWaves) : P.nblk;
  if (variant == 2 && nv <= 20) HB_STEP_LAUNCH(hb_step_newton_big20_kernel, dim3(grid), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (variant == 2) HB_STEP_LAUNCH(hb_step_newton_big28_kernel, dim3(grid), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (variant == 1) HB_STEP_LAUNCH(hb_step_gen_kernel, dim3(grid), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (variant == 3) HB_STEP_LAUNCH(hb_step_gen_big_kernel, dim3(grid), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (solver == 2 && nv <= 28) {
    if (nsteps == 1 && lean_launch(P) && (P.lean_ok & 2)) HB_STEP_LAUNCH(hb_step_newton28_h27_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (nsteps == 1 && lean_launch(P)) HB_STEP_LAUNCH(hb_step_newton28_lean_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (lean_launch(P, true)) HB_STEP_LAUNCH(hb_step_newton28_lean_q_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else HB_STEP_LAUNCH(hb_step_newton28_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  }
  else if (solver == 2) HB_STEP_LAUNCH(hb_step_newton32_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (nv <= 28) {
    // (the two-envs-per-wave kernels write the joint torques when asked: the env adapter's launches take them as well)
    if (lean_launch(P, true) && (P.lean_ok & 2) && duo_pays(P, nsteps)) { g_last_step_kernel = nsteps == 1 ? "hb_step_duo_kernel" : "hb_step_duo_q_kernel"; return launch_step_duo(M_dev, P, nsteps, stream); }
    else if (nsteps == 1 && lean_launch(P) && (P.lean_ok & 2)) HB_STEP_LAUNCH(hb_step_h27_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (nsteps == 1 && lean_launch(P)) HB_STEP_LAUNCH(hb_step_lean_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (lean_launch(P, true) && (P.lean_ok & 2)) HB_STEP_LAUNCH(hb_step_h27_q_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (lean_launch(P, true)) HB_STEP_LAUNCH(hb_step_lean_q_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else HB_STEP_LAUNCH(hb_step_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  }
  else HB_STEP_LAUNCH(hb_step32_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  return hipGetLastError();
}

// One step launch of the classic variant covers all nsteps.  A general variant with stage buffers runs every step as three launches
// on the same stream: poses + work items, narrowphase (a small kernel at 2-4x the step kernel's occupancy: its time is chains of
// dependent loads along the hulls' edge graphs), then the step kernel, which appends the results instead of colliding.
hipError_t launch_step(const DevModel* M_dev, int variant, int solver, int nv, int lds_floats, const BatchPtrs& P, int nsteps, hipStream_t stream) {
  const size_t shmem = (size_t)lds_floats * sizeof(float);
  if (variant == 0 || !P.stage.result) return launch_step_kernel(M_dev, variant, solver, nv, shmem, P, nsteps, stream);
  const char* fast_name = nullptr;  // (a staged step is named after its fast-pass kernel: the one that steps almost every env)
  for (int t = 0; t < nsteps; t++) {
    BatchPtrs Q = P;
    Q.t0 = P.t0 + t;
    if (P.ctrl_mode == 1) Q.ctrl = P.ctrl + (size_t)t * P.n_env * P.stage.nu;
    if (P.qpos_out) Q.qpos_out = P.qpos_out + (size_t)t * P.n_env * P.stage.nq;
    if (P.qvel_out) Q.qvel_out = P.qvel_out + (size_t)t * P.n_env * P.stage.nv;
    if (P.sensor_out) Q.sensor_out = P.sensor_out + (size_t)t * P.n_env * P.sensor_stride;
    hipError_t e = launch_pose_narrow(M_dev, Q, stream);  // (hb_narrow.hip)
    if (e != hipSuccess) return e;
    if (variant == 1 && Q.stage.defer) {
      // the step kernel without the portal-search code; the full one then takes the (rare) env-steps whose qacc came out bad
      if (lean_launch(Q) && (Q.lean_ok & 2)) HB_STEP_LAUNCH(hb_step_gen_fast_h27_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, Q, 1);
      else if (lean_launch(Q)) HB_STEP_LAUNCH(hb_step_gen_fast_lean_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, Q, 1);
      else HB_STEP_LAUNCH(hb_step_gen_fast_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, Q, 1);
      e = hipGetLastError();
      if (e != hipSuccess) return e;
      fast_name = g_last_step_kernel;
      Q.stage.rerun = 1;
    } else if (variant == 3 && Q.stage.dm_fast) {
      // PGS: the one-group kernel (63 rows, 24 contacts, two waves per SIMD) first; the kPgsNefcMax-row kernel then steps what it defers
      HB_STEP_LAUNCH(hb_step_gen_fast1_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      e = hipGetLastError();
      if (e != hipSuccess) return e;
      fast_name = g_last_step_kernel;
      Q.stage.rerun = 1;
    } else if (variant == 2 && Q.stage.dm_fast) {
      // most env-steps fit the one-group Newton instantiation (two waves per SIMD); the four-group kernel then steps the rest
      if (nv <= 20 && lean_launch(Q) && (Q.lean_ok & 4)) HB_STEP_LAUNCH(hb_step_newton_gen20_team_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      else if (nv <= 20 && lean_launch(Q)) HB_STEP_LAUNCH(hb_step_newton_gen20_lean_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      else if (nv <= 20) HB_STEP_LAUNCH(hb_step_newton_gen20_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      else HB_STEP_LAUNCH(hb_step_newton_gen28_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      e = hipGetLastError();
      if (e != hipSuccess) return e;
      fast_name = g_last_step_kernel;
      Q.stage.rerun = 1;
    }
    e = launch_step_kernel(M_dev, variant, solver, nv, shmem, Q, 1, stream);
    if (e != hipSuccess) return e;
    // a long rollout is one call: refresh the heavy-first orders of its launches along the way (the caller does it between calls)
    if (P.order && P.order2 && (t & 7) == 7 && t + 1 < nsteps) {
      e = launch_order(P.counts, const_cast<int*>(P.order), P.n_env, P.blk0, P.nblk, stream, 3, 3);
      if (e == hipSuccess) e = launch_order(P.counts, const_cast<int*>(P.order2), P.n_env, P.blk0, P.nblk, stream, 7, 0);
      if (e != hipSuccess) return e;
    }
  }
  if (fast_name) g_last_step_kernel = fast_name;
  return hipSuccess;
}
hipError_t set_step_lds_limit(int bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)hb_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step_newton28_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step_gen_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step_gen_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step_gen_fast1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step_gen_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step_newton_gen20_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step_newton_gen28_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step_newton_big20_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)hb_step_newton_big28_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)hb_step_newton32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace hb
