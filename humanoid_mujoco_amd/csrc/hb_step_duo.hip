// hb_step_duo.hip — TWO environments per wavefront: the single-step PGS kernel of the reference's 27-dof humanoid.
//
// hb_step_h27_kernel (hb_step.hip) advances one env per 64-lane wave, and on average 34 of the 64 lanes of its vector instructions
// do work: 17 bodies, 27 dofs, 20 geoms and ~11 constraint rows on 64 lanes (profiles/r03_counters.json).  This kernel puts env A on
// lanes 0..31 and env B on lanes 32..63 of ONE wave for every lane-parallel stage (kinematics, comPos, comVel, crb / rne, qM, bias,
// collision, makeConstraint), which roughly halves their instruction count per env, and keeps the stages that already fill the wave
// as they are, run once per env: the eliminations of M and H on the matrix cores (v_mfma_f32_32x32x2_f32) and C = J W, AR = C C' + R.
//
// Constraint rows are PACKED into the wave's 64 row lanes: env A's rows from lane 0, env B's from lane 32 (each <= 31 rows: 99.9 % of
// the benchmark's env-steps), or - one env above 31 rows - the larger env from lane 0 and the other behind it (n_first + n_second + 2
// <= 64, rows aligned to 4).  AR of the packed system is block diagonal (the MFMA operands of an env are masked to its own rows), so
// ONE Gauss-Seidel sweep over the packed rows is the two envs' sweeps one after the other: the row update is exactly the one-env
// kernel's (mul, max, v_readlane, fma, v_writelane), every env converges, and stops, by its own test.  No env-step is handed to another
// kernel: what does not fit two to a wave (more than 12 contacts in an env, rows beyond the packed capacity, a bad qacc that makes
// mj_step run mj_forward a second time) is stepped by this wave with one env at a time, at the one-env kernel's capacities
// (63 rows, 24 contacts).
//
// The arithmetic per env is statement for statement that of step_body<0, 28, 0, 1, 0, 0, 1, 1> (hb_step.hip: same expressions, same
// order, same reductions over the same lanes of the env's half), so the two kernels give the same bits
// (tests/test_gpu_duo.py); reference path: mj_step (simulation/mujoco/include/mujoco/mujoco.h:120), oracle oracle/mjstep_oracle.c.
//
// LDS (one block = one wave = two envs): 20 320 bytes -> 8 blocks per CU, two waves per SIMD, i.e. 16 envs resident per CU where the
// one-env kernel holds 8: all 4096 envs of the benchmark batch are resident at once.
#include "hb_kcommon.hpp"
#include "hb_launch.hpp"

namespace hb {
namespace duo {

constexpr SizedModel Z = kSizedHumanoid27;
constexpr int NQ = Z.nq, NV = Z.nv, NU = Z.nu, NB = Z.nbody, NJ = Z.njnt, NGEOM = Z.ngeom, NT = Z.ntendon, NM = Z.nM, NPAIR = Z.npair, NLEVEL = Z.nlevel, NLIM = Z.nlimcand, NSTATE = Z.nstate;
static_assert(NQ == 28 && NV == 27 && NU <= 24 && NB <= 17 && NGEOM <= 32 && NT <= 32 && Z.ntree == 1 && NLIM <= 64 && NV <= 28, "the duo kernel is written for the 27-dof humanoid's signature");
constexpr int H = 32;        // lanes per env
constexpr int kCsD = 29;     // row stride of C: 28 dof columns + one pad (odd: lane-strided row access is conflict-free)
constexpr int kWsD = 28;     // row stride of W: 16-byte aligned rows of 28
constexpr int kNCh = 12;     // contacts per env while two envs share the wave (24 = kNconMax when one env is stepped alone)
constexpr int kMetaSlots = 11;
constexpr int kDuoAnti = 16;  // heaviest dispatch slots of a launch that are paired with the lightest (the others: with their neighbour)
// per-row meta slots (the one-env kernel's, without its two unused ones)
enum { E_POS = 0, E_MARGIN, E_SOLREF0, E_SOLREF1, E_IMP0, E_IMP1, E_IMP2, E_IMP3, E_IMP4, E_DA, E_MU2 };
// contact record: the one-env kernel's 17 floats, then the dof masks of the two bodies
enum { C_M1 = 17, C_M2 = 18 };

// ---- LDS map (floats) -----------------------------------------------------------------------------------------------------
// per env (kEnvF each): state | scom | M (sparse, + zero / one pads) | diag(H) | qfrc_smooth | v0 (qacc) | v2 | tendon lengths |
//   { cdof | geom positions | geom axes | contacts | pad }  <- W = L^-1 D^-1/2 [28][28] overwrites this group from the half solve on
constexpr int o_qpos = 0, o_qvel = 28, o_warm = 56, o_ctrl = 84, o_scom = 108, o_qM = 112, o_Hd = 360, o_smooth = 388, o_v0 = 416, o_v2 = 444, o_tenlen = 472;
constexpr int o_W = 476, o_cdof = 476, o_gpos = o_cdof + kCdofStride * NV, o_gaxis = o_gpos + 60, o_con = o_gaxis + 60;
constexpr int kEnvF = o_W + kWsD * 28;
static_assert(o_con + kNCh * kConStride <= kEnvF && o_qM + NM + 2 <= o_Hd && o_gpos == 800 && o_con == 920, "per-env LDS map");
// shared by the wave: row meta [kMetaSlots][64] | C [64][kCsD]; the dynamics scratch of both envs aliases these two (dead before mj_collision)
constexpr int o_meta = 2 * kEnvF, o_C = o_meta + kMetaSlots * 64, kLdsF = o_C + 64 * kCsD;
// dynamics scratch per env
constexpr int d_xpq = 0, d_xmat = d_xpq + kXpqStride * NB, d_xipos = d_xmat + 156, d_xanchor = d_xipos + 52, d_xaxis = d_xanchor + 68, d_cinert = d_xaxis + 68, d_if = d_cinert + 172,
              d_va = d_if + kIfStride * NB, kDynF = d_va + 12 * NB;
static_assert(2 * kDynF <= kLdsF - o_meta, "the dynamics scratch of both envs fits under the constraint arrays");
static_assert(kLdsF * 4 <= 20480, "eight blocks per CU");
static_assert(9 * NB <= 156 && 3 * NB <= 52 && 3 * NJ <= 68 && 10 * NB <= 172, "dynamics scratch map");

// sum over the 32 lanes of each half (wave_sum's DPP tree inside the rows of 16, then the two rows of the half): lo for lanes 0..31, hi for 32..63
__device__ __forceinline__ void half_sums(float v, float& lo, float& hi) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  lo = rdlane(v, 0) + rdlane(v, 16);
  hi = rdlane(v, 32) + rdlane(v, 48);
}

// the sparse M (WHICH = 0) or H = M + h B (WHICH = 1: the diagonal from s_Hd) of one env in the accumulator layout (load_sym_pairs of hb_kcommon.hpp)
template <int WHICH>
__device__ __forceinline__ f32x16 load_sym_env(DevModelRef M, const float* s_qM, const float* s_Hd, int lane0) {
  int lane;
  asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane0));
  int e[16];
#pragma unroll
  for (int r = 0; r < 16; r++) e[r] = M.mdense_c[r * 64 + lane];
  f32x16 X;
#pragma unroll
  for (int r = 0; r < 16; r++) X[r] = s_qM[e[r]];
  if constexpr (WHICH == 1) {
    const int li = lane & 31, q = li - 4 * (lane >> 5);
    const float hd = li < NV ? s_Hd[li] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; r++) X[r] = q == crow(r) ? hd : X[r];
  }
  return X;
}

// both envs' matrices with one fetch of the sixteen table words
template <int WHICH>
__device__ __forceinline__ void load_sym_env2(DevModelRef M, const float* qMa, const float* Hda, const float* qMb, const float* Hdb, int lane0, f32x16& Xa, f32x16& Xb) {
  int lane;
  asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane0));
  int e[16];
#pragma unroll
  for (int r = 0; r < 16; r++) e[r] = M.mdense_c[r * 64 + lane];
#pragma unroll
  for (int r = 0; r < 16; r++) { Xa[r] = qMa[e[r]]; Xb[r] = qMb[e[r]]; }
  if constexpr (WHICH == 1) {
    const int li = lane & 31, q = li - 4 * (lane >> 5);
    const float ha = li < NV ? Hda[li] : 1.f, hb_ = li < NV ? Hdb[li] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; r++) { Xa[r] = q == crow(r) ? ha : Xa[r]; Xb[r] = q == crow(r) ? hb_ : Xb[r]; }
  }
}

// W[c][row] = T[row][c] * S(row), rows and columns 0..27 only (stride kWsD): store_w_rows without the identity padding
__device__ __forceinline__ void store_w28(float* W, const f32x16& T, const f32x16& S, int lane) {
  const int c = lane & 31, half = lane >> 5;
  if (c >= 28) return;
  float* p = W + c * kWsD + 4 * half;
#pragma unroll
  for (int g = 0; g < 4; g++)
    if (g < 3 || half == 0) *reinterpret_cast<float4*>(p + 8 * g) = {T[4 * g] * S[4 * g], T[4 * g + 1] * S[4 * g + 1], T[4 * g + 2] * S[4 * g + 2], T[4 * g + 3] * S[4 * g + 3]};
}

// dot32 over the 28 stored columns (the four beyond are zero in the one-env kernel's rows: the same sum)
__device__ __forceinline__ float dot28(const float* row, const float* v) {
  const float4* a = reinterpret_cast<const float4*>(row);
  const float4* b = reinterpret_cast<const float4*>(v);
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < 7; q++) { const float4 x = a[q], y = b[q]; acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; }
  return acc;
}

}  // namespace duo

// MULTI = 0: one step (the step API); 1: nsteps steps with the state on chip in between (the rollouts: every wave advances its two envs through
// all of them, no batch-wide barrier between steps)
template <int MULTI>
__device__ __forceinline__ void step_duo(const DevModel* Mp, const BatchPtrs& P, int nsteps_in) {
  using namespace duo;
  const int nsteps = MULTI ? nsteps_in : 1;
  DevModelRef M = *(const DevModel HB_CONST*)(uintptr_t)Mp;
  extern __shared__ float lds[];
  const int lane0 = threadIdx.x;
  int lane = lane0;
  const int npairs = (P.nblk + 1) >> 1;
  if ((int)blockIdx.x >= npairs) return;
#ifdef HB_STAMPS
  const unsigned long long t_wave0 = __builtin_amdgcn_s_memtime();
#endif
  // The wave's two envs.  The sweeps of two envs run side by side and last as long as the longer one, so envs of like cost share a wave:
  // neighbours in the heavy-first order (slots 2 j, 2 j + 1).  Only the kDuoAnti costliest slots of the launch take the cheapest ones as
  // partners: an env above 31 rows packs its rows in front of a light partner's, while two of them in one wave would have to be stepped
  // one after the other.
  // A launch of SEVERAL steps (MULTI) is one round of waves that lasts as long as its heaviest pair's total, and how heavy an env is persists
  // (rows x sweeps summed over 256 steps: 16 % spread between envs, single ones at 2.5 x the mean) while the sweeps of a pair cost the same
  // whoever the partner is: so there every env takes the one from the other end of the order - the order being by the envs' AVERAGE cost
  // over the previous launch (below: counts[3]).  Modelled on 256 recorded steps (tools/gpu_pair_balance.py, profiles/r04_pair_balance.txt):
  // heaviest pair / mean pair 1.38 with neighbours, 1.18 with opposite ends.
  int slotA, slotB;
  {
    const int kAnti = MULTI ? (P.nblk >> 1) : min(kDuoAnti, P.nblk >> 2), b = (int)blockIdx.x;
    if (b < kAnti) { slotA = b; slotB = P.nblk - 1 - b; }
    else { slotA = kAnti + 2 * (b - kAnti); slotB = slotA + 1 < P.nblk - kAnti ? slotA + 1 : -1; }
    slotA += P.blk0;
    if (slotB >= 0) slotB += P.blk0;
  }
  const int envA = P.order ? P.order[slotA] : slotA;
  const int envB = slotB >= 0 ? (P.order ? P.order[slotB] : slotB) : -1;
  const int h = lane0 >> 5;
  const int env = h ? envB : envA;  // (the lane's env for everything lane-parallel; -1: no second env in this wave)

  float* const E = lds + h * kEnvF;             // the lane's env
  float* const Dn = lds + o_meta + h * kDynF;   // its dynamics scratch
  float* const s_meta = lds + o_meta;
  float* const s_C = lds + o_C;
  float* s_qpos = E + o_qpos; float* s_qvel = E + o_qvel; float* s_warm = E + o_warm; float* s_ctrl = E + o_ctrl;
  float* s_scom = E + o_scom; float* s_qM = E + o_qM; float* s_Hd = E + o_Hd; float* s_smooth = E + o_smooth;
  float* s_v0 = E + o_v0; float* s_v2 = E + o_v2; float* s_tenlen = E + o_tenlen;
  float* s_cdof = E + o_cdof; float* s_gpos = E + o_gpos; float* s_gaxis = E + o_gaxis;
  float* s_xpq = Dn + d_xpq; float* s_xmat = Dn + d_xmat; float* s_xipos = Dn + d_xipos; float* s_xanchor = Dn + d_xanchor; float* s_xaxis = Dn + d_xaxis;
  float* s_cinert = Dn + d_cinert; float* s_if = Dn + d_if; float* s_va = Dn + d_va;

  // ---- state and controls in
  const int l0 = lane0 & 31;
  float time = 0.f;
  {
    const bool have = env >= 0;
    const float* gs = P.state + (size_t)(have ? env : 0) * NSTATE;
    float tq = 0.f, tv = 0.f, tw = 0.f;
    if (have) time = gs[0];
    if (have && l0 < NQ) tq = gs[1 + l0];
    if (have && l0 < NV) { tv = gs[1 + NQ + l0]; tw = gs[1 + NQ + NV + l0]; }
    if (l0 < NQ) s_qpos[l0] = tq;
    if (l0 < 28) { s_qvel[l0] = tv; s_warm[l0] = tw; s_v0[l0] = 0.f; s_v2[l0] = 0.f; }
    // pads behind the sparse matrix (zero, one: what the dense views read outside the sparsity pattern / beyond nv)
    if (l0 < 2) s_qM[NM + l0] = l0 == 1 ? 1.f : 0.f;
  }
  bool eulerdamp;
  {
    bool d = false;
    if (l0 < NV) d = M.dof_damping[l0] > 0.f;
    eulerdamp = __any(d);
  }
  gsync();

#ifdef HB_STAMPS
  unsigned long long stamps_[16] = {0};
#endif
  // controls of step t: ctrl[e][nu] (mode 0), ctrl[t][e][nu] (mode 1), ctrl_tab[t][e][nu] (mode 3), or the benchmark's on-device Halton sequence (mode 2); those of step
  // t + 1 are requested at the top of step t (an HBM round trip that would otherwise open every step)
  auto ctrl_of = [&](int t) -> float {
    if (!(env >= 0 && l0 < NU)) return 0.f;
    if (P.ctrl_mode == 2) return 2.f * halton(1 + P.t0 + t + 1000 * (P.env_offset + env), l0 + 2) - 1.f;
    if (MULTI && P.ctrl_mode == 3) return P.ctrl_tab[t][(size_t)env * NU + l0];  // step calls folded into this launch
    return P.ctrl[(P.ctrl_mode == 1 ? (size_t)t * P.n_env * NU : 0) + (size_t)env * NU + l0];
  };
  float ctrl_pf = ctrl_of(0);
  int status = 0;            // per lane = per env (identical in the lanes of a half)
  int cost_sum = 0, cost_n = 0;  // the env's rows x sweeps over the steps of this launch
  constexpr int kCostHistory = 64;
  const int cost_prev = (MULTI && env >= 0) ? P.counts[kCountStride * (size_t)env + 3] : 0;
  bool ctrl_zeroed = false;  // the env's pass runs on reset data (mj_resetData zeroes ctrl): per half
  bool redo = false;         // the env's pass is the second mj_forward of a step whose first one gave a bad qacc: per half
  for (int step = 0; step < nsteps; step++) {
  if constexpr (MULTI != 0) {
    // The two waves of a SIMD are blocks half a round apart, and the SIMD favours the older one: over a 250-step launch the first half of
    // the blocks took 32.5 M ticks, their SIMD-mates 37 M (tools/gpu_wave_ends.py) - and the launch lasts as long as its last wave.  So the
    // two take the higher priority in turns, step by step.
    // (tried on top: a constant high priority for the heaviest 1/32 or 1/128 of the pairs - the launch's critical path -: no better, 60.6 - 60.8 us per
    // step against 60.5; no priorities at all: 61.1, and 67.8 against 65.9 for launches of 20 steps; the younger wave always high: 66.9)
    if (((step & 1) == 0) == ((int)blockIdx.x < (npairs >> 1))) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
  if (l0 < NU) s_ctrl[l0] = ctrl_pf;
  if (MULTI && step + 1 < nsteps && P.ctrl_mode != 0) ctrl_pf = ctrl_of(step + 1);
  unsigned todo = envB >= 0 ? 3u : 1u;  // envs of the wave still to be stepped (bit 0: A, bit 1: B)
  bool serial = false;       // one env at a time from now on
  while (todo) {
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane0) : "memory");
    const int l = lane & 31;
    const unsigned act = serial ? (todo & (0u - todo)) : todo;  // the envs of this pass
    const bool both = act == 3u;
    const bool on = ((act >> h) & 1u) != 0;  // the lane's env takes part in this pass
    const int ncap = both ? kNCh : kNconMax;  // contacts per env

    // Lane l owns body slot l + 1 (level order) of its env for the whole pass
    const bool bl = on && l + 1 < NB;
    float4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, bp = q0, bq = q0, ip = q0, ch0 = q0, ch1 = q0;
    float4 JA[3], JB[3], JC[3];
#pragma unroll
    for (int jj = 0; jj < 3; jj++) { JA[jj] = q0; JB[jj] = q0; JC[jj] = q0; }
    if (bl) {
      const float4 HB_CONST* R = M.brec + (size_t)(l + 1) * kBrecQuads;
      q0 = R[0]; q1 = R[1]; bp = R[2]; bq = R[3]; ip = R[4]; ch0 = R[7]; ch1 = R[8];
#pragma unroll
      for (int jj = 0; jj < 3; jj++) { JA[jj] = R[9 + 3 * jj]; JB[jj] = R[10 + 3 * jj]; JC[jj] = R[11 + 3 * jj]; }
    }
    int pf_gbody = 0;
    V3 pf_gpos = {0.f, 0.f, 0.f};
    Q4 pf_gquat = {1.f, 0.f, 0.f, 0.f};
    const bool geoml = on && l < NGEOM, dofl = on && l < NV;
    if (geoml) { pf_gbody = M.geom_bodyid[l]; pf_gpos = ld3(M.geom_pos + 3 * l); pf_gquat = ldq(M.geom_quat + 4 * l); }
    float4 pf_dA = {0.f, 0.f, 0.f, 0.f}, pf_dB = pf_dA;
    if (dofl) { pf_dA = M.drec[3 * l]; pf_dB = M.drec[3 * l + 1]; }
    HB_STAMP(0);
    // ---------------------------------------------------------------- mj_checkPos / mj_checkVel
    {
      bool badp = false, badv = false;
      if (on && l < NQ) { const float v = s_qpos[l]; badp = !(fabsf(v) <= HB_MAXVAL); }
      if (dofl) { const float v = s_qvel[l]; badv = !(fabsf(v) <= HB_MAXVAL); }
      const unsigned long long bp64 = __ballot(badp), bv64 = __ballot(badv);
      const bool anyp = (unsigned)(bp64 >> (32 * h)) != 0u, anyv = (unsigned)(bv64 >> (32 * h)) != 0u;
      if (anyp || anyv) {
        status |= anyp ? (1 << 4) : (1 << 5);
        if (l < NQ) s_qpos[l] = M.qpos0[l];
        if (l < NV) { s_qvel[l] = 0.f; s_warm[l] = 0.f; }
        time = 0.f;
        ctrl_zeroed = true;
      }
      if (on && ctrl_zeroed && l < NU) s_ctrl[l] = 0.f;
    }
    gsync();

    HB_STAMP(1);
    // ---------------------------------------------------------------- mj_kinematics
    if (on && l == 0) {
      st3(s_xpq, {0.f, 0.f, 0.f}); stq(s_xpq + 4, {1.f, 0.f, 0.f, 0.f}); st3(s_xipos, {0.f, 0.f, 0.f});
      for (int i = 0; i < 9; i++) s_xmat[i] = (i % 4 == 0) ? 1.f : 0.f;
    }
    gsync();
    const int myb = __float_as_int(q0.x), myp = __float_as_int(q0.y), myjn = __float_as_int(q0.z), myja = __float_as_int(q0.w);
    const int mylevel = bl ? (__float_as_int(q1.x) & 255) : -1, mycn = __float_as_int(q1.w);
    const int myanc2 = (__float_as_int(q1.x) >> 8) & 255, myanc4 = (__float_as_int(q1.x) >> 16) & 255, myanc8 = (__float_as_int(q1.x) >> 24) & 255;
    const float mymass = q1.z;
    const int mych[8] = {__float_as_int(ch0.x), __float_as_int(ch0.y), __float_as_int(ch0.z), __float_as_int(ch0.w),
                         __float_as_int(ch1.x), __float_as_int(ch1.y), __float_as_int(ch1.z), __float_as_int(ch1.w)};
    const bool isfree = bl && myjn == 1 && __float_as_int(JA[0].x) == 0;
    V3 posl = {bp.x, bp.y, bp.z};
    Q4 quatl = {bq.x, bq.y, bq.z, bq.w};
    V3 axl[3], ancl[3];
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
    V3 mypos = posl;
    Q4 myquat = quatl;
    if (bl) {
      reinterpret_cast<float4*>(s_xpq + kXpqStride * myb)[0] = {mypos.x, mypos.y, mypos.z, 0.f};
      reinterpret_cast<float4*>(s_xpq + kXpqStride * myb)[1] = {myquat.w, myquat.x, myquat.y, myquat.z};
    }
    gsync();
    for (int r = 0, span = 1; span < NLEVEL - 1 || r == 0; r++, span <<= 1) {
      const int anc = r == 0 ? myp : (r == 1 ? myanc2 : (r == 2 ? myanc4 : myanc8));
      float4 pp4 = {0.f, 0.f, 0.f, 0.f}, pq4 = {1.f, 0.f, 0.f, 0.f};
      if (bl) { const float4* Pp = reinterpret_cast<const float4*>(s_xpq + kXpqStride * anc); pp4 = Pp[0]; pq4 = Pp[1]; }
      gsync();
      if (bl && anc != 0) {
        const Q4 pq = {pq4.x, pq4.y, pq4.z, pq4.w};
        mypos = V3{pp4.x, pp4.y, pp4.z} + qrot(pq, mypos);
        myquat = qnormalize(qmul(pq, myquat));
        reinterpret_cast<float4*>(s_xpq + kXpqStride * myb)[0] = {mypos.x, mypos.y, mypos.z, 0.f};
        reinterpret_cast<float4*>(s_xpq + kXpqStride * myb)[1] = {myquat.w, myquat.x, myquat.y, myquat.z};
      }
      gsync();
    }
    if (bl) {
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
    if (geoml) {
      const int g = l, b = pf_gbody;
      st3(s_gpos + 3 * g, ld3(s_xpq + kXpqStride * b) + mrot(s_xmat + 9 * b, pf_gpos));
      Q4 q = qmul(ldq(s_xpq + kXpqStride * b + 4), pf_gquat);
      st3(s_gaxis + 3 * g, {2.f * (q.x * q.z + q.w * q.y), 2.f * (q.y * q.z - q.w * q.x), q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z});
    }
    // ---------------------------------------------------------------- mj_comPos (one kinematic tree)
    {
      V3 acc = {0.f, 0.f, 0.f};
      if (bl) acc = ld3(s_xipos + 3 * myb) * mymass;
      const float im = M.tree_invmass[0];
      float xl, xh, yl, yh, zl, zh;
      half_sums(acc.x, xl, xh); half_sums(acc.y, yl, yh); half_sums(acc.z, zl, zh);
      const float sx = (h ? xh : xl) * im, sy = (h ? yh : yl) * im, sz = (h ? zh : zl) * im;
      if (on && l == 0) st3(s_scom, {sx, sy, sz});
    }
    gsync();
    if (bl) {
      const float4 HB_CONST* R = M.brec + (size_t)(l + 1) * kBrecQuads;
      const float4 iq = R[5], in4 = R[6];
      const int b = myb;
      V3 com = ld3(s_scom);
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
    if (on && l < 10) s_cinert[l] = 0.f;
    if (dofl) {
      const int d = l;
      const float4 dA = pf_dA;
      const int j = __float_as_int(dA.x), b = __float_as_int(dA.y), type = __float_as_int(dA.z), k = __float_as_int(dA.w);
      V3 off = ld3(s_scom) - ld3(s_xanchor + 3 * j);
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
    if (on && l < NT) {
      const int t = l;
      const float4 tc = M.trec[3 * t], tq = M.trec[3 * t + 1];
      float len = tc.x * s_qpos[__float_as_int(tq.x)] + tc.y * s_qpos[__float_as_int(tq.y)] + tc.z * s_qpos[__float_as_int(tq.z)] + tc.w * s_qpos[__float_as_int(tq.w)];
      const int nw = M.tendon_num[t];
      for (int w = 4; w < nw; w++) len += M.wrap_prm[M.tendon_adr[t] + w] * s_qpos[M.wrap_qposadr[M.tendon_adr[t] + w]];
      s_tenlen[t] = len;
    }
    gsync();
    HB_STAMP(3);
    // ---------------------------------------------------------------- mj_comVel + mj_rne forward pass
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
    float mycvel[6], mycacc[6];
    for (int i = 0; i < 6; i++) { mycvel[i] = lv[i]; mycacc[i] = la[i]; }
    if (bl) {
      float4* Op = reinterpret_cast<float4*>(s_va + 12 * myb);
      Op[0] = {mycvel[0], mycvel[1], mycvel[2], mycvel[3]};
      Op[1] = {mycvel[4], mycvel[5], mycacc[0], mycacc[1]};
      Op[2] = {mycacc[2], mycacc[3], mycacc[4], mycacc[5]};
    }
    gsync();
    for (int r = 0, span = 1; span < NLEVEL - 1 || r == 0; r++, span <<= 1) {
      const int anc = r == 0 ? myp : (r == 1 ? myanc2 : (r == 2 ? myanc4 : myanc8));
      float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
      if (bl && anc != 0) { const float4* Pp = reinterpret_cast<const float4*>(s_va + 12 * anc); a0 = Pp[0]; a1 = Pp[1]; a2 = Pp[2]; }
      gsync();
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
    for (int i = 0; i < 3; i++) mycacc[3 + i] -= M.gravity[i];  // the world's cacc
    // body-local force cinert cacc + cvel x* (cinert cvel), and the composite inertia seeds
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
    if (on && l < 16) s_if[l] = l < 10 ? s_cinert[l] : 0.f;  // world body
    int pf_pk = 0;
    float2 pf_ad = {0.f, 0.f};
    if (on) { pf_pk = M.mrec[l]; pf_ad = M.mdiag[l]; }
    gsync();
    // mj_crb and the mj_rne backward pass: one sweep up the tree, children into parents (pull form)
    for (int L = NLEVEL - 2; L >= 1; L--) {
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
    float4 pf_bA = {0.f, 0.f, 0.f, 0.f}, pf_bB = pf_bA, pf_bC = pf_bA, pf_a0 = pf_bA, pf_a1 = pf_bA, pf_a2 = pf_bA, pf_a3 = pf_bA;
    if (dofl) { pf_bA = M.drec[3 * l]; pf_bB = M.drec[3 * l + 1]; pf_bC = M.drec[3 * l + 2]; }
    if (on && l < NU) { const float4 HB_CONST* AR4 = M.arec + (size_t)l * 4; pf_a0 = AR4[0]; pf_a1 = AR4[1]; pf_a2 = AR4[2]; pf_a3 = AR4[3]; }
    if (on) {
      for (int e = l; e < NM; e += H) {
        const int pk = pf_pk;
        const float2 ad = pf_ad;
        if (e + H < NM) { pf_pk = M.mrec[e + H]; pf_ad = M.mdiag[e + H]; }
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
        sacc += ad.x;
        // H = M + h diag(damping) differs from M on the diagonal only: the one-env kernel's pair {M, H}, the H half kept for diagonal entries
        const f32x2 mh = {sacc, sacc + (eulerdamp ? M.timestep * ad.y : 0.f)};
        s_qM[e] = mh.x;
        if (i == j) s_Hd[i] = mh.y;
      }
    }
    gsync();
    HB_STAMP(5);
    // ---------------------------------------------------------------- qfrc_bias, mj_passive, mj_fwdActuation -> qfrc_smooth
    if (dofl) {
      const int d = l;
      const float4 dA = pf_bA, dB = pf_bB, dC = pf_bC;
      float bias = 0.f;
      const int b = __float_as_int(dA.y);
      float cdd[6];
      ld_cdof(s_cdof, d, cdd);
      for (int t = 0; t < 6; t++) bias += cdd[t] * s_if[kIfStride * b + 10 + t];
      float passive = 0.f;
      if (__float_as_int(dA.z) >= 2) passive -= dB.w * (s_qpos[__float_as_int(dC.x)] - dC.y);
      passive -= dB.z * s_qvel[d];
      s_smooth[d] = passive - bias;
    }
    gsync();
    if (on && l < NU) {
      const int a = l;
      const int qa = __float_as_int(pf_a0.z), da = __float_as_int(pf_a0.w);
      float ctrl = s_ctrl[a];
      if (__float_as_int(pf_a0.x)) ctrl = clampf(ctrl, pf_a1.x, pf_a1.y);
      float gear = pf_a1.z;
      const float gain = pf_a1.w, bias1 = pf_a2.y;
      float force = gain * ctrl + pf_a2.x + bias1 * gear * s_qpos[qa] + pf_a2.z * gear * s_qvel[da];
      if (__float_as_int(pf_a0.y)) force = clampf(force, pf_a3.x, pf_a3.y);
      atomicAdd(&s_smooth[da], gear * force);
    }
    gsync();

    HB_STAMP(6);
    // ================================================================ constraint arrays from here on (alias the dynamics scratch)
    // contact record s of the lane's env: slots 0..11 in its own block, 12..23 (an env stepped alone) in the other env's
    auto con_rec = [&](int s) -> float* { return (s < kNCh ? E : lds + (1 - h) * kEnvF) + o_con + (s < kNCh ? s : s - kNCh) * kConStride; };
    // ---------------------------------------------------------------- mj_collision
    int ncon = 0;  // per lane = per env
    {
      int* s_list = reinterpret_cast<int*>(s_C) + h * 160;
      int nlist = 0;
      {
        float4 n0 = M.crec[3 * (size_t)l], n1 = M.crec[3 * (size_t)l + 1];
        for (int p0 = 0; p0 < NPAIR; p0 += H) {
          const int p = p0 + l;
          const float4 c0 = n0, c1 = n1;
          if (p0 + H < NPAIR) { const float4 HB_CONST* N = M.crec + 3 * (size_t)(p + H); n0 = N[0]; n1 = N[1]; }
          bool pass = false;
          if (on && p < NPAIR) {
            const int g1 = __float_as_int(c0.x), g2 = __float_as_int(c0.y), t1 = __float_as_int(c0.z) & 255;
            const V3 dp = ld3(s_gpos + 3 * g2) - ld3(s_gpos + 3 * g1);
            if (t1 == 0) pass = dot(dp, ld3(s_gaxis + 3 * g1)) <= c0.w + c1.y;
            else { const float bound = c1.x + c1.y + c0.w; pass = dot(dp, dp) <= bound * bound; }
          }
          const unsigned mine = (unsigned)(__ballot(pass) >> (32 * h));
          if (pass) s_list[nlist + __popc(mine & ((1u << l) - 1u))] = p;
          nlist += __popc(mine);
        }
      }
      gsync();
      const int nlist_max = max(__builtin_amdgcn_readlane(nlist, 0), __builtin_amdgcn_readlane(nlist, 32));
      for (int i0 = 0; i0 < nlist_max; i0 += H) {
        const bool have = i0 + l < nlist;
        const int p = have ? s_list[i0 + l] : 0;
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
        // ordered append inside the env: slot = ncon + (# contacts of lower lanes of the half)
        const unsigned b1 = (unsigned)(__ballot(n >= 1) >> (32 * h)), b2 = (unsigned)(__ballot(n >= 2) >> (32 * h));
        const unsigned lt = (1u << l) - 1u;
        const int slot = ncon + __popc(b1 & lt) + __popc(b2 & lt);
        if (n >= 1 && slot < ncap) {
          float* c = con_rec(slot);
          c[C_DIST] = co0.dist;
          st3(c + C_POS, co0.pos);
          make_frame(c + C_FRAME, co0.n, hint);
          c[C_PAIR] = __int_as_float(p);
        }
        if (n >= 2 && slot + 1 < ncap) {
          float* c = con_rec(slot + 1);
          c[C_DIST] = co1.dist;
          st3(c + C_POS, co1.pos);
          make_frame(c + C_FRAME, co1.n, hint);
          c[C_PAIR] = __int_as_float(p);
        }
        ncon += __popc(b1) + __popc(b2);
      }
    }
    if (both) {
      // more contacts in one of the envs than two envs to a wave hold: one env at a time (nothing of this pass has left the wave)
      if (__ballot(ncon > kNCh)) { serial = true; continue; }
    } else if (ncon > kNconMax) { status |= (1 << 1); ncon = kNconMax; }
    gsync();
    // self collision (CPUEnv._check_self_collision, cpu_env.py:576-584)
    int pairid = 0;
    if (l < ncon) pairid = __float_as_int(con_rec(l)[C_PAIR]);
    bool selfc = false;
    if (l < ncon) selfc = M.pair_self[pairid] != 0;

    HB_STAMP(7);
    // ---------------------------------------------------------------- mj_makeConstraint: count, place, write
    // (a) limit candidates: 2 per limited joint / tendon in constraint order, 32 per round
    constexpr int kLimRounds = (NLIM + H - 1) / H;
    bool lact[kLimRounds];
    float ldist[kLimRounds], lmargin[kLimRounds];
    int lrow[kLimRounds];
    int nefc = 0;  // rows of the lane's env
#pragma unroll
    for (int k = 0; k < kLimRounds; k++) {
      const int c = l + H * k;
      lact[k] = false; ldist[k] = 0.f; lmargin[k] = 0.f;
      if (on && c < NLIM) {
        const float4 HB_CONST* LR = M.lrec + (size_t)c * 4;
        const float4 r0 = LR[0], r1 = LR[1];
        const int kind = __float_as_int(r0.x), id = __float_as_int(r0.y), side = __float_as_int(r0.z);
        lmargin[k] = r1.x;
        const float value = kind == 0 ? s_qpos[__float_as_int(r0.w)] : s_tenlen[id];
        ldist[k] = (float)side * (r1.y - value);
        lact[k] = ldist[k] < lmargin[k];
      }
      const unsigned mine = (unsigned)(__ballot(lact[k]) >> (32 * h));
      lrow[k] = nefc + __popc(mine & ((1u << l) - 1u));
      nefc += __popc(mine);
    }
    // (b) contacts: 0, 1 or 4 rows each
    int myrows = 0;
    bool incl = false;
    float4 pr1 = {0.f, 0.f, 0.f, 0.f}, pr2 = pr1, pr3 = pr1, pr4 = pr1;
    if (l < ncon) {
      const float* c = con_rec(l);
      const float includemargin = M.pair_margin[pairid] - M.pair_gap[pairid];
      incl = c[C_DIST] < includemargin;
      myrows = incl ? (M.pair_dim[pairid] == 1 ? 1 : 4) : 0;
      const float4 HB_CONST* PR = M.prec + (size_t)pairid * 5;
      pr1 = PR[1]; pr2 = PR[2]; pr3 = PR[3]; pr4 = PR[4];
    }
    const unsigned lower = (1u << l) - 1u;
    const unsigned one_row = (unsigned)(__ballot(myrows == 1) >> (32 * h)), four_rows = (unsigned)(__ballot(myrows == 4) >> (32 * h));
    int base = nefc + __popc(one_row & lower) + 4 * __popc(four_rows & lower);  // first row of the lane's contact inside its env
    const int nefc_all = nefc + __popc(one_row) + 4 * __popc(four_rows);        // rows the env asks for
    // placement of the envs' rows on the wave's 64 row lanes
    const int nA = __builtin_amdgcn_readlane(nefc_all, 0), nB = __builtin_amdgcn_readlane(nefc_all, 32);
    int first = 0, split = 64;  // env `first` owns row lanes [0, split), the other env [split, 64); one env alone: all of them
    if (both) {
      if (nA <= 31 && nB <= 31) { first = 0; split = 32; }
      else {
        first = nB > nA ? 1 : 0;
        const int nf = first ? nB : nA, ns = first ? nA : nB;
        split = (nf + 1 + 3) & ~3;
        if (split < 32) split = 32;
        if (split + ns + 1 > 64) { serial = true; continue; }  // the two envs' rows do not fit one wave: one env at a time
      }
    } else first = (act >> 1) & 1u;
    const int rbase = (both && h != first) ? split : 0;  // packed row of the env's row 0 (lane's env)
    bool fits = true;
    if (!both) {  // one env alone: the one-env kernel's capacity and overflow rules
      if (nefc > kNefcMax) { status |= (1 << 2); nefc = kNefcMax; }
      fits = base + myrows <= kNefcMax;
      if ((unsigned)(__ballot(l < ncon && incl && !fits) >> (32 * h))) status |= (1 << 2);
    }
    const bool placedl = l < ncon && incl && fits;
    // rows of the env after its contacts: the end of the last one placed
    const int endv = base + myrows;
    int nefc_after;
    {
      // (the lane index of the last placed contact differs per half: read both candidates and pick)
      const unsigned long long pl64 = __ballot(placedl);
      const unsigned pa = (unsigned)pl64, pb = (unsigned)(pl64 >> 32);
      const int ea = pa ? __builtin_amdgcn_readlane(endv, 31 - __clz(pa)) : __builtin_amdgcn_readlane(nefc, 0);
      const int eb = pb ? __builtin_amdgcn_readlane(endv, 32 + 31 - __clz(pb)) : __builtin_amdgcn_readlane(nefc, 32);
      nefc_after = h ? eb : ea;
    }
    // limit rows
#pragma unroll
    for (int k = 0; k < kLimRounds; k++) {
      if (lact[k] && lrow[k] < kNefcMax) {
        const int c = l + H * k;
        const float4 HB_CONST* LR = M.lrec + (size_t)c * 4;
        const float4 r0 = LR[0], r1 = LR[1], r2 = LR[2], r3 = LR[3];
        const int kind = __float_as_int(r0.x), id = __float_as_int(r0.y), side = __float_as_int(r0.z);
        const int prow = rbase + lrow[k];
        float* Jr = s_C + prow * kCsD;
        for (int kk = 0; kk < kCsD; kk++) Jr[kk] = 0.f;
        if (kind == 0) Jr[__float_as_int(r3.z)] = (float)(-side);
        else for (int w = 0; w < M.tendon_num[id]; w++) Jr[M.wrap_dofadr[M.tendon_adr[id] + w]] = (float)(-side) * M.wrap_prm[M.tendon_adr[id] + w];
        float* e = s_meta + prow;
        e[E_POS * 64] = ldist[k]; e[E_MARGIN * 64] = lmargin[k];
        e[E_SOLREF0 * 64] = r1.z; e[E_SOLREF1 * 64] = r1.w;
        e[(E_IMP0 + 0) * 64] = r2.x; e[(E_IMP0 + 1) * 64] = r2.y; e[(E_IMP0 + 2) * 64] = r2.z; e[(E_IMP0 + 3) * 64] = r2.w;
        e[(E_IMP0 + 4) * 64] = r3.x;
        e[E_DA * 64] = r3.y; e[E_MU2 * 64] = 0.f;
      }
    }
    // contact records: first row, dimension, friction, the two bodies' dof masks; and the rows' meta, written by the contact's own lane
    const int rowv = placedl ? base : -1;
    if (l < ncon) {
      float* c = con_rec(l);
      const int cdim = M.pair_dim[pairid] == 1 ? 1 : 3;
      const float mu = fmaxf(1e-5f, M.pair_friction[3 * pairid]);
      c[C_ROW] = __int_as_float(rowv);
      c[C_DIM] = __int_as_float(cdim);
      c[C_FRIC] = mu;
      c[C_M1] = pr1.x; c[C_M2] = pr1.z;  // (27 dofs: the low words of the masks)
      if (placedl) {
        const int dim = __float_as_int(pr4.y) == 1 ? 1 : 3;
        const float tran = pr2.w;
        float da = dim == 1 ? tran : tran + mu * mu * tran;
        float mus = mu * M.inv_sqrt_impratio;
        const float mu2 = dim == 1 ? 0.f : 2.f * mus * mus;
        const float dist = c[C_DIST];
        const int nr = dim == 1 ? 1 : 4;
        for (int r = 0; r < nr; r++) {
          float* e = s_meta + rbase + base + r;
          e[E_POS * 64] = dist;
          e[E_MARGIN * 64] = pr2.x;
          e[E_SOLREF0 * 64] = pr2.y; e[E_SOLREF1 * 64] = pr2.z;
          e[(E_IMP0 + 0) * 64] = pr3.x; e[(E_IMP0 + 1) * 64] = pr3.y; e[(E_IMP0 + 2) * 64] = pr3.z; e[(E_IMP0 + 3) * 64] = pr3.w;
          e[(E_IMP0 + 4) * 64] = pr4.x;
          e[E_DA * 64] = da;
          e[E_MU2 * 64] = mu2;
        }
      }
    }
    const int selfcol = ((unsigned)(__ballot(selfc) >> (32 * h))) ? 1 : 0;
    gsync();
    // Jacobian rows: loop over the contacts of both envs in step, lanes over dofs
    {
      const int ncon_max = max(__builtin_amdgcn_readlane(ncon, 0), __builtin_amdgcn_readlane(ncon, 32));
      for (int ci = 0; ci < ncon_max; ci++) {
        const int ra = __builtin_amdgcn_readlane(rowv, ci), rb = __builtin_amdgcn_readlane(rowv, 32 + ci);
        const int row = h ? rb : ra;
        if (row < 0 || ci >= ncon) continue;
        const float* c = con_rec(ci);
        const int dim = __float_as_int(c[C_DIM]);
        const unsigned m1 = __float_as_uint(c[C_M1]), m2 = __float_as_uint(c[C_M2]);
        V3 cpos = ld3(c + C_POS);
        V3 off1 = cpos - ld3(s_scom), off2 = off1;
        V3 fn = ld3(c + C_FRAME), ft1 = ld3(c + C_FRAME + 3), ft2 = ld3(c + C_FRAME + 6);
        float mu = c[C_FRIC];
        const int d = l;
        if (d < kCsD) {
          V3 jd = {0.f, 0.f, 0.f};
          if (d < NV) {
            float cdd[6];
            ld_cdof(s_cdof, d, cdd);
            V3 ang = {cdd[0], cdd[1], cdd[2]}, lin = {cdd[3], cdd[4], cdd[5]};
            if ((m2 >> d) & 1u) jd = jd + lin + cross(ang, off2);
            if ((m1 >> d) & 1u) jd = jd - (lin + cross(ang, off1));
          }
          float j0 = dot(fn, jd);
          float* Jr = s_C + (rbase + row) * kCsD + d;
          if (dim == 1) Jr[0] = j0;
          else {
            float j1 = mu * dot(ft1, jd), j2 = mu * dot(ft2, jd);
            Jr[0] = j0 + j1;
            Jr[kCsD] = j0 - j1;
            Jr[2 * kCsD] = j0 + j2;
            Jr[3 * kCsD] = j0 - j2;
          }
        }
      }
    }
    nefc = nefc_after;
    // the envs' row counts as scalars, and the packed rows in use
    const int n0 = __builtin_amdgcn_readlane(nefc, 0), n1 = __builtin_amdgcn_readlane(nefc, 32);
    const int nLo = both ? (first ? n1 : n0) : ((act & 1u) ? n0 : n1);  // rows on lanes [0, split)
    const int nHi = both ? (first ? n0 : n1) : 0;                        // rows on lanes [split, 64)
    const int envLo = both ? first : (int)((act >> 1) & 1u);             // which env owns the low row lanes
    // extra right-hand side: the row behind an env's rows holds its qfrc_smooth (transformed below, with the rows, into y)
    if (on && l < kCsD) s_C[(rbase + nefc) * kCsD + l] = l < NV ? s_smooth[l] : 0.f;
    gsync();

    HB_STAMP(8);
    // ---------------------------------------------------------------- per-row quantities (lane = packed row)
    const bool lolane = lane < split;
    const int renv = lolane ? envLo : 1 - envLo;     // the env of this row lane
    const int ridx = lolane ? lane : lane - split;   // the row inside its env
    const bool rowact = ridx < (lolane ? nLo : nHi);
    float* const RE = lds + renv * kEnvF;            // that env's block
    const int ysrow = (lolane ? 0 : split) + (lolane ? nLo : nHi);  // packed row of that env's qfrc_smooth
    float R = 1.f, Dd = 1.f, aref = 0.f, jw = 0.f, force = 0.f, bvec = 0.f;
    if (rowact) {
      const float* Jr = s_C + lane * kCsD;
      const float* r_qvel = RE + o_qvel;
      const float* r_warm = RE + o_warm;
      float vel = 0.f, jw_ = 0.f;
      {
        float vel2 = 0.f, jw2 = 0.f;
        int k = 0;
        for (; k + 4 <= NV; k += 4) {
          const float j0 = Jr[k], j1 = Jr[k + 1], j2 = Jr[k + 2], j3 = Jr[k + 3];
          vel += j0 * r_qvel[k] + j2 * r_qvel[k + 2]; vel2 += j1 * r_qvel[k + 1] + j3 * r_qvel[k + 3];
          jw_ += j0 * r_warm[k] + j2 * r_warm[k + 2]; jw2 += j1 * r_warm[k + 1] + j3 * r_warm[k + 3];
        }
        for (; k < NV; k++) { const float j = Jr[k]; vel += j * r_qvel[k]; jw_ += j * r_warm[k]; }
        vel += vel2; jw_ += jw2;
      }
      jw = jw_;
      const float* e = s_meta + lane;
      float pos = e[E_POS * 64], margin = e[E_MARGIN * 64];
      float solref0 = e[E_SOLREF0 * 64], solref1 = e[E_SOLREF1 * 64];
      float solimp[5];
      for (int i = 0; i < 5; i++) solimp[i] = e[(E_IMP0 + i) * 64];
      float imp = clampf(impedance(solimp, pos, margin), HB_MINIMP, HB_MAXIMP);
      float mu2 = e[E_MU2 * 64];
      float Rown = fmaxf(HB_MINVAL, (1.f - imp) * e[E_DA * 64] / imp);
      R = mu2 > 0.f ? mu2 * Rown : Rown;
      Dd = 1.f / R;
      float K, Bc;
      kb_from_solref(solref0, solref1, solimp[1], M.timestep, true, K, Bc);
      aref = -Bc * vel - K * imp * (pos - margin);
    }
    gsync();
    HB_STAMP(9);
    // ---------------------------------------------------------------- W = L^-1 D^-1/2 per env; C = J W (rows and the qfrc_smooth rows)
    {
      int lw;
      asm volatile("v_mov_b32 %0, %1" : "=v"(lw) : "v"(lane0));
      if (both) {  // the two eliminations side by side: each fills the other's pivot latencies
        f32x16 Xa, Xb, Ta, Sa, Tb, Sb;
        load_sym_env2<0>(M, lds + o_qM, lds + o_Hd, lds + kEnvF + o_qM, lds + kEnvF + o_Hd, lw, Xa, Xb);
        sym_factor_mfma_x2<14>(Xa, Xb, Ta, Sa, Tb, Sb, lw);
        store_w28(lds + o_W, Ta, Sa, lw);
        store_w28(lds + kEnvF + o_W, Tb, Sb, lw);
      } else {
        float* EE = lds + ((act >> 1) & 1u) * kEnvF;
        f32x16 T, S;
        sym_factor_mfma<14>(load_sym_env<0>(M, EE + o_qM, EE + o_Hd, lw), T, S, lw);
        store_w28(EE + o_W, T, S, lw);
      }
    }
    gsync();
    HB_STAMP(10);
    // packed rows of env e (incl. its qfrc_smooth row): [lo_e, hi_e]
    const int pLo0 = 0, pLo1 = nLo, pHi0 = split, pHi1 = split + nHi;  // low env: [0, nLo]; high env: [split, split + nHi]
    {
      const int col = lane & 31, half = lane >> 5;
      const int last_row = both ? pHi1 : pLo1;
#pragma unroll
      for (int I = 0; I < 2; I++) {
        if (I == 0 || last_row >= 32) {
          const int arow = 32 * I + col;
          const float* Ap = s_C + arow * kCsD + half;
          f32x16 D;
#pragma unroll
          for (int r = 0; r < 16; r++) D[r] = 0.f;
          // the low env's rows of this tile, then the high env's, each against its own W
          const bool tLo = 32 * I <= pLo1, tHi = both && 32 * I + 31 >= pHi0;
          if (tLo) {
            const bool av = arow >= pLo0 && arow <= pLo1;
            const float* Wp = lds + envLo * kEnvF + o_W;
            float a[14];
#pragma unroll
            for (int kk = 0; kk < 14; kk++) a[kk] = av ? Ap[2 * kk] : 0.f;
#pragma unroll
            for (int kk = 0; kk < 14; kk++) D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], Wp[(2 * kk + half) * kWsD + col], D, 0, 0, 0);
          }
          if (tHi) {
            const bool av = arow >= pHi0 && arow <= pHi1;
            const float* Wp = lds + (1 - envLo) * kEnvF + o_W;
            float a[14];
#pragma unroll
            for (int kk = 0; kk < 14; kk++) a[kk] = av ? Ap[2 * kk] : 0.f;
#pragma unroll
            for (int kk = 0; kk < 14; kk++) D = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], Wp[(2 * kk + half) * kWsD + col], D, 0, 0, 0);
          }
          if (col < 28) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
              const int row = 32 * I + (r & 3) + 8 * (r >> 2) + 4 * half;
              if ((row >= pLo0 && row <= pLo1) || (both && row >= pHi0 && row <= pHi1)) s_C[row * kCsD + col] = D[r];
            }
          }
        }
      }
    }
    gsync();
    HB_STAMP(11);
    // ---------------------------------------------------------------- efc_b, AR = C C^T + diag(R) (mj_projectConstraint), block diagonal over the envs
    const float* yv = s_C + ysrow * kCsD;
    float ar[kNefcMax];
    float Aii = 1.f;
    {
      const float* Cr = s_C + lane * kCsD;
      float jas = 0.f, diag = 0.f;
#pragma unroll
      for (int k = 0; k < 28; k++) {
        const float c = rowact ? Cr[k] : 0.f;
        jas += c * yv[k];
        diag += c * c;
      }
      bvec = jas - aref;
      Aii = rowact ? diag + R : 1.f;
      const int col = lane & 31, half = lane >> 5;
      const bool two = (both ? pHi1 : pLo1) > 32;  // rows 32.. in use: the second row tile
      // constraint rows (without the qfrc_smooth rows) of the low / the high env in tile 0 (rows col) and tile 1 (rows 32 + col)
      const bool lo0 = col < nLo, lo1 = 32 + col < nLo;
      const bool hi0 = both && col >= split && col < split + nHi, hi1 = both && 32 + col >= split && 32 + col < split + nHi;
      const float* A0p = s_C + col * kCsD + half;
      const float* A1p = s_C + (32 + col) * kCsD + half;
      const u32x2 rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(R), __float_as_uint(R), false, false);
      const float R0 = __uint_as_float(rr.x), R1 = __uint_as_float(rr.y);
      f32x16 X0, Y0, X1, Y1;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        X0[r] = (row == col && (lo0 || hi0)) ? R0 : 0.f;
        Y1[r] = (row == col && (lo1 || hi1)) ? R1 : 0.f;
        Y0[r] = 0.f;
        X1[r] = 0.f;
      }
      // low env
      {
#pragma unroll
        for (int kk = 0; kk < 14; kk++) {
          const float a0 = lo0 ? A0p[2 * kk] : 0.f;
          X0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, X0, 0, 0, 0);
        }
        if (nLo > 32) {
#pragma unroll
          for (int kk = 0; kk < 14; kk++) {
            const float a0 = lo0 ? A0p[2 * kk] : 0.f;
            const float a1 = lo1 ? A1p[2 * kk] : 0.f;
            Y0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, Y0, 0, 0, 0);
            X1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a0, X1, 0, 0, 0);
            Y1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, Y1, 0, 0, 0);
          }
        }
      }
      // high env: always inside tile 1 (split >= 32)
      if (both && nHi > 0) {
#pragma unroll
        for (int kk = 0; kk < 14; kk++) {
          const float a1 = hi1 ? A1p[2 * kk] : 0.f;
          Y1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, Y1, 0, 0, 0);
        }
      }
      (void)two; (void)hi0;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ra = (r & 3) + 8 * (r >> 2);
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(X0[r]), __float_as_uint(Y0[r]), false, false);
        ar[ra] = __uint_as_float(s0.x);
        ar[ra + 4] = __uint_as_float(s0.y);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(X1[r]), __float_as_uint(Y1[r]), false, false);
        ar[32 + ra] = __uint_as_float(s1.x);
        if (32 + ra + 4 < kNefcMax) ar[32 + ra + 4] = __uint_as_float(s1.y);
      }
    }
    HB_STAMP(12);
    // ---------------------------------------------------------------- mj_fwdConstraint: warm start + PGS over the packed rows
    // sums over the rows of each env: the one-env kernel's wave_sum on the lanes that env's rows would have there
    const bool aligned = !both || split == 32;
    auto env_sums = [&](float v, float& sLo, float& sHi) {
      if (aligned) {
        float a, b;
        half_sums(v, a, b);
        if (both) { sLo = a; sHi = b; }
        else { sLo = a + b; sHi = 0.f; }
      } else {
        // the low env's rows reach beyond lane 31 and the high env's start behind them: the low env's sum over its own lanes, the high
        // env's rows rotated down to lane 0 first (ds_bpermute) so that the tree adds them in the one-env kernel's order
        float a, b;
        half_sums(lolane ? v : 0.f, a, b);
        sLo = a + b;
        const float w = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + split) & 63) << 2, __float_as_int(v)));
        half_sums(lane < nHi ? w : 0.f, a, b);
        sHi = a + b;
      }
    };
    int niterLo = 0, niterHi = 0;
    if (both && split == 32) {
      // Both envs at most 31 rows, env A's on lanes 0.., env B's on lanes 32..: the two Gauss-Seidel sweeps run SIDE BY SIDE - row i of
      // both envs in one row step (every lane proposes; the turn-holders of the two halves are read out by two v_readlane, each half
      // takes its own).  The two dependency chains share their instructions: one row step serves two envs.  Per env the arithmetic,
      // and its order, is the one-env sweep's.
      float arS[31];  // the lane's AR column over the rows of its own env
#pragma unroll
      for (int i = 0; i < 31; i++) arS[i] = h ? ar[32 + i] : ar[i];
      const float nAinv = -1.f / Aii;
      float arf = 0.f;
      const int nmax2 = max(nLo, nHi);
      {
        const float jar = jw - aref;
        force = (rowact && jar < 0.f) ? -Dd * jar : 0.f;
#pragma unroll
        for (int c = 0; c < 8; c++) {
          if (c * 4 < nmax2) {
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (c * 4 + r < 31) { const float fa = rdlane(force, c * 4 + r), fb = rdlane(force, 32 + c * 4 + r); arf += arS[c * 4 + r] * (h ? fb : fa); }
          }
        }
        float cLo, cHi;
        env_sums(rowact ? force * (0.5f * arf + bvec) : 0.f, cLo, cHi);
        const float cost = h ? cHi : cLo;
        if (cost > 0.f) { force = 0.f; arf = 0.f; }
      }
      float res = rowact ? bvec + arf : 0.f;
      const int max_sweeps = M.iterations;
      const float pgs_tol = M.tolerance, pgs_scale = M.pgs_scale;
      bool liveLo = nLo > 0 && max_sweeps > 0, liveHi = nHi > 0 && max_sweeps > 0;
      const bool sweptLo = liveLo, sweptHi = liveHi;
      float force_out = force;
      // while BOTH envs sweep: one row step = row i of both (7 VALU: mul, max, two v_readlane, the half's pick, fma, and ONE v_cndmask under the
      // mask {i, 32 + i} that keeps the two turn-holders' steps - they are the d_ of those very lanes)
      while (liveLo && liveHi) {
        int ne;
        {
          const int a = max(nLo, nHi);
          asm volatile("s_mov_b32 %0, %1" : "=s"(ne) : "s"(a));
        }
        const float nforce = -force, res0 = res;
        float dl = 0.f;
        unsigned long long turn_ = 0x0000000100000001ull;  // the lanes whose turn it is: row 0 of both envs
#define HB_PGS_ROW2(i)                                                               \
  if ((i) < 31) {                                                                    \
    const float d_ = fmaxf(res * nAinv, nforce);                                     \
    int da_, db_;                                                                    \
    unsigned long long ex_;                                                          \
    /* res += arS[i] * (the step of row i of the lane's own env): the two steps stay in scalar registers and each half takes its own */ \
    /* under its half of EXEC (a v_cndmask between two scalars would cost two more VALU moves: one scalar operand per instruction). */ \
    /* The turn-holders' mask {i, 32 + i} walks up one bit per row (thirty-one 64-bit constants would live in scalar registers, and */ \
    /* spill).  One asm statement, ordered so that no VALU instruction reads a scalar register within two instructions of the */      \
    /* statement's start or of the v_readlane that wrote it: the compiler's hazard recogniser does not look inside. */                 \
    asm volatile("s_mov_b64 %[t], exec\n\t"                                          \
                 "v_readlane_b32 %[a], %[d], %[ia]\n\t"                              \
                 "v_readlane_b32 %[b], %[d], %[ib]\n\t"                              \
                 "v_cndmask_b32_e64 %[dl], %[dl], %[d], %[turn]\n\t"                 \
                 "s_lshl_b64 %[turn], %[turn], 1\n\t"                                \
                 "s_mov_b64 exec, %[lo]\n\t"                                         \
                 "v_fmac_f32_e32 %[r], %[a], %[ar]\n\t"                              \
                 "s_mov_b64 exec, %[hi]\n\t"                                         \
                 "v_fmac_f32_e32 %[r], %[b], %[ar]\n\t"                              \
                 "s_mov_b64 exec, %[t]"                                               \
                 : [r] "+v"(res), [dl] "+v"(dl), [turn] "+s"(turn_), [a] "=&s"(da_), [b] "=&s"(db_), [t] "=&s"(ex_) \
                 : [d] "v"(d_), [ar] "v"(arS[(i) < 31 ? (i) : 0]), [ia] "n"(i), [ib] "n"(32 + (i)), [lo] "s"(0x00000000ffffffffull), [hi] "s"(0xffffffff00000000ull) \
                 : "scc");                                                           \
  }
#define HB_PGS_CHUNK2(c) if ((c) * 4 >= ne) break; HB_PGS_ROW2((c) * 4) HB_PGS_ROW2((c) * 4 + 1) HB_PGS_ROW2((c) * 4 + 2) HB_PGS_ROW2((c) * 4 + 3)
        do {
          HB_PGS_CHUNK2(0) HB_PGS_CHUNK2(1) HB_PGS_CHUNK2(2) HB_PGS_CHUNK2(3) HB_PGS_CHUNK2(4) HB_PGS_CHUNK2(5) HB_PGS_CHUNK2(6) HB_PGS_CHUNK2(7)
        } while (0);
#undef HB_PGS_CHUNK2
#undef HB_PGS_ROW2
        const float delta = dl;
        force += delta;
        float iLo, iHi;
        env_sums(delta * (res0 + res), iLo, iHi);
        niterLo++; niterHi++;
        if (-0.5f * iLo * pgs_scale < pgs_tol || niterLo >= max_sweeps) {
          liveLo = false;
          if (!h) { force_out = force; res = 0.f; force = 0.f; }  // rows of a finished env are inert from here on
        }
        if (-0.5f * iHi * pgs_scale < pgs_tol || niterHi >= max_sweeps) {
          liveHi = false;
          if (h) { force_out = force; res = 0.f; force = 0.f; }
        }
      }
      // ONE env still sweeps (the wave sweeps max(sA, sB) times: 29 against a mean of 20 per env): the one-env kernel's row step (5 VALU) on that
      // env's lanes.  The other half's lanes ride along: their res takes garbage, their step stays 0 and their forces are in force_out.
#define HB_PGS_ROW1(base, i)                                                         \
  if ((i) < 31) {                                                                    \
    const float d_ = fmaxf(res * nAinv, nforce);                                     \
    const int di_ = __builtin_amdgcn_readlane(__float_as_int(d_), (base) + (i));     \
    res = __builtin_fmaf(arS[(i) < 31 ? (i) : 0], __int_as_float(di_), res);         \
    dl = hb_writelane(di_, (base) + (i), dl);                                        \
  }
#define HB_PGS_CHUNK1(base, c) if ((c) * 4 >= ne) break; HB_PGS_ROW1(base, (c) * 4) HB_PGS_ROW1(base, (c) * 4 + 1) HB_PGS_ROW1(base, (c) * 4 + 2) HB_PGS_ROW1(base, (c) * 4 + 3)
#define HB_PGS_TAIL(base, live, n_, niter_, mine)                                    \
      while (live) {                                                                 \
        int ne;                                                                      \
        asm volatile("s_mov_b32 %0, %1" : "=s"(ne) : "s"(n_));                       \
        const float nforce = -force, res0 = res;                                     \
        int dl = 0;                                                                  \
        do {                                                                         \
          HB_PGS_CHUNK1(base, 0) HB_PGS_CHUNK1(base, 1) HB_PGS_CHUNK1(base, 2) HB_PGS_CHUNK1(base, 3) HB_PGS_CHUNK1(base, 4) HB_PGS_CHUNK1(base, 5) HB_PGS_CHUNK1(base, 6) HB_PGS_CHUNK1(base, 7) \
        } while (0);                                                                 \
        const float delta = __int_as_float(dl);                                      \
        force += delta;                                                              \
        float iLo, iHi;                                                              \
        env_sums((mine) ? delta * (res0 + res) : 0.f, iLo, iHi);                     \
        niter_++;                                                                    \
        if (-0.5f * ((base) ? iHi : iLo) * pgs_scale < pgs_tol || niter_ >= max_sweeps) { \
          live = false;                                                              \
          if (mine) force_out = force;                                               \
        }                                                                            \
      }
      HB_PGS_TAIL(0, liveLo, nLo, niterLo, !h)
      HB_PGS_TAIL(32, liveHi, nHi, niterHi, h)
#undef HB_PGS_TAIL
#undef HB_PGS_CHUNK1
#undef HB_PGS_ROW1
      force = (h ? sweptHi : sweptLo) ? force_out : force;
    } else {
      const float nAinv = -1.f / Aii;
      float arf = 0.f;
      const int ne_all = both ? (nHi > 0 ? pHi1 : nLo) : nLo;  // packed rows in use: [0, ne_all)
      {
        const float jar = jw - aref;
        force = (rowact && jar < 0.f) ? -Dd * jar : 0.f;
#pragma unroll
        for (int c = 0; c < (kNefcMax + 3) / 4; c++) {
          if (c * 4 < ne_all) {
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (c * 4 + r < kNefcMax) arf += ar[c * 4 + r] * rdlane(force, c * 4 + r);
          }
        }
        float cLo, cHi;
        env_sums(rowact ? force * (0.5f * arf + bvec) : 0.f, cLo, cHi);
        const float cost = lolane ? cLo : cHi;
        if (cost > 0.f) { force = 0.f; arf = 0.f; }
      }
      float res = rowact ? bvec + arf : 0.f;
      const int max_sweeps = M.iterations;
      const float pgs_tol = M.tolerance, pgs_scale = M.pgs_scale;
      bool liveLo = nLo > 0 && max_sweeps > 0, liveHi = nHi > 0 && max_sweeps > 0;
      float force_out = force;  // the env's forces when its sweeps stopped
      while (liveLo || liveHi) {
        // rows of the low env still sweeping: [0, neLo); the sweep then goes on from row 32 up to neEnd (the high env's rows, or what is
        // left of a low env above 31 rows)
        int neLo, neEnd;
        {
          const int a = liveLo ? nLo : 0, b = liveHi ? pHi1 : (liveLo ? nLo : 0);
          asm volatile("s_mov_b32 %0, %1" : "=s"(neLo) : "s"(a));
          asm volatile("s_mov_b32 %0, %1" : "=s"(neEnd) : "s"(b));
        }
        const float nforce = -force, res0 = res;
        int dl = 0;
#define HB_PGS_ROW(i)                                                              \
  if ((i) < kNefcMax) {                                                            \
    const float d_ = fmaxf(res * nAinv, nforce);                                   \
    const int di_ = __builtin_amdgcn_readlane(__float_as_int(d_), (i));            \
    res = __builtin_fmaf(ar[(i) < kNefcMax ? (i) : 0], __int_as_float(di_), res);  \
    dl = hb_writelane(di_, (i), dl);                                               \
  }
#define HB_PGS_ROWS(c) HB_PGS_ROW((c) * 4) HB_PGS_ROW((c) * 4 + 1) HB_PGS_ROW((c) * 4 + 2) HB_PGS_ROW((c) * 4 + 3)
#define HB_PGS_LO(c) if ((c) * 4 >= neLo) goto hb_pgs_upper; HB_PGS_ROWS(c)
#define HB_PGS_UP(c) if ((c) * 4 >= neEnd) break; HB_PGS_ROWS(c)
        do {
          HB_PGS_LO(0) HB_PGS_LO(1) HB_PGS_LO(2) HB_PGS_LO(3) HB_PGS_LO(4) HB_PGS_LO(5) HB_PGS_LO(6) HB_PGS_LO(7)
        hb_pgs_upper:
          HB_PGS_UP(8) HB_PGS_UP(9) HB_PGS_UP(10) HB_PGS_UP(11) HB_PGS_UP(12) HB_PGS_UP(13) HB_PGS_UP(14) HB_PGS_UP(15)
        } while (0);
#undef HB_PGS_UP
#undef HB_PGS_LO
#undef HB_PGS_ROWS
#undef HB_PGS_ROW
        const float delta = __int_as_float(dl);
        force += delta;
        float iLo, iHi;
        env_sums(delta * (res0 + res), iLo, iHi);
        if (liveLo) {
          niterLo++;
          if (-0.5f * iLo * pgs_scale < pgs_tol || niterLo >= max_sweeps) {
            liveLo = false;
            if (lolane) { force_out = force; res = 0.f; force = 0.f; }  // rows of a finished env are inert from here on
          }
        }
        if (liveHi) {
          niterHi++;
          if (-0.5f * iHi * pgs_scale < pgs_tol || niterHi >= max_sweeps) {
            liveHi = false;
            if (!lolane) { force_out = force; res = 0.f; force = 0.f; }
          }
        }
      }
      // (an env without rows, or without sweeps, keeps its warm start: force_out was set from it)
      const bool sweptLo = nLo > 0 && max_sweeps > 0, sweptHi = nHi > 0 && max_sweeps > 0;
      force = (lolane ? sweptLo : sweptHi) ? force_out : force;
    }
    const int niter = (h == envLo) ? niterLo : niterHi;  // per lane = per env
    HB_STAMP(13);
    // ---------------------------------------------------------------- dual finish: s = sum_i f_i C_i ; qacc = W (y + s)
    {
      const float fz = rowact ? force : 0.f;
      const int nmax = max(nLo, nHi);
      const int mybase = (h == envLo) ? 0 : split, myn = (h == envLo) ? nLo : nHi;
      const int kc = l < kCsD ? l : 0;
      float sacc = 0.f;
      for (int i = 0; i < nmax; i++) {
        const float fa = rdlane(fz, i), fb = rdlane(fz, (split + i) & 63);
        const float f = (h == envLo) ? fa : fb;
        if (i < myn) sacc += f * s_C[(mybase + i) * kCsD + kc];
      }
      const float* myy = s_C + (mybase + myn) * kCsD;
      if (dofl) s_v2[l] = myy[l] + sacc;  // y + s
    }
    gsync();
    if (dofl) s_v0[l] = dot28(E + o_W + l * kWsD, s_v2);
    gsync();
    // qfrc_smooth + qfrc_constraint = M qacc, for callers that read the joint torques (the env adapter's reward): the one-env kernel's sum, term for term
    if (P.qfrc_out && on && dofl) {
      float acc = 0.f;
#pragma unroll 8
      for (int j = 0; j < 32; j++) acc = __builtin_fmaf(s_qM[M.mdense[j * 32 + l]], j < NV ? s_v0[j] : 0.f, acc);
      P.qfrc_out[(size_t)env * NV + l] = acc;
    }
    // mj_checkAcc (mujoco.h:307): a bad qacc resets the data and runs mj_forward again; the step then integrates that result
    unsigned again = 0u;  // envs of this pass that run a second forward pass
    {
      bool bad = false;
      if (dofl) bad = !(fabsf(s_v0[l]) <= HB_MAXVAL);
      const unsigned long long b64 = __ballot(bad);
      const bool mybad = (unsigned)(b64 >> (32 * h)) != 0u;
      if (on && mybad) {
        status |= (1 << 6);
        if (l < NQ) s_qpos[l] = M.qpos0[l];
        if (l < 28) { s_qvel[l] = 0.f; s_warm[l] = 0.f; s_v0[l] = 0.f; }
        time = 0.f;
        if (!redo) { redo = true; ctrl_zeroed = true; }
        else redo = false;  // (the reset state itself gives a bad qacc: integrate with qacc = 0)
      } else if (on) redo = false;
      const unsigned long long r64 = __ballot(on && mybad && redo);
      again = ((unsigned)r64 ? 1u : 0u) | ((unsigned)(r64 >> 32) ? 2u : 0u);
      gsync();
    }
    const bool fin = on && !((again >> h) & 1u);  // the lane's env completes its step in this pass
    if (fin) ctrl_zeroed = false;
    // (counts[3], what the heavy-first order sorts by: the step's rows x sweeps - in a launch of several steps their average, with what the
    // env brought along weighing as much as kCostHistory steps: a launch of five steps does not forget what six hundred said)
    if (fin) { cost_sum += nefc * (niter + 4); cost_n++; }
    if (fin && l == 0) { int* c = P.counts + kCountStride * (size_t)env; c[0] = ncon; c[1] = nefc; c[2] = niter; c[3] = MULTI ? (cost_prev * kCostHistory + cost_sum) / (kCostHistory + cost_n) : nefc * (niter + 4); c[4] = selfcol; }

    HB_STAMP(14);
    // ---------------------------------------------------------------- mj_Euler: (M + h diag(damping)) qacc' = qfrc_smooth + qfrc_constraint
    if (fin && l < NV) s_warm[l] = s_v0[l];  // qacc_warmstart <- qacc
    const unsigned finm = act & ~again;
    if (eulerdamp) {
      int le;
      asm volatile("v_mov_b32 %0, %1" : "=v"(le) : "v"(lane0));
      const int li = le & 31;
      if (finm == 3u) {
        float* EA = lds; float* EB = lds + kEnvF;
        const float rhsa = li < NV ? M.timestep * M.dof_damping[li] * (EA + o_v0)[li] : 0.f;
        const float rhsb = li < NV ? M.timestep * M.dof_damping[li] * (EB + o_v0)[li] : 0.f;
        float xa, xb;
        f32x16 Xa, Xb;
        load_sym_env2<1>(M, EA + o_qM, EA + o_Hd, EB + o_qM, EB + o_Hd, le, Xa, Xb);
        sym_solve_mfma_x2<14>(Xa, Xb, rhsa, rhsb, xa, xb, le);
        if (le < NV) { (EA + o_v2)[le] = (EA + o_v0)[le] - xa; (EB + o_v2)[le] = (EB + o_v0)[le] - xb; }
      } else if (finm) {
        float* EE = lds + ((finm >> 1) & 1u) * kEnvF;
        const float rhs = li < NV ? M.timestep * M.dof_damping[li] * (EE + o_v0)[li] : 0.f;
        const float x = sym_solve_mfma<14>(load_sym_env<1>(M, EE + o_qM, EE + o_Hd, le), rhs, le);
        if (le < NV) (EE + o_v2)[le] = (EE + o_v0)[le] - x;
      }
      gsync();
    } else {
      if (fin && l < NV) s_v2[l] = s_v0[l];
      gsync();
    }
    // mj_advance
    const float hstep = M.timestep;
    if (fin && l < NV) s_qvel[l] += hstep * s_v2[l];
    gsync();
    if (fin && l < NJ) {
      const int j = l;
      int qa = M.jnt_qposadr[j], da = M.jnt_dofadr[j];
      if (M.jnt_type[j] == 0) {
        for (int i = 0; i < 3; i++) s_qpos[qa + i] += hstep * s_qvel[da + i];
        float n;
        V3 w = normalized(ld3(s_qvel + da + 3), &n);
        Q4 q = qnormalize(ldq(s_qpos + qa + 3));
        stq(s_qpos + qa + 3, qmul(q, axisangle(w, hstep * n)));
      } else s_qpos[qa] += hstep * s_qvel[da];
    }
    if (fin) time += hstep;
    gsync();
#ifdef HB_STAMPS
    HB_STAMP(15);
    if (lane == 0 && P.stamps) for (int i = 0; i < 16; i++) { P.stamps[(size_t)envA * 16 + i] = stamps_[i]; if (envB >= 0) P.stamps[(size_t)envB * 16 + i] = stamps_[i]; }
#endif
    todo &= ~finm;
  }
  }
#ifdef HB_STAMPS
  // diagnostic build (tools/gpu_wave_ends.py): where the wave ran (HW_ID, XCC_ID), which block it was, and when it ended - slots 1..4 of the
  // stamp record of its first env (the stage stamps of its last step stay in the others)
  if (lane == 0 && P.stamps) {
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = P.stamps + (size_t)envA * 16;
    o[1] = hw; o[2] = xcc; o[3] = blockIdx.x; o[4] = __builtin_amdgcn_s_memtime(); o[5] = t_wave0;
  }
#endif
  // ---- state out
  if (env >= 0) {
    float* gs = P.state + (size_t)env * NSTATE;
    if (l0 == 0) gs[0] = time;
    if (l0 < NQ) gs[1 + l0] = s_qpos[l0];
    if (l0 < NV) { gs[1 + NQ + l0] = s_qvel[l0]; gs[1 + NQ + NV + l0] = s_warm[l0]; }
    if (status && l0 == 0) atomicOr(P.status + env, status);
  }
}

__global__ __launch_bounds__(kGroup, 2) void hb_step_duo_kernel(const DevModel* Mp, const BatchPtrs P) { step_duo<0>(Mp, P, 1); }
// the rollouts' kernel: the step loop inside (hb_rollout_dev / hb_rollout_halton without optional outputs)
__global__ __launch_bounds__(kGroup, 2) void hb_step_duo_q_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_duo<1>(Mp, P, nsteps); }

// lean single-step launches of the 27-dof humanoid's PGS kernel, two envs per wave (launch_step_kernel, hb_step.hip)
hipError_t launch_step_duo(const DevModel* M_dev, const BatchPtrs& P, int nsteps, hipStream_t stream) {
  (void)hipGetLastError();
  if (nsteps == 1) hipLaunchKernelGGL(hb_step_duo_kernel, dim3((P.nblk + 1) / 2), dim3(kGroup), (size_t)duo::kLdsF * sizeof(float), stream, M_dev, P);
  else hipLaunchKernelGGL(hb_step_duo_q_kernel, dim3((P.nblk + 1) / 2), dim3(kGroup), (size_t)duo::kLdsF * sizeof(float), stream, M_dev, P, nsteps);
  return hipGetLastError();
}

}  // namespace hb
