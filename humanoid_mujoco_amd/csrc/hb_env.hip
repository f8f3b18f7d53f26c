// hb_env.hip - everything around the step kernels: MJPC task costs and spline tapes, reset, the env adapter (observations, rewards,
// realism layer, domain randomisation), the policy MLP kernels, the heavy-first order, benchmark controls - and their launchers.
#include <hip/hip_runtime.h>
#include "hb_kcommon.hpp"
#include "hb_launch.hpp"

namespace hb {

// ---- MJPC task cost on the recorded read-out rows --------------------------------------------------------------
// mjpc::Norm (mujoco_mpc/mjpc/norm.cc:50-208), value only
__device__ __forceinline__ float mjpc_norm(int type, const float* x, int n, float p, float q) {
  float y = 0.f;
  switch (type) {
    case 0: for (int i = 0; i < n; i++) y += x[i] * x[i]; return 0.5f * y;                                   // kQuadratic
    case 1: { float c = 0.f; for (int i = 0; i < n; i++) c += x[i] * x[i]; return powf(powf(c, 0.5f * q) + powf(p, q), 1.f / q) - p; }  // kL22
    case 2: { float c = 0.f; for (int i = 0; i < n; i++) c += x[i] * x[i]; return sqrtf(c + p * p) - p; }  // kL2
    case 3: for (int i = 0; i < n; i++) y += p * p * (coshf(x[i] / p) - 1.f); return y;                      // kCosh
    case 5: for (int i = 0; i < n; i++) y += powf(fabsf(x[i]), p); return y;                                   // kPowerLoss
    case 6: for (int i = 0; i < n; i++) y += sqrtf(x[i] * x[i] + p * p) - p; return y;                         // kSmoothAbsLoss
    case 7: for (int i = 0; i < n; i++) y += powf(powf(fabsf(x[i]), q) + powf(p, q), 1.f / q) - p; return y;  // kSmoothAbs2Loss
    case 8: for (int i = 0; i < n; i++) y += p > 0.f ? p * logf(1.f + expf(x[i] / p)) : fmaxf(x[i], 0.f); return y;  // kRectifyLoss
    default: return x[0];                                                                                       // kNull
  }
}

// BaseResidualFn::CostTerms + CostValue (mujoco_mpc/mjpc/task.cc:71-110): term k = weight[k] * Norm(norm[k]) of the next dim[k] residual
// entries; the sum goes through the risk transformation (risk-neutral below kRiskNeutralTolerance = 1e-6).  `terms` nullable.
__device__ __forceinline__ float mjpc_risk(float risk, float c) { return fabsf(risk) >= 1e-6f ? (expf(risk * c) - 1.f) / risk : c; }
__device__ __forceinline__ float mjpc_cost_value(int nterm, const int* dim, const int* norm, const float* weight, const float* p, const float* q, float risk,
                                                 const float* res, float* terms) {
  float cost = 0.f;
  int sh = 0;
  for (int k = 0; k < nterm; k++) {
    const float tk = weight[k] * mjpc_norm(norm[k], res + sh, dim[k], p[k], q[k]);
    if (terms) terms[k] = tk;
    cost += tk;
    sh += dim[k];
  }
  return mjpc_risk(risk, cost);
}

// hb_task_cost: the same cost evaluation for n caller-supplied residual vectors (one thread each)
__global__ void hb_cost_terms_kernel(const float* residual, int n, int nres, const CostSpec K, float* terms, float* cost) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  cost[e] = mjpc_cost_value(K.nterm, K.dim, K.norm, K.weight, K.p, K.q, K.risk, residual + (size_t)e * nres, terms ? terms + (size_t)e * K.nterm : nullptr);
}

// One thread per candidate: Stand::ResidualFn::Residual (tasks/humanoid/stand/stand.cc:41-104) on each of the H rows,
// BaseResidualFn::CostValue (task.cc:71-110), Trajectory::UpdateReturn (trajectory.cc:312-326); a candidate that raised
// a bad-state warning returns kMaxReturnValue (trajectory.cc:29,169-173)
__global__ void hb_stand_cost_kernel(const float* rows, int H, int n_env, const StandTask K, const int* status, float* total, float* costs) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_env) return;
  float sum = 0.f;
  for (int t = 0; t < H; t++) {
    const float* r = rows + ((size_t)t * n_env + e) * K.stride;
    float fz = 0.f, fx = 0.f, fy = 0.f;
    for (int k = 0; k < K.n_feet; k++) { fx += r[K.o_feet + 3 * k]; fy += r[K.o_feet + 3 * k + 1]; fz += r[K.o_feet + 3 * k + 2]; }
    const float inv = 1.f / (float)K.n_feet;
    const float height = r[K.o_head + 2] - fz * inv - K.height_goal;
    const float kFallTime = 0.2f;
    const float dx = fx * inv - (r[K.o_com] + kFallTime * r[K.o_vel]), dy = fy * inv - (r[K.o_com + 1] + kFallTime * r[K.o_vel + 1]);
    const float balance = sqrtf(dx * dx + dy * dy);
    float c = K.weight[0] * mjpc_norm(K.norm[0], &height, 1, K.p[0], K.q[0]);
    c += K.weight[1] * mjpc_norm(K.norm[1], &balance, 1, K.p[1], K.q[1]);
    c += K.weight[2] * mjpc_norm(K.norm[2], r + K.o_vel, 2, K.p[2], K.q[2]);
    c += K.weight[3] * mjpc_norm(K.norm[3], r + K.o_qvel + 6, K.nv - 6, K.p[3], K.q[3]);
    c += K.weight[4] * mjpc_norm(K.norm[4], r + K.o_ctrl, K.nu, K.p[4], K.q[4]);
    c = mjpc_risk(K.risk, c);
    if (costs) costs[(size_t)t * n_env + e] = c;
    sum += c;
  }
  const bool failed = status[e] & ((1 << 4) | (1 << 5) | (1 << 6));
  total[e] = failed ? 1.0e6f : sum / (float)max(H, 1);
}

// Walk::ResidualFn::Residual (tasks/humanoid/walk/walk.cc:44-163) on each of the H rows, then the cost terms in the
// order and with the dimensions the task's user sensors declare (task.cc:71-89), return as in hb_stand_cost_kernel
__global__ void hb_walk_cost_kernel(const float* rows, int H, int n_env, const WalkTask K, const int* status, float* total, float* costs) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_env) return;
  float sum = 0.f;
  for (int t = 0; t < H; t++) {
    const float* r = rows + ((size_t)t * n_env + e) * K.stride;
    float res[96];
    int c = 0;
    const float torso_height = r[K.o_torso + 2];
    res[c++] = torso_height - K.height_goal;
    const float* fr = r + K.o_foot_r;
    const float* fl = r + K.o_foot_l;
    res[c++] = 0.5f * (fl[2] + fr[2]) - r[K.o_pelvis + 2] - 0.2f;
    // balance: capture point against its projection onto the segment between the feet
    float cp[3] = {r[K.o_com] + 0.3f * r[K.o_vel], r[K.o_com + 1] + 0.3f * r[K.o_vel + 1], 1.0e-3f};
    float axis[3] = {fr[0] - fl[0], fr[1] - fl[1], 1.0e-3f};
    float an = sqrtf(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    if (an < 1e-15f) { axis[0] = 1.f; axis[1] = 0.f; axis[2] = 0.f; } else { axis[0] /= an; axis[1] /= an; axis[2] /= an; }  // mju_normalize3
    const float length = 0.5f * an - 0.05f;
    const float center[3] = {0.5f * (fr[0] + fl[0]), 0.5f * (fr[1] + fl[1]), 0.5f * (fr[2] + fl[2])};
    const float vec[3] = {cp[0] - center[0], cp[1] - center[1], cp[2] - center[2]};
    float tt = vec[0] * axis[0] + vec[1] * axis[1] + vec[2] * axis[2];
    tt = fmaxf(-length, fminf(length, tt));
    const float pcp[2] = {axis[0] * tt + center[0], axis[1] * tt + center[1]};
    const float standing = torso_height / sqrtf(torso_height * torso_height + 0.45f * 0.45f) - 0.4f;
    res[c++] = standing * (cp[0] - pcp[0]);
    res[c++] = standing * (cp[1] - pcp[1]);
    // upright: axes are [torso_up, pelvis_up, foot_right_up, foot_left_up, torso_forward, pelvis_forward, foot_right_forward, foot_left_forward]
    const float* ax = r + K.o_axes;
    res[c++] = ax[2] - 1.f;
    res[c++] = 0.3f * (ax[3 + 2] - 1.f);
    for (int f = 0; f < 2; f++) {
      const float* up = ax + 3 * (2 + f);
      res[c++] = 0.1f * standing * up[0]; res[c++] = 0.1f * standing * up[1]; res[c++] = 0.1f * standing * (up[2] - 1.f);
    }
    // posture
    for (int i = 7; i < K.nq; i++) res[c++] = r[K.o_qpos + i];
    // walk
    float fw[2] = {0.f, 0.f};
    for (int k = 4; k < 8; k++) { fw[0] += ax[3 * k]; fw[1] += ax[3 * k + 1]; }
    const float fn = sqrtf(fw[0] * fw[0] + fw[1] * fw[1]);
    if (fn < 1e-15f) { fw[0] = 1.f; fw[1] = 0.f; } else { fw[0] /= fn; fw[1] /= fn; }  // mju_normalize
    const float* tv = r + K.o_linvel;  // torso, foot_right, foot_left
    const float cv[2] = {0.5f * (r[K.o_sub] + tv[0]), 0.5f * (r[K.o_sub + 1] + tv[1])};
    res[c++] = standing * (cv[0] * fw[0] + cv[1] * fw[1] - K.speed_goal);
    // move feet
    res[c++] = standing * (cv[0] - 0.5f * tv[3] - 0.5f * tv[6]);
    res[c++] = standing * (cv[1] - 0.5f * tv[4] - 0.5f * tv[7]);
    // control
    for (int i = 0; i < K.nu; i++) res[c++] = r[K.o_ctrl + i];
    const float cost = mjpc_cost_value(K.nterm, K.dim, K.norm, K.weight, K.p, K.q, K.risk, res, nullptr);
    if (costs) costs[(size_t)t * n_env + e] = cost;
    sum += cost;
  }
  const bool failed = status[e] & ((1 << 4) | (1 << 5) | (1 << 6));
  total[e] = failed ? 1.0e6f : sum / (float)max(H, 1);
}

// ---- SamplingPolicy::Action on the device (mujoco_mpc/mjpc/planners/sampling/policy.cc:50-58): every candidate's
// time spline (mjpc/spline/spline.cc:103-156,240-277: zero-order / linear / cubic Hermite with finite-difference slopes)
// sampled at time0 + t * dt and clamped to ctrlrange, written as the action tape [T][n_env][nu] the rollouts read.
// knots: [n_env][P][nu]; times: [P], increasing, shared by the candidates.
__global__ void hb_spline_tape_kernel(const DevModel M, const float* knots, const float* times, int P, int interp, float time0, float dt, int T, int n_env, float* tape) {
  const int nu = M.nu;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)T * n_env * nu;
  if (idx >= total) return;
  const int i = (int)(idx % nu), e = (int)((idx / nu) % n_env), t = (int)(idx / ((size_t)nu * n_env));
  const float time = time0 + (float)t * dt;
  const float* y = knots + (size_t)e * P * nu + i;  // y[k * nu]: node k
  float v;
  int up = 0;
  while (up < P && times[up] <= time) up++;  // std::upper_bound
  if (P == 0) v = 0.f;
  else if (up == P) v = y[(size_t)(P - 1) * nu];
  else if (up == 0) v = y[0];
  else {
    const int lo = up - 1;
    const float t0 = times[lo], t1 = times[up], x = (time - t0) / (t1 - t0);
    const float p0 = y[(size_t)lo * nu], p1 = y[(size_t)up * nu];
    if (interp == 0) v = p0;
    else if (interp == 1) v = p0 * (1.f - x) + p1 * x;
    else {
      // TimeSpline::Slope: one-sided at the ends, mean of the two one-sided differences inside
      auto slope = [&](int k) {
        if (k == 0) return (y[(size_t)1 * nu] - y[0]) / (times[1] - times[0]);
        const float back = (y[(size_t)k * nu] - y[(size_t)(k - 1) * nu]) / (times[k] - times[k - 1]);
        if (k == P - 1) return back;
        return 0.5f * (y[(size_t)(k + 1) * nu] - y[(size_t)k * nu]) / (times[k + 1] - times[k]) + 0.5f * back;
      };
      const float h = t1 - t0, x2 = x * x, x3 = x2 * x;
      v = (2.f * x3 - 3.f * x2 + 1.f) * p0 + (x3 - 2.f * x2 + x) * h * slope(lo) + (-2.f * x3 + 3.f * x2) * p1 + (x3 - x2) * h * slope(up);
    }
  }
  // Clamp(action, actuator_ctrlrange, nu) (utilities.cc:94-98); an actuator without a control range is left alone
  const float lo_r = M.act_ctrlrange[2 * i], hi_r = M.act_ctrlrange[2 * i + 1];
  if (M.act_ctrllimited[i] || lo_r < hi_r) v = fminf(fmaxf(v, lo_r), hi_r);
  tape[idx] = v;
}

// ------------------------------------------------------------------------------------------
// reset: qpos0/keyframe (+ Halton perturbation), zero velocity/warmstart/time/status
// qpos <- reset pose (+ the Halton perturbation indexed by global env and, for the env adapter, episode), rest zero
// (lane l of nl cooperating lanes writes the entries it owns: the qpos entries of joints l, l + nl, ... - every qpos entry belongs to one
// joint - and a strided share of the velocity and warm-start entries; l = 0, nl = 1: one thread does it all)
__device__ __forceinline__ void reset_state(const DevModel& M, float* s, const float* qpos_src, float perturb, int env_global, int ep, float quat_perturb = 0.f, int l = 0,
                                            int nl = 1) {
  if (l == 0) s[0] = 0.f;
  for (int i = l; i < 2 * M.nv; i += nl) s[1 + M.nq + i] = 0.f;
  const int idx = env_global + 1 + ep * 7919;
  for (int j = l; j < M.njnt; j += nl) {
    const int qa = M.jnt_qposadr[j];
    if (M.jnt_type[j] == 0) {
      for (int i = 0; i < 7; i++) s[1 + qa + i] = qpos_src[qa + i];
      if (perturb > 0.f) {
        s[1 + qa + 2] += perturb * 0.1f * halton(idx, 3);
        // root orientation: every quaternion component +- quat_perturb (cpu_env.py:316-328), left unnormalised as in the reference
        for (int i = 0; i < 4; i++) s[1 + qa + 3 + i] += perturb * quat_perturb * (2.f * halton(idx, 2 + M.njnt + i) - 1.f);
      }
    } else {
      s[1 + qa] = qpos_src[qa];
      if (perturb > 0.f) s[1 + qa] += perturb * 0.2f * (2.f * halton(idx, 2 + j) - 1.f);
    }
  }
}
__global__ void hb_reset_kernel(const DevModel M, float* state, int* status, const uint8_t* mask, const float* qpos_src, const int* episode, int n_env, float perturb,
                                int env_offset, float quat_perturb) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_env) return;
  if (mask && !mask[e]) return;
  reset_state(M, state + (size_t)e * M.nstate, qpos_src, perturb, env_offset + e, episode ? episode[e] : 0, quat_perturb);
  status[e] = 0;
}

// ---- env realism (hb_env_randomization): counter-based random numbers, delay rings, pushes -------------------
// One 32-bit word per (seed, global env, episode, step, stream, element): reproducible, order-free, and the same
// on any split of the batch.  tests/env_ref.py restates these functions in numpy.


// start of an episode: delays drawn (cpu_env.py:135-168), rings logically empty, push schedule cleared
__device__ __forceinline__ void envrand_begin_episode(const DevModel& M, const EnvRand& R, const EnvRandState& S, int e, int env_global, int ep) {
  const float dt = R.control_timestep > 0.f ? R.control_timestep : M.timestep;
  for (int c = 0; c < 4; c++) {
    const float u = rng_uniform(R.seed, env_global, ep, 0, RS_DELAY, c);
    const float d = (R.min_delay + u * (R.max_delay - R.min_delay)) * R.factor;
    S.delay[4 * e + c] = min(kDelaySlots - 1, max(0, (int)rintf(d / dt)));
  }
  S.k_act[e] = 0;
  S.k_obs[e] = 0;
  float* p = S.push + 8 * (size_t)e;
  if (S.xfrc) {
    const int body = (int)p[5];
    if (body > 0 && body < M.nbody) { S.xfrc[((size_t)e * M.nbody + body) * 6] = 0.f; S.xfrc[((size_t)e * M.nbody + body) * 6 + 1] = 0.f; }
  }
  for (int i = 0; i < 8; i++) p[i] = 0.f;
}
__global__ void hb_envrand_reset_kernel(const DevModel M, const EnvRand R, const EnvRandState S, const int* episode, const uint8_t* mask, int n_env, int env_offset) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_env || (mask && !mask[e])) return;
  envrand_begin_episode(M, R, S, e, env_offset + e, episode[e]);
}

// Per-env model parameters of one episode (cpu_env.py:188-264): see hb_domain_randomization in include/hb.h.
enum { RS_DR_MASS = 16, RS_DR_EXTRA, RS_DR_FRIC, RS_DR_ARM, RS_DR_STIFF, RS_DR_MARGIN, RS_DR_RANGE, RS_DR_KP, RS_DR_FRC, RS_DR_FLOOR };
// (NL cooperating lanes, l = this lane's number among them: every table is filled lane-strided; the height map's range is reduced
// over the lanes with shuffles, so NL is 1 or the env kernels' kDrawLanes = 16 consecutive lanes of a wave)
constexpr int kDrawLanes = 16;
template <int NL>
__device__ __forceinline__ void domain_draw(const DevModel& M, const DomainRand& D, float* d, int env_global, int ep, int l = 0) {
  static_assert(NL == 1 || NL == kDrawLanes, "domain_draw: one lane or kDrawLanes");
  const DomainLayout L = domain_layout(M.nbody, M.nv, M.nlimcand, M.nu, M.nhfielddata);
  const float rf = D.factor;
  auto U = [&](int stream, int idx) { return rng_uniform(D.seed, env_global, ep, 0, stream, idx); };
  if (l == 0) d[L.o_mass] = 0.f;
  const int bx = M.nbody > 1 ? 1 + min(M.nbody - 2, (int)(U(RS_DR_EXTRA, 0) * (float)(M.nbody - 1))) : -1;  // the body that carries the extra mass
  for (int sl = 1 + l; sl < M.nbody; sl += NL) {  // brec is level-ordered: slot -> body id, mass
    const float4 q0 = M.brec[(size_t)sl * kBrecQuads], q1 = M.brec[(size_t)sl * kBrecQuads + 1];
    const int b = __float_as_int(q0.x);
    float mass = fmaxf(1e-5f, q1.z + (2.f * U(RS_DR_MASS, b) - 1.f) * D.max_mass_change * rf);
    if (b == bx) mass += U(RS_DR_EXTRA, 1) * D.max_external_mass * rf;
    d[L.o_mass + b] = mass;
  }
  for (int i = l; i < M.nv; i += NL) {
    const float4 dA = M.drec[3 * i], dB = M.drec[3 * i + 1];
    const bool scalar = __float_as_int(dA.z) >= 2;  // hinge / slide
    d[L.o_arm + i] = dB.y + (scalar ? U(RS_DR_ARM, i) * D.armature_max_change * rf : 0.f);
    d[L.o_stiff + i] = dB.w + (scalar ? U(RS_DR_STIFF, i) * D.stiffness_max_change * rf : 0.f);
  }
  for (int c = l; c < M.nlimcand; c += NL) {
    const bool joint = M.lim_kind[c] == 0;
    const int id = M.lim_id[c];
    d[L.o_lmargin + c] = M.lim_margin[c] + (joint ? U(RS_DR_MARGIN, id) * D.margin_max_change * rf : 0.f);  // one margin per joint
    d[L.o_lrange + c] = M.lim_range[c] + (joint ? (2.f * U(RS_DR_RANGE, c) - 1.f) * D.range_max_change * rf : 0.f);
  }
  for (int a = l; a < M.nu; a += NL) {
    float gain = M.act_gain[a], bias1 = M.act_bias[3 * a + 1];
    if (D.kp_nominal > 0.f) {
      gain = D.kp_nominal + (2.f * U(RS_DR_KP, a) - 1.f) * D.kp_max_change * rf;
      if (bias1 != 0.f) bias1 = -gain;
    }
    d[L.o_gain + a] = gain;
    d[L.o_bias1 + a] = bias1;
    d[L.o_frc + 2 * a] = M.act_forcerange[2 * a] + (2.f * U(RS_DR_FRC, 2 * a) - 1.f) * D.force_limit_max_change * rf;
    d[L.o_frc + 2 * a + 1] = M.act_forcerange[2 * a + 1] + (2.f * U(RS_DR_FRC, 2 * a + 1) - 1.f) * D.force_limit_max_change * rf;
  }
  if (l == 0) d[L.o_fric] = (1.f - rf) + (D.friction_min_mult + U(RS_DR_FRIC, 0) * (D.friction_max_mult - D.friction_min_mult)) * rf;
  // floor height maps (CPUEnv._randomize_floor_heightmap, cpu_env.py:267-280: Perlin noise on the grid, shifted and scaled to
  // [0, 1], times MIN + factor (MAX - MIN)).  The reference's noise comes from the third-party perlin_noise package; here:
  // three octaves of smooth value noise from the counter-based generator, normalised the same way.
  const float bump = D.floor_bump_min + rf * (D.floor_bump_max - D.floor_bump_min);
  for (int hf = 0, adr = 0; adr < M.nhfielddata; hf++) {
    const int nr = M.hfield_nrow[hf], nc = M.hfield_ncol[hf], n = nr * nc;
    float* h = d + L.o_hfield + adr;
    if (!(D.floor_bump_max > 0.f)) { for (int i = l; i < n; i += NL) h[i] = M.hfield_data[adr + i]; adr += n; continue; }
    float lo = 3.0e38f, hi = -3.0e38f;
    for (int i = l; i < n; i += NL) {
      const int r = i / nc, c = i - r * nc;
      float v = 0.f, amp = 1.f;
      for (int oct = 0, cells = 2; oct < 3; oct++, cells *= 2, amp *= 0.5f) {  // lattices of 3x3, 5x5, 9x9 nodes over the field
        const float x = (float)c / (float)max(1, nc - 1) * (float)cells, y = (float)r / (float)max(1, nr - 1) * (float)cells;
        const int x0 = min((int)x, cells - 1), y0 = min((int)y, cells - 1);
        float fx = x - (float)x0, fy = y - (float)y0;
        fx = fx * fx * (3.f - 2.f * fx); fy = fy * fy * (3.f - 2.f * fy);  // smoothstep
        auto node = [&](int ix, int iy) { return rng_uniform(D.seed, env_global, ep, hf, RS_DR_FLOOR, (oct * 16 + iy) * 16 + ix); };
        const float a = node(x0, y0), b = node(x0 + 1, y0), cc = node(x0, y0 + 1), dd = node(x0 + 1, y0 + 1);
        v += amp * ((a * (1.f - fx) + b * fx) * (1.f - fy) + (cc * (1.f - fx) + dd * fx) * fy);
      }
      h[i] = v;  // (re-read below by the lane that wrote it)
      lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    if (NL > 1) {
#pragma unroll
      for (int m = NL / 2; m >= 1; m >>= 1) { lo = fminf(lo, __shfl_xor(lo, m, NL)); hi = fmaxf(hi, __shfl_xor(hi, m, NL)); }
    }
    const float sc = hi > lo ? bump / (hi - lo) : 0.f;
    for (int i = l; i < n; i += NL) h[i] = (h[i] - lo) * sc;
    adr += n;
  }
}
__global__ void hb_domain_rand_kernel(const DevModel M, const DomainRand D, float* dr, int stride, const int* episode, const uint8_t* mask, int n_env, int env_offset) {
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) / kDrawLanes, l = threadIdx.x % kDrawLanes;
  if (e >= n_env || (mask && !mask[e])) return;
  domain_draw<kDrawLanes>(M, D, dr + (size_t)e * stride, env_offset + e, episode[e], l);
}

// value through a delay ring: push x as item k, return item k - d (filler before the ring has d items)
__device__ __forceinline__ float ring_delay(float* ring, int stride, int k, int d, float x, float filler) {
  ring[(size_t)(k % kDelaySlots) * stride] = x;
  if (d == 0) return x;
  return k >= d ? ring[(size_t)((k - d) % kDelaySlots) * stride] : filler;
}

// CPUEnv._apply_action + _apply_external_forces (cpu_env.py:612-674) for one env per thread.
// action == nullptr: the reference's step(None), which re-applies the current controls without noise.
__global__ void hb_action_env_kernel(const DevModel M, const EnvRand R, const EnvRandState S, const float* action, float* prev, float* latest, float* ctrl,
                                     const int* episode, const float* state, const uint8_t* mask, int n_env, int env_offset) {
  // (sixteen lanes per env: the actuators lane-strided, the push schedule on the env's first lane)
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) / 16, l = threadIdx.x % 16;
  if (e >= n_env || (mask && !mask[e])) return;
  const int nu = M.nu, ge = env_offset + e, ep = episode[e];
  const int k = S.k_act[e], d = S.delay[4 * e];
  const unsigned kk = R.frozen_noise ? 0u : (unsigned)k;
  for (int i = l; i < nu; i += 16) {
    const size_t ai = (size_t)e * nu + i;
    float a = action ? action[ai] : ctrl[ai];
    if (action && R.action_noise > 0.f) a += R.factor * R.action_noise * rng_normal(R.seed, ge, ep, kk, RS_ACTION, i);
    const float out = ring_delay(S.fifo_act + ((size_t)e * kDelaySlots) * nu + i, nu, k, d, a, 0.f);
    prev[ai] = latest[ai];
    latest[ai] = out;
    ctrl[ai] = out;
  }
  if (l != 0) return;
  S.k_act[e] = k + 1;
  if (R.push_enabled && S.xfrc) {
    float* p = S.push + 8 * (size_t)e;
    float* xf = S.xfrc + (size_t)e * M.nbody * 6;
    const float time = state[(size_t)e * M.nstate];
    if (time >= p[0] + p[1]) {  // window over (or first step): clear the old force, schedule the next push
      const unsigned ev = (unsigned)p[6];
      int body = (int)p[5];
      if (body > 0 && body < M.nbody) { xf[6 * body] = 0.f; xf[6 * body + 1] = 0.f; }
      p[0] = time + R.push_min_interval + rng_uniform(R.seed, ge, ep, ev, RS_PUSH, 0) * (R.push_max_interval - R.push_min_interval);
      p[1] = R.push_min_duration + rng_uniform(R.seed, ge, ep, ev, RS_PUSH, 1) * (R.push_max_duration - R.push_min_duration);
      p[2] = R.factor * (R.push_min_force + rng_uniform(R.seed, ge, ep, ev, RS_PUSH, 2) * (R.push_max_force - R.push_min_force));
      float dx = 2.f * rng_uniform(R.seed, ge, ep, ev, RS_PUSH, 3) - 1.f, dy = 2.f * rng_uniform(R.seed, ge, ep, ev, RS_PUSH, 4) - 1.f;
      const float n = sqrtf(dx * dx + dy * dy);  // never 0: the uniforms are odd multiples of 2^-24
      p[3] = dx / n; p[4] = dy / n;
      body = 1 + min(M.nbody - 2, (int)(rng_uniform(R.seed, ge, ep, ev, RS_PUSH, 5) * (float)(M.nbody - 1)));
      p[5] = (float)body;
      p[6] = (float)(ev + 1);
    }
    if (time > p[0] && time < p[0] + p[1]) {
      const int body = (int)p[5];
      xf[6 * body] = p[3] * p[2];
      xf[6 * body + 1] = p[4] * p[2];
    }
  }
}

// CPUEnv._get_obs's noise and delay lines (cpu_env.py:465-545) applied in place to the true observation o
// env adapter: observation, the 27-DoF analogue of CPUEnv._get_obs (cpu_env.py:465-571):
// [hinge/slide qpos, hinge/slide qvel, root angular velocity, gravity direction in the root body frame]
// (lane l of nl cooperating lanes writes entries l, l + nl, ... of each part)
__device__ __forceinline__ Q4 obs_root_quat(const DevModel& M, const float* s) {
  const int da = M.obs_root_dofadr;
  return da >= 0 ? ldq(s + 1 + M.jnt_qposadr[M.dof_jntid[da]] + 3) : Q4{1.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void compute_obs(const DevModel& M, const float* s, float* o, int l = 0, int nl = 1) {
  const float* qpos = s + 1;
  const float* qvel = s + 1 + M.nq;
  const int nj = (M.nobs - 6) / 2;
  for (int i = l; i < nj; i += nl) { o[i] = qpos[M.jnt_qposadr[M.obs_jnt[i]]]; o[nj + i] = qvel[M.jnt_dofadr[M.obs_jnt[i]]]; }
  const int da = M.obs_root_dofadr;
  // gravity direction in the torso frame: R(q)^T (0,0,-1)  (cpu_env.py:510-519)
  float m[9];
  q2mat(m, qnormalize(obs_root_quat(M, s)));
  for (int c = l; c < 3; c += nl) { o[2 * nj + c] = da >= 0 ? qvel[da + 3 + c] : 0.f; o[2 * nj + 3 + c] = -m[6 + c]; }
}

// the same observation through CPUEnv's sensor model (cpu_env.py:465-571): noise on every reading, each group of readings delayed by
// its own number of control steps (rings of kDelaySlots past readings)
__device__ __forceinline__ void envrand_observe(const DevModel& M, const EnvRand& R, const EnvRandState& S, int e, int env_global, int ep, const float* s, float* o, int l = 0,
                                                int nl = 1) {
  const float* qpos = s + 1;
  const float* qvel = s + 1 + M.nq;
  const int k = S.k_obs[e];
  const unsigned kk = R.frozen_noise ? 0u : (unsigned)k;
  const int nj = (M.nobs - 6) / 2;
  const int dj = S.delay[4 * e + 1], dg = S.delay[4 * e + 2], dv = S.delay[4 * e + 3];
  float* rj = S.fifo_joint + ((size_t)e * kDelaySlots) * 2 * nj;
  for (int i = l; i < nj; i += nl) {
    const float a = qpos[M.jnt_qposadr[M.obs_jnt[i]]] + R.factor * R.joint_angle_noise * rng_normal(R.seed, env_global, ep, kk, RS_JOINT_POS, i);
    const float v = qvel[M.jnt_dofadr[M.obs_jnt[i]]] + R.factor * R.joint_velocity_noise * rng_normal(R.seed, env_global, ep, kk, RS_JOINT_VEL, i);
    o[i] = ring_delay(rj + i, 2 * nj, k, dj, a, 0.f);
    o[nj + i] = ring_delay(rj + nj + i, 2 * nj, k, dj, v, 0.f);
  }
  if (l < 3) {  // (the three components of the gyro and of the gravity direction: lanes 0..2, or one lane all three)
    const int da = M.obs_root_dofadr;
    // gravity direction from the noisy, re-normalised torso quaternion (Rotation.from_quat normalises)
    Q4 q = obs_root_quat(M, s);
    q.w += R.factor * R.imu_noise * rng_normal(R.seed, env_global, ep, kk, RS_IMU, 0);
    q.x += R.factor * R.imu_noise * rng_normal(R.seed, env_global, ep, kk, RS_IMU, 1);
    q.y += R.factor * R.imu_noise * rng_normal(R.seed, env_global, ep, kk, RS_IMU, 2);
    q.z += R.factor * R.imu_noise * rng_normal(R.seed, env_global, ep, kk, RS_IMU, 3);
    float m[9];
    q2mat(m, qnormalize(q));
    float* rg = S.fifo_gyro + ((size_t)e * kDelaySlots) * 3;
    float* rv = S.fifo_grav + ((size_t)e * kDelaySlots) * 3;
    for (int c = l; c < 3; c += nl) {
      const float w = (da >= 0 ? qvel[da + 3 + c] : 0.f) + R.factor * R.gyro_noise * rng_normal(R.seed, env_global, ep, kk, RS_GYRO, c);
      o[2 * nj + c] = ring_delay(rg + c, 3, k, dg, w, 0.f);
      o[2 * nj + 3 + c] = ring_delay(rv + c, 3, k, dv, -m[6 + c], c == 2 ? -1.f : 0.f);
    }
  }
  if (l == 0) S.k_obs[e] = k + 1;
}

__global__ void hb_obs_kernel(const DevModel M, const float* state, float* obs, int n_env) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_env) return;
  compute_obs(M, state + (size_t)e * M.nstate, obs + (size_t)e * M.nobs);
}

// CPUEnv._apply_action bookkeeping (cpu_env.py:656-674): previous <- latest, latest <- action, ctrl <- action
__global__ void hb_action_kernel(const float* action, float* prev, float* latest, float* ctrl, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  prev[i] = latest[i];
  float a = action[i];
  latest[i] = a;
  ctrl[i] = a;
}

__device__ __forceinline__ float scaled_exp(float x) { return expf(-x / 0.5f); }  // reward_functions.py:17-19

// standupReward (reward_functions.py:247-374) + observation + termination + auto-reset.  kEnvLanes lanes per env, 256 / kEnvLanes
// envs per block (it was one thread per env: 37 us of serial work on 32 CUs for 4096 envs, a fifth of VecEnv.step_torch's GPU time).
// The env's state record is staged in LDS (one coalesced pass instead of a strided read per thread); sums over joints, actuators
// and symmetry pairs are lane-strided partial sums reduced over the env's lanes; an auto-reset writes the new state into the same
// LDS copy (every lane the joints it owns), so that the observation of the new episode is read from it after the block barrier.
constexpr int kEnvLanes = 16;
__device__ __forceinline__ float env_lane_sum(float x) {
#pragma unroll
  for (int m = kEnvLanes / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, kEnvLanes);
  return x;
}
__global__ __launch_bounds__(256) void hb_env_kernel(const DevModel M, const EnvConfig cfg, const EnvRand R, const EnvRandState S, float* state, const float* qfrc, const int* counts,
                              float* prev, float* latest, const float* qpos_src, int* episode, int* status, float* obs, float* reward, uint8_t* terminated,
                              uint8_t* truncated, const uint8_t* mask, int observe, const DomainRand D, float* dr, int dr_stride, int n_env, int env_offset, float* term_obs, int* seen) {
  extern __shared__ float sh_state[];
  const int grp = threadIdx.x / kEnvLanes, l = threadIdx.x % kEnvLanes;
  const int e = blockIdx.x * (256 / kEnvLanes) + grp;
  const bool active = e < n_env && (!mask || mask[e]);
  const int nsp = (M.nstate + 3) & ~3;
  float* ls = sh_state + grp * nsp;
  float* s = state + (size_t)(active ? e : 0) * M.nstate;
  if (active) for (int i = l; i < M.nstate; i += kEnvLanes) ls[i] = s[i];
  __syncthreads();
  bool reset = false;
  if (active) {
    const float* qpos = ls + 1;
    const float* qvel = ls + 1 + M.nq;
    const int da = M.obs_root_dofadr;
    // root height and the gravity direction in the torso frame (every lane)
    Q4 q = {1.f, 0.f, 0.f, 0.f};
    float z = 0.f;
    if (da >= 0) { const int qa = M.jnt_qposadr[M.dof_jntid[da]]; q = qnormalize(ldq(qpos + qa + 3)); z = qpos[qa + 2]; }
    float m[9];
    q2mat(m, q);
    const float g[3] = {-m[6], -m[7], -m[8]};
    float r = 0.f;
    // horizontal velocity
    float vx = da >= 0 ? qvel[da] : 0.f, vy = da >= 0 ? qvel[da + 1] : 0.f;
    float dvx = vx - cfg.target_velocity[0], dvy = vy - cfg.target_velocity[1];
    r += cfg.w_hvel * scaled_exp(dvx * dvx + dvy * dvy);
    // upright: |g_local - (0,0,-1)|^2
    r += cfg.w_upright * scaled_exp(g[0] * g[0] + g[1] * g[1] + (g[2] + 1.f) * (g[2] + 1.f));
    // torso height: linear ramp min_z -> target_z, clamped (numpy.interp)
    float t = (z - cfg.min_z) / fmaxf(cfg.target_z - cfg.min_z, 1e-9f);
    r += cfg.w_height * fminf(fmaxf(t, 0.f), 1.f);
    // joint torques on the scalar joints' dofs
    {
      float acc = 0.f, n = 0.f;
      for (int j = l; j < M.njnt; j += kEnvLanes)
        if (M.jnt_type[j] >= 2) {
          float x = fmaxf(fabsf(qfrc[(size_t)e * M.nv + M.jnt_dofadr[j]]) - cfg.safe_torque, 0.f);
          acc += scaled_exp(x * x);
          n += 1.f;
        }
      acc = env_lane_sum(acc); n = env_lane_sum(n);
      if (n > 0.f) r += cfg.w_torque * acc / n;
    }
    // control change / regularisation / symmetry on the (scaled) actions
    const float* pa = prev + (size_t)e * M.nu;
    const float* la = latest + (size_t)e * M.nu;
    const float inv = 1.f / cfg.action_scale;
    if (M.nu > 0) {
      float chg = 0.f, reg = 0.f;
      for (int i = l; i < M.nu; i += kEnvLanes) {
        float d = (la[i] - pa[i]) * inv * cfg.control_frequency;
        chg += scaled_exp(d * d);
        float a = la[i] * inv;
        reg += scaled_exp(a * a);
      }
      chg = env_lane_sum(chg); reg = env_lane_sum(reg);
      r += cfg.w_ctrl_change * chg / (float)M.nu + cfg.w_ctrl_reg * reg / (float)M.nu;
    }
    if (cfg.n_equal + cfg.n_opposite > 0) {
      float sym = 0.f;
      for (int k = l; k < cfg.n_equal + cfg.n_opposite; k += kEnvLanes) {
        const bool eq = k < cfg.n_equal;
        const int a0 = eq ? cfg.equal_pairs[k][0] : cfg.opposite_pairs[k - cfg.n_equal][0], a1 = eq ? cfg.equal_pairs[k][1] : cfg.opposite_pairs[k - cfg.n_equal][1];
        const float d = (eq ? la[a0] - la[a1] : la[a0] + la[a1]) * inv;
        sym += scaled_exp(d * d);
      }
      sym = env_lane_sum(sym);
      r += cfg.w_symmetry * sym / (float)(cfg.n_equal + cfg.n_opposite);
    }
    if (cfg.w_vvel != 0.f) { const float vz = da >= 0 ? qvel[da + 2] : 0.f; r += cfg.w_vvel * scaled_exp(vz * vz); }  // vertical_velocity_penalty
    if (counts[kCountStride * e + 4]) r += cfg.self_collision_penalty;
    const bool upright = fmaxf(fabsf(g[0]), fabsf(g[1])) < cfg.upright_tol;
    const bool timeup = cfg.max_time > 0.f && ls[0] >= cfg.max_time;
    bool term, trunc;
    if (cfg.reward_kind == 1) {  // controlInputReward: fall = terminal (with the terminal reward), time limit = truncation
      term = !upright || z < cfg.min_z_grounded;
      trunc = timeup;
    } else {                     // standupReward: time limit = terminal, standing up = truncation ("is_success")
      term = timeup;
      trunc = z >= cfg.target_z && upright;
    }
    if (term) r = cfg.terminal_reward;
    if (l == 0) { reward[e] = r; terminated[e] = term ? 1 : 0; truncated[e] = trunc ? 1 : 0; }
    reset = (term || trunc) && cfg.auto_reset;
  }
  const bool rand_on = S.k_obs != nullptr;
  int ep = active ? episode[e] : 0;
  // the observation of the state an episode ENDS in, before the env is reset in place: what stable-baselines3's VecEnv hands the learner as
  // infos[i]["terminal_observation"] (DummyVecEnv.step_wait; SAC / PPO bootstrap from it when the episode was truncated) - the same sensor
  // model, noise and delays as any other observation of that episode (CPUEnv.step returns self._get_obs() before anything resets)
  if (reset && term_obs) {
    float* to = term_obs + (size_t)e * M.nobs;
    if (rand_on && observe) envrand_observe(M, R, S, e, env_offset + e, ep, ls, to, l, kEnvLanes);
    else compute_obs(M, ls, to, l, kEnvLanes);
  }
  __syncthreads();  // (every lane has read the old state and the old episode number)
  if (reset) {
    // CPUEnv.reset for this env; the perturbation index advances with the episode count
    ep += 1;
    reset_state(M, ls, qpos_src, cfg.reset_perturb, env_offset + e, ep, cfg.reset_quat_perturb, l, kEnvLanes);
    for (int i = l; i < M.nu; i += kEnvLanes) { prev[(size_t)e * M.nu + i] = 0.f; latest[(size_t)e * M.nu + i] = 0.f; }
    if (l == 0) {
      episode[e] = ep;
      if (seen) seen[e] |= status[e];  // (the warning bits of the episode that ends here stay readable: hb_env_warnings)
      status[e] = 0;
      if (rand_on) envrand_begin_episode(M, R, S, e, env_offset + e, ep);
    }
    static_assert(kEnvLanes == kDrawLanes, "the env kernel draws an episode's model parameters with all lanes of the env");
    if (dr) domain_draw<kDrawLanes>(M, D, dr + (size_t)e * dr_stride, env_offset + e, ep, l);
  }
  __threadfence_block();
  __syncthreads();  // the new state (LDS) and the new episode's delays (global, written by lane 0) are visible to the env's lanes
  if (active) {
    if (reset) for (int i = l; i < M.nstate; i += kEnvLanes) s[i] = ls[i];
    float* o = obs + (size_t)e * M.nobs;
    if (rand_on && observe) envrand_observe(M, R, S, e, env_offset + e, ep, ls, o, l, kEnvLanes);
    else compute_obs(M, ls, o, l, kEnvLanes);
  }
}

// hb_env_reset's collision test (cpu_env.py:411-414): envs of the mask that collide (mode 1: any contact, mode 2:
// self-contact) or ended in their settle step stay in the mask, get a new episode number and are counted
__global__ void hb_reset_check_kernel(const int* counts, const uint8_t* terminated, const uint8_t* truncated, uint8_t* mask, int* episode, int* pending, int mode,
                                      int n_env) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_env || !mask[e]) return;
  const bool hit = mode == 1 ? counts[kCountStride * e] > 0 : counts[kCountStride * e + 4] != 0;
  if (hit || terminated[e] || truncated[e]) { episode[e]++; atomicAdd(pending, 1); }
  else mask[e] = 0;
}


// One dense layer of the policy MLP on the matrix cores: Y[M][N] = act(X[M][K] W[K][N] + b[N]).
// One wave per 32x32 output tile, K swept two columns per v_mfma_f32_32x32x2_f32 (exact f32); the X tile
// is staged through LDS (row stride K+1: conflict-free A-operand reads), W streams from L2 coalesced.
__global__ __launch_bounds__(kGroup) void hb_mlp_layer_kernel(const float* X, const float* W, const float* bias, float* Y, int Mrows, int K, int N, int act) {
  extern __shared__ float xs[];
  const int lane = threadIdx.x, m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int ks = K + 1;
  for (int idx = lane; idx < 32 * K; idx += kGroup) {
    const int r = idx / K, c = idx - r * K;
    xs[r * ks + c] = (m0 + r < Mrows) ? X[(size_t)(m0 + r) * K + c] : 0.f;
  }
  __syncthreads();
  const int col = lane & 31, half = lane >> 5;
  const bool nvld = n0 + col < N;
  f32x16 D;
#pragma unroll
  for (int r = 0; r < 16; r++) D[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 2) {
    const int k = k0 + half;
    const float a = k < K ? xs[col * ks + k] : 0.f;
    const float bv = (nvld && k < K) ? W[(size_t)k * N + n0 + col] : 0.f;
    D = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, D, 0, 0, 0);
  }
  const float bn = nvld ? bias[n0 + col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (row < Mrows && nvld) {
      float v = D[r] + bn;
      Y[(size_t)row * N + n0 + col] = act ? tanhf(v) : v;
    }
  }
}

// The whole policy in one launch: observation -> every MLP layer -> controls, for 16 envs per block (4096 envs = 256
// blocks: one per CU; f32 MFMA throughput per CU is the bound, so the batch is spread over the whole chip).
// Activations never leave LDS (two ping-pong tiles of 16 rows); eight waves share the 16-column output tiles of a
// layer, each sweeping K four columns per v_mfma_f32_16x16x4_f32 (exact f32) with the A operand from LDS and the
// B operand from weights pre-packed on the host in operand order (one coalesced 256-byte wave load per MFMA:
// wp[tile][k/4][lane] = W[4(k/4) + lane/16][16 tile + lane%16]).  A layer with fewer than eight tiles (the
// nu-wide output layer) splits K across the idle waves instead; the partial tiles are summed in a fixed order
// (deterministic, no atomics).
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void hb_policy_kernel(const DevModel M, const PolicyDesc pd, const float* state, float* ctrl, int n_env) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 16, ldx = pd.ldx;
  float* cur = sm;
  float* nxt = sm + 16 * ldx;
  float* part = sm + 32 * ldx;  // [8][16][16] partial tiles
  // observation tile: 32 threads per env row gather the copied entries through the gather table (independent loads,
  // all in flight at once), one thread per row derives the gravity direction from the root quaternion
  {
    const int row = tid >> 5, sub = tid & 31;
    float* o = cur + row * ldx;
    const bool live = m0 + row < n_env;
    const float* s = state + (size_t)(m0 + row) * M.nstate;
    const int ncopy = M.nobs - 3;
    for (int k = sub; k < ncopy; k += 32) {
      const int src = M.obs_src[k];
      o[k] = (live && src >= 0) ? s[src] : 0.f;
    }
    if (sub == 0) {
      Q4 q = {1.f, 0.f, 0.f, 0.f};
      if (live && M.obs_root_qadr >= 0) q = qnormalize(ldq(s + 1 + M.obs_root_qadr + 3));
      float mm[9];
      q2mat(mm, q);
      o[ncopy] = live ? -mm[6] : 0.f; o[ncopy + 1] = live ? -mm[7] : 0.f; o[ncopy + 2] = live ? -mm[8] : 0.f;
      for (int k = M.nobs; k < M.nobs + 3; k++) o[k] = 0.f;  // K is swept four at a time: the pad columns must be finite
    }
  }
  __syncthreads();
  const int col = lane & 15, quad = lane >> 4;  // A: row = col, k offset = quad;  B: k offset = quad, column = col;  D: rows 4 quad + r, column col
  for (int l = 0; l < pd.nl; l++) {
    const int K = pd.sizes[l], N = pd.sizes[l + 1], KK = (K + 3) / 4, ntile = (N + 15) / 16;
    const bool last = l + 1 == pd.nl;
    int S = 1;  // K slices per tile
    while (S * 2 * ntile <= 8) S *= 2;
    const float* wp = pd.w[l];
    const float* bias = pd.b[l];
    if (!last && tid < 48) nxt[(tid / 3) * ldx + N + tid % 3] = 0.f;  // pad columns of the next layer's input
    for (int it = wave; it < ntile * S; it += 8) {
      const int nt = it / S, sl = it - nt * S;
      const int kb = KK * sl / S, ke = KK * (sl + 1) / S;
      f32x4v D = {0.f, 0.f, 0.f, 0.f};
      const float* ap = cur + col * ldx + quad;
      const float* bp = wp + (size_t)nt * KK * 64 + lane;
      int kk = kb;
      for (; kk + 8 <= ke; kk += 8) {  // eight operand pairs in flight per batch of MFMAs
        float a[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { a[u] = ap[4 * (kk + u)]; w[u] = bp[(size_t)(kk + u) * 64]; }
#pragma unroll
        for (int u = 0; u < 8; u++) D = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], w[u], D, 0, 0, 0);
      }
      for (; kk < ke; kk++) D = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * kk], bp[(size_t)kk * 64], D, 0, 0, 0);
      const int n = nt * 16 + col;
      if (S == 1) {
        const float bn = n < N ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * quad + r;
          if (n < N) {
            const float v = tanhf(D[r] + bn);
            if (!last) nxt[row * ldx + n] = v;
            else if (m0 + row < n_env) ctrl[(size_t)(m0 + row) * N + n] = v;
          }
        }
      } else {
        float* pp = part + it * 256;
#pragma unroll
        for (int r = 0; r < 4; r++) pp[(4 * quad + r) * 16 + col] = D[r];
      }
    }
    __syncthreads();
    if (S > 1) {
      for (int idx = tid; idx < ntile * 256; idx += 512) {
        const int nt = idx >> 8, rc = idx & 255, row = rc >> 4, n = nt * 16 + (rc & 15);
        if (n < N) {
          float v = bias[n];
          for (int sl = 0; sl < S; sl++) v += part[(nt * S + sl) * 256 + rc];
          v = tanhf(v);
          if (!last) nxt[row * ldx + n] = v;
          else if (m0 + row < n_env) ctrl[(size_t)(m0 + row) * N + n] = v;
        }
      }
      __syncthreads();
    }
    float* t = cur; cur = nxt; nxt = t;
  }
}

// The same policy without LDS and inside 64 VGPRs: four waves per block of sixteen envs, activations ping-pong through an L2-resident
// scratch (the waves of a block share the CU's vector L1).  Two step-kernel waves per SIMD leave 64 VGPRs, six wave slots and no LDS:
// blocks of THIS kernel run beside them (measured: 6 us slower beside a chip full of step waves than alone), where the LDS variant
// (33 KB per block) waits for two step blocks of a CU to retire (config 4, pipelined: its 10 us became 43).
// Activations are stored in the A-operand order of v_mfma_f32_16x16x4_f32 - element (env row, k) at [k / 4][k % 4][row] - so that a
// k-step's operand is one contiguous 256-byte wave load like the host-packed weights (row-major rows 260 floats apart cost sixteen
// cache lines per load: 52 us for 4096 envs).
// (amdgpu_num_vgpr counts per half of the unified register file: 32 -> 64 registers in all, tools/kernel_resources.sh; with 48 - what
// is left beside two 232-register waves - the kernel spills 23 values and the loop runs 3.10e7 instead of 3.20e7 env-steps/s)
__attribute__((amdgpu_num_vgpr(32))) __global__ __launch_bounds__(256) void hb_policy_lean_kernel(const DevModel M, const PolicyDesc pd, const float* state, float* ctrl,
                                                                                                  float* act, int n_env) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 16, ldx = pd.ldx;
  float* cur = act + (size_t)blockIdx.x * 32 * ldx;  // [ldx / 4][4][16]
  float* nxt = cur + 16 * ldx;
  {
    const int row = tid & 15, sub = tid >> 4;  // sixteen threads per observation column stride, consecutive threads = consecutive rows
    const bool live = m0 + row < n_env;
    const float* s = state + (size_t)(live ? m0 + row : 0) * M.nstate;
    const int ncopy = M.nobs - 3;
    for (int k = sub; k < ncopy; k += 16) {
      const int src = M.obs_src[k];
      cur[(k >> 2) * 64 + (k & 3) * 16 + row] = (live && src >= 0) ? s[src] : 0.f;
    }
    if (sub == 0) {
      Q4 q = {1.f, 0.f, 0.f, 0.f};
      if (live && M.obs_root_qadr >= 0) q = qnormalize(ldq(s + 1 + M.obs_root_qadr + 3));
      // third row of the rotation matrix of q (q2mat's m[6..8])
      const float m6 = 2.f * (q.x * q.z - q.w * q.y), m7 = 2.f * (q.y * q.z + q.w * q.x), m8 = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
      const float g3[3] = {live ? -m6 : 0.f, live ? -m7 : 0.f, live ? -m8 : 0.f};
      for (int c = 0; c < 3; c++) { const int k = ncopy + c; cur[(k >> 2) * 64 + (k & 3) * 16 + row] = g3[c]; }
      for (int k = M.nobs; k < ((M.nobs + 3) & ~3); k++) cur[(k >> 2) * 64 + (k & 3) * 16 + row] = 0.f;  // K is swept four at a time: the pad columns must be finite
    }
  }
  __syncthreads();
  const int col = lane & 15, quad = lane >> 4;  // B: k offset = quad, column = col;  D: rows 4 quad + r, column col
  for (int l = 0; l < pd.nl; l++) {
    const int K = pd.sizes[l], N = pd.sizes[l + 1], KK = (K + 3) / 4, ntile = (N + 15) / 16;
    const bool last = l + 1 == pd.nl;
    const float* wp = pd.w[l];
    const float* bias = pd.b[l];
    if (!last && tid < 16 * (((N + 3) & ~3) - N)) { const int k = N + tid / 16; nxt[(k >> 2) * 64 + (k & 3) * 16 + (tid & 15)] = 0.f; }  // pad columns of the next layer's input
    // two output tiles per wave and pass: they share the A operand, and their accumulators are two independent MFMA chains
    for (int nt = 2 * wave; nt < ntile; nt += 8) {
      const bool two = nt + 1 < ntile;
      f32x4v D0 = {0.f, 0.f, 0.f, 0.f}, D1 = {0.f, 0.f, 0.f, 0.f};
      const float* ap = cur + lane;
      const float* bp0 = wp + (size_t)nt * KK * 64 + lane;
      const float* bp1 = bp0 + (two ? (size_t)KK * 64 : 0);
      int kk = 0;
      for (; kk + 4 <= KK; kk += 4) {  // four k-steps of operands in flight per batch of MFMAs (measured: three are slower; loading the next batch under this one's MFMAs changes nothing)
        float a[4], w0[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = ap[(kk + u) * 64]; w0[u] = bp0[(kk + u) * 64]; w1[u] = bp1[(kk + u) * 64]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          D0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], w0[u], D0, 0, 0, 0);
          D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], w1[u], D1, 0, 0, 0);
        }
      }
      for (; kk < KK; kk++) {
        const float a = ap[kk * 64];
        D0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp0[kk * 64], D0, 0, 0, 0);
        D1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp1[kk * 64], D1, 0, 0, 0);
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if (h == 1 && !two) break;
        const int n = (nt + h) * 16 + col;
        const float bn = n < N ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = 4 * quad + r;
          if (n < N) {
            const float v = tanhf((h ? D1[r] : D0[r]) + bn);
            if (!last) nxt[(n >> 2) * 64 + (n & 3) * 16 + row] = v;
            else if (m0 + row < n_env) ctrl[(size_t)(m0 + row) * N + n] = v;
          }
        }
      }
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
}

// Probe for hb_batch_pipeline: one wave that idles for `ticks` of the 100 MHz wall clock (bounded by the sleep count as well) and
// records when it began and ended.  Two of these on two streams overlap in time exactly when the streams own different hardware queues.
__global__ __launch_bounds__(64) void hb_probe_spin_kernel(unsigned long long* out, unsigned ticks) {
  const unsigned long long t0 = wall_clock64();
  unsigned long long t = t0;
  for (int guard = 0; guard < 2048 && t - t0 < ticks; guard++) {
    __builtin_amdgcn_s_sleep(64);
    t = wall_clock64();
  }
  if (threadIdx.x == 0) { out[0] = t0; out[1] = t; }
}
// Heavy-first dispatch order for the next launch: counting sort of the envs by the cost proxy of their
// last step (constraint rows x solver sweeps, counts[4e+3]), most expensive first (LPT scheduling of
// the 4096 blocks over the resident slots).  One block; the order inside a cost bin is arbitrary,
// which cannot change results (envs are independent).
__global__ __launch_bounds__(1024) void hb_order_kernel(const int* counts, int* order, int* keys, int e0, int n, int slot, int shift) {
  // sorts envs e0 .. e0+n-1 into order[e0 .. e0+n-1], most expensive first; cost = counts[env][slot] >> shift, 256 bins
  __shared__ int hist[256];
  __shared__ int base[256];
  const int tid = threadIdx.x;
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  for (int e = e0 + tid; e < e0 + n; e += blockDim.x) {
    int key = min(255, counts[kCountStride * e + slot] >> shift);
    keys[e] = key;  // read ONCE: the slow lane of two-lane stepping may be writing counts beside this kernel, and a key that changed
                    // between the two passes would leave an env out of the permutation
    atomicAdd(&hist[255 - key], 1);  // bin 0 = most expensive
  }
  __syncthreads();
  // exclusive prefix sum of the 256 bins (Hillis-Steele on 256 threads: 8 rounds instead of a 256-step serial loop
  // on one thread, which was most of this kernel's 9 us on the critical path of every fourth step)
  if (tid < 256) base[tid] = hist[tid];
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    int v = 0;
    if (tid < 256 && tid >= o) v = base[tid - o];
    __syncthreads();
    if (tid < 256) base[tid] += v;
    __syncthreads();
  }
  if (tid < 256) base[tid] += e0 - hist[tid];  // inclusive -> exclusive, offset by the segment start
  __syncthreads();
  for (int e = e0 + tid; e < e0 + n; e += blockDim.x) {
    const int key = keys[e];
    order[atomicAdd(&base[255 - key], 1)] = e;
  }
}

// benchmark controls: ctrl[t][e][i] = 2*H(1+t0+t+1000*(env_offset+e), i+2) - 1  (testspeed.cc:64-80)
__global__ void hb_halton_ctrl_kernel(float* out, int T, int n_env, int nu, int t0, int env_offset) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)T * n_env * nu;
  if (idx >= total) return;
  int i = (int)(idx % nu);
  size_t r = idx / nu;
  int e = (int)(r % n_env), t = (int)(r / n_env);
  out[idx] = 2.f * halton(1 + t0 + t + 1000 * (env_offset + e), i + 2) - 1.f;
}

hipError_t launch_reset(const DevModel& M, float* state, int* status, const uint8_t* mask, const float* qpos_src, const int* episode, int n_env, float perturb,
                        int env_offset, hipStream_t stream, float quat_perturb) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  hipLaunchKernelGGL(hb_reset_kernel, dim3((n_env + 255) / 256), dim3(256), 0, stream, M, state, status, mask, qpos_src, episode, n_env, perturb, env_offset, quat_perturb);
  return hipGetLastError();
}
hipError_t launch_envrand_reset(const DevModel& M, const EnvRand& R, const EnvRandState& S, const int* episode, const uint8_t* mask, int n_env, int env_offset,
                                hipStream_t stream) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  hipLaunchKernelGGL(hb_envrand_reset_kernel, dim3((n_env + 127) / 128), dim3(128), 0, stream, M, R, S, episode, mask, n_env, env_offset);
  return hipGetLastError();
}
hipError_t launch_action_env(const DevModel& M, const EnvRand& R, const EnvRandState& S, const float* action, float* prev, float* latest, float* ctrl,
                             const int* episode, const float* state, const uint8_t* mask, int n_env, int env_offset, hipStream_t stream) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  hipLaunchKernelGGL(hb_action_env_kernel, dim3((n_env + 15) / 16), dim3(256), 0, stream, M, R, S, action, prev, latest, ctrl, episode, state, mask, n_env,
                     env_offset);
  return hipGetLastError();
}
hipError_t launch_reset_check(const int* counts, const uint8_t* terminated, const uint8_t* truncated, uint8_t* mask, int* episode, int* pending, int mode, int n_env,
                              hipStream_t stream) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  hipLaunchKernelGGL(hb_reset_check_kernel, dim3((n_env + 255) / 256), dim3(256), 0, stream, counts, terminated, truncated, mask, episode, pending, mode, n_env);
  return hipGetLastError();
}
hipError_t launch_obs(const DevModel& M, const float* state, float* obs, int n_env, hipStream_t stream) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  hipLaunchKernelGGL(hb_obs_kernel, dim3((n_env + 255) / 256), dim3(256), 0, stream, M, state, obs, n_env);
  return hipGetLastError();
}
hipError_t launch_action(const float* action, float* prev, float* latest, float* ctrl, int n, hipStream_t stream) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  hipLaunchKernelGGL(hb_action_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, action, prev, latest, ctrl, n);
  return hipGetLastError();
}
hipError_t launch_env(const DevModel& M, const EnvConfig& cfg, const EnvRand& R, const EnvRandState& S, float* state, const float* qfrc, const int* counts, float* prev,
                      float* latest, const float* qpos_src, int* episode, int* status, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated,
                      const uint8_t* mask, int observe, const DomainRand& D, float* dr, int dr_stride, int n_env, int env_offset, hipStream_t stream, float* term_obs, int* seen) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  const int per_block = 256 / kEnvLanes;
  hipLaunchKernelGGL(hb_env_kernel, dim3((n_env + per_block - 1) / per_block), dim3(256), (size_t)per_block * ((M.nstate + 3) & ~3) * sizeof(float), stream, M, cfg, R, S, state, qfrc, counts, prev, latest, qpos_src, episode, status, obs,
                     reward, terminated, truncated, mask, observe, D, dr, dr_stride, n_env, env_offset, term_obs, seen);
  return hipGetLastError();
}
hipError_t launch_domain_rand(const DevModel& M, const DomainRand& D, float* dr, int stride, const int* episode, const uint8_t* mask, int n_env, int env_offset,
                              hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(hb_domain_rand_kernel, dim3((n_env + 256 / kDrawLanes - 1) / (256 / kDrawLanes)), dim3(256), 0, stream, M, D, dr, stride, episode, mask, n_env, env_offset);
  return hipGetLastError();
}
hipError_t launch_probe_spin(unsigned long long* out, unsigned ticks, hipStream_t stream) {
  hipLaunchKernelGGL(hb_probe_spin_kernel, dim3(1), dim3(64), 0, stream, out, ticks);
  return hipGetLastError();
}
hipError_t launch_order(const int* counts, int* order, int n_env, int e0, int n, hipStream_t stream, int slot, int shift) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(hb_order_kernel, dim3(1), dim3(1024), 0, stream, counts, order, order + n_env, e0, n, slot, shift);
  return hipGetLastError();
}
hipError_t launch_mlp_layer(const float* X, const float* W, const float* bias, float* Y, int Mrows, int K, int N, int act, hipStream_t stream) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  hipLaunchKernelGGL(hb_mlp_layer_kernel, dim3((Mrows + 31) / 32, (N + 31) / 32), dim3(kGroup), (size_t)32 * (K + 1) * sizeof(float), stream, X, W, bias, Y, Mrows, K, N, act);
  return hipGetLastError();
}
hipError_t launch_policy_lean(const DevModel& M, const PolicyDesc& pd, const float* state, float* ctrl, float* act, int n_env, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(hb_policy_lean_kernel, dim3((n_env + 15) / 16), dim3(256), 0, stream, M, pd, state, ctrl, act, n_env);
  return hipGetLastError();
}
hipError_t launch_policy(const DevModel& M, const PolicyDesc& pd, const float* state, float* ctrl, int n_env, hipStream_t stream) {
  const size_t shmem = ((size_t)32 * pd.ldx + 8 * 256) * sizeof(float);
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  hipLaunchKernelGGL(hb_policy_kernel, dim3((n_env + 15) / 16), dim3(512), shmem, stream, M, pd, state, ctrl, n_env);
  return hipGetLastError();
}
hipError_t launch_halton_ctrl(float* out, int T, int n_env, int nu, int t0, int env_offset, hipStream_t stream) {
  size_t total = (size_t)T * n_env * nu;
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  hipLaunchKernelGGL(hb_halton_ctrl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, out, T, n_env, nu, t0, env_offset);
  return hipGetLastError();
}
hipError_t launch_stand_cost(const float* rows, int H, int n_env, const StandTask& K, const int* status, float* total, float* costs, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(hb_stand_cost_kernel, dim3((n_env + 127) / 128), dim3(128), 0, stream, rows, H, n_env, K, status, total, costs);
  return hipGetLastError();
}
hipError_t launch_walk_cost(const float* rows, int H, int n_env, const WalkTask& K, const int* status, float* total, float* costs, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(hb_walk_cost_kernel, dim3((n_env + 63) / 64), dim3(64), 0, stream, rows, H, n_env, K, status, total, costs);
  return hipGetLastError();
}
hipError_t launch_cost_terms(const float* residual, int n, int nres, const CostSpec& K, float* terms, float* cost, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(hb_cost_terms_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, residual, n, nres, K, terms, cost);
  return hipGetLastError();
}
hipError_t launch_spline_tape(const DevModel& M, const float* knots, const float* times, int P, int interp, float time0, float dt, int T, int n_env, float* tape, hipStream_t stream) {
  (void)hipGetLastError();
  const size_t total = (size_t)T * n_env * M.nu;
  hipLaunchKernelGGL(hb_spline_tape_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, M, knots, times, P, interp, time0, dt, T, n_env, tape);
  return hipGetLastError();
}
}  // namespace hb
