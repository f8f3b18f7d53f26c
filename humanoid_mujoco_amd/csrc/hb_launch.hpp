// hb_launch.hpp — host-callable launchers implemented in the kernel translation units (hb_step.hip, hb_narrow.hip, hb_env.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "hb_device.hpp"
namespace hb {
hipError_t launch_step(const DevModel* M_dev, int variant, int solver, int nv, int lds_floats, const BatchPtrs& P, int nsteps, hipStream_t stream);
// two envs per wave: a lean single-step launch of the 27-dof humanoid's PGS kernel (hb_step_duo.hip; chosen by launch_step)
hipError_t launch_step_duo(const DevModel* M_dev, const BatchPtrs& P, int nsteps, hipStream_t stream);
// name of the step kernel the last launch_step of this thread launched last (hb_last_kernel)
const char* last_step_kernel();
bool fold_pays(int variant, int solver, int nv, const BatchPtrs& P);
// the pose + narrowphase launches of one staged step (hb_narrow.hip)
hipError_t launch_pose_narrow(const DevModel* M_dev, const BatchPtrs& Q, hipStream_t stream);
hipError_t launch_reset(const DevModel& M, float* state, int* status, const uint8_t* mask, const float* qpos_src, const int* episode, int n_env, float perturb,
                        int env_offset, hipStream_t stream, float quat_perturb = 0.f);
hipError_t launch_envrand_reset(const DevModel& M, const EnvRand& R, const EnvRandState& S, const int* episode, const uint8_t* mask, int n_env, int env_offset,
                                hipStream_t stream);
hipError_t launch_action_env(const DevModel& M, const EnvRand& R, const EnvRandState& S, const float* action, float* prev, float* latest, float* ctrl,
                             const int* episode, const float* state, const uint8_t* mask, int n_env, int env_offset, hipStream_t stream);
hipError_t launch_reset_check(const int* counts, const uint8_t* terminated, const uint8_t* truncated, uint8_t* mask, int* episode, int* pending, int mode, int n_env,
                              hipStream_t stream);
hipError_t launch_obs(const DevModel& M, const float* state, float* obs, int n_env, hipStream_t stream);
hipError_t launch_action(const float* action, float* prev, float* latest, float* ctrl, int n, hipStream_t stream);
hipError_t launch_env(const DevModel& M, const EnvConfig& cfg, const EnvRand& R, const EnvRandState& S, float* state, const float* qfrc, const int* counts, float* prev,
                      float* latest, const float* qpos_src, int* episode, int* status, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated,
                      const uint8_t* mask, int observe, const DomainRand& D, float* dr, int dr_stride, int n_env, int env_offset, hipStream_t stream, float* term_obs = nullptr, int* seen = nullptr);
hipError_t launch_domain_rand(const DevModel& M, const DomainRand& D, float* dr, int stride, const int* episode, const uint8_t* mask, int n_env, int env_offset,
                              hipStream_t stream);
hipError_t launch_policy(const DevModel& M, const PolicyDesc& pd, const float* state, float* ctrl, int n_env, hipStream_t stream);
hipError_t launch_policy_lean(const DevModel& M, const PolicyDesc& pd, const float* state, float* ctrl, float* act, int n_env, hipStream_t stream);
// order: [2 * n_env] ints - the permutation, then the sort keys of the pass (read once from counts)
hipError_t launch_order(const int* counts, int* order, int n_env, int e0, int n, hipStream_t stream, int slot = 3, int shift = 3);
hipError_t launch_mlp_layer(const float* X, const float* W, const float* bias, float* Y, int Mrows, int K, int N, int act, hipStream_t stream);
hipError_t launch_halton_ctrl(float* out, int T, int n_env, int nu, int t0, int env_offset, hipStream_t stream);
hipError_t launch_stand_cost(const float* rows, int H, int n_env, const StandTask& K, const int* status, float* total, float* costs, hipStream_t stream);
hipError_t launch_walk_cost(const float* rows, int H, int n_env, const WalkTask& K, const int* status, float* total, float* costs, hipStream_t stream);
hipError_t launch_cost_terms(const float* residual, int n, int nres, const CostSpec& K, float* terms, float* cost, hipStream_t stream);
hipError_t launch_spline_tape(const DevModel& M, const float* knots, const float* times, int P, int interp, float time0, float dt, int T, int n_env, float* tape, hipStream_t stream);
hipError_t launch_probe_spin(unsigned long long* out, unsigned ticks, hipStream_t stream);
hipError_t set_step_lds_limit(int bytes);
}  // namespace hb
