/* hb.h — C-ABI of the MI355X batched humanoid physics-step / rollout engine (libhb.so).
 *
 * Drop-in boundary for the reference's physics-step path.  Every entry point
 * cites the reference interface it replaces; paths are relative to the reference repository
 * (mcgill-robotics/Humanoid-MuJoCo), MuJoCo C API = simulation/mujoco/include/mujoco/mujoco.h.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in any signature.
 *   - every function returns an int status (HB_OK == 0, negative = error) or a handle
 *     (NULL on failure, message in the caller's err buffer).  Nothing aborts the process
 *     (the reference's mju_error would, mujoco.h:810).
 *   - "env-major" arrays are [n_env][width], row e belonging to environment e.
 *   - *_dev variants take DEVICE pointers (hipMalloc / torch tensor data_ptr) on the batch's
 *     device and enqueue on the batch's stream without synchronising; the plain variants take
 *     HOST pointers, copy, and synchronise before returning.
 *   - there is no CPU backend: hb_batch_create fails (HB_ENODEVICE) without a HIP device.
 */
#ifndef HB_H_
#define HB_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HB_OK 0
#define HB_EINVAL (-1)    /* bad argument */
#define HB_ENODEVICE (-2) /* no usable HIP device / HIP call failed */
#define HB_ENOMEM (-3)
#define HB_EUNSUPPORTED (-4)
#define HB_EIO (-5)

/* per-env status bits, the batched counterpart of mjData.warning[] (mjdata.h:54-65,185) */
#define HB_WARN_CONTACTFULL (1 << 1) /* mjWARN_CONTACTFULL: contact buffer overflow, extras dropped */
#define HB_WARN_CNSTRFULL (1 << 2)   /* mjWARN_CNSTRFULL: constraint rows overflow, extras dropped */
#define HB_WARN_BADQPOS (1 << 4)     /* mjWARN_BADQPOS: NaN/huge qpos, env was reset (mujoco.h:301) */
#define HB_WARN_BADQVEL (1 << 5)     /* mjWARN_BADQVEL (mujoco.h:304) */
#define HB_WARN_BADQACC (1 << 6)     /* mjWARN_BADQACC (mujoco.h:307) */

/* state specification bits — identical to mjtState (mjdata.h:27-50); hb_get_state/hb_set_state
 * concatenate the selected components per env in ascending bit order, as mj_getState does. */
#define HB_STATE_TIME (1 << 0)
#define HB_STATE_QPOS (1 << 1)
#define HB_STATE_QVEL (1 << 2)
#define HB_STATE_WARMSTART (1 << 4)
#define HB_STATE_CTRL (1 << 5)
#define HB_STATE_XFRC_APPLIED (1 << 7)
#define HB_STATE_PHYSICS (HB_STATE_QPOS | HB_STATE_QVEL)
#define HB_STATE_INTEGRATION (HB_STATE_TIME | HB_STATE_QPOS | HB_STATE_QVEL | HB_STATE_WARMSTART)

typedef struct hb_model hb_model; /* immutable compiled model; shareable across batches/devices */
typedef struct hb_batch hb_batch; /* n_env independent simulations resident on one GPU */

/* The fields of mjOption this engine honours (mjmodel.h:403-445). */
typedef struct hb_options {
  double timestep;
  double gravity[3];
  double impratio;
  double tolerance;  /* solver early-exit threshold on the scaled cost improvement (Newton: also on the scaled gradient) */
  int iterations;    /* PGS sweep cap / Newton iteration cap */
  int solver;        /* 0 = PGS (mjSOL_PGS), 2 = Newton (mjSOL_NEWTON, the reference's default); CG is not implemented */
  int cone;          /* 0 = pyramidal; the only cone implemented */
  int integrator;    /* 0 = Euler (semi-implicit, implicit joint damping) */
  int disableflags;  /* mjtDisableBit (mjmodel.h:50-68) */
  int ls_iterations;   /* Newton: cap on line-search evaluations per iteration (mjOption.ls_iterations, mjmodel.h:434) */
  double ls_tolerance; /* Newton: line-search slope tolerance relative to `tolerance` (mjmodel.h:411) */
} hb_options;

/* sizes a caller needs to allocate buffers (mjModel.nq/nv/nu/..., mjmodel.h:560-620) */
typedef struct hb_sizes {
  int nq, nv, nu, nbody, njnt, ngeom, ntendon, nM, nkey, npair;
  int nobs;     /* width of the env observation (hb_get_obs) */
  int ncon_max; /* contact capacity per env */
  int nefc_max; /* constraint-row capacity per env */
} hb_sizes;

/* ---- model --------------------------------------------------------------------------------- */

/* Replaces mj_loadXML (mujoco.h:103) / mj_loadModel (mujoco.h:163).  `path` ends in ".xml"
 * (MJCF subset, compiled on the host) or ".hbm" (this engine's compiled text model). */
hb_model* hb_model_load(const char* path, char* err, int err_sz);
/* Replaces mj_loadXML with an in-memory string (mjVFS use in the reference). */
hb_model* hb_model_load_xml_string(const char* xml, char* err, int err_sz);
/* Replaces mj_saveModel (mujoco.h:159). */
int hb_model_save(const hb_model* m, const char* path, char* err, int err_sz);
/* Replaces mj_deleteModel (mujoco.h:166). */
void hb_model_free(hb_model* m);
int hb_model_sizes(const hb_model* m, hb_sizes* out);
/* mjOption get/set (model->opt in the reference, e.g. simulation/cpu_env.py:87 sets
 * opt.timestep).  Options are copied into a batch at hb_batch_create. */
int hb_options_get(const hb_model* m, hb_options* out);
int hb_options_set(hb_model* m, const hb_options* in);
/* The order in which mj_collision emits contacts = the order of the model's candidate geom pairs.  Physically irrelevant, but PGS cut at a
 * finite sweep count depends on it (up to 4e-2 in qacc on the benchmark workload at 50 sweeps, 3e-14 for converged Newton:
 * profiles/r03_contact_order.txt).  order 1 (what the MJCF compiler writes): body pairs ascending, then the geoms of the first and of the
 * second body - MuJoCo's collision driver enumerates body pairs, then the geoms inside a pair (the documented structure of
 * engine_collision_driver; the pinned MuJoCo 3.1.4 is not in the reference tree to check the details against); order 0: geom pairs
 * ascending (this engine's rounds 1-3, and .hbm files written then).  -1: query.  Returns the order in force.  Call before
 * hb_batch_create. */
int hb_model_pair_order(hb_model* m, int order);
/* mj_name2id (mujoco.h:516) for kind in {"body","joint","geom","actuator","tendon","key"}; -1 if absent. */
int hb_model_name2id(const hb_model* m, const char* kind, const char* name);
/* Copies a named fp64 model array (mjModel field name, e.g. "body_mass", "qpos0") into out[cap];
 * returns its length, or HB_EINVAL. */
int hb_model_get_array(const hb_model* m, const char* field, double* out, int cap);

/* ---- batch --------------------------------------------------------------------------------- */

/* Replaces n_env calls of mj_makeData (mujoco.h:173).  device = HIP ordinal (>= 0). */
hb_batch* hb_batch_create(const hb_model* m, int n_env, int device, char* err, int err_sz);
/* Replaces mj_deleteData (mujoco.h:206). */
void hb_batch_free(hb_batch* b);
int hb_batch_n_env(const hb_batch* b);
/* The HIP stream (hipStream_t as void*) that orders all work of this batch.  Work the caller enqueues on
 * it before a step call (e.g. a policy writing the controls) is seen by that step; every hb_* call is
 * ordered after the steps enqueued before it.  The call itself performs hb_batch_join. */
void* hb_batch_stream(hb_batch* b);
int hb_batch_sync(hb_batch* b);
/* Pipelined stepping (off by default; on = 1: the default count, on = 2..8: that many segments).  The default is three segments -
 * the first on the batch's own stream - when their streams are seen to run kernels side by side, two when the process' other streams
 * leave them no hardware queue each (ROCm shares GPU_MAX_HW_QUEUES = 4 queues among all streams of a process; the call probes
 * with three idle waves of 60 us, once).  More than four concurrently active queues are slower whatever the setting.  When on, hb_step_dev /
 * hb_rollout*_dev cut the batch into fixed env segments and step each with its own launch on an internal stream: segment c of step t+1
 * follows only segment c of step t, so the slowest envs of one step overlap the start of the next
 * (envs are independent: results are identical to the unpipelined launch).  The internal streams fork
 * from the batch's stream at every step call and are joined back by the next hb_* call of any other
 * kind.  A caller that enqueues its OWN work on hb_batch_stream() after step calls must call
 * hb_batch_join first (or fetch the stream again); callers that only use hb_* functions need nothing.
 * A pipelined batch also folds hb_step_dev calls made back to back into one launch (see hb_step_dev): the same rule covers it. */
int hb_batch_pipeline(hb_batch* b, int on);
/* Number of env segments step calls are cut into at the moment (1: unpipelined). */
int hb_batch_segments(const hb_batch* b);
int hb_batch_join(hb_batch* b);

/* Replaces mj_resetData (keyframe < 0) / mj_resetDataKeyframe (mujoco.h:180,186) for the envs
 * whose mask byte is non-zero (mask == NULL: all).  perturb != 0 adds the deterministic
 * Halton perturbation of SURVEY.md §8(d) (hinges +-0.2 rad, root z +0.1 m) indexed by
 * env_offset + e, mirroring CPUEnv.reset's joint/height randomisation (cpu_env.py:282-328). */
int hb_reset(hb_batch* b, const uint8_t* mask, int keyframe, int perturb, int env_offset);

/* Replaces `for e: mju_copy(d->ctrl, ...); mj_step(m, d)` (mujoco.h:120; call sites
 * simulation/cpu_env.py:683-684, simulation/mujoco/sample/testspeed.cc:93-96).
 * ctrl: env-major [n_env][nu] float32, applied for n_substeps consecutive steps. */
int hb_step(hb_batch* b, const float* ctrl, int n_substeps);
/* The same with the controls already in device memory, asynchronous: the call returns at once, its results are there after hb_batch_sync
 * or behind anything enqueued on hb_batch_stream later, and ctrl_dev must stay allocated and untouched until then.  On a PIPELINED batch
 * (hb_batch_pipeline) of a primitive-geometry model whose waves all fit on the chip at once (8 per CU: up to 2048 envs on MI355X, 4096
 * for the models with the two-envs-per-wave kernel, HB_TUNE_DUO), calls made back to back - nothing else of the batch's API in between -
 * are executed as ONE kernel launch of up to HB_TUNE_FOLD steps, step t reading the controls of call t: no env waits for the batch's
 * slowest one between steps (MI355X: 2048 envs 42 us per step against 64, 4096 envs 62 against 78).  The states are bit-identical to one
 * launch per call; the work starts when HB_TUNE_FOLD steps are held or when any other hb_* call of the batch (hb_batch_sync,
 * hb_batch_stream, hb_batch_join, a read) arrives. */
int hb_step_dev(hb_batch* b, const float* ctrl_dev, int n_substeps);

/* Replaces the open-loop rollout loops (mujoco_mpc/mjpc/trajectory.cc:141-179,
 * simulation/mujoco/sample/testspeed.cc:84-103): T steps in ONE launch with state resident
 * on chip; ctrl is [T][n_env][nu]; qpos_out (nullable) receives [T][n_env][nq] after each step. */
int hb_rollout(hb_batch* b, const float* ctrl, int T, float* qpos_out);
int hb_rollout_dev(hb_batch* b, const float* ctrl_dev, int T, float* qpos_out_dev);
/* Same rollout with controls generated on device from the Halton sequence of
 * testspeed.cc:64-80 (ctrl[t,e,i] = 2*H(1+t0+t+1000*(env_offset+e), i+2)-1): the benchmark
 * workload of SURVEY.md §8(d), no control tensor in HBM. */
int hb_rollout_halton(hb_batch* b, int T, int t0, int env_offset, float* qpos_out_dev);

/* Replaces mj_forward (mujoco.h:129): recompute everything up to qacc without integrating. */
int hb_forward(hb_batch* b, const float* ctrl);

/* ---- wire format of a state ------------------------------------------------------------ */

/* One env's state as the `State` message of the reference's gRPC agent service (mujoco_mpc/mjpc/grpc/agent.proto:75-83:
 * time = 1, qpos = 2, qvel = 3, act = 4, mocap_pos = 5, mocap_quat = 6, userdata = 7; doubles, repeated fields packed),
 * in protobuf wire encoding, so that existing clients' GetState / SetState payloads can be produced and consumed
 * without a protobuf dependency.  This engine has no activations, mocap bodies or user data: those fields are not
 * written, and a message that carries any of them is refused (HB_EUNSUPPORTED).
 * hb_state_to_proto returns the number of bytes the message takes (also when buf is NULL or cap is too small: call
 * twice), hb_state_from_proto sets the fields present (absent ones keep their values; qacc_warmstart is reset to 0 as
 * mj_setState leaves it for a new state). */
int hb_state_to_proto(hb_batch* b, int env, unsigned char* buf, int cap);
int hb_state_from_proto(hb_batch* b, int env, const unsigned char* buf, int len);

/* ---- planner rollouts (MJPC's Trajectory::Rollout, mujoco_mpc/mjpc/trajectory.cc:100-210) -------------------- */

/* The sensors MJPC's humanoid tasks build their residuals from (tasks/humanoid_cap/stand/task.xml:22-40): framepos of
 * up to 16 bodies or sites (a site is a body plus an offset in the body frame), and subtreecom / subtreelinvel (mj_subtreeVel, mujoco.h:346) of one kinematic tree, named by its root
 * body (a direct child of the world; < 0: none).  Per env and step the read-out is
 * [framepos 0 (3) | ... | subtreecom (3) | subtreelinvel (3)], hb_sensor_size() floats. */
#define HB_MAX_FRAMEPOS 16
typedef struct hb_sensor_spec {
  int n_framepos;
  int framepos_body[HB_MAX_FRAMEPOS];
  int subtree_body;
  float framepos_offset[HB_MAX_FRAMEPOS][3]; /* position in the body's frame: zeros = objtype "xbody" (the body frame); a site's pos = objtype
                                               "site"; the body's ipos (hb_model_get_array "body_ipos") = objtype "body" (inertial frame) */
  /* further read-outs, appended behind the above in this order (3 floats each): */
  int n_frameaxis;            /* framexaxis / framezaxis of a body frame (objtype "xbody") */
  int frameaxis_body[8];
  int frameaxis_which[8];     /* 0: x axis, 2: z axis */
  int n_framelinvel;          /* framelinvel, objtype "body": linear velocity of the body's inertial frame origin, world axes */
  int framelinvel_body[8];
  int n_subtreelinvel;        /* subtreelinvel of further bodies (any body of the model, not only a tree root) */
  int subtreelinvel_body[4];
} hb_sensor_spec;
int hb_sensor_size(const hb_sensor_spec* spec);
/* mj_setState of ONE state on every env: the N candidate action sequences of a sampling planner all start from the
 * current state (sampling/planner.cc:342-380).  `state` is one record of hb_state_size(b, spec) floats. */
int hb_set_state_broadcast(hb_batch* b, unsigned spec, const float* state);
int hb_set_state_broadcast_f64(hb_batch* b, unsigned spec, const double* state);
/* hb_rollout plus the sensor read-out of every step: sensor_out[t][e][:] holds the sensors mj_step evaluates at the
 * state BEFORE step t's integration (what mjData.sensordata holds after the t-th mj_step call).  Host pointers;
 * qpos_out nullable. */
int hb_rollout_sensors(hb_batch* b, const float* ctrl, int T, const hb_sensor_spec* spec, float* sensor_out, float* qpos_out);
/* SamplingPolicy::Action for every candidate on the device (mujoco_mpc/mjpc/planners/sampling/policy.cc:50-58 over
 * mjpc/spline/spline.cc:103-156,240-277): knots[e][k][nu] are the spline nodes of candidate e (host, [n_env][n_points][nu]),
 * times[n_points] the node times shared by the candidates, interpolation 0 = zero-order, 1 = linear, 2 = cubic Hermite
 * with finite-difference slopes.  The splines are sampled at time0 + t * timestep for t = 0..T-1, clamped to ctrlrange
 * and left on the device as the action tape of the rollouts that follow: pass HB_CTRL_TAPE as their `ctrl` (the
 * hb_rollout_task_* functions accept it; horizon - 1 <= T).  n_points * nu floats per candidate cross PCIe instead of
 * T * nu. */
#define HB_CTRL_TAPE ((const float*)(uintptr_t)1)
int hb_ctrl_tape_splines(hb_batch* b, const float* knots, const float* times, int n_points, int interpolation, double time0, int T);
/* The tape hb_ctrl_tape_splines left on the device, copied back: out[T][n_env][nu] (host), T <= the tape's length; HB_EINVAL when
 * there is no tape (any call that writes the batch's controls discards it).  This is TimeSpline::Sample for every candidate at
 * every step time, which is how the reference's own expectations for the spline (mujoco_mpc/mjpc/test/spline/spline_test.cc:52-157)
 * are checked against the device code (tests/test_gpu_mjpc_expectations.py). */
int hb_ctrl_tape_read(hb_batch* b, int T, float* out);
/* BaseResidualFn::CostTerms and CostValue (mujoco_mpc/mjpc/task.cc:71-110) for n residual vectors at once, on the device, with the
 * code the hb_rollout_task_* kernels use: term k = weight[k] * Norm(norm[k]; norm_p[k]) over the next dim[k] entries of the
 * residual (mjpc/norm.h:24-36), cost = the sum through the risk transformation (exp(risk c) - 1) / risk, or the sum itself when
 * |risk| < 1e-6.  residual[n][n_residual], terms[n][n_term] (nullable), cost[n]: host pointers.  sum(dim) must equal n_residual. */
typedef struct hb_cost_spec {
  int n_term;
  int dim[8];
  int norm[8];          /* mjpc::NormType */
  float weight[8];
  float norm_p[8][2];
  float risk;
} hb_cost_spec;
int hb_task_cost(hb_batch* b, const float* residual, int n, int n_residual, const hb_cost_spec* spec, float* terms, float* cost);
/* Trajectory::NoisyRollout's perturbation (mujoco_mpc/mjpc/trajectory.cc:147-156): before every step of the calls that
 * follow, every xfrc_applied entry of every env becomes rate * xfrc + scale * N(0, 1), rate = exp(-timestep / xfrc_rate),
 * scale = xfrc_std * sqrt(1 - rate^2) (an Ornstein-Uhlenbeck process with stationary deviation xfrc_std); xfrc_std = 0
 * switches it off (the xfrc_applied values stay as they are).  The draws are counter-based (seed, global env, call, step,
 * entry): reproducible, unlike the reference's absl::BitGen, and independent across envs. */
int hb_rollout_noise(hb_batch* b, float xfrc_std, float xfrc_rate, unsigned seed);
/* Open-loop rollout that records the trajectory the way MJPC's Trajectory::Rollout does (mujoco_mpc/mjpc/trajectory.cc:
 * 141-190): states after every step, qpos_out[t][e][nq] and qvel_out[t][e][nv] (host pointers, each nullable; times are
 * t0 + (t + 1) * timestep and the actions are the caller's own tape), and failed[e] = 1 when the env raised a bad-state
 * warning on the way (CheckWarnings, utilities.cc:787-799; nullable). */
int hb_rollout_trajectory(hb_batch* b, const float* ctrl, int T, float* qpos_out, float* qvel_out, int* failed);
/* ---- transition derivatives for the gradient-based planners ------------------------------------------------------
 * mjd_transitionFD (mujoco.h:1242-1251) for T (state, control) points at once, the way ModelDerivatives::Compute fans it
 * over its thread pool (mujoco_mpc/mjpc/planners/model_derivatives.cc:44-105): finite differences of the discrete
 * dynamics x' = step(x, u) in tangent-space coordinates, x = (qpos, qvel), dx = (dqpos in R^nv, dqvel),
 *     A[t] = d x'/d x   [2nv][2nv],     B[t] = d x'/d u   [2nv][nu]      (row-major doubles; either may be NULL).
 * Every perturbed copy of every point is one env of the batch: T * (1 + k * (2nv + nu)) envs are needed, k = 2 when
 * `centered`, else 1 (HB_EINVAL if the batch is smaller); all of them advance in ONE step launch.  x[t] = qpos[nq] |
 * qvel[nv]; warmstart[t][nv] (nullable: zeros) is used by the nominal and by every perturbed copy, as mjd_transitionFD
 * does.  Position perturbations and differences go through mj_integratePos / mj_differentiatePos (quaternion dofs in
 * the tangent space); a control is nudged inside its ctrlrange.  The arithmetic is fp32 on the device: eps should be
 * about 1e-3 (rounding in x' divided by eps is the noise floor), not the 1e-6 an fp64 caller would use.
 * The batch's states are overwritten. */
int hb_transition_fd(hb_batch* b, const double* x, const double* u, const double* warmstart, int T, double eps, int centered, double* A, double* B);
/* The same with the sensor derivatives of mjd_transitionFD: d(sensor) = C dx + D du, for the read-out row of `spec`
 * (hb_sensor_size(spec) values, evaluated in the step's forward pass, i.e. at (x, u) itself: the residual Jacobians a
 * gradient-based planner builds its cost derivatives from).  C[t][ns][2nv], D[t][ns][nu], either nullable. */
int hb_transition_fd_sensors(hb_batch* b, const double* x, const double* u, const double* warmstart, int T, double eps, int centered,
                             const hb_sensor_spec* spec, double* A, double* B, double* C, double* D);

/* ---- a planner iteration's cost evaluation on the device: MJPC's "Humanoid Stand" task ------------------------------
 * (mujoco_mpc/mjpc/tasks/humanoid/stand/{stand.cc:41-104, task.xml:14-36}; the two-foot variant of
 * tasks/humanoid_cap/stand is the same residual with n_feet = 2).  Residual, in order: Height (head z minus mean foot z
 * minus height_goal), Balance (distance in xy of the mean foot position from the capture point com + 0.2 s * com
 * velocity), CoM Vel. (xy), Joint Vel. (qvel[6:]), Control (ctrl); cost = sum_k weight[k] * Norm(norm[k]; norm_p[k])
 * (mjpc/norm.h:24-36, task.cc:71-110), risk transformation as in task.cc:104-109. */
typedef struct hb_task_stand {
  int head_body;
  int n_feet;              /* 1..4 foot frames (the reference's sites sp0..sp3): body + offset in the body frame */
  int foot_body[4];
  float foot_offset[4][3];
  int subtree_body;        /* root body of the tree whose subtreecom / subtreelinvel enter (torso) */
  float height_goal;       /* task parameter "Height Goal" */
  int norm[5];             /* mjpc::NormType per term */
  float weight[5];
  float norm_p[5][2];      /* norm parameters p, q */
  float risk;              /* task_risk (0: risk neutral) */
} hb_task_stand;
/* The values of the reference's task.xml for this model: bodies "head", "foot_left", "foot_right", "torso" by name, feet
 * offsets of sites sp0..sp3 (humanoid.xml.patch:170-171,217-218); HB_EINVAL if the model has no such bodies. */
int hb_task_stand_default(const hb_model* m, hb_task_stand* out);
/* Trajectory::Rollout + UpdateReturn (trajectory.cc:100-210,312-326) for N candidates at once: from the batch's current
 * state, horizon - 1 steps with ctrl[t][e][nu] (host, [horizon-1][n_env][nu]; the last action is repeated for the final
 * mj_forward, zero when horizon = 1), the residual of every one of the `horizon` states, its cost, and
 * total_return[e] = mean cost (1e6 for a candidate that raised a bad-state warning: kMaxReturnValue).  costs
 * ([horizon][n_env], nullable) receives the stage costs.  Nothing but these floats crosses PCIe on the way back.  The
 * per-env status words are cleared first: failure is a property of this rollout. */
int hb_rollout_task_stand(hb_batch* b, const float* ctrl, int horizon, const hb_task_stand* task, float* total_return, float* costs);

/* MJPC's "Humanoid Walk" task the same way (mujoco_mpc/mjpc/tasks/humanoid/walk/{walk.cc:44-163, task.xml:14-86}).
 * Residual, in order: torso height (1), pelvis/feet (1), balance (2: capture point against its projection onto the
 * segment between the feet), upright (8), posture (qpos[7:]), walk (1), move feet (2), control (nu).  The cost terms are
 * the task's user sensors IN THEIR ORDER with THEIR dimensions (Height 1, Pelvis/Feet 1, Balance 2, Upright 8, Posture
 * nq-7, Velocity 2, Walk 1, Control nu), applied to consecutive slices of that vector as BaseResidualFn::CostTerms does. */
typedef struct hb_task_walk {
  int torso_body, pelvis_body, foot_right_body, foot_left_body, waist_lower_body;
  float height_goal;  /* residual_Torso */
  float speed_goal;   /* residual_Speed */
  int n_term;
  int dim[8];
  int norm[8];
  float weight[8];
  float norm_p[8][2];
  float risk;
} hb_task_walk;
int hb_task_walk_default(const hb_model* m, hb_task_walk* out);
int hb_rollout_task_walk(hb_batch* b, const float* ctrl, int horizon, const hb_task_walk* task, float* total_return, float* costs);

/* The same read-out at the current state (mj_forward, no integration): the terminal residual of a trajectory. */
int hb_sensors(hb_batch* b, const float* ctrl, const hb_sensor_spec* spec, float* sensor_out);

/* Replaces mj_stateSize / mj_getState / mj_setState (mujoco.h:378-384). */
int hb_state_size(const hb_batch* b, unsigned spec);
int hb_get_state(hb_batch* b, unsigned spec, float* out);
int hb_set_state(hb_batch* b, unsigned spec, const float* in);
/* fp64 variants for bit-exact hand-over of oracle/CPU states */
int hb_get_state_f64(hb_batch* b, unsigned spec, double* out);
int hb_set_state_f64(hb_batch* b, unsigned spec, const double* in);

/* Env adapter outputs, the 27-DoF analogue of CPUEnv._get_obs/_get_reward (cpu_env.py:465-616):
 * obs[n_env][nobs] = [hinge qpos (nv-6), hinge qvel (nv-6), root angular velocity (3),
 * gravity direction in the torso frame (3)]; reward/terminated/truncated may be NULL; when requested they are
 * evaluated with the current hb_env_config without resetting anything (pure query). */
int hb_get_obs(hb_batch* b, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated);

/* Per-env status bits (HB_WARN_*), accumulated since the last hb_reset; replaces polling
 * mjData.warning (mujoco_mpc/mjpc/utilities.cc:787-799 CheckWarnings). */
int hb_get_status(hb_batch* b, int* status);
/* The same for a training loop that resets finished envs in place (hb_env_step with auto_reset): hb_get_status is cleared by an env's reset,
 * so a warning of an episode that ended before the next poll would be lost.  warnings[e] = the bits env e has raised since the previous
 * call of this function, in whatever episodes; the call clears what finished episodes left behind. */
int hb_env_warnings(hb_batch* b, int* warnings);
/* Per-env counters of the last step: ncon, nefc, solver iterations (mjData.ncon/nefc/
 * solver_niter, mjdata.h:196-201) — what testspeed.cc:97-98 accumulates. */
int hb_get_counts(hb_batch* b, int* ncon, int* nefc, int* niter);
/* Name of the step kernel the batch's last step / rollout / forward launch ran ("hb_step_duo_kernel", "hb_step_h27_q_kernel", ...; ""
 * before the first).  Which instantiation a launch takes depends on the model's solver and sizes and on the optional inputs / outputs
 * the call asked for: the parity tests assert that the kernel they checked is the kernel the benchmark times, and
 * bench.py names the kernel of its roofline object by this string.  The pointer stays valid for the life of the library. */
const char* hb_last_kernel(hb_batch* b);
/* How many times the batch has launched its step kernel(s) so far - the launches of one call's env segments (hb_batch_pipeline) count as
 * one, a staged step's kernels as one, hb_step_dev calls folded into one launch as one: bench.py divides its timed region by the
 * difference to get the launch duration its roofline object quotes.  Step calls still held back are launched first. */
long long hb_batch_step_launches(hb_batch* b);
/* "<device name> (<gfx arch>, <n> CUs) #<index>" of the GPU the batch lives on: what every rank of a multi-GPU bench.py run reports about
 * itself (the reference's testspeed.cc prints its thread count: sample/testspeed.cc:203-210). */
int hb_batch_device_name(const hb_batch* b, char* out, int cap);
/* Run-time choices between kernels and schedules that give the SAME results (every one of them is held bit-identical to its alternative
 * by a test): for the tests that compare them and for measurements.  Takes effect from the next launch on.  Nothing of the reference
 * corresponds: mj_step (mujoco.h:120) has one code path.
 *   HB_TUNE_DUO            two envs per wavefront for the lean launches of the 27-dof humanoid's PGS kernel (csrc/hb_step_duo.hip; DESIGN.md
 *                          3.8): 1 (default) where it pays - pipelined step calls of batches from 2.5 x the chip's wave slots on (5120 envs on
 *                          MI355X), unpipelined ones from 1.5 x (3072), launches of several steps from more than 1 x (2049); 0 never; 2 always
 *   HB_TUNE_LEAN           1 (default): launches without optional inputs / outputs take the lean instantiations of step_body; 0: the full ones
 *   HB_TUNE_SIZED          1 (default): models with the size signature of the 27-dof humanoid / the reference's robot take the kernels that
 *                          have those sizes as constants; 0: the generic ones
 *   HB_TUNE_STAGED         1 (default): models with mesh hulls / height fields step in stages (pose, narrowphase, step kernels: DESIGN.md
 *                          3.6); 0: everything inside one step kernel
 *   HB_TUNE_FASTPASS       1 (default): the staged step runs the one-group fast kernel first and the full one for what that defers; 0: full only
 *   HB_TUNE_NARROW_PRIM    1 (default): a model without mesh geoms takes the narrowphase kernel that has no hull climb in it
 *   HB_TUNE_SCHEDULE       1 (default): blocks are dispatched heavy-first (hb_order_kernel); 0: in env order
 *   HB_TUNE_REORDER_PERIOD the heavy-first order is re-sorted every n-th step call (default 4), and after every launch of eight steps or more
 *                          (the multi-step two-envs-per-wave kernel pairs its envs by that order: DESIGN.md 3.9)
 *   HB_TUNE_POLICY_LEAN    hb_rollout_policy: 1 (default) the LDS-free policy kernel beside the other segments' step kernels when
 *                          pipelined; 0 never; 2 always
 *   HB_TUNE_FOLD           hb_step_dev calls enqueued back to back run as ONE launch of up to this many steps (default and maximum 256; 1:
 *                          every call its own launch).  See hb_step_dev.
 * Environment variables the library reads (all others of earlier rounds are gone): HB_DEBUG (name failing HIP calls on stderr), HB_DUO
 * (HB_TUNE_DUO's value for new batches), HB_BOX_CULL=0 (model tables without the oriented-box cull of portal-search pairs: a test),
 * and in the diagnostic build (-DHB_STAMPS) HB_STOP_PHASE and HB_MPR_LIMIT (tools/gpu_narrow_limits.sh). */
enum { HB_TUNE_DUO = 0, HB_TUNE_LEAN, HB_TUNE_SIZED, HB_TUNE_STAGED, HB_TUNE_FASTPASS, HB_TUNE_NARROW_PRIM, HB_TUNE_SCHEDULE, HB_TUNE_REORDER_PERIOD,
       HB_TUNE_POLICY_LEAN, HB_TUNE_FOLD, HB_TUNE_COUNT };
int hb_batch_tune(hb_batch* b, int knob, int value);
/* Narrowphase work of the last step of each env, for models that collide through mesh hulls or height fields (the staged step:
 * DESIGN.md 3.6): nwork = work items (a candidate pair that passed the broadphase, or one prism of a height-field pair's sub-grid),
 * nsearch = those of them that needed a portal search (mjc_Convex / mjc_ConvexHField: libccd MPR), kcycles = shader clock cycles / 1024
 * the env's narrowphase wave took (saturating at 255; the cost the heavy-first dispatch of that launch sorts by).  Zeros for other
 * models.  Any pointer may be NULL. */
int hb_get_collision_counts(hb_batch* b, int* nwork, int* nsearch, int* kcycles);

/* Diagnostics of the last step for parity tests (mjData.qacc, efc_force, contact[]; mjdata.h:
 * 362,376,427): enable once, then read after a step.  efc_force is [n_env][nefc_max];
 * contact is [n_env][ncon_max][16] = dist, pos[3], frame[9], dim, geom1, geom2. */
int hb_diag_enable(hb_batch* b, int on);
int hb_get_qacc(hb_batch* b, float* qacc);
int hb_get_efc_force(hb_batch* b, float* efc_force);
int hb_get_contacts(hb_batch* b, float* contact);

/* ---- env adapter: the 27-DoF analogue of CPUEnv.step/reset (simulation/cpu_env.py:374-416,656-693) --------- */

#define HB_ENV_MAX_PAIRS 16
/* Reward / termination parameters of standupReward (simulation/reward_functions.py:247-374), made explicit.
 * hb_env_default_config fills the reference's weights and model-derived heights. */
typedef struct hb_env_config {
  float target_velocity[2];     /* control_input_velocity (cpu_env.py:593) */
  float target_z, min_z;        /* TARGET_Z_POS / MIN_Z_POS_FOR_REWARD (reward_functions.py:303-305) */
  float max_time;               /* MAX_SIM_TIME_STANDUP; <= 0 disables the time limit */
  float safe_torque;            /* MAX__SAFE_JOINT_TORQUE */
  float control_frequency;      /* CONTROL_FREQUENCY (simulation_parameters.py:51) */
  float action_scale;           /* previous/latest action are divided by this (pi/2 at cpu_env.py:601-602) */
  float w_hvel, w_upright, w_height, w_torque, w_ctrl_change, w_ctrl_reg, w_symmetry;
  float self_collision_penalty; /* SELF_COLLISION_PENALTY */
  float terminal_reward;        /* TERMINAL_REWARD (overrides the step reward at the time limit) */
  float upright_tol;            /* success needs max|g_local[0:2]| below this (0.7) */
  int n_equal, n_opposite;      /* actuator index pairs of symmetry_reward (reward_functions.py:98-113) */
  int equal_pairs[HB_ENV_MAX_PAIRS][2];
  int opposite_pairs[HB_ENV_MAX_PAIRS][2];
  int auto_reset;               /* 1: envs that terminate/truncate are reset inside hb_env_step (VecEnv semantics) */
  int reset_keyframe;           /* -1 = qpos0 */
  float reset_perturb;          /* 0..1 scale of the Halton joint/height perturbation (randomization_factor) */
  /* 0: standupReward semantics (terminated = time limit, with terminal_reward; truncated = success),
   * 1: controlInputReward semantics (reward_functions.py:116-245: terminated = fallen over or torso below
   *    min_z_grounded, with terminal_reward; truncated = time limit) */
  int reward_kind;
  float w_vvel;                 /* VERTICAL_VELOCITY_PENALTY_WEIGHT (0 for standupReward) */
  float min_z_grounded;         /* MIN_Z_BEFORE_GROUNDED */
  /* reset-until-collision-free (cpu_env.py:411-414: reset steps once and starts over while anything collides):
   * 0 off, 1 any contact (the reference's test), 2 self-collision only; hb_env_reset only, at most 8 draws */
  int reset_collision_mode;
  /* CPUEnv._randomize_starting_position also perturbs the root quaternion, +-QUAT_INITIAL_OFFSET_MAX (0.1) per component, scaled by
   * reset_perturb like the rest (cpu_env.py:300-328; the result is not renormalised there either: mj_kinematics does that) */
  float reset_quat_perturb;
  /* 0: the joint entries of the observation are in joint order; 1: in ACTUATOR order, the reference's JOINT_NAMES order
   * (simulation_parameters.py:84-103 = the <motor> order of assets/humanoid.xml:97-110): needs one actuator per scalar joint */
  int obs_actuator_order;
} hb_env_config;

int hb_env_default_config(const hb_model* m, hb_env_config* out);
/* The reference's own values for its own robot (simulation/assets/world.xml + humanoid.xml; reward_functions.py:289-372,
 * simulation_parameters.py:51-103): TARGET_Z_POS -0.375, MIN_Z_POS_FOR_REWARD -0.6, safe torque 1 N m, 500 Hz, the symmetry pairs
 * by joint name (equal: elbows; opposite: hip roll, hip pitch, knee, shoulder pitch, shoulder roll), observation in JOINT_NAMES
 * order, reset = keyframe "standup_reset" if the model has it (lying on the floor: INITIAL_QUAT_STANDUP, Z_INITIAL_POS_STANDUP) with
 * the joint / height / quaternion perturbations of CPUEnv.reset.  HB_EINVAL if the model lacks the reference's joint names. */
int hb_env_team_config(const hb_model* m, hb_env_config* out);
int hb_env_configure(hb_batch* b, const hb_env_config* cfg);

/* Sensor / actuation realism of CPUEnv (cpu_env.py:135-186,465-545,612-674; values of simulation_parameters.py:5-48),
 * all on the device and per env.  `factor` is the reference's randomization_factor and scales every magnitude.
 *  - action noise, then an integer-step action delay FIFO (filled with zeros at reset);
 *  - observation noise on joint angles / velocities, gyro, and the torso quaternion before the gravity vector is
 *    taken, then one delay FIFO per channel group (joints, gyro, gravity; fillers 0 / 0 / (0,0,-1));
 *  - delays are drawn per env and episode: round(U(min_delay, max_delay) * factor / control_timestep) steps;
 *  - pushes: a horizontal force of U(min,max)*factor N on a random body for U(duration) s every U(interval) s,
 *    through xfrc_applied (cpu_env.py:612-654).
 * Random numbers are a counter-based hash of (seed, global env index, episode, step, stream, element), so a run
 * is reproducible and independent of the batch split across GPUs.  frozen_noise = 1 reproduces a quirk of the
 * reference (its PRNG key is never split, so every step of an episode adds the same noise vector). */
typedef struct hb_env_randomization {
  float factor;
  unsigned int seed;
  float control_timestep;      /* seconds per hb_env_step call; <= 0: the model's timestep */
  float joint_angle_noise;     /* rad   (JOINT_ANGLE_NOISE_STDDEV, degrees in the reference) */
  float joint_velocity_noise;  /* rad/s (JOINT_VELOCITY_NOISE_STDDEV) */
  float gyro_noise;            /* rad/s (GYRO_NOISE_STDDEV) */
  float imu_noise;             /* added to each torso quaternion component (IMU_NOISE_STDDEV in rad) */
  float action_noise;          /* rad   (JOINT_ACTION_NOISE_STDDEV) */
  float min_delay, max_delay;  /* s */
  int frozen_noise;
  int push_enabled;
  float push_min_interval, push_max_interval, push_min_duration, push_max_duration, push_min_force, push_max_force;
} hb_env_randomization;

/* The reference's values (factor = 1, pushes on, fresh noise every step). */
int hb_env_default_randomization(const hb_model* m, hb_env_randomization* out);
/* Installs (cfg != NULL and cfg->factor > 0) or removes the above; takes effect at the next hb_env_reset
 * (call it before resetting).  Delays beyond 63 control steps are refused. */
int hb_env_randomize(hb_batch* b, const hb_env_randomization* cfg);
/* Domain randomisation of the model per env and episode (cpu_env.py:188-264, values of simulation_parameters.py:5-37),
 * drawn on the device at hb_env_reset and at every auto-reset, all scaled by `factor` as in the reference:
 *  - body masses += U(-max_mass_change, +max_mass_change) (floor 1e-5 kg) and one random body += U(0, max_external_mass);
 *    like the reference, derived constants (inverse weights, subtree masses) are NOT recomputed;
 *  - floor sliding friction *= (1 - factor) + U(friction_min_mult, friction_max_mult) * factor   (first plane geom);
 *  - per hinge/slide joint: armature += U(0, armature_max_change), stiffness += U(0, stiffness_max_change),
 *    limit margin += U(0, margin_max_change), each limit bound += U(-range_max_change, +range_max_change);
 *  - per actuator: force range bounds += U(-force_limit_max_change, +...); if kp_nominal > 0 the gain becomes
 *    kp_nominal + U(-kp_max_change, +kp_max_change) and, for actuators with an affine bias, biasprm[1] = -gain.
 * Random numbers come from the same counter-based generator as hb_env_randomization (seed, global env, episode). */
typedef struct hb_domain_randomization {
  float factor;
  unsigned int seed;
  float friction_min_mult, friction_max_mult;
  float max_mass_change, max_external_mass;
  float armature_max_change, stiffness_max_change, margin_max_change, range_max_change;
  float kp_nominal, kp_max_change, force_limit_max_change;
  float floor_bump_min, floor_bump_max;  /* height fields: per-env elevations, smooth noise scaled to [0, min + factor (max - min)]
                                            (CPUEnv._randomize_floor_heightmap, cpu_env.py:267-280); max = 0: the model's own data */
} hb_domain_randomization;
/* The reference's values (factor = 1, kp_nominal = 0: gains stay the model's — JOINT_P_GAIN = 2 is the team robot's). */
int hb_env_default_domain_randomization(const hb_model* m, hb_domain_randomization* out);
/* Installs (cfg != NULL and cfg->factor > 0) or removes per-env model parameters; call before hb_env_reset. */
int hb_env_domain_randomize(hb_batch* b, const hb_domain_randomization* cfg);
/* Current per-env parameters, for inspection: out[n_env][stride] with per env
 * mass[nbody] | armature[nv] | stiffness[nv] | limit margin[nlim] | limit bound[nlim] | gain[nu] | biasprm1[nu] |
 * forcerange[2 nu] | floor friction scale | hfield_data[nhfielddata]; returns stride (also when out == NULL), 0 when off. nlim = 2 per limited
 * joint / tendon in constraint order (lower, upper). */
int hb_env_get_domain_params(hb_batch* b, float* out);

/* CPUEnv.reset for every env: returns obs[n_env][nobs]. */
int hb_env_reset(hb_batch* b, float* obs);
/* CPUEnv.step: action[n_env][nu] -> ctrl (unscaled; the physics clamps to ctrlrange), n_substeps x mj_step,
 * reward, termination, observation; finished envs are reset when auto_reset is set and then report the
 * observation of their new episode. Host pointers.  The four outputs are one record on the device ([obs | reward | terminated |
 * truncated]); a caller whose four buffers lie back to back in that order (reward == obs + n_env * nobs, terminated ==
 * (uint8_t*)(reward + n_env), truncated == terminated + n_env; page-locked for a DMA transfer: hb_host_alloc) gets them in ONE
 * device-to-host transfer, anyone else in four. */
int hb_env_step(hb_batch* b, const float* action, int n_substeps, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated);
/* The observations of the states episodes ENDED in.  hb_env_step resets a finished env in place and reports the first observation of its
 * new episode; stable-baselines3's vectorised envs (the reference trains through DummyVecEnv: rl/train.py:134-136) keep the last
 * observation of the old one as infos[i]["terminal_observation"], and its off-policy learners bootstrap from it when the episode was
 * truncated - which in the reference's standupReward is the SUCCESS case (reward_functions.py:371-372, cpu_env.py:686-693).  The first call
 * (terminal_obs may be NULL) switches the recording on; from the next hb_env_step on, row e of the [n_env][nobs] table is rewritten
 * whenever env e finishes an episode (same sensor model, noise and delays as that episode's other observations) and keeps its value
 * otherwise.  With a non-NULL pointer the table is copied to the host (after the steps enqueued so far). */
int hb_env_terminal_obs(hb_batch* b, float* terminal_obs);
/* The same, enqueued only: VecEnv.step_async of the reference's training loop (stable-baselines3 VecEnv: step_async() then
 * step_wait(); rl/train.py:134-136 builds a DummyVecEnv whose step() is that pair).  The action buffer must stay untouched and the
 * output buffers unread until hb_batch_sync(b) returns (= step_wait); page-locked buffers (hb_host_alloc) make the copies truly
 * asynchronous.  The host is free in between - e.g. to run the learner's update. */
int hb_env_step_async(hb_batch* b, const float* action, int n_substeps, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated);
/* Same with device pointers, asynchronous on the batch's stream (policy on the same GPU). */
int hb_env_step_dev(hb_batch* b, const float* action_dev, int n_substeps, float* obs_dev, float* reward_dev, uint8_t* terminated_dev, uint8_t* truncated_dev);

/* ---- policy in the loop (BASELINE.json configs[3]: MLP policy inference fused into the rollout loop) ---------- */

/* Installs an MLP policy obs[nobs] -> sizes[1] -> ... -> sizes[n_layers] == nu with tanh after every layer
 * (sizes and activation of simulation/hyperparam_config.py:21-27, rl/train.py:163-167).  weights[l] is row-major
 * [sizes[l]][sizes[l+1]] float32 (the transpose of torch.nn.Linear.weight), biases[l] is [sizes[l+1]].  Host pointers;
 * copied to the device.  n_layers <= 4, every size <= 512. */
int hb_policy_set_mlp(hb_batch* b, int n_layers, const int* sizes, const float* const* weights, const float* const* biases);
/* One policy evaluation on the current states: ctrl_out[n_env][nu] (host, nullable) and the batch's control buffer. */
int hb_policy_eval(hb_batch* b, float* ctrl_out);
/* T closed-loop steps on the device: observation -> MLP (f32 MFMA GEMMs) -> mj_step, no host round trip.
 * qpos_out_dev (nullable, device) receives [T][n_env][nq]. */
int hb_rollout_policy(hb_batch* b, int T, float* qpos_out_dev);

/* ---- host-side helpers so a C/C++/ctypes caller needs no HIP headers ---------------------------- */

/* Device buffer on the batch's GPU (hipMalloc / hipFree). */
void* hb_dev_alloc(hb_batch* b, uint64_t bytes);
void hb_dev_free(hb_batch* b, void* p);
/* Page-locked host memory (hipHostMalloc / hipHostFree).  Host pointers handed to hb_step, hb_env_step, hb_get_obs ...
 * may be any memory; when they are page-locked the copies are DMA transfers that overlap instead of staged ones
 * (VecEnv keeps its action and observation arrays in such buffers). */
void* hb_host_alloc(uint64_t bytes);
void hb_host_free(void* p);
int hb_memcpy_h2d(hb_batch* b, void* dst_dev, const void* src, uint64_t bytes);
int hb_memcpy_d2h(hb_batch* b, void* dst, const void* src_dev, uint64_t bytes);
/* Fills out_dev[T][n_env][nu] with the benchmark's deterministic controls, the generator of
 * simulation/mujoco/sample/testspeed.cc:64-80 (CtrlNoise) with ctrlnoise = 1:
 * ctrl[t,e,i] = 2*Halton(1 + t0 + t + 1000*(env_offset+e), i+2) - 1. */
int hb_halton_ctrl_dev(hb_batch* b, int T, int t0, int env_offset, float* out_dev);
/* HIP-event stopwatch on the batch's stream (hipEventRecord on that stream; hipEventElapsedTime). */
int hb_timer_start(hb_batch* b);
int hb_timer_stop(hb_batch* b, float* elapsed_ms);
/* Per-kernel timing of the step kernel alone: when enabled, every 8th step launch is bracketed by its own pair
 * of HIP events on the launch stream (up to 256 samples); hb_step_timing_read synchronises and returns their mean. */
int hb_step_timing(hb_batch* b, int enable);
int hb_step_timing_read(hb_batch* b, float* mean_us, int* samples);
/* Diagnostic builds only (libhb_stamps.so, -DHB_STAMPS): per-env s_memtime stamps at the 16 phase
 * boundaries of the last step; the first call arms the buffer.  HB_EUNSUPPORTED in the product build. */
int hb_get_stamps(hb_batch* b, unsigned long long* out);

const char* hb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HB_H_ */
