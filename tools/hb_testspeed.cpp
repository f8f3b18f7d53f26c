// hb_testspeed — host C++ rollout driver over the C-ABI, in the shape of the reference's
// simulation/mujoco/sample/testspeed.cc (load model, make data, Halton control noise, step loop,
// print steps/s, contacts/step, constraints/step) but for n_env environments on one GPU.
// usage: hb_testspeed model.{xml,hbm} [nstep=1000] [n_env=4096] [device=0] [PGS|Newton]
// The per-stage table of testspeed.cc:235-288 (mjData.timer) is printed when the library is the diagnostic build
// (build/hb_testspeed_stamps, linked against build/libhb_stamps.so: s_memtime stamps at the stage boundaries of the step
// kernel); the product library carries no stamps (they cost about a tenth of a wave's cycles) and says so.
#include "hb.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s model.{xml,hbm} [nstep] [n_env] [device] [PGS|Newton]\n", argv[0]); return 2; }
  int nstep = argc > 2 ? atoi(argv[2]) : 1000, n_env = argc > 3 ? atoi(argv[3]) : 4096, device = argc > 4 ? atoi(argv[4]) : 0;
  char err[1024] = "";
  hb_model* m = hb_model_load(argv[1], err, sizeof err);
  if (!m) { fprintf(stderr, "could not load model: %s\n", err); return 1; }
  hb_sizes sz;
  hb_model_sizes(m, &sz);
  if (argc > 5) {  // solver override (the committed benchmark model carries PGS / 50; the reference's XML means Newton / 100)
    hb_options o;
    hb_options_get(m, &o);
    if (!strcmp(argv[5], "Newton")) { o.solver = 2; o.iterations = 100; }
    else if (!strcmp(argv[5], "PGS")) { o.solver = 0; o.iterations = 50; }
    else { fprintf(stderr, "unknown solver %s\n", argv[5]); return 2; }
    if (hb_options_set(m, &o) != HB_OK) { fprintf(stderr, "could not set options\n"); return 1; }
  }
  hb_batch* b = hb_batch_create(m, n_env, device, err, sizeof err);
  if (!b) { fprintf(stderr, "could not create batch: %s\n", err); return 1; }
  hb_reset(b, nullptr, -1, /*perturb=*/1, /*env_offset=*/0);
  // controls for the whole run live in HBM, generated there (testspeed.cc:64-80)
  float* ctrl = (float*)hb_dev_alloc(b, (uint64_t)nstep * n_env * sz.nu * sizeof(float));
  if (!ctrl) { fprintf(stderr, "out of device memory\n"); return 1; }
  hb_halton_ctrl_dev(b, nstep, 0, 0, ctrl);
  hb_batch_pipeline(b, 1);  // two env segments on two streams: one step's tail overlaps the next
  hb_step_dev(b, ctrl, 1);  // warm-up launch
  hb_reset(b, nullptr, -1, 1, 0);
  hb_batch_sync(b);
  long long contacts = 0, constraints = 0;
  std::vector<int> ncon(n_env), nefc(n_env);
  auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < nstep; t++) hb_step_dev(b, ctrl + (size_t)t * n_env * sz.nu, 1);
  hb_batch_sync(b);
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  hb_get_counts(b, ncon.data(), nefc.data(), nullptr);
  for (int e = 0; e < n_env; e++) { contacts += ncon[e]; constraints += nefc[e]; }
  std::vector<int> status(n_env);
  hb_get_status(b, status.data());
  int flagged = 0;
  for (int s : status) flagged += s != 0;
  hb_options opt;
  hb_options_get(m, &opt);
  printf("\nSimulation time      : %.3f s\n", sec);
  printf("Environments         : %d on device %d\n", n_env, device);
  printf("Steps per second     : %.0f (env-steps/s)\n", (double)nstep * n_env / sec);
  printf("Realtime factor      : %.1f x per env\n", nstep * opt.timestep / sec);
  printf("Time per batch step  : %.1f us\n", 1e6 * sec / nstep);
  printf("Contacts / env (last): %.3f\n", (double)contacts / n_env);
  printf("Constraints / env    : %.3f\n", (double)constraints / n_env);
  printf("Degrees of freedom   : %d\n", sz.nv);
  printf("Solver               : %s, at most %d iterations\n", opt.solver == 2 ? "Newton" : "PGS", opt.iterations);
  printf("Envs with warnings   : %d\n", flagged);
  {  // internal profiler (mjTIMER_* taxonomy, mjdata.h:68-91): mean cycles per env-step of one more step, per stage
    std::vector<unsigned long long> st((size_t)n_env * 16);
    if (hb_get_stamps(b, st.data()) == HB_OK) {  // first call arms the buffer
      hb_step_dev(b, ctrl + (size_t)(nstep - 1) * n_env * sz.nu, 1);
      hb_batch_sync(b);
      hb_get_stamps(b, st.data());
      double ph[15] = {0};
      for (int e = 0; e < n_env; e++) for (int i = 0; i < 15; i++) ph[i] += (double)(st[(size_t)e * 16 + i + 1] - st[(size_t)e * 16 + i]) / n_env;
      double tot = 0;
      for (double v : ph) tot += v;
      struct { const char* name; unsigned mask; } rows[] = {  // bit i: stage i of the kernel's stamp list (tools/gpu_phase_profile.py)
          {"step", 0x7fff}, {"position", 0x0fbe}, {"  kinematics", 0x0006}, {"  inertia (crb, qM)", 0x0038}, {"  collision", 0x0080}, {"  make", 0x0300},
          {"  project (J W, AR)", 0x0c00}, {"velocity+actuation", 0x0040}, {"constraint (solver)", 0x3000}, {"advance (Euler)", 0x4000}, {"load state, ctrl, checks", 0x0001}};
      printf("\nInternal profiler, shader cycles per env-step (one wave per env; stage stamps cost ~10 %% themselves)\n");
      for (auto& r : rows) {
        double v = 0;
        for (int i = 0; i < 15; i++) if ((r.mask >> i) & 1u) v += ph[i];
        printf(" %-26s : %9.0f  (%6.2f %%)\n", r.name, v, 100 * v / tot);
      }
      printf(" (the sweep up the tree of stage 'inertia' also carries mj_rne's backward pass and mj_comVel: one merged pass)\n");
    } else printf("Internal profiler    : not in this build (use build/hb_testspeed_stamps: stage stamps exist in the diagnostic library only)\n");
  }
  hb_dev_free(b, ctrl);
  hb_batch_free(b);
  hb_model_free(m);
  return 0;
}
