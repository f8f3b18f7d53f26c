// hb_narrow.hip - the staged step of the general variants: hb_pose_kernel (poses, broadphase, work items) and the narrowphase kernels
// (one work item per lane).  See hb_step.hip for the step kernels that gather their results.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifdef HB_STAMPS
// diagnostic build only (tools/gpu_narrow_limits.sh): HB_MPR_LIMIT="a,b" in the environment cuts every portal search off after a support
// calls and every hull climb after b rounds - WRONG contacts, but the launch durations say where the narrowphase's time goes
#define HB_NARROW_DIAG 1
namespace hb { __device__ int g_mpr_limit[2] = {1 << 30, 1 << 30}; }
#endif
#include "hb_kcommon.hpp"
#include "hb_collide.hpp"
#include "hb_launch.hpp"

namespace hb {

// ---- staged step of the general variants: poses + work lists, then the narrowphase, each in a kernel of its own ------------------
// hb_pose_kernel: one wave per env.  The state checks and mj_kinematics of step_body, statement for statement (the step kernel
// repeats them: a pose costs less to recompute than to hand over), the geoms' world poses, broadphase and work items.
__global__ __launch_bounds__(kGroup, 4) void hb_pose_kernel(const DevModel* Mp, const BatchPtrs P) {
  DevModelRef M = *(const DevModel HB_CONST*)(uintptr_t)Mp;
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= P.nblk) return;
  const int env = P.blk0 + (int)blockIdx.x;
  if (blockIdx.x == 0 && lane == 0 && P.stage.defer_count) P.stage.defer_count[P.blk0] = 0;  // (the list the step's fast pass fills for its second pass)
  if (P.env_mask && !P.env_mask[env]) { if (lane == 0) { P.stage.nwork[env] = 0; P.stage.nsearch[2 * env] = 0; P.stage.nsearch[2 * env + 1] = 0; } return; }
  const int nq = M.nq, nv = M.nv, nb = M.nbody, ng = M.ngeom;
  float* s_qpos = lds;
  float* s_xpq = s_qpos + ((nq + 3) & ~3);
  float* s_gpos = s_xpq + kXpqStride * nb;
  float* s_gaxis = s_gpos + ((3 * ng + 3) & ~3);
  float* s_gquat = s_gaxis + ((3 * ng + 3) & ~3);
  int* s_scratch = reinterpret_cast<int*>(s_gquat + 4 * ng);
  const float* gstate = P.state + (size_t)env * M.nstate;
  // mj_checkPos / mj_checkVel: a bad state is reset before the step, and the step's poses are those of qpos0
  bool badp = false, badv = false;
  for (int i = lane; i < nq; i += kGroup) { const float v = gstate[1 + i]; s_qpos[i] = v; badp |= !(fabsf(v) <= HB_MAXVAL); }
  for (int i = lane; i < nv; i += kGroup) { const float v = gstate[1 + nq + i]; badv |= !(fabsf(v) <= HB_MAXVAL); }
  if (__any(badp) || __any(badv)) for (int i = lane; i < nq; i += kGroup) s_qpos[i] = M.qpos0[i];
  const bool bl = lane + 1 < nb;
  float4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, bp = q0, bq = q0;
  float4 JA[3], JB[3], JC[3];
#pragma unroll
  for (int jj = 0; jj < 3; jj++) { JA[jj] = q0; JB[jj] = q0; JC[jj] = q0; }
  if (bl) {
    const float4 HB_CONST* R = M.brec + (size_t)(lane + 1) * kBrecQuads;
    q0 = R[0]; q1 = R[1]; bp = R[2]; bq = R[3];
#pragma unroll
    for (int jj = 0; jj < 3; jj++) { JA[jj] = R[9 + 3 * jj]; JB[jj] = R[10 + 3 * jj]; JC[jj] = R[11 + 3 * jj]; }
  }
  int pf_gbody = 0;
  V3 pf_gpos = {0.f, 0.f, 0.f};
  Q4 pf_gquat = {1.f, 0.f, 0.f, 0.f};
  if (lane < ng) { pf_gbody = M.geom_bodyid[lane]; pf_gpos = ld3(M.geom_pos + 3 * lane); pf_gquat = ldq(M.geom_quat + 4 * lane); }
  if (lane == 0) { st3(s_xpq, {0.f, 0.f, 0.f}); stq(s_xpq + 4, {1.f, 0.f, 0.f, 0.f}); }
  gsync();
  const int myb = __float_as_int(q0.x), myp = __float_as_int(q0.y), myjn = __float_as_int(q0.z);
  const int myanc2 = (__float_as_int(q1.x) >> 8) & 255, myanc4 = (__float_as_int(q1.x) >> 16) & 255, myanc8 = (__float_as_int(q1.x) >> 24) & 255;
  const bool isfree = bl && myjn == 1 && __float_as_int(JA[0].x) == 0;
  V3 posl = {bp.x, bp.y, bp.z};
  Q4 quatl = {bq.x, bq.y, bq.z, bq.w};
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
        const V3 axl = qrot(quatl, laxis);
        const V3 ancl = qrot(quatl, lpos) + posl;
        const float dq = s_qpos[qa] - JA[jj].w;
        if (__float_as_int(JA[jj].x) == 2) posl = posl + axl * dq;
        else {
          quatl = qmul(quatl, axisangle(laxis, dq));
          posl = ancl - qrot(quatl, lpos);
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
  for (int r = 0, span = 1; span < M.nlevel - 1 || r == 0; r++, span <<= 1) {
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
  // geoms: world position, z axis and orientation (the step kernel rotates the offset with the body's matrix: the same q2mat here)
  if (lane < ng) {
    const int g = lane, b = pf_gbody;
    float mat[9];
    q2mat(mat, ldq(s_xpq + kXpqStride * b + 4));
    const V3 gp = ld3(s_xpq + kXpqStride * b) + mrot(mat, pf_gpos);
    const Q4 q = qmul(ldq(s_xpq + kXpqStride * b + 4), pf_gquat);
    const V3 ga = {2.f * (q.x * q.z + q.w * q.y), 2.f * (q.y * q.z - q.w * q.x), q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z};
    st3(s_gpos + 3 * g, gp); st3(s_gaxis + 3 * g, ga); stq(s_gquat + 4 * g, q);
    float* o = P.stage.geom + (size_t)env * ng * 10;  // per env: positions[3 ng] | z axes[3 ng] | quaternions[4 ng]
    st3(o + 3 * g, gp); st3(o + 3 * ng + 3 * g, ga); stq(o + 6 * ng + 4 * g, q);
  }
  gsync();
  int status = 0;
  const int nwork = build_work_list(M, lane, s_gpos, s_gaxis, s_gquat, s_scratch, status);
  const int* s_list = s_scratch;
  const int* s_pinfo = s_scratch + kListMax;
  const int* s_work = s_pinfo + 4 * kListMax;
  // Every item but its portal search (the analytic pairs completely; a prism's height test): results of the items that are done
  // go straight to the step kernel's input, the others are listed for hb_narrow_kernel, which packs them 64 to a wave whatever env
  // they belong to (an env has about nine: one wave per env would run mostly empty).
  const DomainLayout DL = domain_layout(M.nbody, M.nv, M.nlimcand, M.nu, M.nhfielddata);
  const float* hdata = P.dr ? P.dr + (size_t)env * P.dr_stride + DL.o_hfield : (const float*)M.hfield_data;
  float4* R = P.stage.result + (size_t)env * kWorkMax * 4;
  int4* items = P.stage.item + (size_t)env * kWorkMax;
  int nsearch1 = 0, nsearch2 = 0;
  for (int w0 = 0; w0 < nwork; w0 += kGroup) {
    const int w = w0 + lane;
    const bool have = w < nwork;
    const int item = have ? s_work[w] : 0;
    const int idx = item >> 16, sub = item & 0xffff;
    const int p = have ? s_list[idx] : 0;
    const int rmin = s_pinfo[4 * idx], cmin = s_pinfo[4 * idx + 1], ncols = s_pinfo[4 * idx + 2];
    ConOut co0, co1;
    int n;
    V3 hint;
    const int kind = eval_work_item<1>(M, hdata, have, p, sub, rmin, cmin, ncols, __int_as_float(s_pinfo[4 * idx + 3]), s_gpos, s_gaxis, s_gquat, co0, co1, n, hint);
    if (have && !kind) {
      R[4 * w] = {co0.dist, co0.pos.x, co0.pos.y, co0.pos.z};
      R[4 * w + 1] = {co0.n.x, co0.n.y, co0.n.z, __int_as_float(n)};
      R[4 * w + 2] = {co1.dist, co1.pos.x, co1.pos.y, co1.pos.z};
      R[4 * w + 3] = {co1.n.x, co1.n.y, co1.n.z, __int_as_float(p)};
    }
    // (the env's own slots: prism searches from the front, pair searches from the back)
    const unsigned long long need1 = __ballot(have && kind == 1), need2 = __ballot(have && kind == 2);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int4 rec = {env, p | (w << 16), sub | (ncols << 16), rmin | (cmin << 16)};
    if (have && kind == 1) items[nsearch1 + __popcll(need1 & below)] = rec;
    if (have && kind == 2) items[kWorkMax - 1 - (nsearch2 + __popcll(need2 & below))] = rec;
    nsearch1 += __popcll(need1); nsearch2 += __popcll(need2);
  }
  if (lane == 0) {
    P.stage.nwork[env] = nwork;
    P.stage.nsearch[2 * env] = nsearch1; P.stage.nsearch[2 * env + 1] = nsearch2;
    int* c = P.counts + kCountStride * (size_t)env;
    c[5] = nwork; c[6] = nsearch1 + nsearch2;
    if (status) atomicOr(P.status + env, status);
  }
}

// hb_narrow_kernel: the portal searches, one wave per env (and per 64 of its searches): lane l runs the env's l-th search, prisms
// first, exactly as the fused step kernel would (eval_work_item).  The waves are mostly empty (an env has about nine searches), but
// there are as many of them as the chip holds at once; packing the searches of all envs densely into waves (a prefix sum over the
// per-env counts, 64 / 16 / 4 searches per wave, one kernel per kind of search) measured slower: a wave's time is set by its
// longest search and the divergence between its lanes, not by how many lanes it has.  Round 4 tried it again on top of the
// four-lane searches and the one-loop portal search (a flat list filled by one atomicAdd per env in hb_pose_kernel, sixteen consecutive
// searches per wave): the narrowphase launch stayed at 85 us - it lasts as long as its longest search, 170-190 k cycles - and the 4096
// atomics on one counter cost the pose kernel 23 us.
// A model with meshes (MESH = 1): FOUR lanes per search (hb_mpr.hpp: climb4 - the four share the rounds of a hull climb), sixteen searches per
// wave; without meshes a lane per search.  Where that came from: the diagnostic build's HB_MPR_LIMIT (tools/gpu_narrow_limits.sh,
// profiles/r04_narrow_limits.txt) showed the launch VALU-bound at 4.6 active lanes, 10 us per support call allowed, 8 of them the climb.
// PAIR = 1: TWO envs per wave (lanes 0..31: one env's searches, 32..63: the other's).  Half the waves, each a little longer: the unpipelined
// step of 4096 robots 267 -> 233 us with one lane per search (profiles/r04_narrow_pairs.txt), and with four lanes per search the launch is
// 84 us in this form (profiles/r04_team_counters.json: 11.7 active lanes per vector instruction).  The launch then is ONE round of waves and
// lasts as long as its slowest; for pipelined segments the two forms measure the same (169 us per step of 4096 robots), and
// launch_pose_narrow takes the pair form for a launch that covers its whole batch.
template <int MESH, int PAIR = 0>
__device__ __forceinline__ void narrow_body(const DevModel* Mp, const BatchPtrs& P) {
  DevModelRef M = *(const DevModel HB_CONST*)(uintptr_t)Mp;
  const int lane = threadIdx.x;
  // heavy first (BatchPtrs::order2: the envs of this launch sorted by the time their wave took in an earlier step): the launch ends
  // when its slowest wave does, and a slow wave that starts in the last round ends late
  constexpr int G = MESH ? 4 : 1;  // lanes per search
  constexpr int kPer = (PAIR ? 32 : kGroup) / G;  // searches of an env per chunk
  const int nslot = PAIR ? (P.nblk + 1) >> 1 : P.nblk;
  const int nwaves = (int)gridDim.x / nslot;  // waves per slot: wave c takes chunks c, c + nwaves, ... of its env(s)
  const int pr = (int)blockIdx.x % nslot;
  const int half = PAIR ? lane >> 5 : 0, l = PAIR ? lane & 31 : lane;
  const int slot = P.blk0 + (PAIR ? 2 * pr + half : pr);
  const bool live = !PAIR || 2 * pr + half < P.nblk;
  const int env = live ? (P.order2 ? P.order2[slot] : slot) : 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  const int n1 = live ? P.stage.nsearch[2 * env] : 0, n2 = live ? P.stage.nsearch[2 * env + 1] : 0;
  const bool first = (int)blockIdx.x / nslot == 0;
  if (first && l == 0 && live) P.counts[kCountStride * (size_t)env + 7] = 0;
  const int ng = M.ngeom;
  const DomainLayout DL = domain_layout(M.nbody, M.nv, M.nlimcand, M.nu, M.nhfielddata);
  const float* g = P.stage.geom + (size_t)env * ng * 10;
  const float* hdata = P.dr ? P.dr + (size_t)env * P.dr_stride + DL.o_hfield : (const float*)M.hfield_data;
  auto run_chunk = [&](int chunk) {
    const int j = chunk * kPer + l / G;
    const bool have = j < n1 + n2;
    int4 it = {env, 0, 1 << 16, 0};
    if (have) it = P.stage.item[(size_t)env * kWorkMax + (j < n1 ? j : kWorkMax - 1 - (j - n1))];
    const int p = it.y & 0xffff, w = it.y >> 16;
    ConOut co0, co1;
    int n;
    V3 hint;
    eval_work_item<2, MESH, G>(M, hdata, have, p, it.z & 0xffff, it.w & 0xffff, it.w >> 16, it.z >> 16, 0.f, g, g + 3 * ng, g + 6 * ng, co0, co1, n, hint);
    if (have && (l & (G - 1)) == 0) {
      float4* R = P.stage.result + ((size_t)env * kWorkMax + w) * 4;
      R[0] = {co0.dist, co0.pos.x, co0.pos.y, co0.pos.z};
      R[1] = {co0.n.x, co0.n.y, co0.n.z, __int_as_float(n)};
      R[2] = {co1.dist, co1.pos.x, co1.pos.y, co1.pos.z};
      R[3] = {co1.n.x, co1.n.y, co1.n.z, __int_as_float(p)};
    }
  };
  if constexpr (G == 1) {  // (a wave per chunk: launch_pose_narrow; a loop here costs the kernel without the hull climb 50 more spilled registers)
    if (__any((int)blockIdx.x / nslot * kPer < n1 + n2)) run_chunk((int)blockIdx.x / nslot);
  } else {
    for (int chunk = (int)blockIdx.x / nslot; __any(chunk * kPer < n1 + n2); chunk += nwaves) run_chunk(chunk);
  }
  // (the cost the heavy-first order of the next launches sorts by: the time of the env's - first - wave)
  if (first && l == 0 && live) P.counts[kCountStride * (size_t)env + 7] = (int)min(255ull, (__builtin_amdgcn_s_memtime() - t_begin) >> 10);
}
__global__ __launch_bounds__(kGroup, 2) void hb_narrow_kernel(const DevModel* Mp, const BatchPtrs P) { narrow_body<1>(Mp, P); }
// a model without mesh geoms (configs[4]: capsules and spheres over the height field's prisms): no hull climb in the kernel
__global__ __launch_bounds__(kGroup, 2) void hb_narrow_prim_kernel(const DevModel* Mp, const BatchPtrs P) { narrow_body<0>(Mp, P); }
// two envs per wave (narrow_body's PAIR): the unpipelined launches
__global__ __launch_bounds__(kGroup, 2) void hb_narrow2_kernel(const DevModel* Mp, const BatchPtrs P) { narrow_body<1, 1>(Mp, P); }
__global__ __launch_bounds__(kGroup, 2) void hb_narrow2_prim_kernel(const DevModel* Mp, const BatchPtrs P) { narrow_body<0, 1>(Mp, P); }

hipError_t launch_pose_narrow(const DevModel* M_dev, const BatchPtrs& Q, hipStream_t stream) {
  (void)hipGetLastError();
#ifdef HB_NARROW_DIAG
  static const bool limits_set = [] {
    if (const char* e = getenv("HB_MPR_LIMIT")) {
      int v[2] = {1 << 30, 1 << 30};
      sscanf(e, "%d,%d", &v[0], &v[1]);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mpr_limit), v, sizeof v);
    }
    return true;
  }();
  (void)limits_set;
#endif
  hipLaunchKernelGGL(hb_pose_kernel, dim3(Q.nblk), dim3(kGroup), (size_t)Q.stage.pose_lds, stream, M_dev, Q);
  const bool pairs = Q.nblk == Q.n_env && Q.n_env >= 512;  // a launch that covers its whole batch: nothing overlaps its tail anyway
  if (pairs) {
    // (waves per pair of envs: a wave without searches left ends at once.  Meshes: four lanes per search, two waves hold sixteen searches per env -
    // the reference's robot has at most eleven - and loop for more; without meshes a lane per search, a wave per 32 searches of each env)
    const int slots = (Q.nblk + 1) / 2;
    if (Q.stage.no_mesh) hipLaunchKernelGGL(hb_narrow2_prim_kernel, dim3(slots * (kWorkMax / 32)), dim3(kGroup), 0, stream, M_dev, Q);
    else hipLaunchKernelGGL(hb_narrow2_kernel, dim3(slots * 2), dim3(kGroup), 0, stream, M_dev, Q);
  } else if (Q.stage.no_mesh) hipLaunchKernelGGL(hb_narrow_prim_kernel, dim3(Q.nblk * (kWorkMax / kGroup)), dim3(kGroup), 0, stream, M_dev, Q);
  else hipLaunchKernelGGL(hb_narrow_kernel, dim3(Q.nblk * 2), dim3(kGroup), 0, stream, M_dev, Q);
  return hipGetLastError();
}

}  // namespace hb
