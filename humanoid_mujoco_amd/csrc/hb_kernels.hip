// hb_kernels.hip — the fused physics-step kernel for gfx950 (MI355X).
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
#include <hip/hip_runtime.h>
#include "hb_device.hpp"

namespace hb {

// Diagnostic build only (-DHB_STAMPS): per-phase cycle stamps of the last step, written to
// BatchPtrs::diag_contact's tail is NOT used; stamps go to their own buffer P.stamps.
#ifdef HB_STAMPS
// (P.stop_phase = k > 0: the wave leaves at stamp k - 1 without writing anything - tools/gpu_phase_instructions.py counts a launch's instructions
// up to every stamp with the PMC counters and differences them)
#define HB_STAMP(i) do { if (lane == 0 && P.stamps) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamps_[i] = t_; } if (P.stop_phase == (i) + 1) return; } while (0)
#else
#define HB_STAMP(i) do {} while (0)
#endif
#if defined(HB_STAMPS) && defined(HB_PROBE_NEWTON)
// diagnostic: cycles per section of the Newton solve, accumulated over the iterations of one step (slots 0..7 of the stamps)
#define HB_NP(i) do { if (P.stamps) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); np_acc[i] += t_ - np_t; np_t = t_; } } while (0)
#else
#define HB_NP(i) do {} while (0)
#endif
#define HB_MINVAL 1e-15f
#define HB_MAXVAL 1e10f
#define HB_MINIMP 0.0001f
#define HB_MAXIMP 0.9999f

// wave-level ordering point for LDS traffic between lanes of one wavefront.  A wavefront's DS
// instructions execute in issue order, so no s_barrier is needed; the fences stop the compiler
// from moving LDS accesses across this point.
__device__ __forceinline__ void gsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

__device__ __forceinline__ float rdlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
// v_writelane_b32 (clang has no builtin for it; bind the LLVM intrinsic by name)
extern "C" __device__ int hb_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 64 lanes, identical (and scalar) in every lane: four DPP steps inside each row of 16,
// then the four row sums through scalar registers; no LDS traffic.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
}
// value known to be identical in every lane -> tell the compiler (scalar register, scalar branches)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniformf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ V3 ld3(const float HB_CONST* p) { return {p[0], p[1], p[2]}; }
#endif
__device__ __forceinline__ void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 normalized(V3 v, float* n_out = nullptr) {
  float n = sqrtf(dot(v, v));
  if (n_out) *n_out = n;
  if (n < HB_MINVAL) return {1.f, 0.f, 0.f};
  float inv = 1.f / n;
  return v * inv;
}

struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 ldq(const float* p) { return {p[0], p[1], p[2], p[3]}; }
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ Q4 ldq(const float HB_CONST* p) { return {p[0], p[1], p[2], p[3]}; }
#endif
__device__ __forceinline__ void stq(float* p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qconj(Q4 q) { return {q.w, -q.x, -q.y, -q.z}; }
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
  const float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  if (n2 < HB_MINVAL * HB_MINVAL) return {1.f, 0.f, 0.f, 0.f};
  const float inv = rsqrtf(n2);
  return {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
__device__ __forceinline__ void q2mat(float* m, Q4 q) {
  float q00 = q.w * q.w, q11 = q.x * q.x, q22 = q.y * q.y, q33 = q.z * q.z;
  float q01 = q.w * q.x, q02 = q.w * q.y, q03 = q.w * q.z, q12 = q.x * q.y, q13 = q.x * q.z, q23 = q.y * q.z;
  m[0] = q00 + q11 - q22 - q33; m[1] = 2.f * (q12 - q03); m[2] = 2.f * (q13 + q02);
  m[3] = 2.f * (q12 + q03); m[4] = q00 - q11 + q22 - q33; m[5] = 2.f * (q23 - q01);
  m[6] = 2.f * (q13 - q02); m[7] = 2.f * (q23 + q01); m[8] = q00 - q11 - q22 + q33;
}
__device__ __forceinline__ V3 mrot(const float* m, V3 v) {
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
// rotate v by the unit quaternion q: v + 2 w (u x v) + 2 u x (u x v), u = (x, y, z) - 21 flops instead of the
// 40 of building the rotation matrix first (the same rotation; rounding differs in the last bits)
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
  const V3 u = {q.x, q.y, q.z};
  V3 t = cross(u, v);
  t = {t.x + t.x, t.y + t.y, t.z + t.z};
  const V3 c = cross(u, t);
  return {v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z};
}
__device__ __forceinline__ Q4 axisangle(V3 axis, float ang) {
  float s, c;
  sincosf(0.5f * ang, &s, &c);
  return {c, axis.x * s, axis.y * s, axis.z * s};
}

// spatial algebra on 6-vectors (rotation, translation); cinert layout as mjData.cinert (mjdata.h:269)
__device__ __forceinline__ void mul_inert_vec(float* r, const float* i, const float* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
__device__ __forceinline__ void cross_motion(float* r, const float* vel, const float* v) {
  r[0] = -vel[2] * v[1] + vel[1] * v[2];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = -vel[1] * v[0] + vel[0] * v[1];
  r[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
__device__ __forceinline__ void cross_force(float* r, const float* vel, const float* f) {
  r[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  r[3] = -vel[2] * f[4] + vel[1] * f[5];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = -vel[1] * f[3] + vel[0] * f[4];
}

// radical inverse, mju_Halton (mujoco.h:1231); used by simulation/mujoco/sample/testspeed.cc:76
__device__ __forceinline__ float halton(int index, int base) {
  float f = 1.f / (float)base, fb = f, hn = 0.f;
  while (index > 0) {
    int n1 = index / base, r = index - n1 * base;
    hn += f * (float)r;
    f *= fb;
    index = n1;
  }
  return hn;
}

// ------------------------------------------------------------------------------------------
// narrowphase helpers (engine_collision_primitive restatement, see oracle)
struct ConOut { float dist; V3 pos; V3 n; };

__device__ __forceinline__ bool plane_sphere(ConOut& c, float margin, V3 ppos, V3 normal, V3 spos, float radius) {
  float cdist = dot(spos - ppos, normal);
  if (cdist > margin + radius) return false;
  c.dist = cdist - radius;
  c.pos = spos + normal * (-c.dist * 0.5f - radius);
  c.n = normal;
  return true;
}
__device__ __forceinline__ bool sphere_sphere(ConOut& c, float margin, V3 p1, float r1, V3 p2, float r2) {
  V3 dif = p2 - p1;
  float cdist = sqrtf(dot(dif, dif));
  if (cdist > margin + r1 + r2) return false;
  c.dist = cdist - r1 - r2;
  V3 n = cdist < HB_MINVAL ? V3{1.f, 0.f, 0.f} : dif * (1.f / cdist);
  c.pos = p1 + n * (r1 + c.dist * 0.5f);
  c.n = n;
  return true;
}
__device__ __forceinline__ int capsule_capsule(ConOut& c0, ConOut& c1, float margin, V3 pos1, V3 axis1, float r1, float len1, V3 pos2, V3 axis2, float r2, float len2) {
  V3 dif = pos1 - pos2;
  float ma = dot(axis1, axis1), mb = -dot(axis1, axis2), mc = dot(axis2, axis2);
  float u = -dot(axis1, dif), v = dot(axis2, dif);
  float det = ma * mc - mb * mb;
  if (fabsf(det) >= HB_MINVAL) {
    float x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > len1) { x1 = len1; x2 = (v - mb * len1) / mc; }
    else if (x1 < -len1) { x1 = -len1; x2 = (v + mb * len1) / mc; }
    if (x2 > len2) { x2 = len2; x1 = clampf((u - mb * len2) / ma, -len1, len1); }
    else if (x2 < -len2) { x2 = -len2; x1 = clampf((u + mb * len2) / ma, -len1, len1); }
    return sphere_sphere(c0, margin, pos1 + axis1 * x1, r1, pos2 + axis2 * x2, r2) ? 1 : 0;
  }
  // parallel axes: up to two contacts from the segment ends (first two hits in this order)
  ConOut t0, t1, t2, t3;
  float x2 = clampf((v - mb * len1) / mc, -len2, len2);
  const bool h0 = sphere_sphere(t0, margin, pos1 + axis1 * len1, r1, pos2 + axis2 * x2, r2);
  x2 = clampf((v + mb * len1) / mc, -len2, len2);
  const bool h1 = sphere_sphere(t1, margin, pos1 - axis1 * len1, r1, pos2 + axis2 * x2, r2);
  float x1 = clampf((u - mb * len2) / ma, -len1, len1);
  const bool h2 = sphere_sphere(t2, margin, pos1 + axis1 * x1, r1, pos2 + axis2 * len2, r2);
  x1 = clampf((u + mb * len2) / ma, -len1, len1);
  const bool h3 = sphere_sphere(t3, margin, pos1 + axis1 * x1, r1, pos2 - axis2 * len2, r2);
  // first two hits in order
  int n = 0;
  if (h0) { c0 = t0; n = 1; }
  if (h1) { if (n == 0) c0 = t1; else c1 = t1; n++; }
  if (h2 && n < 2) { if (n == 0) c0 = t2; else c1 = t2; n++; }
  if (h3 && n < 2) { if (n == 0) c0 = t3; else c1 = t3; n++; }
  return n;
}


// closest point of triangle abc to p (Ericson, Real-Time Collision Detection 5.1.5)
__device__ __forceinline__ V3 closest_on_triangle(V3 p, V3 a, V3 b, V3 c) {
  const V3 ab = b - a, ac = c - a, ap = p - a;
  const float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.f && d2 <= 0.f) return a;
  const V3 bp = p - b;
  const float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.f && d4 <= d3) return b;
  const float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) return a + ab * (d1 / (d1 - d3));
  const V3 cp = p - c;
  const float d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.f && d5 <= d6) return c;
  const float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) return a + ac * (d2 / (d2 - d6));
  const float va = d3 * d6 - d5 * d4;
  if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) return b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)));
  const float denom = 1.f / (va + vb + vc);
  return a + ab * (vb * denom) + ac * (vc * denom);
}

}  // namespace hb
#include "hb_mpr.hpp"
namespace hb {

// complete a contact frame from its normal and an optional tangent hint (mju_makeFrame)
__device__ __forceinline__ void make_frame(float* f, V3 n, V3 hint) {
  n = normalized(n);
  V3 t = hint;
  if (dot(t, t) < 0.25f) t = (n.y < 0.5f && n.y > -0.5f) ? V3{0.f, 1.f, 0.f} : V3{0.f, 0.f, 1.f};
  t = t - n * dot(n, t);
  t = normalized(t);
  V3 b = cross(n, t);
  st3(f, n); st3(f + 3, t); st3(f + 6, b);
}

// impedance sigmoid (getimpedance restatement); solimp = d0, dmax, width, midpoint, power
__device__ __forceinline__ float impedance(const float* solimp, float pos, float margin) {
  float d0 = clampf(solimp[0], HB_MINIMP, HB_MAXIMP), d1 = clampf(solimp[1], HB_MINIMP, HB_MAXIMP);
  float width = fmaxf(0.f, solimp[2]), mid = clampf(solimp[3], HB_MINIMP, HB_MAXIMP), power = fmaxf(1.f, solimp[4]);
  if (d0 == d1 || width <= HB_MINVAL) return 0.5f * (d0 + d1);
  float x = fabsf((pos - margin) / width);
  if (x >= 1.f) return d1;
  if (x <= 0.f) return d0;
  // both halves of the sigmoid are the same power curve, mirrored: one evaluation, and the usual exponent 2
  // (MuJoCo's default solimp) needs no powf at all
  const bool lower = x <= mid;
  const float t = lower ? x : 1.f - x, mm = lower ? mid : 1.f - mid;
  float y;
  if (power == 1.f) y = t;  // x or 1 - x: the curve is the identity
  else if (power == 2.f) y = t * t / mm;
  else y = powf(t, power) / powf(mm, power - 1.f);
  if (!lower) y = 1.f - y;
  return d0 + y * (d1 - d0);
}

// reference spring (K) and damper (B) of a constraint row from solref (mj_makeImpedance; oracle: make_constraint)
__device__ __forceinline__ void kb_from_solref(float solref0, float solref1, float solimp1, float timestep, bool refsafe, float& K, float& B) {
  const float dmax = clampf(solimp1, HB_MINIMP, HB_MAXIMP);
  if (solref0 > 0.f) {
    float tc = solref0;
    if (refsafe) tc = fmaxf(tc, 2.f * timestep);
    K = 1.f / fmaxf(HB_MINVAL, dmax * dmax * tc * tc * solref1 * solref1);
    B = 2.f / fmaxf(HB_MINVAL, dmax * tc);
  } else { K = -solref0 / fmaxf(HB_MINVAL, dmax * dmax); B = -solref1 / fmaxf(HB_MINVAL, dmax); }
}

// LDS record strides (floats).  Records read as ds_read_b128 by lanes that index different bodies / dofs are 16-byte aligned AND an odd
// multiple of 16 bytes apart: with the natural power-of-two strides (8, 16 floats) lanes b and b + 8 (b + 4) hit the same banks
// (measured: 14 % of LDS-active cycles were bank conflicts, profiles/r02_counters.json)
constexpr int kCdofStride = 12;  // per dof: angular[3], -, linear[3], -, (pad 4)
constexpr int kXpqStride = 12;   // per body: xpos[3], -, xquat[4], (pad 4)
constexpr int kIfStride = 20;    // per body: composite inertia[10] | cfrc[6], (pad 4)
constexpr int kWs = 36;  // 16-byte aligned rows: a row times a vector is eight ds_read_b128 pairs (dot32)
// one dof's motion axis record: s_cdof[8 d ..] = angular[3], -, linear[3], - (two ds_read_b128)
__device__ __forceinline__ void ld_cdof(const float* s_cdof, int d, float out[6]) {
  const float4* p = reinterpret_cast<const float4*>(s_cdof + kCdofStride * d);
  const float4 a = p[0], l = p[1];
  out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = l.x; out[4] = l.y; out[5] = l.z;
}

// 32-term dot product of a W row with a dof vector, both 16-byte aligned and zero beyond nv
__device__ __forceinline__ float dot32(const float* row, const float* v) {
  const float4* a = reinterpret_cast<const float4*>(row);
  const float4* b = reinterpret_cast<const float4*>(v);
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < 8; q++) { const float4 x = a[q], y = b[q]; acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; }
  return acc;
}

// ------------------------------------------------------------------------------------------

// ---- counter-based random numbers (env realism, rollout noise): one 32-bit word per (seed, global env, episode, step,
// stream, element): reproducible, order-free, the same on any split of the batch.  tests/env_ref.py restates them in numpy.
__device__ __forceinline__ unsigned rng_mix(unsigned h, unsigned v) {
  h ^= v; h *= 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13; h *= 0xC2B2AE3Du; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ unsigned rng_u32(unsigned seed, unsigned env, unsigned ep, unsigned step, unsigned stream, unsigned idx) {
  unsigned h = rng_mix(0x6A09E667u, seed);
  h = rng_mix(h, env); h = rng_mix(h, ep); h = rng_mix(h, step); h = rng_mix(h, stream); h = rng_mix(h, idx);
  return h;
}
__device__ __forceinline__ float rng_uniform(unsigned seed, unsigned env, unsigned ep, unsigned step, unsigned stream, unsigned idx) {
  return ((float)(rng_u32(seed, env, ep, step, stream, idx) >> 8) + 0.5f) * (1.f / 16777216.f);  // (0, 1)
}
__device__ __forceinline__ float rng_normal(unsigned seed, unsigned env, unsigned ep, unsigned step, unsigned stream, unsigned idx) {
  const float u1 = rng_uniform(seed, env, ep, step, stream, 2 * idx), u2 = rng_uniform(seed, env, ep, step, stream, 2 * idx + 1);
  return sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);  // Box-Muller
}
enum { RS_ACTION = 1, RS_JOINT_POS, RS_JOINT_VEL, RS_GYRO, RS_IMU, RS_DELAY, RS_PUSH, RS_XFRC };

// ---- dense helpers of the Newton solver ------------------------------------------------------------------------
// A symmetric nv x nv matrix (nv <= 32, identity beyond nv) lives one ROW PER LANE: lane l (and its mirror l + 32)
// holds row l & 31 in 32 registers.  Vectors live one element per lane (lanes 0..31).  Everything is readlane + fma
// on statically indexed registers: no LDS traffic, no cross-lane reductions.

// sum_j row[j] * x_j over LDS rows (N terms, the tails are zero by construction), x_j taken from lane j: all loads of
// the unrolled body are issued before the first use; the two-row form shares the broadcasts
template <int N>
__device__ __forceinline__ float rowdot(const float* row, float x) {
  float v[N];
#pragma unroll
  for (int j = 0; j < N; j++) v[j] = row[j];
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    a0 = __builtin_fmaf(v[j], rdlane(x, j), a0);
    a1 = __builtin_fmaf(v[j + 1], rdlane(x, j + 1), a1);
  }
  return a0 + a1;
}
template <int N>
__device__ __forceinline__ void rowdot2(const float* rowA, const float* rowB, float x, float& ra, float& rb) {
  float va[N], vb[N];
#pragma unroll
  for (int j = 0; j < N; j++) { va[j] = rowA[j]; vb[j] = rowB[j]; }
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    const float x0 = rdlane(x, j), x1 = rdlane(x, j + 1);
    a0 = __builtin_fmaf(va[j], x0, a0); b0 = __builtin_fmaf(vb[j], x0, b0);
    a1 = __builtin_fmaf(va[j + 1], x1, a1); b1 = __builtin_fmaf(vb[j + 1], x1, b1);
  }
  ra = a0 + a1; rb = b0 + b1;
}

// Right-looking Cholesky A = L L' in place (mju_cholFactor, mujoco.h:1211, incl. its diagonal floor) of the leading
// N x N block (N = nv rounded up to 4; identity beyond nv).  Every lane updates its whole row, so that lane i ends up with
//   element k < i: L[i][k] d_k;   element k > i: S_i[i][k] d_i^2 = L[k][i] d_i   (S_i: the Schur complement at pivot i),
// i.e. row i of L and column i of L, each pre-scaled so that the two triangular solves below are one v_readlane and one
// fma per step.  Returns d_i = 1 / L[i][i].
// The dependent chain of a pivot is readlane - rsq - mul - readlane - fma: the NEXT pivot column is updated first, with
// its multiplier taken by v_readlane; the rest of the trailing update goes through a 64-float LDS line (one ds_write,
// broadcast ds_read_b128s) as packed math on register pairs (v_pk_fma_f32) and overlaps the following pivots.
template <int N>
__device__ __forceinline__ float chol_rows(f32x2 (&A)[16], float* s_l, int li, int lane) {
  float dv = 1.f;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const float akk = A[k >> 1][k & 1];
    const float piv = fmaxf(rdlane(akk, k), HB_MINVAL);
    const float d = __builtin_amdgcn_rsqf(piv);
    const float l = akk * d;
    s_l[lane] = l;  // all 64 lanes (the upper half lands in the next 32 floats): an unconditional store keeps the code straight-line
    if (li == k) dv = d;
    const float lm = li > k ? -l : 0.f;  // rows at and above the pivot are final
    if (k + 1 < N) A[(k + 1) >> 1][(k + 1) & 1] = __builtin_fmaf(lm, rdlane(l, k + 1), A[(k + 1) >> 1][(k + 1) & 1]);
    const f32x2 lm2 = {lm, lm};
#pragma unroll
    for (int c = (k + 2) / 4; c < N / 4; c++) {
      const float4 lv = *reinterpret_cast<const float4*>(s_l + 4 * c);
      const float lq[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int lo = 4 * c + 2 * h, hi = lo + 1;
        if (lo > k + 1) A[2 * c + h] = lm2 * f32x2{lq[2 * h], lq[2 * h + 1]} + A[2 * c + h];
        else if (hi > k + 1) A[2 * c + h][1] = __builtin_fmaf(lm, lq[2 * h + 1], A[2 * c + h][1]);
      }
    }
    A[k >> 1][k & 1] = li > k ? l * d : akk;
  }
  const float dv2 = dv * dv;
#pragma unroll
  for (int k = 1; k < N; k++) A[k >> 1][k & 1] = li < k ? A[k >> 1][k & 1] * dv2 : A[k >> 1][k & 1];
  return dv;
}

// x = (L L')^-1 g for the factor left by chol_rows (mju_cholSolve, mujoco.h:1214): column-oriented forward and backward
// substitution; with the pre-scaled factor the element solved at step k is lane k's running value itself: v_readlane
// broadcasts it, one fma updates every other row, v_writelane (off the chain) keeps it
template <int N>
__device__ __forceinline__ float chol_solve_rows(const f32x2 (&A)[16], float dv, float g) {
  float r = g;
  int y = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int rk = __builtin_amdgcn_readlane(__float_as_int(r), k);
    y = hb_writelane(rk, k, y);
    r = __builtin_fmaf(-A[k >> 1][k & 1], __int_as_float(rk), r);  // rows below k; rows above are done (their r is dead)
  }
  r = __int_as_float(y) * (dv * dv);  // y_k = r_k d_k, and the backward pass runs on d_i-scaled rows
  int x = 0;
#pragma unroll
  for (int k = N - 1; k >= 0; k--) {
    const int xk = __builtin_amdgcn_readlane(__float_as_int(r), k);
    x = hb_writelane(xk, k, x);
    r = __builtin_fmaf(-A[k >> 1][k & 1], __int_as_float(xk), r);  // rows above k
  }
  return __int_as_float(x);
}

// ---- symmetric elimination on the matrix cores (Newton instantiation of order <= 28) -----------------------------
// A symmetric 32 x 32 matrix S lives in the accumulator layout of v_mfma_f32_32x32x2_f32 (f32x16 per lane: lane = column
// + 32 * half, register r = row crow(r) + 4 * half).  In that layout ROW k of S is one register on the 32 lanes of one
// half - which is exactly the shape of an MFMA operand - and by symmetry it is also column k.  Gaussian elimination of
// two pivots is therefore ONE rank-2 MFMA update  S -= a b'  with b = the two pivot rows (one per half, moved by a
// v_permlane32_swap) and a = -b / D masked to the rows below the pivot: about twenty-five VALU instructions per pivot
// pair instead of a trailing update of N - k columns.  The same multipliers applied to T (started at I) by a second
// MFMA leave T = L^-1; the right-hand side rides along as row / column 31, so z_k = U[k][31] / D_k = (D^-1 L^-1 g)_k
// falls out of the pivot rows, and x = T' z is sixteen lane-local fmas plus one swap.  NP pivot pairs (order 2 NP <= 30);
// rows beyond are identity padding and are never pivots.
// (Building T in place of the eliminated triangle - pivot row with a doubled diagonal, one MFMA per pair - was measured
// too: fewer MFMAs but more VALU work per pair, and VALU issue is what the two waves of a SIMD compete for: slower.)
__device__ __forceinline__ constexpr int crow(int r) { return (r & 3) + 8 * (r >> 2); }

template <int NP>
__device__ __forceinline__ float sym_solve_mfma(f32x16 X, float g, int lane) {
  const int li = lane & 31, half = lane >> 5;
  const bool up = half != 0;
  {  // right-hand side into row 31 and column 31 (g is mirrored in both halves and zero beyond nv): S += e31 g' + g e31'
    const float e31 = li == 31 ? 1.f : 0.f;
    X = __builtin_amdgcn_mfma_f32_32x32x2f32(up ? g : e31, up ? e31 : g, X, 0, 0, 0);
  }
  f32x16 T, Z;
  const int q = li - 4 * half;
#pragma unroll
  for (int r = 0; r < 16; r++) { T[r] = q == crow(r) ? 1.f : 0.f; Z[r] = 0.f; }
  const float lik = (float)(li - half);  // row index minus the pivot slot of this half
#pragma unroll
  for (int b = 0; b < NP; b++) {
    const int k0 = 2 * b, r0 = (k0 & 3) + 4 * (k0 >> 3), h0 = (k0 >> 2) & 1, L0 = 32 * h0;
    const float rowa = X[r0], rowb = X[r0 + 1], ta = T[r0], tb = T[r0 + 1];  // rows k0, k0 + 1 on the lanes of half h0
    const float inv0 = __builtin_amdgcn_rcpf(fmaxf(rdlane(rowa, L0 + k0), HB_MINVAL));
    const float m = rdlane(rowb, L0 + k0) * inv0;
    const float rowb1 = __builtin_fmaf(-m, rowa, rowb), tb1 = __builtin_fmaf(-m, ta, tb);  // row k0 + 1 after pivot k0
    const float inv1 = __builtin_amdgcn_rcpf(fmaxf(rdlane(rowb1, L0 + k0 + 1), HB_MINVAL));
    const u32x2 sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(rowa), __float_as_uint(rowb1), false, false);
    const u32x2 st = __builtin_amdgcn_permlane32_swap(__float_as_uint(ta), __float_as_uint(tb1), false, false);
    const float vb = __uint_as_float(h0 ? sx.y : sx.x), vt = __uint_as_float(h0 ? st.y : st.x);  // half 0: row k0, half 1: row k0 + 1
    const float below = __builtin_amdgcn_fmed3f(lik - (float)k0, 0.f, 1.f);                       // 1 on the rows below this half's pivot
    const float va = -(vb * (up ? inv1 : inv0)) * below;
    X = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, X, 0, 0, 0);
    T = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vt, T, 0, 0, 0);
    const float z0 = rdlane(rowa, L0 + 31) * inv0, z1 = rdlane(rowb1, L0 + 31) * inv1;
    if (half == h0) { Z[r0] = z0; Z[r0 + 1] = z1; }
  }
  float p = 0.f, p1 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r += 2) { p = __builtin_fmaf(T[r], Z[r], p); p1 = __builtin_fmaf(T[r + 1], Z[r + 1], p1); }
  p += p1;
  const u32x2 sp = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
  return p + __uint_as_float(up ? sp.x : sp.y);
}

// a dof vector valid on lanes 0..31, copied to both halves
__device__ __forceinline__ float rdlane_mirror(float x, int lane) {
  const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return lane >= 32 ? __uint_as_float(sw.x) : x;
}

// the dense mass matrix ([32][kCs] in LDS, identity beyond nv) in the accumulator layout
__device__ __forceinline__ f32x16 load_sym(const float* s_Md, int stride, int lane) {
  const float* pl = s_Md + 4 * (lane >> 5) * stride + (lane & 31);
  f32x16 X;
#pragma unroll
  for (int r = 0; r < 16; r++) X[r] = pl[crow(r) * stride];
  return X;
}

// the {M, H} pairs in the accumulator layout (WHICH = 0: M, 1: H = M + h B), straight from the sparse storage
template <int WHICH>
__device__ __forceinline__ f32x16 load_sym_pairs(DevModelRef M, const f32x2* s_qLD, int lane0) {
  // (the lane id is re-materialised so that the sixteen table words are fetched again at every use instead of being
  // kept in registers from the W stage, across the PGS sweeps, to the Euler solve)
  int lane;
  asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane0));
  int e[16];
#pragma unroll
  for (int r = 0; r < 16; r++) e[r] = M.mdense_c[r * 64 + lane];
  f32x16 X;
#pragma unroll
  for (int r = 0; r < 16; r++) X[r] = s_qLD[e[r]][WHICH];
  return X;
}

// Elimination only (PGS instantiation): T = L^-1 and, per register and half, D^-1/2 of that register's row, so that
// W = T' D^-1/2 (M^-1 = W W') can be written out.  Same rank-2 updates as sym_solve_mfma, no right-hand side.
template <int NP>
__device__ __forceinline__ void sym_factor_mfma(f32x16 X, f32x16& T, f32x16& S, int lane) {
  const int li = lane & 31, half = lane >> 5;
  const bool up = half != 0;
  const int q = li - 4 * half;
#pragma unroll
  for (int r = 0; r < 16; r++) { T[r] = q == crow(r) ? 1.f : 0.f; S[r] = 1.f; }
  const float lik = (float)(li - half);
#pragma unroll
  for (int b = 0; b < NP; b++) {
    const int k0 = 2 * b, r0 = (k0 & 3) + 4 * (k0 >> 3), h0 = (k0 >> 2) & 1, L0 = 32 * h0;
    const float rowa = X[r0], rowb = X[r0 + 1], ta = T[r0], tb = T[r0 + 1];
    const float d0 = fmaxf(rdlane(rowa, L0 + k0), HB_MINVAL);
    const float inv0 = __builtin_amdgcn_rcpf(d0);
    const float m = rdlane(rowb, L0 + k0) * inv0;
    const float rowb1 = __builtin_fmaf(-m, rowa, rowb), tb1 = __builtin_fmaf(-m, ta, tb);
    const float d1 = fmaxf(rdlane(rowb1, L0 + k0 + 1), HB_MINVAL);
    const float inv1 = __builtin_amdgcn_rcpf(d1);
    const u32x2 sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(rowa), __float_as_uint(rowb1), false, false);
    const u32x2 st = __builtin_amdgcn_permlane32_swap(__float_as_uint(ta), __float_as_uint(tb1), false, false);
    const float vb = __uint_as_float(h0 ? sx.y : sx.x), vt = __uint_as_float(h0 ? st.y : st.x);
    const float below = __builtin_amdgcn_fmed3f(lik - (float)k0, 0.f, 1.f);
    const float va = -(vb * (up ? inv1 : inv0)) * below;
    X = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, X, 0, 0, 0);
    T = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vt, T, 0, 0, 0);
    if (half == h0) { S[r0] = __builtin_amdgcn_rsqf(d0); S[r0 + 1] = __builtin_amdgcn_rsqf(d1); }
  }
}

// W[c][row] = T[row][c] * S(row): lane (c, half) owns four runs of four consecutive rows: four 16-byte stores
__device__ __forceinline__ void store_w_rows(float* W, int stride, const f32x16& T, const f32x16& S, int lane) {
  float* p = W + (lane & 31) * stride + 4 * (lane >> 5);
#pragma unroll
  for (int g = 0; g < 4; g++)
    *reinterpret_cast<float4*>(p + 8 * g) = {T[4 * g] * S[4 * g], T[4 * g + 1] * S[4 * g + 1], T[4 * g + 2] * S[4 * g + 2], T[4 * g + 3] * S[4 * g + 3]};
}

// WT[row][c] = T[row][c] * S(row): the transpose of store_w_rows' matrix (consecutive lanes, consecutive addresses)
__device__ __forceinline__ void store_w_cols(float* WT, int stride, const f32x16& T, const f32x16& S, int lane) {
  float* p = WT + 4 * (lane >> 5) * stride + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; r++) p[crow(r) * stride] = T[r] * S[r];
}

// ---- general narrowphase (COLL = 1): every pair kind of the classic one plus mesh hulls and height-field prisms ----------------
// exclusive prefix sum of a small non-negative count over the 64 lanes (and the total)
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total) {
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
  total = __builtin_amdgcn_readlane(x, 63);
  return x - v;
}

// a mesh geom's hull for the support function: the start records of its mesh
__device__ __forceinline__ void set_mesh(DevModelRef M, CObj& o, int g) {
  o.vert = M.mesh_start + (M.geom_meshnum[g] > 0 ? kMeshStart * M.geom_dataid[g] : 0);
  o.nbr = M.mesh_nbr;
}

// separating-axis test of the oriented bounding boxes of geoms g1, g2 (centres dp apart, orientations q1, q2, each box grown by
// `grow`): false only if an axis separates them (Gottschalk's 15 axes; the epsilon on |R| keeps near-parallel edge pairs from
// reporting a separation that rounding made up, and the slack keeps boxes that touch to within rounding together)
__device__ __forceinline__ bool boxes_touch(DevModelRef M, int g1, int g2, V3 dp, Q4 q1, Q4 q2, float grow) {
  const float slack = 1e-6f;
  float A[9], B[9], R[9], AR[9];
  q2mat(A, q1); q2mat(B, q2);
  const V3 ha = ld3(M.geom_half + 3 * g1), hb3 = ld3(M.geom_half + 3 * g2);
  const float a[3] = {ha.x + grow + slack, ha.y + grow + slack, ha.z + grow + slack}, b[3] = {hb3.x + grow + slack, hb3.y + grow + slack, hb3.z + grow + slack};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { R[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j]; AR[3 * i + j] = fabsf(R[3 * i + j]) + 1e-6f; }  // A' B
  const float t[3] = {A[0] * dp.x + A[3] * dp.y + A[6] * dp.z, A[1] * dp.x + A[4] * dp.y + A[7] * dp.z, A[2] * dp.x + A[5] * dp.y + A[8] * dp.z};  // A' dp
  bool apart = false;
#pragma unroll
  for (int i = 0; i < 3; i++) apart |= fabsf(t[i]) > a[i] + b[0] * AR[3 * i] + b[1] * AR[3 * i + 1] + b[2] * AR[3 * i + 2];
#pragma unroll
  for (int j = 0; j < 3; j++) apart |= fabsf(t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j]) > a[0] * AR[j] + a[1] * AR[3 + j] + a[2] * AR[6 + j] + b[j];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const float ra = a[i1] * AR[3 * i2 + j] + a[i2] * AR[3 * i1 + j], rb = b[j1] * AR[3 * i + j2] + b[j2] * AR[3 * i + j1];
      apart |= fabsf(t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j]) > ra + rb;
    }
  return !apart;
}

// lowest point (z, relative to the geom's position) of geom g with orientation q in the frame the query is made in: the support
// function along -z.  Single precision: the value only decides whether a prism under the geom is searched (the prism's top is
// compared with it), and a vertex within rounding of the lowest one gives the same answer to within that rounding.
__device__ __forceinline__ float lowest_point(DevModelRef M, int g, int type, float r, float h, Q4 q) {
  float m[9];
  q2mat(m, q);
  const V3 ld = {-m[6], -m[7], -m[8]};  // mat' (0, 0, -1): the query direction in the geom's frame
  if (type == 2) return -r;
  if (type == 3) return -r - fabsf(ld.z) * h;  // mat (ld r + (0, 0, sign(ld.z) h)) . z = -r - |ld.z| h
  if (type != 7 || M.geom_meshnum[g] <= 0) return -M.geom_rbound[g];
  const float4 HB_CONST* start = M.mesh_start + kMeshStart * M.geom_dataid[g];
  const float ax = fabsf(ld.x), ay = fabsf(ld.y), az = fabsf(ld.z);
  const int axis = ax >= ay ? (ax >= az ? 0 : 2) : (ay >= az ? 1 : 2);
  const float major = axis == 0 ? ld.x : (axis == 1 ? ld.y : ld.z);
  const float inv = 1.f / fabsf(major);
  const float u = (axis == 0 ? ld.y : (axis == 1 ? ld.z : ld.x)) * inv, v = (axis == 0 ? ld.z : (axis == 1 ? ld.x : ld.y)) * inv;
  const int iu = min(max((int)floorf((u + 1.f) * 2.f), 0), 3), iv = min(max((int)floorf((v + 1.f) * 2.f), 0), 3);
  const float4 s0 = start[(2 * axis + (major < 0.f ? 1 : 0)) * 16 + iu * 4 + iv];
  float bd = __builtin_fmaf(s0.x, ld.x, __builtin_fmaf(s0.y, ld.y, s0.z * ld.z));  // (one fixed operation sequence for start and neighbours: hb_mpr.hpp, hull_val)
  int link = __float_as_int(s0.w);
  for (int guard = 0; guard < 256; guard++) {
    const int adr = link >> 8, nch = link & 255;
    bool moved = false;
    int nlink = link;
    for (int c = 0; c < nch; c++) {
      float4 nb[kMeshChunk];
#pragma unroll
      for (int i = 0; i < kMeshChunk; i++) nb[i] = M.mesh_nbr[adr + c * kMeshChunk + i];
#pragma unroll
      for (int i = 0; i < kMeshChunk; i++) {
        const float val = __builtin_fmaf(nb[i].x, ld.x, __builtin_fmaf(nb[i].y, ld.y, nb[i].z * ld.z));
        if (val > bd) { bd = val; nlink = __float_as_int(nb[i].w); moved = true; }
      }
    }
    if (!moved) break;
    link = nlink;
  }
  return -bd;  // the support point's z in the query frame is -(v . ld)
}

// mj_collision for models with mesh geoms and / or a height field (the reference's own robot: simulation/assets/world.xml:14-58).
// Three passes over LDS lists: (1) broadphase per candidate pair, survivors in pair order; (2) work items: one per pair, or one
// per prism of the sub-grid under the geom for a height-field pair (mjc_ConvexHField's double loop, flattened); (3) narrowphase,
// one work item per lane, contacts appended in work-item order (= the oracle's order: pair, then grid row, then strip position).
// The passes are separate functions because the STAGED step (launch_step) runs them in separate kernels: (1) + (2) in
// hb_pose_kernel, (3) in hb_narrow_kernel at four times the occupancy the step kernel allows, and the step kernel itself only
// appends the results (collide_gather).
//
// passes (1) and (2): s_scratch receives the pair list, the sub-grids of height-field pairs and the work items; returns the number of work items
__device__ __forceinline__ int build_work_list(DevModelRef M, int lane, const float* s_gpos, const float* s_gaxis, const float* s_gquat, int* s_scratch, int& status) {
  int* s_list = s_scratch;                   // [kListMax]
  int* s_pinfo = s_scratch + kListMax;       // [kListMax][4]: rmin, cmin, ncols of a height-field pair's sub-grid, lowest point of the geom (float bits)
  int* s_work = s_pinfo + 4 * kListMax;      // [kWorkMax]: list index << 16 | sub-item
  int nlist = 0;
  for (int p0 = 0; p0 < M.npair; p0 += kGroup) {
    const int p = p0 + lane;
    bool pass = false;
    if (p < M.npair) {
      const float4 c0 = M.crec[3 * (size_t)p], c1 = M.crec[3 * (size_t)p + 1];
      const int g1 = __float_as_int(c0.x), g2 = __float_as_int(c0.y), t1 = __float_as_int(c0.z) & 255;
      const V3 dp = ld3(s_gpos + 3 * g2) - ld3(s_gpos + 3 * g1);
      if (t1 == 0) pass = dot(dp, ld3(s_gaxis + 3 * g1)) <= c0.w + c1.y;
      else if (t1 == 1) pass = true;
      else {
        const float bound = c1.x + c1.y + c0.w;
        pass = dot(dp, dp) <= bound * bound;
        // a pair that goes to the portal search: the geoms' oriented bounding boxes first (each grown by half the margin).  Boxes
        // that a separating axis keeps apart hold hulls that do not touch: the search would say so too, after two hull climbs
        // per support query (the robot's limbs are long and thin: most pairs that pass the bounding spheres stop here)
        const int t2 = (__float_as_int(c0.z) >> 8) & 255;
        if (pass && M.box_cull && (t1 == 7 || t2 == 7)) pass = boxes_touch(M, g1, g2, dp, ldq(s_gquat + 4 * g1), ldq(s_gquat + 4 * g2), 0.5f * c0.w);
      }
    }
    const unsigned long long bal = __ballot(pass);
    const int slot = nlist + __popcll(bal & ((1ull << lane) - 1ull));
    if (pass && slot < kListMax) s_list[slot] = p;
    nlist += __popcll(bal);
  }
  nlist = uniform(nlist);
  if (nlist > kListMax) { status |= (1 << 1); nlist = kListMax; }
  gsync();
  int nwork = 0;
  for (int i0 = 0; i0 < nlist; i0 += kGroup) {
    const int idx = i0 + lane;
    int cnt = 0;
    if (idx < nlist) {
      const int p = s_list[idx];
      const float4 c0 = M.crec[3 * (size_t)p], c1 = M.crec[3 * (size_t)p + 1];
      const int g1 = __float_as_int(c0.x), g2 = __float_as_int(c0.y), t1 = __float_as_int(c0.z) & 255;
      cnt = 1;
      if (t1 == 0 && ((__float_as_int(c0.z) >> 8) & 255) == 7) cnt = 2;  // mjc_PlaneConvex: up to four contacts, two per work item
      if (t1 == 1) {
        // mjc_ConvexHField's culling.  The sub-grid comes from the geom's bounding sphere in place of its exact bounding box (a superset
        // of MuJoCo's prisms in x and y: the extra ones lie outside the geom's footprint and cannot touch it), the height test from the
        // geom's exact lowest point in the field's frame (one support query along -z; MuJoCo's box has the same bottom), so that a
        // prism under a raised limb is not searched at all.
        float hm[9];
        q2mat(hm, ldq(M.geom_quat + 4 * g1));
        const int hid = M.geom_dataid[g1];
        const float sx = M.hfield_size[4 * hid], sy = M.hfield_size[4 * hid + 1], sz = M.hfield_size[4 * hid + 2], sb = M.hfield_size[4 * hid + 3];
        const int nrow = M.hfield_nrow[hid], ncol = M.hfield_ncol[hid];
        const V3 dif = ld3(s_gpos + 3 * g2) - ld3(s_gpos + 3 * g1);
        const V3 q = {hm[0] * dif.x + hm[3] * dif.y + hm[6] * dif.z, hm[1] * dif.x + hm[4] * dif.y + hm[7] * dif.z, hm[2] * dif.x + hm[5] * dif.y + hm[8] * dif.z};
        const float reach = c1.y + c0.w;
        if (sx < q.x - reach || -sx > q.x + reach || sy < q.y - reach || -sy > q.y + reach || sz < q.z - reach || -sb > q.z + reach) cnt = 0;
        else {
          int cmin = (int)floorf((q.x - reach + sx) / (2.f * sx) * (float)(ncol - 1)), cmax = (int)ceilf((q.x + reach + sx) / (2.f * sx) * (float)(ncol - 1));
          int rmin = (int)floorf((q.y - reach + sy) / (2.f * sy) * (float)(nrow - 1)), rmax = (int)ceilf((q.y + reach + sy) / (2.f * sy) * (float)(nrow - 1));
          cmin = max(cmin, 0); rmin = max(rmin, 0); cmax = min(cmax, ncol - 1); rmax = min(rmax, nrow - 1);
          const int ncols = max(cmax - cmin, 0), nrows = max(rmax - rmin, 0);
          cnt = nrows * 2 * ncols;
          const float4 c2 = M.crec[3 * (size_t)p + 2];
          const float loz = q.z + lowest_point(M, g2, (__float_as_int(c0.z) >> 8) & 255, c2.x, c2.y, qmul(qconj(ldq(M.geom_quat + 4 * g1)), ldq(s_gquat + 4 * g2)));
          s_pinfo[4 * idx] = rmin; s_pinfo[4 * idx + 1] = cmin; s_pinfo[4 * idx + 2] = ncols; s_pinfo[4 * idx + 3] = __float_as_int(loz);
        }
      }
    }
    int total;
    const int base = nwork + wave_excl_scan(cnt, lane, total);
    for (int k = 0; k < cnt; k++) if (base + k < kWorkMax) s_work[base + k] = (idx << 16) | k;
    nwork += total;
  }
  nwork = uniform(nwork);
  if (nwork > kWorkMax) { status |= (1 << 1); nwork = kWorkMax; }
  gsync();
  return nwork;
}

// pass (3) for one work item: pair p (sub-item `sub` of the sub-grid rmin, cmin, ncols for a height-field pair) -> n contacts (0..2).
// MODE 0: all of it.  MODE 1 (hb_pose_kernel): everything but the portal search; returns whether the item needs one (then n = 0).
// MODE 2 (hb_narrow_kernel): an item MODE 1 said needs the portal search.
template <int MODE, int MESH = 1>
__device__ __forceinline__ int eval_work_item(DevModelRef M, const float* hdata_all, bool have, int p, int sub, int rmin, int cmin, int ncols, float loz,
                                              const float* s_gpos, const float* s_gaxis, const float* s_gquat, ConOut& co0, ConOut& co1, int& n, V3& hint) {
  float4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
  if (have) { const float4 HB_CONST* N = M.crec + 3 * (size_t)p; c0 = N[0]; c1 = N[1]; c2 = N[2]; }
  co0.dist = 0.f; co0.pos = {0.f, 0.f, 0.f}; co0.n = {0.f, 0.f, 1.f}; co1 = co0;
  n = 0;
  hint = {0.f, 0.f, 0.f};
  const int g1 = __float_as_int(c0.x), g2 = __float_as_int(c0.y);
  const int t1 = __float_as_int(c0.z) & 255, t2 = (__float_as_int(c0.z) >> 8) & 255;
  const float margin = c0.w;
  // the two objects of an MPR test (one call site below)
  CObj o1, o2;
  int mpr_kind = 0;  // 0: no MPR for this item, 1: prism vs geom (field frame), 2: geom vs geom (world frame)
  float hm[9];
  V3 pos1 = {0.f, 0.f, 0.f};
  if (have) {
    pos1 = ld3(s_gpos + 3 * g1);
    const V3 pos2 = ld3(s_gpos + 3 * g2), ax2 = ld3(s_gaxis + 3 * g2);
    const float rb1 = c1.x, rb2 = c1.y, r2 = c2.x, l2 = c2.y;
    (void)rb1;
    if (t1 == 1) {
      q2mat(hm, ldq(M.geom_quat + 4 * g1));
      const int hid = M.geom_dataid[g1];
      const float sx = M.hfield_size[4 * hid], sy = M.hfield_size[4 * hid + 1], sz = M.hfield_size[4 * hid + 2], sb = M.hfield_size[4 * hid + 3];
      const int nrow = M.hfield_nrow[hid], ncol = M.hfield_ncol[hid];
      const float* data = hdata_all + M.hfield_adr[hid];
      const int r = rmin + sub / (2 * ncols), j = sub % (2 * ncols);
      const float dx = 2.f * sx / (float)(ncol - 1), dy = 2.f * sy / (float)(nrow - 1);
      // strip vertex s of grid row r: column cmin + s / 2, grid row r + 1 for even s, r for odd s (mjc_ConvexHField: dr = {1, 0})
      V3 tv[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int sidx = j + k, cc = cmin + (sidx >> 1), rr = r + ((sidx & 1) ? 0 : 1);
        tv[k] = {dx * (float)cc - sx, dy * (float)rr - sy, data[rr * ncol + cc] * sz + margin};
      }
      // geom 2 in the field's frame
      const V3 dif = pos2 - pos1;
      o2.pos = {hm[0] * dif.x + hm[3] * dif.y + hm[6] * dif.z, hm[1] * dif.x + hm[4] * dif.y + hm[7] * dif.z, hm[2] * dif.x + hm[5] * dif.y + hm[8] * dif.z};
      if (MODE == 2 || !(tv[0].z < loz && tv[1].z < loz && tv[2].z < loz)) {  // prism below the geom's lowest point (loz: build_work_list); MODE 2: tested before
        float m2[9];
        q2mat(m2, ldq(s_gquat + 4 * g2));
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b = 0; b < 3; b++) o2.mat[3 * a + b] = hm[a] * m2[b] + hm[3 + a] * m2[3 + b] + hm[6 + a] * m2[6 + b];  // hm' m2
        o2.type = t2; o2.r = r2; o2.h = l2; o2.margin = margin;
        if constexpr (MESH != 0) set_mesh(M, o2, g2);
        o1.type = -1; o1.pos = {0.f, 0.f, 0.f}; o1.r = o1.h = o1.margin = 0.f; o1.vert = M.mesh_start; o1.nbr = M.mesh_nbr;
#pragma unroll
        for (int a = 0; a < 9; a++) o1.mat[a] = 0.f;
        o1.p0 = {tv[0].x, tv[0].y, -sb}; o1.p1 = {tv[1].x, tv[1].y, -sb}; o1.p2 = {tv[2].x, tv[2].y, -sb};
        o1.p3 = tv[0]; o1.p4 = tv[1]; o1.p5 = tv[2];
        mpr_kind = 1;
      }
    } else if (t1 == 0 && t2 == 7) {
      // mjc_PlaneConvex: no portal search (never reaches the narrowphase kernel); work item `sub` carries contacts 2 sub, 2 sub + 1
      if constexpr (MODE != 2) {
        const V3 normal = ld3(s_gaxis + 3 * g1);
        if (dot(pos2 - pos1, normal) <= margin + rb2) {
          q2mat(o2.mat, ldq(s_gquat + 4 * g2));
          o2.type = t2; o2.pos = pos2; o2.r = r2; o2.h = l2; o2.margin = 0.f; set_mesh(M, o2, g2);
          const int total = plane_hull(o2, pos1, normal, margin, 0.3f * rb2, 2 * sub, co0.dist, co0.pos, co1.dist, co1.pos);
          co0.n = normal; co1.n = normal;
          n = min(max(total - 2 * sub, 0), 2);
        }
      }
    } else if (MESH != 0 && (t1 == 7 || t2 == 7)) {
      // mjc_Convex: both geoms in the world frame, each inflated by half the margin
      q2mat(o1.mat, ldq(s_gquat + 4 * g1));
      q2mat(o2.mat, ldq(s_gquat + 4 * g2));
      o1.type = t1; o1.pos = pos1; o1.r = c1.z; o1.h = c1.w; o1.margin = 0.5f * margin; set_mesh(M, o1, g1);
      o2.type = t2; o2.pos = pos2; o2.r = r2; o2.h = l2; o2.margin = 0.5f * margin; set_mesh(M, o2, g2);
      o1.p0 = o1.p1 = o1.p2 = o1.p3 = o1.p4 = o1.p5 = V3{0.f, 0.f, 0.f};
      mpr_kind = 2;
    } else if (MODE == 2) {  // (the analytic pairs never reach the narrowphase kernel)
    } else if (t1 == 0) {
      const V3 normal = ld3(s_gaxis + 3 * g1);
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
    } else {
      const float r1 = c1.z, l1 = c1.w;
      if (t1 == 2 && t2 == 2) n = sphere_sphere(co0, margin, pos1, r1, pos2, r2) ? 1 : 0;
      else if (t1 == 2) {
        const float x = clampf(dot(ax2, pos1 - pos2), -l2, l2);
        n = sphere_sphere(co0, margin, pos1, r1, pos2 + ax2 * x, r2) ? 1 : 0;
      } else n = capsule_capsule(co0, co1, margin, pos1, ld3(s_gaxis + 3 * g1), r1, l1, pos2, ax2, r2, l2);
    }
  }
  if constexpr (MODE == 1) return mpr_kind;
  if (mpr_kind) {
    float depth;
    V3 dir, vec;
    const bool hit = mpr_penetration<MESH>(o1, o2, M.mpr_iterations, (double)M.mpr_tolerance, depth, dir, vec);
    if (mpr_kind == 1) {
      if (hit && depth >= 2.220446e-16f) {
        co0.dist = -depth;
        co0.n = mrot(hm, dir);
        co0.pos = mrot(hm, vec) + pos1;
        n = 1;
      }
    } else if (hit && !(dir.x == 0.f && dir.y == 0.f && dir.z == 0.f)) {
      co0.dist = margin - depth;
      co0.n = dir;
      co0.pos = vec;
      n = 1;
    }
    if (n) {  // mjc_fixNormal: spheres and capsules know their own normal
      float m1[9], m2[9];
      q2mat(m1, ldq(s_gquat + 4 * g1));
      q2mat(m2, ldq(s_gquat + 4 * g2));
      V3 n1, n2;
      const bool h1 = analytic_normal(t1, pos1, m1, c1.w, co0.pos, n1), h2 = analytic_normal(t2, ld3(s_gpos + 3 * g2), m2, c2.y, co0.pos, n2);
      if (h1 || h2) {
        V3 nn = {0.f, 0.f, 0.f};
        if (h1) nn = nn + n1;
        if (h2) nn = nn - n2;
        float len;
        nn = normalized(nn, &len);
        if (len >= HB_MINVAL) co0.n = nn;
      }
    }
  }
  return mpr_kind;
}

// ordered append of one round's results: slot = ncon + (# contacts of lower lanes)
template <int NC>
__device__ __forceinline__ void append_contacts(int lane, float* s_con, int& ncon, int n, const ConOut& co0, const ConOut& co1, V3 hint, int p) {
  const unsigned long long b1 = __ballot(n >= 1), b2 = __ballot(n >= 2);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int slot = ncon + __popcll(b1 & lt) + __popcll(b2 & lt);
  if (n >= 1 && slot < NC) {
    float* c = s_con + slot * kConStride;
    c[C_DIST] = co0.dist;
    st3(c + C_POS, co0.pos);
    make_frame(c + C_FRAME, co0.n, hint);
    c[C_PAIR] = __int_as_float(p);
  }
  if (n >= 2 && slot + 1 < NC) {
    float* c = s_con + (slot + 1) * kConStride;
    c[C_DIST] = co1.dist;
    st3(c + C_POS, co1.pos);
    make_frame(c + C_FRAME, co1.n, hint);
    c[C_PAIR] = __int_as_float(p);
  }
  ncon += __popcll(b1) + __popcll(b2);
}

// the fused form: all three passes in the step kernel
template <int NC>
__device__ __forceinline__ int collide_general(DevModelRef M, const float* hdata_all, int lane, const float* s_gpos, const float* s_gaxis, const float* s_gquat,
                                               float* s_con, int* s_scratch, int& status) {
  const int* s_list = s_scratch;
  const int* s_pinfo = s_scratch + kListMax;
  const int* s_work = s_pinfo + 4 * kListMax;
  const int nwork = build_work_list(M, lane, s_gpos, s_gaxis, s_gquat, s_scratch, status);
  int ncon = 0;
  for (int w0 = 0; w0 < nwork; w0 += kGroup) {
    const bool have = w0 + lane < nwork;
    const int item = have ? s_work[w0 + lane] : 0;
    const int idx = item >> 16, sub = item & 0xffff;
    const int p = have ? s_list[idx] : 0;
    ConOut co0, co1;
    int n;
    V3 hint;
    eval_work_item<0>(M, hdata_all, have, p, sub, s_pinfo[4 * idx], s_pinfo[4 * idx + 1], s_pinfo[4 * idx + 2], __int_as_float(s_pinfo[4 * idx + 3]), s_gpos, s_gaxis, s_gquat, co0, co1, n, hint);
    append_contacts<NC>(lane, s_con, ncon, n, co0, co1, hint, p);
  }
  if (ncon > NC) { status |= (1 << 1); ncon = NC; }
  return ncon;
}

// the staged form's third part: the work items were evaluated by hb_narrow_kernel; append its results in work-item order
template <int NC>
__device__ __forceinline__ int collide_gather(DevModelRef M, int lane, int env, const StageBufs& G, const float* s_gaxis, float* s_con, int& status) {
  // (overflow of the pair / work lists was flagged by hb_pose_kernel, which clamps the count it stores)
  const int nwork = min(max(G.nwork[env], 0), kWorkMax);
  int ncon = 0;
  const float4* R = G.result + ((size_t)env * kWorkMax) * 4;
  for (int w0 = 0; w0 < nwork; w0 += kGroup) {
    const bool have = w0 + lane < nwork;
    ConOut co0, co1;
    co0.dist = 0.f; co0.pos = {0.f, 0.f, 0.f}; co0.n = {0.f, 0.f, 1.f}; co1 = co0;
    int n = 0, p = 0;
    V3 hint = {0.f, 0.f, 0.f};
    if (have) {
      const float4 a = R[4 * (w0 + lane)], b = R[4 * (w0 + lane) + 1], c = R[4 * (w0 + lane) + 2], d = R[4 * (w0 + lane) + 3];
      co0.dist = a.x; co0.pos = {a.y, a.z, a.w}; co0.n = {b.x, b.y, b.z}; n = __float_as_int(b.w);
      co1.dist = c.x; co1.pos = {c.y, c.z, c.w}; co1.n = {d.x, d.y, d.z}; p = __float_as_int(d.w);
      if (n >= 1) {  // the frame hint of a plane-capsule pair: the capsule's axis
        const float4 c0 = M.crec[3 * (size_t)p];
        const int t1 = __float_as_int(c0.z) & 255, t2 = (__float_as_int(c0.z) >> 8) & 255;
        if (t1 == 0 && t2 == 3) hint = ld3(s_gaxis + 3 * __float_as_int(c0.y));
      }
    }
    append_contacts<NC>(lane, s_con, ncon, n, co0, co1, hint, p);
  }
  if (ncon > NC) { status |= (1 << 1); ncon = NC; }
  return ncon;
}

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
template <int SOLVER, int NDENSE, int COLL = 0, int NG = 1, int DEFER = 0, int SMALL = 0, int LEAN = 0, int SIZED = 0>
__device__ __forceinline__ void step_body(const DevModel* Mp, const BatchPtrs& P, int nsteps_in, int env_fixed = -1, int ring = -1) {
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
  static_assert(SIZED == 0 || (NG == 1 && SMALL == 0 && ((NDENSE == 28 && (COLL == 0 || SOLVER == 0)) || (NDENSE == 20 && COLL == 1 && SOLVER == 2))), "the size-specialised instantiations: the humanoid (classic or variant-1 layout) and the robot (Newton, variant-1 layout)");
  static_assert(NG == 1 || SOLVER == 2 || (COLL == 1 && NG == kPgsGroups && DEFER == 0), "PGS on more than one row group: the general variant's kPgsGroups instantiation");
  static_assert(SMALL == 0 || (SOLVER == 0 && NDENSE <= 28 && COLL == 0 && NG == 1 && DEFER == 0), "the small instantiation: classic PGS kernel of dense order <= 28");
  constexpr int kNR = SMALL ? kSmallNefcMax : (NG == 1 ? kNefcMax : 64 * NG);  // row capacity of this instantiation
  constexpr int kNC = SMALL ? kSmallNconMax : (NG == 1 ? kNconMax : kBigNconMax);  // contact capacity
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
  bool coherent = env_fixed >= 0;  // (the slow lane's kernel; the small kernel sets it for an env it takes back from the slow lane)
  if (env_fixed < 0 && (int)blockIdx.x >= P.nblk) return;
  const int slot = P.blk0 + (int)blockIdx.x;
  const int env = env_fixed >= 0 ? env_fixed : (P.order ? P.order[slot] : slot);  // (env_fixed: the slow lane's kernel names the env)
  if (P_env_mask && !P_env_mask[env]) return;  // masked stepping (hb_env_reset's settle step)
  if constexpr (COLL != 0 && DEFER == 0) {
    if (P.stage.rerun) {  // second pass of a staged step: only the envs the fast pass deferred
      const int d = P.stage.defer[env];
      if (!d) return;
      if (lane0 == 0) P.stage.defer[env] = 0;
    }
  }
  // two-lane stepping (BatchPtrs::lane): the small kernel leaves the slow-lane envs to hb_step_slow_kernel
  if constexpr (SMALL != 0) {
    // the first block of a small launch tells the slow lane that the GPU has got to this call, and with which controls
    if (blockIdx.x == 0 && lane0 == 0) {
      const int r = P.lane_tag % kLaneRing;
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(P.lane_ring->ctrl + r), (unsigned long long)(uintptr_t)P.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(P.lane_ring->t0 + r, P.t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(P.lane_ring->mode + r, P.ctrl_mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(P.lane_ring->released + P.lane_seg, P.lane_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int slow = P.lane[env];
    if (slow) {
      // A slow env comes back to the fast lane at the first call at which its slow lane has caught up: lane_done[e] == lane_tag - 1 means
      // every earlier step is complete, and the compare-and-swap to -lane_tag CLAIMS this call's step (the slow lane claims its steps the
      // same way, so exactly one of the two steps the env for this call).  Otherwise the env stays slow; at the first call of a window
      // (lane_mode 2: a new list) it is carried over into the new list.
      int got = 0;
      if (lane0 == 0) {
        int expect = P.lane_tag - 1;
        got = __hip_atomic_compare_exchange_strong(P.lane_done + env, &expect, -P.lane_tag, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
      }
      got = uniform(got);
      if (!got) {
        if (P.lane_mode == 2 && lane0 == 0 && P.lane_win[env] != P.lane_wid[P.lane_seg]) {
          P.lane_win[env] = P.lane_wid[P.lane_seg];
          P.lane_list[P.lane_par[P.lane_seg] * P.n_env_total + P.lane_lo[P.lane_seg] + atomicAdd(P.lane_count + 4 * P.lane_par[P.lane_seg] + P.lane_seg, 1)] = env;
        }
        return;
      }
      coherent = true;  // the slow lane's stores of this env's state, not a stale cache line
      if (lane0 == 0) P.lane[env] = 0;
    }
  }
  // an env-step that overflows the small instantiation (nothing has been written at that point): to the slow lane
  auto to_slow_lane = [&]() {
    if (lane0 == 0) {
      P.lane_done[env] = P.lane_tag - 1;  // (the fast lane has completed every step before this one)
      P.lane[env] = P.lane_tag;
      if (P.lane_win[env] != P.lane_wid[P.lane_seg]) {  // (an env that left and re-entered the slow lane inside a window is already listed)
        P.lane_win[env] = P.lane_wid[P.lane_seg];
        P.lane_list[P.lane_par[P.lane_seg] * P.n_env_total + P.lane_lo[P.lane_seg] + atomicAdd(P.lane_count + 4 * P.lane_par[P.lane_seg] + P.lane_seg, 1)] = env;
      }
    }
  };
  (void)to_slow_lane;
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
  // (ring >= 0: the slow lane steps this env with the controls of an earlier step call)
  const float* ctrl_src = P.ctrl;
  int ctrl_mode = P.ctrl_mode, ctrl_t0 = P.t0;
  if (ring >= 0) {
    ctrl_src = reinterpret_cast<const float*>((uintptr_t)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(P.lane_ring->ctrl + ring), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    ctrl_mode = __hip_atomic_load(P.lane_ring->mode + ring, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ctrl_t0 = __hip_atomic_load(P.lane_ring->t0 + ring, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (ctrl_mode != 2 && lane < HB_SZ(nu)) ctrl_pf = ctrl_src[(size_t)env * HB_SZ(nu) + lane];
  float* gstate = P.state + (size_t)env * HB_SZ(nstate);
  // (coherent: the state record changes hands between the slow lane's kernel and a small launch running beside it - agent-scope accesses
  // that go past the non-coherent caches, entry by entry; everywhere else ordinary loads and stores)
  auto ld_state = [&](int i) -> float { return coherent ? __hip_atomic_load(gstate + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gstate[i]; };
  auto st_state = [&](int i, float v) { if (coherent) __hip_atomic_store(gstate + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else gstate[i] = v; };
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
      const float* c = ctrl_src + (ctrl_mode == 1 ? (size_t)step * P.n_env * HB_SZ(nu) : 0) + (size_t)env * HB_SZ(nu);
      if (lane < HB_SZ(nu)) s_ctrl[lane] = ctrl_pf;
      for (int i = lane + kGroup; i < HB_SZ(nu); i += kGroup) s_ctrl[i] = c[i];
      if (ctrl_mode == 1 && step + 1 < nsteps && lane < HB_SZ(nu)) ctrl_pf = c[(size_t)P.n_env * HB_SZ(nu) + lane];
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
        if constexpr (SMALL != 0) { to_slow_lane(); return; }  // more contacts than this instantiation holds
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
            if (lane == 0) P.stage.defer[env] = 1;
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
        if constexpr (SMALL != 0) { to_slow_lane(); return; }
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
        if constexpr (SMALL != 0) { to_slow_lane(); return; }
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
      if (!SMALL || (lw & 31) < NDENSE) store_w_rows(s_W, kWs, T, S, lw);  // (the small layout has no room for the padding rows, and nothing reads them)
      gsync();
    }
    if constexpr (NG == 1) {
    {
      const int col = lane & 31, half = lane >> 5;
#pragma unroll
      for (int I = 0; I < 2; I++) {
        if (I == 0 || (!SMALL && nefc >= 32)) {
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
      const bool two = !SMALL && nefc > 32;  // rows 32..62 in use: all four tiles, else only tile (0,0)
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
        if (ra + 4 < kNR) ar[ra + 4] = __uint_as_float(s0.y);  // (the small instantiation has no row 31)
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
    if constexpr (SMALL == 0) { if (lane < kNR) s_force[lane] = rowact ? force : 0.f; }
    gsync();
    HB_STAMP(13);
    // ---------------------------------------------------------------- dual finish: s = sum_i f_i C_i ; qacc = W (y + s)
    const bool want_qfrc = P_qfrc_out != nullptr;
    if constexpr (SMALL != 0) {
      // the forces straight out of their lanes (nefc is uniform): no LDS copy.  Every lane runs the loop - v_readlane reads lanes that a
      // divergent region has switched off, and the value they hold must have been computed there
      const float fz = rowact ? force : 0.f;
      const int kc = lane < kCs ? lane : 0;
      float sacc = 0.f;
      for (int i = 0; i < nefc; i++) sacc += rdlane(fz, i) * s_C[i * cs + kc];
      if (lane < nv) s_v2[lane] = yv[lane] + sacc;  // y + s (nv <= 28: one pass)
    } else {
      for (int k = lane; k < nv; k += kGroup) {
        float sacc = 0.f;
        for (int i = 0; i < nefc; i++) sacc += s_force[i] * s_C[i * cs + k];
        s_v2[k] = yv[k] + sacc;  // y + s
      }
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
          if (lane == 0) P.stage.defer[env] = 1;
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
// the small instantiation (31 rows, 12 contacts: three waves per SIMD); single-step launches only - an overflowing env-step leaves without
// having written anything, and the slow lane (hb_step_kernel, lane_mode 3) steps that env from then on
__attribute__((amdgpu_num_vgpr(112))) __global__ __launch_bounds__(kGroup, 2) void hb_step_lean_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 0, 1, 0, 0, 1>(Mp, P, nsteps); }
// (the single-step lean kernel with the sizes and the LDS layout of the reference's 27-dof humanoid as constants)
__attribute__((amdgpu_num_vgpr(112))) __global__ __launch_bounds__(kGroup, 2) void hb_step_h27_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 0, 1, 0, 0, 1, 1>(Mp, P, nsteps); }
__attribute__((amdgpu_num_vgpr(112))) __global__ __launch_bounds__(kGroup, 2) void hb_step_h27_q_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 0, 1, 0, 0, 2, 1>(Mp, P, nsteps); }
__attribute__((amdgpu_num_vgpr(112))) __global__ __launch_bounds__(kGroup, 2) void hb_step_lean_q_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 0, 1, 0, 0, 2>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 3) void hb_step_small_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 0, 1, 0, 1>(Mp, P, nsteps); }
// The slow lane: a few blocks walk every segment's list of slow envs and step them with the full instantiation.  A slow env-step is
// often the heavy kind (more than 31 rows: up to 50 sweeps over them, about 100 us against the 80 us period of the small launches), and a
// launch lasts as long as its slowest block - so the slow lane is NOT in lock step with the fast one.  A slow env depends on nothing the
// small launches do: lane_done[e] is the tag of the last step completed for it, and a block steps its env through ALL the step calls the
// GPU has got to so far (LaneRing::released, with the controls of those calls) - also calls whose small launch started while it was
// running.  An env that was heavy for a step or two catches up at the pace of its own light steps, and the small kernel takes it back at
// the first call that finds it caught up; nothing ever waits for the slow lane but a join.
// (A list entry a LATER small launch is writing beside this kernel shows as -1 or not at all: the next launch takes it.)
__global__ __launch_bounds__(kGroup, 2) void hb_step_slow_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) {
  const int seg = blockIdx.y;
  const int* list = P.lane_list + P.lane_par[seg] * P.n_env_total + P.lane_lo[seg];
  const int count = min(uniform(P.lane_count[4 * P.lane_par[seg] + seg]), P.lane_n[seg]);
  for (int i = blockIdx.x; i < count; i += gridDim.x) {
    const int e = uniform(list[i]);
    if (e < 0) continue;
    if (uniform(P.lane[e]) == 0) continue;
    // step after step, each CLAIMED by compare-and-swap on lane_done[e] (t - 1 -> -t) and published when complete (-t -> t): the small
    // kernel claims a step the same way when it takes the env back - whoever loses a claim leaves the env to the winner
    for (;;) {
      // (relaxed agent-scope accesses: they go past the non-coherent caches one word at a time; an acquire / release FENCE at agent
      // scope would invalidate / write back the whole L2 of the XCD under the small launches' feet - measured: 250 instead of 90 us per step)
      const int released = uniform(__hip_atomic_load(P.lane_ring->released + seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      int done = uniform(__hip_atomic_load(P.lane_done + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (done < 0 || done >= released || uniform(P.lane[e]) == 0) break;
      int won = 0;
      if (threadIdx.x == 0) {
        int expect = done;
        won = __hip_atomic_compare_exchange_strong(P.lane_done + e, &expect, -(done + 1), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
      }
      if (!uniform(won)) break;
      if (threadIdx.x == 0) atomicAdd(&P.lane_ring->slow_steps, 1);
      step_body<0, 28>(Mp, P, nsteps, e, (done + 1) % kLaneRing);
      gsync();
      // every lane's (write-through, agent-scope) stores of the new state have left the wave before the flag does
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) __hip_atomic_store(P.lane_done + e, done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__global__ __launch_bounds__(kGroup, 2) void hb_step32_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 32>(Mp, P, nsteps); }
// General instantiations (mesh hulls, height-field prisms, condim 4 / 6): PGS on 63 rows (configs[4]: the 27-dof humanoid on terrain),
// Newton on 256 rows (the reference's own robot, simulation/assets/world.xml: 18 dofs -> dense order 20; up to 28 dofs)
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, 1>(Mp, P, nsteps); }
// PGS on kPgsNefcMax rows (AR in LDS: one env per CU) for condim 4 / 6 models, and the one-group fast pass that defers to it (variant 3)
__global__ __launch_bounds__(kGroup, 1) void hb_step_gen_big_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, kPgsGroups>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_fast1_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_fast_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, 1, 2>(Mp, P, nsteps); }  // staged step, fast pass
__global__ __launch_bounds__(kGroup, 1) void hb_step_newton_big20_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 20, 1, kBigGroups>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 1) void hb_step_newton_big28_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28, 1, kBigGroups>(Mp, P, nsteps); }
// fast pass of a variant-2 model's staged step: Newton on one row group, general collision results, deferring what does not fit
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton_gen20_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 20, 1, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton_gen28_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28, 1, 1, 1>(Mp, P, nsteps); }
// Newton instantiations: dense order 28 (nv <= 28: the 27-dof humanoid) and 32
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton28_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28>(Mp, P, nsteps); }
// lean instantiations (step_body's LEAN: no optional inputs / outputs in the launch) of the kernels the plain step API spends its time in
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton28_lean_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28, 0, 1, 0, 0, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton28_h27_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28, 0, 1, 0, 0, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton28_lean_q_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 28, 0, 1, 0, 0, 2>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_fast_lean_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, 1, 2, 0, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton_gen20_team_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 20, 1, 1, 1, 0, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_gen_fast_h27_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<0, 28, 1, 1, 2, 0, 1, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton_gen20_lean_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 20, 1, 1, 1, 0, 1>(Mp, P, nsteps); }
__global__ __launch_bounds__(kGroup, 2) void hb_step_newton32_kernel(const DevModel* Mp, const BatchPtrs P, int nsteps) { step_body<2, 32>(Mp, P, nsteps); }

// ---- staged step of the general variants: poses + work lists, then the narrowphase, each in a kernel of its own ------------------
// hb_pose_kernel: one wave per env.  The state checks and mj_kinematics of step_body, statement for statement (the step kernel
// repeats them: a pose costs less to recompute than to hand over), the geoms' world poses, broadphase and work items.
__global__ __launch_bounds__(kGroup, 4) void hb_pose_kernel(const DevModel* Mp, const BatchPtrs P) {
  DevModelRef M = *(const DevModel HB_CONST*)(uintptr_t)Mp;
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= P.nblk) return;
  const int env = P.blk0 + (int)blockIdx.x;
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
// longest search and the divergence between its lanes, not by how many lanes it has.
template <int MESH>
__device__ __forceinline__ void narrow_body(const DevModel* Mp, const BatchPtrs& P) {
  DevModelRef M = *(const DevModel HB_CONST*)(uintptr_t)Mp;
  const int lane = threadIdx.x;
  // heavy first (BatchPtrs::order2: the envs of this launch sorted by the time their wave took in an earlier step): the launch ends
  // when its slowest wave does, and a slow wave that starts in the last round ends late
  const int chunk = (int)blockIdx.x / P.nblk, slot = P.blk0 + (int)blockIdx.x % P.nblk;
  const int env = P.order2 ? P.order2[slot] : slot;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  const int n1 = P.stage.nsearch[2 * env], n2 = P.stage.nsearch[2 * env + 1];
  const int j = chunk * kGroup + lane;
  if (chunk * kGroup >= n1 + n2) { if (chunk == 0 && lane == 0) P.counts[kCountStride * (size_t)env + 7] = 0; return; }
  const bool have = j < n1 + n2;
  int4 it = {env, 0, 1 << 16, 0};
  if (have) it = P.stage.item[(size_t)env * kWorkMax + (j < n1 ? j : kWorkMax - 1 - (j - n1))];
  const int ng = M.ngeom;
  const DomainLayout DL = domain_layout(M.nbody, M.nv, M.nlimcand, M.nu, M.nhfielddata);
  const int p = it.y & 0xffff, w = it.y >> 16;
  const float* g = P.stage.geom + (size_t)env * ng * 10;
  const float* hdata = P.dr ? P.dr + (size_t)env * P.dr_stride + DL.o_hfield : (const float*)M.hfield_data;
  ConOut co0, co1;
  int n;
  V3 hint;
  eval_work_item<2, MESH>(M, hdata, have, p, it.z & 0xffff, it.w & 0xffff, it.w >> 16, it.z >> 16, 0.f, g, g + 3 * ng, g + 6 * ng, co0, co1, n, hint);
  if (have) {
    float4* R = P.stage.result + ((size_t)env * kWorkMax + w) * 4;
    R[0] = {co0.dist, co0.pos.x, co0.pos.y, co0.pos.z};
    R[1] = {co0.n.x, co0.n.y, co0.n.z, __int_as_float(n)};
    R[2] = {co1.dist, co1.pos.x, co1.pos.y, co1.pos.z};
    R[3] = {co1.n.x, co1.n.y, co1.n.z, __int_as_float(p)};
  }
  if (chunk == 0 && lane == 0) P.counts[kCountStride * (size_t)env + 7] = (int)min(255ull, (__builtin_amdgcn_s_memtime() - t_begin) >> 10);
}
__global__ __launch_bounds__(kGroup, 2) void hb_narrow_kernel(const DevModel* Mp, const BatchPtrs P) { narrow_body<1>(Mp, P); }
// a model without mesh geoms (configs[4]: capsules and spheres over the height field's prisms): no hull climb in the kernel
__global__ __launch_bounds__(kGroup, 3) void hb_narrow_prim_kernel(const DevModel* Mp, const BatchPtrs P) { narrow_body<0>(Mp, P); }

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
                              uint8_t* truncated, const uint8_t* mask, int observe, const DomainRand D, float* dr, int dr_stride, int n_env, int env_offset) {
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
  __syncthreads();  // (every lane has read the old state and the old episode number)
  if (reset) {
    // CPUEnv.reset for this env; the perturbation index advances with the episode count
    ep += 1;
    reset_state(M, ls, qpos_src, cfg.reset_perturb, env_offset + e, ep, cfg.reset_quat_perturb, l, kEnvLanes);
    for (int i = l; i < M.nu; i += kEnvLanes) { prev[(size_t)e * M.nu + i] = 0.f; latest[(size_t)e * M.nu + i] = 0.f; }
    if (l == 0) {
      episode[e] = ep;
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

// ------------------------------------------------------------------------------------------
// host-callable launchers (declared in hb_launch.hpp)
}  // namespace hb

#include "hb_launch.hpp"

namespace hb {

// the lean instantiations apply when the launch has none of the optional inputs / outputs (HB_LEAN=0: never)
static bool lean_launch(const BatchPtrs& P, bool with_qfrc = false) {
  static const bool lean_on = !(getenv("HB_LEAN") && atoi(getenv("HB_LEAN")) == 0);
  return lean_on && (P.lean_ok & 1) && !P.xfrc && (with_qfrc || !P.qfrc_out) && !P.sensor_out && !P.qpos_out && !P.qvel_out && !P.diag_qacc && !P.diag_force && !P.diag_contact && !P.dr && !P.env_mask &&
         P.integrate && !P.lane;
}
static hipError_t launch_step_kernel(const DevModel* M_dev, int variant, int solver, int nv, size_t shmem, const BatchPtrs& P, int nsteps, hipStream_t stream) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  if (variant == 2 && nv <= 20) hipLaunchKernelGGL(hb_step_newton_big20_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (variant == 2) hipLaunchKernelGGL(hb_step_newton_big28_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (variant == 1) hipLaunchKernelGGL(hb_step_gen_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (variant == 3) hipLaunchKernelGGL(hb_step_gen_big_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (solver == 2 && nv <= 28) {
    if (nsteps == 1 && lean_launch(P) && (P.lean_ok & 2)) hipLaunchKernelGGL(hb_step_newton28_h27_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (nsteps == 1 && lean_launch(P)) hipLaunchKernelGGL(hb_step_newton28_lean_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (lean_launch(P, true)) hipLaunchKernelGGL(hb_step_newton28_lean_q_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else hipLaunchKernelGGL(hb_step_newton28_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  }
  else if (solver == 2) hipLaunchKernelGGL(hb_step_newton32_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  else if (nv <= 28) {
    if (nsteps == 1 && lean_launch(P) && (P.lean_ok & 2)) hipLaunchKernelGGL(hb_step_h27_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (nsteps == 1 && lean_launch(P)) hipLaunchKernelGGL(hb_step_lean_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (lean_launch(P, true) && (P.lean_ok & 2)) hipLaunchKernelGGL(hb_step_h27_q_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else if (lean_launch(P, true)) hipLaunchKernelGGL(hb_step_lean_q_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
    else hipLaunchKernelGGL(hb_step_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  }
  else hipLaunchKernelGGL(hb_step32_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, P, nsteps);
  return hipGetLastError();
}

// One step launch of the classic variant covers all nsteps.  A general variant with stage buffers runs every step as three launches
// on the same stream: poses + work items, narrowphase (a small kernel at 2-4x the step kernel's occupancy: its time is chains of
// dependent loads along the hulls' edge graphs), then the step kernel, which appends the results instead of colliding.
hipError_t launch_step(const DevModel* M_dev, int variant, int solver, int nv, int lds_floats, const BatchPtrs& P, int nsteps, hipStream_t stream) {
  const size_t shmem = (size_t)lds_floats * sizeof(float);
  if (variant == 0 || !P.stage.result) return launch_step_kernel(M_dev, variant, solver, nv, shmem, P, nsteps, stream);
  for (int t = 0; t < nsteps; t++) {
    BatchPtrs Q = P;
    Q.t0 = P.t0 + t;
    if (P.ctrl_mode == 1) Q.ctrl = P.ctrl + (size_t)t * P.n_env * P.stage.nu;
    if (P.qpos_out) Q.qpos_out = P.qpos_out + (size_t)t * P.n_env * P.stage.nq;
    if (P.qvel_out) Q.qvel_out = P.qvel_out + (size_t)t * P.n_env * P.stage.nv;
    if (P.sensor_out) Q.sensor_out = P.sensor_out + (size_t)t * P.n_env * P.sensor_stride;
    (void)hipGetLastError();
    hipLaunchKernelGGL(hb_pose_kernel, dim3(P.nblk), dim3(kGroup), (size_t)P.stage.pose_lds, stream, M_dev, Q);
    if (Q.stage.no_mesh) hipLaunchKernelGGL(hb_narrow_prim_kernel, dim3(P.nblk * (kWorkMax / kGroup)), dim3(kGroup), 0, stream, M_dev, Q);
    else hipLaunchKernelGGL(hb_narrow_kernel, dim3(P.nblk * (kWorkMax / kGroup)), dim3(kGroup), 0, stream, M_dev, Q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (variant == 1 && Q.stage.defer) {
      // the step kernel without the portal-search code; the full one then takes the (rare) env-steps whose qacc came out bad
      if (lean_launch(Q) && (Q.lean_ok & 2)) hipLaunchKernelGGL(hb_step_gen_fast_h27_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, Q, 1);
      else if (lean_launch(Q)) hipLaunchKernelGGL(hb_step_gen_fast_lean_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, Q, 1);
      else hipLaunchKernelGGL(hb_step_gen_fast_kernel, dim3(P.nblk), dim3(kGroup), shmem, stream, M_dev, Q, 1);
      e = hipGetLastError();
      if (e != hipSuccess) return e;
      Q.stage.rerun = 1;
    } else if (variant == 3 && Q.stage.dm_fast) {
      // PGS: the one-group kernel (63 rows, 24 contacts, two waves per SIMD) first; the kPgsNefcMax-row kernel then steps what it defers
      hipLaunchKernelGGL(hb_step_gen_fast1_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      e = hipGetLastError();
      if (e != hipSuccess) return e;
      Q.stage.rerun = 1;
    } else if (variant == 2 && Q.stage.dm_fast) {
      // most env-steps fit the one-group Newton instantiation (two waves per SIMD); the four-group kernel then steps the rest
      if (nv <= 20 && lean_launch(Q) && (Q.lean_ok & 4)) hipLaunchKernelGGL(hb_step_newton_gen20_team_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      else if (nv <= 20 && lean_launch(Q)) hipLaunchKernelGGL(hb_step_newton_gen20_lean_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      else if (nv <= 20) hipLaunchKernelGGL(hb_step_newton_gen20_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      else hipLaunchKernelGGL(hb_step_newton_gen28_kernel, dim3(P.nblk), dim3(kGroup), (size_t)Q.stage.fast_lds, stream, Q.stage.dm_fast, Q, 1);
      e = hipGetLastError();
      if (e != hipSuccess) return e;
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
  return hipSuccess;
}
hipError_t launch_step_slow(const DevModel* M_dev, int lds_floats, const BatchPtrs& P, int blocks, int nseg, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(hb_step_slow_kernel, dim3(blocks, nseg), dim3(kGroup), (size_t)lds_floats * sizeof(float), stream, M_dev, P, 1);
  return hipGetLastError();
}
hipError_t launch_step_small(const DevModel* M_small, int lds_floats, const BatchPtrs& P, hipStream_t stream) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(hb_step_small_kernel, dim3(P.nblk), dim3(kGroup), (size_t)lds_floats * sizeof(float), stream, M_small, P, 1);
  return hipGetLastError();
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
                      const uint8_t* mask, int observe, const DomainRand& D, float* dr, int dr_stride, int n_env, int env_offset, hipStream_t stream) {
  (void)hipGetLastError();  // the result below must be this launch's, not an older call's sticky error
  const int per_block = 256 / kEnvLanes;
  hipLaunchKernelGGL(hb_env_kernel, dim3((n_env + per_block - 1) / per_block), dim3(256), (size_t)per_block * ((M.nstate + 3) & ~3) * sizeof(float), stream, M, cfg, R, S, state, qfrc, counts, prev, latest, qpos_src, episode, status, obs,
                     reward, terminated, truncated, mask, observe, D, dr, dr_stride, n_env, env_offset);
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
