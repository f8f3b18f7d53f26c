// hb_kcommon.hpp - device helpers shared by the kernel translation units (hb_step.hip, hb_step_duo.hip, hb_narrow.hip, hb_env.hip):
// wave idioms, 3-vector / quaternion / spatial algebra, the primitive colliders, constraint impedance, counter-based random numbers and
// the dense eliminations on the matrix cores.  Everything is __forceinline__: no device function crosses a translation unit.
#pragma once
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

// Two independent problems side by side (two envs per wave, hb_step_duo.hip): the pivot chains of the two eliminations do not depend on
// each other, so pair b of the one is issued into the latencies of pair b of the other (readlane - rcp - fma - swap - MFMA - MFMA).  Per
// problem the arithmetic is sym_factor_mfma's / sym_solve_mfma's, statement for statement.
template <int NP>
__device__ __forceinline__ void sym_factor_mfma_x2(f32x16 Xa, f32x16 Xb, f32x16& Ta, f32x16& Sa, f32x16& Tb, f32x16& Sb, int lane) {
  const int li = lane & 31, half = lane >> 5;
  const bool up = half != 0;
  const int q = li - 4 * half;
#pragma unroll
  for (int r = 0; r < 16; r++) { Ta[r] = q == crow(r) ? 1.f : 0.f; Sa[r] = 1.f; Tb[r] = Ta[r]; Sb[r] = 1.f; }
  const float lik = (float)(li - half);
#pragma unroll
  for (int b = 0; b < NP; b++) {
    const int k0 = 2 * b, r0 = (k0 & 3) + 4 * (k0 >> 3), h0 = (k0 >> 2) & 1, L0 = 32 * h0;
    const float below = __builtin_amdgcn_fmed3f(lik - (float)k0, 0.f, 1.f);
#define HB_FACTOR_PAIR(X, T, S)                                                                                                   \
    {                                                                                                                             \
      const float rowa = X[r0], rowb = X[r0 + 1], ta = T[r0], tb = T[r0 + 1];                                                     \
      const float d0 = fmaxf(rdlane(rowa, L0 + k0), HB_MINVAL);                                                                   \
      const float inv0 = __builtin_amdgcn_rcpf(d0);                                                                               \
      const float m = rdlane(rowb, L0 + k0) * inv0;                                                                               \
      const float rowb1 = __builtin_fmaf(-m, rowa, rowb), tb1 = __builtin_fmaf(-m, ta, tb);                                       \
      const float d1 = fmaxf(rdlane(rowb1, L0 + k0 + 1), HB_MINVAL);                                                              \
      const float inv1 = __builtin_amdgcn_rcpf(d1);                                                                               \
      const u32x2 sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(rowa), __float_as_uint(rowb1), false, false);            \
      const u32x2 st = __builtin_amdgcn_permlane32_swap(__float_as_uint(ta), __float_as_uint(tb1), false, false);                \
      const float vb = __uint_as_float(h0 ? sx.y : sx.x), vt = __uint_as_float(h0 ? st.y : st.x);                                 \
      const float va = -(vb * (up ? inv1 : inv0)) * below;                                                                        \
      X = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, X, 0, 0, 0);                                                               \
      T = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vt, T, 0, 0, 0);                                                               \
      if (half == h0) { S[r0] = __builtin_amdgcn_rsqf(d0); S[r0 + 1] = __builtin_amdgcn_rsqf(d1); }                               \
    }
    HB_FACTOR_PAIR(Xa, Ta, Sa)
    HB_FACTOR_PAIR(Xb, Tb, Sb)
#undef HB_FACTOR_PAIR
  }
}

template <int NP>
__device__ __forceinline__ void sym_solve_mfma_x2(f32x16 Xa, f32x16 Xb, float ga, float gb, float& xa, float& xb, int lane) {
  const int li = lane & 31, half = lane >> 5;
  const bool up = half != 0;
  {
    const float e31 = li == 31 ? 1.f : 0.f;
    Xa = __builtin_amdgcn_mfma_f32_32x32x2f32(up ? ga : e31, up ? e31 : ga, Xa, 0, 0, 0);
    Xb = __builtin_amdgcn_mfma_f32_32x32x2f32(up ? gb : e31, up ? e31 : gb, Xb, 0, 0, 0);
  }
  f32x16 Ta, Za, Tb, Zb;
  const int q = li - 4 * half;
#pragma unroll
  for (int r = 0; r < 16; r++) { Ta[r] = q == crow(r) ? 1.f : 0.f; Za[r] = 0.f; Tb[r] = Ta[r]; Zb[r] = 0.f; }
  const float lik = (float)(li - half);
#pragma unroll
  for (int b = 0; b < NP; b++) {
    const int k0 = 2 * b, r0 = (k0 & 3) + 4 * (k0 >> 3), h0 = (k0 >> 2) & 1, L0 = 32 * h0;
    const float below = __builtin_amdgcn_fmed3f(lik - (float)k0, 0.f, 1.f);
#define HB_SOLVE_PAIR(X, T, Z)                                                                                                    \
    {                                                                                                                             \
      const float rowa = X[r0], rowb = X[r0 + 1], ta = T[r0], tb = T[r0 + 1];                                                     \
      const float inv0 = __builtin_amdgcn_rcpf(fmaxf(rdlane(rowa, L0 + k0), HB_MINVAL));                                          \
      const float m = rdlane(rowb, L0 + k0) * inv0;                                                                               \
      const float rowb1 = __builtin_fmaf(-m, rowa, rowb), tb1 = __builtin_fmaf(-m, ta, tb);                                       \
      const float inv1 = __builtin_amdgcn_rcpf(fmaxf(rdlane(rowb1, L0 + k0 + 1), HB_MINVAL));                                     \
      const u32x2 sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(rowa), __float_as_uint(rowb1), false, false);            \
      const u32x2 st = __builtin_amdgcn_permlane32_swap(__float_as_uint(ta), __float_as_uint(tb1), false, false);                \
      const float vb = __uint_as_float(h0 ? sx.y : sx.x), vt = __uint_as_float(h0 ? st.y : st.x);                                 \
      const float va = -(vb * (up ? inv1 : inv0)) * below;                                                                        \
      X = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, X, 0, 0, 0);                                                               \
      T = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vt, T, 0, 0, 0);                                                               \
      const float z0 = rdlane(rowa, L0 + 31) * inv0, z1 = rdlane(rowb1, L0 + 31) * inv1;                                          \
      if (half == h0) { Z[r0] = z0; Z[r0 + 1] = z1; }                                                                             \
    }
    HB_SOLVE_PAIR(Xa, Ta, Za)
    HB_SOLVE_PAIR(Xb, Tb, Zb)
#undef HB_SOLVE_PAIR
  }
#define HB_SOLVE_OUT(T, Z, out_)                                                                                                    \
  {                                                                                                                               \
    float p = 0.f, p1 = 0.f;                                                                                                      \
    _Pragma("unroll") for (int r = 0; r < 16; r += 2) { p = __builtin_fmaf(T[r], Z[r], p); p1 = __builtin_fmaf(T[r + 1], Z[r + 1], p1); } \
    p += p1;                                                                                                                      \
    const u32x2 sp = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);                      \
    out_ = p + __uint_as_float(up ? sp.x : sp.y);                                                                                 \
  }
  HB_SOLVE_OUT(Ta, Za, xa)
  HB_SOLVE_OUT(Tb, Zb, xb)
#undef HB_SOLVE_OUT
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

}  // namespace hb
