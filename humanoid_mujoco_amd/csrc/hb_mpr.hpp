// hb_mpr.hpp — convex narrowphase on the device (fp32): libccd's Minkowski Portal Refinement as MuJoCo uses it for mesh
// geoms (through their convex hulls) and for every geom against a height-field prism (engine_collision_convex.c: mjc_Convex,
// mjc_ConvexHField; mujoco.h:355 mj_collision).  Mirrors oracle/mjstep_oracle.c (mpr_penetration, ccd_support, fix_normal)
// statement for statement so that the discrete decisions of the portal search agree wherever fp32 allows.
// One LANE runs one (pair, prism) test: the loops below are per lane and divergent; a mesh's support function is an
// exhaustive sweep over its hull vertices (16-byte records in the model tables, served by L1 / L2).
// Included by hb_kernels.hip only (uses its V3 / Q4 helpers).
#pragma once

namespace hb {

#define HB_CCD_EPS 1.1920929e-7f  // FLT_EPSILON: libccd's CCD_EPS at this precision
__device__ __forceinline__ bool ccd_is_zero(float x) { return fabsf(x) < HB_CCD_EPS; }
__device__ __forceinline__ bool ccd_eq(float a, float b) {
  const float ab = fabsf(a - b);
  if (ab < HB_CCD_EPS) return true;
  a = fabsf(a); b = fabsf(b);
  return b > a ? ab < HB_CCD_EPS * b : ab < HB_CCD_EPS * a;
}

// a convex object in the frame the test runs in: a geom (sphere 2, capsule 3, mesh hull 7) or a height-field prism (-1)
struct CObj {
  int type;
  V3 pos;
  float mat[9];  // row-major rotation of the geom frame
  float r, h;    // size[0], size[1]
  const float4 HB_CONST* vert;  // mesh: hull vertices in the geom frame
  int nvert;
  float margin;
  V3 p0, p1, p2, p3, p4, p5;  // prism: bottom triangle 0..2, top triangle 3..5
};

__device__ __forceinline__ V3 ccd_center(const CObj& o) {
  if (o.type < 0) return (o.p0 + o.p1 + o.p2 + o.p3 + o.p4 + o.p5) * (1.f / 6.f);
  return o.pos;
}
// the point farthest along dir (unit)
__device__ __forceinline__ V3 ccd_support(const CObj& o, V3 dir) {
  if (o.type < 0) {
    V3 best = o.p0;
    float bd = dot(o.p0, dir), v;
    v = dot(o.p1, dir); if (v > bd) { bd = v; best = o.p1; }
    v = dot(o.p2, dir); if (v > bd) { bd = v; best = o.p2; }
    v = dot(o.p3, dir); if (v > bd) { bd = v; best = o.p3; }
    v = dot(o.p4, dir); if (v > bd) { bd = v; best = o.p4; }
    v = dot(o.p5, dir); if (v > bd) { bd = v; best = o.p5; }
    return best;
  }
  const V3 ld = {o.mat[0] * dir.x + o.mat[3] * dir.y + o.mat[6] * dir.z, o.mat[1] * dir.x + o.mat[4] * dir.y + o.mat[7] * dir.z,
                 o.mat[2] * dir.x + o.mat[5] * dir.y + o.mat[8] * dir.z};  // mat' dir
  V3 res;
  if (o.type == 2) res = ld * o.r;
  else if (o.type == 3) { res = ld * o.r; res.z += ld.z >= 0.f ? o.h : -o.h; }
  else {
    float bd = -3.0e38f;
    res = {0.f, 0.f, 0.f};
    for (int i = 0; i < o.nvert; i++) {
      const float4 q = o.vert[i];
      const float v = q.x * ld.x + q.y * ld.y + q.z * ld.z;
      if (v > bd) { bd = v; res = {q.x, q.y, q.z}; }
    }
  }
  res = res + ld * o.margin;
  return mrot(o.mat, res) + o.pos;
}

struct CSup { V3 v, v1; };  // a point of the Minkowski difference obj1 - obj2 and its witness on obj1 (the one on obj2 is v1 - v)
__device__ __forceinline__ CSup mpr_support(const CObj& o1, const CObj& o2, V3 dir) {
  CSup s;
  s.v1 = ccd_support(o1, dir);
  const V3 w2 = ccd_support(o2, dir * -1.f);
  s.v = s.v1 - w2;
  return s;
}
__device__ __forceinline__ V3 mpr_portal_dir(const CSup& P1, const CSup& P2, const CSup& P3) { return normalized(cross(P2.v - P1.v, P3.v - P1.v)); }
__device__ __forceinline__ bool mpr_reach_tolerance(const CSup& P1, const CSup& P2, const CSup& P3, const CSup& v4, V3 dir, float tol) {
  const float dv4 = dot(v4.v, dir);
  const float d = fminf(fminf(dv4 - dot(P1.v, dir), dv4 - dot(P2.v, dir)), dv4 - dot(P3.v, dir));
  return ccd_eq(d, tol) || d < tol;
}
__device__ __forceinline__ void mpr_expand_portal(const CSup& P0, CSup& P1, CSup& P2, CSup& P3, const CSup& v4) {
  const V3 v4v0 = cross(v4.v, P0.v);
  if (dot(P1.v, v4v0) > 0.f) { if (dot(P2.v, v4v0) > 0.f) P1 = v4; else P3 = v4; }
  else { if (dot(P3.v, v4v0) > 0.f) P2 = v4; else P1 = v4; }
}

// ccdMPRPenetration: true (and depth, dir from obj1 into obj2, pos) when the objects intersect
__device__ __forceinline__ bool mpr_penetration(const CObj& o1, const CObj& o2, int max_iterations, float tolerance, float& depth, V3& pdir, V3& pos) {
  CSup P0, P1, P2, P3, v4;
  const V3 origin = {0.f, 0.f, 0.f};
  // ---- discoverPortal
  P0.v1 = ccd_center(o1);
  P0.v = P0.v1 - ccd_center(o2);
  if (ccd_eq(P0.v.x, 0.f) && ccd_eq(P0.v.y, 0.f) && ccd_eq(P0.v.z, 0.f)) P0.v.x += HB_CCD_EPS * 10.f;
  V3 dir = normalized(P0.v * -1.f);
  P1 = mpr_support(o1, o2, dir);
  float dt = dot(P1.v, dir);
  if (ccd_is_zero(dt) || dt < 0.f) return false;
  dir = cross(P0.v, P1.v);
  if (ccd_is_zero(dot(dir, dir))) {
    pos = P1.v1 - P1.v * 0.5f;  // 0.5 (v1 + v2), v2 = v1 - v
    if (ccd_eq(P1.v.x, 0.f) && ccd_eq(P1.v.y, 0.f) && ccd_eq(P1.v.z, 0.f)) { depth = 0.f; pdir = origin; return true; }  // touching
    float n;
    pdir = normalized(P1.v, &n);
    depth = n;
    return true;
  }
  dir = normalized(dir);
  P2 = mpr_support(o1, o2, dir);
  dt = dot(P2.v, dir);
  if (ccd_is_zero(dt) || dt < 0.f) return false;
  dir = normalized(cross(P1.v - P0.v, P2.v - P0.v));
  if (dot(dir, P0.v) > 0.f) { const CSup t = P1; P1 = P2; P2 = t; dir = dir * -1.f; }
  for (int guard = 0;; guard++) {
    if (guard > 1000) return false;
    P3 = mpr_support(o1, o2, dir);
    dt = dot(P3.v, dir);
    if (ccd_is_zero(dt) || dt < 0.f) return false;
    bool cont = false;
    dt = dot(cross(P1.v, P3.v), P0.v);
    if (dt < 0.f && !ccd_is_zero(dt)) { P2 = P3; cont = true; }
    if (!cont) {
      dt = dot(cross(P3.v, P2.v), P0.v);
      if (dt < 0.f && !ccd_is_zero(dt)) { P1 = P3; cont = true; }
    }
    if (!cont) break;
    dir = normalized(cross(P1.v - P0.v, P2.v - P0.v));
  }
  // ---- refinePortal
  for (int guard = 0;; guard++) {
    if (guard > 1000) return false;
    dir = mpr_portal_dir(P1, P2, P3);
    dt = dot(dir, P1.v);
    if (ccd_is_zero(dt) || dt > 0.f) break;
    v4 = mpr_support(o1, o2, dir);
    dt = dot(v4.v, dir);
    if (!(ccd_is_zero(dt) || dt > 0.f) || mpr_reach_tolerance(P1, P2, P3, v4, dir, tolerance)) return false;
    mpr_expand_portal(P0, P1, P2, P3, v4);
  }
  // ---- findPenetr
  for (int it = 0;; it++) {
    dir = mpr_portal_dir(P1, P2, P3);
    v4 = mpr_support(o1, o2, dir);
    if (mpr_reach_tolerance(P1, P2, P3, v4, dir, tolerance) || it > max_iterations) {
      const V3 w = closest_on_triangle(origin, P1.v, P2.v, P3.v);
      depth = sqrtf(dot(w, w));
      if (ccd_is_zero(depth)) pdir = origin;
      else pdir = normalized(w);
      // findPos: barycentric coordinates of the origin in the portal tetrahedron
      dir = mpr_portal_dir(P1, P2, P3);
      float b0 = dot(cross(P1.v, P2.v), P3.v), b1 = dot(cross(P3.v, P2.v), P0.v), b2 = dot(cross(P0.v, P1.v), P3.v), b3 = dot(cross(P2.v, P1.v), P0.v);
      float sum = b0 + b1 + b2 + b3;
      if (ccd_is_zero(sum) || sum < 0.f) {
        b0 = 0.f;
        b1 = dot(cross(P2.v, P3.v), dir); b2 = dot(cross(P3.v, P1.v), dir); b3 = dot(cross(P1.v, P2.v), dir);
        sum = b1 + b2 + b3;
      }
      const float inv = 1.f / sum;
      // 0.5 (p1 + p2) with p2_i = v1_i - v_i
      const V3 p1 = P0.v1 * b0 + P1.v1 * b1 + P2.v1 * b2 + P3.v1 * b3;
      const V3 pv = P0.v * b0 + P1.v * b1 + P2.v * b2 + P3.v * b3;
      pos = (p1 - pv * 0.5f) * inv;
      return true;
    }
    mpr_expand_portal(P0, P1, P2, P3, v4);
  }
}

// mjc_fixNormal: a sphere or capsule supplies its own surface normal at the contact point (oracle: fix_normal)
__device__ __forceinline__ bool analytic_normal(int type, V3 gpos, const float* gmat, float h, V3 cpos, V3& n_out) {
  if (type != 2 && type != 3) return false;
  const V3 dif = cpos - gpos;
  V3 lp = {gmat[0] * dif.x + gmat[3] * dif.y + gmat[6] * dif.z, gmat[1] * dif.x + gmat[4] * dif.y + gmat[7] * dif.z, gmat[2] * dif.x + gmat[5] * dif.y + gmat[8] * dif.z};
  if (type == 3) lp.z = lp.z > h ? lp.z - h : (lp.z < -h ? lp.z + h : 0.f);
  float n;
  lp = normalized(lp, &n);
  if (n < HB_MINVAL) return false;
  n_out = mrot(gmat, lp);
  return true;
}

}  // namespace hb
