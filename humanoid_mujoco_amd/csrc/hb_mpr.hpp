// hb_mpr.hpp — convex narrowphase on the device (fp64 arithmetic on fp32 geometry): libccd's Minkowski Portal Refinement as MuJoCo uses it for mesh
// geoms (through their convex hulls) and for every geom against a height-field prism (engine_collision_convex.c: mjc_Convex,
// mjc_ConvexHField; mujoco.h:355 mj_collision).  Mirrors oracle/mjstep_oracle.c (mpr_penetration, ccd_support, fix_normal)
// statement for statement so that the discrete decisions of the portal search agree wherever the fp32 poses allow.
// One LANE runs one (pair, prism) test: the loops below are per lane and divergent; a mesh's support function is a
// climb along the hull's edge graph (16-byte records in the model tables, served by L1 / L2) from a cube map of start vertices.
// Included by hb_kcommon.hpp only (uses its V3 / Q4 helpers).
#pragma once

namespace hb {

// The portal search runs in DOUBLE precision on the device too.  Measured with an fp32 version: against a height-field prism
// (metres wide, a geom centimetres across) sign tests of the portal expansion flip under fp32 rounding, the search then stops on
// a portal far from the surface and reports a penetration of the prism's whole depth (1 m): envs exploded at a rate of 2 % per
// thousand steps on the terrain benchmark.  An fp64 fma issues like an unpacked fp32 one on MI355X (tools/micro/fp64_rates.hip), and
// with the same arithmetic as the oracle the discrete decisions agree as well.  Geometry stays fp32 in registers (CObj) and is widened at use.
#define HB_CCD_EPS 2.220446049250313e-16  // DBL_EPSILON: libccd's CCD_EPS in its double-precision build (what MuJoCo links)
struct V3d { double x, y, z; };
__device__ __forceinline__ V3d operator+(V3d a, V3d b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3d operator-(V3d a, V3d b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3d operator*(V3d a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double dot(V3d a, V3d b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3d cross(V3d a, V3d b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3d widen(V3 a) { return {(double)a.x, (double)a.y, (double)a.z}; }
__device__ __forceinline__ V3 narrow(V3d a) { return {(float)a.x, (float)a.y, (float)a.z}; }
__device__ __forceinline__ V3d normalized(V3d v, double* n_out = nullptr) {
  const double n = sqrt(dot(v, v));
  if (n_out) *n_out = n;
  if (n < 1e-15) return {1.0, 0.0, 0.0};
  const double inv = 1.0 / n;
  return v * inv;
}
__device__ __forceinline__ bool ccd_is_zero(double x) { return fabs(x) < HB_CCD_EPS; }
__device__ __forceinline__ bool ccd_eq(double a, double b) {
  const double ab = fabs(a - b);
  if (ab < HB_CCD_EPS) return true;
  a = fabs(a); b = fabs(b);
  return b > a ? ab < HB_CCD_EPS * b : ab < HB_CCD_EPS * a;
}
// closest point of triangle abc to p (Ericson, Real-Time Collision Detection 5.1.5), double precision
__device__ __forceinline__ V3d closest_on_triangle(V3d p, V3d a, V3d b, V3d c) {
  const V3d ab = b - a, ac = c - a, ap = p - a;
  const double d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.0 && d2 <= 0.0) return a;
  const V3d bp = p - b;
  const double d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.0 && d4 <= d3) return b;
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) return a + ab * (d1 / (d1 - d3));
  const V3d cp = p - c;
  const double d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0.0 && d5 <= d6) return c;
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) return a + ac * (d2 / (d2 - d6));
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) return b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)));
  const double denom = 1.0 / (va + vb + vc);
  return a + ab * (vb * denom) + ac * (vc * denom);
}

// a convex object in the frame the test runs in: a geom (sphere 2, capsule 3, mesh hull 7) or a height-field prism (-1)
struct CObj {
  int type;
  V3 pos;
  float mat[9];  // row-major rotation of the geom frame
  float r, h;    // size[0], size[1]
  // mesh: the hull as an edge graph.  vert = the mesh's cube map of start records (x, y, z, link), nbr = the neighbour records of
  // the whole model: link = first record << 8 | chunks, record = (x, y, z of the neighbour, the neighbour's own link).
  const float4 HB_CONST* vert;
  const float4 HB_CONST* nbr;
  float margin;
  V3 p0, p1, p2, p3, p4, p5;  // prism: bottom triangle 0..2, top triangle 3..5
};

__device__ __forceinline__ V3d ccd_center(const CObj& o) {
  if (o.type < 0) return (widen(o.p0) + widen(o.p1) + widen(o.p2) + widen(o.p3) + widen(o.p4) + widen(o.p5)) * (1.0 / 6.0);
  return widen(o.pos);
}
// ---- mesh support: steepest ascent along the hull's edges (oracle: ccd_support).  Every neighbour of the current vertex is
// evaluated, the best one taken if it is strictly better (ties to the first in the list); on a convex polytope a vertex with no
// better neighbour is a maximiser (exact ties - a direction perpendicular to a flat facet - are broken by a second, generic
// direction: see the oracle), so the vertex a climb ends on does not depend on where it starts.  Every climb starts from the
// mesh's cube map of support vertices (the cell the direction falls in: 1.2 - 1.5 rounds on the reference's hulls, of which the
// last only confirms; from the previous call's vertex it was 2 - 5, from vertex 0 4 - 7: tools/mpr_stats.py).  A round is one batch
// of kMeshChunk independent 16-byte loads per chunk of the vertex's padded neighbour list.
struct Climb { V3d ld; V3 best; int link; double bd; };
// The value of a vertex along the query direction and along the tie direction, each as ONE fixed sequence of fused operations: a
// vertex also appears as padding in its own neighbour list, and a climb only ends if that copy compares EQUAL to the vertex
// (left to the compiler, the two sides of a comparison may be contracted differently and differ in the last bit: a vertex that
// "improves" on itself for ever).
__device__ __forceinline__ double hull_val(float x, float y, float z, V3d ld) { return __builtin_fma((double)x, ld.x, __builtin_fma((double)y, ld.y, (double)z * ld.z)); }
__device__ __forceinline__ double hull_tie(float x, float y, float z) { return __builtin_fma((double)x, 0.41421356237309503, __builtin_fma((double)y, 0.7320508075688772, (double)z)); }
__device__ __forceinline__ V3d local_dir(const CObj& o, V3d dir) {  // mat' dir
  const double m0 = o.mat[0], m1 = o.mat[1], m2 = o.mat[2], m3 = o.mat[3], m4 = o.mat[4], m5 = o.mat[5], m6 = o.mat[6], m7 = o.mat[7], m8 = o.mat[8];
  return {m0 * dir.x + m3 * dir.y + m6 * dir.z, m1 * dir.x + m4 * dir.y + m7 * dir.z, m2 * dir.x + m5 * dir.y + m8 * dir.z};
}
__device__ __forceinline__ V3d world_point(const CObj& o, V3d ld, V3d res) {  // mat (res + ld margin) + pos
  res = res + ld * (double)o.margin;
  const double m0 = o.mat[0], m1 = o.mat[1], m2 = o.mat[2], m3 = o.mat[3], m4 = o.mat[4], m5 = o.mat[5], m6 = o.mat[6], m7 = o.mat[7], m8 = o.mat[8];
  return V3d{m0 * res.x + m1 * res.y + m2 * res.z, m3 * res.x + m4 * res.y + m5 * res.z, m6 * res.x + m7 * res.y + m8 * res.z} + widen(o.pos);
}
__device__ __forceinline__ void climb_start(const CObj& o, Climb& c) {
  const V3d ld = c.ld;
  const double ax = fabs(ld.x), ay = fabs(ld.y), az = fabs(ld.z);
  const int axis = ax >= ay ? (ax >= az ? 0 : 2) : (ay >= az ? 1 : 2);
  const double major = axis == 0 ? ld.x : (axis == 1 ? ld.y : ld.z);
  const float inv = 1.f / (float)fabs(major);  // (single precision: this only picks the start)
  const float u = (float)(axis == 0 ? ld.y : (axis == 1 ? ld.z : ld.x)) * inv, v = (float)(axis == 0 ? ld.z : (axis == 1 ? ld.x : ld.y)) * inv;
  const int iu = min(max((int)floorf((u + 1.f) * 2.f), 0), 3), iv = min(max((int)floorf((v + 1.f) * 2.f), 0), 3);
  const float4 s0 = o.vert[(2 * axis + (major < 0.0 ? 1 : 0)) * 16 + iu * 4 + iv];
  c.best = {s0.x, s0.y, s0.z};
  c.link = __float_as_int(s0.w);
  c.bd = hull_val(s0.x, s0.y, s0.z, ld);
}
// one chunk of neighbour records against the running best (nb, nlink, c.bd) of the round.  Branch-free: values and tie values of
// all records first (independent chains), then one compare-and-select step per record (measured: with a branch around the rare
// tie case and another around the update, the scalar branch overhead of the eight records was most of a round's time)
__device__ __forceinline__ void climb_eval(const float4 (&q)[kMeshChunk], Climb& c, V3& nb, int& nlink, bool& moved) {
  double v[kMeshChunk], t[kMeshChunk];
#pragma unroll
  for (int i = 0; i < kMeshChunk; i++) { v[i] = hull_val(q[i].x, q[i].y, q[i].z, c.ld); t[i] = hull_tie(q[i].x, q[i].y, q[i].z); }
  double bt = hull_tie(nb.x, nb.y, nb.z);
#pragma unroll
  for (int i = 0; i < kMeshChunk; i++) {
    // strictly better, or exactly equal (the vertex's own padding copies; otherwise rare) and ahead along the generic second direction
    const bool take = (v[i] > c.bd) | ((v[i] == c.bd) & (t[i] > bt));
    c.bd = take ? v[i] : c.bd;
    bt = take ? t[i] : bt;
    nb.x = take ? q[i].x : nb.x; nb.y = take ? q[i].y : nb.y; nb.z = take ? q[i].z : nb.z;
    nlink = take ? __float_as_int(q[i].w) : nlink;
    moved |= take;
  }
}
// (every climb is bounded: a hull has at most kClimbMax vertices worth of strictly improving moves; the bound only matters for
// corrupt tables or non-finite directions, where a kernel that never ends would take the device with it)
constexpr int kClimbMax = 1024;
// GROUP = 4 (the narrowphase kernels of models with meshes): FOUR lanes run one test, identically but for the rounds of the climbs - lane r of
// the four loads and values records r and r + 4 of every chunk, then the four exchange their best (key: value, tie value, position in the
// list - the order the one-lane scan takes records in) in two steps inside the quad and fetch the winner's record from the lane that
// holds it.  A round is ~ 90 vector instructions instead of ~ 200, and the instructions of a wave serve sixteen tests.
// (Tried on top, round 4, and dropped: the climbs of the two objects of a support call round by round side by side, their loads in flight
// together - the narrowphase launch 84.7 -> 91.3 us: a hull against a hull is rare, and every other search paid for the second half.)
__device__ __forceinline__ double quad_xchg(double v, bool far) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int plo = far ? __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true);
  const int phi = far ? __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
  return __hiloint2double(phi, plo);
}
__device__ __forceinline__ int quad_xchg(int v, bool far) {
  return far ? __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
}
__device__ __forceinline__ void climb4(const CObj& o, Climb& c) {
#ifdef HB_NARROW_DIAG
  const int rounds_max = min(kClimbMax, g_mpr_limit[1]);
#else
  constexpr int rounds_max = kClimbMax;
#endif
  const int lane = (int)threadIdx.x, r = lane & 3;
  for (int guard = 0; guard < rounds_max; guard++) {
    const int adr = c.link >> 8, nch = c.link & 255;
    // the lane's best so far: the vertex the round starts on (position -1: ahead of every record of the list)
    double bv = c.bd, bt = hull_tie(c.best.x, c.best.y, c.best.z);
    int bi = -1, bl = c.link;
    float bx = c.best.x, by = c.best.y, bz = c.best.z;
    for (int k = 0; k < nch; k++) {
      const float4 q0 = o.nbr[adr + k * kMeshChunk + r], q1 = o.nbr[adr + k * kMeshChunk + r + 4];
      const double v0 = hull_val(q0.x, q0.y, q0.z, c.ld), t0 = hull_tie(q0.x, q0.y, q0.z);
      const double v1 = hull_val(q1.x, q1.y, q1.z, c.ld), t1 = hull_tie(q1.x, q1.y, q1.z);
      bool take = (v0 > bv) | ((v0 == bv) & (t0 > bt));
      bv = take ? v0 : bv; bt = take ? t0 : bt; bi = take ? k * kMeshChunk + r : bi;
      bx = take ? q0.x : bx; by = take ? q0.y : by; bz = take ? q0.z : bz; bl = take ? __float_as_int(q0.w) : bl;
      take = (v1 > bv) | ((v1 == bv) & (t1 > bt));
      bv = take ? v1 : bv; bt = take ? t1 : bt; bi = take ? k * kMeshChunk + r + 4 : bi;
      bx = take ? q1.x : bx; by = take ? q1.y : by; bz = take ? q1.z : bz; bl = take ? __float_as_int(q1.w) : bl;
    }
    // the four lanes' best: among equal (value, tie value) the earlier position, as the one-lane scan keeps the first
#pragma unroll
    for (int step = 0; step < 2; step++) {
      const double pv = quad_xchg(bv, step != 0), pt = quad_xchg(bt, step != 0);
      const int pi = quad_xchg(bi, step != 0);
      const bool take = (pv > bv) | ((pv == bv) & ((pt > bt) | ((pt == bt) & (pi < bi))));
      bv = take ? pv : bv; bt = take ? pt : bt; bi = take ? pi : bi;
    }
    if (bi < 0) break;  // no record is better than the vertex: the climb ends on it
    const int owner = ((lane & ~3) | (bi & 3)) << 2;
    c.best.x = __int_as_float(__builtin_amdgcn_ds_bpermute(owner, __float_as_int(bx)));
    c.best.y = __int_as_float(__builtin_amdgcn_ds_bpermute(owner, __float_as_int(by)));
    c.best.z = __int_as_float(__builtin_amdgcn_ds_bpermute(owner, __float_as_int(bz)));
    c.link = __builtin_amdgcn_ds_bpermute(owner, bl);
    c.bd = bv;
  }
}
__device__ __forceinline__ void climb(const CObj& o, Climb& c) {
#ifdef HB_NARROW_DIAG
  const int rounds_max = min(kClimbMax, g_mpr_limit[1]);
#else
  constexpr int rounds_max = kClimbMax;
#endif
  for (int guard = 0; guard < rounds_max; guard++) {
    const int adr = c.link >> 8, nch = c.link & 255;
    bool moved = false;
    V3 nb = c.best;
    int nlink = c.link;
    for (int k = 0; k < nch; k++) {
      float4 q[kMeshChunk];
#pragma unroll
      for (int i = 0; i < kMeshChunk; i++) q[i] = o.nbr[adr + k * kMeshChunk + i];
      climb_eval(q, c, nb, nlink, moved);
    }
    if (!moved) break;
    c.best = nb; c.link = nlink;
  }
}
// the point farthest along dir (unit)
// (MESH = 0: an instantiation for models without mesh geoms - the hull climb, its loads and its registers are not in the kernel)
template <int MESH = 1, int GROUP = 1>
__device__ __forceinline__ V3d ccd_support(const CObj& o, V3d dir) {
  if (o.type < 0) {
    V3d best = widen(o.p0), c;
    double bd = dot(best, dir), v;
    c = widen(o.p1); v = dot(c, dir); if (v > bd) { bd = v; best = c; }
    c = widen(o.p2); v = dot(c, dir); if (v > bd) { bd = v; best = c; }
    c = widen(o.p3); v = dot(c, dir); if (v > bd) { bd = v; best = c; }
    c = widen(o.p4); v = dot(c, dir); if (v > bd) { bd = v; best = c; }
    c = widen(o.p5); v = dot(c, dir); if (v > bd) { bd = v; best = c; }
    return best;
  }
  const V3d ld = local_dir(o, dir);
  V3d res;
  if (o.type == 2) res = ld * (double)o.r;
  else if (o.type == 3) { res = ld * (double)o.r; res.z += ld.z >= 0.0 ? (double)o.h : -(double)o.h; }
  else if constexpr (MESH != 0) {
    Climb c;
    c.ld = ld;
    climb_start(o, c);
    if constexpr (GROUP == 4) climb4(o, c);
    else climb(o, c);
    res = widen(c.best);
  } else res = ld * (double)o.r;  // (never reached: the host takes this instantiation only for a model without meshes)
  return world_point(o, ld, res);
}

// mjc_PlaneConvex for a mesh hull (oracle: plane_convex): the hull's support point along -normal is the first contact if it is within the
// margin, up to three more come from the vertices adjacent to it in the hull's graph, in list order, each within the margin and at least
// tol away from the first point.  Returns the count (0..4) and the contacts `first_k`, `first_k + 1` of that list in (da, pa), (db, pb):
// a work item carries two contacts, so a plane - mesh pair is two items.
__device__ __forceinline__ int plane_hull(const CObj& o, V3 ppos, V3 normal, float margin, float tol, int first_k, float& da, V3& pa, float& db, V3& pb) {
  const V3d nrm = widen(normal), pp = widen(ppos);
  Climb c;
  c.ld = local_dir(o, nrm * -1.0);
  climb_start(o, c);
  climb(o, c);
  const V3d first = world_point(o, c.ld, widen(c.best));  // (o.margin = 0)
  const double dist = dot(first - pp, nrm);
  if (dist > (double)margin) return 0;
  int n = 0;
  auto put = [&](V3d pnt, double d) {
    const V3 pos = narrow(pnt - nrm * (0.5 * d));
    if (n == first_k) { da = (float)d; pa = pos; }
    if (n == first_k + 1) { db = (float)d; pb = pos; }
    n++;
  };
  put(first, dist);
  const int adr = c.link >> 8, nch = c.link & 255;
  for (int i = 0; i < nch * kMeshChunk && n < 4; i++) {
    const float4 q = o.nbr[adr + i];
    if (q.x == c.best.x && q.y == c.best.y && q.z == c.best.z) continue;  // the padding of the list: copies of the vertex itself
    const V3d pnt = world_point(o, c.ld, V3d{(double)q.x, (double)q.y, (double)q.z});
    const double di = dot(pnt - pp, nrm);
    const V3d df = pnt - first;
    if (di > (double)margin || sqrt(dot(df, df)) < (double)tol) continue;
    put(pnt, di);
  }
  return n;
}

struct CSup { V3d v, v1; };  // a point of the Minkowski difference obj1 - obj2 and its witness on obj1 (the one on obj2 is v1 - v)
template <int MESH = 1, int GROUP = 1>
__device__ __forceinline__ CSup mpr_support(const CObj& o1, const CObj& o2, V3d dir) {
  CSup s;
  s.v1 = ccd_support<MESH, GROUP>(o1, dir);
  const V3d w2 = ccd_support<MESH, GROUP>(o2, dir * -1.0);
  s.v = s.v1 - w2;
  return s;
}
__device__ __forceinline__ V3d mpr_portal_dir(const CSup& P1, const CSup& P2, const CSup& P3) { return normalized(cross(P2.v - P1.v, P3.v - P1.v)); }
__device__ __forceinline__ bool mpr_reach_tolerance(const CSup& P1, const CSup& P2, const CSup& P3, const CSup& v4, V3d dir, double tol) {
  const double dv4 = dot(v4.v, dir);
  const double d = fmin(fmin(dv4 - dot(P1.v, dir), dv4 - dot(P2.v, dir)), dv4 - dot(P3.v, dir));
  return ccd_eq(d, tol) || d < tol;
}
// (which portal vertex v4 replaces is data: written as a store through a chosen pointer the three vertices live in scratch memory, a
// round trip through the cache hierarchy on every use; as selects they stay in registers)
__device__ __forceinline__ V3d pick(bool c, V3d a, V3d b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
__device__ __forceinline__ void mpr_expand_portal(const CSup& P0, CSup& P1, CSup& P2, CSup& P3, const CSup& v4) {
  const V3d v4v0 = cross(v4.v, P0.v);
  const bool a = dot(P1.v, v4v0) > 0.0, b = dot(P2.v, v4v0) > 0.0, c = dot(P3.v, v4v0) > 0.0;
  const bool to1 = a ? b : !c, to2 = !a && c, to3 = a && !b;
  P1.v = pick(to1, v4.v, P1.v); P1.v1 = pick(to1, v4.v1, P1.v1);
  P2.v = pick(to2, v4.v, P2.v); P2.v1 = pick(to2, v4.v1, P2.v1);
  P3.v = pick(to3, v4.v, P3.v); P3.v1 = pick(to3, v4.v1, P3.v1);
}

// ccdMPRPenetration: true (and depth, dir from obj1 into obj2, pos) when the objects intersect.
// libccd's three loops (discoverPortal, refinePortal, findPenetration: the oracle's mpr_penetration has them as written there) run here
// as ONE loop around ONE support call, the lane's place in the algorithm kept in `state`: a wave runs a different test in every lane,
// and lanes inside different loops of the original would execute those loops one after the other (the wave's time the sum over the
// loops of the slowest lane in each); with one loop it is the number of support calls of the longest test.  Per lane the statements,
// their order and their operands are the original's.
template <int MESH = 1, int GROUP = 1>
__device__ __forceinline__ bool mpr_penetration(const CObj& o1, const CObj& o2, int max_iterations, double tolerance, float& depth_out, V3& pdir_out, V3& pos_out) {
  CSup P0, P1, P2, P3;
  const V3d origin = {0.0, 0.0, 0.0};
  P1.v = origin; P1.v1 = origin; P2 = P1; P3 = P1;
  // ---- discoverPortal
  P0.v1 = ccd_center(o1);
  P0.v = P0.v1 - ccd_center(o2);
  if (ccd_eq(P0.v.x, 0.0) && ccd_eq(P0.v.y, 0.0) && ccd_eq(P0.v.z, 0.0)) P0.v.x += HB_CCD_EPS * 10.0;
  V3d dir = normalized(P0.v * -1.0);
  enum { S_P1 = 0, S_P2, S_DISCOVER, S_REFINE, S_FIND, S_DONE };
  int state = S_P1;
  int count = 0;  // iterations of the loop the lane is in (discover / refine: libccd's guard; find: `it`)
  bool hit = false;
  // the head of refinePortal's loop: the portal's direction; leaves for findPenetration when the origin is on its outer side
  auto refine_head = [&]() {
    dir = mpr_portal_dir(P1, P2, P3);
    const double dt = dot(dir, P1.v);
    if (ccd_is_zero(dt) || dt > 0.0) { state = S_FIND; count = 0; }  // (findPenetration starts from the same portal direction)
  };
#ifdef HB_NARROW_DIAG
  int diag_calls = 0;
#endif
  while (state != S_DONE) {
#ifdef HB_NARROW_DIAG
    if (diag_calls++ >= g_mpr_limit[0]) break;
#endif
    const CSup s = mpr_support<MESH, GROUP>(o1, o2, dir);
    if (state == S_P1) {
      P1 = s;
      const double dt = dot(P1.v, dir);
      if (ccd_is_zero(dt) || dt < 0.0) state = S_DONE;
      else {
        dir = cross(P0.v, P1.v);
        if (ccd_is_zero(dot(dir, dir))) {
          pos_out = narrow(P1.v1 - P1.v * 0.5);  // 0.5 (v1 + v2), v2 = v1 - v
          if (ccd_eq(P1.v.x, 0.0) && ccd_eq(P1.v.y, 0.0) && ccd_eq(P1.v.z, 0.0)) { depth_out = 0.f; pdir_out = {0.f, 0.f, 0.f}; }  // touching
          else {
            double n;
            pdir_out = narrow(normalized(P1.v, &n));
            depth_out = (float)n;
          }
          hit = true;
          state = S_DONE;
        } else {
          dir = normalized(dir);
          state = S_P2;
        }
      }
    } else if (state == S_P2) {
      P2 = s;
      const double dt = dot(P2.v, dir);
      if (ccd_is_zero(dt) || dt < 0.0) state = S_DONE;
      else {
        dir = normalized(cross(P1.v - P0.v, P2.v - P0.v));
        if (dot(dir, P0.v) > 0.0) { const CSup t = P1; P1 = P2; P2 = t; dir = dir * -1.0; }
        state = S_DISCOVER; count = 0;
      }
    } else if (state == S_DISCOVER) {
      P3 = s;
      double dt = dot(P3.v, dir);
      if (ccd_is_zero(dt) || dt < 0.0) state = S_DONE;
      else {
        bool cont = false;
        dt = dot(cross(P1.v, P3.v), P0.v);
        if (dt < 0.0 && !ccd_is_zero(dt)) { P2 = P3; cont = true; }
        if (!cont) {
          dt = dot(cross(P3.v, P2.v), P0.v);
          if (dt < 0.0 && !ccd_is_zero(dt)) { P1 = P3; cont = true; }
        }
        if (cont) {
          dir = normalized(cross(P1.v - P0.v, P2.v - P0.v));
          if (++count > 1000) state = S_DONE;
        } else {
          state = S_REFINE; count = 0;
          refine_head();
        }
      }
    } else if (state == S_REFINE) {
      const double dt = dot(s.v, dir);
      if (!(ccd_is_zero(dt) || dt > 0.0) || mpr_reach_tolerance(P1, P2, P3, s, dir, tolerance)) state = S_DONE;
      else {
        mpr_expand_portal(P0, P1, P2, P3, s);
        if (++count > 1000) state = S_DONE;
        else refine_head();
      }
    } else {  // S_FIND
      if (mpr_reach_tolerance(P1, P2, P3, s, dir, tolerance) || count > max_iterations) {
        const V3d w = closest_on_triangle(origin, P1.v, P2.v, P3.v);
        const double depth = sqrt(dot(w, w));
        const V3d pdir = ccd_is_zero(depth) ? origin : normalized(w);
        // findPos: barycentric coordinates of the origin in the portal tetrahedron
        dir = mpr_portal_dir(P1, P2, P3);
        double b0 = dot(cross(P1.v, P2.v), P3.v), b1 = dot(cross(P3.v, P2.v), P0.v), b2 = dot(cross(P0.v, P1.v), P3.v), b3 = dot(cross(P2.v, P1.v), P0.v);
        double sum = b0 + b1 + b2 + b3;
        if (ccd_is_zero(sum) || sum < 0.0) {
          b0 = 0.0;
          b1 = dot(cross(P2.v, P3.v), dir); b2 = dot(cross(P3.v, P1.v), dir); b3 = dot(cross(P1.v, P2.v), dir);
          sum = b1 + b2 + b3;
        }
        const double inv = 1.0 / sum;
        // 0.5 (p1 + p2) with p2_i = v1_i - v_i
        const V3d p1 = P0.v1 * b0 + P1.v1 * b1 + P2.v1 * b2 + P3.v1 * b3;
        const V3d pv = P0.v * b0 + P1.v * b1 + P2.v * b2 + P3.v * b3;
        const V3d pos = (p1 - pv * 0.5) * inv;
        depth_out = (float)depth; pdir_out = narrow(pdir); pos_out = narrow(pos);
        hit = true;
        state = S_DONE;
      } else {
        mpr_expand_portal(P0, P1, P2, P3, s);
        count++;
        dir = mpr_portal_dir(P1, P2, P3);
      }
    }
  }
  return hit;
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
