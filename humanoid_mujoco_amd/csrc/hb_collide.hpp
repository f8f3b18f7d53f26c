// hb_collide.hpp - the general narrowphase (mesh hulls and height-field prisms through MPR beside the primitive pairs): work lists,
// work items and the ordered append, shared by the step kernels of the general variants (hb_step.hip) and the staged step's pose /
// narrowphase kernels (hb_narrow.hip).
#pragma once
#include "hb_kcommon.hpp"
namespace hb {

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
template <int MODE, int MESH = 1, int GROUP = 1>
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
    const bool hit = mpr_penetration<MESH, GROUP>(o1, o2, M.mpr_iterations, (double)M.mpr_tolerance, depth, dir, vec);
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

}  // namespace hb
