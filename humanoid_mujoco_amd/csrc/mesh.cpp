// mesh.cpp — mesh assets for the model compiler: STL reader and 3-D convex hull.
//
// The reference's own robot (simulation/assets/humanoid.xml:4-13, world.xml:18) collides through
// `geom type="mesh"`; MuJoCo collides a mesh geom through its CONVEX HULL (computed at compile time by
// qhull) and needs only the hull's vertices: the narrowphase is support-function based (libccd MPR,
// oracle/mjstep_oracle.c: mpr_penetration).  This file produces those vertices: binary / ASCII STL
// (or an inline vertex list) -> unique points -> incremental convex hull -> hull vertices.
// The hull is this engine's own implementation (incremental insertion with horizon edges); the vertex set
// of a convex hull is unique, so any correct algorithm gives what qhull gives up to the tolerance on
// nearly coplanar points (which do not move the support function by more than that tolerance).
#include "hb_model.hpp"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace hb {
namespace {

struct P3 { double x, y, z; };
inline P3 sub(const P3& a, const P3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 crs(const P3& a, const P3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dt(const P3& a, const P3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct Face {
  int v[3];
  P3 n;      // outward unit normal
  double d;  // plane offset: n . x = d
  bool alive;
};

bool make_face(const std::vector<P3>& p, int a, int b, int c, Face& f) {
  P3 n = crs(sub(p[b], p[a]), sub(p[c], p[a]));
  double len = std::sqrt(dt(n, n));
  if (len < 1e-300) return false;
  f.v[0] = a; f.v[1] = b; f.v[2] = c;
  f.n = {n.x / len, n.y / len, n.z / len};
  f.d = dt(f.n, p[a]);
  f.alive = true;
  return true;
}

}  // namespace

// Convex hull of `pts` (3 doubles per point): indices of the hull's vertices, ascending, and (optionally) the hull's triangles as
// triples of positions in `hull`.  false: degenerate input (fewer than four points that are not coplanar).
static bool hull_once(const std::vector<double>& pts, const std::vector<double>& orig, std::vector<int>& hull, std::string& err, std::vector<int>* tris, bool& open_hull);

// Nearly coplanar input (a subdivided CAD face rounded to float32) can leave the incremental build with an open hull; such input is
// built again on joggled copies of the points (deterministic offsets of 1e-8, 1e-7 of the extent: qhull's "QJ"), which have no exact
// coplanarities; the vertex SET comes out of the joggled topology, the coordinates stay the file's, and the result is verified against
// them (closed manifold, Euler's formula, no input point outside a face by more than 2e-6 of the extent).
bool convex_hull_vertices(const std::vector<double>& pts, std::vector<int>& hull, std::string& err, std::vector<int>* tris) {
  bool open_hull = false;
  if (hull_once(pts, pts, hull, err, tris, open_hull) || !open_hull) return err.empty() && !hull.empty();
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (size_t i = 0; i < pts.size(); i++) { lo[i % 3] = std::min(lo[i % 3], pts[i]); hi[i % 3] = std::max(hi[i % 3], pts[i]); }
  const double scale = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
  for (double amp : {1e-8, 1e-7}) {
    std::vector<double> jog(pts);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < jog.size(); i++) {
      x += 0x9E3779B97F4A7C15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      jog[i] += amp * scale * ((double)(z >> 11) / 9007199254740992.0 - 0.5);
    }
    std::string e2;
    if (hull_once(jog, pts, hull, e2, tris, open_hull)) { err.clear(); return true; }
    err = e2;
    if (!open_hull) return false;
  }
  return false;
}

// one incremental build on `pts`; the final convexity check runs against `orig` (the file's coordinates).  open_hull: the failure was
// an open / non-manifold result (worth another attempt on joggled points), not degenerate input
static bool hull_once(const std::vector<double>& pts, const std::vector<double>& orig, std::vector<int>& hull, std::string& err, std::vector<int>* tris, bool& open_hull) {
  open_hull = false;
  hull.clear();
  err.clear();
  const int n = (int)pts.size() / 3;
  std::vector<P3> p(n);
  P3 lo = {1e300, 1e300, 1e300}, hi = {-1e300, -1e300, -1e300};
  for (int i = 0; i < n; i++) {
    p[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    lo = {std::min(lo.x, p[i].x), std::min(lo.y, p[i].y), std::min(lo.z, p[i].z)};
    hi = {std::max(hi.x, p[i].x), std::max(hi.y, p[i].y), std::max(hi.z, p[i].z)};
  }
  if (n < 4) { err = "mesh: fewer than four vertices"; return false; }
  const double scale = std::max(hi.x - lo.x, std::max(hi.y - lo.y, hi.z - lo.z));
  if (!(scale > 0)) { err = "mesh: all vertices coincide"; return false; }
  const double eps = 1e-10 * scale;  // a point closer than this to a face plane counts as on it (not a new vertex); qhull merges nearly coplanar
  // facets with a coarser tolerance, so for rounded float32 meshes this hull keeps some more (harmless) vertices than MuJoCo's
  // initial tetrahedron: the two points farthest apart along the widest axis, the point farthest from their line, the
  // point farthest from that triangle's plane
  int i0 = 0, i1 = 0;
  {
    const int ax = (hi.x - lo.x >= hi.y - lo.y && hi.x - lo.x >= hi.z - lo.z) ? 0 : (hi.y - lo.y >= hi.z - lo.z ? 1 : 2);
    auto get = [&](int i) { return ax == 0 ? p[i].x : (ax == 1 ? p[i].y : p[i].z); };
    for (int i = 1; i < n; i++) { if (get(i) < get(i0)) i0 = i; if (get(i) > get(i1)) i1 = i; }
  }
  int i2 = -1;
  double best = 0;
  for (int i = 0; i < n; i++) {
    P3 c = crs(sub(p[i1], p[i0]), sub(p[i], p[i0]));
    double a = dt(c, c);
    if (a > best) { best = a; i2 = i; }
  }
  if (i2 < 0 || std::sqrt(best) < eps * scale) { err = "mesh: vertices are collinear"; return false; }
  Face base;
  make_face(p, i0, i1, i2, base);
  int i3 = -1;
  best = 0;
  for (int i = 0; i < n; i++) {
    double a = std::fabs(dt(base.n, p[i]) - base.d);
    if (a > best) { best = a; i3 = i; }
  }
  if (i3 < 0 || best < eps) { err = "mesh: vertices are coplanar"; return false; }
  std::vector<Face> faces;
  auto add_oriented = [&](int a, int b, int c, const P3& inside) {
    Face f;
    if (!make_face(p, a, b, c, f)) return;
    if (dt(f.n, inside) - f.d > 0) { std::swap(f.v[1], f.v[2]); f.n = {-f.n.x, -f.n.y, -f.n.z}; f.d = -f.d; }
    faces.push_back(f);
  };
  const P3 cen = {(p[i0].x + p[i1].x + p[i2].x + p[i3].x) / 4, (p[i0].y + p[i1].y + p[i2].y + p[i3].y) / 4, (p[i0].z + p[i1].z + p[i2].z + p[i3].z) / 4};
  add_oriented(i0, i1, i2, cen); add_oriented(i0, i1, i3, cen); add_oriented(i0, i2, i3, cen); add_oriented(i1, i2, i3, cen);
  if (faces.size() != 4) { err = "mesh: degenerate initial simplex"; return false; }
  // incremental insertion in index order
  std::vector<char> done(n, 0);
  done[i0] = done[i1] = done[i2] = done[i3] = 1;
  std::vector<int> order;
  for (int i = 0; i < n; i++) if (!done[i]) order.push_back(i);
  std::vector<int> visible;
  std::map<std::pair<int, int>, int> edges;
  for (int idx : order) {
    visible.clear();
    for (int f = 0; f < (int)faces.size(); f++)
      if (faces[f].alive && dt(faces[f].n, p[idx]) - faces[f].d > eps) visible.push_back(f);
    if (visible.empty()) continue;  // inside the current hull (or on it)
    edges.clear();
    for (int f : visible)
      for (int k = 0; k < 3; k++) edges[{faces[f].v[k], faces[f].v[(k + 1) % 3]}] = f;
    for (int f : visible) faces[f].alive = false;
    // horizon: directed edges of visible faces whose reverse is not an edge of a visible face; the new face keeps the edge's direction
    for (auto& e : edges) {
      if (edges.count({e.first.second, e.first.first})) continue;
      Face f;
      if (make_face(p, e.first.first, e.first.second, idx, f)) faces.push_back(f);
    }
    if (faces.size() > 200000) {  // compact
      std::vector<Face> keep;
      for (auto& f : faces) if (f.alive) keep.push_back(f);
      faces.swap(keep);
    }
  }
  std::vector<char> used(n, 0);
  for (auto& f : faces) if (f.alive) for (int k = 0; k < 3; k++) used[f.v[k]] = 1;
  hull.clear();
  for (int i = 0; i < n; i++) if (used[i]) hull.push_back(i);
  if (hull.size() < 4) { err = "mesh: convex hull collapsed"; return false; }
  // The result must be a closed, convex 2-manifold: every directed edge has exactly one reverse, V - E + F = 2, and no input point lies
  // outside any face by more than the tolerance.  (Nearly coplanar input can leave the visible set of an insertion not simply connected;
  // a horizon face that make_face drops, or a directed edge seen twice, would give an open hull - and the device's hill climb along the
  // hull's edge graph can stop at a non-maximal vertex where an edge is missing.  Refused here instead.)
  {
    std::map<std::pair<int, int>, int> dir;
    size_t nf = 0;
    for (auto& f : faces) {
      if (!f.alive) continue;
      nf++;
      for (int k = 0; k < 3; k++) dir[{f.v[k], f.v[(k + 1) % 3]}]++;
    }
    for (auto& e : dir)
      if (e.second != 1 || !dir.count({e.first.second, e.first.first})) { err = "mesh: convex hull is not a closed manifold (nearly coplanar or duplicate vertices?)"; open_hull = true; hull.clear(); return false; }
    const long long V = (long long)hull.size(), E = (long long)dir.size() / 2, F = (long long)nf;
    if (V - E + F != 2) { err = "mesh: convex hull fails Euler's formula (V - E + F = " + std::to_string(V - E + F) + ")"; open_hull = true; hull.clear(); return false; }
    // (2e-6 of the extent: float32 coordinates are rounded by 6e-8 of it, and a small face - three neighbouring points of a subdivided
    // CAD facet - tilts by that rounding over its own size, which a point at the far end of the facet sees magnified by the ratio)
    const double tol = 2e-6 * scale;
    for (auto& f : faces) {
      if (!f.alive) continue;
      // (the face through the file's coordinates of its three vertices, against the file's coordinates of every point)
      const P3 a = {orig[3 * f.v[0]], orig[3 * f.v[0] + 1], orig[3 * f.v[0] + 2]}, b = {orig[3 * f.v[1]], orig[3 * f.v[1] + 1], orig[3 * f.v[1] + 2]},
               c = {orig[3 * f.v[2]], orig[3 * f.v[2] + 1], orig[3 * f.v[2] + 2]};
      P3 nn = crs(sub(b, a), sub(c, a));
      const double len = std::sqrt(dt(nn, nn));
      const double e1 = dt(sub(b, a), sub(b, a)), e2 = dt(sub(c, a), sub(c, a)), e3 = dt(sub(c, b), sub(c, b));
      if (len < 1e-4 * std::max(e1, std::max(e2, e3))) continue;  // a sliver between nearly collinear vertices: its normal means nothing
      nn = {nn.x / len, nn.y / len, nn.z / len};
      const double dd = dt(nn, a);
      for (int i = 0; i < n; i++) {
        const double out = nn.x * orig[3 * i] + nn.y * orig[3 * i + 1] + nn.z * orig[3 * i + 2] - dd;
        if (out > tol) { err = "mesh: convex hull leaves input vertices outside (by " + std::to_string(out / scale) + " of the extent)"; open_hull = true; hull.clear(); return false; }
      }
    }
  }
  if (tris) {
    std::vector<int> local(n, -1);
    for (size_t k = 0; k < hull.size(); k++) local[hull[k]] = (int)k;
    tris->clear();
    for (auto& f : faces) if (f.alive) for (int k = 0; k < 3; k++) tris->push_back(local[f.v[k]]);
  }
  return true;
}

// Volume, centre of mass and inertia tensor about the centre of mass (unit density; xx, yy, zz, xy, xz, yz) of the solid bounded by a
// closed, outward-oriented triangle surface: the sum over the signed tetrahedra (origin, a, b, c) of the closed-form tetrahedron
// integrals.  This is what MuJoCo's compiler derives a mesh geom's mass, inertial frame and inertia from (the reference's robot:
// simulation/assets/humanoid.xml:22-93 gives its mesh geoms no size, world.xml:18 its defaults); it integrates over the mesh's own
// faces, this compiler over the faces of the convex hull it keeps - the same solid for a convex mesh, the hull's for any other.
bool mesh_mass_properties(const std::vector<double>& v, const std::vector<int>& tris, double& volume, double com[3], double inertia[6]) {
  // second moments about the origin: integral of x_i x_j over a tetrahedron (0, a, b, c) = det / 120 * (sum over vertex pairs incl. equal ones)
  double vol = 0, m1[3] = {0, 0, 0}, m2[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  // (integrated about a point inside the solid: the first vertex - smaller cancellations than about a far-away file origin)
  const double o[3] = {v[0], v[1], v[2]};
  for (size_t t = 0; t + 2 < tris.size(); t += 3) {
    double a[3], b[3], c[3];
    for (int i = 0; i < 3; i++) { a[i] = v[3 * tris[t] + i] - o[i]; b[i] = v[3 * tris[t + 1] + i] - o[i]; c[i] = v[3 * tris[t + 2] + i] - o[i]; }
    const double det = a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
    vol += det / 6;
    for (int i = 0; i < 3; i++) m1[i] += det / 24 * (a[i] + b[i] + c[i]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        m2[i][j] += det / 120 * (2 * (a[i] * a[j] + b[i] * b[j] + c[i] * c[j]) + a[i] * b[j] + b[i] * a[j] + a[i] * c[j] + c[i] * a[j] + b[i] * c[j] + c[i] * b[j]);
  }
  if (!(vol > 0)) return false;
  volume = vol;
  double cl[3];
  for (int i = 0; i < 3; i++) { cl[i] = m1[i] / vol; com[i] = o[i] + cl[i]; }
  // central second moments, then the inertia tensor
  double c2[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c2[i][j] = m2[i][j] - vol * cl[i] * cl[j];
  inertia[0] = c2[1][1] + c2[2][2]; inertia[1] = c2[0][0] + c2[2][2]; inertia[2] = c2[0][0] + c2[1][1];
  inertia[3] = -c2[0][1]; inertia[4] = -c2[0][2]; inertia[5] = -c2[1][2];
  return true;
}

// Vertices of an STL file (binary, or ASCII "solid ... vertex x y z"), duplicates removed, in first-appearance order.
bool read_stl_vertices(const std::string& path, std::vector<double>& pts, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = "cannot open mesh file: " + path; return false; }
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string b = ss.str();
  std::vector<float> raw;
  bool binary = false;
  if (b.size() >= 84) {
    uint32_t ntri;
    memcpy(&ntri, b.data() + 80, 4);
    if ((size_t)84 + (size_t)50 * ntri == b.size()) {
      binary = true;
      if (ntri < 1 || ntri > 20000000u) { err = "mesh: bad triangle count in " + path; return false; }
      raw.resize((size_t)9 * ntri);
      for (uint32_t t = 0; t < ntri; t++) memcpy(&raw[(size_t)9 * t], b.data() + 84 + (size_t)50 * t + 12, 36);
    }
  }
  if (!binary) {
    if (b.compare(0, 5, "solid") != 0) { err = "mesh: not an STL file (size does not match a binary STL and it does not start with 'solid'): " + path; return false; }
    std::istringstream is(b);
    std::string tok;
    while (is >> tok)
      if (tok == "vertex") { float x, y, z; if (!(is >> x >> y >> z)) { err = "mesh: bad vertex in ASCII STL " + path; return false; } raw.push_back(x); raw.push_back(y); raw.push_back(z); }
    if (raw.size() < 9) { err = "mesh: no triangles in " + path; return false; }
  }
  // unique points (exact float equality, as the file stores them), first appearance order
  std::map<std::tuple<float, float, float>, int> seen;
  pts.clear();
  for (size_t i = 0; i + 2 < raw.size(); i += 3) {
    if (!std::isfinite(raw[i]) || !std::isfinite(raw[i + 1]) || !std::isfinite(raw[i + 2])) { err = "mesh: non-finite vertex in " + path; return false; }
    auto key = std::make_tuple(raw[i], raw[i + 1], raw[i + 2]);
    if (seen.emplace(key, (int)seen.size()).second) { pts.push_back(raw[i]); pts.push_back(raw[i + 1]); pts.push_back(raw[i + 2]); }
  }
  return true;
}

}  // namespace hb
