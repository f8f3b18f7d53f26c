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
bool convex_hull_vertices(const std::vector<double>& pts, std::vector<int>& hull, std::string& err, std::vector<int>* tris) {
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
  if (tris) {
    std::vector<int> local(n, -1);
    for (size_t k = 0; k < hull.size(); k++) local[hull[k]] = (int)k;
    tris->clear();
    for (auto& f : faces) if (f.alive) for (int k = 0; k < 3; k++) tris->push_back(local[f.v[k]]);
  }
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
