// mjcf.cpp — MJCF subset compiler: XML -> hb::Model.
//
// Replaces mj_loadXML (reference: simulation/mujoco/include/mujoco/mujoco.h:103; called from
// simulation/cpu_env.py:86 and simulation/mujoco/sample/testspeed.cc:159) for the MJCF features
// the hot-path models use (simulation/mujoco/model/humanoid/humanoid.xml): nested <default>
// classes, childclass, bodies with free/hinge/slide joints, plane/sphere/capsule geoms
// (fromto, zaxis, euler, axisangle, xyaxes, quat), geom-derived inertia at density 1000,
// <inertial>, fixed tendons, motor/position/general actuators on joints, <contact><exclude>,
// keyframes, <option> + <flag>.  Angles default to degrees (<compiler angle>).
// The MuJoCo compiler source is not in the reference tree; semantics follow the public MJCF
// documentation [recall], see DESIGN.md "Model compiler".
#include "hb_model.hpp"
#include "hmath.hpp"
#include "xml_mini.hpp"
#include <algorithm>
#include <array>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>

namespace hb {
namespace {

const double PI = 3.14159265358979323846;
typedef std::map<std::string, std::string> AttrMap;

struct DefaultClass {
  std::map<std::string, AttrMap> kind;  // "geom","joint","actuator","tendon" -> attrs
};

struct Ctx {
  std::string err;
  bool degree = true;
  std::string eulerseq = "xyz";
  bool autolimits = true;
  double default_density = 1000.0;
  std::map<std::string, DefaultClass> classes;
  std::map<std::string, std::string> hfield_names;  // name -> index (as string)
  std::map<std::string, int> mesh_names;            // name -> mesh id
  std::vector<double> mesh_center;                  // per mesh: centre of mass of its hull in the file's coordinates (the hull is stored relative to it ...
  std::vector<double> mesh_quat;                    //   ... in its principal axes of inertia: their orientation in the file's frame, 4 per mesh)
  std::vector<double> mesh_volume, mesh_inertia;    // per mesh: volume; principal moments at unit density (3 per mesh, descending)
  std::vector<double> mesh_rbound;                  // per mesh: largest vertex distance from that centre
  std::string basedir, meshdir;
  Model* m = nullptr;
  bool fail(const std::string& e) { if (err.empty()) err = e; return false; }
};

bool parse_doubles(const std::string& s, std::vector<double>& out) {
  out.clear();
  const char* p = s.c_str();
  char* e;
  for (;;) {
    while (*p && isspace((unsigned char)*p)) p++;
    if (!*p) break;
    double v = strtod(p, &e);
    if (e == p) return false;
    out.push_back(v);
    p = e;
  }
  return true;
}

// attribute lookup in merged attrs
struct Attrs {
  AttrMap a;
  Ctx* c;
  std::string where;
  bool has(const char* k) const { return a.count(k) != 0; }
  std::string str(const char* k, const std::string& d = "") const { auto it = a.find(k); return it == a.end() ? d : it->second; }
  // read up to n doubles into v (which holds defaults); returns count read or -1
  int vec(const char* k, double* v, int nmax, int nmin = 1) const {
    auto it = a.find(k);
    if (it == a.end()) return 0;
    std::vector<double> t;
    if (!parse_doubles(it->second, t) || (int)t.size() > nmax || (int)t.size() < nmin) {
      c->fail("mjcf: bad value for '" + std::string(k) + "' in " + where + ": \"" + it->second + "\"");
      return -1;
    }
    for (size_t i = 0; i < t.size(); i++) v[i] = t[i];
    return (int)t.size();
  }
  double num(const char* k, double d) const { double v = d; vec(k, &v, 1); return v; }
  int integer(const char* k, int d) const { double v = d; vec(k, &v, 1); return (int)v; }
  // tri-state bool: "true"/"false"/"auto" -> 1/0/-1
  int tri(const char* k, int d) const {
    auto it = a.find(k);
    if (it == a.end()) return d;
    if (it->second == "true") return 1;
    if (it->second == "false") return 0;
    if (it->second == "auto") return -1;
    c->fail("mjcf: bad boolean for '" + std::string(k) + "' in " + where);
    return d;
  }
};

void overlay(AttrMap& dst, const XmlNode& n) { for (auto& a : n.attrs) dst[a.first] = a.second; }

// <default> tree: a child class starts as a copy of its parent
void read_defaults(Ctx& c, const XmlNode& n, const std::string& parent, bool top) {
  std::string cname = top ? "main" : (n.attr("class") ? *n.attr("class") : "");
  if (!top && cname.empty()) { c.fail("mjcf: nested <default> without class name"); return; }
  if (top && n.attr("class")) cname = *n.attr("class");
  DefaultClass dc;
  if (!parent.empty()) dc = c.classes[parent];
  else if (c.classes.count(cname)) dc = c.classes[cname];
  for (auto& ch : n.children) {
    if (ch->name == "default") continue;
    std::string kind = ch->name;
    if (kind == "motor" || kind == "general" || kind == "position" || kind == "velocity") kind = "actuator";
    overlay(dc.kind[kind], *ch);
  }
  c.classes[cname] = dc;
  for (auto& ch : n.children) if (ch->name == "default") read_defaults(c, *ch, cname, false);
}

Attrs merged(Ctx& c, const XmlNode& n, const std::string& kind, const std::string& childclass, bool use_defaults = true) {
  Attrs r;
  r.c = &c;
  r.where = "<" + n.name + (n.attr("name") ? " name=\"" + *n.attr("name") + "\"" : "") + ">";
  if (use_defaults) {
    std::string cls = n.attr("class") ? *n.attr("class") : (childclass.empty() ? "main" : childclass);
    auto it = c.classes.find(cls);
    if (it == c.classes.end()) {
      if (cls != "main") c.fail("mjcf: unknown default class '" + cls + "' in " + r.where);
    } else {
      auto k = it->second.kind.find(kind);
      if (k != it->second.kind.end()) r.a = k->second;
    }
  }
  overlay(r.a, n);
  return r;
}

// orientation from quat / axisangle / euler / xyaxes / zaxis
bool read_orientation(Ctx& c, const Attrs& a, double* quat) {
  quat[0] = 1; quat[1] = quat[2] = quat[3] = 0;
  double v[6];
  if (a.has("quat")) {
    if (a.vec("quat", quat, 4, 4) < 0) return false;
    hm::normalize4(quat);
  } else if (a.has("axisangle")) {
    if (a.vec("axisangle", v, 4, 4) < 0) return false;
    double ang = c.degree ? v[3] * PI / 180 : v[3];
    hm::normalize3(v);
    hm::axis_angle2quat(quat, v, ang);
  } else if (a.has("euler")) {
    if (a.vec("euler", v, 3, 3) < 0) return false;
    for (int i = 0; i < 3; i++) {
      double ang = c.degree ? v[i] * PI / 180 : v[i];
      char ax = c.eulerseq[i];
      double e[3] = {0, 0, 0};
      e[(tolower(ax) - 'x')] = 1;
      double qi[4], t[4];
      hm::axis_angle2quat(qi, e, ang);
      if (islower(ax)) hm::mul_quat(t, quat, qi);  // intrinsic: post-multiply
      else hm::mul_quat(t, qi, quat);               // extrinsic: pre-multiply
      memcpy(quat, t, sizeof t);
    }
    hm::normalize4(quat);
  } else if (a.has("xyaxes")) {
    if (a.vec("xyaxes", v, 6, 6) < 0) return false;
    double x[3] = {v[0], v[1], v[2]}, y[3] = {v[3], v[4], v[5]}, z[3];
    hm::normalize3(x);
    double d = hm::dot3(x, y);
    for (int i = 0; i < 3; i++) y[i] -= d * x[i];
    hm::normalize3(y);
    hm::cross(z, x, y);
    double mat[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]};
    hm::mat2quat(quat, mat);
  } else if (a.has("zaxis")) {
    if (a.vec("zaxis", v, 3, 3) < 0) return false;
    hm::z2quat(quat, v);
  }
  return c.err.empty();
}

struct GeomTmp {
  int type;
  double size[3], pos[3], quat[4];
  double mass, inertia[3];  // principal, in geom frame
  double mesh_volume = 0, mesh_inertia[3] = {0, 0, 0};  // a mesh geom: its hull's volume and principal moments at unit density
};

// mass and principal inertia of a primitive at given density (or explicit mass)
void geom_inertia(GeomTmp& g, double density, double mass_attr) {
  double vol = 0;
  double r = g.size[0], h = g.size[1];
  if (g.type == GEOM_SPHERE) vol = 4.0 / 3.0 * PI * r * r * r;
  else if (g.type == GEOM_CAPSULE) vol = PI * r * r * (2 * h) + 4.0 / 3.0 * PI * r * r * r;
  else if (g.type == GEOM_CYLINDER) vol = PI * r * r * (2 * h);
  else vol = 0;
  // (a mesh geom: volume and principal moments of its hull, in whose centre-of-mass / principal frame the geom already sits: compile_geom)
  if (g.type == GEOM_MESH) vol = g.mesh_volume;
  double mass = mass_attr >= 0 ? mass_attr : density * vol;
  g.mass = mass;
  g.inertia[0] = g.inertia[1] = g.inertia[2] = 0;
  if (g.type == GEOM_MESH && vol > 0) for (int i = 0; i < 3; i++) g.inertia[i] = mass / vol * g.mesh_inertia[i];
  if (g.type == GEOM_SPHERE) {
    g.inertia[0] = g.inertia[1] = g.inertia[2] = 0.4 * mass * r * r;
  } else if (g.type == GEOM_CAPSULE) {
    double height = 2 * h;
    double msph = mass * (4.0 / 3.0 * r) / (4.0 / 3.0 * r + height);  // two hemispheres together
    double mcyl = mass - msph;
    double ix = mcyl * (3 * r * r + height * height) / 12.0;
    double iz = mcyl * r * r / 2.0;
    double isph = 0.4 * msph * r * r;
    // hemispheres displaced along z: each hemisphere com at h + 3r/8; parallel-axis via full formula
    ix += isph + msph * height * (3 * r + 2 * height) / 8.0;
    iz += isph;
    g.inertia[0] = g.inertia[1] = ix;
    g.inertia[2] = iz;
  } else if (g.type == GEOM_CYLINDER) {
    const double height = 2 * h;
    g.inertia[0] = g.inertia[1] = mass * (3 * r * r + height * height) / 12.0;
    g.inertia[2] = mass * r * r / 2.0;
  }
}

struct BodyBuild {
  std::vector<GeomTmp> geoms;
  bool has_inertial = false;
  double ipos[3], iquat[4], mass, inertia[3];
};

void push3(vecd& v, const double* x) { v.push_back(x[0]); v.push_back(x[1]); v.push_back(x[2]); }
void push4(vecd& v, const double* x) { for (int i = 0; i < 4; i++) v.push_back(x[i]); }

int find_name(const std::vector<std::string>& names, const std::string& n) {
  for (size_t i = 0; i < names.size(); i++) if (names[i] == n) return (int)i;
  return -1;
}

bool compile_body(Ctx& c, const XmlNode& n, int parent, const std::string& childclass_in, int depth);

bool add_geom(Ctx& c, const XmlNode& n, int body, const std::string& childclass, BodyBuild& bb) {
  Model& m = *c.m;
  Attrs a = merged(c, n, "geom", childclass);
  std::string ts = a.str("type", "sphere");
  int type;
  if (ts == "plane") type = GEOM_PLANE;
  else if (ts == "sphere") type = GEOM_SPHERE;
  else if (ts == "capsule") type = GEOM_CAPSULE;
  else if (ts == "hfield") type = GEOM_HFIELD;
  else if (ts == "mesh") type = GEOM_MESH;
  else if (ts == "cylinder") type = GEOM_CYLINDER;  // accepted as a non-colliding (visual) geom only: build_pairs refuses it in a collision pair
  else return c.fail("mjcf: geom type '" + ts + "' is not supported (plane, sphere, capsule, hfield, mesh; cylinder when non-colliding) in " + a.where);
  GeomTmp g;
  g.type = type;
  g.size[0] = g.size[1] = g.size[2] = 0;
  if (a.vec("size", g.size, 3) < 0) return false;
  g.pos[0] = g.pos[1] = g.pos[2] = 0;
  if (a.vec("pos", g.pos, 3, 3) < 0) return false;
  if (!read_orientation(c, a, g.quat)) return false;
  if (a.has("fromto")) {
    if (type != GEOM_CAPSULE && type != GEOM_CYLINDER) return c.fail("mjcf: fromto requires a capsule or cylinder in " + a.where);
    double ft[6];
    if (a.vec("fromto", ft, 6, 6) < 0) return false;
    double vec[3] = {ft[0] - ft[3], ft[1] - ft[4], ft[2] - ft[5]};
    double len = hm::norm3(vec);
    if (len < 1e-12) return c.fail("mjcf: fromto points too close in " + a.where);
    // with fromto, size holds only the radius
    g.size[1] = len / 2;
    for (int i = 0; i < 3; i++) g.pos[i] = 0.5 * (ft[i] + ft[i + 3]);
    hm::z2quat(g.quat, vec);
  }
  if (type == GEOM_SPHERE && g.size[0] <= 0) return c.fail("mjcf: sphere needs size>0 in " + a.where);
  if (type == GEOM_CAPSULE && (g.size[0] <= 0 || g.size[1] <= 0)) return c.fail("mjcf: capsule needs radius and half-length in " + a.where);
  int dataid = -1;
  if (type == GEOM_MESH) {
    // MuJoCo collides a mesh geom through the convex hull of its mesh (mesh.cpp), and its compiler re-centres a mesh on its centre of mass
    // and aligns it with its principal axes of inertia; the offset and the rotation go into the geom's frame (same surface in world
    // coordinates; what moves is the frame origin - the interior point of the MPR portal search - and the geom's inertial frame)
    std::string mn = a.str("mesh");
    auto it = c.mesh_names.find(mn);
    if (it == c.mesh_names.end()) return c.fail("mjcf: unknown mesh '" + mn + "' in " + a.where);
    dataid = it->second;
    double off[3], q[4];
    hm::rot_vec_quat(off, &c.mesh_center[3 * dataid], g.quat);
    for (int i = 0; i < 3; i++) g.pos[i] += off[i];
    hm::mul_quat(q, g.quat, &c.mesh_quat[4 * dataid]);
    hm::normalize4(q);
    memcpy(g.quat, q, sizeof q);
    g.mesh_volume = c.mesh_volume[dataid];
    for (int i = 0; i < 3; i++) g.mesh_inertia[i] = c.mesh_inertia[3 * dataid + i];
    g.size[0] = g.size[1] = g.size[2] = 0;
  }
  if (type == GEOM_HFIELD) {
    std::string hn = a.str("hfield");
    auto it = c.hfield_names.find(hn);
    if (it == c.hfield_names.end()) return c.fail("mjcf: unknown hfield '" + hn + "' in " + a.where);
    dataid = atoi(it->second.c_str());
  }
  geom_inertia(g, a.num("density", c.default_density), a.has("mass") ? a.num("mass", 0) : -1.0);
  bb.geoms.push_back(g);

  m.geom_name.push_back(a.str("name"));
  m.geom_type.push_back(type);
  m.geom_bodyid.push_back(body);
  m.geom_contype.push_back(a.integer("contype", 1));
  m.geom_conaffinity.push_back(a.integer("conaffinity", 1));
  int condim = a.integer("condim", 3);
  if (condim != 1 && condim != 3 && condim != 4 && condim != 6) return c.fail("mjcf: condim must be 1,3,4 or 6 in " + a.where);
  m.geom_condim.push_back(condim);
  m.geom_priority.push_back(a.integer("priority", 0));
  m.geom_dataid.push_back(dataid);
  push3(m.geom_size, g.size);
  push3(m.geom_pos, g.pos);
  push4(m.geom_quat, g.quat);
  double rb = 0;
  if (type == GEOM_SPHERE) rb = g.size[0];
  else if (type == GEOM_CAPSULE) rb = g.size[0] + g.size[1];
  else if (type == GEOM_CYLINDER) rb = std::sqrt(g.size[0] * g.size[0] + g.size[1] * g.size[1]);
  else if (type == GEOM_MESH) rb = c.mesh_rbound[dataid];
  else if (type == GEOM_HFIELD) {
    const double* hs = &m.hfield_size[4 * dataid];
    rb = std::sqrt(hs[0] * hs[0] + hs[1] * hs[1] + std::max(hs[2], hs[3]) * std::max(hs[2], hs[3]));
  }
  m.geom_rbound.push_back(rb);
  double fr[3] = {1, 0.005, 0.0001};
  if (a.vec("friction", fr, 3) < 0) return false;
  push3(m.geom_friction, fr);
  m.geom_solmix.push_back(a.num("solmix", 1));
  double sr[2] = {0.02, 1};
  if (a.vec("solref", sr, 2) < 0) return false;
  m.geom_solref.push_back(sr[0]); m.geom_solref.push_back(sr[1]);
  double si[5] = {0.9, 0.95, 0.001, 0.5, 2};
  if (a.vec("solimp", si, 5) < 0) return false;
  for (int i = 0; i < 5; i++) m.geom_solimp.push_back(si[i]);
  m.geom_margin.push_back(a.num("margin", 0));
  m.geom_gap.push_back(a.num("gap", 0));
  return c.err.empty();
}

bool add_joint(Ctx& c, const XmlNode& n, int body, const std::string& childclass) {
  Model& m = *c.m;
  bool freejoint = n.name == "freejoint";
  Attrs a = merged(c, n, "joint", childclass, !freejoint);
  std::string ts = freejoint ? "free" : a.str("type", "hinge");
  int type;
  if (ts == "free") type = JNT_FREE;
  else if (ts == "hinge") type = JNT_HINGE;
  else if (ts == "slide") type = JNT_SLIDE;
  else return c.fail("mjcf: joint type '" + ts + "' is not supported (free, hinge, slide) in " + a.where);
  if (type == JNT_FREE && m.body_parentid[body] != 0) return c.fail("mjcf: free joint only allowed on a child of the world body");
  if (type == JNT_FREE && m.body_jntnum[body] != 0) return c.fail("mjcf: free joint must be the only joint of its body");
  int nq = type == JNT_FREE ? 7 : 1, nv = type == JNT_FREE ? 6 : 1;
  int jid = m.njnt++;
  if (m.body_jntnum[body] == 0) { m.body_jntadr[body] = jid; m.body_dofadr[body] = m.nv; }
  m.body_jntnum[body]++;
  m.body_dofnum[body] += nv;
  m.jnt_name.push_back(a.str("name"));
  m.jnt_type.push_back(type);
  m.jnt_qposadr.push_back(m.nq);
  m.jnt_dofadr.push_back(m.nv);
  m.jnt_bodyid.push_back(body);
  double pos[3] = {0, 0, 0}, axis[3] = {0, 0, 1};
  if (type != JNT_FREE) {
    if (a.vec("pos", pos, 3, 3) < 0 || a.vec("axis", axis, 3, 3) < 0) return false;
    if (hm::norm3(axis) < 1e-12) return c.fail("mjcf: zero joint axis in " + a.where);
    hm::normalize3(axis);
  }
  push3(m.jnt_pos, pos);
  push3(m.jnt_axis, axis);
  bool ang = (type == JNT_HINGE) && c.degree;
  double range[2] = {0, 0};
  int nr = 0;
  double stiffness = 0, damping = 0, armature = 0, frictionloss = 0, margin = 0, ref = 0, springref = 0;
  int limited = 0;
  double solref[2] = {0.02, 1}, solimp[5] = {0.9, 0.95, 0.001, 0.5, 2};
  if (type != JNT_FREE) {
    nr = a.vec("range", range, 2, 2);
    if (nr < 0) return false;
    if (ang) { range[0] *= PI / 180; range[1] *= PI / 180; }
    int lt = a.tri("limited", -1);
    limited = lt >= 0 ? lt : ((c.autolimits && nr == 2) ? 1 : 0);
    if (limited && !(range[0] < range[1])) return c.fail("mjcf: limited joint needs range[0]<range[1] in " + a.where);
    stiffness = a.num("stiffness", 0);
    damping = a.num("damping", 0);
    armature = a.num("armature", 0);
    frictionloss = a.num("frictionloss", 0);
    margin = a.num("margin", 0);
    ref = a.num("ref", 0);
    springref = a.num("springref", 0);
    if (ang) { ref *= PI / 180; springref *= PI / 180; }
    if (a.vec("solreflimit", solref, 2) < 0 || a.vec("solimplimit", solimp, 5) < 0) return false;
  } else {
    // <joint type="free"> may still carry damping/armature; <freejoint> has none
    if (!freejoint) { damping = a.num("damping", 0); armature = a.num("armature", 0); }
  }
  if (frictionloss != 0) return c.fail("mjcf: joint frictionloss is not supported in " + a.where);
  m.jnt_limited.push_back(limited);
  m.jnt_range.push_back(range[0]); m.jnt_range.push_back(range[1]);
  m.jnt_stiffness.push_back(stiffness);
  m.jnt_margin.push_back(margin);
  m.jnt_solref.push_back(solref[0]); m.jnt_solref.push_back(solref[1]);
  for (int i = 0; i < 5; i++) m.jnt_solimp.push_back(solimp[i]);
  // qpos0 / qpos_spring
  if (type == JNT_FREE) {
    for (int i = 0; i < 3; i++) { m.qpos0.push_back(m.body_pos[3 * body + i]); }
    for (int i = 0; i < 4; i++) { m.qpos0.push_back(m.body_quat[4 * body + i]); }
    for (int i = 0; i < 7; i++) m.qpos_spring.push_back(m.qpos0[m.nq + i]);
  } else {
    m.qpos0.push_back(ref);
    m.qpos_spring.push_back(springref);
  }
  // dofs: parent = previous dof of this body, else last dof of nearest ancestor that has dofs
  for (int k = 0; k < nv; k++) {
    int d = m.nv + k;
    int parent_dof = -1;
    if (d > m.body_dofadr[body]) parent_dof = d - 1;
    else {
      int b = m.body_parentid[body];
      while (b > 0 && m.body_dofnum[b] == 0) b = m.body_parentid[b];
      if (b > 0) parent_dof = m.body_dofadr[b] + m.body_dofnum[b] - 1;
    }
    m.dof_bodyid.push_back(body);
    m.dof_jntid.push_back(jid);
    m.dof_parentid.push_back(parent_dof);
    m.dof_armature.push_back(armature);
    m.dof_damping.push_back(damping);
    m.dof_frictionloss.push_back(0);
  }
  m.nq += nq;
  m.nv += nv;
  return c.err.empty();
}

bool finish_body_inertia(Ctx& c, int body, BodyBuild& bb) {
  Model& m = *c.m;
  double ipos[3] = {0, 0, 0}, iquat[4] = {1, 0, 0, 0}, mass = 0, inertia[3] = {0, 0, 0};
  if (body == 0) {
    // the world body is static and massless whatever geoms it carries (floor, the reference's cylinder axis markers)
  } else if (bb.has_inertial) {
    memcpy(ipos, bb.ipos, sizeof ipos); memcpy(iquat, bb.iquat, sizeof iquat);
    mass = bb.mass; memcpy(inertia, bb.inertia, sizeof inertia);
  } else {
    std::vector<GeomTmp*> sel;
    for (auto& g : bb.geoms) if (g.mass > 0) sel.push_back(&g);
    if (sel.size() == 1) {
      memcpy(ipos, sel[0]->pos, sizeof ipos); memcpy(iquat, sel[0]->quat, sizeof iquat);
      mass = sel[0]->mass; memcpy(inertia, sel[0]->inertia, sizeof inertia);
    } else if (sel.size() > 1) {
      for (auto g : sel) { mass += g->mass; for (int i = 0; i < 3; i++) ipos[i] += g->mass * g->pos[i]; }
      for (int i = 0; i < 3; i++) ipos[i] /= mass;
      double I[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (auto g : sel) {
        double R[9];
        hm::quat2mat(R, g->quat);
        for (int r = 0; r < 3; r++)
          for (int s = 0; s < 3; s++)
            for (int k = 0; k < 3; k++) I[3 * r + s] += R[3 * r + k] * g->inertia[k] * R[3 * s + k];
        double d[3] = {g->pos[0] - ipos[0], g->pos[1] - ipos[1], g->pos[2] - ipos[2]};
        double dd = hm::dot3(d, d);
        for (int r = 0; r < 3; r++)
          for (int s = 0; s < 3; s++) I[3 * r + s] += g->mass * ((r == s ? dd : 0.0) - d[r] * d[s]);
      }
      double w[3], V[9];
      hm::eig3(I, w, V);
      // sort descending, keep a right-handed frame
      int idx[3] = {0, 1, 2};
      for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) if (w[idx[j]] > w[idx[i]]) std::swap(idx[i], idx[j]);
      double Rm[9];
      for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) Rm[3 * r + k] = V[3 * r + idx[k]];
      double c0[3] = {Rm[0], Rm[3], Rm[6]}, c1[3] = {Rm[1], Rm[4], Rm[7]}, c2[3];
      hm::cross(c2, c0, c1);
      Rm[2] = c2[0]; Rm[5] = c2[1]; Rm[8] = c2[2];
      hm::mat2quat(iquat, Rm);
      for (int k = 0; k < 3; k++) inertia[k] = w[idx[k]];
    }
  }
  for (int i = 0; i < 3; i++) { m.body_ipos[3 * body + i] = ipos[i]; m.body_inertia[3 * body + i] = inertia[i]; }
  for (int i = 0; i < 4; i++) m.body_iquat[4 * body + i] = iquat[i];
  m.body_mass[body] = mass;
  return true;
}

bool compile_body(Ctx& c, const XmlNode& n, int parent, const std::string& childclass_in, int depth) {
  Model& m = *c.m;
  int body = m.nbody++;
  std::string childclass = n.attr("childclass") ? *n.attr("childclass") : childclass_in;
  Attrs a;
  a.c = &c;
  a.where = "<body name=\"" + (n.attr("name") ? *n.attr("name") : std::string()) + "\">";
  overlay(a.a, n);
  m.body_name.push_back(body == 0 ? "world" : a.str("name"));
  m.body_parentid.push_back(parent < 0 ? 0 : parent);
  m.body_depth.push_back(depth);
  double pos[3] = {0, 0, 0}, quat[4] = {1, 0, 0, 0};
  if (body > 0) {
    if (a.vec("pos", pos, 3, 3) < 0) return false;
    if (!read_orientation(c, a, quat)) return false;
  }
  push3(m.body_pos, pos);
  push4(m.body_quat, quat);
  m.body_jntnum.push_back(0); m.body_jntadr.push_back(-1);
  m.body_dofnum.push_back(0); m.body_dofadr.push_back(-1);
  m.body_geomnum.push_back(0); m.body_geomadr.push_back(-1);
  m.body_mass.push_back(0); m.body_subtreemass.push_back(0);
  for (int i = 0; i < 3; i++) { m.body_ipos.push_back(0); m.body_inertia.push_back(0); }
  push4(m.body_iquat, quat);
  m.body_invweight0.push_back(0); m.body_invweight0.push_back(0);
  m.body_rootid.push_back(0); m.body_weldid.push_back(0);

  BodyBuild bb;
  // joints first (they define dof order), then geoms, in document order within each kind
  for (auto& ch : n.children)
    if (ch->name == "joint" || ch->name == "freejoint") { if (!add_joint(c, *ch, body, childclass)) return false; }
  for (auto& ch : n.children) {
    if (ch->name == "geom") {
      if (m.body_geomnum[body] == 0) m.body_geomadr[body] = m.ngeom;
      if (!add_geom(c, *ch, body, childclass, bb)) return false;
      m.body_geomnum[body]++;
      m.ngeom++;
    } else if (ch->name == "inertial") {
      Attrs ia;
      ia.c = &c; ia.where = "<inertial>";
      overlay(ia.a, *ch);
      bb.has_inertial = true;
      bb.ipos[0] = bb.ipos[1] = bb.ipos[2] = 0;
      if (ia.vec("pos", bb.ipos, 3, 3) < 0) return false;
      if (!read_orientation(c, ia, bb.iquat)) return false;
      bb.mass = ia.num("mass", 0);
      bb.inertia[0] = bb.inertia[1] = bb.inertia[2] = 0;
      if (ia.has("diaginertia")) { if (ia.vec("diaginertia", bb.inertia, 3, 3) < 0) return false; }
      else if (ia.has("fullinertia")) {
        double f[6];
        if (ia.vec("fullinertia", f, 6, 6) < 0) return false;
        double I[9] = {f[0], f[3], f[4], f[3], f[1], f[5], f[4], f[5], f[2]}, w[3], V[9];
        hm::eig3(I, w, V);
        double c0[3] = {V[0], V[3], V[6]}, c1[3] = {V[1], V[4], V[7]}, c2[3];
        hm::cross(c2, c0, c1);
        V[2] = c2[0]; V[5] = c2[1]; V[8] = c2[2];
        double q[4], t[4];
        hm::mat2quat(q, V);
        hm::mul_quat(t, bb.iquat, q);
        memcpy(bb.iquat, t, sizeof t);
        memcpy(bb.inertia, w, sizeof w);
      }
    }
  }
  if (!finish_body_inertia(c, body, bb)) return false;
  // rootid / weldid
  if (body > 0) {
    int p = m.body_parentid[body];
    m.body_rootid[body] = (p == 0) ? body : m.body_rootid[p];
    m.body_weldid[body] = m.body_jntnum[body] > 0 ? body : m.body_weldid[p];
  }
  for (auto& ch : n.children)
    if (ch->name == "body") { if (!compile_body(c, *ch, body, childclass, depth + 1)) return false; }
  return c.err.empty();
}

bool read_option(Ctx& c, const XmlNode& n) {
  Model& m = *c.m;
  Attrs a;
  a.c = &c; a.where = "<option>";
  overlay(a.a, n);
  m.timestep = a.num("timestep", m.timestep);
  m.impratio = a.num("impratio", m.impratio);
  m.tolerance = a.num("tolerance", m.tolerance);
  m.iterations = a.integer("iterations", m.iterations);
  m.ls_iterations = a.integer("ls_iterations", m.ls_iterations);
  m.ls_tolerance = a.num("ls_tolerance", m.ls_tolerance);
  if (a.vec("gravity", m.gravity, 3, 3) < 0) return false;
  std::string s = a.str("solver", "");
  if (s == "PGS") m.solver = SOL_PGS;
  else if (s == "Newton") m.solver = SOL_NEWTON;
  else if (s == "CG") return c.fail("mjcf: solver 'CG' is not implemented; this engine runs PGS or Newton (see DESIGN.md)");
  s = a.str("cone", "pyramidal");
  if (s != "pyramidal") return c.fail("mjcf: only the pyramidal friction cone is implemented");
  s = a.str("integrator", "Euler");
  if (s != "Euler") return c.fail("mjcf: only the Euler integrator is implemented");
  if (const XmlNode* f = n.child("flag")) {
    struct { const char* name; int bit; } flags[] = {
        {"constraint", DSBL_CONSTRAINT}, {"limit", DSBL_LIMIT}, {"contact", DSBL_CONTACT}, {"passive", DSBL_PASSIVE},
        {"gravity", DSBL_GRAVITY}, {"clampctrl", DSBL_CLAMPCTRL}, {"warmstart", DSBL_WARMSTART},
        {"filterparent", DSBL_FILTERPARENT}, {"actuation", DSBL_ACTUATION}, {"refsafe", DSBL_REFSAFE},
        {"eulerdamp", DSBL_EULERDAMP}, {"equality", DSBL_EQUALITY}, {"frictionloss", DSBL_FRICTIONLOSS}};
    for (auto& fl : flags)
      if (const std::string* v = f->attr(fl.name)) {
        if (*v == "disable") m.disableflags |= fl.bit;
        else if (*v == "enable") m.disableflags &= ~fl.bit;
        else return c.fail(std::string("mjcf: flag ") + fl.name + " must be enable or disable");
      }
  }
  return c.err.empty();
}

// <mesh name= file= | vertex= scale=>: convex hull vertices of the mesh (what mesh collision needs)
bool read_mesh(Ctx& c, const XmlNode& n) {
  Model& m = *c.m;
  Attrs a;
  a.c = &c; a.where = "<mesh" + (n.attr("name") ? " name=\"" + *n.attr("name") + "\"" : std::string()) + ">";
  overlay(a.a, n);
  std::vector<double> pts;
  std::string name = a.str("name");
  if (a.has("file")) {
    std::string f = a.str("file");
    std::string path = (f.size() && f[0] == '/') ? f : c.basedir + c.meshdir + f;
    std::string lower = f;
    for (auto& ch : lower) ch = (char)tolower((unsigned char)ch);
    if (lower.size() < 4 || lower.substr(lower.size() - 4) != ".stl") return c.fail("mjcf: only STL mesh files are supported in " + a.where);
    if (!read_stl_vertices(path, pts, c.err)) return false;
    if (name.empty()) { size_t s0 = f.find_last_of('/'); name = f.substr(s0 == std::string::npos ? 0 : s0 + 1); name = name.substr(0, name.size() - 4); }
  } else if (a.has("vertex")) {
    if (!parse_doubles(a.str("vertex"), pts) || pts.size() % 3 != 0) return c.fail("mjcf: mesh vertex must hold 3 numbers per vertex in " + a.where);
  } else return c.fail("mjcf: mesh needs file= or vertex= in " + a.where);
  double sc[3] = {1, 1, 1};
  if (a.vec("scale", sc, 3, 3) < 0) return false;
  for (size_t i = 0; i < pts.size(); i++) pts[i] *= sc[i % 3];
  std::vector<int> hull, tris;
  if (!convex_hull_vertices(pts, hull, c.err, &tris)) { c.err += " in " + a.where; return false; }
  // the hull's volume, centre of mass and principal axes of inertia (signed tetrahedra over its faces: mesh.cpp); the stored vertices are
  // relative to that frame
  std::vector<double> hv;
  for (int v : hull) for (int i = 0; i < 3; i++) hv.push_back(pts[3 * v + i]);
  double vol = 0, cen[3], I6[6], rb = 0;
  if (!mesh_mass_properties(hv, tris, vol, cen, I6)) return c.fail("mjcf: mesh hull has no volume in " + a.where);
  double Rm[9], w3[3];
  {
    const double I[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    double w[3], V[9];
    hm::eig3(I, w, V);
    int idx[3] = {0, 1, 2};  // moments descending, right-handed axes (as finish_body_inertia orders a body's)
    for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) if (w[idx[j]] > w[idx[i]]) std::swap(idx[i], idx[j]);
    for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) Rm[3 * r + k] = V[3 * r + idx[k]];
    double c0[3] = {Rm[0], Rm[3], Rm[6]}, c1[3] = {Rm[1], Rm[4], Rm[7]}, c2[3];
    hm::cross(c2, c0, c1);
    Rm[2] = c2[0]; Rm[5] = c2[1]; Rm[8] = c2[2];
    for (int k = 0; k < 3; k++) w3[k] = w[idx[k]];
    // (a tensor that is diagonal to rounding - a box, a symmetric part: the eigenvectors of equal or nearly equal moments are arbitrary -
    // takes the file's axes, permuted into descending order, moments equal to 1e-12 counted as equal)
    const double offd = std::fabs(I6[3]) + std::fabs(I6[4]) + std::fabs(I6[5]), tr = I6[0] + I6[1] + I6[2];
    if (offd <= 1e-12 * tr) {
      int pm[3] = {0, 1, 2};
      for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) if (I6[pm[j]] > I6[pm[i]] + 1e-12 * tr) std::swap(pm[i], pm[j]);
      for (int i = 0; i < 9; i++) Rm[i] = 0.0;
      Rm[3 * pm[0] + 0] = 1.0; Rm[3 * pm[1] + 1] = 1.0;
      double c0[3] = {Rm[0], Rm[3], Rm[6]}, c1[3] = {Rm[1], Rm[4], Rm[7]}, c2[3];
      hm::cross(c2, c0, c1);
      Rm[2] = c2[0]; Rm[5] = c2[1]; Rm[8] = c2[2];
      for (int k = 0; k < 3; k++) w3[k] = I6[pm[k]];
    }
  }
  double mq[4];
  hm::mat2quat(mq, Rm);
  hm::normalize4(mq);
  c.mesh_names[name] = m.nmesh;
  m.mesh_name.push_back(name);
  m.mesh_vertadr.push_back(m.nmeshvert);
  m.mesh_vertnum.push_back((int)hull.size());
  for (int v : hull) {
    const double d[3] = {pts[3 * v] - cen[0], pts[3 * v + 1] - cen[1], pts[3 * v + 2] - cen[2]};
    double q[3];  // R' d: coordinates along the principal axes
    for (int k = 0; k < 3; k++) q[k] = Rm[k] * d[0] + Rm[3 + k] * d[1] + Rm[6 + k] * d[2];
    rb = std::max(rb, hm::norm3(q));
    push3(m.mesh_vert, q);
  }
  {  // edge graph of the hull: neighbours of every vertex, ascending
    std::vector<std::set<int>> nb(hull.size());
    for (size_t t = 0; t + 2 < tris.size(); t += 3)
      for (int k = 0; k < 3; k++) { const int u = tris[t + k], w = tris[t + (k + 1) % 3]; nb[u].insert(w); nb[w].insert(u); }
    for (auto& sset : nb) {
      m.mesh_nbradr.push_back(m.nmeshnbr);
      m.mesh_nbrnum.push_back((int)sset.size());
      for (int w : sset) m.mesh_nbr.push_back(w);
      m.nmeshnbr += (int)sset.size();
    }
  }
  push3(c.mesh_center, cen);
  for (int i = 0; i < 4; i++) c.mesh_quat.push_back(mq[i]);
  c.mesh_volume.push_back(vol);
  for (int i = 0; i < 3; i++) c.mesh_inertia.push_back(w3[i]);
  c.mesh_rbound.push_back(rb);
  m.nmeshvert += (int)hull.size();
  m.nmesh++;
  return true;
}

bool read_assets(Ctx& c, const XmlNode& n) {
  Model& m = *c.m;
  for (auto& ch : n.children) {
    if (ch->name == "mesh") { if (!read_mesh(c, *ch)) return false; continue; }
    if (ch->name != "hfield") continue;
    Attrs a;
    a.c = &c; a.where = "<hfield>";
    overlay(a.a, *ch);
    int nrow = a.integer("nrow", 0), ncol = a.integer("ncol", 0);
    double size[4] = {0, 0, 0, 0};
    if (a.vec("size", size, 4, 4) < 0) return false;
    if (nrow < 2 || ncol < 2) return c.fail("mjcf: hfield needs nrow,ncol >= 2 (file-based hfields are not supported)");
    std::vector<double> el(nrow * ncol, 0.0);
    if (a.has("elevation")) {
      std::vector<double> t;
      if (!parse_doubles(a.str("elevation"), t) || (int)t.size() != nrow * ncol) return c.fail("mjcf: hfield elevation must have nrow*ncol values");
      // MJCF lists rows top-to-bottom in the file; store row 0 = min y like mjModel.hfield_data
      for (int r = 0; r < nrow; r++) for (int cc = 0; cc < ncol; cc++) el[r * ncol + cc] = t[(nrow - 1 - r) * ncol + cc];
      double mx = 0;
      for (double v : el) mx = std::max(mx, v);
      double mn = mx;
      for (double v : el) mn = std::min(mn, v);
      if (mx > mn) for (auto& v : el) v = (v - mn) / (mx - mn);
      else for (auto& v : el) v = 0;
    }
    c.hfield_names[a.str("name")] = std::to_string(m.nhfield);
    m.hfield_nrow.push_back(nrow); m.hfield_ncol.push_back(ncol);
    m.hfield_adr.push_back(m.nhfielddata);
    for (int i = 0; i < 4; i++) m.hfield_size.push_back(size[i]);
    for (double v : el) m.hfield_data.push_back(v);
    m.nhfielddata += nrow * ncol;
    m.nhfield++;
  }
  return c.err.empty();
}

bool read_tendons(Ctx& c, const XmlNode& n) {
  Model& m = *c.m;
  for (auto& ch : n.children) {
    if (ch->name != "fixed") return c.fail("mjcf: only <fixed> tendons are supported");
    Attrs a = merged(c, *ch, "tendon", "");
    m.tendon_name.push_back(a.str("name"));
    m.tendon_adr.push_back(m.nwrap);
    int num = 0;
    for (auto& w : ch->children) {
      if (w->name != "joint") continue;
      const std::string* jn = w->attr("joint");
      int j = jn ? find_name(m.jnt_name, *jn) : -1;
      if (j < 0) return c.fail("mjcf: unknown joint in tendon " + a.where);
      if (m.jnt_type[j] != JNT_HINGE && m.jnt_type[j] != JNT_SLIDE) return c.fail("mjcf: fixed tendon needs scalar joints");
      double coef = 1;
      if (const std::string* cs = w->attr("coef")) coef = strtod(cs->c_str(), nullptr);
      m.wrap_objid.push_back(j);
      m.wrap_prm.push_back(coef);
      m.nwrap++;
      num++;
    }
    m.tendon_num.push_back(num);
    double range[2] = {0, 0};
    int nr = a.vec("range", range, 2, 2);
    if (nr < 0) return false;
    int lt = a.tri("limited", -1);
    m.tendon_limited.push_back(lt >= 0 ? lt : ((c.autolimits && nr == 2) ? 1 : 0));
    m.tendon_range.push_back(range[0]); m.tendon_range.push_back(range[1]);
    m.tendon_margin.push_back(a.num("margin", 0));
    double sr[2] = {0.02, 1}, si[5] = {0.9, 0.95, 0.001, 0.5, 2};
    if (a.vec("solreflimit", sr, 2) < 0 || a.vec("solimplimit", si, 5) < 0) return false;
    m.tendon_solref_lim.push_back(sr[0]); m.tendon_solref_lim.push_back(sr[1]);
    for (int i = 0; i < 5; i++) m.tendon_solimp_lim.push_back(si[i]);
    if (a.num("stiffness", 0) != 0 || a.num("damping", 0) != 0 || a.num("frictionloss", 0) != 0)
      return c.fail("mjcf: tendon stiffness/damping/frictionloss are not supported");
    m.tendon_invweight0.push_back(0);
    m.tendon_length0.push_back(0);
    m.ntendon++;
  }
  return c.err.empty();
}

bool read_actuators(Ctx& c, const XmlNode& n) {
  Model& m = *c.m;
  for (auto& ch : n.children) {
    const std::string& k = ch->name;
    if (k != "motor" && k != "position" && k != "general") return c.fail("mjcf: actuator <" + k + "> is not supported (motor, position, general)");
    Attrs a = merged(c, *ch, "actuator", "");
    std::string jn = a.str("joint");
    int j = find_name(m.jnt_name, jn);
    if (jn.empty() || j < 0) return c.fail("mjcf: actuator needs a valid joint= in " + a.where);
    if (m.jnt_type[j] != JNT_HINGE && m.jnt_type[j] != JNT_SLIDE) return c.fail("mjcf: actuator joint must be hinge or slide");
    m.actuator_name.push_back(a.str("name"));
    m.actuator_trnid.push_back(j);
    double gear[6] = {1, 0, 0, 0, 0, 0};
    if (a.vec("gear", gear, 6) < 0) return false;
    m.actuator_gear.push_back(gear[0]);
    double cr[2] = {0, 0}, fr[2] = {0, 0};
    int ncr = a.vec("ctrlrange", cr, 2, 2), nfr = a.vec("forcerange", fr, 2, 2);
    if (ncr < 0 || nfr < 0) return false;
    int cl = a.tri("ctrllimited", -1), fl = a.tri("forcelimited", -1);
    m.actuator_ctrllimited.push_back(cl >= 0 ? cl : ((c.autolimits && ncr == 2) ? 1 : 0));
    m.actuator_forcelimited.push_back(fl >= 0 ? fl : ((c.autolimits && nfr == 2) ? 1 : 0));
    m.actuator_ctrlrange.push_back(cr[0]); m.actuator_ctrlrange.push_back(cr[1]);
    m.actuator_forcerange.push_back(fr[0]); m.actuator_forcerange.push_back(fr[1]);
    double gain = 1, bias[3] = {0, 0, 0};
    if (k == "position") {
      double kp = a.num("kp", 1), kv = a.num("kv", 0);
      gain = kp; bias[1] = -kp; bias[2] = -kv;
    } else if (k == "general") {
      double gp[10] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0}, bp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (a.vec("gainprm", gp, 10) < 0 || a.vec("biasprm", bp, 10) < 0) return false;
      if (a.str("gaintype", "fixed") != "fixed") return c.fail("mjcf: only gaintype fixed is supported");
      std::string bt = a.str("biastype", "none");
      if (bt != "none" && bt != "affine") return c.fail("mjcf: only biastype none/affine is supported");
      if (a.str("dyntype", "none") != "none") return c.fail("mjcf: actuator dynamics are not supported");
      gain = gp[0];
      if (bt == "affine") { bias[0] = bp[0]; bias[1] = bp[1]; bias[2] = bp[2]; }
    }
    m.actuator_gainprm.push_back(gain);
    for (int i = 0; i < 3; i++) m.actuator_biasprm.push_back(bias[i]);
    m.nu++;
  }
  return c.err.empty();
}

}  // namespace
void sort_pairs(Model& m, int order) {
  std::vector<int> idx(m.pair_geom1.size());
  for (size_t i = 0; i < idx.size(); i++) idx[i] = (int)i;
  auto key = [&](int p) {
    const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
    const int lo = std::min(g1, g2), hi = std::max(g1, g2);
    if (order == 0) return std::array<int, 4>{lo, hi, 0, 0};
    const int b1 = m.geom_bodyid[lo], b2 = m.geom_bodyid[hi];  // (geoms are numbered body by body: the lower geom sits on the lower body)
    return b1 <= b2 ? std::array<int, 4>{b1, b2, lo, hi} : std::array<int, 4>{b2, b1, hi, lo};
  };
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key(a) < key(b); });
  veci g1(idx.size()), g2(idx.size());
  for (size_t i = 0; i < idx.size(); i++) { g1[i] = m.pair_geom1[idx[i]]; g2[i] = m.pair_geom2[idx[i]]; }
  m.pair_geom1.swap(g1); m.pair_geom2.swap(g2);
  m.pair_order = order ? 1 : 0;
}
namespace {
void build_pairs(Model& m) {
  m.pair_geom1.clear(); m.pair_geom2.clear();
  bool filterparent = !(m.disableflags & DSBL_FILTERPARENT);
  for (int g1 = 0; g1 < m.ngeom; g1++)
    for (int g2 = g1 + 1; g2 < m.ngeom; g2++) {
      int b1 = m.geom_bodyid[g1], b2 = m.geom_bodyid[g2];
      int w1 = m.body_weldid[b1], w2 = m.body_weldid[b2];
      if (w1 == w2) continue;
      int wp1 = m.body_weldid[m.body_parentid[w1]], wp2 = m.body_weldid[m.body_parentid[w2]];
      if (filterparent && w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) continue;
      bool excluded = false;
      for (int e = 0; e < m.nexclude; e++)
        if ((m.exclude_body1[e] == b1 && m.exclude_body2[e] == b2) || (m.exclude_body1[e] == b2 && m.exclude_body2[e] == b1)) excluded = true;
      if (excluded) continue;
      if (!((m.geom_contype[g1] & m.geom_conaffinity[g2]) || (m.geom_contype[g2] & m.geom_conaffinity[g1]))) continue;
      int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
      int a = g1, b = g2;
      if (t1 > t2) { std::swap(a, b); std::swap(t1, t2); }
      if (t2 == GEOM_PLANE || t2 == GEOM_HFIELD) continue;  // plane-plane, plane-hfield, hfield-hfield: no collider
      if (t1 == GEOM_CYLINDER || t2 == GEOM_CYLINDER || t1 == GEOM_ELLIPSOID || t2 == GEOM_ELLIPSOID || t1 == GEOM_BOX || t2 == GEOM_BOX) {
        m.pair_unsupported = "collision between geoms '" + m.geom_name[a] + "' and '" + m.geom_name[b] + "': cylinder / ellipsoid / box colliders are not implemented";
        continue;
      }
      m.pair_geom1.push_back(a);
      m.pair_geom2.push_back(b);
    }
  m.npair = (int)m.pair_geom1.size();
  sort_pairs(m, 1);
}

void expand_includes(XmlNode& n, const std::string& basedir, std::string& err, int depth) {
  if (depth > 8) { err = "mjcf: include nesting too deep"; return; }
  for (size_t i = 0; i < n.children.size();) {
    XmlNode* ch = n.children[i].get();
    if (ch->name == "include") {
      const std::string* f = ch->attr("file");
      if (!f) { err = "mjcf: <include> without file"; return; }
      std::string path = (f->size() && (*f)[0] == '/') ? *f : basedir + *f;
      std::ifstream in(path, std::ios::binary);
      if (!in) { err = "mjcf: cannot open include " + path; return; }
      std::stringstream ss;
      ss << in.rdbuf();
      std::string text = ss.str();
      XmlParser p(text);
      auto root = p.parse(err);
      if (!root) return;
      expand_includes(*root, basedir, err, depth + 1);
      if (!err.empty()) return;
      n.children.erase(n.children.begin() + i);
      size_t k = i;
      for (auto& rc : root->children) n.children.insert(n.children.begin() + (k++), std::move(rc));
      i = k;
    } else {
      expand_includes(*ch, basedir, err, depth);
      if (!err.empty()) return;
      i++;
    }
  }
}

bool compile_root(Ctx& c, XmlNode& root) {
  Model& m = *c.m;
  if (root.name != "mujoco") return c.fail("mjcf: root element must be <mujoco>");
  expand_includes(root, c.basedir, c.err, 0);
  if (!c.err.empty()) return false;
  m.timestep = 0.002;
  // pass 1: compiler, option, defaults, assets (may appear in any order, several times)
  for (auto& ch : root.children) {
    if (ch->name == "compiler") {
      if (const std::string* s = ch->attr("angle")) c.degree = (*s != "radian");
      if (const std::string* s = ch->attr("eulerseq")) { if (s->size() != 3) return c.fail("mjcf: bad eulerseq"); c.eulerseq = *s; }
      if (const std::string* s = ch->attr("autolimits")) c.autolimits = (*s == "true");
      if (const std::string* s = ch->attr("coordinate")) if (*s != "local") return c.fail("mjcf: only local coordinates are supported");
      if (const std::string* s = ch->attr("meshdir")) { c.meshdir = *s; if (!c.meshdir.empty() && c.meshdir.back() != '/') c.meshdir += '/'; }
    } else if (ch->name == "option") {
      if (!read_option(c, *ch)) return false;
    } else if (ch->name == "default") {
      read_defaults(c, *ch, "", true);
      if (!c.err.empty()) return false;
    } else if (ch->name == "asset") {
      if (!read_assets(c, *ch)) return false;
    }
  }
  // pass 2: kinematic tree
  // (several <worldbody> elements - an included file's and the including file's - are one world body: children in document order)
  bool world_done = false;
  XmlNode* world = nullptr;
  for (auto& ch : root.children) {
    if (ch->name != "worldbody") continue;
    if (!world) { world = ch.get(); continue; }
    for (auto& gc : ch->children) world->children.push_back(std::move(gc));
    ch->children.clear();
  }
  if (world) {
    if (!compile_body(c, *world, -1, "", 0)) return false;
    world_done = true;
  }
  if (!world_done) return c.fail("mjcf: missing <worldbody>");
  // a moving body needs mass: its own or that of the jointless bodies welded to it (the reference's robot hangs its torso,
  // the only massive part of the trunk, off a massless free-floating link: simulation/assets/humanoid.xml:16-21)
  {
    std::vector<double> wmass(m.nbody, 0.0);
    for (int b = 1; b < m.nbody; b++) wmass[m.body_weldid[b]] += m.body_mass[b];
    for (int b = 1; b < m.nbody; b++)
      if (m.body_dofnum[b] > 0 && !(wmass[b] > 0)) return c.fail("mjcf: moving body '" + m.body_name[b] + "' has no mass");
  }
  // pass 3: everything that refers to bodies and joints by name
  for (auto& ch : root.children) {
    if (ch->name == "contact") {
      for (auto& e : ch->children) {
        if (e->name == "pair") return c.fail("mjcf: explicit contact <pair> is not supported");
        if (e->name != "exclude") continue;
        const std::string *b1 = e->attr("body1"), *b2 = e->attr("body2");
        int i1 = b1 ? find_name(m.body_name, *b1) : -1, i2 = b2 ? find_name(m.body_name, *b2) : -1;
        if (i1 < 0 || i2 < 0) return c.fail("mjcf: unknown body in <exclude>");
        m.exclude_body1.push_back(i1); m.exclude_body2.push_back(i2);
        m.nexclude++;
      }
    } else if (ch->name == "tendon") {
      if (!read_tendons(c, *ch)) return false;
    } else if (ch->name == "actuator") {
      if (!read_actuators(c, *ch)) return false;
    } else if (ch->name == "equality") {
      if (!ch->children.empty()) return c.fail("mjcf: equality constraints are not supported");
    }
  }
  // Madr / nM
  m.dof_Madr.assign(m.nv, 0);
  m.nM = 0;
  for (int i = 0; i < m.nv; i++) {
    m.dof_Madr[i] = m.nM;
    for (int j = i; j >= 0; j = m.dof_parentid[j]) m.nM++;
  }
  m.dof_invweight0.assign(m.nv, 0);
  m.dof_M0.assign(m.nv, 0);
  // keyframes
  for (auto& ch : root.children) {
    if (ch->name != "keyframe") continue;
    for (auto& k : ch->children) {
      if (k->name != "key") continue;
      std::vector<double> q(m.qpos0);
      if (const std::string* s = k->attr("qpos")) {
        std::vector<double> t;
        if (!parse_doubles(*s, t) || (int)t.size() != m.nq) return c.fail("mjcf: keyframe qpos must have nq values");
        q = t;
      }
      m.key_name.push_back(k->attr("name") ? *k->attr("name") : "");
      for (double v : q) m.key_qpos.push_back(v);
      m.nkey++;
    }
  }
  build_pairs(m);
  if (!m.pair_unsupported.empty()) return c.fail("mjcf: " + m.pair_unsupported);
  if (!validate_model(m, c.err)) return false;
  return set_const(m, c.err);
}

}  // namespace

bool compile_mjcf_string(const std::string& xml, Model& m, std::string& err) {
  XmlParser p(xml);
  auto root = p.parse(err);
  if (!root) return false;
  m = Model();
  Ctx c;
  c.m = &m;
  if (!compile_root(c, *root)) { err = c.err.empty() ? "mjcf: compile failed" : c.err; return false; }
  return true;
}

bool compile_mjcf_file(const std::string& path, Model& m, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = "cannot open model file: " + path; return false; }
  std::stringstream ss;
  ss << f.rdbuf();
  std::string text = ss.str();
  XmlParser p(text);
  auto root = p.parse(err);
  if (!root) return false;
  m = Model();
  Ctx c;
  c.m = &m;
  size_t slash = path.find_last_of('/');
  c.basedir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
  if (!compile_root(c, *root)) { err = c.err.empty() ? "mjcf: compile failed" : c.err; return false; }
  return true;
}

}  // namespace hb
