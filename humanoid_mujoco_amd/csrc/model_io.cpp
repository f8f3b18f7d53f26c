// model_io.cpp — ".hbm" text serialisation of hb::Model.
// Plays the role of mj_saveModel / mj_loadModel (reference: simulation/mujoco/include/mujoco/
// mujoco.h:159-163) for this engine.  Format: first line "HBM1", then one record per line:
//   i <name> <int>          d <name> <double>
//   I <name> <n> <ints…>    D <name> <n> <doubles…>    S <name> <n> <tokens…>
// terminated by "END".  Doubles use %.17g so a round trip is bit exact.  Unknown records are
// skipped on load (forward compatible); the oracle has its own independent parser.
#include "hb_model.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace hb {

namespace {
struct Writer {
  std::ostringstream os;
  char buf[64];
  void operator()(const char* n, int& v) { os << "i " << n << " " << v << "\n"; }
  void operator()(const char* n, double& v) {
    snprintf(buf, sizeof buf, "%.17g", v);
    os << "d " << n << " " << buf << "\n";
  }
  void operator()(const char* n, double* v, int cnt) {
    os << "D " << n << " " << cnt;
    for (int i = 0; i < cnt; i++) { snprintf(buf, sizeof buf, " %.17g", v[i]); os << buf; }
    os << "\n";
  }
  void operator()(const char* n, veci& v) {
    os << "I " << n << " " << v.size();
    for (int x : v) os << " " << x;
    os << "\n";
  }
  void operator()(const char* n, vecd& v) {
    os << "D " << n << " " << v.size();
    for (double x : v) { snprintf(buf, sizeof buf, " %.17g", x); os << buf; }
    os << "\n";
  }
  void operator()(const char* n, std::vector<std::string>& v) {
    os << "S " << n << " " << v.size();
    for (auto s : v) {
      if (s.empty()) s = "-";
      for (auto& c : s) if (c == ' ' || c == '\t' || c == '\n') c = '_';
      os << " " << s;
    }
    os << "\n";
  }
};

struct Record {
  char kind;
  std::vector<std::string> tok;
};

struct Reader {
  std::map<std::string, Record>* recs;
  std::string err;
  Record* find(const char* n, char kind) {
    auto it = recs->find(n);
    if (it == recs->end()) return nullptr;  // missing field keeps its default
    if (it->second.kind != kind) { err = std::string("hbm: wrong record kind for ") + n; return nullptr; }
    return &it->second;
  }
  void operator()(const char* n, int& v) { if (auto r = find(n, 'i')) v = atoi(r->tok[0].c_str()); }
  void operator()(const char* n, double& v) { if (auto r = find(n, 'd')) v = strtod(r->tok[0].c_str(), nullptr); }
  void operator()(const char* n, double* v, int cnt) {
    if (auto r = find(n, 'D')) {
      if ((int)r->tok.size() != cnt) { err = std::string("hbm: bad length for ") + n; return; }
      for (int i = 0; i < cnt; i++) v[i] = strtod(r->tok[i].c_str(), nullptr);
    }
  }
  void operator()(const char* n, veci& v) {
    if (auto r = find(n, 'I')) { v.resize(r->tok.size()); for (size_t i = 0; i < v.size(); i++) v[i] = atoi(r->tok[i].c_str()); }
  }
  void operator()(const char* n, vecd& v) {
    if (auto r = find(n, 'D')) { v.resize(r->tok.size()); for (size_t i = 0; i < v.size(); i++) v[i] = strtod(r->tok[i].c_str(), nullptr); }
  }
  void operator()(const char* n, std::vector<std::string>& v) {
    if (auto r = find(n, 'S')) { v = r->tok; for (auto& s : v) if (s == "-") s.clear(); }
  }
};
}  // namespace

bool save_hbm(const Model& m, const std::string& path, std::string& err) {
  Writer w;
  w.os << "HBM1\n";
  const_cast<Model&>(m).visit(w);
  w.os << "END\n";
  std::ofstream f(path, std::ios::binary);
  if (!f) { err = "cannot open for writing: " + path; return false; }
  f << w.os.str();
  return (bool)f;
}

bool load_hbm_string(const std::string& text, Model& m, std::string& err) {
  std::istringstream is(text);
  std::string line;
  if (!std::getline(is, line) || line.compare(0, 4, "HBM1") != 0) { err = "not an HBM1 model"; return false; }
  std::map<std::string, Record> recs;
  bool ended = false;
  while (std::getline(is, line)) {
    if (line.compare(0, 3, "END") == 0) { ended = true; break; }
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ls(line);
    std::string kind, name;
    ls >> kind >> name;
    if (kind.size() != 1 || name.empty()) { err = "hbm: malformed record: " + line.substr(0, 40); return false; }
    Record r;
    r.kind = kind[0];
    std::string t;
    if (r.kind == 'i' || r.kind == 'd') {
      if (!(ls >> t)) { err = "hbm: missing value for " + name; return false; }
      r.tok.push_back(t);
    } else {
      size_t n = 0;
      if (!(ls >> n)) { err = "hbm: missing count for " + name; return false; }
      if (n > line.size()) { err = "hbm: count mismatch for " + name; return false; }  // every entry takes at least two characters
      r.tok.reserve(n);
      while (ls >> t) r.tok.push_back(t);
      if (r.tok.size() != n) { err = "hbm: count mismatch for " + name; return false; }
    }
    recs[name] = r;
  }
  if (!ended) { err = "hbm: truncated file (no END)"; return false; }
  m = Model();
  Reader rd;
  rd.recs = &recs;
  m.visit(rd);
  if (!rd.err.empty()) { err = rd.err; return false; }
  return validate_model(m, err);
}

// Every array length against its size field and every id / address against its range: what build_device_model and the
// kernels index without further checks.  Run on every loaded .hbm (a truncated or edited file must give an error string,
// not a host out-of-bounds read or a GPU fault) and on every compiled MJCF.
bool validate_model(const Model& m, std::string& err) {
  auto bad = [&](const std::string& what) { err = "model: " + what; return false; };
  if (m.nbody < 1 || m.nbody > 64) return bad("nbody must be 1..64");
  if (m.nq < 0 || m.nv < 0 || m.nu < 0 || m.njnt < 0 || m.ngeom < 0 || m.ntendon < 0 || m.nwrap < 0 || m.nkey < 0 || m.nexclude < 0 || m.npair < 0 ||
      m.nhfield < 0 || m.nhfielddata < 0 || m.nM < 0 || m.nmesh < 0 || m.nmeshvert < 0 || m.nmeshnbr < 0)
    return bad("negative size");
  if (m.nq > 4096 || m.nv > 4096 || m.nu > 4096 || m.njnt > 4096 || m.ngeom > 4096 || m.nwrap > 65536 || m.npair > (1 << 20)) return bad("size out of range");
  struct { const char* name; size_t have, want; } lens[] = {
#define HB_LEN(x, n) {#x, m.x.size(), (size_t)(n)}
      HB_LEN(body_parentid, m.nbody), HB_LEN(body_rootid, m.nbody), HB_LEN(body_weldid, m.nbody), HB_LEN(body_jntnum, m.nbody), HB_LEN(body_jntadr, m.nbody),
      HB_LEN(body_dofnum, m.nbody), HB_LEN(body_dofadr, m.nbody), HB_LEN(body_geomnum, m.nbody), HB_LEN(body_geomadr, m.nbody), HB_LEN(body_depth, m.nbody),
      HB_LEN(body_pos, 3 * m.nbody), HB_LEN(body_quat, 4 * m.nbody), HB_LEN(body_ipos, 3 * m.nbody), HB_LEN(body_iquat, 4 * m.nbody), HB_LEN(body_mass, m.nbody),
      HB_LEN(body_subtreemass, m.nbody), HB_LEN(body_inertia, 3 * m.nbody), HB_LEN(body_invweight0, 2 * m.nbody),
      HB_LEN(jnt_type, m.njnt), HB_LEN(jnt_qposadr, m.njnt), HB_LEN(jnt_dofadr, m.njnt), HB_LEN(jnt_bodyid, m.njnt), HB_LEN(jnt_limited, m.njnt),
      HB_LEN(jnt_pos, 3 * m.njnt), HB_LEN(jnt_axis, 3 * m.njnt), HB_LEN(jnt_stiffness, m.njnt), HB_LEN(jnt_range, 2 * m.njnt), HB_LEN(jnt_margin, m.njnt),
      HB_LEN(jnt_solref, 2 * m.njnt), HB_LEN(jnt_solimp, 5 * m.njnt),
      HB_LEN(dof_bodyid, m.nv), HB_LEN(dof_jntid, m.nv), HB_LEN(dof_parentid, m.nv), HB_LEN(dof_Madr, m.nv), HB_LEN(dof_armature, m.nv), HB_LEN(dof_damping, m.nv),
      HB_LEN(dof_frictionloss, m.nv), HB_LEN(dof_invweight0, m.nv), HB_LEN(dof_M0, m.nv),
      HB_LEN(geom_type, m.ngeom), HB_LEN(geom_bodyid, m.ngeom), HB_LEN(geom_contype, m.ngeom), HB_LEN(geom_conaffinity, m.ngeom), HB_LEN(geom_condim, m.ngeom),
      HB_LEN(geom_priority, m.ngeom), HB_LEN(geom_dataid, m.ngeom), HB_LEN(geom_size, 3 * m.ngeom), HB_LEN(geom_pos, 3 * m.ngeom), HB_LEN(geom_quat, 4 * m.ngeom),
      HB_LEN(geom_rbound, m.ngeom), HB_LEN(geom_friction, 3 * m.ngeom), HB_LEN(geom_solmix, m.ngeom), HB_LEN(geom_solref, 2 * m.ngeom), HB_LEN(geom_solimp, 5 * m.ngeom),
      HB_LEN(geom_margin, m.ngeom), HB_LEN(geom_gap, m.ngeom),
      HB_LEN(hfield_nrow, m.nhfield), HB_LEN(hfield_ncol, m.nhfield), HB_LEN(hfield_adr, m.nhfield), HB_LEN(hfield_size, 4 * m.nhfield), HB_LEN(hfield_data, m.nhfielddata),
      HB_LEN(mesh_vertadr, m.nmesh), HB_LEN(mesh_vertnum, m.nmesh), HB_LEN(mesh_vert, 3 * m.nmeshvert), HB_LEN(mesh_name, m.nmesh),
      HB_LEN(mesh_nbradr, m.nmeshvert), HB_LEN(mesh_nbrnum, m.nmeshvert), HB_LEN(mesh_nbr, m.nmeshnbr),
      HB_LEN(tendon_adr, m.ntendon), HB_LEN(tendon_num, m.ntendon), HB_LEN(tendon_limited, m.ntendon), HB_LEN(wrap_objid, m.nwrap), HB_LEN(tendon_range, 2 * m.ntendon),
      HB_LEN(tendon_margin, m.ntendon), HB_LEN(tendon_solref_lim, 2 * m.ntendon), HB_LEN(tendon_solimp_lim, 5 * m.ntendon), HB_LEN(tendon_invweight0, m.ntendon),
      HB_LEN(tendon_length0, m.ntendon), HB_LEN(wrap_prm, m.nwrap),
      HB_LEN(actuator_trnid, m.nu), HB_LEN(actuator_ctrllimited, m.nu), HB_LEN(actuator_forcelimited, m.nu), HB_LEN(actuator_gear, m.nu),
      HB_LEN(actuator_ctrlrange, 2 * m.nu), HB_LEN(actuator_forcerange, 2 * m.nu), HB_LEN(actuator_gainprm, m.nu), HB_LEN(actuator_biasprm, 3 * m.nu),
      HB_LEN(exclude_body1, m.nexclude), HB_LEN(exclude_body2, m.nexclude), HB_LEN(pair_geom1, m.npair), HB_LEN(pair_geom2, m.npair),
      HB_LEN(qpos0, m.nq), HB_LEN(qpos_spring, m.nq), HB_LEN(key_qpos, (size_t)m.nkey * m.nq),
      HB_LEN(body_name, m.nbody), HB_LEN(jnt_name, m.njnt), HB_LEN(geom_name, m.ngeom), HB_LEN(tendon_name, m.ntendon), HB_LEN(actuator_name, m.nu),
      HB_LEN(key_name, m.nkey),
#undef HB_LEN
  };
  for (auto& l : lens)
    if (l.have != l.want) return bad(std::string("array ") + l.name + " has " + std::to_string(l.have) + " entries, its size field says " + std::to_string(l.want));
  auto in = [](int v, int lo, int hi) { return v >= lo && v < hi; };  // lo <= v < hi
  int nq = 0, nv = 0;
  for (int j = 0; j < m.njnt; j++) {
    const int t = m.jnt_type[j];
    if (t != JNT_FREE && t != JNT_BALL && t != JNT_SLIDE && t != JNT_HINGE) return bad("joint type out of range");
    if (m.jnt_qposadr[j] != nq || m.jnt_dofadr[j] != nv) return bad("joint addresses are not consecutive");
    nq += t == JNT_FREE ? 7 : (t == JNT_BALL ? 4 : 1);
    nv += t == JNT_FREE ? 6 : (t == JNT_BALL ? 3 : 1);
    if (!in(m.jnt_bodyid[j], 1, m.nbody)) return bad("jnt_bodyid out of range");
  }
  if (nq != m.nq || nv != m.nv) return bad("nq / nv do not match the joints");
  if (m.body_parentid[0] != 0) return bad("the world body must be its own parent");
  for (int b = 0; b < m.nbody; b++) {
    if (b > 0 && !in(m.body_parentid[b], 0, b)) return bad("body_parentid must precede the body");
    if (!in(m.body_rootid[b], 0, m.nbody) || !in(m.body_weldid[b], 0, m.nbody)) return bad("body root / weld id out of range");
    const int jn = m.body_jntnum[b], dn = m.body_dofnum[b], gn = m.body_geomnum[b];
    if (jn < 0 || dn < 0 || gn < 0) return bad("negative per-body count");
    if (jn > 0 && (m.body_jntadr[b] < 0 || m.body_jntadr[b] + jn > m.njnt)) return bad("body joint range out of bounds");
    if (dn > 0 && (m.body_dofadr[b] < 0 || m.body_dofadr[b] + dn > m.nv)) return bad("body dof range out of bounds");
    if (gn > 0 && (m.body_geomadr[b] < 0 || m.body_geomadr[b] + gn > m.ngeom)) return bad("body geom range out of bounds");
    for (int j = 0; j < jn; j++) if (m.jnt_bodyid[m.body_jntadr[b] + j] != b) return bad("jnt_bodyid does not match the body's joint range");
  }
  int nM = 0;
  for (int d = 0; d < m.nv; d++) {
    if (!in(m.dof_bodyid[d], 1, m.nbody) || !in(m.dof_jntid[d], 0, m.njnt)) return bad("dof body / joint id out of range");
    if (!in(m.dof_parentid[d], -1, d)) return bad("dof_parentid must be -1 or precede the dof");
    if (m.dof_Madr[d] != nM) return bad("dof_Madr is not the ancestor-chain layout");
    for (int j = d; j >= 0; j = m.dof_parentid[j]) nM++;
  }
  if (nM != m.nM) return bad("nM does not match the dof tree");
  for (int g = 0; g < m.ngeom; g++) {
    if (!in(m.geom_bodyid[g], 0, m.nbody)) return bad("geom_bodyid out of range");
    const int t = m.geom_type[g];
    if (t != GEOM_PLANE && t != GEOM_HFIELD && t != GEOM_SPHERE && t != GEOM_CAPSULE && t != GEOM_MESH && t != GEOM_CYLINDER) return bad("geom type not supported");
    if (t == GEOM_HFIELD && !in(m.geom_dataid[g], 0, m.nhfield)) return bad("height-field geom without a valid field");
    if (t == GEOM_MESH && !in(m.geom_dataid[g], 0, m.nmesh)) return bad("mesh geom without a valid mesh");
    const int cd = m.geom_condim[g];
    if (cd != 1 && cd != 3 && cd != 4 && cd != 6) return bad("geom condim must be 1, 3, 4 or 6");
  }
  for (int p = 0; p < m.npair; p++)
    if (!in(m.pair_geom1[p], 0, m.ngeom) || !in(m.pair_geom2[p], 0, m.ngeom)) return bad("pair geom out of range");
  for (int k = 0; k < m.nmesh; k++)
    for (int v = 0; v < m.mesh_vertnum[k] && m.mesh_vertadr[k] >= 0 && m.mesh_vertadr[k] + m.mesh_vertnum[k] <= m.nmeshvert; v++) {
      const int g = m.mesh_vertadr[k] + v;
      if (m.mesh_nbrnum[g] < 3 || m.mesh_nbradr[g] < 0 || m.mesh_nbradr[g] + m.mesh_nbrnum[g] > m.nmeshnbr) return bad("mesh edge graph out of bounds");
      for (int i = 0; i < m.mesh_nbrnum[g]; i++) if (!in(m.mesh_nbr[m.mesh_nbradr[g] + i], 0, m.mesh_vertnum[k])) return bad("mesh edge graph: neighbour out of range");
    }
  for (int k = 0; k < m.nmesh; k++)
    if (m.mesh_vertnum[k] < 4 || m.mesh_vertadr[k] < 0 || m.mesh_vertadr[k] + m.mesh_vertnum[k] > m.nmeshvert) return bad("mesh vertex range out of bounds");
  for (int p = 0; p < m.npair; p++) {
    const int t1 = m.geom_type[m.pair_geom1[p]], t2 = m.geom_type[m.pair_geom2[p]];
    if (t1 == GEOM_CYLINDER || t2 == GEOM_CYLINDER) return bad("cylinder geoms have no collider");
  }
  for (int h = 0; h < m.nhfield; h++) {
    if (m.hfield_nrow[h] < 2 || m.hfield_ncol[h] < 2 || m.hfield_nrow[h] > 4096 || m.hfield_ncol[h] > 4096) return bad("height-field dimensions out of range");
    if (m.hfield_adr[h] < 0 || (long long)m.hfield_adr[h] + (long long)m.hfield_nrow[h] * m.hfield_ncol[h] > m.nhfielddata) return bad("height-field data out of bounds");
  }
  for (int t = 0; t < m.ntendon; t++)
    if (m.tendon_num[t] < 0 || m.tendon_adr[t] < 0 || m.tendon_adr[t] + m.tendon_num[t] > m.nwrap) return bad("tendon wrap range out of bounds");
  for (int w = 0; w < m.nwrap; w++) {
    if (!in(m.wrap_objid[w], 0, m.njnt)) return bad("wrap_objid out of range");
    if (m.jnt_type[m.wrap_objid[w]] != JNT_SLIDE && m.jnt_type[m.wrap_objid[w]] != JNT_HINGE) return bad("fixed tendon over a non-scalar joint");
  }
  for (int a = 0; a < m.nu; a++) {
    if (!in(m.actuator_trnid[a], 0, m.njnt)) return bad("actuator_trnid out of range");
    if (m.jnt_type[m.actuator_trnid[a]] != JNT_SLIDE && m.jnt_type[m.actuator_trnid[a]] != JNT_HINGE) return bad("actuator on a non-scalar joint");
  }
  for (int e = 0; e < m.nexclude; e++)
    if (!in(m.exclude_body1[e], 0, m.nbody) || !in(m.exclude_body2[e], 0, m.nbody)) return bad("exclude body out of range");
  if (!(m.timestep > 0) || !(m.impratio > 0) || !(m.meaninertia > 0) || m.iterations < 0 || m.ls_iterations < 0) return bad("option out of range");
  return true;
}

bool load_hbm(const std::string& path, Model& m, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = "cannot open model file: " + path; return false; }
  std::stringstream ss;
  ss << f.rdbuf();
  return load_hbm_string(ss.str(), m, err);
}

}  // namespace hb
