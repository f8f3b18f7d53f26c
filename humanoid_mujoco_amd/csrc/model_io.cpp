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
  // consistency of the sizes the kernels rely on
  if (m.nbody <= 0 || (int)m.body_parentid.size() != m.nbody || (int)m.qpos0.size() != m.nq ||
      (int)m.dof_Madr.size() != m.nv || (int)m.geom_type.size() != m.ngeom ||
      (int)m.pair_geom1.size() != m.npair) {
    err = "hbm: inconsistent sizes";
    return false;
  }
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
