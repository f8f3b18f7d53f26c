// hb_api.cpp — the C-ABI of libhb.so (include/hb.h): model handles, device model tables,
// batches of environments on one GPU, and the launch plumbing around the kernel translation units (hb_step.hip, hb_narrow.hip, hb_env.hip).
//
// Host C++ only; no PyTorch.  One hb_batch owns one HIP stream and all device memory of its
// envs; the model is immutable and shareable (reference ownership rules: SURVEY.md §8b).
#include "../../include/hb.h"
#include "hb_device.hpp"
#include "hb_launch.hpp"
#include "hb_model.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace hb;

struct hb_model {
  Model m;
};

namespace {

void set_err(char* err, int err_sz, const std::string& s) {
  if (err && err_sz > 0) { snprintf(err, err_sz, "%s", s.c_str()); }
}

// flat table builder: ints and floats pushed into two arrays; offsets resolved after upload
struct TableBuilder {
  std::vector<int> iv;
  std::vector<float> fv;
  std::vector<unsigned long long> uv;
  size_t addi(const std::vector<int>& v) { size_t o = iv.size(); iv.insert(iv.end(), v.begin(), v.end()); if (v.empty()) iv.push_back(0); return o; }
  size_t addf(const std::vector<double>& v) { size_t o = fv.size(); for (double x : v) fv.push_back((float)x); if (v.empty()) fv.push_back(0.f); return o; }
  size_t addraw(const std::vector<float>& v) { while (fv.size() % 4) fv.push_back(0.f); size_t o = fv.size(); fv.insert(fv.end(), v.begin(), v.end()); return o; }
  size_t addu(const std::vector<unsigned long long>& v) { size_t o = uv.size(); uv.insert(uv.end(), v.begin(), v.end()); if (v.empty()) uv.push_back(0); return o; }
};

struct DeviceModel {
  DevModel dm;
  int* d_int = nullptr;
  float* d_flt = nullptr;
  unsigned long long* d_u64 = nullptr;
  float* d_qpos_src = nullptr;  // qpos0 followed by keyframes, fp32
  DevModel* d_dm = nullptr;     // device copy of dm (the step kernel reads the tables through it)
  DevModel* d_dm_fast = nullptr;  // variant 2 only: the same model with the variant-1 LDS layout (fast step kernel of the staged step)
  int fast_lds_floats = 0;
  bool sized_h27 = false;  // sizes and LDS layout equal kSizedHumanoid27's: the size-specialised step kernel applies
  bool sized_team = false; // the fast layout equals kSizedTeamV1's
  // observation order tables (device pointers): joint order and, when it exists, actuator order (hb_env_config.obs_actuator_order)
  const int *obs_jnt_joint = nullptr, *obs_src_joint = nullptr, *obs_jnt_act = nullptr, *obs_src_act = nullptr;
  bool has_act_order = false;
  ~DeviceModel() {
    if (d_int) (void)hipFree(d_int);
    if (d_flt) (void)hipFree(d_flt);
    if (d_u64) (void)hipFree(d_u64);
    if (d_qpos_src) (void)hipFree(d_qpos_src);
    if (d_dm) (void)hipFree(d_dm);
    if (d_dm_fast) (void)hipFree(d_dm_fast);
  }
};

// contact parameter mixing per candidate pair (mj_contactParam restatement; static per pair)
void mix_pair(const Model& m, int g1, int g2, int& dim, double* fr, double* solref, double* solimp, double& margin, double& gap) {
  dim = std::max(m.geom_condim[g1], m.geom_condim[g2]);
  int p1 = m.geom_priority[g1], p2 = m.geom_priority[g2];
  if (p1 != p2) {
    int g = p1 > p2 ? g1 : g2;
    dim = m.geom_condim[g];
    for (int i = 0; i < 3; i++) fr[i] = m.geom_friction[3 * g + i];
    for (int i = 0; i < 2; i++) solref[i] = m.geom_solref[2 * g + i];
    for (int i = 0; i < 5; i++) solimp[i] = m.geom_solimp[5 * g + i];
  } else {
    for (int i = 0; i < 3; i++) fr[i] = std::max(m.geom_friction[3 * g1 + i], m.geom_friction[3 * g2 + i]);
    double s1 = m.geom_solmix[g1], s2 = m.geom_solmix[g2], mix;
    const double MINVAL = 1e-15;
    if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
    else mix = s1 < MINVAL ? 0.0 : 1.0;
    const double *r1 = &m.geom_solref[2 * g1], *r2 = &m.geom_solref[2 * g2];
    if (r1[0] > 0 && r2[0] > 0) for (int i = 0; i < 2; i++) solref[i] = mix * r1[i] + (1 - mix) * r2[i];
    else for (int i = 0; i < 2; i++) solref[i] = std::min(r1[i], r2[i]);
    for (int i = 0; i < 5; i++) solimp[i] = mix * m.geom_solimp[5 * g1 + i] + (1 - mix) * m.geom_solimp[5 * g2 + i];
  }
  margin = std::max(m.geom_margin[g1], m.geom_margin[g2]);
  gap = std::max(m.geom_gap[g1], m.geom_gap[g2]);
}

// Which instantiation of the step kernel a model needs (DevModel::variant): the classic one handles plane / sphere / capsule pairs
// with condim 1 / 3; a mesh geom, a height field or a condim 4 / 6 pair takes the general collision + constraint assembly, with PGS on
// 63 rows or Newton on kBigNefcMax rows.  false: no instantiation fits.
bool model_variant(const Model& m, int& variant, int& ncon_max, int& nefc_max, std::string& err) {
  bool general = false, wide = false;  // wide: a pair of contact dimension 4 / 6 (six / ten pyramid rows per contact)
  for (int p = 0; p < m.npair; p++) {
    const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
    const int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
    if (t1 == GEOM_MESH || t2 == GEOM_MESH || t1 == GEOM_HFIELD || t2 == GEOM_HFIELD) general = true;
    const int dim = m.geom_priority[g1] != m.geom_priority[g2] ? m.geom_condim[m.geom_priority[g1] > m.geom_priority[g2] ? g1 : g2] : std::max(m.geom_condim[g1], m.geom_condim[g2]);
    if (dim != 1 && dim != 3) { general = true; wide = true; }
  }
  variant = 0; ncon_max = kNconMax; nefc_max = kNefcMax;
  if (!general) return true;
  if (m.nv > 28) { err = "models with mesh geoms, height fields or condim 4 / 6 support at most 28 degrees of freedom in this build"; return false; }
  if (m.npair > 65535) { err = "general collision: more than 65535 candidate pairs"; return false; }  // (packed work-item words, hb_pose_kernel)
  for (int h = 0; h < m.nhfield; h++)
    if (m.hfield_nrow[h] > 32767 || m.hfield_ncol[h] > 32767) { err = "height fields larger than 32767 x 32767 are not supported"; return false; }
  if (m.solver == SOL_NEWTON) { variant = 2; ncon_max = kBigNconMax; nefc_max = kBigNefcMax; }
  else if (wide) { variant = 3; ncon_max = kBigNconMax; nefc_max = kPgsNefcMax; }  // PGS with six / ten rows per contact: kPgsNefcMax rows, AR in LDS
  else variant = 1;
  return true;
}

bool build_device_model(const Model& m, DeviceModel& D, std::string& err) {
  if (m.nv > 32) { err = "this build supports nv <= 32 degrees of freedom"; return false; }
  if (m.nbody > 64 || m.ngeom > 64) { err = "this build supports at most 64 bodies and 64 geoms"; return false; }
  for (int g = 0; g < m.ngeom; g++)
    if (m.geom_type[g] == GEOM_HFIELD && m.geom_bodyid[g] != 0) { err = "height fields must be attached to the world body"; return false; }
  for (int j = 0; j < m.njnt; j++)
    if (m.jnt_type[j] == JNT_BALL) { err = "ball joints are not supported"; return false; }
  TableBuilder T;
  DevModel& dm = D.dm;
  memset(&dm, 0, sizeof dm);
  int nb = m.nbody, nv = m.nv;
  dm.nq = m.nq; dm.nv = nv; dm.nu = m.nu; dm.nbody = nb; dm.njnt = m.njnt; dm.ngeom = m.ngeom; dm.ntendon = m.ntendon; dm.nM = m.nM; dm.npair = m.npair; dm.nhfielddata = m.nhfielddata;
  dm.nstate = 1 + m.nq + 2 * nv;
  dm.timestep = (float)m.timestep;
  for (int i = 0; i < 3; i++) dm.gravity[i] = (float)m.gravity[i];
  dm.inv_sqrt_impratio = (float)(1.0 / std::sqrt(m.impratio));
  dm.tolerance = (float)m.tolerance;
  dm.pgs_scale = (float)(1.0 / (m.meaninertia * std::max(1, nv)));
  dm.iterations = m.iterations;
  dm.disableflags = m.disableflags;
  dm.solver = m.solver; dm.ls_iterations = m.ls_iterations; dm.ls_tolerance = (float)m.ls_tolerance;
  if (!model_variant(m, dm.variant, dm.ncon_max, dm.nefc_max, err)) return false;
  dm.mpr_iterations = 50; dm.mpr_tolerance = 1e-6f;  // mjOption.mpr_iterations / mpr_tolerance defaults (mjmodel.h:413,437)

  // trees, levels, children, dof masks
  std::vector<int> treeid(nb, 0), roots;
  for (int b = 1; b < nb; b++) {
    if (m.body_parentid[b] == 0) { treeid[b] = (int)roots.size(); roots.push_back(b); }
    else treeid[b] = treeid[m.body_parentid[b]];
  }
  dm.ntree = (int)roots.size();
  std::vector<double> tree_invmass;
  for (int r : roots) tree_invmass.push_back(m.body_subtreemass[r] > 1e-15 ? 1.0 / m.body_subtreemass[r] : 0.0);
  int maxdepth = 0;
  for (int b = 0; b < nb; b++) maxdepth = std::max(maxdepth, m.body_depth[b]);
  dm.nlevel = maxdepth + 1;
  std::vector<int> level_adr(dm.nlevel, 0), level_num(dm.nlevel, 0), level_body;
  for (int L = 0; L <= maxdepth; L++) {
    level_adr[L] = (int)level_body.size();
    for (int b = 0; b < nb; b++) if (m.body_depth[b] == L) { level_body.push_back(b); level_num[L]++; }
  }
  std::vector<int> childadr(nb, 0), childnum(nb, 0), child_list;
  for (int b = 0; b < nb; b++) {
    childadr[b] = (int)child_list.size();
    for (int c = 1; c < nb; c++) if (m.body_parentid[c] == b && c != b) { child_list.push_back(c); childnum[b]++; }
  }
  std::vector<unsigned long long> dofmask(nb, 0);
  for (int b = 1; b < nb; b++)
    for (int a = b; a > 0; a = m.body_parentid[a])
      for (int k = 0; k < m.body_dofnum[a]; k++) dofmask[b] |= 1ull << (m.body_dofadr[a] + k);
  // dof ancestry
  std::vector<int> Mi(m.nM), Mj(m.nM);
  for (int i = 0; i < nv; i++) {
    int adr = m.dof_Madr[i];
    for (int j = i; j >= 0; j = m.dof_parentid[j]) { Mi[adr] = i; Mj[adr] = j; adr++; }
  }
  if (m.nM > 1023) { err = "sparse mass matrix too large for the packed index tables"; return false; }
  // ---- LDS layout (a function of the variant: a variant-2 model also gets the variant-1 layout for its fast step kernel)
  auto lay = [&](DevModel& dm) -> bool {
  int off = 0;
  auto take = [&](int n) { int o = off; off += (n + 3) & ~3; return o; };
  // row stride of C: 33 (K = 32 for the matrix cores plus a pad column); the big Newton layout stores J rows only as wide as its dense
  // order (NDENSE + 1: 21 for the reference's 18-dof robot, else 29): 256 rows of 33 floats would be 34 KB per env
  // (variant 1 with the Newton solver is the fast layout of a variant-2 model: its one-group Newton kernel reads J rows like the big one)
  dm.cstride = (dm.variant == 2 || (dm.variant == 1 && dm.solver == 2)) ? (nv <= 20 ? 21 : 29) : 33;
  dm.o_gquat = dm.variant ? take(4 * m.ngeom) : 0;
  dm.o_qpos = take(m.nq); dm.o_qvel = take(nv); dm.o_warm = take(nv); dm.o_ctrl = take(std::max(1, m.nu));
  dm.o_gpos = take(3 * m.ngeom); dm.o_gaxis = take(3 * m.ngeom); dm.o_scom = take(3 * std::max(1, dm.ntree)); dm.o_cdof = take(12 * nv);  /* angular[3], -, linear[3], -, pad[4] per dof (kCdofStride: conflict-free b128 reads) */
  dm.o_qLD = take(2 * m.nM + 4); dm.o_smooth = take(nv);  // qLD: {M, H} pairs + the zero and one pad pairs of the dense views
  dm.o_vec0 = take(32); dm.o_vec1 = take(32); dm.o_vec2 = take(32);  /* read 32 wide */ dm.o_tenlen = take(std::max(1, m.ntendon));
  int region = off;
  dm.o_xpos = take(12 * nb);  /* xpos[3], -, xquat[4], pad[4] records (kXpqStride) */ dm.o_xmat = take(9 * nb); dm.o_xipos = take(3 * nb);
  dm.o_xanchor = take(3 * m.njnt); dm.o_xaxis = take(3 * m.njnt); dm.o_cinert = take(10 * nb); dm.o_crb = take(20 * nb);  // crb: inertia[10] | cfrc[6] | pad[4] records (kIfStride)
  dm.o_cvel = take(12 * nb);  // cvel[6] | cacc[6] records
  int endA = off;
  off = region;
  dm.o_meta = 0;
  dm.o_AR = 0;
  if (dm.variant == 2) {
    // Newton on kBigNefcMax rows: contacts, J rows, the dense M ([32][33], where the classic layout keeps the row meta), the
    // compact row meta, one D / force word per row
    dm.o_con = take(dm.ncon_max * kConStride);
    dm.o_C = take(std::max(dm.nefc_max * dm.cstride + 64, kListMax * 5 + kWorkMax));  // (the collision pass borrows the head of C for its lists)
    dm.o_efc = take(32 * 36);
    dm.o_meta = take(dm.nefc_max * kMetaStride);
  } else if (dm.variant == 3) {
    // PGS on kPgsNefcMax rows: contacts, J / C rows (+ the qfrc_smooth row), W, the compact row meta, the matrix AR
    dm.o_con = take(dm.ncon_max * kConStride);
    dm.o_C = take(std::max((dm.nefc_max + 1) * dm.cstride + 64, kListMax * 5 + kWorkMax));
    dm.o_efc = take(32 * 36);
    dm.o_meta = take(dm.nefc_max * kMetaStride);
    dm.o_AR = take(dm.nefc_max * dm.nefc_max);
  } else {
  dm.o_con = take(kNconMax * kConStride); dm.o_C = take((kNefcMax + 1) * dm.cstride);
  // per-row meta (13 slots x kNefcMax) is dead once the row quantities are in registers; W = L^-1 D^-1/2
  // ([32][33]) is built over it before the J W product and lives until the dual finish
  dm.o_efc = take(std::max(13 * kNefcMax, 32 * 36));
  // mj_Euler builds W_H and its transpose (2 x [32][36]) from the start of C, running on into the row-meta area
  if (dm.o_efc != dm.o_C + (((kNefcMax + 1) * dm.cstride + 3) & ~3) || dm.o_efc + std::max(13 * kNefcMax, 32 * 36) - dm.o_C < 2 * 32 * 36) {
    err = "internal: LDS layout leaves no room for the Euler solve's W_H pair"; return false;
  }
  if (dm.variant == 1) dm.o_meta = take(64 * kMetaStride);
  }
  dm.o_force = take(std::max(kGroup, dm.nefc_max));
  int endB = off;
  // xipos and scom/cdof are read while region B is being written (xfrc, Jacobians): keep xipos out of the alias
  dm.lds_floats = std::max(endA, endB);
  if (dm.lds_floats * 4 > 160 * 1024) { err = "model needs more LDS than one CU has"; return false; }
  return true;
  };
  if (!lay(dm)) return false;

  std::vector<int> mdense((size_t)32 * 32, m.nM);
  for (int i = 0; i < 32; i++) mdense[(size_t)i * 32 + i] = m.nM + 1;
  for (int e = 0; e < m.nM; e++) { mdense[(size_t)Mj[e] * 32 + Mi[e]] = e; mdense[(size_t)Mi[e] * 32 + Mj[e]] = e; }
  std::vector<int> mdense_c((size_t)16 * 64);
  for (int r = 0; r < 16; r++)
    for (int ln = 0; ln < 64; ln++) mdense_c[(size_t)r * 64 + ln] = mdense[(size_t)(ln & 31) * 32 + ((r & 3) + 8 * (r >> 2) + 4 * (ln >> 5))];
  // pairs
  std::vector<int> pair_dim;
  std::vector<double> pair_fr, pair_solref, pair_solimp, pair_margin, pair_gap;
  for (int p = 0; p < m.npair; p++) {
    int dim; double fr[3], sr[2], si[5], mg, gp;
    mix_pair(m, m.pair_geom1[p], m.pair_geom2[p], dim, fr, sr, si, mg, gp);
    if (dim != 1 && dim != 3 && dim != 4 && dim != 6) { err = "contact dimension " + std::to_string(dim) + " does not exist (condim 1, 3, 4, 6)"; return false; }
    pair_dim.push_back(dim);
    for (double v : fr) pair_fr.push_back(v);
    for (double v : sr) pair_solref.push_back(v);
    for (double v : si) pair_solimp.push_back(v);
    pair_margin.push_back(mg); pair_gap.push_back(gp);
  }
  // domain randomisation scales the sliding friction of the floor (first plane geom): per pair (floor's own
  // coefficient if it takes part, else 0; the other geom's coefficient / the mixed one)
  std::vector<double> pair_fricab;
  {
    int floor_geom = -1;
    for (int g = 0; g < m.ngeom && floor_geom < 0; g++) if (m.geom_type[g] == GEOM_PLANE) floor_geom = g;
    for (int p = 0; p < m.npair; p++) {
      const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
      if (g1 == floor_geom || g2 == floor_geom) {
        const int other = g1 == floor_geom ? g2 : g1;
        pair_fricab.push_back(m.geom_friction[3 * floor_geom]); pair_fricab.push_back(m.geom_friction[3 * other]);
      } else { pair_fricab.push_back(0.0); pair_fricab.push_back(pair_fr[3 * p]); }
    }
    if (pair_fricab.empty()) { pair_fricab.push_back(0.0); pair_fricab.push_back(0.0); }
  }
  // limit candidates
  std::vector<int> lim_kind, lim_id, lim_side;
  std::vector<double> lim_range, lim_margin, lim_solref, lim_solimp, lim_invw;
  for (int j = 0; j < m.njnt; j++) {
    if (!m.jnt_limited[j] || (m.jnt_type[j] != JNT_HINGE && m.jnt_type[j] != JNT_SLIDE)) continue;
    for (int side = -1; side <= 1; side += 2) {
      lim_kind.push_back(0); lim_id.push_back(j); lim_side.push_back(side);
      lim_range.push_back(m.jnt_range[2 * j + (side + 1) / 2]); lim_margin.push_back(m.jnt_margin[j]);
      for (int i = 0; i < 2; i++) lim_solref.push_back(m.jnt_solref[2 * j + i]);
      for (int i = 0; i < 5; i++) lim_solimp.push_back(m.jnt_solimp[5 * j + i]);
      lim_invw.push_back(m.dof_invweight0[m.jnt_dofadr[j]]);
    }
  }
  for (int t = 0; t < m.ntendon; t++) {
    if (!m.tendon_limited[t]) continue;
    for (int side = -1; side <= 1; side += 2) {
      lim_kind.push_back(1); lim_id.push_back(t); lim_side.push_back(side);
      lim_range.push_back(m.tendon_range[2 * t + (side + 1) / 2]); lim_margin.push_back(m.tendon_margin[t]);
      for (int i = 0; i < 2; i++) lim_solref.push_back(m.tendon_solref_lim[2 * t + i]);
      for (int i = 0; i < 5; i++) lim_solimp.push_back(m.tendon_solimp_lim[5 * t + i]);
      lim_invw.push_back(m.tendon_invweight0[t]);
    }
  }
  dm.nlimcand = (int)lim_kind.size();
  std::vector<int> wrap_dofadr, wrap_qposadr;
  for (int w = 0; w < m.nwrap; w++) { wrap_dofadr.push_back(m.jnt_dofadr[m.wrap_objid[w]]); wrap_qposadr.push_back(m.jnt_qposadr[m.wrap_objid[w]]); }
  std::vector<int> act_qposadr, act_dofadr;
  for (int a = 0; a < m.nu; a++) { act_qposadr.push_back(m.jnt_qposadr[m.actuator_trnid[a]]); act_dofadr.push_back(m.jnt_dofadr[m.actuator_trnid[a]]); }
  // env adapter
  dm.obs_root_body = -1; dm.obs_root_dofadr = -1;
  int nscalar = 0;
  for (int j = 0; j < m.njnt; j++) {
    if (m.jnt_type[j] == JNT_FREE && dm.obs_root_dofadr < 0) { dm.obs_root_dofadr = m.jnt_dofadr[j]; dm.obs_root_body = m.jnt_bodyid[j]; }
    if (m.jnt_type[j] == JNT_HINGE || m.jnt_type[j] == JNT_SLIDE) nscalar++;
  }
  dm.nobs = 2 * nscalar + 6;
  // observation gather table: state-record offsets of the scalar joints' qpos and qvel, then the root's angular
  // velocity (3) - the first nobs - 3 observation entries are plain copies out of the state record
  std::vector<int> obs_src;
  for (int j = 0; j < m.njnt; j++) if (m.jnt_type[j] == JNT_HINGE || m.jnt_type[j] == JNT_SLIDE) obs_src.push_back(1 + m.jnt_qposadr[j]);
  for (int j = 0; j < m.njnt; j++) if (m.jnt_type[j] == JNT_HINGE || m.jnt_type[j] == JNT_SLIDE) obs_src.push_back(1 + m.nq + m.jnt_dofadr[j]);
  for (int i = 0; i < 3; i++) obs_src.push_back(dm.obs_root_dofadr >= 0 ? 1 + m.nq + dm.obs_root_dofadr + 3 + i : -1);
  dm.obs_root_qadr = -1;
  for (int j = 0; j < m.njnt; j++) if (m.jnt_type[j] == JNT_FREE) { dm.obs_root_qadr = m.jnt_qposadr[j]; break; }
  // the scalar joints in observation order: joint order, and - when every scalar joint has exactly one actuator - actuator order
  // (the reference's JOINT_NAMES order, hb_env_config.obs_actuator_order); the second half of each table is the obs_src of that order
  std::vector<int> obs_jnt, obs_jnt_act, obs_src_act;
  for (int j = 0; j < m.njnt; j++) if (m.jnt_type[j] == JNT_HINGE || m.jnt_type[j] == JNT_SLIDE) obs_jnt.push_back(j);
  {
    std::vector<int> seen(m.njnt, 0);
    bool ok = m.nu == nscalar;
    for (int a = 0; a < m.nu && ok; a++) { if (seen[m.actuator_trnid[a]]++) ok = false; obs_jnt_act.push_back(m.actuator_trnid[a]); }
    if (!ok) obs_jnt_act.clear();
    D.has_act_order = ok;
    for (int j : obs_jnt_act) obs_src_act.push_back(1 + m.jnt_qposadr[j]);
    for (int j : obs_jnt_act) obs_src_act.push_back(1 + m.nq + m.jnt_dofadr[j]);
    for (int i = 0; i < 3; i++) obs_src_act.push_back(dm.obs_root_dofadr >= 0 ? 1 + m.nq + dm.obs_root_dofadr + 3 + i : -1);
  }

  // level-ordered body records, dof records, packed M entries (layouts in hb_device.hpp)
  std::vector<float> brec((size_t)nb * kBrecQuads * 4, 0.f);
  auto fi = [](int v) { float f; memcpy(&f, &v, 4); return f; };
  for (int sl = 0; sl < nb; sl++) {
    int b = level_body[sl];
    float* r = &brec[(size_t)sl * kBrecQuads * 4];
    if (m.body_jntnum[b] > 3) { err = "at most 3 joints per body are supported (body '" + m.body_name[b] + "')"; return false; }
    if (childnum[b] > 8) { err = "at most 8 child bodies per body are supported (body '" + m.body_name[b] + "')"; return false; }
    r[0] = fi(b); r[1] = fi(m.body_parentid[b]); r[2] = fi(m.body_jntnum[b]); r[3] = fi(m.body_jntadr[b]);
    {  // depth, and the ancestors 2, 4 and 8 links up (the world once the chain ends): the kinematics pass composes
       // poses by pointer jumping, log2(depth) rounds instead of one per level
      auto up = [&](int x, int k) { while (k-- > 0 && x > 0) x = m.body_parentid[x]; return x; };
      if (m.body_depth[b] > 16) { err = "kinematic trees deeper than 16 bodies are not supported"; return false; }
      r[4] = fi(m.body_depth[b] | (up(b, 2) << 8) | (up(b, 4) << 16) | (up(b, 8) << 24));
    }
    r[5] = fi(treeid[b]); r[6] = (float)m.body_mass[b]; r[7] = fi(childnum[b]);
    for (int i = 0; i < 3; i++) { r[8 + i] = (float)m.body_pos[3 * b + i]; r[16 + i] = (float)m.body_ipos[3 * b + i]; r[24 + i] = (float)m.body_inertia[3 * b + i]; }
    for (int i = 0; i < 4; i++) { r[12 + i] = (float)m.body_quat[4 * b + i]; r[20 + i] = (float)m.body_iquat[4 * b + i]; }
    for (int c = 0; c < 8; c++) r[28 + c] = fi(c < childnum[b] ? child_list[childadr[b] + c] : 0);
    for (int jj = 0; jj < m.body_jntnum[b]; jj++) {
      int j = m.body_jntadr[b] + jj;
      float* q = r + 36 + 12 * jj;
      q[0] = fi(m.jnt_type[j]); q[1] = fi(m.jnt_qposadr[j]); q[2] = fi(m.jnt_dofadr[j]); q[3] = (float)m.qpos0[m.jnt_qposadr[j]];
      for (int i = 0; i < 3; i++) { q[4 + i] = (float)m.jnt_axis[3 * j + i]; q[8 + i] = (float)m.jnt_pos[3 * j + i]; }
    }
  }
  std::vector<float> drec((size_t)nv * 12, 0.f);
  for (int d = 0; d < nv; d++) {
    int j = m.dof_jntid[d], b = m.dof_bodyid[d];
    float* r = &drec[(size_t)d * 12];
    r[0] = fi(j); r[1] = fi(b); r[2] = fi(m.jnt_type[j]); r[3] = fi(d - m.jnt_dofadr[j]);
    r[4] = fi(treeid[b]); r[5] = (float)m.dof_armature[d]; r[6] = (float)m.dof_damping[d]; r[7] = (float)m.jnt_stiffness[j];
    r[8] = fi(m.jnt_qposadr[j]); r[9] = (float)m.qpos_spring[m.jnt_qposadr[j]];
  }
  std::vector<int> mrec(m.nM);
  std::vector<float> mdiag((size_t)m.nM * 2, 0.f);
  for (int e = 0; e < m.nM; e++) {
    mrec[e] = Mi[e] | (Mj[e] << 8) | (m.dof_bodyid[Mi[e]] << 16);
    if (Mi[e] == Mj[e]) { mdiag[2 * e] = (float)m.dof_armature[Mi[e]]; mdiag[2 * e + 1] = (float)m.dof_damping[Mi[e]]; }
  }

  // ---- offsets into the flat tables
  struct IO { const int** p; size_t o; };
  struct FO { const float** p; size_t o; };
  std::vector<IO> io;
  std::vector<FO> fo;
#define TI(field, vec) io.push_back({&dm.field, T.addi(vec)})
#define TF(field, vec) fo.push_back({&dm.field, T.addf(vec)})
  TI(body_treeid, treeid); TF(body_invweight0, m.body_invweight0); TF(tree_invmass, tree_invmass);
  TI(jnt_type, m.jnt_type); TI(jnt_qposadr, m.jnt_qposadr); TI(jnt_dofadr, m.jnt_dofadr); TF(qpos0, m.qpos0); TI(dof_jntid, m.dof_jntid); TI(dof_Madr, m.dof_Madr); TF(dof_damping, m.dof_damping); TI(mrec, mrec);
  TI(mdense, mdense); TI(mdense_c, mdense_c);
  TI(geom_type, m.geom_type); TI(geom_bodyid, m.geom_bodyid); TI(geom_dataid, m.geom_dataid);
  std::vector<int> geom_meshadr(m.ngeom, 0), geom_meshnum(m.ngeom, 0);
  for (int g = 0; g < m.ngeom; g++)
    if (m.geom_type[g] == GEOM_MESH) { geom_meshadr[g] = m.mesh_vertadr[m.geom_dataid[g]]; geom_meshnum[g] = m.mesh_vertnum[m.geom_dataid[g]]; }
  TI(geom_meshadr, geom_meshadr); TI(geom_meshnum, geom_meshnum);
  TI(hfield_nrow, m.hfield_nrow); TI(hfield_ncol, m.hfield_ncol); TI(hfield_adr, m.hfield_adr); TF(hfield_size, m.hfield_size); TF(hfield_data, m.hfield_data);
  TF(geom_size, m.geom_size); TF(geom_pos, m.geom_pos); TF(geom_quat, m.geom_quat); TF(geom_rbound, m.geom_rbound);
  // half extents of every geom's bounding box in its own frame (the oriented boxes the broadphase of the general path tests before a
  // pair of geoms is handed to the portal search): the hull's coordinate range for a mesh, the obvious for spheres and capsules
  std::vector<double> geom_half((size_t)std::max(1, m.ngeom) * 3, 0.0);
  for (int g = 0; g < m.ngeom; g++) {
    double* h = &geom_half[3 * (size_t)g];
    h[0] = h[1] = h[2] = m.geom_rbound[g];
    if (m.geom_type[g] == GEOM_SPHERE) h[0] = h[1] = h[2] = m.geom_size[3 * g];
    else if (m.geom_type[g] == GEOM_CAPSULE) { h[0] = h[1] = m.geom_size[3 * g]; h[2] = m.geom_size[3 * g] + m.geom_size[3 * g + 1]; }
    else if (m.geom_type[g] == GEOM_MESH && m.geom_dataid[g] >= 0) {
      const int k = m.geom_dataid[g];
      h[0] = h[1] = h[2] = 0.0;
      for (int v = 0; v < m.mesh_vertnum[k]; v++)
        for (int c = 0; c < 3; c++) h[c] = std::max(h[c], std::fabs((double)(float)m.mesh_vert[3 * (size_t)(m.mesh_vertadr[k] + v) + c]));
    }
  }
  TF(geom_half, geom_half);
  dm.box_cull = !(getenv("HB_BOX_CULL") && atoi(getenv("HB_BOX_CULL")) == 0);
  std::vector<int> pair_self;  // both geoms of the pair on the robot (neither on the world body)
  for (int p = 0; p < m.npair; p++) pair_self.push_back(m.geom_bodyid[m.pair_geom1[p]] != 0);
  TI(pair_geom1, m.pair_geom1); TI(pair_geom2, m.pair_geom2); TI(pair_dim, pair_dim); TI(pair_self, pair_self);
  TF(pair_fricab, pair_fricab);
  TF(pair_friction, pair_fr); TF(pair_solref, pair_solref); TF(pair_solimp, pair_solimp); TF(pair_margin, pair_margin); TF(pair_gap, pair_gap);
  TI(lim_kind, lim_kind); TI(lim_id, lim_id); TI(lim_side, lim_side);
  TF(lim_range, lim_range); TF(lim_margin, lim_margin); TF(lim_solref, lim_solref); TF(lim_solimp, lim_solimp); TF(lim_invweight, lim_invw);
  TI(obs_src, obs_src); TI(obs_jnt, obs_jnt);
  const size_t o_obs_jnt_act = T.addi(obs_jnt_act.empty() ? std::vector<int>{0} : obs_jnt_act), o_obs_src_act = T.addi(obs_src_act.empty() ? std::vector<int>{0} : obs_src_act);
  TI(tendon_adr, m.tendon_adr); TI(tendon_num, m.tendon_num); TI(wrap_dofadr, wrap_dofadr); TI(wrap_qposadr, wrap_qposadr);
  TF(wrap_prm, m.wrap_prm);
  TI(act_qposadr, act_qposadr); TI(act_dofadr, act_dofadr); TI(act_ctrllimited, m.actuator_ctrllimited); TI(act_forcelimited, m.actuator_forcelimited);
  TF(act_gear, m.actuator_gear); TF(act_ctrlrange, m.actuator_ctrlrange); TF(act_forcerange, m.actuator_forcerange); TF(act_gain, m.actuator_gainprm);
  TF(act_bias, m.actuator_biasprm);
#undef TI
#undef TF
  size_t o_mask = T.addu(dofmask);
  // per collision pair, everything mj_makeConstraint needs of it in one 5-quad record (one scalar fetch per contact):
  // [0] body1, body2, tree1, tree2   [1] dof mask of body1 (lo, hi), of body2 (lo, hi)
  // [2] margin - gap, solref[2], invweight0 sum   [3] solimp[0..3]   [4] solimp[4], condim, -, -
  std::vector<float> prec((size_t)std::max(1, m.npair) * 20, 0.f);
  for (int p = 0; p < m.npair; p++) {
    float* r = &prec[(size_t)p * 20];
    const int b1 = m.geom_bodyid[m.pair_geom1[p]], b2 = m.geom_bodyid[m.pair_geom2[p]];
    r[0] = fi(b1); r[1] = fi(b2); r[2] = fi(treeid[b1]); r[3] = fi(treeid[b2]);
    r[4] = fi((int)(dofmask[b1] & 0xffffffffull)); r[5] = fi((int)(dofmask[b1] >> 32));
    r[6] = fi((int)(dofmask[b2] & 0xffffffffull)); r[7] = fi((int)(dofmask[b2] >> 32));
    r[8] = (float)(pair_margin[p] - pair_gap[p]); r[9] = (float)pair_solref[2 * p]; r[10] = (float)pair_solref[2 * p + 1];
    r[11] = (float)(m.body_invweight0[2 * b1] + m.body_invweight0[2 * b2]);
    for (int i = 0; i < 4; i++) r[12 + i] = (float)pair_solimp[5 * p + i];
    r[16] = (float)pair_solimp[5 * p + 4]; r[17] = fi(pair_dim[p]);
  }
  // per collision pair, what mj_collision needs of it in one 3-quad record (one vector fetch per lane and round):
  // [0] geom1, geom2, type1 | type2 << 8, margin   [1] rbound1, rbound2, size1[0], size1[1]   [2] size2[0], size2[1], -, -
  std::vector<float> crec((size_t)(std::max(1, m.npair) + 64) * 12, 0.f);  // + one round of padding for the prefetch
  for (int p = 0; p < m.npair; p++) {
    float* r = &crec[(size_t)p * 12];
    const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
    r[0] = fi(g1); r[1] = fi(g2); r[2] = fi(m.geom_type[g1] | (m.geom_type[g2] << 8)); r[3] = (float)pair_margin[p];
    r[4] = (float)m.geom_rbound[g1]; r[5] = (float)m.geom_rbound[g2]; r[6] = (float)m.geom_size[3 * g1]; r[7] = (float)m.geom_size[3 * g1 + 1];
    r[8] = (float)m.geom_size[3 * g2]; r[9] = (float)m.geom_size[3 * g2 + 1];
  }
  // per fixed tendon, its first four wraps in one 3-quad record (a tendon with more falls back to the wrap tables):
  // [0] coefficients   [1] qpos addresses   [2] dof addresses; unused slots have coefficient 0 and address 0
  std::vector<float> trec((size_t)std::max(1, m.ntendon) * 12, 0.f);
  for (int t = 0; t < m.ntendon; t++)
    for (int w = 0; w < std::min(4, m.tendon_num[t]); w++) {
      const int a = m.tendon_adr[t] + w;
      trec[(size_t)t * 12 + w] = (float)m.wrap_prm[a];
      trec[(size_t)t * 12 + 4 + w] = fi(wrap_qposadr[a]);
      trec[(size_t)t * 12 + 8 + w] = fi(wrap_dofadr[a]);
    }
  // per limit candidate, everything mj_instantiateLimit needs of it in one 4-quad record (one round trip instead of the
  // dependent walk candidate -> joint -> addresses):
  // [0] kind, id, side, qpos address (joints)   [1] margin, range, solref[2]   [2] solimp[0..3]   [3] solimp[4], invweight, dof address (joints), -
  std::vector<float> lrec((size_t)std::max(1, dm.nlimcand) * 16, 0.f);
  for (int c = 0; c < dm.nlimcand; c++) {
    float* r = &lrec[(size_t)c * 16];
    const bool joint = lim_kind[c] == 0;
    r[0] = fi(lim_kind[c]); r[1] = fi(lim_id[c]); r[2] = fi(lim_side[c]); r[3] = fi(joint ? m.jnt_qposadr[lim_id[c]] : 0);
    r[4] = (float)lim_margin[c]; r[5] = (float)lim_range[c]; r[6] = (float)lim_solref[2 * c]; r[7] = (float)lim_solref[2 * c + 1];
    for (int i = 0; i < 5; i++) r[8 + i] = (float)lim_solimp[5 * c + i];
    r[13] = (float)lim_invw[c]; r[14] = fi(joint ? m.jnt_dofadr[lim_id[c]] : 0);
  }
  // per actuator, one 4-quad record: [0] ctrllimited, forcelimited, qpos address, dof address   [1] ctrlrange[2], gear, gain
  // [2] biasprm[0..2], -   [3] forcerange[2], -, -
  std::vector<float> arec((size_t)std::max(1, m.nu) * 16, 0.f);
  for (int a = 0; a < m.nu; a++) {
    float* r = &arec[(size_t)a * 16];
    r[0] = fi(m.actuator_ctrllimited[a]); r[1] = fi(m.actuator_forcelimited[a]); r[2] = fi(act_qposadr[a]); r[3] = fi(act_dofadr[a]);
    r[4] = (float)m.actuator_ctrlrange[2 * a]; r[5] = (float)m.actuator_ctrlrange[2 * a + 1]; r[6] = (float)m.actuator_gear[a]; r[7] = (float)m.actuator_gainprm[a];
    for (int i = 0; i < 3; i++) r[8 + i] = (float)m.actuator_biasprm[3 * a + i];
    r[12] = (float)m.actuator_forcerange[2 * a]; r[13] = (float)m.actuator_forcerange[2 * a + 1];
  }
  // hull vertices as 16-byte records (x, y, z, link) and the edge graph with inlined coordinates, each vertex's neighbour list padded
  // to whole chunks of kMeshChunk records with copies of the vertex itself (hb_device.hpp); per mesh the cube map of start vertices
  std::vector<int> padadr(std::max(1, m.nmeshvert), 0), padchunks(std::max(1, m.nmeshvert), 0);
  int npad = 0;
  for (int g = 0; g < m.nmeshvert; g++) {
    padadr[g] = npad;
    padchunks[g] = (m.mesh_nbrnum[g] + kMeshChunk - 1) / kMeshChunk;
    npad += padchunks[g] * kMeshChunk;
    if (padchunks[g] > 255 || padadr[g] >= (1 << 23)) { err = "mesh edge graph too large for the packed link words"; return false; }
  }
  std::vector<float> meshv((size_t)std::max(1, m.nmeshvert) * 4, 0.f), meshn((size_t)std::max(1, npad) * 4, 0.f), meshs((size_t)std::max(1, m.nmesh) * kMeshStart * 4, 0.f);
  auto link_of = [&](int g) { return fi((padadr[g] << 8) | padchunks[g]); };
  for (int k = 0; k < m.nmesh; k++) {
    for (int v = 0; v < m.mesh_vertnum[k]; v++) {
      const int g = m.mesh_vertadr[k] + v;
      for (int i = 0; i < 3; i++) meshv[(size_t)4 * g + i] = (float)m.mesh_vert[3 * g + i];
      meshv[(size_t)4 * g + 3] = link_of(g);
      for (int i = 0; i < padchunks[g] * kMeshChunk; i++) {
        const int w = i < m.mesh_nbrnum[g] ? m.mesh_vertadr[k] + m.mesh_nbr[m.mesh_nbradr[g] + i] : g;
        const size_t r = (size_t)padadr[g] + i;
        for (int c = 0; c < 3; c++) meshn[4 * r + c] = (float)m.mesh_vert[3 * w + c];
        meshn[4 * r + 3] = link_of(w);
      }
    }
    // cube map: face f = 2 * axis + (negative ? 1 : 0), cell (iu, iv) over the other two axes in cyclic order, u, v in [-1, 1]
    for (int f = 0; f < 6; f++)
      for (int iu = 0; iu < 4; iu++)
        for (int iv = 0; iv < 4; iv++) {
          const int ax = f >> 1;
          double d[3];
          d[ax] = (f & 1) ? -1.0 : 1.0; d[(ax + 1) % 3] = -0.75 + 0.5 * iu; d[(ax + 2) % 3] = -0.75 + 0.5 * iv;
          int best = m.mesh_vertadr[k];
          double bd = -1e300;
          for (int v = 0; v < m.mesh_vertnum[k]; v++) {
            const int g = m.mesh_vertadr[k] + v;
            // (the float-rounded coordinates the device climbs on)
            const double val = (double)(float)m.mesh_vert[3 * g] * d[0] + (double)(float)m.mesh_vert[3 * g + 1] * d[1] + (double)(float)m.mesh_vert[3 * g + 2] * d[2];
            if (val > bd) { bd = val; best = g; }
          }
          float* rec = &meshs[((size_t)k * kMeshStart + f * 16 + iu * 4 + iv) * 4];
          for (int c = 0; c < 3; c++) rec[c] = (float)m.mesh_vert[3 * best + c];
          rec[3] = link_of(best);
        }
  }
  const size_t o_meshv = T.addraw(meshv), o_meshn = T.addraw(meshn), o_meshs = T.addraw(meshs);
  const size_t o_arec = T.addraw(arec);
  size_t o_brec = T.addraw(brec), o_drec = T.addraw(drec), o_mdiag = T.addraw(mdiag), o_prec = T.addraw(prec), o_crec = T.addraw(crec), o_trec = T.addraw(trec),
         o_lrec = T.addraw(lrec);

  // ---- upload
  if (hipMalloc((void**)&D.d_int, T.iv.size() * sizeof(int)) != hipSuccess || hipMalloc((void**)&D.d_flt, T.fv.size() * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&D.d_u64, T.uv.size() * sizeof(unsigned long long)) != hipSuccess) { err = "hipMalloc failed for model tables"; return false; }
  if (hipMemcpy(D.d_int, T.iv.data(), T.iv.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(D.d_flt, T.fv.data(), T.fv.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(D.d_u64, T.uv.data(), T.uv.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy failed for model tables"; return false; }
  for (auto& x : io) *x.p = D.d_int + x.o;
  for (auto& x : fo) *x.p = D.d_flt + x.o;
  dm.body_dofmask = D.d_u64 + o_mask;
  dm.brec = reinterpret_cast<const float4*>(D.d_flt + o_brec);
  dm.drec = reinterpret_cast<const float4*>(D.d_flt + o_drec);
  dm.prec = reinterpret_cast<const float4*>(D.d_flt + o_prec);
  dm.crec = reinterpret_cast<const float4*>(D.d_flt + o_crec);
  dm.trec = reinterpret_cast<const float4*>(D.d_flt + o_trec);
  dm.lrec = reinterpret_cast<const float4*>(D.d_flt + o_lrec);
  D.obs_jnt_joint = dm.obs_jnt; D.obs_src_joint = dm.obs_src;
  D.obs_jnt_act = D.d_int + o_obs_jnt_act; D.obs_src_act = D.d_int + o_obs_src_act;
  dm.arec = reinterpret_cast<const float4*>(D.d_flt + o_arec);
  dm.mesh_vert = reinterpret_cast<const float4*>(D.d_flt + o_meshv);
  dm.mesh_nbr = reinterpret_cast<const float4*>(D.d_flt + o_meshn);
  dm.mesh_start = reinterpret_cast<const float4*>(D.d_flt + o_meshs);
  dm.mdiag = reinterpret_cast<const float2*>(D.d_flt + o_mdiag);
  std::vector<float> qsrc;
  for (double v : m.qpos0) qsrc.push_back((float)v);
  for (double v : m.key_qpos) qsrc.push_back((float)v);
  if (hipMalloc((void**)&D.d_qpos_src, qsrc.size() * sizeof(float)) != hipSuccess ||
      hipMemcpy(D.d_qpos_src, qsrc.data(), qsrc.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMalloc failed for qpos sources"; return false; }
  if (hipMalloc((void**)&D.d_dm, sizeof(DevModel)) != hipSuccess || hipMemcpy(D.d_dm, &dm, sizeof(DevModel), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMalloc failed for the device model"; return false; }
  // A variant-2 model (Newton on 256 rows in four register groups: one wave per SIMD) almost always has at most 63 rows and 24
  // contacts in a step: its staged step first runs the one-group Newton instantiation (two waves per SIMD) on the variant-1 LDS
  // layout and falls back to the four-group kernel for the envs that overflow (launch_step).  Same tables, other offsets.
  {
    // the size-specialised kernel (hb_step_h27_kernel): the model's sizes and the layout just computed against the compile-time mirror
    const SizedModel z = dm.variant == 1 ? kSizedHumanoid27V1 : kSizedHumanoid27;
    D.sized_h27 = (dm.variant == 0 || (dm.variant == 1 && dm.solver == 0)) && dm.o_gquat == z.o_gquat && dm.o_meta == z.o_meta && m.nq == z.nq && nv == z.nv && m.nu == z.nu && nb == z.nbody && m.njnt == z.njnt && m.ngeom == z.ngeom &&
                  m.ntendon == z.ntendon && m.nM == z.nM && dm.ntree == z.ntree && m.npair == z.npair && dm.nlevel == z.nlevel && dm.nlimcand == z.nlimcand && dm.nstate == z.nstate && dm.cstride == z.cstride &&
                  dm.o_qpos == z.o_qpos && dm.o_qvel == z.o_qvel && dm.o_warm == z.o_warm && dm.o_ctrl == z.o_ctrl && dm.o_gpos == z.o_gpos && dm.o_gaxis == z.o_gaxis &&
                  dm.o_scom == z.o_scom && dm.o_cdof == z.o_cdof && dm.o_qLD == z.o_qLD && dm.o_smooth == z.o_smooth && dm.o_vec0 == z.o_vec0 && dm.o_vec1 == z.o_vec1 &&
                  dm.o_vec2 == z.o_vec2 && dm.o_tenlen == z.o_tenlen && dm.o_xpos == z.o_xpos && dm.o_xmat == z.o_xmat && dm.o_xipos == z.o_xipos &&
                  dm.o_xanchor == z.o_xanchor && dm.o_xaxis == z.o_xaxis && dm.o_cinert == z.o_cinert && dm.o_crb == z.o_crb && dm.o_cvel == z.o_cvel &&
                  dm.o_con == z.o_con && dm.o_C == z.o_C && dm.o_efc == z.o_efc && dm.o_force == z.o_force && dm.lds_floats == z.lds_floats;
  }
  D.fast_lds_floats = 0;
  if (dm.variant == 2 || dm.variant == 3) {
    DevModel fm = dm;
    fm.variant = 1; fm.ncon_max = kNconMax; fm.nefc_max = kNefcMax;
    if (!lay(fm)) return false;
    if (hipMalloc((void**)&D.d_dm_fast, sizeof(DevModel)) != hipSuccess || hipMemcpy(D.d_dm_fast, &fm, sizeof(DevModel), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMalloc failed for the device model"; return false; }
    D.fast_lds_floats = fm.lds_floats;
    {
      const SizedModel z = kSizedTeamV1;  // the robot's fast layout against its compile-time mirror (hb_step_newton_gen20_team_kernel)
      D.sized_team = dm.variant == 2 && fm.o_gquat == z.o_gquat && fm.o_meta == z.o_meta && m.nq == z.nq && nv == z.nv && m.nu == z.nu && nb == z.nbody && m.njnt == z.njnt &&
                     m.ngeom == z.ngeom && m.ntendon == z.ntendon && m.nM == z.nM && fm.ntree == z.ntree && m.npair == z.npair && fm.nlevel == z.nlevel && fm.nlimcand == z.nlimcand &&
                     fm.nstate == z.nstate && fm.cstride == z.cstride && fm.o_qpos == z.o_qpos && fm.o_qvel == z.o_qvel && fm.o_warm == z.o_warm && fm.o_ctrl == z.o_ctrl &&
                     fm.o_gpos == z.o_gpos && fm.o_gaxis == z.o_gaxis && fm.o_scom == z.o_scom && fm.o_cdof == z.o_cdof && fm.o_qLD == z.o_qLD && fm.o_smooth == z.o_smooth &&
                     fm.o_vec0 == z.o_vec0 && fm.o_vec1 == z.o_vec1 && fm.o_vec2 == z.o_vec2 && fm.o_tenlen == z.o_tenlen && fm.o_xpos == z.o_xpos && fm.o_xmat == z.o_xmat &&
                     fm.o_xipos == z.o_xipos && fm.o_xanchor == z.o_xanchor && fm.o_xaxis == z.o_xaxis && fm.o_cinert == z.o_cinert && fm.o_crb == z.o_crb && fm.o_cvel == z.o_cvel &&
                     fm.o_con == z.o_con && fm.o_C == z.o_C && fm.o_efc == z.o_efc && fm.o_force == z.o_force && fm.lds_floats == z.lds_floats;
    }
  }
  return true;
}

}  // namespace

struct hb_batch {
  const hb_model* model = nullptr;
  DeviceModel D;
  int n_env = 0, device = 0;
  hipStream_t stream = nullptr;
  float *d_state = nullptr, *d_ctrl = nullptr, *d_xfrc = nullptr, *d_diag_qacc = nullptr, *d_diag_force = nullptr, *d_diag_contact = nullptr;
  float *d_obs = nullptr, *d_reward = nullptr;
  int* d_seen = nullptr;        // [n_env] warning bits of episodes that ended since the last hb_env_warnings
  float* d_term_obs = nullptr;  // [n_env][nobs] observations of the states episodes ended in (hb_env_terminal_obs), null until asked for
  uint8_t *d_term = nullptr, *d_trunc = nullptr, *d_mask = nullptr;
  int *d_status = nullptr, *d_counts = nullptr;
  size_t ctrl_cap = 0;  // floats
  float* d_qpos_out = nullptr;
  size_t qpos_out_cap = 0;
  float* d_qvel_out = nullptr;
  size_t qvel_out_cap = 0;
  float* d_task_out = nullptr;  // task returns and stage costs
  float xfrc_std = 0.f, xfrc_rate = 0.f;  // rollout noise (hb_rollout_noise)
  int tape_steps = 0;                      // steps of the action tape hb_ctrl_tape_splines left in d_ctrl (0: none)
  float* d_knots = nullptr;               // spline nodes and node times staged for it
  size_t knots_cap = 0;
  unsigned xfrc_seed = 0, xfrc_calls = 0;
  size_t task_out_cap = 0;
  float* d_sensor_out = nullptr;
  size_t sensor_out_cap = 0;
  bool diag = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  unsigned long long* d_stamps = nullptr;
  StageBufs stage = {};  // staged step of the general variants (null: fused)
  // env adapter (hb_env_*)
  EnvConfig env_cfg = {};
  bool env_ready = false;
  float *d_prev = nullptr, *d_latest = nullptr, *d_qfrc = nullptr, *d_action = nullptr;
  int* d_episode = nullptr;
  int env_offset = 0;
  // realism layer (hb_env_randomize)
  EnvRand env_rand = {};
  EnvRandState rs = {};     // device arrays; all null while off
  bool rand_on = false;
  DomainRand dom_rand = {};    // hb_env_domain_randomize
  float* d_dr = nullptr;       // [n_env][dr_stride] per-env model parameters, null while off
  int dr_stride = 0;
  uint8_t* d_rmask = nullptr;  // hb_env_reset's pending-envs mask
  int* d_pending = nullptr;
  // policy MLP (hb_policy_*)
  int mlp_layers = 0;
  int mlp_sizes[5] = {0, 0, 0, 0, 0};
  float* d_mlp_w[4] = {nullptr, nullptr, nullptr, nullptr};
  float* d_mlp_wp[4] = {nullptr, nullptr, nullptr, nullptr};  // packed for hb_policy_kernel (all widths <= 256), else null
  bool mlp_fused = false;
  float* d_mlp_b[4] = {nullptr, nullptr, nullptr, nullptr};
  float* d_mlp_h[2] = {nullptr, nullptr};  // hidden activations, ping-pong
  float* d_mlp_act = nullptr;              // activation scratch of the LDS-free policy kernel: [n_env / 16 + kPipes][2][16][widest + 4]
  // optional per-kernel timing of the step kernel (hb_step_timing)
  bool time_steps = false;
  std::vector<hipEvent_t> tev;  // pairs
  int tev_used = 0;
  long long launch_count = 0;
  const char* last_kernel = "";  // hb_last_kernel
  // run-time choices between kernels / schedules that give the same results (hb_batch_tune, include/hb.h: HB_TUNE_*), indexed by knob
  int tune[HB_TUNE_COUNT] = {getenv("HB_DUO") ? atoi(getenv("HB_DUO")) : 1, 1, 1, 1, 1, 1, 1, 4, 1, kFoldMax};
  // hb_step_dev calls not launched yet (fold_steps): the launch parameters they share, and the controls of each
  BatchPtrs fold_P;
  const float* fold_ctrl[kFoldMax] = {};
  int fold_n = 0;
  int* d_order = nullptr;   // heavy-first dispatch order (hb_order_kernel), valid once a step has run
  int* d_order2 = nullptr;  // the same for the narrowphase launch of a staged step
  int order_mode = 0;       // 0: none yet, 1: one permutation of the whole batch, 2: one permutation per pipe segment
  bool schedule = true;  // heavy-first dispatch order (HB_TUNE_SCHEDULE)
  // Pipelined stepping (hb_batch_pipeline): the batch is cut into npipe fixed env segments, each stepped by
  // its own launch on its own stream.  Envs are independent, so segment c of step t+1 only has to follow
  // segment c of step t: the tail of one step (its slowest envs) overlaps the head of the next.  `stream`
  // stays the batch's ordering point: pipes fork from it at every step call and are joined back into it
  // before anything else is enqueued on it.
  static constexpr int kPipes = 8;  // most segments; npipe of them in use (streams are created when first asked for)
  int npipe = 0;                    // 0: unpipelined
  int probed_segments = 0;          // what hb_batch_pipeline(b, 1)'s probe of the segment streams found (0: not probed yet); the streams are kept, so is the answer
  bool forked = false;
  hipStream_t pipe[kPipes] = {};
  hipEvent_t ev_fork = nullptr, ev_pipe[kPipes] = {};
  int join_error = 0;
  bool main_dirty = true;  // work was enqueued on `stream` since the pipes last forked from it
};

namespace {

// HB_DEBUG=1 in the environment names the failing HIP call on stderr
static bool hb_debug() { static const bool on = getenv("HB_DEBUG") != nullptr; return on; }
// a call whose failure is not fatal (teardown paths): still named under HB_DEBUG, and never left behind as the
// thread's sticky last error for an unrelated launch to trip over
#define HB_IGN(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    if (hb_debug()) fprintf(stderr, "[hb] %s:%d: %s -> %s (ignored)\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
    (void)hipGetLastError(); } } while (0)
#define HB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    if (hb_debug()) fprintf(stderr, "[hb] %s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
    return HB_ENODEVICE; } } while (0)

// the batch's control buffer for WRITING: whatever hb_ctrl_tape_splines left there is gone afterwards, so a later
// HB_CTRL_TAPE rollout must fail (HB_EINVAL) instead of rolling out stale controls
float* ctrl_for_write(hb_batch* b) { b->tape_steps = 0; return b->d_ctrl; }

int ensure_ctrl(hb_batch* b, size_t floats) {
  if (floats <= b->ctrl_cap) return HB_OK;
  if (b->d_ctrl) HB_IGN(hipFree(b->d_ctrl));
  b->d_ctrl = nullptr; b->ctrl_cap = 0; b->tape_steps = 0;
  if (hipMalloc((void**)&b->d_ctrl, floats * sizeof(float)) != hipSuccess) return HB_ENOMEM;
  b->ctrl_cap = floats;
  return HB_OK;
}

// the staged step's buffers as this launch sees them (HB_TUNE_STAGED / FASTPASS / NARROW_PRIM switch parts of it off: all null = the fused step)
StageBufs staged(const hb_batch* b) {
  StageBufs sb = b->stage;
  if (!b->tune[HB_TUNE_STAGED]) { sb = StageBufs{}; return sb; }
  if (!b->tune[HB_TUNE_FASTPASS]) { sb.defer = nullptr; sb.defer_list = nullptr; sb.defer_count = nullptr; sb.dm_fast = nullptr; sb.fast_lds = 0; }
  if (!b->tune[HB_TUNE_NARROW_PRIM]) sb.no_mesh = 0;
  return sb;
}
bool staged_on(const hb_batch* b) { return b->stage.result && b->tune[HB_TUNE_STAGED]; }

BatchPtrs make_ptrs(hb_batch* b) {
  BatchPtrs P;
  memset(&P, 0, sizeof P);
  P.state = b->d_state; P.status = b->d_status; P.counts = b->d_counts; P.xfrc = b->d_xfrc; P.qfrc_out = b->d_qfrc;
  if (b->diag) { P.diag_qacc = b->d_diag_qacc; P.diag_force = b->d_diag_force; P.diag_contact = b->d_diag_contact; }
  P.n_env = b->n_env;
  P.integrate = 1;
  if (b->schedule && b->order_mode) { P.order = b->d_order; P.order2 = staged_on(b) ? b->d_order2 : nullptr; }
  P.blk0 = 0; P.nblk = b->n_env;
  P.dr = b->d_dr; P.dr_stride = b->dr_stride;
  P.stamps = b->d_stamps;
#ifdef HB_STAMPS
  if (const char* sp = getenv("HB_STOP_PHASE")) P.stop_phase = atoi(sp);
#endif
  P.stage = staged(b);
  P.duo = b->tune[HB_TUNE_DUO];
  const bool sized_on = b->tune[HB_TUNE_SIZED] != 0;
  P.lean_ok = (b->D.dm.disableflags == 0 && b->tune[HB_TUNE_LEAN] ? 1 : 0) | (b->D.sized_h27 && sized_on ? 2 : 0) | (b->D.sized_team && sized_on ? 4 : 0);
  if (b->diag) P.stage.dm_fast = nullptr;  // the diagnostic buffers are laid out for the kernel of the model's own variant
  if (b->xfrc_std > 0.f && b->d_xfrc) {
    const double rate = b->xfrc_rate > 0.f ? std::exp(-b->model->m.timestep / b->xfrc_rate) : 0.0;  // trajectory.cc:149-150
    P.xfrc_rate = (float)rate; P.xfrc_scale = (float)(b->xfrc_std * std::sqrt(1.0 - rate * rate));
    P.xfrc_seed = b->xfrc_seed; P.xfrc_call = b->xfrc_calls++;
  }
  return P;
}


// order `stream` behind every pipe (no-op unless steps are in flight on the pipes)
void join_pipes(hb_batch* b) {
  if (b->forked) b->main_dirty = true;  // the batch's stream now carries the join: the next fork must carry it to the pipes
  if (!b->forked) return;
  for (int c = 0; c < b->npipe; c++) {
    if (hipEventRecord(b->ev_pipe[c], b->pipe[c]) != hipSuccess || hipStreamWaitEvent(b->stream, b->ev_pipe[c], 0) != hipSuccess) b->join_error = 1;
  }
  b->forked = false;
}
// the batch's stream, ordered behind all enqueued steps: every use of the stream outside launch_steps goes through here
int flush_steps(hb_batch* b);
hipStream_t main_stream(hb_batch* b) {
  if (b->fold_n && flush_steps(b) != HB_OK) b->join_error = 1;  // (hb_batch_sync reports it)
  join_pipes(b);
  b->main_dirty = true;  // the caller is about to enqueue something the next step's launches must follow
  return b->stream;
}

// heavy-first re-sort every N-th step call (HB_REORDER_PERIOD overrides, for experiments)
int reorder_period(const hb_batch* b) { return b->tune[HB_TUNE_REORDER_PERIOD] < 1 ? 1 : b->tune[HB_TUNE_REORDER_PERIOD]; }
// number of segments the next step call is cut into (1: one launch on the batch's stream)
int segment_count(const hb_batch* b) { return (b->npipe > 1 && !b->time_steps && b->n_env >= 64 * b->npipe) ? b->npipe : 1; }
struct Segment { int lo, hi; hipStream_t st; };
Segment segment(hb_batch* b, int c, int nseg) {
  if (nseg == 1) return {0, b->n_env, b->stream};
  return {(int)((long long)b->n_env * c / nseg), (int)((long long)b->n_env * (c + 1) / nseg), b->pipe[c]};
}
// fork: the pipes see everything enqueued on the batch's stream so far (controls written there, resets, ...)
int fork_pipes(hb_batch* b, int nseg) {
  // (step calls held back - fold_steps - come first whoever launches next; flush_steps itself gets here with nothing held any more)
  if (b->fold_n) { const int rc = flush_steps(b); if (rc != HB_OK) return rc; }
  if (nseg == 1) { join_pipes(b); return HB_OK; }
  // (nothing enqueued on the batch's stream since the last fork: the pipes already follow all of it, and a marker on a stream that
  // shares a hardware queue with a busy one would wait behind that one's work)
  if (b->main_dirty) {
    HB_HIP(hipEventRecord(b->ev_fork, b->stream));
    for (int c = 0; c < nseg; c++) HB_HIP(hipStreamWaitEvent(b->pipe[c], b->ev_fork, 0));
    b->main_dirty = false;
  }
  b->forked = true;
  return HB_OK;
}
// one segment's launch of the step kernel, then (heavy-first scheduling, every 4th call) the tiny kernel that
// orders the segment's next launch by the cost of this one; costs change slowly, and the sort sits on the
// critical path of its stream
int launch_segment(hb_batch* b, BatchPtrs P, int nsteps, const Segment& sg, int nseg, bool reorder) {
  P.blk0 = sg.lo; P.nblk = sg.hi - sg.lo;
  // a whole-batch permutation would mix segments: a segment only uses the order of its own envs
  P.order = (b->schedule && (nseg == 1 ? b->order_mode != 0 : b->order_mode == 2)) ? b->d_order : nullptr;
  P.order2 = (P.order && staged_on(b)) ? b->d_order2 : nullptr;
  HB_HIP(launch_step(b->D.d_dm, b->D.dm.variant, b->D.dm.solver, b->D.dm.nv, b->D.dm.lds_floats, P, nsteps, sg.st)); b->last_kernel = last_step_kernel();
  // (the key of the counting sort is 8 bits of the cost: of a single step's rows x sweeps - up to ~ 1600 - the bits above the lowest three; of the
  // AVERAGE over a launch of several steps, which the two-envs-per-wave kernel leaves behind and pairs its envs by - 180 .. 700 -, one bit more)
  if (reorder) HB_HIP(launch_order(b->d_counts, b->d_order, b->n_env, sg.lo, sg.hi - sg.lo, sg.st, /*slot=*/3, /*shift=*/(nsteps >= 8 && b->D.dm.variant == 0) ? 2 : 3));
  if (reorder && staged_on(b)) HB_HIP(launch_order(b->d_counts, b->d_order2, b->n_env, sg.lo, sg.hi - sg.lo, sg.st, /*slot=*/7, /*shift=*/0));
  return HB_OK;
}
// `refreshed`: the launch rewrote the permutations itself (launch_step's in-rollout refresh of a staged multi-step launch sorts the slots
// of each launch: the whole batch when there is one segment, each segment's own envs otherwise)
void steps_enqueued(hb_batch* b, int nseg, bool reorder, bool refreshed = false) {
  if (reorder || refreshed) b->order_mode = nseg == 1 ? 1 : 2;
  b->launch_count++;
}

int launch_steps_now(hb_batch* b, BatchPtrs& P, int nsteps, int ncalls = 1) {
  // a launch of several steps has no batch-wide barrier between its steps: nothing for segments to overlap, and three launches that each
  // bring their own rounds of waves fill the chip worse than one (4096 envs, 64 steps: 103 us per step against 71, profiles/r04_fold_sizes.txt)
  const int nseg = (b->D.dm.variant == 0 && nsteps >= 5) ? 1 : segment_count(b);
  const bool sample = nseg == 1 && b->time_steps && (b->launch_count % 8 == 0) && b->tev_used + 2 <= (int)b->tev.size();
  // (a launch of several steps is followed by its re-sort every time: it pairs its envs by the order, and one sort is nothing beside it)
  const bool reorder = b->schedule && (ncalls >= reorder_period(b) || nsteps >= 8 || b->launch_count % reorder_period(b) == 0);
  int rc = fork_pipes(b, nseg);
  if (rc != HB_OK) return rc;
  if (sample) HB_HIP(hipEventRecord(b->tev[b->tev_used], b->stream));
  for (int c = 0; c < nseg; c++) {
    rc = launch_segment(b, P, nsteps, segment(b, c, nseg), nseg, reorder);
    if (rc != HB_OK) return rc;
  }
  if (sample) { HB_HIP(hipEventRecord(b->tev[b->tev_used + 1], b->stream)); b->tev_used += 2; }
  // (the condition of launch_step's refresh: staged, ordered, at least one (t & 7) == 7 with a step behind it)
  const bool refreshed = staged_on(b) && b->D.dm.variant != 0 && b->schedule && nsteps > 8 && (nseg == 1 ? b->order_mode != 0 : b->order_mode == 2);
  steps_enqueued(b, nseg, reorder, refreshed);
  return HB_OK;
}

// Step calls enqueued back to back run as ONE launch.  hb_step_dev is asynchronous: until the caller synchronises, reads something or
// enqueues other work (all of which pass main_stream), nobody can tell K launches of one step from one launch of K steps - except the clock:
// a launch of one step lasts as long as its slowest env and the next one waits for it, a launch of K steps lets every wave run on into
// its envs' next step (the rollout kernels: no batch-wide barrier).  That pays when the launch's waves are all on the chip at once
// (hb_step.hip: fold_pays - up to 2048 envs, 4096 for the models with the two-envs-per-wave kernel).  Such a step call is held back (its
// launch parameters and its control pointer) until one
// of: kFoldMax steps are held, a call with other parameters arrives, anything touches the batch's stream.  The held calls then run as
// one multi-step launch whose step t reads the controls of call t (BatchPtrs::ctrl_tab, ctrl_mode 3).  Results are bit-identical to
// the unfolded launches (tests/test_gpu_fold.py); HB_TUNE_FOLD = 1 switches it off.
// Only for a PIPELINED batch: its caller has already taken on the one obligation this adds - hb_batch_join (or fetching the stream again)
// before enqueueing work of its own on the batch's stream behind step calls (include/hb.h: hb_batch_pipeline).  An unpipelined batch
// keeps its plain stream semantics: every call is launched when it is made.
int flush_steps(hb_batch* b) {
  if (!b->fold_n) return HB_OK;
  BatchPtrs P = b->fold_P;
  const int n = b->fold_n;
  b->fold_n = 0;  // (first: the launch below passes fork_pipes / join_pipes, never main_stream, but nothing may re-enter with steps held)
  HB_HIP(hipSetDevice(b->device));
  bool same = true;
  for (int t = 1; t < n; t++) same = same && b->fold_ctrl[t] == b->fold_ctrl[0];
  if (same) { P.ctrl = b->fold_ctrl[0]; P.ctrl_mode = 0; }  // (one call, with or without substeps: exactly the launch it always was)
  else { P.ctrl = nullptr; P.ctrl_mode = 3; for (int t = 0; t < n; t++) P.ctrl_tab[t] = b->fold_ctrl[t]; }
  return launch_steps_now(b, P, n, n);
}
int launch_steps(hb_batch* b, BatchPtrs& P, int nsteps, bool foldable = false) {
  const int cap = std::min(b->tune[HB_TUNE_FOLD], kFoldMax);
#ifdef HB_STAMPS
  foldable = false;  // (the diagnostic build samples single launches)
#endif
  foldable = foldable && b->npipe > 1 && cap > 1 && nsteps <= cap && P.ctrl_mode == 0 && !b->time_steps && !b->diag && !P.stamps && P.xfrc_scale == 0.f &&
             fold_pays(b->D.dm.variant, b->D.dm.solver, b->D.dm.nv, P);
  if (!foldable) {
    const int rc = flush_steps(b);
    return rc != HB_OK ? rc : launch_steps_now(b, P, nsteps);
  }
  BatchPtrs key = P;
  key.ctrl = nullptr;
  if (b->fold_n && (memcmp(&key, &b->fold_P, sizeof key) != 0 || b->fold_n + nsteps > cap)) {
    const int rc = flush_steps(b);
    if (rc != HB_OK) return rc;
  }
  if (!b->fold_n) b->fold_P = key;
  for (int t = 0; t < nsteps; t++) b->fold_ctrl[b->fold_n++] = P.ctrl;
  return b->fold_n >= cap ? flush_steps(b) : HB_OK;
}

// field offsets of the per-env state record for a state spec
struct SpecLayout { int total; };
int spec_size(const Model& m, unsigned spec) {
  int n = 0;
  if (spec & HB_STATE_TIME) n += 1;
  if (spec & HB_STATE_QPOS) n += m.nq;
  if (spec & HB_STATE_QVEL) n += m.nv;
  if (spec & HB_STATE_WARMSTART) n += m.nv;
  if (spec & HB_STATE_XFRC_APPLIED) n += 6 * m.nbody;
  return n;
}
const unsigned kSupportedSpec = HB_STATE_TIME | HB_STATE_QPOS | HB_STATE_QVEL | HB_STATE_WARMSTART | HB_STATE_XFRC_APPLIED;

template <class T>
int get_state_impl(hb_batch* b, unsigned spec, T* out) {
  if (!b || !out || (spec & ~kSupportedSpec) || !spec) return HB_EINVAL;
  const Model& m = b->model->m;
  int ns = b->D.dm.nstate, n = b->n_env, w = spec_size(m, spec);
  std::vector<float> host((size_t)n * ns), xf;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  HB_HIP(hipMemcpy(host.data(), b->d_state, host.size() * sizeof(float), hipMemcpyDeviceToHost));
  if (spec & HB_STATE_XFRC_APPLIED) {
    xf.assign((size_t)n * 6 * m.nbody, 0.f);
    if (b->d_xfrc) HB_HIP(hipMemcpy(xf.data(), b->d_xfrc, xf.size() * sizeof(float), hipMemcpyDeviceToHost));
  }
  for (int e = 0; e < n; e++) {
    const float* s = &host[(size_t)e * ns];
    T* o = out + (size_t)e * w;
    if (spec & HB_STATE_TIME) *o++ = (T)s[0];
    if (spec & HB_STATE_QPOS) for (int i = 0; i < m.nq; i++) *o++ = (T)s[1 + i];
    if (spec & HB_STATE_QVEL) for (int i = 0; i < m.nv; i++) *o++ = (T)s[1 + m.nq + i];
    if (spec & HB_STATE_WARMSTART) for (int i = 0; i < m.nv; i++) *o++ = (T)s[1 + m.nq + m.nv + i];
    if (spec & HB_STATE_XFRC_APPLIED) for (int i = 0; i < 6 * m.nbody; i++) *o++ = (T)xf[(size_t)e * 6 * m.nbody + i];
  }
  return HB_OK;
}

template <class T>
int set_state_impl(hb_batch* b, unsigned spec, const T* in) {
  if (!b || !in || (spec & ~kSupportedSpec) || !spec) return HB_EINVAL;
  const Model& m = b->model->m;
  int ns = b->D.dm.nstate, n = b->n_env, w = spec_size(m, spec);
  std::vector<float> host((size_t)n * ns);
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  HB_HIP(hipMemcpy(host.data(), b->d_state, host.size() * sizeof(float), hipMemcpyDeviceToHost));
  std::vector<float> xf;
  if (spec & HB_STATE_XFRC_APPLIED) xf.assign((size_t)n * 6 * m.nbody, 0.f);
  for (int e = 0; e < n; e++) {
    float* s = &host[(size_t)e * ns];
    const T* o = in + (size_t)e * w;
    if (spec & HB_STATE_TIME) s[0] = (float)*o++;
    if (spec & HB_STATE_QPOS) for (int i = 0; i < m.nq; i++) s[1 + i] = (float)*o++;
    if (spec & HB_STATE_QVEL) for (int i = 0; i < m.nv; i++) s[1 + m.nq + i] = (float)*o++;
    if (spec & HB_STATE_WARMSTART) for (int i = 0; i < m.nv; i++) s[1 + m.nq + m.nv + i] = (float)*o++;
    if (spec & HB_STATE_XFRC_APPLIED) for (int i = 0; i < 6 * m.nbody; i++) xf[(size_t)e * 6 * m.nbody + i] = (float)*o++;
  }
  HB_HIP(hipMemcpy(b->d_state, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  if (spec & HB_STATE_XFRC_APPLIED) {
    if (!b->d_xfrc) { if (hipMalloc((void**)&b->d_xfrc, xf.size() * sizeof(float)) != hipSuccess) return HB_ENOMEM; }
    HB_HIP(hipMemcpy(b->d_xfrc, xf.data(), xf.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return HB_OK;
}

template <class T>
int set_state_broadcast_impl(hb_batch* b, unsigned spec, const T* one) {
  if (!b || !one) return HB_EINVAL;
  if ((spec & ~kSupportedSpec) || !spec) return HB_EINVAL;
  const int w = spec_size(b->model->m, spec);
  std::vector<T> all((size_t)b->n_env * w);
  for (int e = 0; e < b->n_env; e++) memcpy(&all[(size_t)e * w], one, (size_t)w * sizeof(T));
  return set_state_impl<T>(b, spec, all.data());
}

}  // namespace

extern "C" {

const char* hb_version(void) { return "hb 0.1 (gfx950)"; }

hb_model* hb_model_load(const char* path, char* err, int err_sz) {
  if (!path) { set_err(err, err_sz, "null path"); return nullptr; }
  hb_model* h = nullptr;
  try {  // nothing may propagate through the C boundary (a malformed file can make the loaders throw bad_alloc / length_error)
    h = new hb_model;
    std::string e, p = path;
    bool ok = (p.size() > 4 && p.substr(p.size() - 4) == ".hbm") ? load_hbm(p, h->m, e) : compile_mjcf_file(p, h->m, e);
    if (!ok) { set_err(err, err_sz, e); delete h; return nullptr; }
    return h;
  } catch (const std::exception& ex) {
    set_err(err, err_sz, std::string("model load failed: ") + ex.what());
    delete h;
    return nullptr;
  }
}

hb_model* hb_model_load_xml_string(const char* xml, char* err, int err_sz) {
  if (!xml) { set_err(err, err_sz, "null xml"); return nullptr; }
  hb_model* h = nullptr;
  try {
    h = new hb_model;
    std::string e;
    if (!compile_mjcf_string(xml, h->m, e)) { set_err(err, err_sz, e); delete h; return nullptr; }
    return h;
  } catch (const std::exception& ex) {
    set_err(err, err_sz, std::string("model load failed: ") + ex.what());
    delete h;
    return nullptr;
  }
}

int hb_model_save(const hb_model* m, const char* path, char* err, int err_sz) {
  if (!m || !path) return HB_EINVAL;
  std::string e;
  if (!save_hbm(m->m, path, e)) { set_err(err, err_sz, e); return HB_EIO; }
  return HB_OK;
}

void hb_model_free(hb_model* m) { delete m; }

int hb_model_sizes(const hb_model* h, hb_sizes* out) {
  if (!h || !out) return HB_EINVAL;
  const Model& m = h->m;
  out->nq = m.nq; out->nv = m.nv; out->nu = m.nu; out->nbody = m.nbody; out->njnt = m.njnt; out->ngeom = m.ngeom; out->ntendon = m.ntendon;
  out->nM = m.nM; out->nkey = m.nkey; out->npair = m.npair;
  int nscalar = 0;
  for (int j = 0; j < m.njnt; j++) if (m.jnt_type[j] >= JNT_SLIDE) nscalar++;
  out->nobs = 2 * nscalar + 6;
  { int variant; std::string e; if (!model_variant(m, variant, out->ncon_max, out->nefc_max, e)) { out->ncon_max = kNconMax; out->nefc_max = kNefcMax; } }
  return HB_OK;
}

int hb_options_get(const hb_model* h, hb_options* o) {
  if (!h || !o) return HB_EINVAL;
  const Model& m = h->m;
  o->timestep = m.timestep; memcpy(o->gravity, m.gravity, sizeof o->gravity); o->impratio = m.impratio; o->tolerance = m.tolerance;
  o->iterations = m.iterations; o->solver = m.solver; o->cone = m.cone; o->integrator = m.integrator; o->disableflags = m.disableflags;
  o->ls_iterations = m.ls_iterations; o->ls_tolerance = m.ls_tolerance;
  return HB_OK;
}

int hb_options_set(hb_model* h, const hb_options* o) {
  if (!h || !o) return HB_EINVAL;
  if ((o->solver != SOL_PGS && o->solver != SOL_NEWTON) || o->cone != 0 || o->integrator != 0) return HB_EUNSUPPORTED;
  if (!(o->timestep > 0) || !(o->impratio > 0) || o->iterations < 0 || o->ls_iterations < 0 || !(o->ls_tolerance >= 0)) return HB_EINVAL;
  Model& m = h->m;
  m.timestep = o->timestep; memcpy(m.gravity, o->gravity, sizeof o->gravity); m.impratio = o->impratio; m.tolerance = o->tolerance;
  m.iterations = o->iterations; m.disableflags = o->disableflags; m.solver = o->solver;
  m.ls_iterations = o->ls_iterations; m.ls_tolerance = o->ls_tolerance;
  return HB_OK;
}

int hb_model_pair_order(hb_model* h, int order) {
  if (!h) return HB_EINVAL;
  if (order == 0 || order == 1) sort_pairs(h->m, order);
  else if (order != -1) return HB_EINVAL;
  return h->m.pair_order;
}

int hb_model_name2id(const hb_model* h, const char* kind, const char* name) {
  if (!h || !kind || !name) return -1;
  const Model& m = h->m;
  const std::vector<std::string>* v = nullptr;
  std::string k = kind;
  if (k == "body") v = &m.body_name; else if (k == "joint") v = &m.jnt_name; else if (k == "geom") v = &m.geom_name;
  else if (k == "actuator") v = &m.actuator_name; else if (k == "tendon") v = &m.tendon_name; else if (k == "key") v = &m.key_name;
  if (!v) return -1;
  for (size_t i = 0; i < v->size(); i++) if ((*v)[i] == name) return (int)i;
  return -1;
}

int hb_model_get_array(const hb_model* h, const char* field, double* out, int cap) {
  if (!h || !field) return HB_EINVAL;
  struct Finder {
    const char* want; const vecd* found = nullptr; const veci* foundi = nullptr;
    void operator()(const char* n, vecd& v) { if (!strcmp(n, want)) found = &v; }
    void operator()(const char* n, veci& v) { if (!strcmp(n, want)) foundi = &v; }
    void operator()(const char*, int&) {}
    void operator()(const char*, double&) {}
    void operator()(const char*, double*, int) {}
    void operator()(const char*, std::vector<std::string>&) {}
  } f;
  f.want = field;
  const_cast<Model&>(h->m).visit(f);
  if (f.found) { int n = (int)f.found->size(); if (out) for (int i = 0; i < n && i < cap; i++) out[i] = (*f.found)[i]; return n; }
  if (f.foundi) { int n = (int)f.foundi->size(); if (out) for (int i = 0; i < n && i < cap; i++) out[i] = (*f.foundi)[i]; return n; }
  return HB_EINVAL;
}

hb_batch* hb_batch_create(const hb_model* m, int n_env, int device, char* err, int err_sz) {
  if (!m || n_env <= 0) { set_err(err, err_sz, "bad arguments"); return nullptr; }
  int ndev = 0;
  if (device < 0 || hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) {
    set_err(err, err_sz, "no HIP device available: this engine has no CPU backend (device=" + std::to_string(device) + ", visible=" + std::to_string(ndev) + ")");
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) { set_err(err, err_sz, "hipSetDevice failed"); return nullptr; }
  hb_batch* b = new hb_batch;
  b->model = m; b->n_env = n_env; b->device = device;
  std::string e;
  if (!build_device_model(m->m, b->D, e)) { set_err(err, err_sz, e); delete b; return nullptr; }
  const DevModel& dm = b->D.dm;
  bool ok = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming) == hipSuccess;
  // (the pipes' streams are created when hb_batch_pipeline asks for them: the runtime maps streams
  // onto four hardware queues, and streams that share one serialise - a batch should not own more streams than it uses)
  ok = ok && hipMalloc((void**)&b->d_state, (size_t)n_env * dm.nstate * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc((void**)&b->d_status, (size_t)n_env * sizeof(int)) == hipSuccess;
  ok = ok && hipMalloc((void**)&b->d_counts, (size_t)n_env * kCountStride * sizeof(int)) == hipSuccess;
  ok = ok && hipMemset(b->d_counts, 0, (size_t)n_env * kCountStride * sizeof(int)) == hipSuccess;
  ok = ok && ensure_ctrl(b, (size_t)n_env * std::max(1, dm.nu)) == HB_OK;
  ok = ok && hipMalloc((void**)&b->d_order, (size_t)2 * n_env * sizeof(int)) == hipSuccess;  // the permutation | the sort keys of a pass
  ok = ok && hipMalloc((void**)&b->d_order2, (size_t)2 * n_env * sizeof(int)) == hipSuccess;
  // general variants: the staged step (pose -> narrowphase -> step kernels); hb_batch_tune(HB_TUNE_STAGED, 0) keeps everything in the step kernel
  if (dm.variant != 0) {
    StageBufs& sb = b->stage;
    ok = ok && hipMalloc((void**)&sb.geom, (size_t)n_env * std::max(1, dm.ngeom) * 10 * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void**)&sb.item, (size_t)n_env * kWorkMax * sizeof(int4)) == hipSuccess;
    ok = ok && hipMalloc((void**)&sb.nwork, (size_t)n_env * sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc((void**)&sb.nsearch, (size_t)n_env * 2 * sizeof(int)) == hipSuccess;
    ok = ok && hipMemset(sb.nsearch, 0, (size_t)n_env * 2 * sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc((void**)&sb.result, (size_t)n_env * kWorkMax * 4 * sizeof(float4)) == hipSuccess;
    ok = ok && hipMemset(sb.nwork, 0, (size_t)n_env * sizeof(int)) == hipSuccess;
    sb.nq = dm.nq; sb.nv = dm.nv; sb.nu = dm.nu;
    sb.no_mesh = b->model->m.nmesh == 0 ? 1 : 0;
    sb.pose_lds = pose_lds_floats(dm.nq, dm.nbody, dm.ngeom) * (int)sizeof(float);
    if (dm.variant == 1 || b->D.d_dm_fast) {
      ok = ok && hipMalloc((void**)&sb.defer, (size_t)n_env * 3 * sizeof(int)) == hipSuccess;  // flags | list | counters (StageBufs)
      ok = ok && hipMemset(sb.defer, 0, (size_t)n_env * 3 * sizeof(int)) == hipSuccess;
      if (ok) { sb.defer_list = sb.defer + n_env; sb.defer_count = sb.defer + 2 * (size_t)n_env; }
      if (dm.variant == 2 || dm.variant == 3) { sb.dm_fast = b->D.d_dm_fast; sb.fast_lds = b->D.fast_lds_floats * (int)sizeof(float); }
      if (ok && sb.fast_lds > 64 * 1024) ok = set_step_lds_limit(sb.fast_lds) == hipSuccess;
    }
  }
  if (hb_debug()) fprintf(stderr, "[hb] LDS per env: %d bytes (%d envs per CU)\n", dm.lds_floats * 4, 160 * 1024 / (dm.lds_floats * 4));
  if (hb_debug()) fprintf(stderr, "[hb] tree levels %d, limit candidates %d, trees %d; size-specialised kernel: %s\n", dm.nlevel, dm.nlimcand, dm.ntree, (b->D.sized_h27 || b->D.sized_team) ? "yes" : "no");
  if (hb_debug() && b->D.fast_lds_floats) fprintf(stderr, "[hb] LDS per env of the staged step's fast kernel: %d bytes (%d envs per CU)\n", b->D.fast_lds_floats * 4, 160 * 1024 / (b->D.fast_lds_floats * 4));
  if (ok && dm.lds_floats * 4 > 64 * 1024) ok = set_step_lds_limit(dm.lds_floats * 4) == hipSuccess;
  if (!ok) { set_err(err, err_sz, "device allocation failed"); hb_batch_free(b); return nullptr; }
  if (hb_reset(b, nullptr, -1, 0, 0) != HB_OK) { set_err(err, err_sz, "initial reset failed"); hb_batch_free(b); return nullptr; }
  return b;
}

static void envrand_free_fwd(hb_batch* b);
void hb_batch_free(hb_batch* b) {
  if (!b) return;
  b->fold_n = 0;  // (step calls nobody will read the results of)
  HB_IGN(hipSetDevice(b->device));
  for (int c = 0; c < hb_batch::kPipes; c++) {
    if (b->pipe[c] && b->pipe[c] != b->stream) { HB_IGN(hipStreamSynchronize(b->pipe[c])); HB_IGN(hipStreamDestroy(b->pipe[c])); }
    if (b->ev_pipe[c]) HB_IGN(hipEventDestroy(b->ev_pipe[c]));
  }
  if (b->ev_fork) HB_IGN(hipEventDestroy(b->ev_fork));
  if (b->stream) { HB_IGN(hipStreamSynchronize(b->stream)); HB_IGN(hipStreamDestroy(b->stream)); }
  for (auto e : b->tev) HB_IGN(hipEventDestroy(e));
  for (int i = 0; i < 4; i++) { if (b->d_mlp_w[i]) HB_IGN(hipFree(b->d_mlp_w[i])); if (b->d_mlp_b[i]) HB_IGN(hipFree(b->d_mlp_b[i])); if (b->d_mlp_wp[i]) HB_IGN(hipFree(b->d_mlp_wp[i])); }
  for (int i = 0; i < 2; i++) if (b->d_mlp_h[i]) HB_IGN(hipFree(b->d_mlp_h[i]));
  if (b->d_mlp_act) HB_IGN(hipFree(b->d_mlp_act));
  if (b->ev0) HB_IGN(hipEventDestroy(b->ev0));
  if (b->ev1) HB_IGN(hipEventDestroy(b->ev1));
  envrand_free_fwd(b);
  if (b->d_sensor_out) HB_IGN(hipFree(b->d_sensor_out));
  if (b->d_dr) HB_IGN(hipFree(b->d_dr));
  if (b->d_rmask) HB_IGN(hipFree(b->d_rmask));
  if (b->d_pending) HB_IGN(hipFree(b->d_pending));
  void* ptrs[] = {b->stage.geom, b->stage.item, b->stage.nsearch, b->stage.nwork, b->stage.result, b->stage.defer, b->d_term_obs, b->d_seen, b->d_state, b->d_ctrl, b->d_xfrc, b->d_diag_qacc, b->d_diag_force, b->d_diag_contact, b->d_obs, b->d_mask,
                  b->d_status, b->d_counts, b->d_qpos_out, b->d_qvel_out, b->d_task_out, b->d_knots, b->d_order, b->d_order2, b->d_prev, b->d_latest, b->d_qfrc, b->d_action, b->d_episode};
  for (void* p : ptrs) if (p) HB_IGN(hipFree(p));
  delete b;
}

int hb_batch_n_env(const hb_batch* b) { return b ? b->n_env : HB_EINVAL; }
void* hb_batch_stream(hb_batch* b) { return b ? (void*)main_stream(b) : nullptr; }
int hb_batch_sync(hb_batch* b) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  if (b->join_error) { b->join_error = 0; return HB_ENODEVICE; }
  return HB_OK;
}
// the streams of the first n segments, created on first use.  Segment 0 runs on the batch's own stream: one stream fewer for the
// process' hardware queues (below), and nothing else is enqueued there while steps are in flight
static int make_pipes(hb_batch* b, int n) {
  for (int c = 0; c < n; c++) {
    if (!b->pipe[c] && c == 0) b->pipe[0] = b->stream;
    if (!b->pipe[c]) {
      // (streams of different priorities - ROCm keeps one pool of hardware queues per priority level - were measured too: 95 us per step for three
      // segments against 88)
      const hipError_t e = hipStreamCreateWithFlags(&b->pipe[c], hipStreamNonBlocking);
      if (e != hipSuccess) return HB_ENOMEM;
    }
    if (!b->ev_pipe[c] && hipEventCreateWithFlags(&b->ev_pipe[c], hipEventDisableTiming) != hipSuccess) return HB_ENOMEM;
  }
  return HB_OK;
}
// Do kernels on the first n segment streams run at the same time?  ROCm maps a process' streams onto GPU_MAX_HW_QUEUES (default 4)
// hardware queues, the null stream included, and two streams on one queue run their kernels one after the other: three segments
// of which two share a queue step 4096 envs in 147 us instead of 88.  Nothing in HIP tells which queue a stream got, so this looks:
// one idling wave per stream (60 us each), all n overlapping pairwise in the GPU's own clock or not.
// returns -1 when all n run side by side, else the higher stream index of the first pair that does not (-2: the probe itself failed)
static int pipes_conflict(hb_batch* b, int n) {
  unsigned long long* d = nullptr;
  unsigned long long h[2 * hb_batch::kPipes] = {};
  if (hipStreamSynchronize(b->stream) != hipSuccess || hipMalloc((void**)&d, sizeof h) != hipSuccess) return -2;
  bool ok = hipMemset(d, 0, sizeof h) == hipSuccess;
  for (int c = 0; c < n && ok; c++) ok = launch_probe_spin(d + 2 * c, 6000, b->pipe[c]) == hipSuccess;
  for (int c = 0; c < n; c++) ok = hipStreamSynchronize(b->pipe[c]) == hipSuccess && ok;
  ok = ok && hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
  HB_IGN(hipFree(d));
  if (!ok) return -2;
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (!(h[2 * i] < h[2 * j + 1] && h[2 * j] < h[2 * i + 1] && h[2 * i + 1] > h[2 * i] && h[2 * j + 1] > h[2 * j])) return j;
  return -1;
}
int hb_batch_pipeline(hb_batch* b, int on) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  (void)main_stream(b);
  if (on < 0 || on > hb_batch::kPipes) return HB_EINVAL;
  // 1: default segment count.  Measured on MI355X, 4096 envs, us per step (tools/gpu_pipeline_queues.py, every case a fresh process):
  // 2 segments 92, 3 segments 88, 4 segments 87 with GPU_MAX_HW_QUEUES >= 5 and 142 without, 5 and more 107 - 170 however many queues
  // the runtime is given (a fifth active queue is multiplexed).  Three it is, when the three streams are seen to run side by side
  // (other streams of the process can take the queues: pipes_conflict); two otherwise.  No environment is set here.
  int want = on == 1 ? 3 : on;
  int rc = make_pipes(b, want);
  if (rc != HB_OK) return rc;
  if (on == 1 && b->probed_segments) want = b->probed_segments;  // (probed before on these very streams: no second probe, the same shape every time)
  else if (on == 1) {
    // a stream that shares its queue is replaced by a fresh one (the runtime deals queues out in turn: the next stream lands on another
    // one) while the old one still holds its place; a few tries, then two segments
    hipStream_t spare[4] = {};
    int bad = pipes_conflict(b, want);
    for (int t = 0; t < 4 && bad > 0; t++) {
      hipStream_t s = nullptr;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
      spare[t] = b->pipe[bad];
      b->pipe[bad] = s;
      bad = pipes_conflict(b, want);
    }
    for (hipStream_t s : spare) if (s) HB_IGN(hipStreamDestroy(s));
    if (bad == -2) return HB_ENODEVICE;  // the probe itself failed (a HIP error, not a shared queue): an error, not a quiet two segments
    if (bad != -1) want = 2;
    b->probed_segments = want;
    if (hb_debug()) fprintf(stderr, "[hb] hb_batch_pipeline: %d env segments (%s)\n", want, bad == -1 ? "every segment stream has a hardware queue of its own" : "two of three segment streams share a hardware queue");
  }
  b->npipe = want;
  if (b->order_mode == 2) b->order_mode = 0;  // segment boundaries changed: per-segment permutations are stale (one of the whole batch stays good for unsegmented launches)
  return HB_OK;
}
int hb_batch_segments(const hb_batch* b) { return b ? segment_count(b) : HB_EINVAL; }
int hb_batch_join(hb_batch* b) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  (void)main_stream(b);  // (launches the step calls held back, then joins the segments' streams)
  return b->join_error ? HB_ENODEVICE : HB_OK;
}

static int reset_impl(hb_batch* b, const uint8_t* mask, int keyframe, float perturb_scale, int env_offset) {
  if (!b) return HB_EINVAL;
  const Model& m = b->model->m;
  if (keyframe >= m.nkey) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  const uint8_t* dmask = nullptr;
  if (mask) {
    if (!b->d_mask && hipMalloc((void**)&b->d_mask, b->n_env) != hipSuccess) return HB_ENOMEM;
    HB_HIP(hipMemcpyAsync(b->d_mask, mask, b->n_env, hipMemcpyHostToDevice, main_stream(b)));
    dmask = b->d_mask;
  }
  b->env_offset = env_offset;
  const float* src = b->D.d_qpos_src + (keyframe < 0 ? 0 : (size_t)(1 + keyframe) * m.nq);
  HB_HIP(launch_reset(b->D.dm, b->d_state, b->d_status, dmask, src, nullptr, b->n_env, perturb_scale, env_offset, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_reset(hb_batch* b, const uint8_t* mask, int keyframe, int perturb, int env_offset) {
  return reset_impl(b, mask, keyframe, perturb ? 1.f : 0.f, env_offset);
}

int hb_step_dev(hb_batch* b, const float* ctrl_dev, int n_substeps) {
  if (!b || n_substeps < 1 || (!ctrl_dev && b->D.dm.nu > 0)) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  BatchPtrs P = make_ptrs(b);
  P.ctrl = ctrl_dev; P.ctrl_mode = 0;
  return launch_steps(b, P, n_substeps, /*foldable=*/true);
}

int hb_step(hb_batch* b, const float* ctrl, int n_substeps) {
  if (!b || n_substeps < 1 || (!ctrl && b->D.dm.nu > 0)) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  size_t n = (size_t)b->n_env * b->D.dm.nu;
  if (n) HB_HIP(hipMemcpyAsync(ctrl_for_write(b), ctrl, n * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
  int rc = hb_step_dev(b, b->d_ctrl, n_substeps);
  if (rc != HB_OK) return rc;
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_forward(hb_batch* b, const float* ctrl) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  size_t n = (size_t)b->n_env * b->D.dm.nu;
  if (ctrl && n) HB_HIP(hipMemcpyAsync(ctrl_for_write(b), ctrl, n * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
  else if (n) HB_HIP(hipMemsetAsync(ctrl_for_write(b), 0, n * sizeof(float), main_stream(b)));
  BatchPtrs P = make_ptrs(b);
  P.ctrl = b->d_ctrl; P.ctrl_mode = 0; P.integrate = 0;
  HB_HIP(launch_step(b->D.d_dm, b->D.dm.variant, b->D.dm.solver, b->D.dm.nv, b->D.dm.lds_floats, P, 1, main_stream(b))); b->last_kernel = last_step_kernel();
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_rollout_dev(hb_batch* b, const float* ctrl_dev, int T, float* qpos_out_dev) {
  if (!b || T < 1 || (!ctrl_dev && b->D.dm.nu > 0)) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  BatchPtrs P = make_ptrs(b);
  P.ctrl = ctrl_dev; P.ctrl_mode = 1; P.qpos_out = qpos_out_dev;
  return launch_steps(b, P, T);
}

int hb_rollout(hb_batch* b, const float* ctrl, int T, float* qpos_out) {
  if (!b || T < 1 || (!ctrl && b->D.dm.nu > 0)) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  size_t n = (size_t)T * b->n_env * b->D.dm.nu;
  int rc = ensure_ctrl(b, std::max<size_t>(n, 1));
  if (rc != HB_OK) return rc;
  if (n) HB_HIP(hipMemcpyAsync(ctrl_for_write(b), ctrl, n * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
  size_t nq_out = (size_t)T * b->n_env * b->D.dm.nq;
  if (qpos_out && nq_out > b->qpos_out_cap) {
    if (b->d_qpos_out) HB_IGN(hipFree(b->d_qpos_out));
    b->d_qpos_out = nullptr; b->qpos_out_cap = 0;
    if (hipMalloc((void**)&b->d_qpos_out, nq_out * sizeof(float)) != hipSuccess) return HB_ENOMEM;
    b->qpos_out_cap = nq_out;
  }
  rc = hb_rollout_dev(b, b->d_ctrl, T, qpos_out ? b->d_qpos_out : nullptr);
  if (rc != HB_OK) return rc;
  if (qpos_out) HB_HIP(hipMemcpyAsync(qpos_out, b->d_qpos_out, nq_out * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

// grows a device trace buffer to `need` floats
static int ensure_trace(float** buf, size_t* cap, size_t need) {
  if (need <= *cap) return HB_OK;
  if (*buf) HB_IGN(hipFree(*buf));
  *buf = nullptr; *cap = 0;
  if (hipMalloc((void**)buf, need * sizeof(float)) != hipSuccess) return HB_ENOMEM;
  *cap = need;
  return HB_OK;
}

// ---- mjd_transitionFD over a batch ------------------------------------------------------------------------------
namespace {
// mj_integratePos (mujoco.h:466) with dt = 1 on one state: qpos <- qpos (+) dq, dq in R^nv
void integrate_pos(const Model& m, double* qpos, const double* dq) {
  for (int j = 0; j < m.njnt; j++) {
    const int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
    if (m.jnt_type[j] == JNT_FREE) {
      for (int i = 0; i < 3; i++) qpos[qa + i] += dq[da + i];
      double v[3] = {dq[da + 3], dq[da + 4], dq[da + 5]};
      const double ang = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      if (ang > 1e-15) {
        const double s = std::sin(0.5 * ang) / ang, c = std::cos(0.5 * ang);
        const double r[4] = {c, v[0] * s, v[1] * s, v[2] * s};
        double* q = qpos + qa + 3;
        const double o[4] = {q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3], q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2],
                             q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1], q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0]};
        const double n = std::sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
        for (int i = 0; i < 4; i++) q[i] = o[i] / n;
      }
    } else qpos[qa] += dq[da];
  }
}
// mj_differentiatePos (mujoco.h:463) with dt = 1: dq = q2 (-) q1 in R^nv
void differentiate_pos(const Model& m, double* dq, const double* q1, const double* q2) {
  for (int j = 0; j < m.njnt; j++) {
    const int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
    if (m.jnt_type[j] == JNT_FREE) {
      for (int i = 0; i < 3; i++) dq[da + i] = q2[qa + i] - q1[qa + i];
      // rotation taking q1 to q2, in q1's frame: conj(q1) * q2, as a rotation vector (mju_subQuat)
      const double* a = q1 + qa + 3;
      const double* c = q2 + qa + 3;
      double d[4] = {a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3], a[0] * c[1] - a[1] * c[0] - a[2] * c[3] + a[3] * c[2],
                     a[0] * c[2] + a[1] * c[3] - a[2] * c[0] - a[3] * c[1], a[0] * c[3] - a[1] * c[2] + a[2] * c[1] - a[3] * c[0]};
      const double sn = std::sqrt(d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
      double ang = 2 * std::atan2(sn, d[0]);
      if (ang > M_PI) ang -= 2 * M_PI;
      const double k = sn > 1e-15 ? ang / sn : 0.0;
      for (int i = 0; i < 3; i++) dq[da + 3 + i] = d[1 + i] * k;
    } else dq[da] = q2[qa] - q1[qa];
  }
}
}  // namespace

static int transition_fd_impl(hb_batch* b, const double* x, const double* u, const double* warm, int T, double eps, int centered, const hb_sensor_spec* spec,
                              double* A, double* B, double* C, double* D) {
  if (!b || !x || T < 1 || !(eps > 0) || (!u && b->D.dm.nu > 0) || (!A && !B && !C && !D) || ((C || D) && !spec)) return HB_EINVAL;
  const Model& m = b->model->m;
  const int nq = m.nq, nv = m.nv, nu = m.nu, nx = 2 * nv, ncol = nx + nu, k = centered ? 2 : 1, per = 1 + k * ncol;
  if ((long long)T * per > b->n_env) return HB_EINVAL;
  const int N = b->n_env, rec = 1 + nq + 2 * nv;
  std::vector<double> st((size_t)N * rec, 0.0), step_of((size_t)T * per, 0.0);
  std::vector<float> ctrl((size_t)N * std::max(1, nu), 0.f);
  // unused envs: a valid rest state (they step along and are ignored)
  for (int e = T * per; e < N; e++) for (int i = 0; i < nq; i++) st[(size_t)e * rec + 1 + i] = m.qpos0[i];
  std::vector<double> dq(nv);
  for (int t = 0; t < T; t++) {
    const double* xt = x + (size_t)t * (nq + nv);
    for (int c = 0; c < per; c++) {
      const int e = t * per + c;
      double* s = st.data() + (size_t)e * rec;
      for (int i = 0; i < nq; i++) s[1 + i] = xt[i];
      for (int i = 0; i < nv; i++) s[1 + nq + i] = xt[nq + i];
      if (warm) for (int i = 0; i < nv; i++) s[1 + nq + nv + i] = warm[(size_t)t * nv + i];
      for (int i = 0; i < nu; i++) ctrl[(size_t)e * nu + i] = (float)u[(size_t)t * nu + i];
      if (c == 0) continue;
      const int col = (c - 1) % ncol;
      const double sign = (c - 1) / ncol == 0 ? 1.0 : -1.0;  // second block: the minus side of a centered difference
      double h = sign * eps;
      if (col < nv) {  // position, in the tangent space
        std::fill(dq.begin(), dq.end(), 0.0);
        dq[col] = h;
        integrate_pos(m, s + 1, dq.data());
      } else if (col < nx) s[1 + nq + (col - nv)] += h;
      else {
        const int a = col - nx;
        double v = u[(size_t)t * nu + a] + h;
        if (m.actuator_ctrllimited[a]) v = std::min(std::max(v, m.actuator_ctrlrange[2 * a]), m.actuator_ctrlrange[2 * a + 1]);  // nudge inside the range
        ctrl[(size_t)e * nu + a] = (float)v;
        h = v - u[(size_t)t * nu + a];
      }
      step_of[e] = h;  // the step actually taken
    }
  }
  int rc = hb_set_state_f64(b, HB_STATE_INTEGRATION, st.data());
  if (rc != HB_OK) return rc;
  std::vector<float> rows;
  int ns = 0;
  if (C || D) {
    // one step with the read-out row of every env (evaluated in the forward pass, before the integration)
    ns = hb_sensor_size(spec);
    if (ns <= 0) return HB_EINVAL;
    rows.resize((size_t)N * ns);
    rc = hb_rollout_sensors(b, ctrl.data(), 1, spec, rows.data(), nullptr);
  } else rc = hb_step(b, ctrl.data(), 1);
  if (rc != HB_OK) return rc;
  rc = hb_get_state_f64(b, HB_STATE_INTEGRATION, st.data());
  if (rc != HB_OK) return rc;
  std::vector<double> dp(nx), dm(nx);
  auto diff = [&](int e_ref, int e, double* out) {  // x'(e) (-) x'(e_ref) in tangent coordinates
    const double* r = st.data() + (size_t)e_ref * rec;
    const double* s = st.data() + (size_t)e * rec;
    differentiate_pos(m, out, r + 1, s + 1);
    for (int i = 0; i < nv; i++) out[nv + i] = s[1 + nq + i] - r[1 + nq + i];
  };
  for (int t = 0; t < T; t++) {
    const int e0 = t * per;
    for (int col = 0; col < ncol; col++) {
      const int ep = e0 + 1 + col, em = centered ? e0 + 1 + ncol + col : e0;
      const double hp = step_of[ep], hm = centered ? step_of[em] : 0.0;
      std::vector<double> d(nx, 0.0);
      if (hp - hm != 0.0) {
        diff(e0, ep, dp.data());
        if (centered) diff(e0, em, dm.data()); else std::fill(dm.begin(), dm.end(), 0.0);
        for (int i = 0; i < nx; i++) d[i] = (dp[i] - dm[i]) / (hp - hm);
      }
      if (col < nx) { if (A) for (int i = 0; i < nx; i++) A[((size_t)t * nx + i) * nx + col] = d[i]; }
      else if (B) for (int i = 0; i < nx; i++) B[((size_t)t * nx + i) * nu + (col - nx)] = d[i];
      if (C || D) {
        const float* rp = rows.data() + (size_t)ep * ns;
        const float* rm = rows.data() + (size_t)em * ns;
        for (int i = 0; i < ns; i++) {
          const double g = hp - hm != 0.0 ? ((double)rp[i] - (double)rm[i]) / (hp - hm) : 0.0;
          if (col < nx) { if (C) C[((size_t)t * ns + i) * nx + col] = g; }
          else if (D) D[((size_t)t * ns + i) * nu + (col - nx)] = g;
        }
      }
    }
  }
  return HB_OK;
}

int hb_transition_fd(hb_batch* b, const double* x, const double* u, const double* warm, int T, double eps, int centered, double* A, double* B) {
  if (!A && !B) return HB_EINVAL;
  return transition_fd_impl(b, x, u, warm, T, eps, centered, nullptr, A, B, nullptr, nullptr);
}
int hb_transition_fd_sensors(hb_batch* b, const double* x, const double* u, const double* warm, int T, double eps, int centered, const hb_sensor_spec* spec,
                             double* A, double* B, double* C, double* D) {
  return transition_fd_impl(b, x, u, warm, T, eps, centered, spec, A, B, C, D);
}

int hb_ctrl_tape_splines(hb_batch* b, const float* knots, const float* times, int n_points, int interpolation, double time0, int T) {
  // an empty spline samples as zeros and a one-node spline as its node, whatever the interpolation (spline.cc:103-118; spline_test.cc:41-64)
  if (!b || n_points < 0 || n_points > 64 || (n_points > 0 && (!knots || !times)) || interpolation < 0 || interpolation > 2 || T < 1 || b->D.dm.nu < 1) return HB_EINVAL;
  for (int k = 1; k < n_points; k++) if (!(times[k] > times[k - 1])) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  const int N = b->n_env, nu = b->D.dm.nu;
  const size_t nk = (size_t)N * n_points * nu;
  int rc = ensure_trace(&b->d_knots, &b->knots_cap, nk + 64);
  if (rc != HB_OK) return rc;
  rc = ensure_ctrl(b, (size_t)T * N * nu);
  if (rc != HB_OK) return rc;
  if (n_points > 0) {
    HB_HIP(hipMemcpyAsync(b->d_knots, knots, nk * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
    HB_HIP(hipMemcpyAsync(b->d_knots + nk, times, (size_t)n_points * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
  }
  HB_HIP(launch_spline_tape(b->D.dm, b->d_knots, b->d_knots + nk, n_points, interpolation, (float)time0, (float)b->model->m.timestep, T, N, b->d_ctrl, main_stream(b)));
  b->tape_steps = T;
  return HB_OK;
}

int hb_ctrl_tape_read(hb_batch* b, int T, float* out) {
  if (!b || !out || T < 1 || T > b->tape_steps) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipMemcpyAsync(out, b->d_ctrl, (size_t)T * b->n_env * b->D.dm.nu * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_task_cost(hb_batch* b, const float* residual, int n, int n_residual, const hb_cost_spec* spec, float* terms, float* cost) {
  if (!b || !residual || !spec || !cost || n < 1 || n_residual < 1 || spec->n_term < 1 || spec->n_term > 8) return HB_EINVAL;
  CostSpec K;
  memset(&K, 0, sizeof K);
  int total_dim = 0;
  for (int k = 0; k < spec->n_term; k++) {
    if (spec->dim[k] < 1 || spec->norm[k] < -1 || spec->norm[k] > 8 || spec->norm[k] == 4) return HB_EINVAL;  // kJunction (4) has no value-only form here
    K.dim[k] = spec->dim[k]; K.norm[k] = spec->norm[k]; K.weight[k] = spec->weight[k]; K.p[k] = spec->norm_p[k][0]; K.q[k] = spec->norm_p[k][1];
    total_dim += spec->dim[k];
  }
  if (total_dim != n_residual) return HB_EINVAL;  // "mismatch between total user-sensor dimension and actual length of residual"
  K.nterm = spec->n_term; K.risk = spec->risk;
  HB_HIP(hipSetDevice(b->device));
  const size_t nr = (size_t)n * n_residual, nt = (size_t)n * spec->n_term;
  int rc = ensure_trace(&b->d_sensor_out, &b->sensor_out_cap, nr);
  if (rc != HB_OK) return rc;
  if ((rc = ensure_trace(&b->d_task_out, &b->task_out_cap, nt + n)) != HB_OK) return rc;
  HB_HIP(hipMemcpyAsync(b->d_sensor_out, residual, nr * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
  HB_HIP(launch_cost_terms(b->d_sensor_out, n, n_residual, K, terms ? b->d_task_out + n : nullptr, b->d_task_out, main_stream(b)));
  HB_HIP(hipMemcpyAsync(cost, b->d_task_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  if (terms) HB_HIP(hipMemcpyAsync(terms, b->d_task_out + n, nt * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_rollout_noise(hb_batch* b, float xfrc_std, float xfrc_rate, unsigned seed) {
  if (!b || !(xfrc_std >= 0.f) || !(xfrc_rate >= 0.f)) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  if (xfrc_std > 0.f && !b->d_xfrc) {
    const size_t nx = (size_t)b->n_env * b->D.dm.nbody * 6;
    if (hipMalloc((void**)&b->d_xfrc, nx * sizeof(float)) != hipSuccess) return HB_ENOMEM;
    HB_HIP(hipMemset(b->d_xfrc, 0, nx * sizeof(float)));
  }
  b->xfrc_std = xfrc_std; b->xfrc_rate = xfrc_rate; b->xfrc_seed = seed; b->xfrc_calls = 0;
  return HB_OK;
}

int hb_rollout_trajectory(hb_batch* b, const float* ctrl, int T, float* qpos_out, float* qvel_out, int* failed) {
  if (!b || T < 1 || (!ctrl && b->D.dm.nu > 0)) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  const size_t n = (size_t)T * b->n_env * b->D.dm.nu;
  int rc = ensure_ctrl(b, std::max<size_t>(n, 1));
  if (rc != HB_OK) return rc;
  if (n) HB_HIP(hipMemcpyAsync(ctrl_for_write(b), ctrl, n * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
  const size_t nq_out = (size_t)T * b->n_env * b->D.dm.nq, nv_out = (size_t)T * b->n_env * b->D.dm.nv;
  if (qpos_out && (rc = ensure_trace(&b->d_qpos_out, &b->qpos_out_cap, nq_out)) != HB_OK) return rc;
  if (qvel_out && (rc = ensure_trace(&b->d_qvel_out, &b->qvel_out_cap, nv_out)) != HB_OK) return rc;
  BatchPtrs P = make_ptrs(b);
  P.ctrl = b->d_ctrl; P.ctrl_mode = 1;
  P.qpos_out = qpos_out ? b->d_qpos_out : nullptr;
  P.qvel_out = qvel_out ? b->d_qvel_out : nullptr;
  rc = launch_steps(b, P, T);
  if (rc != HB_OK) return rc;
  if (qpos_out) HB_HIP(hipMemcpyAsync(qpos_out, b->d_qpos_out, nq_out * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  if (qvel_out) HB_HIP(hipMemcpyAsync(qvel_out, b->d_qvel_out, nv_out * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  if (failed) {
    // CheckWarnings (mujoco_mpc/mjpc/utilities.cc:787-799): a trajectory that raised a bad-state warning is a failure
    std::vector<int> st(b->n_env);
    rc = hb_get_status(b, st.data());
    if (rc != HB_OK) return rc;
    for (int e = 0; e < b->n_env; e++) failed[e] = (st[e] & (HB_WARN_BADQPOS | HB_WARN_BADQVEL | HB_WARN_BADQACC)) ? 1 : 0;
  }
  return HB_OK;
}

// ---- agent.proto State <-> one env's state record (protobuf wire format, no protobuf dependency) ----
namespace {
size_t pb_varint(unsigned long long v, unsigned char* out) {
  size_t n = 0;
  do { unsigned char c = v & 0x7f; v >>= 7; if (v) c |= 0x80; if (out) out[n] = c; n++; } while (v);
  return n;
}
// appends `tag`, and for packed doubles the byte length, then the little-endian doubles; counts when out == nullptr
size_t pb_doubles(int field, const double* v, int n, bool packed, unsigned char* out) {
  size_t k = 0;
  k += pb_varint(((unsigned long long)field << 3) | (packed ? 2 : 1), out ? out + k : nullptr);
  if (packed) k += pb_varint((unsigned long long)n * 8, out ? out + k : nullptr);
  if (out) memcpy(out + k, v, (size_t)n * 8);  // hosts of this engine are little-endian
  return k + (size_t)n * 8;
}
bool pb_read_varint(const unsigned char* buf, int len, int& pos, unsigned long long& v) {
  v = 0;
  for (int shift = 0; pos < len && shift < 64; shift += 7) {
    const unsigned char c = buf[pos++];
    v |= (unsigned long long)(c & 0x7f) << shift;
    if (!(c & 0x80)) return true;
  }
  return false;
}
}  // namespace

int hb_state_to_proto(hb_batch* b, int env, unsigned char* buf, int cap) {
  if (!b || env < 0 || env >= b->n_env || cap < 0) return HB_EINVAL;
  const Model& m = b->model->m;
  const int ns = b->D.dm.nstate;
  std::vector<float> rec(ns);
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  HB_HIP(hipMemcpy(rec.data(), b->d_state + (size_t)env * ns, (size_t)ns * sizeof(float), hipMemcpyDeviceToHost));
  std::vector<double> d(rec.begin(), rec.end());
  const size_t need = pb_doubles(1, &d[0], 1, false, nullptr) + pb_doubles(2, &d[1], m.nq, true, nullptr) + pb_doubles(3, &d[1 + m.nq], m.nv, true, nullptr);
  if (buf && (size_t)cap >= need) {
    size_t k = pb_doubles(1, &d[0], 1, false, buf);
    k += pb_doubles(2, &d[1], m.nq, true, buf + k);
    k += pb_doubles(3, &d[1 + m.nq], m.nv, true, buf + k);
  }
  return (int)need;
}

int hb_state_from_proto(hb_batch* b, int env, const unsigned char* buf, int len) {
  if (!b || env < 0 || env >= b->n_env || !buf || len < 0) return HB_EINVAL;
  const Model& m = b->model->m;
  const int ns = b->D.dm.nstate;
  std::vector<float> rec(ns);
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  HB_HIP(hipMemcpy(rec.data(), b->d_state + (size_t)env * ns, (size_t)ns * sizeof(float), hipMemcpyDeviceToHost));
  int pos = 0, nqpos = 0, nqvel = 0;  // repeated doubles may arrive packed or one by one; both are appended in order
  bool touched = false;
  while (pos < len) {
    unsigned long long key, l;
    if (!pb_read_varint(buf, len, pos, key)) return HB_EINVAL;
    const int field = (int)(key >> 3), wt = (int)(key & 7);
    if (wt == 1) {  // one double
      if (pos + 8 > len) return HB_EINVAL;
      double v; memcpy(&v, buf + pos, 8); pos += 8;
      if (field == 1) rec[0] = (float)v;
      else if (field == 2) { if (nqpos >= m.nq) return HB_EINVAL; rec[1 + nqpos++] = (float)v; }
      else if (field == 3) { if (nqvel >= m.nv) return HB_EINVAL; rec[1 + m.nq + nqvel++] = (float)v; }
      else if (field >= 4 && field <= 7) return HB_EUNSUPPORTED;
      touched = true;
    } else if (wt == 2) {  // packed doubles (or an unknown length-delimited field)
      if (!pb_read_varint(buf, len, pos, l) || l > (unsigned long long)(len - pos)) return HB_EINVAL;
      if (field >= 4 && field <= 7) { if (l) return HB_EUNSUPPORTED; }
      else if (field == 2 || field == 3) {
        if (l % 8) return HB_EINVAL;
        for (unsigned long long k = 0; k < l / 8; k++) {
          double v; memcpy(&v, buf + pos + 8 * k, 8);
          if (field == 2) { if (nqpos >= m.nq) return HB_EINVAL; rec[1 + nqpos++] = (float)v; }
          else { if (nqvel >= m.nv) return HB_EINVAL; rec[1 + m.nq + nqvel++] = (float)v; }
        }
        touched = true;
      }
      pos += (int)l;
    } else if (wt == 0) { if (!pb_read_varint(buf, len, pos, l)) return HB_EINVAL; }
    else if (wt == 5) { if (pos + 4 > len) return HB_EINVAL; pos += 4; }
    else return HB_EINVAL;
  }
  if ((nqpos && nqpos != m.nq) || (nqvel && nqvel != m.nv)) return HB_EINVAL;  // a partial vector is an error, an absent one is not
  if (touched) for (int i = 0; i < m.nv; i++) rec[1 + m.nq + m.nv + i] = 0.f;
  HB_HIP(hipMemcpy(b->d_state + (size_t)env * ns, rec.data(), (size_t)ns * sizeof(float), hipMemcpyHostToDevice));
  return HB_OK;
}

int hb_sensor_size(const hb_sensor_spec* spec) {
  if (!spec || spec->n_framepos < 0 || spec->n_framepos > HB_MAX_FRAMEPOS) return HB_EINVAL;
  if (spec->n_frameaxis < 0 || spec->n_frameaxis > 8 || spec->n_framelinvel < 0 || spec->n_framelinvel > 8 || spec->n_subtreelinvel < 0 || spec->n_subtreelinvel > 4) return HB_EINVAL;
  return 3 * spec->n_framepos + (spec->subtree_body >= 0 ? 6 : 0) + 3 * (spec->n_frameaxis + spec->n_framelinvel + spec->n_subtreelinvel);
}

// fills the sensor fields of P and sizes the device read-out buffer for T steps
static int sensor_setup(hb_batch* b, const hb_sensor_spec* spec, int T, BatchPtrs& P) {
  const Model& m = b->model->m;
  const int ns = hb_sensor_size(spec);
  if (ns <= 0) return HB_EINVAL;
  for (int k = 0; k < spec->n_framepos; k++) if (spec->framepos_body[k] < 0 || spec->framepos_body[k] >= m.nbody) return HB_EINVAL;
  int tree = -1;
  if (spec->subtree_body >= 0) {
    if (spec->subtree_body < 1 || spec->subtree_body >= m.nbody || m.body_parentid[spec->subtree_body] != 0) return HB_EINVAL;  // a tree root
    for (int bd = 1, t = 0; bd <= spec->subtree_body; bd++)
      if (m.body_parentid[bd] == 0) { if (bd == spec->subtree_body) tree = t; t++; }
    if (tree < 0) return HB_EINVAL;
  }
  const size_t need = (size_t)T * b->n_env * ns;
  if (need > b->sensor_out_cap) {
    if (b->d_sensor_out) HB_IGN(hipFree(b->d_sensor_out));
    b->d_sensor_out = nullptr; b->sensor_out_cap = 0;
    if (hipMalloc((void**)&b->d_sensor_out, need * sizeof(float)) != hipSuccess) return HB_ENOMEM;
    b->sensor_out_cap = need;
  }
  P.sensor_out = b->d_sensor_out; P.sensor_stride = ns; P.sensor_nframe = spec->n_framepos; P.sensor_tree = tree;
  for (int k = 0; k < spec->n_framepos; k++) {
    P.sensor_body[k] = spec->framepos_body[k];
    for (int i = 0; i < 3; i++) P.sensor_off[k][i] = spec->framepos_offset[k][i];
  }
  P.sensor_naxis = spec->n_frameaxis; P.sensor_nlinvel = spec->n_framelinvel; P.sensor_nsub = spec->n_subtreelinvel;
  for (int k = 0; k < spec->n_frameaxis; k++) {
    if (spec->frameaxis_body[k] < 0 || spec->frameaxis_body[k] >= m.nbody || (spec->frameaxis_which[k] != 0 && spec->frameaxis_which[k] != 2)) return HB_EINVAL;
    P.sensor_axis_body[k] = spec->frameaxis_body[k]; P.sensor_axis_which[k] = spec->frameaxis_which[k];
  }
  for (int k = 0; k < spec->n_framelinvel; k++) {
    if (spec->framelinvel_body[k] < 1 || spec->framelinvel_body[k] >= m.nbody) return HB_EINVAL;
    P.sensor_linvel_body[k] = spec->framelinvel_body[k];
  }
  for (int k = 0; k < spec->n_subtreelinvel; k++) {
    const int root = spec->subtreelinvel_body[k];
    if (root < 1 || root >= m.nbody) return HB_EINVAL;
    unsigned long long mask = 0;
    double mass = 0;
    for (int bd = 1; bd < m.nbody; bd++)
      for (int a = bd; a > 0; a = m.body_parentid[a])
        if (a == root) { mask |= 1ull << bd; mass += m.body_mass[bd]; break; }
    P.sensor_submask[k] = mask;
    P.sensor_subinv[k] = mass > 1e-15 ? (float)(1.0 / mass) : 0.f;
  }
  return HB_OK;
}

int hb_rollout_sensors(hb_batch* b, const float* ctrl, int T, const hb_sensor_spec* spec, float* sensor_out, float* qpos_out) {
  if (!b || T < 1 || !spec || !sensor_out || (!ctrl && b->D.dm.nu > 0)) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  const size_t n = (size_t)T * b->n_env * b->D.dm.nu;
  int rc = ensure_ctrl(b, std::max<size_t>(n, 1));
  if (rc != HB_OK) return rc;
  if (n) HB_HIP(hipMemcpyAsync(ctrl_for_write(b), ctrl, n * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
  const size_t nq_out = (size_t)T * b->n_env * b->D.dm.nq;
  if (qpos_out && nq_out > b->qpos_out_cap) {
    if (b->d_qpos_out) HB_IGN(hipFree(b->d_qpos_out));
    b->d_qpos_out = nullptr; b->qpos_out_cap = 0;
    if (hipMalloc((void**)&b->d_qpos_out, nq_out * sizeof(float)) != hipSuccess) return HB_ENOMEM;
    b->qpos_out_cap = nq_out;
  }
  BatchPtrs P = make_ptrs(b);
  P.ctrl = b->d_ctrl; P.ctrl_mode = 1; P.qpos_out = qpos_out ? b->d_qpos_out : nullptr;
  rc = sensor_setup(b, spec, T, P);
  if (rc != HB_OK) return rc;
  rc = launch_steps(b, P, T);
  if (rc != HB_OK) return rc;
  HB_HIP(hipMemcpyAsync(sensor_out, b->d_sensor_out, (size_t)T * b->n_env * P.sensor_stride * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  if (qpos_out) HB_HIP(hipMemcpyAsync(qpos_out, b->d_qpos_out, nq_out * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

// Trajectory::Rollout's data flow for a task cost on the device: `horizon - 1` steps from the current state with the
// read-out row of every step (sensors of `spec`, then the state / control parts in `flags`), then one mj_forward with the
// last action repeated (zero when horizon = 1) for the terminal row.  Leaves `horizon` rows of `*stride` floats in
// b->d_sensor_out and room for horizon + 1 floats per env in b->d_task_out; the status words are cleared first.
static int rollout_rows(hb_batch* b, const float* ctrl, int H, const hb_sensor_spec* spec, int flags, int* stride) {
  const DevModel& dm = b->D.dm;
  const int N = b->n_env, nu = dm.nu;
  const size_t n = (size_t)(H - 1) * N * nu;
  int rc = HB_OK;
  if (ctrl == HB_CTRL_TAPE) {
    if (H < 2 || b->tape_steps < H - 1) return HB_EINVAL;  // the tape hb_ctrl_tape_splines left is shorter than this rollout
  } else {
    rc = ensure_ctrl(b, std::max<size_t>(std::max<size_t>(n, (size_t)N * nu), 1));
    if (rc != HB_OK) return rc;
    b->tape_steps = 0;
    if (n) HB_HIP(hipMemcpyAsync(ctrl_for_write(b), ctrl, n * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
    else if (nu) HB_HIP(hipMemsetAsync(ctrl_for_write(b), 0, (size_t)N * nu * sizeof(float), main_stream(b)));
  }
  // failure is a property of THIS rollout (CheckWarnings looks at the warnings of the rollout's own mjData)
  HB_HIP(hipMemsetAsync(b->d_status, 0, (size_t)N * sizeof(int), main_stream(b)));
  BatchPtrs P = make_ptrs(b);
  rc = sensor_setup(b, spec, H, P);
  if (rc != HB_OK) return rc;
  *stride = P.sensor_stride + ((flags & 4) ? dm.nq : 0) + ((flags & 1) ? dm.nv : 0) + ((flags & 2) ? nu : 0);
  if ((rc = ensure_trace(&b->d_sensor_out, &b->sensor_out_cap, (size_t)H * N * *stride)) != HB_OK) return rc;
  if ((rc = ensure_trace(&b->d_task_out, &b->task_out_cap, (size_t)(H + 1) * N)) != HB_OK) return rc;
  P.sensor_out = b->d_sensor_out; P.sensor_stride = *stride; P.sensor_flags = flags;
  if (H > 1) {
    P.ctrl = b->d_ctrl; P.ctrl_mode = 1;
    rc = launch_steps(b, P, H - 1);
    if (rc != HB_OK) return rc;
  }
  // final mj_forward with the last action repeated (trajectory.cc:188-202)
  BatchPtrs F = P;
  F.ctrl = b->d_ctrl + (H > 1 ? (size_t)(H - 2) * N * nu : 0); F.ctrl_mode = 0; F.integrate = 0;
  F.sensor_out = b->d_sensor_out + (size_t)(H - 1) * N * *stride;
  F.blk0 = 0; F.nblk = N;
  HB_HIP(launch_step(b->D.d_dm, dm.variant, dm.solver, dm.nv, dm.lds_floats, F, 1, main_stream(b))); b->last_kernel = last_step_kernel();
  return HB_OK;
}

static int task_results(hb_batch* b, int H, float* total_return, float* costs) {
  const int N = b->n_env;
  HB_HIP(hipMemcpyAsync(total_return, b->d_task_out, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  if (costs) HB_HIP(hipMemcpyAsync(costs, b->d_task_out + N, (size_t)H * N * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_task_walk_default(const hb_model* h, hb_task_walk* t) {
  if (!h || !t) return HB_EINVAL;
  memset(t, 0, sizeof *t);
  const char* names[5] = {"torso", "pelvis", "foot_right", "foot_left", "waist_lower"};
  int id[5];
  for (int k = 0; k < 5; k++) if ((id[k] = hb_model_name2id(h, "body", names[k])) < 0) return HB_EINVAL;
  t->torso_body = id[0]; t->pelvis_body = id[1]; t->foot_right_body = id[2]; t->foot_left_body = id[3]; t->waist_lower_body = id[4];
  t->height_goal = 1.35f; t->speed_goal = 0.5f;
  hb_sizes sz;
  hb_model_sizes(h, &sz);
  // user sensors of tasks/humanoid/walk/task.xml:28-35: name, dim, "norm weight lo hi [p [q]]"
  const int dim[8] = {1, 1, 2, 8, sz.nq - 7, 2, 1, sz.nu}, norm[8] = {7, 8, 1, 2, 0, 7, 7, 3};
  const float w[8] = {5.f, 1.f, 5.f, 5.f, 0.025f, 0.625f, 1.f, 0.1f};
  const float p[8] = {0.1f, 0.05f, 0.02f, 0.01f, 0.f, 0.2f, 0.5f, 0.3f}, q[8] = {4.f, 0.f, 4.f, 0.f, 0.f, 4.f, 3.f, 0.f};
  t->n_term = 8;
  for (int k = 0; k < 8; k++) { t->dim[k] = dim[k]; t->norm[k] = norm[k]; t->weight[k] = w[k]; t->norm_p[k][0] = p[k]; t->norm_p[k][1] = q[k]; }
  return HB_OK;
}

int hb_rollout_task_walk(hb_batch* b, const float* ctrl, int H, const hb_task_walk* task, float* total_return, float* costs) {
  if (!b || !task || !total_return || H < 1 || (H > 1 && !ctrl && b->D.dm.nu > 0)) return HB_EINVAL;
  const Model& m = b->model->m;
  const DevModel& dm = b->D.dm;
  const int bodies[5] = {task->torso_body, task->pelvis_body, task->foot_right_body, task->foot_left_body, task->waist_lower_body};
  for (int bd : bodies) if (bd < 1 || bd >= m.nbody) return HB_EINVAL;
  if (dm.nq < 7 || task->n_term < 1 || task->n_term > 8) return HB_EINVAL;
  const int nres = 1 + 1 + 2 + 8 + (dm.nq - 7) + 1 + 2 + dm.nu;
  int total_dim = 0;
  for (int k = 0; k < task->n_term; k++) {
    if (task->dim[k] < 1 || task->norm[k] < -1 || task->norm[k] > 8 || task->norm[k] == 4) return HB_EINVAL;
    total_dim += task->dim[k];
  }
  if (total_dim != nres || nres > 96) return HB_EINVAL;  // "mismatch between total user-sensor dimension and actual length of residual" (walk.cc:150-162)
  HB_HIP(hipSetDevice(b->device));
  // read-out rows: framepos (objtype body: inertial frames) torso, foot_right, foot_left, pelvis | subtreecom, subtreelinvel (torso's tree)
  // | up axes x4, forward axes x4 | framelinvel torso, foot_right, foot_left | subtreelinvel waist_lower | qpos | ctrl
  hb_sensor_spec spec;
  memset(&spec, 0, sizeof spec);
  const int fp[4] = {task->torso_body, task->foot_right_body, task->foot_left_body, task->pelvis_body};
  spec.n_framepos = 4;
  for (int k = 0; k < 4; k++) {
    spec.framepos_body[k] = fp[k];
    for (int i = 0; i < 3; i++) spec.framepos_offset[k][i] = (float)m.body_ipos[3 * fp[k] + i];
  }
  int root = task->torso_body;
  while (m.body_parentid[root] != 0) root = m.body_parentid[root];
  if (root != task->torso_body) return HB_EINVAL;  // torso_subcom / torso_subcomvel are read as a whole tree
  spec.subtree_body = root;
  const int axb[4] = {task->torso_body, task->pelvis_body, task->foot_right_body, task->foot_left_body};
  spec.n_frameaxis = 8;
  for (int k = 0; k < 4; k++) { spec.frameaxis_body[k] = axb[k]; spec.frameaxis_which[k] = 2; spec.frameaxis_body[4 + k] = axb[k]; spec.frameaxis_which[4 + k] = 0; }
  spec.n_framelinvel = 3;
  spec.framelinvel_body[0] = task->torso_body; spec.framelinvel_body[1] = task->foot_right_body; spec.framelinvel_body[2] = task->foot_left_body;
  spec.n_subtreelinvel = 1;
  spec.subtreelinvel_body[0] = task->waist_lower_body;
  int stride = 0;
  int rc = rollout_rows(b, ctrl, H, &spec, /*qpos | ctrl*/ 4 | 2, &stride);
  if (rc != HB_OK) return rc;
  WalkTask K;
  memset(&K, 0, sizeof K);
  K.o_torso = 0; K.o_foot_r = 3; K.o_foot_l = 6; K.o_pelvis = 9; K.o_com = 12; K.o_vel = 15; K.o_axes = 18; K.o_linvel = K.o_axes + 24; K.o_sub = K.o_linvel + 9;
  K.o_qpos = K.o_sub + 3; K.o_ctrl = K.o_qpos + dm.nq; K.nq = dm.nq; K.nu = dm.nu; K.stride = stride;
  if (K.o_ctrl + dm.nu != stride) return HB_EINVAL;
  K.height_goal = task->height_goal; K.speed_goal = task->speed_goal; K.risk = task->risk; K.nterm = task->n_term;
  for (int k = 0; k < task->n_term; k++) { K.dim[k] = task->dim[k]; K.norm[k] = task->norm[k]; K.weight[k] = task->weight[k]; K.p[k] = task->norm_p[k][0]; K.q[k] = task->norm_p[k][1]; }
  HB_HIP(launch_walk_cost(b->d_sensor_out, H, b->n_env, K, b->d_status, b->d_task_out, costs ? b->d_task_out + b->n_env : nullptr, main_stream(b)));
  return task_results(b, H, total_return, costs);
}

int hb_task_stand_default(const hb_model* h, hb_task_stand* t) {
  if (!h || !t) return HB_EINVAL;
  memset(t, 0, sizeof *t);
  const int head = hb_model_name2id(h, "body", "head"), fl = hb_model_name2id(h, "body", "foot_left"), fr = hb_model_name2id(h, "body", "foot_right"),
            torso = hb_model_name2id(h, "body", "torso");
  if (head < 0 || fl < 0 || fr < 0 || torso < 0) return HB_EINVAL;
  t->head_body = head; t->subtree_body = torso; t->n_feet = 4;
  const int fb[4] = {fl, fl, fr, fr};
  const float fx[4] = {-0.07f, 0.14f, -0.07f, 0.14f};
  for (int k = 0; k < 4; k++) { t->foot_body[k] = fb[k]; t->foot_offset[k][0] = fx[k]; }
  t->height_goal = 1.4f;
  const int norm[5] = {6, 6, 0, 0, 3};
  const float w[5] = {100.f, 50.f, 10.f, 0.01f, 0.025f}, p[5] = {0.1f, 0.1f, 0.f, 0.f, 0.3f};
  for (int k = 0; k < 5; k++) { t->norm[k] = norm[k]; t->weight[k] = w[k]; t->norm_p[k][0] = p[k]; }
  return HB_OK;
}

int hb_rollout_task_stand(hb_batch* b, const float* ctrl, int H, const hb_task_stand* task, float* total_return, float* costs) {
  if (!b || !task || !total_return || H < 1 || (H > 1 && !ctrl && b->D.dm.nu > 0)) return HB_EINVAL;
  const Model& m = b->model->m;
  const DevModel& dm = b->D.dm;
  if (task->n_feet < 1 || task->n_feet > 4 || task->head_body < 0 || task->head_body >= m.nbody || dm.nv < 6) return HB_EINVAL;
  for (int k = 0; k < task->n_feet; k++) if (task->foot_body[k] < 0 || task->foot_body[k] >= m.nbody) return HB_EINVAL;
  for (int k = 0; k < 5; k++) if (task->norm[k] < -1 || task->norm[k] > 8 || task->norm[k] == 4) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  // read-out rows: [head | feet | subtreecom | subtreelinvel | qvel | ctrl]
  hb_sensor_spec spec;
  memset(&spec, 0, sizeof spec);
  spec.n_framepos = 1 + task->n_feet;
  spec.framepos_body[0] = task->head_body;
  // "head_position" is a framepos with objtype="body": MuJoCo's body objtype is the INERTIAL frame (xipos = xpos + R ipos)
  for (int i = 0; i < 3; i++) spec.framepos_offset[0][i] = (float)m.body_ipos[3 * task->head_body + i];
  for (int k = 0; k < task->n_feet; k++) {
    spec.framepos_body[1 + k] = task->foot_body[k];
    for (int i = 0; i < 3; i++) spec.framepos_offset[1 + k][i] = task->foot_offset[k][i];
  }
  spec.subtree_body = task->subtree_body;
  if (spec.subtree_body < 0) return HB_EINVAL;
  const int nu = dm.nu, nv = dm.nv;
  int stride = 0;
  int rc = rollout_rows(b, ctrl, H, &spec, /*qvel | ctrl*/ 1 | 2, &stride);
  if (rc != HB_OK) return rc;
  StandTask K;
  memset(&K, 0, sizeof K);
  K.n_feet = task->n_feet; K.o_head = 0; K.o_feet = 3; K.o_com = 3 * spec.n_framepos; K.o_vel = K.o_com + 3; K.o_qvel = K.o_com + 6; K.o_ctrl = K.o_qvel + nv;
  K.nv = nv; K.nu = nu; K.stride = stride;
  if (K.o_ctrl + nu != stride) return HB_EINVAL;
  K.height_goal = task->height_goal; K.risk = task->risk;
  for (int k = 0; k < 5; k++) { K.norm[k] = task->norm[k]; K.weight[k] = task->weight[k]; K.p[k] = task->norm_p[k][0]; K.q[k] = task->norm_p[k][1]; }
  HB_HIP(launch_stand_cost(b->d_sensor_out, H, b->n_env, K, b->d_status, b->d_task_out, costs ? b->d_task_out + b->n_env : nullptr, main_stream(b)));
  return task_results(b, H, total_return, costs);
}

int hb_sensors(hb_batch* b, const float* ctrl, const hb_sensor_spec* spec, float* sensor_out) {
  if (!b || !spec || !sensor_out) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  const size_t n = (size_t)b->n_env * b->D.dm.nu;
  if (ctrl && n) HB_HIP(hipMemcpyAsync(ctrl_for_write(b), ctrl, n * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
  else if (n) HB_HIP(hipMemsetAsync(ctrl_for_write(b), 0, n * sizeof(float), main_stream(b)));
  BatchPtrs P = make_ptrs(b);
  P.ctrl = b->d_ctrl; P.ctrl_mode = 0; P.integrate = 0;
  int rc = sensor_setup(b, spec, 1, P);
  if (rc != HB_OK) return rc;
  HB_HIP(launch_step(b->D.d_dm, b->D.dm.variant, b->D.dm.solver, b->D.dm.nv, b->D.dm.lds_floats, P, 1, main_stream(b))); b->last_kernel = last_step_kernel();
  HB_HIP(hipMemcpyAsync(sensor_out, b->d_sensor_out, (size_t)b->n_env * P.sensor_stride * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_set_state_broadcast(hb_batch* b, unsigned spec, const float* state) { return set_state_broadcast_impl<float>(b, spec, state); }
int hb_set_state_broadcast_f64(hb_batch* b, unsigned spec, const double* state) { return set_state_broadcast_impl<double>(b, spec, state); }

int hb_rollout_halton(hb_batch* b, int T, int t0, int env_offset, float* qpos_out_dev) {
  if (!b || T < 1) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  BatchPtrs P = make_ptrs(b);
  P.ctrl = nullptr; P.ctrl_mode = 2; P.t0 = t0; P.env_offset = env_offset; P.qpos_out = qpos_out_dev;
  return launch_steps(b, P, T);
}

int hb_state_size(const hb_batch* b, unsigned spec) {
  if (!b || (spec & ~kSupportedSpec)) return HB_EINVAL;
  return spec_size(b->model->m, spec);
}
int hb_get_state(hb_batch* b, unsigned spec, float* out) { return get_state_impl<float>(b, spec, out); }
int hb_set_state(hb_batch* b, unsigned spec, const float* in) { return set_state_impl<float>(b, spec, in); }
int hb_get_state_f64(hb_batch* b, unsigned spec, double* out) { return get_state_impl<double>(b, spec, out); }
int hb_set_state_f64(hb_batch* b, unsigned spec, const double* in) { return set_state_impl<double>(b, spec, in); }

static int env_alloc(hb_batch* b) {
  if (b->env_ready) return HB_OK;
  const DevModel& dm = b->D.dm;
  size_t n = b->n_env;
  HB_HIP(hipSetDevice(b->device));
  if (!b->d_obs) {
    // one record [obs n x nobs | reward n | terminated n | truncated n]: a host that keeps its four buffers in the same order and
    // back to back (engine.py does) gets them in one transfer instead of four
    if (hipMalloc((void**)&b->d_obs, n * dm.nobs * sizeof(float) + n * sizeof(float) + 2 * n) != hipSuccess) return HB_ENOMEM;
    b->d_reward = b->d_obs + n * dm.nobs;
    b->d_term = reinterpret_cast<uint8_t*>(b->d_reward + n);
    b->d_trunc = b->d_term + n;
  }
  size_t nu = std::max(1, dm.nu);
  if (hipMalloc((void**)&b->d_prev, n * nu * sizeof(float)) != hipSuccess || hipMalloc((void**)&b->d_latest, n * nu * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&b->d_action, n * nu * sizeof(float)) != hipSuccess || hipMalloc((void**)&b->d_qfrc, n * dm.nv * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&b->d_episode, n * sizeof(int)) != hipSuccess || hipMalloc((void**)&b->d_seen, n * sizeof(int)) != hipSuccess) return HB_ENOMEM;
  HB_HIP(hipMemset(b->d_seen, 0, n * sizeof(int)));
  HB_HIP(hipMemset(b->d_prev, 0, n * nu * sizeof(float)));
  HB_HIP(hipMemset(b->d_latest, 0, n * nu * sizeof(float)));
  HB_HIP(hipMemset(b->d_qfrc, 0, n * dm.nv * sizeof(float)));
  HB_HIP(hipMemset(b->d_episode, 0, n * sizeof(int)));
  hb_env_config def;
  hb_env_default_config(b->model, &def);
  static_assert(sizeof(hb_env_config) == sizeof(EnvConfig), "hb_env_config and EnvConfig must have the same layout");
  memcpy(&b->env_cfg, &def, sizeof def);
  b->env_ready = true;
  return HB_OK;
}

// reward / termination / observation of every env (or those of `mask`).  observe: push the observation through the
// realism layer's noise and delay lines (a step of the episode) instead of returning the true one.
static int env_eval(hb_batch* b, bool allow_reset, bool observe, const uint8_t* mask, float* d_obs, float* d_reward, uint8_t* d_term, uint8_t* d_trunc) {
  EnvConfig cfg = b->env_cfg;
  if (!allow_reset) cfg.auto_reset = 0;
  const Model& m = b->model->m;
  const float* src = b->D.d_qpos_src + (cfg.reset_keyframe < 0 || cfg.reset_keyframe >= m.nkey ? 0 : (size_t)(1 + cfg.reset_keyframe) * m.nq);
  EnvRandState S = b->rs;
  if (!b->rand_on) memset(&S, 0, sizeof S);
  HB_HIP(launch_env(b->D.dm, cfg, b->env_rand, S, b->d_state, b->d_qfrc, b->d_counts, b->d_prev, b->d_latest, src, b->d_episode, b->d_status, d_obs, d_reward,
                    d_term, d_trunc, mask, observe ? 1 : 0, b->dom_rand, b->d_dr, b->dr_stride, b->n_env, b->env_offset, main_stream(b), observe ? b->d_term_obs : nullptr, b->d_seen));
  return HB_OK;
}

int hb_get_obs(hb_batch* b, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated) {
  if (!b || !obs) return HB_EINVAL;
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  int n = b->n_env, nobs = b->D.dm.nobs;
  if (reward || terminated || truncated) {
    rc = env_eval(b, false, false, nullptr, b->d_obs, b->d_reward, b->d_term, b->d_trunc);  // pure evaluation: no reset, no bookkeeping, true observation
    if (rc != HB_OK) return rc;
  } else {
    HB_HIP(launch_obs(b->D.dm, b->d_state, b->d_obs, n, main_stream(b)));
  }
  HB_HIP(hipMemcpyAsync(obs, b->d_obs, (size_t)n * nobs * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  if (reward) HB_HIP(hipMemcpyAsync(reward, b->d_reward, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  if (terminated) HB_HIP(hipMemcpyAsync(terminated, b->d_term, n, hipMemcpyDeviceToHost, main_stream(b)));
  if (truncated) HB_HIP(hipMemcpyAsync(truncated, b->d_trunc, n, hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_env_default_config(const hb_model* h, hb_env_config* c) {
  if (!h || !c) return HB_EINVAL;
  const Model& m = h->m;
  memset(c, 0, sizeof *c);
  double z0 = 1.0;
  for (int j = 0; j < m.njnt; j++) if (m.jnt_type[j] == JNT_FREE) { z0 = m.qpos0[m.jnt_qposadr[j] + 2]; break; }
  c->target_z = (float)(0.94 * z0);  // the reference targets its standing height (Z_INITIAL_POS); 6 % slack for the soft stance
  c->min_z = (float)(0.3 * z0);
  c->max_time = 10.f;                 // MAX_SIM_TIME_STANDUP
  double gear = 0;
  for (int a = 0; a < m.nu; a++) gear += std::fabs(m.actuator_gear[a]);
  c->safe_torque = (float)(m.nu ? 0.05 * gear / m.nu : 1.0);  // reference: 1.0 N m = 5 % of its 20 N m motors
  c->control_frequency = (float)(1.0 / m.timestep);
  c->action_scale = 1.5707963267948966f;
  c->w_hvel = 5.f; c->w_upright = 10.f; c->w_height = 15.f; c->w_torque = 2.5f; c->w_ctrl_change = 2.f; c->w_ctrl_reg = 0.5f; c->w_symmetry = 1.f;
  c->self_collision_penalty = -20.f; c->terminal_reward = -100.f; c->upright_tol = 0.7f;
  // symmetry pairs: actuators named <x>_right / <x>_left (mirrored joint axes in the model => equal controls)
  for (int a = 0; a < m.nu && c->n_equal < HB_ENV_MAX_PAIRS; a++) {
    const std::string& n = m.actuator_name[a];
    const std::string suf = "_right";
    if (n.size() > suf.size() && n.compare(n.size() - suf.size(), suf.size(), suf) == 0) {
      std::string other = n.substr(0, n.size() - suf.size()) + "_left";
      for (int k = 0; k < m.nu; k++) if (m.actuator_name[k] == other) { c->equal_pairs[c->n_equal][0] = k; c->equal_pairs[c->n_equal][1] = a; c->n_equal++; break; }
    }
  }
  c->auto_reset = 1; c->reset_keyframe = -1; c->reset_perturb = 1.f;
  c->reward_kind = 0; c->w_vvel = 0.f;
  c->min_z_grounded = (float)(0.25 * z0);  // the reference's MIN_Z_BEFORE_GROUNDED sits a quarter of the way up its robot
  c->reset_collision_mode = 0;
  return HB_OK;
}

int hb_env_team_config(const hb_model* h, hb_env_config* c) {
  if (!h || !c) return HB_EINVAL;
  const Model& m = h->m;
  int rc = hb_env_default_config(h, c);
  if (rc != HB_OK) return rc;
  auto act = [&](const char* name) { for (int a = 0; a < m.nu; a++) if (m.actuator_name[a] == name) return a; return -1; };
  // reward_functions.py:289-339 (standupReward) and simulation_parameters.py:51-77
  c->target_z = -0.375f;            // TARGET_Z_POS = Z_INITIAL_POS
  c->min_z = -0.6f;                 // MIN_Z_POS_FOR_REWARD
  c->max_time = 10.f;               // MAX_SIM_TIME_STANDUP
  c->safe_torque = 1.0f;            // MAX__SAFE_JOINT_TORQUE
  c->control_frequency = 500.f;     // CONTROL_FREQUENCY
  c->n_equal = c->n_opposite = 0;
  const char* eq[][2] = {{"left_elbow", "right_elbow"}};
  const char* op[][2] = {{"left_hip_roll", "right_hip_roll"}, {"left_hip_pitch", "right_hip_pitch"}, {"left_knee", "right_knee"},
                         {"left_shoulder_pitch", "right_shoulder_pitch"}, {"left_shoulder_roll", "right_shoulder_roll"}};
  for (auto& pr : eq) { const int a = act(pr[0]), b2 = act(pr[1]); if (a < 0 || b2 < 0) return HB_EINVAL; c->equal_pairs[c->n_equal][0] = a; c->equal_pairs[c->n_equal][1] = b2; c->n_equal++; }
  for (auto& pr : op) { const int a = act(pr[0]), b2 = act(pr[1]); if (a < 0 || b2 < 0) return HB_EINVAL; c->opposite_pairs[c->n_opposite][0] = a; c->opposite_pairs[c->n_opposite][1] = b2; c->n_opposite++; }
  c->reset_keyframe = -1;
  for (int k = 0; k < m.nkey; k++) if (m.key_name[k] == "standup_reset") c->reset_keyframe = k;
  c->reset_perturb = 1.f;
  c->reset_quat_perturb = 0.1f;     // QUAT_INITIAL_OFFSET_MAX
  c->obs_actuator_order = 1;        // JOINT_NAMES order = the <motor> order of the reference's humanoid.xml
  c->min_z_grounded = -0.6f;
  c->reset_collision_mode = 1;      // CPUEnv.reset starts over while anything is in contact (cpu_env.py:411-414)
  return HB_OK;
}

int hb_env_configure(hb_batch* b, const hb_env_config* cfg) {
  if (!b || !cfg) return HB_EINVAL;
  if (cfg->n_equal < 0 || cfg->n_equal > HB_ENV_MAX_PAIRS || cfg->n_opposite < 0 || cfg->n_opposite > HB_ENV_MAX_PAIRS || !(cfg->action_scale > 0)) return HB_EINVAL;
  int nu = b->D.dm.nu;
  for (int k = 0; k < cfg->n_equal; k++) for (int t = 0; t < 2; t++) if (cfg->equal_pairs[k][t] < 0 || cfg->equal_pairs[k][t] >= nu) return HB_EINVAL;
  for (int k = 0; k < cfg->n_opposite; k++) for (int t = 0; t < 2; t++) if (cfg->opposite_pairs[k][t] < 0 || cfg->opposite_pairs[k][t] >= nu) return HB_EINVAL;
  if (cfg->reset_keyframe >= b->model->m.nkey) return HB_EINVAL;
  if (cfg->reward_kind < 0 || cfg->reward_kind > 1 || cfg->reset_collision_mode < 0 || cfg->reset_collision_mode > 2) return HB_EINVAL;
  if (!(cfg->reset_quat_perturb >= 0.f) || (cfg->obs_actuator_order != 0 && cfg->obs_actuator_order != 1)) return HB_EINVAL;
  if (cfg->obs_actuator_order && !b->D.has_act_order) return HB_EINVAL;  // needs exactly one actuator per scalar joint
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  memcpy(&b->env_cfg, cfg, sizeof *cfg);
  // the env / observation / policy kernels take the DevModel by value from this host copy: point it at the chosen order
  b->D.dm.obs_jnt = cfg->obs_actuator_order ? b->D.obs_jnt_act : b->D.obs_jnt_joint;
  b->D.dm.obs_src = cfg->obs_actuator_order ? b->D.obs_src_act : b->D.obs_src_joint;
  return HB_OK;
}

int hb_env_default_randomization(const hb_model* h, hb_env_randomization* r) {
  if (!h || !r) return HB_EINVAL;
  memset(r, 0, sizeof *r);
  const float deg = 0.017453292519943295f;
  r->factor = 1.f; r->seed = 0; r->control_timestep = (float)h->m.timestep;
  r->joint_angle_noise = 2.f * deg;     // JOINT_ANGLE_NOISE_STDDEV     (simulation_parameters.py:39-45)
  r->joint_velocity_noise = 5.f * deg;  // JOINT_VELOCITY_NOISE_STDDEV
  r->gyro_noise = 2.f * deg;            // GYRO_NOISE_STDDEV
  r->imu_noise = 5.f * deg;             // IMU_NOISE_STDDEV
  r->action_noise = 0.5f * deg;         // JOINT_ACTION_NOISE_STDDEV
  r->min_delay = 0.01f; r->max_delay = 0.05f;  // MIN_DELAY, MAX_DELAY
  r->frozen_noise = 0;
  r->push_enabled = 1;                  // *_EXTERNAL_FORCE_* (simulation_parameters.py:14-20)
  r->push_min_interval = 1.f; r->push_max_interval = 3.f; r->push_min_duration = 0.05f; r->push_max_duration = 0.15f;
  r->push_min_force = 5.f; r->push_max_force = 15.f;
  return HB_OK;
}

// releases the realism layer's device arrays
static void envrand_free_fwd(hb_batch* b) { void* ptrs[] = {b->rs.k_act, b->rs.k_obs, b->rs.delay, b->rs.fifo_act, b->rs.fifo_joint, b->rs.fifo_gyro, b->rs.fifo_grav, b->rs.push}; for (void* p : ptrs) if (p) HB_IGN(hipFree(p)); memset(&b->rs, 0, sizeof b->rs); b->rand_on = false; }
static void envrand_free(hb_batch* b) { envrand_free_fwd(b); }
int hb_env_randomize(hb_batch* b, const hb_env_randomization* cfg) {
  if (!b) return HB_EINVAL;
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  if (!cfg || !(cfg->factor > 0.f)) { envrand_free(b); return HB_OK; }
  const DevModel& dm = b->D.dm;
  const float dt = cfg->control_timestep > 0.f ? cfg->control_timestep : dm.timestep;
  if (!(cfg->max_delay >= cfg->min_delay) || cfg->min_delay < 0.f || cfg->max_delay * cfg->factor / dt > (float)(kDelaySlots - 1)) return HB_EINVAL;
  if (cfg->push_enabled && (!(cfg->push_max_interval >= cfg->push_min_interval) || !(cfg->push_max_duration >= cfg->push_min_duration) ||
                            !(cfg->push_max_force >= cfg->push_min_force) || dm.nbody < 2)) return HB_EINVAL;
  static_assert(sizeof(hb_env_randomization) == sizeof(EnvRand), "hb_env_randomization and EnvRand must have the same layout");
  const size_t n = b->n_env, nu = std::max(1, dm.nu), nj2 = std::max(2, dm.nobs - 6);
  if (!b->rs.k_act) {
    bool ok = hipMalloc((void**)&b->rs.k_act, n * sizeof(int)) == hipSuccess && hipMalloc((void**)&b->rs.k_obs, n * sizeof(int)) == hipSuccess &&
              hipMalloc((void**)&b->rs.delay, n * 4 * sizeof(int)) == hipSuccess && hipMalloc((void**)&b->rs.fifo_act, n * kDelaySlots * nu * sizeof(float)) == hipSuccess &&
              hipMalloc((void**)&b->rs.fifo_joint, n * kDelaySlots * nj2 * sizeof(float)) == hipSuccess &&
              hipMalloc((void**)&b->rs.fifo_gyro, n * kDelaySlots * 3 * sizeof(float)) == hipSuccess &&
              hipMalloc((void**)&b->rs.fifo_grav, n * kDelaySlots * 3 * sizeof(float)) == hipSuccess && hipMalloc((void**)&b->rs.push, n * 8 * sizeof(float)) == hipSuccess;
    if (!ok) { envrand_free(b); return HB_ENOMEM; }
    HB_HIP(hipMemset(b->rs.push, 0, n * 8 * sizeof(float)));
  }
  if (cfg->push_enabled && !b->d_xfrc) {
    const size_t nx = n * 6 * dm.nbody;
    if (hipMalloc((void**)&b->d_xfrc, nx * sizeof(float)) != hipSuccess) return HB_ENOMEM;
    HB_HIP(hipMemset(b->d_xfrc, 0, nx * sizeof(float)));
  }
  memcpy(&b->env_rand, cfg, sizeof *cfg);
  b->rs.xfrc = cfg->push_enabled ? b->d_xfrc : nullptr;
  b->rand_on = true;
  // a consistent episode state until the caller resets: delays drawn, rings empty
  HB_HIP(launch_envrand_reset(dm, b->env_rand, b->rs, b->d_episode, nullptr, b->n_env, b->env_offset, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_env_default_domain_randomization(const hb_model* h, hb_domain_randomization* d) {
  if (!h || !d) return HB_EINVAL;
  memset(d, 0, sizeof *d);
  d->factor = 1.f; d->seed = 0;
  d->friction_min_mult = 0.5f; d->friction_max_mult = 1.f;   // FLOOR_FRICTION_*_MULTIPLIER (simulation_parameters.py:5-7)
  d->max_mass_change = 0.05f; d->max_external_mass = 0.2f;   // MAX_MASS_CHANGE_PER_LIMB, MAX_EXTERNAL_MASS_ADDED
  d->armature_max_change = 0.0005f; d->stiffness_max_change = 0.f; d->margin_max_change = 0.05f; d->range_max_change = 0.1f;  // JOINT_*_MAX_CHANGE
  d->kp_nominal = 0.f; d->kp_max_change = 0.5f;              // JOINT_P_GAIN(_MAX_CHANGE); nominal 0: keep the model's gains
  d->force_limit_max_change = 0.05f;                         // JOINT_FORCE_LIMIT_MAX_CHANGE
  d->floor_bump_min = 0.f; d->floor_bump_max = h->m.nhfield > 0 ? 0.1f : 0.f;  // MIN/MAX_FLOOR_BUMP_HEIGHT (simulation_parameters.py:47-48); only with a height field
  return HB_OK;
}

int hb_env_domain_randomize(hb_batch* b, const hb_domain_randomization* cfg) {
  if (!b) return HB_EINVAL;
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  if (!cfg || !(cfg->factor > 0.f)) {
    if (b->d_dr) HB_IGN(hipFree(b->d_dr));
    b->d_dr = nullptr; b->dr_stride = 0;
    return HB_OK;
  }
  if (!(cfg->friction_max_mult >= cfg->friction_min_mult) || cfg->friction_min_mult < 0.f || cfg->max_mass_change < 0.f || cfg->max_external_mass < 0.f ||
      cfg->armature_max_change < 0.f || cfg->stiffness_max_change < 0.f || cfg->margin_max_change < 0.f || cfg->range_max_change < 0.f || cfg->kp_max_change < 0.f ||
      cfg->force_limit_max_change < 0.f || cfg->floor_bump_min < 0.f || cfg->floor_bump_max < 0.f) return HB_EINVAL;
  static_assert(sizeof(hb_domain_randomization) == sizeof(DomainRand), "hb_domain_randomization and DomainRand must have the same layout");
  const DevModel& dm = b->D.dm;
  const DomainLayout L = domain_layout(dm.nbody, dm.nv, dm.nlimcand, dm.nu, dm.nhfielddata);
  if (!b->d_dr && hipMalloc((void**)&b->d_dr, (size_t)b->n_env * L.stride * sizeof(float)) != hipSuccess) return HB_ENOMEM;
  b->dr_stride = L.stride;
  memcpy(&b->dom_rand, cfg, sizeof *cfg);
  // valid parameters at once (the draw of episode 0); hb_env_reset draws again for the episode numbers it assigns
  HB_HIP(launch_domain_rand(dm, b->dom_rand, b->d_dr, b->dr_stride, b->d_episode, nullptr, b->n_env, b->env_offset, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_env_get_domain_params(hb_batch* b, float* out) {
  if (!b) return HB_EINVAL;
  if (!b->d_dr) return 0;
  if (out) {
    HB_HIP(hipSetDevice(b->device));
    HB_HIP(hipStreamSynchronize(main_stream(b)));
    HB_HIP(hipMemcpy(out, b->d_dr, (size_t)b->n_env * b->dr_stride * sizeof(float), hipMemcpyDeviceToHost));
  }
  return b->dr_stride;
}

// CPUEnv._apply_action (+ pushes) -> n_substeps x mj_step -> reward / termination / observation, for every env or those of `mask`
static int env_step_impl(hb_batch* b, const float* action_dev, int n_substeps, const uint8_t* mask, bool allow_reset, float* obs_dev, float* reward_dev,
                         uint8_t* terminated_dev, uint8_t* truncated_dev) {
  const int n = b->n_env * b->D.dm.nu;
  if (b->rand_on) {
    HB_HIP(launch_action_env(b->D.dm, b->env_rand, b->rs, action_dev, b->d_prev, b->d_latest, ctrl_for_write(b), b->d_episode, b->d_state, mask, b->n_env, b->env_offset,
                             main_stream(b)));
  } else if (n && action_dev) {
    HB_HIP(launch_action(action_dev, b->d_prev, b->d_latest, ctrl_for_write(b), n, main_stream(b)));
  }
  BatchPtrs P = make_ptrs(b);
  P.ctrl = b->d_ctrl; P.ctrl_mode = 0; P.env_mask = mask;
  int rc = launch_steps(b, P, n_substeps);
  if (rc != HB_OK) return rc;
  return env_eval(b, allow_reset, true, mask, obs_dev, reward_dev, terminated_dev, truncated_dev);
}

int hb_env_reset(hb_batch* b, float* obs) {
  if (!b || !obs) return HB_EINVAL;
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  const EnvConfig& c = b->env_cfg;
  const Model& m = b->model->m;
  const size_t n = b->n_env, nu = std::max(1, b->D.dm.nu);
  HB_HIP(hipMemsetAsync(b->d_prev, 0, n * nu * sizeof(float), main_stream(b)));
  HB_HIP(hipMemsetAsync(b->d_latest, 0, n * nu * sizeof(float), main_stream(b)));
  HB_HIP(hipMemsetAsync(b->d_episode, 0, n * sizeof(int), main_stream(b)));
  if (b->d_ctrl) HB_HIP(hipMemsetAsync(ctrl_for_write(b), 0, n * nu * sizeof(float), main_stream(b)));
  if (c.reset_collision_mode == 0) {
    rc = reset_impl(b, nullptr, c.reset_keyframe, c.reset_perturb, b->env_offset);
    if (rc != HB_OK) return rc;
    if (b->rand_on) HB_HIP(launch_envrand_reset(b->D.dm, b->env_rand, b->rs, b->d_episode, nullptr, b->n_env, b->env_offset, main_stream(b)));
    if (b->d_dr) HB_HIP(launch_domain_rand(b->D.dm, b->dom_rand, b->d_dr, b->dr_stride, b->d_episode, nullptr, b->n_env, b->env_offset, main_stream(b)));
  } else {
    // The reference's protocol (cpu_env.py:374-416): randomise, take one step with the current (zero) controls, and
    // start over with a new draw while that step ends in a collision or in a terminal state.  Pending envs carry
    // a mask; everything (reset, realism layer, physics, evaluation) runs masked, at most eight draws.
    if (!b->d_rmask && (hipMalloc((void**)&b->d_rmask, n) != hipSuccess || hipMalloc((void**)&b->d_pending, sizeof(int)) != hipSuccess)) return HB_ENOMEM;
    HB_HIP(hipMemsetAsync(b->d_rmask, 1, n, main_stream(b)));
    const float* src = b->D.d_qpos_src + (c.reset_keyframe < 0 ? 0 : (size_t)(1 + c.reset_keyframe) * m.nq);
    const float dtc = b->rand_on && b->env_rand.control_timestep > 0.f ? b->env_rand.control_timestep : (float)m.timestep;
    const int substeps = std::max(1, (int)std::lround(dtc / m.timestep));
    for (int attempt = 0; attempt < 8; attempt++) {
      HB_HIP(launch_reset(b->D.dm, b->d_state, b->d_status, b->d_rmask, src, b->d_episode, b->n_env, c.reset_perturb, b->env_offset, main_stream(b), c.reset_quat_perturb));
      if (b->rand_on) HB_HIP(launch_envrand_reset(b->D.dm, b->env_rand, b->rs, b->d_episode, b->d_rmask, b->n_env, b->env_offset, main_stream(b)));
      if (b->d_dr) HB_HIP(launch_domain_rand(b->D.dm, b->dom_rand, b->d_dr, b->dr_stride, b->d_episode, b->d_rmask, b->n_env, b->env_offset, main_stream(b)));
      rc = env_step_impl(b, nullptr, substeps, b->d_rmask, false, b->d_obs, b->d_reward, b->d_term, b->d_trunc);
      if (rc != HB_OK) return rc;
      HB_HIP(hipMemsetAsync(b->d_pending, 0, sizeof(int), main_stream(b)));
      HB_HIP(launch_reset_check(b->d_counts, b->d_term, b->d_trunc, b->d_rmask, b->d_episode, b->d_pending, c.reset_collision_mode, b->n_env, main_stream(b)));
      int pending = 0;
      HB_HIP(hipMemcpyAsync(&pending, b->d_pending, sizeof(int), hipMemcpyDeviceToHost, main_stream(b)));
      HB_HIP(hipStreamSynchronize(main_stream(b)));
      if (pending == 0) break;
    }
  }
  // the observation the reference returns from reset(): one more pass through the noise and delay lines
  rc = env_eval(b, false, true, nullptr, b->d_obs, b->d_reward, b->d_term, b->d_trunc);
  if (rc != HB_OK) return rc;
  HB_HIP(hipMemcpyAsync(obs, b->d_obs, n * b->D.dm.nobs * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_env_step_dev(hb_batch* b, const float* action_dev, int n_substeps, float* obs_dev, float* reward_dev, uint8_t* terminated_dev, uint8_t* truncated_dev) {
  if (!b || !action_dev || n_substeps < 1 || !obs_dev || !reward_dev || !terminated_dev || !truncated_dev) return HB_EINVAL;
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  return env_step_impl(b, action_dev, n_substeps, nullptr, true, obs_dev, reward_dev, terminated_dev, truncated_dev);
}

static int env_step_host(hb_batch* b, const float* action, int n_substeps, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated, bool wait) {
  if (!b || !action || !obs || !reward || !terminated || !truncated) return HB_EINVAL;
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  int n = b->n_env, nu = b->D.dm.nu, nobs = b->D.dm.nobs;
  if (nu) HB_HIP(hipMemcpyAsync(b->d_action, action, (size_t)n * nu * sizeof(float), hipMemcpyHostToDevice, main_stream(b)));
  rc = hb_env_step_dev(b, b->d_action, n_substeps, b->d_obs, b->d_reward, b->d_term, b->d_trunc);
  if (rc != HB_OK) return rc;
  const size_t ob = (size_t)n * nobs * sizeof(float), rb = (size_t)n * sizeof(float);
  if (reinterpret_cast<const uint8_t*>(reward) == reinterpret_cast<const uint8_t*>(obs) + ob && terminated == reinterpret_cast<const uint8_t*>(reward) + rb && truncated == terminated + n) {
    HB_HIP(hipMemcpyAsync(obs, b->d_obs, ob + rb + 2 * (size_t)n, hipMemcpyDeviceToHost, main_stream(b)));  // the caller's buffers are one record too
  } else {
    HB_HIP(hipMemcpyAsync(obs, b->d_obs, ob, hipMemcpyDeviceToHost, main_stream(b)));
    HB_HIP(hipMemcpyAsync(reward, b->d_reward, rb, hipMemcpyDeviceToHost, main_stream(b)));
    HB_HIP(hipMemcpyAsync(terminated, b->d_term, n, hipMemcpyDeviceToHost, main_stream(b)));
    HB_HIP(hipMemcpyAsync(truncated, b->d_trunc, n, hipMemcpyDeviceToHost, main_stream(b)));
  }
  if (wait) HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}
int hb_env_terminal_obs(hb_batch* b, float* terminal_obs) {
  if (!b) return HB_EINVAL;
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  HB_HIP(hipSetDevice(b->device));
  const size_t bytes = (size_t)b->n_env * b->D.dm.nobs * sizeof(float);
  if (!b->d_term_obs) {  // first call: from the next hb_env_step on the env kernel records them
    if (hipMalloc((void**)&b->d_term_obs, bytes) != hipSuccess) return HB_ENOMEM;
    HB_HIP(hipMemsetAsync(b->d_term_obs, 0, bytes, main_stream(b)));
  }
  if (terminal_obs) {
    HB_HIP(hipMemcpyAsync(terminal_obs, b->d_term_obs, bytes, hipMemcpyDeviceToHost, main_stream(b)));
    HB_HIP(hipStreamSynchronize(main_stream(b)));
  }
  return HB_OK;
}
int hb_env_step(hb_batch* b, const float* action, int n_substeps, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated) {
  return env_step_host(b, action, n_substeps, obs, reward, terminated, truncated, true);
}
int hb_env_step_async(hb_batch* b, const float* action, int n_substeps, float* obs, float* reward, uint8_t* terminated, uint8_t* truncated) {
  return env_step_host(b, action, n_substeps, obs, reward, terminated, truncated, false);
}

int hb_policy_set_mlp(hb_batch* b, int n_layers, const int* sizes, const float* const* weights, const float* const* biases) {
  if (!b || !sizes || !weights || !biases || n_layers < 1 || n_layers > 4) return HB_EINVAL;
  if (sizes[0] != b->D.dm.nobs || sizes[n_layers] != b->D.dm.nu) return HB_EINVAL;
  for (int l = 0; l <= n_layers; l++) if (sizes[l] < 1 || sizes[l] > 512) return HB_EINVAL;
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  int maxh = 1;
  bool fused = true;  // one-launch policy kernel: every width fits its LDS tiles
  for (int l = 0; l <= n_layers; l++) fused = fused && sizes[l] <= 256;
  for (int l = 0; l < n_layers; l++) {
    if (!weights[l] || !biases[l]) return HB_EINVAL;
    if (b->d_mlp_wp[l]) { HB_IGN(hipFree(b->d_mlp_wp[l])); b->d_mlp_wp[l] = nullptr; }
    if (fused) {
      // B-operand order of v_mfma_f32_16x16x4_f32: wp[tile][k/4][lane] = W[4(k/4) + lane/16][16 tile + lane%16], zero padded
      const int K = sizes[l], N = sizes[l + 1], KK = (K + 3) / 4, ntile = (N + 15) / 16;
      std::vector<float> wp((size_t)ntile * KK * 64, 0.f);
      for (int nt = 0; nt < ntile; nt++)
        for (int kk = 0; kk < KK; kk++)
          for (int ln = 0; ln < 64; ln++) {
            const int k = 4 * kk + (ln >> 4), n = 16 * nt + (ln & 15);
            if (k < K && n < N) wp[((size_t)nt * KK + kk) * 64 + ln] = weights[l][(size_t)k * N + n];
          }
      if (hipMalloc((void**)&b->d_mlp_wp[l], wp.size() * sizeof(float)) != hipSuccess) return HB_ENOMEM;
      HB_HIP(hipMemcpy(b->d_mlp_wp[l], wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (b->d_mlp_w[l]) { HB_IGN(hipFree(b->d_mlp_w[l])); b->d_mlp_w[l] = nullptr; }
    if (b->d_mlp_b[l]) { HB_IGN(hipFree(b->d_mlp_b[l])); b->d_mlp_b[l] = nullptr; }
    size_t nw = (size_t)sizes[l] * sizes[l + 1];
    if (hipMalloc((void**)&b->d_mlp_w[l], nw * sizeof(float)) != hipSuccess || hipMalloc((void**)&b->d_mlp_b[l], sizes[l + 1] * sizeof(float)) != hipSuccess) return HB_ENOMEM;
    HB_HIP(hipMemcpy(b->d_mlp_w[l], weights[l], nw * sizeof(float), hipMemcpyHostToDevice));
    HB_HIP(hipMemcpy(b->d_mlp_b[l], biases[l], sizes[l + 1] * sizeof(float), hipMemcpyHostToDevice));
    if (l + 1 < n_layers) maxh = std::max(maxh, sizes[l + 1]);
  }
  for (int i = 0; i < 2; i++) {
    if (b->d_mlp_h[i]) { HB_IGN(hipFree(b->d_mlp_h[i])); b->d_mlp_h[i] = nullptr; }
    if (hipMalloc((void**)&b->d_mlp_h[i], (size_t)b->n_env * maxh * sizeof(float)) != hipSuccess) return HB_ENOMEM;
  }
  if (b->d_mlp_act) { HB_IGN(hipFree(b->d_mlp_act)); b->d_mlp_act = nullptr; }
  if (fused) {
    int widest = 1;
    for (int l = 0; l <= n_layers; l++) widest = std::max(widest, sizes[l]);
    const size_t floats = ((size_t)(b->n_env + 15) / 16 + hb_batch::kPipes) * 32 * (widest + 4);
    if (hipMalloc((void**)&b->d_mlp_act, floats * sizeof(float)) != hipSuccess) return HB_ENOMEM;
    HB_HIP(hipMemset(b->d_mlp_act, 0, floats * sizeof(float)));
  }
  b->mlp_layers = n_layers;
  b->mlp_fused = fused;
  for (int l = 0; l <= n_layers; l++) b->mlp_sizes[l] = sizes[l];
  return HB_OK;
}

// obs -> MLP -> ctrl for envs [lo, hi) on `st`
// (seg >= 0: env segment `seg` of a pipelined closed loop: the LDS-free kernel, which the GPU places beside the running step kernels)
static int policy_forward(hb_batch* b, int lo, int hi, hipStream_t st, int seg = -1) {
  if (b->mlp_layers < 1) return HB_EINVAL;
  const DevModel& dm = b->D.dm;
  if (b->mlp_fused) {
    PolicyDesc pd;
    memset(&pd, 0, sizeof pd);
    pd.nl = b->mlp_layers;
    int widest = 1;
    for (int l = 0; l <= b->mlp_layers; l++) { pd.sizes[l] = b->mlp_sizes[l]; widest = std::max(widest, b->mlp_sizes[l]); }
    for (int l = 0; l < b->mlp_layers; l++) { pd.w[l] = b->d_mlp_wp[l]; pd.b[l] = b->d_mlp_b[l]; }
    pd.ldx = widest + 4;  // + the K pad columns (K is swept four at a time); 16-row tiles
    const bool lean_ok = b->tune[HB_TUNE_POLICY_LEAN] != 0;
    const bool lean_all = b->tune[HB_TUNE_POLICY_LEAN] == 2;
    if (lean_all && seg < 0) seg = 0;
    if (seg >= 0 && lean_ok && b->d_mlp_act)
      HB_HIP(launch_policy_lean(dm, pd, b->d_state + (size_t)lo * dm.nstate, ctrl_for_write(b) + (size_t)lo * dm.nu,
                                b->d_mlp_act + ((size_t)(lo + 15) / 16 + seg) * 32 * pd.ldx, hi - lo, st));
    else
      HB_HIP(launch_policy(dm, pd, b->d_state + (size_t)lo * dm.nstate, ctrl_for_write(b) + (size_t)lo * dm.nu, hi - lo, st));
    return HB_OK;
  }
  // wide layers: one launch per layer, activations through HBM
  HB_HIP(launch_obs(dm, b->d_state + (size_t)lo * dm.nstate, b->d_obs + (size_t)lo * b->mlp_sizes[0], hi - lo, st));
  const float* x = b->d_obs + (size_t)lo * b->mlp_sizes[0];
  for (int l = 0; l < b->mlp_layers; l++) {
    float* y = ((l + 1 == b->mlp_layers) ? ctrl_for_write(b) : b->d_mlp_h[l & 1]) + (size_t)lo * b->mlp_sizes[l + 1];
    HB_HIP(launch_mlp_layer(x, b->d_mlp_w[l], b->d_mlp_b[l], y, hi - lo, b->mlp_sizes[l], b->mlp_sizes[l + 1], 1, st));
    x = y;
  }
  return HB_OK;
}

int hb_policy_eval(hb_batch* b, float* ctrl_out) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  int rc = policy_forward(b, 0, b->n_env, main_stream(b));
  if (rc != HB_OK) return rc;
  if (ctrl_out) HB_HIP(hipMemcpyAsync(ctrl_out, b->d_ctrl, (size_t)b->n_env * b->D.dm.nu * sizeof(float), hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}

int hb_rollout_policy(hb_batch* b, int T, float* qpos_out_dev) {
  if (!b || T < 1) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  // every segment is its own obs -> MLP -> mj_step chain: with pipelining on, the chains run side by side
  const int nseg = segment_count(b);
  int rc = fork_pipes(b, nseg);
  if (rc != HB_OK) return rc;
  for (int t = 0; t < T; t++) {
    const bool reorder = b->schedule && (b->launch_count % reorder_period(b) == 0);
    BatchPtrs P = make_ptrs(b);
    P.qfrc_out = nullptr;  // (the env adapter's read-out: nothing in this loop reads it, and without it the step launches are the lean kernels)
    P.ctrl = b->d_ctrl; P.ctrl_mode = 0;
    P.qpos_out = qpos_out_dev ? qpos_out_dev + (size_t)t * b->n_env * b->D.dm.nq : nullptr;
    for (int c = 0; c < nseg; c++) {
      const Segment sg = segment(b, c, nseg);
      rc = policy_forward(b, sg.lo, sg.hi, sg.st, nseg > 1 ? c : -1);
      if (rc == HB_OK) rc = launch_segment(b, P, 1, sg, nseg, reorder);
      if (rc != HB_OK) return rc;
    }
    steps_enqueued(b, nseg, reorder);
  }
  return HB_OK;
}

int hb_get_status(hb_batch* b, int* status) {
  if (!b || !status) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  HB_HIP(hipMemcpy(status, b->d_status, (size_t)b->n_env * sizeof(int), hipMemcpyDeviceToHost));
  return HB_OK;
}

int hb_env_warnings(hb_batch* b, int* warnings) {
  if (!b || !warnings) return HB_EINVAL;
  int rc = env_alloc(b);
  if (rc != HB_OK) return rc;
  HB_HIP(hipSetDevice(b->device));
  hipStream_t st = main_stream(b);
  std::vector<int> seen((size_t)b->n_env);
  HB_HIP(hipMemcpyAsync(warnings, b->d_status, seen.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  HB_HIP(hipMemcpyAsync(seen.data(), b->d_seen, seen.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  HB_HIP(hipMemsetAsync(b->d_seen, 0, seen.size() * sizeof(int), st));
  HB_HIP(hipStreamSynchronize(st));
  for (size_t e = 0; e < seen.size(); e++) warnings[e] |= seen[e];
  return HB_OK;
}

int hb_get_counts(hb_batch* b, int* ncon, int* nefc, int* niter) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  std::vector<int> h((size_t)b->n_env * kCountStride);
  HB_HIP(hipMemcpy(h.data(), b->d_counts, h.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (int e = 0; e < b->n_env; e++) {
    if (ncon) ncon[e] = h[kCountStride * e];
    if (nefc) nefc[e] = h[kCountStride * e + 1];
    if (niter) niter[e] = h[kCountStride * e + 2];
  }
  return HB_OK;
}

int hb_get_collision_counts(hb_batch* b, int* nwork, int* nsearch, int* kcycles) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  std::vector<int> h((size_t)b->n_env * kCountStride);
  HB_HIP(hipMemcpy(h.data(), b->d_counts, h.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (int e = 0; e < b->n_env; e++) {
    if (nwork) nwork[e] = h[kCountStride * e + 5];
    if (nsearch) nsearch[e] = h[kCountStride * e + 6];
    if (kcycles) kcycles[e] = h[kCountStride * e + 7];
  }
  return HB_OK;
}

const char* hb_last_kernel(hb_batch* b) {
  if (!b) return "";
  if (b->fold_n && flush_steps(b) != HB_OK) b->join_error = 1;  // (step calls held back: launched now, so that the name is theirs)
  return b->last_kernel;
}
long long hb_batch_step_launches(hb_batch* b) {
  if (!b) return HB_EINVAL;
  if (b->fold_n && flush_steps(b) != HB_OK) b->join_error = 1;
  return b->launch_count;
}
int hb_batch_device_name(const hb_batch* b, char* out, int cap) {
  if (!b || !out || cap < 2) return HB_EINVAL;
  hipDeviceProp_t prop;
  HB_HIP(hipGetDeviceProperties(&prop, b->device));
  snprintf(out, (size_t)cap, "%s (%s, %d CUs) #%d", prop.name, prop.gcnArchName, prop.multiProcessorCount, b->device);
  return HB_OK;
}
int hb_batch_tune(hb_batch* b, int knob, int value) {
  if (!b || knob < 0 || knob >= HB_TUNE_COUNT || value < 0) return HB_EINVAL;
  if (knob == HB_TUNE_DUO && value > 2) return HB_EINVAL;
  if (knob == HB_TUNE_FOLD && (value < 1 || value > kFoldMax)) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  (void)main_stream(b);  // the choice holds from the next launch on: whatever is in flight on the segments' streams is joined first
  b->tune[knob] = value;
  if (knob == HB_TUNE_SCHEDULE) { b->schedule = value != 0; b->order_mode = 0; }
  if (knob == HB_TUNE_STAGED || knob == HB_TUNE_FASTPASS) b->order_mode = 0;
  return HB_OK;
}

int hb_diag_enable(hb_batch* b, int on) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  if (on && !b->d_diag_qacc) {
    size_t n = b->n_env;
    if (hipMalloc((void**)&b->d_diag_qacc, n * b->D.dm.nv * sizeof(float)) != hipSuccess || hipMalloc((void**)&b->d_diag_force, n * b->D.dm.nefc_max * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&b->d_diag_contact, n * b->D.dm.ncon_max * kDiagConStride * sizeof(float)) != hipSuccess) return HB_ENOMEM;
  }
  b->diag = on != 0;
  return HB_OK;
}

static int copy_out(hb_batch* b, float* out, const float* dev, size_t n) {
  if (!b || !out) return HB_EINVAL;
  if (!b->diag || !dev) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  HB_HIP(hipMemcpy(out, dev, n * sizeof(float), hipMemcpyDeviceToHost));
  return HB_OK;
}
int hb_get_qacc(hb_batch* b, float* qacc) { return copy_out(b, qacc, b ? b->d_diag_qacc : nullptr, b ? (size_t)b->n_env * b->D.dm.nv : 0); }
int hb_get_efc_force(hb_batch* b, float* f) { return copy_out(b, f, b ? b->d_diag_force : nullptr, b ? (size_t)b->n_env * b->D.dm.nefc_max : 0); }
int hb_get_contacts(hb_batch* b, float* c) { return copy_out(b, c, b ? b->d_diag_contact : nullptr, b ? (size_t)b->n_env * b->D.dm.ncon_max * kDiagConStride : 0); }

void* hb_dev_alloc(hb_batch* b, uint64_t bytes) {
  if (!b || hipSetDevice(b->device) != hipSuccess) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return p;
}
void hb_dev_free(hb_batch* b, void* p) {
  if (!b || !p) return;
  HB_IGN(hipSetDevice(b->device));
  HB_IGN(hipStreamSynchronize(main_stream(b)));  // (step calls held back, fold_steps, may read the buffer: launched and waited for)
  HB_IGN(hipFree(p));
}
void* hb_host_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void hb_host_free(void* p) { if (p) HB_IGN(hipHostFree(p)); }
int hb_memcpy_h2d(hb_batch* b, void* dst_dev, const void* src, uint64_t bytes) {
  if (!b || !dst_dev || !src) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}
int hb_memcpy_d2h(hb_batch* b, void* dst, const void* src_dev, uint64_t bytes) {
  if (!b || !dst || !src_dev) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, main_stream(b)));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  return HB_OK;
}
int hb_halton_ctrl_dev(hb_batch* b, int T, int t0, int env_offset, float* out_dev) {
  if (!b || T < 1 || !out_dev) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(launch_halton_ctrl(out_dev, T, b->n_env, b->D.dm.nu, t0, env_offset, main_stream(b)));
  return HB_OK;
}
int hb_get_stamps(hb_batch* b, unsigned long long* out) {
#ifdef HB_STAMPS
  if (!b || !out) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  if (!b->d_stamps) {
    if (hipMalloc((void**)&b->d_stamps, (size_t)b->n_env * 16 * sizeof(unsigned long long)) != hipSuccess) return HB_ENOMEM;
    HB_HIP(hipMemset(b->d_stamps, 0, (size_t)b->n_env * 16 * sizeof(unsigned long long)));
    return HB_OK;  // first call arms the stamps; call again after a step to read them
  }
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  HB_HIP(hipMemcpy(out, b->d_stamps, (size_t)b->n_env * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return HB_OK;
#else
  (void)b; (void)out;
  return HB_EUNSUPPORTED;
#endif
}
int hb_step_timing(hb_batch* b, int enable) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  if (enable && b->tev.empty()) {
    b->tev.resize(512);
    for (auto& e : b->tev) HB_HIP(hipEventCreate(&e));
  }
  b->time_steps = enable != 0;
  b->tev_used = 0;
  return HB_OK;
}
int hb_step_timing_read(hb_batch* b, float* mean_us, int* samples) {
  if (!b || !mean_us) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipStreamSynchronize(main_stream(b)));
  double tot = 0;
  int n = 0;
  for (int i = 0; i + 1 < b->tev_used; i += 2) {
    float ms = 0;
    HB_HIP(hipEventElapsedTime(&ms, b->tev[i], b->tev[i + 1]));
    tot += ms; n++;
  }
  *mean_us = n ? (float)(1e3 * tot / n) : 0.f;
  if (samples) *samples = n;
  b->tev_used = 0;
  return HB_OK;
}
int hb_timer_start(hb_batch* b) {
  if (!b) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  if (!b->ev0) { HB_HIP(hipEventCreate(&b->ev0)); HB_HIP(hipEventCreate(&b->ev1)); }
  HB_HIP(hipEventRecord(b->ev0, main_stream(b)));
  return HB_OK;
}
int hb_timer_stop(hb_batch* b, float* elapsed_ms) {
  if (!b || !elapsed_ms || !b->ev0) return HB_EINVAL;
  HB_HIP(hipSetDevice(b->device));
  HB_HIP(hipEventRecord(b->ev1, main_stream(b)));
  HB_HIP(hipEventSynchronize(b->ev1));
  HB_HIP(hipEventElapsedTime(elapsed_ms, b->ev0, b->ev1));
  return HB_OK;
}

}  // extern "C"
