// setconst.cpp — compile-time constants that mj_setConst derives from the model at qpos0
// (reference API: simulation/mujoco/include/mujoco/mujoco.h:221; fields mjmodel.h:547,679,
// 681,725,727,971,973): body_subtreemass, dof_M0, dof_invweight0, body_invweight0,
// tendon_length0, tendon_invweight0, stat.meaninertia.  They feed constraint regularisation
// (efc_diagApprox) and the PGS termination scale.  fp64, host only, runs once per model load.
//
// This is product code (the model loader needs it); it deliberately does not share code with
// oracle/ — the oracle recomputes the same quantities independently and tests compare them.
#include "hb_model.hpp"
#include "hmath.hpp"
#include <cstring>

namespace hb {
namespace {

struct Kin {
  std::vector<double> xpos, xquat, xmat, xipos, ximat, xanchor, xaxis, subtree_com, cinert, cdof, crb, qM, qLD, qLDiagInv;
};

void inert_com(double* res, const double* inert, const double* mat, const double* dif, double mass) {
  // res[0..5] = R diag(inert) R^T (xx,yy,zz,xy,xz,yz) shifted by dif; res[6..8] = mass*dif; res[9] = mass
  double t[9];
  for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) t[3 * r + k] = mat[3 * r + k] * inert[k];
  auto rr = [&](int a, int b) { return t[3 * a] * mat[3 * b] + t[3 * a + 1] * mat[3 * b + 1] + t[3 * a + 2] * mat[3 * b + 2]; };
  res[0] = rr(0, 0) + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
  res[1] = rr(1, 1) + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
  res[2] = rr(2, 2) + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
  res[3] = rr(0, 1) - mass * dif[0] * dif[1];
  res[4] = rr(0, 2) - mass * dif[0] * dif[2];
  res[5] = rr(1, 2) - mass * dif[1] * dif[2];
  res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2];
  res[9] = mass;
}

void mul_inert_vec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}

void position_stage(const Model& m, const double* qpos, Kin& k) {
  int nb = m.nbody, nv = m.nv;
  k.xpos.assign(3 * nb, 0); k.xquat.assign(4 * nb, 0); k.xmat.assign(9 * nb, 0); k.xipos.assign(3 * nb, 0);
  k.ximat.assign(9 * nb, 0); k.xanchor.assign(3 * m.njnt, 0); k.xaxis.assign(3 * m.njnt, 0);
  k.subtree_com.assign(3 * nb, 0); k.cinert.assign(10 * nb, 0); k.cdof.assign(6 * nv, 0); k.crb.assign(10 * nb, 0);
  k.qM.assign(m.nM, 0); k.qLD.assign(m.nM, 0); k.qLDiagInv.assign(nv, 0);
  k.xquat[0] = 1; k.xmat[0] = k.xmat[4] = k.xmat[8] = 1; k.ximat[0] = k.ximat[4] = k.ximat[8] = 1;
  for (int b = 1; b < nb; b++) {
    double pos[3], quat[4];
    int p = m.body_parentid[b];
    bool isfree = m.body_jntnum[b] == 1 && m.jnt_type[m.body_jntadr[b]] == JNT_FREE;
    if (isfree) {
      int j = m.body_jntadr[b], qa = m.jnt_qposadr[j];
      for (int i = 0; i < 3; i++) pos[i] = qpos[qa + i];
      for (int i = 0; i < 4; i++) quat[i] = qpos[qa + 3 + i];
      hm::normalize4(quat);
      for (int i = 0; i < 3; i++) { k.xanchor[3 * j + i] = pos[i]; k.xaxis[3 * j + i] = m.jnt_axis[3 * j + i]; }
    } else {
      hm::rot_vec_quat(pos, &m.body_pos[3 * b], &k.xquat[4 * p]);
      for (int i = 0; i < 3; i++) pos[i] += k.xpos[3 * p + i];
      hm::mul_quat(quat, &k.xquat[4 * p], &m.body_quat[4 * b]);
      for (int jj = 0; jj < m.body_jntnum[b]; jj++) {
        int j = m.body_jntadr[b] + jj, qa = m.jnt_qposadr[j];
        double axis[3], anchor[3];
        hm::rot_vec_quat(axis, &m.jnt_axis[3 * j], quat);
        hm::rot_vec_quat(anchor, &m.jnt_pos[3 * j], quat);
        for (int i = 0; i < 3; i++) anchor[i] += pos[i];
        for (int i = 0; i < 3; i++) { k.xanchor[3 * j + i] = anchor[i]; k.xaxis[3 * j + i] = axis[i]; }
        double dq = qpos[qa] - m.qpos0[qa];
        if (m.jnt_type[j] == JNT_SLIDE) {
          for (int i = 0; i < 3; i++) pos[i] += axis[i] * dq;
        } else {
          double ql[4], t[4], v[3];
          hm::axis_angle2quat(ql, &m.jnt_axis[3 * j], dq);
          hm::mul_quat(t, quat, ql);
          memcpy(quat, t, sizeof t);
          hm::rot_vec_quat(v, &m.jnt_pos[3 * j], quat);
          for (int i = 0; i < 3; i++) pos[i] = anchor[i] - v[i];
        }
      }
      hm::normalize4(quat);
    }
    memcpy(&k.xpos[3 * b], pos, sizeof pos);
    memcpy(&k.xquat[4 * b], quat, sizeof quat);
    hm::quat2mat(&k.xmat[9 * b], quat);
    double v[3], qi[4];
    hm::rot_vec_quat(v, &m.body_ipos[3 * b], quat);
    for (int i = 0; i < 3; i++) k.xipos[3 * b + i] = pos[i] + v[i];
    hm::mul_quat(qi, quat, &m.body_iquat[4 * b]);
    hm::quat2mat(&k.ximat[9 * b], qi);
  }
  // subtree centres of mass
  std::vector<double> sm(nb, 0);
  for (int b = 0; b < nb; b++) { sm[b] = m.body_mass[b]; for (int i = 0; i < 3; i++) k.subtree_com[3 * b + i] = m.body_mass[b] * k.xipos[3 * b + i]; }
  for (int b = nb - 1; b > 0; b--) { int p = m.body_parentid[b]; sm[p] += sm[b]; for (int i = 0; i < 3; i++) k.subtree_com[3 * p + i] += k.subtree_com[3 * b + i]; }
  for (int b = 0; b < nb; b++)
    for (int i = 0; i < 3; i++) k.subtree_com[3 * b + i] = sm[b] > 1e-15 ? k.subtree_com[3 * b + i] / sm[b] : k.xipos[3 * b + i];
  // cinert, cdof
  for (int b = 1; b < nb; b++) {
    const double* com = &k.subtree_com[3 * m.body_rootid[b]];
    double dif[3] = {k.xipos[3 * b] - com[0], k.xipos[3 * b + 1] - com[1], k.xipos[3 * b + 2] - com[2]};
    inert_com(&k.cinert[10 * b], &m.body_inertia[3 * b], &k.ximat[9 * b], dif, m.body_mass[b]);
  }
  for (int j = 0; j < m.njnt; j++) {
    int b = m.jnt_bodyid[j], da = m.jnt_dofadr[j];
    const double* com = &k.subtree_com[3 * m.body_rootid[b]];
    double off[3] = {com[0] - k.xanchor[3 * j], com[1] - k.xanchor[3 * j + 1], com[2] - k.xanchor[3 * j + 2]};
    if (m.jnt_type[j] == JNT_FREE) {
      for (int i = 0; i < 3; i++) k.cdof[6 * (da + i) + 3 + i] = 1;
      for (int i = 0; i < 3; i++) {
        double ax[3] = {k.xmat[9 * b + i], k.xmat[9 * b + 3 + i], k.xmat[9 * b + 6 + i]};
        double* cd = &k.cdof[6 * (da + 3 + i)];
        cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2];
        hm::cross(cd + 3, ax, off);
      }
    } else if (m.jnt_type[j] == JNT_SLIDE) {
      for (int i = 0; i < 3; i++) k.cdof[6 * da + 3 + i] = k.xaxis[3 * j + i];
    } else {
      double* cd = &k.cdof[6 * da];
      for (int i = 0; i < 3; i++) cd[i] = k.xaxis[3 * j + i];
      hm::cross(cd + 3, &k.xaxis[3 * j], off);
    }
  }
  // composite rigid body -> qM
  k.crb = k.cinert;
  for (int b = nb - 1; b > 0; b--) { int p = m.body_parentid[b]; if (p > 0) for (int i = 0; i < 10; i++) k.crb[10 * p + i] += k.crb[10 * b + i]; }
  for (int i = 0; i < nv; i++) {
    int adr = m.dof_Madr[i];
    double buf[6];
    mul_inert_vec(buf, &k.crb[10 * m.dof_bodyid[i]], &k.cdof[6 * i]);
    k.qM[adr] = m.dof_armature[i];
    for (int j = i; j >= 0; j = m.dof_parentid[j]) {
      double s = 0;
      for (int t = 0; t < 6; t++) s += k.cdof[6 * j + t] * buf[t];
      k.qM[adr++] += s;
    }
  }
  // L^T D L factorisation on the ancestor-chain layout
  k.qLD = k.qM;
  for (int kk = nv - 1; kk >= 0; kk--) {
    int Mkk = m.dof_Madr[kk], Mki = Mkk + 1, i = m.dof_parentid[kk];
    if (k.qLD[Mkk] < 1e-15) k.qLD[Mkk] = 1e-15;
    while (i >= 0) {
      double tmp = k.qLD[Mki] / k.qLD[Mkk];
      int cnt = (i < nv - 1 ? m.dof_Madr[i + 1] : m.nM) - m.dof_Madr[i];
      for (int t = 0; t < cnt; t++) k.qLD[m.dof_Madr[i] + t] -= k.qLD[Mki + t] * tmp;
      k.qLD[Mki] = tmp;
      i = m.dof_parentid[i];
      Mki++;
    }
  }
  for (int i = 0; i < nv; i++) k.qLDiagInv[i] = 1.0 / k.qLD[m.dof_Madr[i]];
}

void solve_m(const Model& m, const Kin& k, double* x) {
  int nv = m.nv;
  for (int kk = nv - 1; kk >= 0; kk--) {
    int Mki = m.dof_Madr[kk] + 1, i = m.dof_parentid[kk];
    while (i >= 0) { x[i] -= k.qLD[Mki] * x[kk]; Mki++; i = m.dof_parentid[i]; }
  }
  for (int i = 0; i < nv; i++) x[i] *= k.qLDiagInv[i];
  for (int kk = 0; kk < nv; kk++) {
    int Mki = m.dof_Madr[kk] + 1, i = m.dof_parentid[kk];
    while (i >= 0) { x[kk] -= k.qLD[Mki] * x[i]; Mki++; i = m.dof_parentid[i]; }
  }
}

// Jacobian of a point attached to a body: jacp[3*nv], jacr[3*nv] (row-major 3 x nv)
void jac_point(const Model& m, const Kin& k, double* jacp, double* jacr, const double* point, int body) {
  int nv = m.nv;
  for (int i = 0; i < 3 * nv; i++) jacp[i] = jacr[i] = 0;
  const double* com = &k.subtree_com[3 * m.body_rootid[body]];
  double off[3] = {point[0] - com[0], point[1] - com[1], point[2] - com[2]};
  while (body > 0 && m.body_dofnum[body] == 0) body = m.body_parentid[body];
  if (body == 0) return;
  int i = m.body_dofadr[body] + m.body_dofnum[body] - 1;
  while (i >= 0) {
    const double* cd = &k.cdof[6 * i];
    double c[3];
    hm::cross(c, cd, off);
    for (int r = 0; r < 3; r++) { jacr[r * nv + i] = cd[r]; jacp[r * nv + i] = cd[3 + r] + c[r]; }
    i = m.dof_parentid[i];
  }
}

}  // namespace

bool set_const(Model& m, std::string& err) {
  int nv = m.nv, nb = m.nbody;
  if (nv == 0) { err = "model has no degrees of freedom"; return false; }
  Kin k;
  position_stage(m, m.qpos0.data(), k);
  // subtree mass
  for (int b = 0; b < nb; b++) m.body_subtreemass[b] = m.body_mass[b];
  for (int b = nb - 1; b > 0; b--) m.body_subtreemass[m.body_parentid[b]] += m.body_subtreemass[b];
  // dof_M0, meaninertia
  double mean = 0;
  for (int i = 0; i < nv; i++) { m.dof_M0[i] = k.qM[m.dof_Madr[i]]; mean += m.dof_M0[i]; }
  m.meaninertia = mean / nv;
  for (int i = 0; i < nv; i++)
    if (!(k.qLD[m.dof_Madr[i]] > 1e-12)) { err = "mass matrix is singular at qpos0 (massless chain or missing armature?)"; return false; }
  // body_invweight0: (J M^-1 J^T) averaged over translational / rotational blocks, at the body com
  std::vector<double> jacp(3 * nv), jacr(3 * nv), col(nv);
  for (int b = 1; b < nb; b++) {
    jac_point(m, k, jacp.data(), jacr.data(), &k.xipos[3 * b], b);
    double tran = 0, rot = 0;
    for (int r = 0; r < 3; r++) {
      for (int i = 0; i < nv; i++) col[i] = jacp[r * nv + i];
      solve_m(m, k, col.data());
      for (int i = 0; i < nv; i++) tran += jacp[r * nv + i] * col[i];
      for (int i = 0; i < nv; i++) col[i] = jacr[r * nv + i];
      solve_m(m, k, col.data());
      for (int i = 0; i < nv; i++) rot += jacr[r * nv + i] * col[i];
    }
    m.body_invweight0[2 * b] = tran / 3;
    m.body_invweight0[2 * b + 1] = rot / 3;
  }
  m.body_invweight0[0] = m.body_invweight0[1] = 0;
  // dof_invweight0: diag(M^-1), averaged over the 3 translational / 3 rotational dofs of a free joint
  std::vector<double> dinv(nv);
  for (int i = 0; i < nv; i++) {
    for (int t = 0; t < nv; t++) col[t] = 0;
    col[i] = 1;
    solve_m(m, k, col.data());
    dinv[i] = col[i];
  }
  for (int j = 0; j < m.njnt; j++) {
    int da = m.jnt_dofadr[j];
    if (m.jnt_type[j] == JNT_FREE) {
      double a = (dinv[da] + dinv[da + 1] + dinv[da + 2]) / 3, b = (dinv[da + 3] + dinv[da + 4] + dinv[da + 5]) / 3;
      for (int i = 0; i < 3; i++) { m.dof_invweight0[da + i] = a; m.dof_invweight0[da + 3 + i] = b; }
    } else {
      m.dof_invweight0[da] = dinv[da];
    }
  }
  // tendons
  for (int t = 0; t < m.ntendon; t++) {
    double len = 0;
    for (int i = 0; i < nv; i++) col[i] = 0;
    for (int w = 0; w < m.tendon_num[t]; w++) {
      int j = m.wrap_objid[m.tendon_adr[t] + w];
      double coef = m.wrap_prm[m.tendon_adr[t] + w];
      len += coef * m.qpos0[m.jnt_qposadr[j]];
      col[m.jnt_dofadr[j]] += coef;
    }
    std::vector<double> J(col);
    solve_m(m, k, col.data());
    double w = 0;
    for (int i = 0; i < nv; i++) w += J[i] * col[i];
    m.tendon_length0[t] = len;
    m.tendon_invweight0[t] = w;
  }
  return true;
}

}  // namespace hb
