/* mjstep_oracle.c — fp64 CPU restatement of the reference's mj_step pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (humanoid_mujoco_amd/, include/) may
 * link, import or execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker and the CPU baseline.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in a third-party dependency that is absent
 * from /root/reference: google-deepmind/mujoco (pinned 3.1.4 by mujoco_mpc/CMakeLists.txt:58-61;
 * headers vendored at 3.1.1 under simulation/mujoco/include/mujoco/).  Neither its sources nor
 * a Linux binary nor the Python package are available and the reference holds no golden vectors
 * for mj_step.  This file restates the published algorithm (MuJoCo
 * "Computation" documentation and the API contracts in mujoco.h) stage by stage; every function
 * cites the declaration it follows.  It is validated by closed-form cases and invariants in
 * tests/test_oracle_*.py, not against MuJoCo output.  Both constraint solvers of the path are here: PGS
 * (the benchmark configuration) and Newton (mjOption's default, what the reference's humanoid.xml runs);
 * they solve the dual and the primal form of one convex problem, and tests/test_oracle_newton.py pins each
 * with the other (Newton vs PGS run to convergence: the same qacc to 1e-6).
 *
 * Call sites of the path in the reference: simulation/cpu_env.py:684 (mujoco.mj_step),
 * mujoco_mpc/mjpc/trajectory.cc:158, simulation/mujoco/sample/testspeed.cc:96.
 *
 * Build: gcc -O2 -shared -fPIC -o liboracle.so mjstep_oracle.c -lm -lpthread   (see Makefile)
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static long g_pgs_reverts = 0;

enum { SOL_PGS = 0, SOL_CG = 1, SOL_NEWTON = 2 }; /* mjtSolver, mjmodel.h:159-163 */

#define MINVAL 1E-15   /* mjMINVAL */
#define MAXVAL 1E+10   /* mjmodel.h:23 mjMAXVAL */
#define MINIMP 0.0001  /* mjmodel.h:25 */
#define MAXIMP 0.9999  /* mjmodel.h:26 */
#define MINMU 1E-5     /* mjmodel.h:24 */

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { GEOM_PLANE = 0, GEOM_HFIELD = 1, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_CYLINDER = 5, GEOM_MESH = 7 }; /* mjtGeom, mjmodel.h:94-103 */
enum { CNSTR_LIMIT_JOINT = 3, CNSTR_LIMIT_TENDON = 4, CNSTR_CONTACT_FRICTIONLESS = 5, CNSTR_CONTACT_PYRAMIDAL = 6 }; /* mjmodel.h:256-265 */
enum { DSBL_CONSTRAINT = 1, DSBL_LIMIT = 8, DSBL_CONTACT = 16, DSBL_PASSIVE = 32, DSBL_GRAVITY = 64, DSBL_CLAMPCTRL = 128,
       DSBL_WARMSTART = 256, DSBL_ACTUATION = 1024, DSBL_REFSAFE = 2048, DSBL_EULERDAMP = 16384 };
enum { WARN_CONTACTFULL = 1, WARN_CNSTRFULL = 2, WARN_BADQPOS = 4, WARN_BADQVEL = 5, WARN_BADQACC = 6 }; /* mjdata.h:54-65 */

#define OM_MAXCON 128
#define OM_MAXEFC 640

typedef struct {
  int nq, nv, nu, nbody, njnt, ngeom, ntendon, nwrap, nM, nkey, npair, nhfield, nhfielddata, nmesh, nmeshvert, nmeshnbr;
  int mpr_iterations;   /* mjOption.mpr_iterations, mjmodel.h:437 (default 50) */
  double mpr_tolerance; /* mjOption.mpr_tolerance, mjmodel.h:413 (default 1e-6) */
  double timestep, impratio, tolerance, meaninertia, gravity[3];
  int integrator, cone, solver, iterations, disableflags;
  int ls_iterations;    /* mjOption.ls_iterations, mjmodel.h:434 (default 50) */
  double ls_tolerance;  /* mjOption.ls_tolerance, mjmodel.h:411 (default 0.01) */
  int *body_parentid, *body_rootid, *body_weldid, *body_jntnum, *body_jntadr, *body_dofnum, *body_dofadr;
  double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_subtreemass, *body_inertia, *body_invweight0;
  int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  double *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp;
  int *dof_bodyid, *dof_jntid, *dof_parentid, *dof_Madr;
  double *dof_armature, *dof_damping, *dof_invweight0, *dof_M0;
  int *geom_type, *geom_bodyid, *geom_condim, *geom_priority, *geom_dataid;
  double *geom_size, *geom_pos, *geom_quat, *geom_rbound, *geom_friction, *geom_solmix, *geom_solref, *geom_solimp, *geom_margin, *geom_gap;
  int *hfield_nrow, *hfield_ncol, *hfield_adr;
  double *hfield_size, *hfield_data;
  int *mesh_vertadr, *mesh_vertnum; /* convex-hull vertices of each mesh in the geom frame (what mesh collision uses) */
  double* mesh_vert;
  int *mesh_nbradr, *mesh_nbrnum, *mesh_nbr; /* the hull's edge graph: per hull vertex its neighbours (indices local to the mesh) */
  int *tendon_adr, *tendon_num, *tendon_limited, *wrap_objid;
  double *tendon_range, *tendon_margin, *tendon_solref_lim, *tendon_solimp_lim, *tendon_invweight0, *tendon_length0, *wrap_prm;
  int *actuator_trnid, *actuator_ctrllimited, *actuator_forcelimited;
  double *actuator_gear, *actuator_ctrlrange, *actuator_forcerange, *actuator_gainprm, *actuator_biasprm;
  int *pair_geom1, *pair_geom2;
  double *qpos0, *qpos_spring, *key_qpos;
} om_model;

typedef struct {  /* mjContact, mjdata.h:99-136 */
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
  int dim, geom1, geom2, efc_address;
} om_contact;

typedef struct {  /* the subset of mjData (mjdata.h:164-431) this path touches */
  double time;
  double *qpos, *qvel, *ctrl, *qacc_warmstart, *qfrc_applied, *xfrc_applied, *qacc;
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat;
  double *subtree_com, *cinert, *cdof, *cdof_dot, *crb, *cvel, *cacc, *cfrc_body;
  double *qM, *qLD, *qLDiagInv, *qH, *qHDiagInv;
  double *ten_length, *ten_J, *actuator_force;
  double *qfrc_passive, *qfrc_bias, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint;
  int ncon, nefc, nl;
  om_contact* contact;
  int *efc_type, *efc_id;
  double *efc_J, *efc_pos, *efc_margin, *efc_diagApprox, *efc_R, *efc_D, *efc_KBIP, *efc_vel, *efc_aref, *efc_b, *efc_force, *efc_AR, *efc_B;
  int solver_niter;
  int solver_nls;       /* line-search evaluations of the last Newton solve (statistics) */
  int warning[8];
  /* counters for statistics (testspeed.cc:97-98 accumulates ncon/nefc the same way) */
  long long sum_ncon, sum_nefc, sum_iter, nstep;
  int max_ncon, max_nefc;
} om_data;

/* ------------------------------------------------------------------ .hbm loader ---------- */

typedef struct { char kind; char name[48]; int n; int* iv; double* dv; } om_rec;

static om_rec* find_rec(om_rec* r, int nr, const char* name) {
  for (int i = 0; i < nr; i++) if (!strcmp(r[i].name, name)) return r + i;
  return NULL;
}
static int rec_int(om_rec* r, int nr, const char* name, int def) { om_rec* x = find_rec(r, nr, name); return (x && x->kind == 'i') ? x->iv[0] : def; }
static double rec_dbl(om_rec* r, int nr, const char* name, double def) { om_rec* x = find_rec(r, nr, name); return (x && x->kind == 'd') ? x->dv[0] : def; }
static int* rec_iarr(om_rec* r, int nr, const char* name, int need) {
  om_rec* x = find_rec(r, nr, name);
  int* out = (int*)calloc(need > 0 ? need : 1, sizeof(int));
  if (x && x->kind == 'I') for (int i = 0; i < need && i < x->n; i++) out[i] = x->iv[i];
  return out;
}
static double* rec_darr(om_rec* r, int nr, const char* name, int need) {
  om_rec* x = find_rec(r, nr, name);
  double* out = (double*)calloc(need > 0 ? need : 1, sizeof(double));
  if (x && x->kind == 'D') for (int i = 0; i < need && i < x->n; i++) out[i] = x->dv[i];
  return out;
}

om_model* om_load(const char* path, char* err, int errsz) {
  FILE* f = fopen(path, "rb");
  if (!f) { snprintf(err, errsz, "cannot open %s", path); return NULL; }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char* text = (char*)malloc(sz + 1);
  if (fread(text, 1, sz, f) != (size_t)sz) { fclose(f); free(text); snprintf(err, errsz, "short read"); return NULL; }
  text[sz] = 0;
  fclose(f);
  if (strncmp(text, "HBM1", 4)) { free(text); snprintf(err, errsz, "not an HBM1 file"); return NULL; }
  int cap = 256, nr = 0;
  om_rec* recs = (om_rec*)calloc(cap, sizeof(om_rec));
  char* save = NULL;
  for (char* line = strtok_r(text, "\n", &save); line; line = strtok_r(NULL, "\n", &save)) {
    if (!strncmp(line, "END", 3)) break;
    if (!strncmp(line, "HBM1", 4) || line[0] == '#' || !line[0]) continue;
    char kind = line[0];
    char* p = line + 2;
    om_rec* r = recs + nr;
    int k = 0;
    while (*p && *p != ' ' && k < 47) r->name[k++] = *p++;
    r->name[k] = 0;
    r->kind = kind;
    if (kind == 'i') { r->n = 1; r->iv = (int*)malloc(sizeof(int)); r->iv[0] = (int)strtol(p, NULL, 10); }
    else if (kind == 'd') { r->n = 1; r->dv = (double*)malloc(sizeof(double)); r->dv[0] = strtod(p, NULL); }
    else if (kind == 'I' || kind == 'D') {
      char* e;
      long n = strtol(p, &e, 10);
      p = e;
      r->n = (int)n;
      if (kind == 'I') { r->iv = (int*)malloc(sizeof(int) * (n + 1)); for (long i = 0; i < n; i++) { r->iv[i] = (int)strtol(p, &e, 10); p = e; } }
      else { r->dv = (double*)malloc(sizeof(double) * (n + 1)); for (long i = 0; i < n; i++) { r->dv[i] = strtod(p, &e); p = e; } }
    } else continue; /* S records (names) are not needed here */
    if (++nr == cap) { cap *= 2; recs = (om_rec*)realloc(recs, cap * sizeof(om_rec)); memset(recs + nr, 0, (cap - nr) * sizeof(om_rec)); }
  }
  om_model* m = (om_model*)calloc(1, sizeof(om_model));
#define RI(x) m->x = rec_int(recs, nr, #x, 0)
#define RD(x) m->x = rec_dbl(recs, nr, #x, 0)
#define AI(x, n) m->x = rec_iarr(recs, nr, #x, n)
#define AD(x, n) m->x = rec_darr(recs, nr, #x, n)
  RI(nq); RI(nv); RI(nu); RI(nbody); RI(njnt); RI(ngeom); RI(ntendon); RI(nwrap); RI(nM); RI(nkey); RI(npair); RI(nhfield); RI(nhfielddata); RI(nmesh); RI(nmeshvert); RI(nmeshnbr);
  m->mpr_iterations = 50; m->mpr_tolerance = 1e-6;
  RD(timestep); RD(impratio); RD(tolerance); RD(meaninertia);
  RI(integrator); RI(cone); RI(solver); RI(iterations); RI(disableflags);
  m->ls_iterations = rec_int(recs, nr, "ls_iterations", 50); m->ls_tolerance = rec_dbl(recs, nr, "ls_tolerance", 0.01);
  { double* g = rec_darr(recs, nr, "gravity", 3); memcpy(m->gravity, g, sizeof m->gravity); free(g); }
  int nb = m->nbody, nj = m->njnt, nv = m->nv, ng = m->ngeom, nt = m->ntendon, nu = m->nu;
  AI(body_parentid, nb); AI(body_rootid, nb); AI(body_weldid, nb); AI(body_jntnum, nb); AI(body_jntadr, nb); AI(body_dofnum, nb); AI(body_dofadr, nb);
  AD(body_pos, 3 * nb); AD(body_quat, 4 * nb); AD(body_ipos, 3 * nb); AD(body_iquat, 4 * nb); AD(body_mass, nb); AD(body_subtreemass, nb);
  AD(body_inertia, 3 * nb); AD(body_invweight0, 2 * nb);
  AI(jnt_type, nj); AI(jnt_qposadr, nj); AI(jnt_dofadr, nj); AI(jnt_bodyid, nj); AI(jnt_limited, nj);
  AD(jnt_pos, 3 * nj); AD(jnt_axis, 3 * nj); AD(jnt_stiffness, nj); AD(jnt_range, 2 * nj); AD(jnt_margin, nj); AD(jnt_solref, 2 * nj); AD(jnt_solimp, 5 * nj);
  AI(dof_bodyid, nv); AI(dof_jntid, nv); AI(dof_parentid, nv); AI(dof_Madr, nv);
  AD(dof_armature, nv); AD(dof_damping, nv); AD(dof_invweight0, nv); AD(dof_M0, nv);
  AI(geom_type, ng); AI(geom_bodyid, ng); AI(geom_condim, ng); AI(geom_priority, ng); AI(geom_dataid, ng);
  AD(geom_size, 3 * ng); AD(geom_pos, 3 * ng); AD(geom_quat, 4 * ng); AD(geom_rbound, ng); AD(geom_friction, 3 * ng); AD(geom_solmix, ng);
  AD(geom_solref, 2 * ng); AD(geom_solimp, 5 * ng); AD(geom_margin, ng); AD(geom_gap, ng);
  AI(hfield_nrow, m->nhfield); AI(hfield_ncol, m->nhfield); AI(hfield_adr, m->nhfield); AD(hfield_size, 4 * m->nhfield); AD(hfield_data, m->nhfielddata);
  AI(mesh_vertadr, m->nmesh); AI(mesh_vertnum, m->nmesh); AD(mesh_vert, 3 * m->nmeshvert);
  AI(mesh_nbradr, m->nmeshvert); AI(mesh_nbrnum, m->nmeshvert); AI(mesh_nbr, m->nmeshnbr);
  AI(tendon_adr, nt); AI(tendon_num, nt); AI(tendon_limited, nt); AI(wrap_objid, m->nwrap);
  AD(tendon_range, 2 * nt); AD(tendon_margin, nt); AD(tendon_solref_lim, 2 * nt); AD(tendon_solimp_lim, 5 * nt); AD(tendon_invweight0, nt);
  AD(tendon_length0, nt); AD(wrap_prm, m->nwrap);
  AI(actuator_trnid, nu); AI(actuator_ctrllimited, nu); AI(actuator_forcelimited, nu);
  AD(actuator_gear, nu); AD(actuator_ctrlrange, 2 * nu); AD(actuator_forcerange, 2 * nu); AD(actuator_gainprm, nu); AD(actuator_biasprm, 3 * nu);
  AI(pair_geom1, m->npair); AI(pair_geom2, m->npair);
  AD(qpos0, m->nq); AD(qpos_spring, m->nq); AD(key_qpos, m->nkey * m->nq);
  for (int i = 0; i < nr; i++) { free(recs[i].iv); free(recs[i].dv); }
  free(recs);
  free(text);
  if (m->nq <= 0 || m->nv <= 0 || m->nbody <= 1) { snprintf(err, errsz, "bad model sizes"); free(m); return NULL; }
  return m;
}

/* ------------------------------------------------------------------ small math ----------- */

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double normalize3(double* v) {
  double n = sqrt(dot3(v, v));
  if (n < MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; }
  else { v[0] /= n; v[1] /= n; v[2] /= n; }
  return n;
}
static void normalize4(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void mulquat(double* r, const double* a, const double* b) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof t);
}
static void quat2mat(double* m, const double* q) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02);
  m[3] = 2 * (q12 + q03); m[4] = q00 - q11 + q22 - q33; m[5] = 2 * (q23 - q01);
  m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01); m[8] = q00 - q11 - q22 + q33;
}
static void mulmatvec3(double* r, const double* m, const double* v) {
  double t[3] = {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2], m[6] * v[0] + m[7] * v[1] + m[8] * v[2]};
  memcpy(r, t, sizeof t);
}
static void rotvecquat(double* r, const double* v, const double* q) { double m[9]; quat2mat(m, q); mulmatvec3(r, m, v); }
static void axisangle2quat(double* q, const double* axis, double ang) {
  double s = sin(ang * 0.5);
  q[0] = cos(ang * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* spatial algebra, 6-vectors are (rotation, translation) — mjdata.h:268-270,316-317 */
static void mul_inert_vec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void cross_motion(double* r, const double* vel, const double* v) {
  r[0] = -vel[2] * v[1] + vel[1] * v[2];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = -vel[1] * v[0] + vel[0] * v[1];
  r[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
static void cross_force(double* r, const double* vel, const double* f) {
  r[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  r[3] = -vel[2] * f[4] + vel[1] * f[5];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = -vel[1] * f[3] + vel[0] * f[4];
}

/* ------------------------------------------------------------------ data ------------------ */

om_data* om_make_data(const om_model* m) { /* mj_makeData, mujoco.h:173 */
  om_data* d = (om_data*)calloc(1, sizeof(om_data));
  int nb = m->nbody, nv = m->nv, nj = m->njnt, ng = m->ngeom;
#define AL(x, n) d->x = (double*)calloc((n) > 0 ? (n) : 1, sizeof(double))
  AL(qpos, m->nq); AL(qvel, nv); AL(ctrl, m->nu); AL(qacc_warmstart, nv); AL(qfrc_applied, nv); AL(xfrc_applied, 6 * nb); AL(qacc, nv);
  AL(xpos, 3 * nb); AL(xquat, 4 * nb); AL(xmat, 9 * nb); AL(xipos, 3 * nb); AL(ximat, 9 * nb); AL(xanchor, 3 * nj); AL(xaxis, 3 * nj);
  AL(geom_xpos, 3 * ng); AL(geom_xmat, 9 * ng);
  AL(subtree_com, 3 * nb); AL(cinert, 10 * nb); AL(cdof, 6 * nv); AL(cdof_dot, 6 * nv); AL(crb, 10 * nb); AL(cvel, 6 * nb); AL(cacc, 6 * nb); AL(cfrc_body, 6 * nb);
  AL(qM, m->nM); AL(qLD, m->nM); AL(qLDiagInv, nv); AL(qH, m->nM); AL(qHDiagInv, nv);
  AL(ten_length, m->ntendon); AL(ten_J, m->ntendon * nv); AL(actuator_force, m->nu);
  AL(qfrc_passive, nv); AL(qfrc_bias, nv); AL(qfrc_actuator, nv); AL(qfrc_smooth, nv); AL(qacc_smooth, nv); AL(qfrc_constraint, nv);
  d->contact = (om_contact*)calloc(OM_MAXCON, sizeof(om_contact));
  d->efc_type = (int*)calloc(OM_MAXEFC, sizeof(int));
  d->efc_id = (int*)calloc(OM_MAXEFC, sizeof(int));
  AL(efc_J, OM_MAXEFC * nv); AL(efc_B, OM_MAXEFC * nv); AL(efc_pos, OM_MAXEFC); AL(efc_margin, OM_MAXEFC); AL(efc_diagApprox, OM_MAXEFC);
  AL(efc_R, OM_MAXEFC); AL(efc_D, OM_MAXEFC); AL(efc_KBIP, 4 * OM_MAXEFC); AL(efc_vel, OM_MAXEFC); AL(efc_aref, OM_MAXEFC); AL(efc_b, OM_MAXEFC);
  AL(efc_force, OM_MAXEFC);
  d->efc_AR = (double*)calloc((size_t)OM_MAXEFC * OM_MAXEFC, sizeof(double));
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  return d;
}

void om_free_data(om_data* d) {
  if (!d) return;
  double** p[] = {&d->qpos, &d->qvel, &d->ctrl, &d->qacc_warmstart, &d->qfrc_applied, &d->xfrc_applied, &d->qacc, &d->xpos, &d->xquat, &d->xmat,
                  &d->xipos, &d->ximat, &d->xanchor, &d->xaxis, &d->geom_xpos, &d->geom_xmat, &d->subtree_com, &d->cinert, &d->cdof, &d->cdof_dot,
                  &d->crb, &d->cvel, &d->cacc, &d->cfrc_body, &d->qM, &d->qLD, &d->qLDiagInv, &d->qH, &d->qHDiagInv, &d->ten_length, &d->ten_J,
                  &d->actuator_force, &d->qfrc_passive, &d->qfrc_bias, &d->qfrc_actuator, &d->qfrc_smooth, &d->qacc_smooth, &d->qfrc_constraint,
                  &d->efc_J, &d->efc_B, &d->efc_pos, &d->efc_margin, &d->efc_diagApprox, &d->efc_R, &d->efc_D, &d->efc_KBIP, &d->efc_vel,
                  &d->efc_aref, &d->efc_b, &d->efc_force, &d->efc_AR};
  for (size_t i = 0; i < sizeof p / sizeof p[0]; i++) free(*p[i]);
  free(d->contact); free(d->efc_type); free(d->efc_id);
  free(d);
}

/* mj_resetData / mj_resetDataKeyframe, mujoco.h:180,186 */
void om_reset(const om_model* m, om_data* d, int key) {
  memcpy(d->qpos, (key >= 0 && key < m->nkey) ? m->key_qpos + key * m->nq : m->qpos0, sizeof(double) * m->nq);
  memset(d->qvel, 0, sizeof(double) * m->nv);
  memset(d->ctrl, 0, sizeof(double) * m->nu);
  memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
  memset(d->qacc, 0, sizeof(double) * m->nv);
  memset(d->qfrc_applied, 0, sizeof(double) * m->nv);
  memset(d->xfrc_applied, 0, sizeof(double) * 6 * m->nbody);
  memset(d->warning, 0, sizeof d->warning);
  d->time = 0; d->ncon = d->nefc = 0;
  d->sum_ncon = d->sum_nefc = d->sum_iter = d->nstep = 0; d->max_ncon = d->max_nefc = 0;
}

/* ------------------------------------------------------------------ position stage ------- */

/* mj_kinematics, mujoco.h:310 */
static void kinematics(const om_model* m, om_data* d) {
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  memset(d->xpos, 0, 3 * sizeof(double)); memset(d->xipos, 0, 3 * sizeof(double));
  memset(d->xmat, 0, 9 * sizeof(double)); d->xmat[0] = d->xmat[4] = d->xmat[8] = 1;
  memcpy(d->ximat, d->xmat, 9 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    double pos[3], quat[4];
    int p = m->body_parentid[b];
    if (m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] == JNT_FREE) {
      int j = m->body_jntadr[b], qa = m->jnt_qposadr[j];
      memcpy(pos, d->qpos + qa, sizeof pos);
      memcpy(quat, d->qpos + qa + 3, sizeof quat);
      normalize4(quat);
      memcpy(d->xanchor + 3 * j, pos, sizeof pos);
      memcpy(d->xaxis + 3 * j, m->jnt_axis + 3 * j, sizeof pos);
    } else {
      mulmatvec3(pos, d->xmat + 9 * p, m->body_pos + 3 * b);
      for (int i = 0; i < 3; i++) pos[i] += d->xpos[3 * p + i];
      mulquat(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
      for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
        int j = m->body_jntadr[b] + jj, qa = m->jnt_qposadr[j];
        double* axis = d->xaxis + 3 * j;
        double* anchor = d->xanchor + 3 * j;
        rotvecquat(axis, m->jnt_axis + 3 * j, quat);
        rotvecquat(anchor, m->jnt_pos + 3 * j, quat);
        for (int i = 0; i < 3; i++) anchor[i] += pos[i];
        double dq = d->qpos[qa] - m->qpos0[qa];
        if (m->jnt_type[j] == JNT_SLIDE) {
          for (int i = 0; i < 3; i++) pos[i] += axis[i] * dq;
        } else { /* hinge: rotate about the local axis, then correct for the off-centre anchor */
          double ql[4], v[3];
          axisangle2quat(ql, m->jnt_axis + 3 * j, dq);
          mulquat(quat, quat, ql);
          rotvecquat(v, m->jnt_pos + 3 * j, quat);
          for (int i = 0; i < 3; i++) pos[i] = anchor[i] - v[i];
        }
      }
      normalize4(quat);
    }
    memcpy(d->xpos + 3 * b, pos, sizeof pos);
    memcpy(d->xquat + 4 * b, quat, sizeof quat);
    quat2mat(d->xmat + 9 * b, quat);
    double v[3], qi[4];
    mulmatvec3(v, d->xmat + 9 * b, m->body_ipos + 3 * b);
    for (int i = 0; i < 3; i++) d->xipos[3 * b + i] = pos[i] + v[i];
    mulquat(qi, quat, m->body_iquat + 4 * b);
    quat2mat(d->ximat + 9 * b, qi);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double v[3], q[4];
    mulmatvec3(v, d->xmat + 9 * b, m->geom_pos + 3 * g);
    for (int i = 0; i < 3; i++) d->geom_xpos[3 * g + i] = d->xpos[3 * b + i] + v[i];
    mulquat(q, d->xquat + 4 * b, m->geom_quat + 4 * g);
    quat2mat(d->geom_xmat + 9 * g, q);
  }
}

/* mj_comPos, mujoco.h:313 */
static void com_pos(const om_model* m, om_data* d) {
  int nb = m->nbody;
  for (int b = 0; b < nb; b++) for (int i = 0; i < 3; i++) d->subtree_com[3 * b + i] = m->body_mass[b] * d->xipos[3 * b + i];
  for (int b = nb - 1; b > 0; b--) for (int i = 0; i < 3; i++) d->subtree_com[3 * m->body_parentid[b] + i] += d->subtree_com[3 * b + i];
  for (int b = 0; b < nb; b++) {
    if (m->body_subtreemass[b] < MINVAL) memcpy(d->subtree_com + 3 * b, d->xipos + 3 * b, 3 * sizeof(double));
    else for (int i = 0; i < 3; i++) d->subtree_com[3 * b + i] /= m->body_subtreemass[b];
  }
  memset(d->cinert, 0, 10 * sizeof(double));
  for (int b = 1; b < nb; b++) {
    const double* com = d->subtree_com + 3 * m->body_rootid[b];
    double dif[3] = {d->xipos[3 * b] - com[0], d->xipos[3 * b + 1] - com[1], d->xipos[3 * b + 2] - com[2]};
    const double* mat = d->ximat + 9 * b;
    const double* in = m->body_inertia + 3 * b;
    double mass = m->body_mass[b], t[9];
    double* res = d->cinert + 10 * b;
    for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) t[3 * r + k] = mat[3 * r + k] * in[k];
    res[0] = t[0] * mat[0] + t[1] * mat[1] + t[2] * mat[2] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
    res[1] = t[3] * mat[3] + t[4] * mat[4] + t[5] * mat[5] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
    res[2] = t[6] * mat[6] + t[7] * mat[7] + t[8] * mat[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
    res[3] = t[0] * mat[3] + t[1] * mat[4] + t[2] * mat[5] - mass * dif[0] * dif[1];
    res[4] = t[0] * mat[6] + t[1] * mat[7] + t[2] * mat[8] - mass * dif[0] * dif[2];
    res[5] = t[3] * mat[6] + t[4] * mat[7] + t[5] * mat[8] - mass * dif[1] * dif[2];
    res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2]; res[9] = mass;
  }
  memset(d->cdof, 0, 6 * m->nv * sizeof(double));
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const double* com = d->subtree_com + 3 * m->body_rootid[b];
    double off[3] = {com[0] - d->xanchor[3 * j], com[1] - d->xanchor[3 * j + 1], com[2] - d->xanchor[3 * j + 2]};
    if (m->jnt_type[j] == JNT_FREE) {
      for (int i = 0; i < 3; i++) d->cdof[6 * (da + i) + 3 + i] = 1;
      for (int i = 0; i < 3; i++) { /* rotations about the body axes (columns of xmat) */
        double ax[3] = {d->xmat[9 * b + i], d->xmat[9 * b + 3 + i], d->xmat[9 * b + 6 + i]};
        double* cd = d->cdof + 6 * (da + 3 + i);
        memcpy(cd, ax, sizeof ax);
        cross3(cd + 3, ax, off);
      }
    } else if (m->jnt_type[j] == JNT_SLIDE) {
      memcpy(d->cdof + 6 * da + 3, d->xaxis + 3 * j, 3 * sizeof(double));
    } else {
      memcpy(d->cdof + 6 * da, d->xaxis + 3 * j, 3 * sizeof(double));
      cross3(d->cdof + 6 * da + 3, d->xaxis + 3 * j, off);
    }
  }
}

/* mj_tendon (fixed tendons only), mujoco.h:322 */
static void tendon(const om_model* m, om_data* d) {
  memset(d->ten_J, 0, sizeof(double) * m->ntendon * m->nv);
  for (int t = 0; t < m->ntendon; t++) {
    d->ten_length[t] = 0;
    for (int w = 0; w < m->tendon_num[t]; w++) {
      int j = m->wrap_objid[m->tendon_adr[t] + w];
      double coef = m->wrap_prm[m->tendon_adr[t] + w];
      d->ten_length[t] += coef * d->qpos[m->jnt_qposadr[j]];
      d->ten_J[t * m->nv + m->jnt_dofadr[j]] = coef;
    }
  }
}

/* mj_crb, mujoco.h:328 */
static void crb(const om_model* m, om_data* d) {
  memcpy(d->crb, d->cinert, sizeof(double) * 10 * m->nbody);
  for (int b = m->nbody - 1; b > 0; b--) { int p = m->body_parentid[b]; if (p > 0) for (int i = 0; i < 10; i++) d->crb[10 * p + i] += d->crb[10 * b + i]; }
  memset(d->qM, 0, sizeof(double) * m->nM);
  for (int i = 0; i < m->nv; i++) {
    int adr = m->dof_Madr[i];
    double buf[6];
    d->qM[adr] = m->dof_armature[i];
    mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double s = 0;
      for (int t = 0; t < 6; t++) s += d->cdof[6 * j + t] * buf[t];
      d->qM[adr++] += s;
    }
  }
}

/* mj_factorM (L^T D L on the ancestor-chain layout), mujoco.h:331 */
static void factor_i(const om_model* m, const double* M, double* LD, double* DiagInv) {
  int nv = m->nv;
  if (LD != M) memcpy(LD, M, sizeof(double) * m->nM);
  for (int k = nv - 1; k >= 0; k--) {
    int Mkk = m->dof_Madr[k], Mki = Mkk + 1, i = m->dof_parentid[k];
    if (LD[Mkk] < MINVAL) LD[Mkk] = MINVAL;
    while (i >= 0) {
      double tmp = LD[Mki] / LD[Mkk];
      int cnt = (i < nv - 1 ? m->dof_Madr[i + 1] : m->nM) - m->dof_Madr[i];
      for (int t = 0; t < cnt; t++) LD[m->dof_Madr[i] + t] -= LD[Mki + t] * tmp;
      LD[Mki] = tmp;
      i = m->dof_parentid[i];
      Mki++;
    }
  }
  for (int i = 0; i < nv; i++) DiagInv[i] = 1.0 / LD[m->dof_Madr[i]];
}

/* mj_solveM, mujoco.h:334 */
static void solve_ld(const om_model* m, double* x, const double* LD, const double* DiagInv) {
  int nv = m->nv;
  for (int k = nv - 1; k >= 0; k--) {
    if (x[k] == 0) continue;
    int Mki = m->dof_Madr[k] + 1, i = m->dof_parentid[k];
    while (i >= 0) { x[i] -= LD[Mki] * x[k]; Mki++; i = m->dof_parentid[i]; }
  }
  for (int i = 0; i < nv; i++) x[i] *= DiagInv[i];
  for (int k = 0; k < nv; k++) {
    int Mki = m->dof_Madr[k] + 1, i = m->dof_parentid[k];
    while (i >= 0) { x[k] -= LD[Mki] * x[i]; Mki++; i = m->dof_parentid[i]; }
  }
}

/* mj_jac, mujoco.h:421: jacp/jacr are 3 x nv row-major; may be NULL */
static void jac(const om_model* m, const om_data* d, double* jacp, double* jacr, const double* point, int body) {
  int nv = m->nv;
  if (jacp) memset(jacp, 0, sizeof(double) * 3 * nv);
  if (jacr) memset(jacr, 0, sizeof(double) * 3 * nv);
  const double* com = d->subtree_com + 3 * m->body_rootid[body];
  double off[3] = {point[0] - com[0], point[1] - com[1], point[2] - com[2]};
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body == 0) return;
  int i = m->body_dofadr[body] + m->body_dofnum[body] - 1;
  while (i >= 0) {
    const double* cd = d->cdof + 6 * i;
    double c[3];
    cross3(c, cd, off);
    for (int r = 0; r < 3; r++) {
      if (jacr) jacr[r * nv + i] = cd[r];
      if (jacp) jacp[r * nv + i] = cd[3 + r] + c[r];
    }
    i = m->dof_parentid[i];
  }
}

/* ------------------------------------------------------------------ collision ------------ */

/* mju_makeFrame: complete a contact frame whose first row is the normal */
static void make_frame(double* f) {
  normalize3(f);
  if (sqrt(dot3(f + 3, f + 3)) < 0.5) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double dd = dot3(f, f + 3);
  for (int i = 0; i < 3; i++) f[3 + i] -= dd * f[i];
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}

/* plane (pos1, normal) vs sphere — engine_collision_primitive mjc_PlaneSphere [recall] */
static int plane_sphere(om_contact* c, double margin, const double* pos1, const double* normal, const double* pos2, double radius) {
  double tmp[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
  double cdist = dot3(tmp, normal);
  if (cdist > margin + radius) return 0;
  c->dist = cdist - radius;
  for (int i = 0; i < 3; i++) c->pos[i] = pos2[i] + normal[i] * (-c->dist / 2 - radius);
  memcpy(c->frame, normal, 3 * sizeof(double));
  memset(c->frame + 3, 0, 6 * sizeof(double));
  return 1;
}
static int sphere_sphere(om_contact* c, double margin, const double* pos1, double r1, const double* pos2, double r2) {
  double dif[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
  double cdist = sqrt(dot3(dif, dif));
  if (cdist > margin + r1 + r2) return 0;
  c->dist = cdist - r1 - r2;
  if (cdist < MINVAL) { dif[0] = 1; dif[1] = dif[2] = 0; }
  else for (int i = 0; i < 3; i++) dif[i] /= cdist;
  for (int i = 0; i < 3; i++) c->pos[i] = pos1[i] + dif[i] * (r1 + c->dist / 2);
  memcpy(c->frame, dif, sizeof dif);
  memset(c->frame + 3, 0, 6 * sizeof(double));
  return 1;
}
static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static int capsule_capsule(om_contact* c, double margin, const double* pos1, const double* axis1, double r1, double len1, const double* pos2,
                           const double* axis2, double r2, double len2) {
  double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  double ma = dot3(axis1, axis1), mb = -dot3(axis1, axis2), mc = dot3(axis2, axis2);
  double u = -dot3(axis1, dif), v = dot3(axis2, dif);
  double det = ma * mc - mb * mb;
  double x1, x2, v1[3], v2[3];
  if (fabs(det) >= MINVAL) {
    x1 = (mc * u - mb * v) / det;
    x2 = (ma * v - mb * u) / det;
    if (x1 > len1) { x1 = len1; x2 = (v - mb * len1) / mc; }
    else if (x1 < -len1) { x1 = -len1; x2 = (v + mb * len1) / mc; }
    if (x2 > len2) { x2 = len2; x1 = clipd((u - mb * len2) / ma, -len1, len1); }
    else if (x2 < -len2) { x2 = -len2; x1 = clipd((u + mb * len2) / ma, -len1, len1); }
    for (int i = 0; i < 3; i++) { v1[i] = pos1[i] + axis1[i] * x1; v2[i] = pos2[i] + axis2[i] * x2; }
    return sphere_sphere(c, margin, v1, r1, v2, r2);
  }
  /* parallel axes: up to two contacts from the segment ends */
  int n = 0;
  for (int i = 0; i < 3; i++) v1[i] = pos1[i] + axis1[i] * len1;
  x2 = clipd((v - mb * len1) / mc, -len2, len2);
  for (int i = 0; i < 3; i++) v2[i] = pos2[i] + axis2[i] * x2;
  n += sphere_sphere(c + n, margin, v1, r1, v2, r2);
  for (int i = 0; i < 3; i++) v1[i] = pos1[i] - axis1[i] * len1;
  x2 = clipd((v + mb * len1) / mc, -len2, len2);
  for (int i = 0; i < 3; i++) v2[i] = pos2[i] + axis2[i] * x2;
  n += sphere_sphere(c + n, margin, v1, r1, v2, r2);
  if (n >= 2) return n;
  for (int i = 0; i < 3; i++) v2[i] = pos2[i] + axis2[i] * len2;
  x1 = clipd((u - mb * len2) / ma, -len1, len1);
  for (int i = 0; i < 3; i++) v1[i] = pos1[i] + axis1[i] * x1;
  n += sphere_sphere(c + n, margin, v1, r1, v2, r2);
  if (n >= 2) return n;
  for (int i = 0; i < 3; i++) v2[i] = pos2[i] - axis2[i] * len2;
  x1 = clipd((u + mb * len2) / ma, -len1, len1);
  for (int i = 0; i < 3; i++) v1[i] = pos1[i] + axis1[i] * x1;
  n += sphere_sphere(c + n, margin, v1, r1, v2, r2);
  return n;
}

/* ------------------------------------------------------------------ convex collision (libccd MPR) ----
 * MuJoCo collides mesh geoms (through their convex hulls), and ANY geom against a height field, with libccd's Minkowski
 * Portal Refinement: engine_collision_convex.c (mjc_Convex, mjc_ConvexHField; declared through mj_collision, mujoco.h:355)
 * over third-party libccd (src/mpr.c: ccdMPRPenetration).  Neither source is in /root/reference (MuJoCo 3.1.4 is fetched at
 * configure time, mujoco_mpc/CMakeLists.txt:58-88; libccd is one of ITS dependencies), so this is a restatement of the
 * published algorithm (G. Snethen, "XenoCollide", Game Programming Gems 7; libccd mpr.c) [recall]: same portal discovery,
 * refinement, expansion rule, tolerance test and penetration read-out, with libccd's double-precision epsilon. */

#define CCD_EPS 2.220446049250313e-16 /* DBL_EPSILON: libccd built in double precision */
static int ccd_is_zero(double x) { return fabs(x) < CCD_EPS; }
static int ccd_eq(double a, double b) {
  double ab = fabs(a - b);
  if (ab < CCD_EPS) return 1;
  a = fabs(a); b = fabs(b);
  return b > a ? ab < CCD_EPS * b : ab < CCD_EPS * a;
}
static int ccd_vec_eq(const double* a, const double* b) { return ccd_eq(a[0], b[0]) && ccd_eq(a[1], b[1]) && ccd_eq(a[2], b[2]); }

/* one collision object in the frame the test runs in: a geom (sphere / capsule / mesh hull) or a height-field prism */
typedef struct {
  int type;                 /* GEOM_SPHERE, GEOM_CAPSULE, GEOM_MESH, or -1: prism */
  double pos[3], mat[9];    /* frame of the geom (row-major rotation) */
  double size[3];
  const double* vert;       /* mesh: hull vertices in the geom frame */
  int nvert;
  const int *nbradr, *nbrnum, *nbr; /* mesh: the hull's edge graph (NULL: exhaustive search) */
  int cur;                  /* mesh: vertex the last support call ended on (the next climb starts there; 0 at the start of a test) */
  double margin;            /* mjccd_support inflates the shape by this much along the direction */
  double prism[6][3];       /* prism: bottom triangle 0..2, top triangle 3..5 */
} ccd_obj;

/* mjccd_center / prism_center */
static void ccd_center(const ccd_obj* o, double* c) {
  if (o->type < 0) {
    c[0] = c[1] = c[2] = 0;
    for (int i = 0; i < 6; i++) for (int k = 0; k < 3; k++) c[k] += o->prism[i][k];
    for (int k = 0; k < 3; k++) c[k] /= 6.0;
  } else memcpy(c, o->pos, 3 * sizeof(double));
}
/* diagnostic counters of the narrowphase (single-threaded use: tools/mpr_stats.py): tests, support calls on meshes, climb rounds
 * (one evaluation of all neighbours of the current vertex), neighbour evaluations, portal iterations */
static _Thread_local long long om_stat[8]; /* per thread: the rollout threads of om_rollout_threads do not share them */
void om_mpr_stats(long long* out, int reset) { for (int i = 0; i < 8; i++) { out[i] = om_stat[i]; if (reset) om_stat[i] = 0; } }
/* mjccd_support / prism_support: the point of the object farthest along dir (dir is unit: every caller in mpr.c normalises) */
static void ccd_support(ccd_obj* o, const double* dir, double* out) {
  if (o->type < 0) {
    int best = 0;
    double bd = -1e300;
    for (int i = 0; i < 6; i++) { double v = dot3(o->prism[i], dir); if (v > bd) { bd = v; best = i; } }
    memcpy(out, o->prism[best], 3 * sizeof(double));
    return;
  }
  double ld[3], res[3];
  for (int i = 0; i < 3; i++) ld[i] = o->mat[i] * dir[0] + o->mat[3 + i] * dir[1] + o->mat[6 + i] * dir[2];  /* mat' dir */
  if (o->type == GEOM_SPHERE) { for (int i = 0; i < 3; i++) res[i] = ld[i] * o->size[0]; }
  else if (o->type == GEOM_CAPSULE) {
    for (int i = 0; i < 3; i++) res[i] = ld[i] * o->size[0];
    res[2] += ld[2] >= 0 ? o->size[1] : -o->size[1];
  } else if (o->nbr) {
    /* mesh: steepest ascent along the hull's edges from the vertex the previous call of this test ended on (MuJoCo hill-climbs
     * on its mesh graph too; on a convex polytope a vertex with no better neighbour is a maximiser).  Every neighbour is
     * evaluated, the best one taken if it is strictly better, ties to the first in the list. */
    /* Exact ties (a direction perpendicular to a flat facet, e.g. an axis direction on a CAD part) are broken by a second, generic
     * direction, as if the direction were ld + epsilon tie: without it a climb that starts on the facet opposite the maximum sees
     * only equal neighbours and stalls there. */
    static const double tie[3] = {0.41421356237309503, 0.7320508075688772, 1.0};
    int cur = o->cur;
    double bd = dot3(o->vert + 3 * cur, ld), bt = dot3(o->vert + 3 * cur, tie);
    om_stat[1]++;
    for (;;) {
      int best = cur;
      const int* nb = o->nbr + o->nbradr[cur];
      om_stat[2]++; om_stat[3] += o->nbrnum[cur];
      for (int i = 0; i < o->nbrnum[cur]; i++) {
        double v = dot3(o->vert + 3 * nb[i], ld), t = dot3(o->vert + 3 * nb[i], tie);
        if (v > bd || (v == bd && t > bt)) { bd = v; bt = t; best = nb[i]; }
      }
      if (best == cur) break;
      cur = best;
    }
    o->cur = cur;
    memcpy(res, o->vert + 3 * cur, sizeof res);
  } else { /* mesh without a graph (test hook): exhaustive search over the hull vertices, the same maximiser */
    int best = 0;
    double bd = -1e300;
    for (int i = 0; i < o->nvert; i++) { double v = dot3(o->vert + 3 * i, ld); if (v > bd) { bd = v; best = i; } }
    memcpy(res, o->vert + 3 * best, sizeof res);
  }
  for (int i = 0; i < 3; i++) res[i] += ld[i] * o->margin;
  for (int i = 0; i < 3; i++) out[i] = o->mat[3 * i] * res[0] + o->mat[3 * i + 1] * res[1] + o->mat[3 * i + 2] * res[2] + o->pos[i];
}

typedef struct { double v[3], v1[3], v2[3]; } ccd_sup;  /* a point of the Minkowski difference obj1 - obj2 and its two witnesses */
static void mpr_support(ccd_obj* o1, ccd_obj* o2, const double* dir, ccd_sup* s) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  om_stat[4]++;
  ccd_support(o1, dir, s->v1);
  ccd_support(o2, nd, s->v2);
  for (int i = 0; i < 3; i++) s->v[i] = s->v1[i] - s->v2[i];
}
static void vsub(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static void mpr_portal_dir(const ccd_sup* P, double* dir) {
  double a[3], b[3];
  vsub(a, P[2].v, P[1].v); vsub(b, P[3].v, P[1].v);
  cross3(dir, a, b);
  normalize3(dir);
}
static int mpr_reach_tolerance(const ccd_sup* P, const ccd_sup* v4, const double* dir, double tol) {
  double dv1 = dot3(P[1].v, dir), dv2 = dot3(P[2].v, dir), dv3 = dot3(P[3].v, dir), dv4 = dot3(v4->v, dir);
  double d = fmin(fmin(dv4 - dv1, dv4 - dv2), dv4 - dv3);
  return ccd_eq(d, tol) || d < tol;
}
static void mpr_expand_portal(ccd_sup* P, const ccd_sup* v4) {
  double v4v0[3];
  cross3(v4v0, v4->v, P[0].v);
  if (dot3(P[1].v, v4v0) > 0) { if (dot3(P[2].v, v4v0) > 0) P[1] = *v4; else P[3] = *v4; }
  else { if (dot3(P[3].v, v4v0) > 0) P[2] = *v4; else P[1] = *v4; }
}
/* squared distance of the origin from triangle (a, b, c) and the closest point (ccdVec3PointTriDist2 with P = origin) */
/* closest point of triangle abc to p (Ericson, Real-Time Collision Detection 5.1.5): stands in for libccd's ccdVec3PointTriDist2 */
static void closest_on_triangle(double* out, const double* p, const double* a, const double* b, const double* c) {
  double ab[3], ac[3], ap[3];
  for (int i = 0; i < 3; i++) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { memcpy(out, a, 3 * sizeof(double)); return; }
  double bp[3] = {p[0] - b[0], p[1] - b[1], p[2] - b[2]};
  double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { memcpy(out, b, 3 * sizeof(double)); return; }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); for (int i = 0; i < 3; i++) out[i] = a[i] + v * ab[i]; return; }
  double cp[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
  double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { memcpy(out, c, 3 * sizeof(double)); return; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double w = d2 / (d2 - d6); for (int i = 0; i < 3; i++) out[i] = a[i] + w * ac[i]; return; }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); for (int i = 0; i < 3; i++) out[i] = b[i] + w * (c[i] - b[i]); return; }
  double denom = 1.0 / (va + vb + vc), v = vb * denom, w = vc * denom;
  for (int i = 0; i < 3; i++) out[i] = a[i] + ab[i] * v + ac[i] * w;
}
static double origin_tri_dist2(const double* a, const double* b, const double* c, double* witness) {
  const double o[3] = {0, 0, 0};
  closest_on_triangle(witness, o, a, b, c);
  return dot3(witness, witness);
}

/* ccdMPRPenetration: 0 and (depth, dir, pos) when the objects intersect, -1 otherwise.  dir points from obj1 into obj2. */
static int mpr_penetration_core(ccd_obj* o1, ccd_obj* o2, int max_iterations, double tolerance, double* depth, double* pdir, double* pos) {
  ccd_sup P[4];
  double dir[3], va[3], vb[3];
  const double origin[3] = {0, 0, 0};
  /* ---- discoverPortal */
  om_stat[0]++;
  ccd_center(o1, P[0].v1); ccd_center(o2, P[0].v2);
  vsub(P[0].v, P[0].v1, P[0].v2);
  if (ccd_vec_eq(P[0].v, origin)) P[0].v[0] += CCD_EPS * 10;  /* centres coincide: nudge */
  for (int i = 0; i < 3; i++) dir[i] = -P[0].v[i];
  normalize3(dir);
  mpr_support(o1, o2, dir, &P[1]);
  double dt = dot3(P[1].v, dir);
  if (ccd_is_zero(dt) || dt < 0) return -1;
  cross3(dir, P[0].v, P[1].v);
  if (ccd_is_zero(dot3(dir, dir))) {
    if (ccd_vec_eq(P[1].v, origin)) { /* origin lies on v1: touching contact (findPenetrTouch) */
      *depth = 0; pdir[0] = pdir[1] = pdir[2] = 0;
      for (int i = 0; i < 3; i++) pos[i] = 0.5 * (P[1].v1[i] + P[1].v2[i]);
      return 0;
    }
    /* origin lies on the v0-v1 segment (findPenetrSegment) */
    for (int i = 0; i < 3; i++) { pos[i] = 0.5 * (P[1].v1[i] + P[1].v2[i]); pdir[i] = P[1].v[i]; }
    *depth = normalize3(pdir);
    return 0;
  }
  normalize3(dir);
  mpr_support(o1, o2, dir, &P[2]);
  dt = dot3(P[2].v, dir);
  if (ccd_is_zero(dt) || dt < 0) return -1;
  vsub(va, P[1].v, P[0].v); vsub(vb, P[2].v, P[0].v);
  cross3(dir, va, vb);
  normalize3(dir);
  if (dot3(dir, P[0].v) > 0) { ccd_sup t = P[1]; P[1] = P[2]; P[2] = t; for (int i = 0; i < 3; i++) dir[i] = -dir[i]; }  /* portal faces oriented away from the origin */
  for (int guard = 0; ; guard++) {
    if (guard > 1000) return -1;  /* (libccd loops until the portal closes; this bound is never reached on convex input) */
    mpr_support(o1, o2, dir, &P[3]);
    dt = dot3(P[3].v, dir);
    if (ccd_is_zero(dt) || dt < 0) return -1;
    int cont = 0;
    cross3(va, P[1].v, P[3].v);  /* origin outside (v1, v0, v3): v3 replaces v2 */
    dt = dot3(va, P[0].v);
    if (dt < 0 && !ccd_is_zero(dt)) { P[2] = P[3]; cont = 1; }
    if (!cont) {
      cross3(va, P[3].v, P[2].v);  /* origin outside (v3, v0, v2): v3 replaces v1 */
      dt = dot3(va, P[0].v);
      if (dt < 0 && !ccd_is_zero(dt)) { P[1] = P[3]; cont = 1; }
    }
    if (!cont) break;
    vsub(va, P[1].v, P[0].v); vsub(vb, P[2].v, P[0].v);
    cross3(dir, va, vb);
    normalize3(dir);
  }
  /* ---- refinePortal */
  ccd_sup v4;
  for (int guard = 0; ; guard++) {
    if (guard > 1000) return -1;
    mpr_portal_dir(P, dir);
    dt = dot3(dir, P[1].v);
    if (ccd_is_zero(dt) || dt > 0) break;  /* the portal encapsulates the origin */
    mpr_support(o1, o2, dir, &v4);
    dt = dot3(v4.v, dir);
    if (!(ccd_is_zero(dt) || dt > 0) || mpr_reach_tolerance(P, &v4, dir, tolerance)) return -1;  /* cannot reach the origin: no intersection */
    mpr_expand_portal(P, &v4);
  }
  /* ---- findPenetr */
  for (int it = 0; ; it++) {
    mpr_portal_dir(P, dir);
    mpr_support(o1, o2, dir, &v4);
    if (mpr_reach_tolerance(P, &v4, dir, tolerance) || it > max_iterations) {
      *depth = sqrt(origin_tri_dist2(P[1].v, P[2].v, P[3].v, pdir));
      if (ccd_is_zero(*depth)) { pdir[0] = pdir[1] = pdir[2] = 0; }
      else normalize3(pdir);
      /* findPos: barycentric coordinates of the origin in the portal tetrahedron */
      double b[4], t[3];
      mpr_portal_dir(P, dir);
      cross3(t, P[1].v, P[2].v); b[0] = dot3(t, P[3].v);
      cross3(t, P[3].v, P[2].v); b[1] = dot3(t, P[0].v);
      cross3(t, P[0].v, P[1].v); b[2] = dot3(t, P[3].v);
      cross3(t, P[2].v, P[1].v); b[3] = dot3(t, P[0].v);
      double sum = b[0] + b[1] + b[2] + b[3];
      if (ccd_is_zero(sum) || sum < 0) {
        b[0] = 0;
        cross3(t, P[2].v, P[3].v); b[1] = dot3(t, dir);
        cross3(t, P[3].v, P[1].v); b[2] = dot3(t, dir);
        cross3(t, P[1].v, P[2].v); b[3] = dot3(t, dir);
        sum = b[1] + b[2] + b[3];
      }
      double inv = 1.0 / sum;
      for (int k = 0; k < 3; k++) {
        double p1 = 0, p2 = 0;
        for (int i = 0; i < 4; i++) { p1 += b[i] * P[i].v1[k]; p2 += b[i] * P[i].v2[k]; }
        pos[k] = 0.5 * (p1 + p2) * inv;
      }
      return 0;
    }
    mpr_expand_portal(P, &v4);
  }
}

static void ccd_obj_from_geom(const om_model* m, const om_data* d, int g, double margin, ccd_obj* o) {
  o->type = m->geom_type[g];
  memcpy(o->pos, d->geom_xpos + 3 * g, sizeof o->pos);
  memcpy(o->mat, d->geom_xmat + 9 * g, sizeof o->mat);
  memcpy(o->size, m->geom_size + 3 * g, sizeof o->size);
  o->vert = NULL; o->nvert = 0; o->margin = margin; o->nbr = NULL; o->nbradr = o->nbrnum = NULL; o->cur = 0;
  if (o->type == GEOM_MESH) {
    int k = m->geom_dataid[g];
    o->vert = m->mesh_vert + 3 * m->mesh_vertadr[k]; o->nvert = m->mesh_vertnum[k];
    if (m->nmeshnbr > 0) { o->nbradr = m->mesh_nbradr + m->mesh_vertadr[k]; o->nbrnum = m->mesh_nbrnum + m->mesh_vertadr[k]; o->nbr = m->mesh_nbr; }
  }
}

/* mjc_fixNormal [recall]: a sphere or capsule knows its own surface normal at the contact point; a mesh or prism does not.
 * The contact normal (pointing from geom1 to geom2) is replaced by the analytic one where a geom has it, averaged if both do. */
static void fix_normal(const om_model* m, const om_data* d, om_contact* c, int g1, int g2) {
  double nrm[2][3];
  int have[2] = {0, 0}, gid[2] = {g1, g2};
  for (int i = 0; i < 2; i++) {
    int g = gid[i], t = m->geom_type[g];
    if (t != GEOM_SPHERE && t != GEOM_CAPSULE) continue;
    const double *xp = d->geom_xpos + 3 * g, *mat = d->geom_xmat + 9 * g;
    double dif[3] = {c->pos[0] - xp[0], c->pos[1] - xp[1], c->pos[2] - xp[2]}, lp[3];
    for (int k = 0; k < 3; k++) lp[k] = mat[k] * dif[0] + mat[3 + k] * dif[1] + mat[6 + k] * dif[2];
    if (t == GEOM_CAPSULE) { double h = m->geom_size[3 * g + 1]; lp[2] = lp[2] > h ? lp[2] - h : (lp[2] < -h ? lp[2] + h : 0); }
    if (normalize3(lp) < MINVAL) continue;
    for (int k = 0; k < 3; k++) nrm[i][k] = (mat[3 * k] * lp[0] + mat[3 * k + 1] * lp[1] + mat[3 * k + 2] * lp[2]) * (i == 0 ? 1.0 : -1.0);  /* outward of geom1, inward of geom2 */
    have[i] = 1;
  }
  if (!have[0] && !have[1]) return;
  double n[3];
  for (int k = 0; k < 3; k++) n[k] = (have[0] ? nrm[0][k] : 0) + (have[1] ? nrm[1][k] : 0);
  if (normalize3(n) < MINVAL) return;
  memcpy(c->frame, n, sizeof n);
}

/* mjc_Convex: two convex geoms (at least one of them a mesh here), one contact (multiccd is off by default, mjmodel.h:75) */
/* (diagnostics: om_stat[5] = most support calls of one test, om_stat[6] = tests with more than 16, om_stat[7] = tests that hit) */
static int mpr_penetration(ccd_obj* o1, ccd_obj* o2, int max_iterations, double tolerance, double* depth, double* pdir, double* pos) {
  const long long before = om_stat[4];
  const int rc = mpr_penetration_core(o1, o2, max_iterations, tolerance, depth, pdir, pos);
  const long long used = om_stat[4] - before;
  if (used > om_stat[5]) om_stat[5] = used;
  if (used > 16) om_stat[6]++;
  if (rc == 0) om_stat[7]++;
  return rc;
}
static int convex_convex(const om_model* m, const om_data* d, om_contact* con, int g1, int g2, double margin) {
  ccd_obj o1, o2;
  ccd_obj_from_geom(m, d, g1, 0.5 * margin, &o1);
  ccd_obj_from_geom(m, d, g2, 0.5 * margin, &o2);
  double depth, dir[3], pos[3];
  if (mpr_penetration(&o1, &o2, m->mpr_iterations, m->mpr_tolerance, &depth, dir, pos) != 0) return 0;
  if (dir[0] == 0 && dir[1] == 0 && dir[2] == 0) return 0;  /* contact found but its normal is undefined */
  con->dist = margin - depth;
  memcpy(con->frame, dir, sizeof dir);
  memcpy(con->pos, pos, sizeof pos);
  memset(con->frame + 3, 0, 6 * sizeof(double));
  fix_normal(m, d, con, g1, g2);
  return 1;
}

/* mjc_PlaneConvex [recall] (engine_collision_convex.c; mujoco.h:355 mj_collision): plane g1 against mesh g2 (through its hull).  The hull's
 * support point along the plane's inward direction -normal is the first contact if it is within the margin; up to three more come
 * from the hull vertices ADJACENT to it in the mesh graph, in the order of the graph, each within the margin and at least
 * tolplanemesh x rbound away from the first point (so that a face lying on the plane is held at several corners instead of one).
 * dist = signed distance of the vertex from the plane, pos = the vertex moved half of that back along the normal, frame normal = the
 * plane's.  Recalled constants: tolplanemesh 0.3, at most 3 extra contacts.  The graph is this build's own hull, so
 * which neighbours exist - and their order - is a property of that hull. */
#define OM_TOLPLANEMESH 0.3
static int plane_convex(const om_model* m, const om_data* d, om_contact* con, int g1, int g2, double margin) {
  const double *pos1 = d->geom_xpos + 3 * g1, *mat1 = d->geom_xmat + 9 * g1;
  const double normal[3] = {mat1[2], mat1[5], mat1[8]}, down[3] = {-normal[0], -normal[1], -normal[2]};
  ccd_obj o;
  ccd_obj_from_geom(m, d, g2, 0.0, &o);
  double first[3];
  ccd_support(&o, down, first);
  double dif[3] = {first[0] - pos1[0], first[1] - pos1[1], first[2] - pos1[2]};
  double dist = dot3(dif, normal);
  if (dist > margin) return 0;
  int n = 0;
  con[n].dist = dist;
  for (int k = 0; k < 3; k++) { con[n].pos[k] = first[k] - 0.5 * dist * normal[k]; con[n].frame[k] = normal[k]; }
  memset(con[n].frame + 3, 0, 6 * sizeof(double));
  n++;
  if (!o.nbr) return n;
  const double tol = OM_TOLPLANEMESH * m->geom_rbound[g2];
  const int* nb = o.nbr + o.nbradr[o.cur];
  for (int i = 0; i < o.nbrnum[o.cur] && n < 4; i++) {
    const double* v = o.vert + 3 * nb[i];
    double pnt[3];
    for (int k = 0; k < 3; k++) pnt[k] = o.mat[3 * k] * v[0] + o.mat[3 * k + 1] * v[1] + o.mat[3 * k + 2] * v[2] + o.pos[k];
    double dp[3] = {pnt[0] - pos1[0], pnt[1] - pos1[1], pnt[2] - pos1[2]}, df[3] = {pnt[0] - first[0], pnt[1] - first[1], pnt[2] - first[2]};
    const double di = dot3(dp, normal);
    if (di > margin || sqrt(dot3(df, df)) < tol) continue;
    con[n].dist = di;
    for (int k = 0; k < 3; k++) { con[n].pos[k] = pnt[k] - 0.5 * di * normal[k]; con[n].frame[k] = normal[k]; }
    memset(con[n].frame + 3, 0, 6 * sizeof(double));
    n++;
  }
  return n;
}

/* mjc_ConvexHField: geom g2 (sphere, capsule or mesh) against height field g1.  Everything runs in the field's frame: the
 * geom's bounding box there picks a sub-grid; every grid cell of it is two triangular prisms (from the field's base up to the
 * surface triangle); each prism that reaches the geom's height is tested with MPR and gives at most one contact. */
#define OM_MAXCONPAIR 50 /* mjMAXCONPAIR, mjmodel.h:28 */
static int convex_hfield(const om_model* m, const om_data* d, om_contact* con, int maxcon, int g1, int g2, double margin) {
  const double *pos1 = d->geom_xpos + 3 * g1, *mat1 = d->geom_xmat + 9 * g1;
  int hid = m->geom_dataid[g1];
  const double* size1 = m->hfield_size + 4 * hid;
  int nrow = m->hfield_nrow[hid], ncol = m->hfield_ncol[hid];
  const double* data = m->hfield_data + m->hfield_adr[hid];
  ccd_obj o2, prism;
  ccd_obj_from_geom(m, d, g2, 0, &o2);
  /* geom2 in the field's frame */
  double dif[3] = {o2.pos[0] - pos1[0], o2.pos[1] - pos1[1], o2.pos[2] - pos1[2]}, pos[3], mat[9];
  for (int i = 0; i < 3; i++) pos[i] = mat1[i] * dif[0] + mat1[3 + i] * dif[1] + mat1[6 + i] * dif[2];
  double r2 = m->geom_rbound[g2];
  for (int i = 0; i < 2; i++) if (size1[i] < pos[i] - r2 - margin || -size1[i] > pos[i] + r2 + margin) return 0;  /* box-sphere test */
  if (size1[2] < pos[2] - r2 - margin || -size1[3] > pos[2] + r2 + margin) return 0;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += mat1[3 * k + i] * o2.mat[3 * k + j]; mat[3 * i + j] = s; }  /* mat1' mat2 */
  memcpy(o2.pos, pos, sizeof pos); memcpy(o2.mat, mat, sizeof mat);
  /* bounding box of geom2 in the field's frame from its support points */
  double lo[3], hi[3], sp[3];
  for (int k = 0; k < 3; k++) {
    double dir[3] = {0, 0, 0};
    dir[k] = 1; ccd_support(&o2, dir, sp); hi[k] = sp[k];
    dir[k] = -1; ccd_support(&o2, dir, sp); lo[k] = sp[k];
  }
  if (lo[0] - margin > size1[0] || hi[0] + margin < -size1[0] || lo[1] - margin > size1[1] || hi[1] + margin < -size1[1] ||
      lo[2] - margin > size1[2] || hi[2] + margin < -size1[3]) return 0;  /* box-box test */
  int cmin = (int)floor((lo[0] + size1[0]) / (2 * size1[0]) * (ncol - 1)), cmax = (int)ceil((hi[0] + size1[0]) / (2 * size1[0]) * (ncol - 1));
  int rmin = (int)floor((lo[1] + size1[1]) / (2 * size1[1]) * (nrow - 1)), rmax = (int)ceil((hi[1] + size1[1]) / (2 * size1[1]) * (nrow - 1));
  if (cmin < 0) cmin = 0; if (rmin < 0) rmin = 0; if (cmax > ncol - 1) cmax = ncol - 1; if (rmax > nrow - 1) rmax = nrow - 1;
  o2.margin = margin;  /* (the prism tops are raised by the margin as well: mjc_ConvexHField [recall]) */
  double dx = 2.0 * size1[0] / (ncol - 1), dy = 2.0 * size1[1] / (nrow - 1);
  const int dr[2] = {1, 0};  /* triangulation direction: the strip visits (r+1, c) before (r, c) [recall] */
  memset(&prism, 0, sizeof prism);
  prism.type = -1;
  prism.prism[0][2] = prism.prism[1][2] = prism.prism[2][2] = -size1[3];
  int cnt = 0;
  for (int r = rmin; r < rmax && cnt < maxcon; r++) {
    int nvert = 0;
    for (int c = cmin; c <= cmax && cnt < maxcon; c++)
      for (int i = 0; i < 2 && cnt < maxcon; i++) {
        /* addVert: shift the strip by one vertex */
        for (int k = 0; k < 3; k++) { prism.prism[0][k] = prism.prism[1][k]; prism.prism[1][k] = prism.prism[2][k]; prism.prism[3][k] = prism.prism[4][k]; prism.prism[4][k] = prism.prism[5][k]; }
        prism.prism[2][0] = prism.prism[5][0] = dx * c - size1[0];
        prism.prism[2][1] = prism.prism[5][1] = dy * (r + dr[i]) - size1[1];
        prism.prism[5][2] = data[(r + dr[i]) * ncol + c] * size1[2] + margin;
        if (++nvert <= 2) continue;
        if (prism.prism[3][2] < lo[2] && prism.prism[4][2] < lo[2] && prism.prism[5][2] < lo[2]) continue;  /* prism below the geom */
        double depth, dir[3], vec[3];
        o2.cur = 0;
        if (mpr_penetration(&prism, &o2, m->mpr_iterations, m->mpr_tolerance, &depth, dir, vec) == 0 && !ccd_is_zero(depth)) {
          om_contact* cc = con + cnt;
          cc->dist = -depth;
          for (int k = 0; k < 3; k++) {
            cc->frame[k] = mat1[3 * k] * dir[0] + mat1[3 * k + 1] * dir[1] + mat1[3 * k + 2] * dir[2];
            cc->pos[k] = mat1[3 * k] * vec[0] + mat1[3 * k + 1] * vec[1] + mat1[3 * k + 2] * vec[2] + pos1[k];
          }
          memset(cc->frame + 3, 0, 6 * sizeof(double));
          cnt++;
        }
      }
  }
  for (int k = 0; k < cnt; k++) fix_normal(m, d, con + k, g1, g2);
  return cnt;
}

/* contact parameter mixing — mj_contactParam [recall], mjmodel.h:733-741 */
static void contact_param(const om_model* m, om_contact* c, int g1, int g2) {
  c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
  double fr[3];
  int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
  if (p1 != p2) {
    int g = p1 > p2 ? g1 : g2;
    c->dim = m->geom_condim[g];
    memcpy(fr, m->geom_friction + 3 * g, sizeof fr);
    memcpy(c->solref, m->geom_solref + 2 * g, sizeof c->solref);
    memcpy(c->solimp, m->geom_solimp + 5 * g, sizeof c->solimp);
  } else {
    for (int i = 0; i < 3; i++) fr[i] = fmax(m->geom_friction[3 * g1 + i], m->geom_friction[3 * g2 + i]);
    double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
    if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
    else mix = s1 < MINVAL ? 0.0 : 1.0;
    const double *r1 = m->geom_solref + 2 * g1, *r2 = m->geom_solref + 2 * g2;
    if (r1[0] > 0 && r2[0] > 0) for (int i = 0; i < 2; i++) c->solref[i] = mix * r1[i] + (1 - mix) * r2[i];
    else for (int i = 0; i < 2; i++) c->solref[i] = fmin(r1[i], r2[i]);
    for (int i = 0; i < 5; i++) c->solimp[i] = mix * m->geom_solimp[5 * g1 + i] + (1 - mix) * m->geom_solimp[5 * g2 + i];
  }
  c->friction[0] = c->friction[1] = fmax(MINMU, fr[0]);
  c->friction[2] = fmax(MINMU, fr[1]);
  c->friction[3] = c->friction[4] = fmax(MINMU, fr[2]);
}

/* ANALYSIS KNOB (tests/test_oracle_contact_order.py, tools/contact_order_sensitivity.py): the order in which mj_collision emits
 * contacts is one of the restatement's unverifiable choices (this file: static (geom1, geom2) order; MuJoCo 3.1: sorted body pairs,
 * then a BVH traversal inside a body pair).  The order changes nothing physical, but Gauss-Seidel cut at a finite sweep count depends
 * on it.  To MEASURE that dependence the contact list can be re-ordered after the collision pass: 0 as emitted (default; what the
 * device mirrors), 1 reversed inside every body pair (a different intra-pair traversal), 2 fully reversed, 3 a seeded shuffle. */
static int g_contact_order = 0;
static unsigned g_contact_seed = 0;
void om_set_contact_order(int mode, unsigned seed) { g_contact_order = mode; g_contact_seed = seed; }
static void permute_contacts(const om_model* m, om_data* d) {
  const int n = d->ncon;
  if (!g_contact_order || n < 2) return;
  om_contact tmp;
  if (g_contact_order == 1) {
    for (int a = 0; a < n;) {
      int b = a;
      const int b1 = m->geom_bodyid[d->contact[a].geom1], b2 = m->geom_bodyid[d->contact[a].geom2];
      while (b + 1 < n && m->geom_bodyid[d->contact[b + 1].geom1] == b1 && m->geom_bodyid[d->contact[b + 1].geom2] == b2) b++;
      for (int i = a, j = b; i < j; i++, j--) { tmp = d->contact[i]; d->contact[i] = d->contact[j]; d->contact[j] = tmp; }
      a = b + 1;
    }
  } else if (g_contact_order == 2) {
    for (int i = 0, j = n - 1; i < j; i++, j--) { tmp = d->contact[i]; d->contact[i] = d->contact[j]; d->contact[j] = tmp; }
  } else {
    unsigned long long x = 0x9E3779B97F4A7C15ull * (g_contact_seed + 1);
    for (int i = n - 1; i > 0; i--) {  /* Fisher-Yates on a splitmix64 stream */
      x += 0x9E3779B97F4A7C15ull;
      unsigned long long z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      const int j = (int)(z % (unsigned long long)(i + 1));
      tmp = d->contact[i]; d->contact[i] = d->contact[j]; d->contact[j] = tmp;
    }
  }
}

/* mj_collision, mujoco.h:355: candidate pairs are the statically filtered list in the model
 * (same-body / parent-child / exclude / contype filters applied at compile time), visited in
 * (geom1, geom2) order; a bounding-sphere test stands in for the broadphase. */
static void collision(const om_model* m, om_data* d) {
  d->ncon = 0;
  if (m->disableflags & (DSBL_CONSTRAINT | DSBL_CONTACT)) return;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
    double gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
    const double *pos1 = d->geom_xpos + 3 * g1, *pos2 = d->geom_xpos + 3 * g2, *mat1 = d->geom_xmat + 9 * g1, *mat2 = d->geom_xmat + 9 * g2;
    const double *s1 = m->geom_size + 3 * g1, *s2 = m->geom_size + 3 * g2;
    om_contact con[OM_MAXCONPAIR];
    int n = 0;
    if (t1 == GEOM_HFIELD) {
      n = convex_hfield(m, d, con, OM_MAXCONPAIR, g1, g2, margin);  /* MuJoCo's scheme: every geom type goes through the prisms */
    } else if (t1 == GEOM_MESH || t2 == GEOM_MESH) {
      double dp[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
      if (t1 == GEOM_PLANE) {
        double normal[3] = {mat1[2], mat1[5], mat1[8]};
        if (dot3(dp, normal) > margin + m->geom_rbound[g2]) continue;
        n = plane_convex(m, d, con, g1, g2, margin);
      } else {
        double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
        if (dot3(dp, dp) > bound * bound) continue;
        n = convex_convex(m, d, con, g1, g2, margin);
      }
    } else if (t1 == GEOM_PLANE) {
      double normal[3] = {mat1[2], mat1[5], mat1[8]};
      double dp[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
      if (dot3(dp, normal) > margin + m->geom_rbound[g2]) continue;
      if (t2 == GEOM_SPHERE) n = plane_sphere(con, margin, pos1, normal, pos2, s2[0]);
      else { /* capsule: two end spheres, frames aligned with the capsule axis */
        double axis[3] = {mat2[2], mat2[5], mat2[8]}, e1[3], e2[3];
        for (int i = 0; i < 3; i++) { e1[i] = pos2[i] + axis[i] * s2[1]; e2[i] = pos2[i] - axis[i] * s2[1]; }
        int n1 = plane_sphere(con, margin, pos1, normal, e1, s2[0]);
        int n2 = plane_sphere(con + n1, margin, pos1, normal, e2, s2[0]);
        n = n1 + n2;
        for (int k = 0; k < n; k++) memcpy(con[k].frame + 3, axis, sizeof axis);
      }
    } else {
      double dp[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
      double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
      if (dot3(dp, dp) > bound * bound) continue;
      if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE) n = sphere_sphere(con, margin, pos1, s1[0], pos2, s2[0]);
      else if (t1 == GEOM_SPHERE && t2 == GEOM_CAPSULE) {
        double axis[3] = {mat2[2], mat2[5], mat2[8]};
        double vec[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
        double x = clipd(dot3(axis, vec), -s2[1], s2[1]);
        double pt[3] = {pos2[0] + axis[0] * x, pos2[1] + axis[1] * x, pos2[2] + axis[2] * x};
        n = sphere_sphere(con, margin, pos1, s1[0], pt, s2[0]);
      } else {
        double a1[3] = {mat1[2], mat1[5], mat1[8]}, a2[3] = {mat2[2], mat2[5], mat2[8]};
        n = capsule_capsule(con, margin, pos1, a1, s1[0], s1[1], pos2, a2, s2[0], s2[1]);
      }
    }
    for (int k = 0; k < n; k++) {
      if (d->ncon >= OM_MAXCON) { d->warning[WARN_CONTACTFULL]++; return; }
      om_contact* c = d->contact + d->ncon++;
      *c = con[k];
      make_frame(c->frame);
      c->includemargin = margin - gap;
      c->geom1 = g1; c->geom2 = g2;
      contact_param(m, c, g1, g2);
    }
  }
  permute_contacts(m, d);
}

/* ------------------------------------------------------------------ constraints ---------- */

static int add_row(const om_model* m, om_data* d, int type, int id, double pos, double margin) {
  if (d->nefc >= OM_MAXEFC) { d->warning[WARN_CNSTRFULL]++; return -1; }
  int i = d->nefc++;
  memset(d->efc_J + (size_t)i * m->nv, 0, sizeof(double) * m->nv);
  d->efc_type[i] = type; d->efc_id[i] = id; d->efc_pos[i] = pos; d->efc_margin[i] = margin;
  return i;
}

/* impedance sigmoid — getimpedance in engine_core_constraint.c [recall]; solimp mjmodel.h:711 */
static double impedance(const double* solimp_in, double pos, double margin) {
  double si[5];
  si[0] = clipd(solimp_in[0], MINIMP, MAXIMP); si[1] = clipd(solimp_in[1], MINIMP, MAXIMP);
  si[2] = fmax(0, solimp_in[2]); si[3] = clipd(solimp_in[3], MINIMP, MAXIMP); si[4] = fmax(1, solimp_in[4]);
  if (si[0] == si[1] || si[2] <= MINVAL) return 0.5 * (si[0] + si[1]);
  double x = fabs((pos - margin) / si[2]);
  if (x >= 1) return si[1];
  if (x <= 0) return si[0];
  double y;
  if (si[4] == 1) y = x;
  else if (x <= si[3]) y = pow(x, si[4]) / pow(si[3], si[4] - 1);
  else y = 1 - pow(1 - x, si[4]) / pow(1 - si[3], si[4] - 1);
  return si[0] + y * (si[1] - si[0]);
}

/* mj_makeConstraint, mujoco.h:358: rows in order limits(joint), limits(tendon), contacts;
 * then efc_diagApprox, efc_R/D, efc_KBIP. */
static void make_constraint(const om_model* m, om_data* d) {
  int nv = m->nv;
  d->nefc = d->nl = 0;
  if (m->disableflags & DSBL_CONSTRAINT) return;
  if (!(m->disableflags & DSBL_LIMIT)) {
    for (int j = 0; j < m->njnt; j++) {
      if (!m->jnt_limited[j] || (m->jnt_type[j] != JNT_HINGE && m->jnt_type[j] != JNT_SLIDE)) continue;
      double value = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
      for (int side = -1; side <= 1; side += 2) {
        double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - value);
        if (dist < margin) {
          int i = add_row(m, d, CNSTR_LIMIT_JOINT, j, dist, margin);
          if (i < 0) return;
          d->efc_J[(size_t)i * nv + m->jnt_dofadr[j]] = -side;
        }
      }
    }
    for (int t = 0; t < m->ntendon; t++) {
      if (!m->tendon_limited[t]) continue;
      double value = d->ten_length[t], margin = m->tendon_margin[t];
      for (int side = -1; side <= 1; side += 2) {
        double dist = side * (m->tendon_range[2 * t + (side + 1) / 2] - value);
        if (dist < margin) {
          int i = add_row(m, d, CNSTR_LIMIT_TENDON, t, dist, margin);
          if (i < 0) return;
          for (int k = 0; k < nv; k++) d->efc_J[(size_t)i * nv + k] = -side * d->ten_J[t * nv + k];
        }
      }
    }
    d->nl = d->nefc;
  }
  if (!(m->disableflags & DSBL_CONTACT)) {
    double* jp1 = (double*)malloc(sizeof(double) * 3 * nv * 6);
    double *jp2 = jp1 + 3 * nv, *jd = jp2 + 3 * nv, *jr1 = jd + 6 * nv, *jr2 = jr1 + 3 * nv;  /* jd: 6 rows (3 translational, 3 rotational) */
    for (int ci = 0; ci < d->ncon; ci++) {
      om_contact* c = d->contact + ci;
      c->efc_address = -1;
      if (c->dist >= c->includemargin) continue; /* in the gap: excluded */
      int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
      jac(m, d, jp1, NULL, c->pos, b1);
      jac(m, d, jp2, NULL, c->pos, b2);
      /* translational Jacobian difference rotated into the contact frame: jd[r] = frame[r] . (J2 - J1) */
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < nv; k++) {
          double s = 0;
          for (int a = 0; a < 3; a++) s += c->frame[3 * r + a] * (jp2[a * nv + k] - jp1[a * nv + k]);
          jd[r * nv + k] = s;
        }
      if (c->dim == 1) {
        int i = add_row(m, d, CNSTR_CONTACT_FRICTIONLESS, ci, c->dist, c->includemargin);
        if (i < 0) break;
        c->efc_address = i;
        memcpy(d->efc_J + (size_t)i * nv, jd, sizeof(double) * nv);
      } else {
        /* pyramidal cone of dimension dim: 2 (dim - 1) rows  normal +- friction[k-1] * (direction k), k = 1 .. dim-1, where
         * directions 1, 2 are the tangents (relative linear velocity), 3 the spin about the normal and 4, 5 the rolling about the
         * tangents (relative ANGULAR velocity in the contact frame): mj_makeConstraint / mj_instantiateContact [recall] */
        if (c->dim > 3) {
          jac(m, d, NULL, jr1, c->pos, b1);
          jac(m, d, NULL, jr2, c->pos, b2);
          for (int r = 0; r < 3; r++)
            for (int k = 0; k < nv; k++) {
              double s = 0;
              for (int a = 0; a < 3; a++) s += c->frame[3 * r + a] * (jr2[a * nv + k] - jr1[a * nv + k]);
              jd[(3 + r) * nv + k] = s;
            }
        }
        int first = -1;
        for (int k = 1; k < c->dim; k++)
          for (int sgn = 1; sgn >= -1; sgn -= 2) {
            int i = add_row(m, d, CNSTR_CONTACT_PYRAMIDAL, ci, c->dist, c->includemargin);
            if (i < 0) { first = -2; break; }
            if (first < 0) first = i;
            for (int q = 0; q < nv; q++) d->efc_J[(size_t)i * nv + q] = jd[q] + sgn * c->friction[k - 1] * jd[k * nv + q];
          }
        if (first == -2) break;
        c->efc_address = first;
      }
    }
    free(jp1);
  }
  /* diagApprox — mj_diagApprox [recall] */
  for (int i = 0; i < d->nefc; i++) {
    int id = d->efc_id[i];
    switch (d->efc_type[i]) {
      case CNSTR_LIMIT_JOINT: d->efc_diagApprox[i] = m->dof_invweight0[m->jnt_dofadr[id]]; break;
      case CNSTR_LIMIT_TENDON: d->efc_diagApprox[i] = m->tendon_invweight0[id]; break;
      default: {
        om_contact* c = d->contact + id;
        int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
        double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
        if (d->efc_type[i] == CNSTR_CONTACT_FRICTIONLESS) d->efc_diagApprox[i] = tran;
        else {
          int j = i - c->efc_address;
          double fri = c->friction[j / 2];
          double rot = m->body_invweight0[2 * b1 + 1] + m->body_invweight0[2 * b2 + 1];
          d->efc_diagApprox[i] = tran + fri * fri * (j / 2 < 2 ? tran : rot); /* directions 1, 2 are translational, 3..5 rotational */
        }
      }
    }
  }
  /* impedance, R, D, KBIP — mj_makeImpedance [recall] */
  for (int i = 0; i < d->nefc; i++) {
    int id = d->efc_id[i];
    const double *solref, *solimp;
    switch (d->efc_type[i]) {
      case CNSTR_LIMIT_JOINT: solref = m->jnt_solref + 2 * id; solimp = m->jnt_solimp + 5 * id; break;
      case CNSTR_LIMIT_TENDON: solref = m->tendon_solref_lim + 2 * id; solimp = m->tendon_solimp_lim + 5 * id; break;
      default: solref = d->contact[id].solref; solimp = d->contact[id].solimp;
    }
    double imp = impedance(solimp, d->efc_pos[i], d->efc_margin[i]);
    imp = clipd(imp, MINIMP, MAXIMP);
    d->efc_R[i] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[i] / imp);
    double dmax = clipd(solimp[1], MINIMP, MAXIMP), K, B;
    if (solref[0] > 0) {
      double tc = solref[0], dr = solref[1];
      if (!(m->disableflags & DSBL_REFSAFE)) tc = fmax(tc, 2 * m->timestep);
      K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
      B = 2 / fmax(MINVAL, dmax * tc);
    } else { K = -solref[0] / fmax(MINVAL, dmax * dmax); B = -solref[1] / fmax(MINVAL, dmax); }
    d->efc_KBIP[4 * i] = K; d->efc_KBIP[4 * i + 1] = B; d->efc_KBIP[4 * i + 2] = imp; d->efc_KBIP[4 * i + 3] = 0;
  }
  /* pyramidal contacts: all rows of a contact share Rpy = 2 mu^2 R(first row), mu = friction/sqrt(impratio) */
  for (int ci = 0; ci < d->ncon; ci++) {
    om_contact* c = d->contact + ci;
    if (c->efc_address < 0 || c->dim == 1) continue;
    int i0 = c->efc_address, nr = 2 * (c->dim - 1);
    c->mu = c->friction[0] / sqrt(m->impratio);
    double Rpy = 2 * c->mu * c->mu * d->efc_R[i0];
    for (int j = 0; j < nr; j++) d->efc_R[i0 + j] = Rpy;
  }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1 / d->efc_R[i];
}

/* mj_projectConstraint, mujoco.h:364: AR = J M^-1 J^T + diag(R).  B = M^-1 J^T kept for qacc. */
static void project_constraint(const om_model* m, om_data* d) {
  int nv = m->nv, n = d->nefc;
  for (int i = 0; i < n; i++) {
    double* b = d->efc_B + (size_t)i * nv;
    memcpy(b, d->efc_J + (size_t)i * nv, sizeof(double) * nv);
    solve_ld(m, b, d->qLD, d->qLDiagInv);
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = 0;
      const double *Ji = d->efc_J + (size_t)i * nv, *Bj = d->efc_B + (size_t)j * nv;
      for (int k = 0; k < nv; k++) s += Ji[k] * Bj[k];
      d->efc_AR[(size_t)i * n + j] = d->efc_AR[(size_t)j * n + i] = s;
    }
  for (int i = 0; i < n; i++) d->efc_AR[(size_t)i * n + i] += d->efc_R[i];
}

/* ------------------------------------------------------------------ velocity stage ------- */

/* mj_comVel, mujoco.h:340 */
static void com_vel(const om_model* m, om_data* d) {
  memset(d->cvel, 0, 6 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    double cvel[6], t[6];
    memcpy(cvel, d->cvel + 6 * m->body_parentid[b], sizeof cvel);
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == JNT_FREE) {
        memset(d->cdof_dot + 6 * da, 0, 18 * sizeof(double));
        for (int k = 0; k < 3; k++) for (int i = 0; i < 6; i++) cvel[i] += d->cdof[6 * (da + k) + i] * d->qvel[da + k];
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot + 6 * (da + k), cvel, d->cdof + 6 * (da + k));
        for (int k = 3; k < 6; k++) for (int i = 0; i < 6; i++) cvel[i] += d->cdof[6 * (da + k) + i] * d->qvel[da + k];
      } else {
        cross_motion(t, cvel, d->cdof + 6 * da);
        memcpy(d->cdof_dot + 6 * da, t, sizeof t);
        for (int i = 0; i < 6; i++) cvel[i] += d->cdof[6 * da + i] * d->qvel[da];
      }
    }
    memcpy(d->cvel + 6 * b, cvel, sizeof cvel);
  }
}

/* mj_passive, mujoco.h:343: joint springs and dampers */
static void passive(const om_model* m, om_data* d) {
  memset(d->qfrc_passive, 0, sizeof(double) * m->nv);
  if (m->disableflags & DSBL_PASSIVE) return;
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_stiffness[j] == 0) continue;
    if (m->jnt_type[j] == JNT_HINGE || m->jnt_type[j] == JNT_SLIDE)
      d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[m->jnt_qposadr[j]] - m->qpos_spring[m->jnt_qposadr[j]]);
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
}

/* mj_rne with flg_acc=0, mujoco.h:349: bias force = Coriolis + centrifugal + gravity */
static void rne(const om_model* m, om_data* d, double* result) {
  memset(d->cacc, 0, 6 * sizeof(double));
  if (!(m->disableflags & DSBL_GRAVITY)) for (int i = 0; i < 3; i++) d->cacc[3 + i] = -m->gravity[i];
  memset(d->cfrc_body, 0, 6 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    double t[6], t1[6];
    memcpy(d->cacc + 6 * b, d->cacc + 6 * m->body_parentid[b], 6 * sizeof(double));
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int da = m->body_dofadr[b] + k;
      for (int i = 0; i < 6; i++) d->cacc[6 * b + i] += d->cdof_dot[6 * da + i] * d->qvel[da];
    }
    mul_inert_vec(t, d->cinert + 10 * b, d->cacc + 6 * b);
    mul_inert_vec(t1, d->cinert + 10 * b, d->cvel + 6 * b);
    cross_force(d->cfrc_body + 6 * b, d->cvel + 6 * b, t1);
    for (int i = 0; i < 6; i++) d->cfrc_body[6 * b + i] += t[i];
  }
  for (int b = m->nbody - 1; b > 0; b--) { int p = m->body_parentid[b]; if (p > 0) for (int i = 0; i < 6; i++) d->cfrc_body[6 * p + i] += d->cfrc_body[6 * b + i]; }
  for (int i = 0; i < m->nv; i++) {
    double s = 0;
    for (int t = 0; t < 6; t++) s += d->cdof[6 * i + t] * d->cfrc_body[6 * m->dof_bodyid[i] + t];
    result[i] = s;
  }
}

/* mj_referenceConstraint, mujoco.h:367 */
static void reference_constraint(const om_model* m, om_data* d) {
  int nv = m->nv;
  for (int i = 0; i < d->nefc; i++) {
    double s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qvel[k];
    d->efc_vel[i] = s;
    d->efc_aref[i] = -d->efc_KBIP[4 * i + 1] * s - d->efc_KBIP[4 * i] * d->efc_KBIP[4 * i + 2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
}

/* ------------------------------------------------------------------ acceleration stage --- */

/* mj_fwdActuation, mujoco.h:253 */
static void fwd_actuation(const om_model* m, om_data* d) {
  memset(d->qfrc_actuator, 0, sizeof(double) * m->nv);
  if (m->disableflags & DSBL_ACTUATION) return;
  for (int a = 0; a < m->nu; a++) {
    double ctrl = d->ctrl[a];
    if (m->actuator_ctrllimited[a] && !(m->disableflags & DSBL_CLAMPCTRL)) ctrl = clipd(ctrl, m->actuator_ctrlrange[2 * a], m->actuator_ctrlrange[2 * a + 1]);
    int j = m->actuator_trnid[a];
    double gear = m->actuator_gear[a];
    double length = gear * d->qpos[m->jnt_qposadr[j]], velocity = gear * d->qvel[m->jnt_dofadr[j]];
    double force = m->actuator_gainprm[a] * ctrl + m->actuator_biasprm[3 * a] + m->actuator_biasprm[3 * a + 1] * length + m->actuator_biasprm[3 * a + 2] * velocity;
    if (m->actuator_forcelimited[a]) force = clipd(force, m->actuator_forcerange[2 * a], m->actuator_forcerange[2 * a + 1]);
    d->actuator_force[a] = force;
    d->qfrc_actuator[m->jnt_dofadr[j]] += gear * force;
  }
}

/* mj_fwdAcceleration, mujoco.h:256 */
static void fwd_acceleration(const om_model* m, om_data* d) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
  /* mj_xfrcAccumulate: Cartesian wrenches applied at body coms */
  double* jp = (double*)malloc(sizeof(double) * 6 * nv);
  double* jr = jp + 3 * nv;
  for (int b = 1; b < m->nbody; b++) {
    const double* f = d->xfrc_applied + 6 * b;
    if (f[0] == 0 && f[1] == 0 && f[2] == 0 && f[3] == 0 && f[4] == 0 && f[5] == 0) continue;
    jac(m, d, jp, jr, d->xipos + 3 * b, b);
    for (int k = 0; k < nv; k++)
      for (int r = 0; r < 3; r++) d->qfrc_smooth[k] += jp[r * nv + k] * f[r] + jr[r * nv + k] * f[3 + r];
  }
  free(jp);
  memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double) * nv);
  solve_ld(m, d->qacc_smooth, d->qLD, d->qLDiagInv);
}

/* mj_constraintUpdate restricted to limit/contact rows (mujoco.h:371): f = max(0, -D*jar) */
static void constraint_update(om_data* d, const double* jar) {
  for (int i = 0; i < d->nefc; i++) d->efc_force[i] = jar[i] < 0 ? -d->efc_D[i] * jar[i] : 0;
}

/* ---- Newton solver (mjSOL_NEWTON, mjmodel.h:162; the reference's default: its humanoid XML sets no solver) ----
 * mj_solNewton -> the primal solver of MuJoCo's engine_solver.c [recall; the source is not vendored, see header].
 * Published algorithm (MuJoCo "Computation" chapter, "Solver algorithms / Newton"): minimise over qacc
 *     cost(qacc) = 1/2 (qacc - qacc_smooth)' M (qacc - qacc_smooth) + s(J qacc - aref),
 * where for the inequality rows of this path (limits, pyramidal contacts) s = sum_i 1/2 D_i min(0, jar_i)^2,
 * by Newton steps  search = -H^-1 grad,  H = M + J' diag(D_i [jar_i < 0]) J  (Cholesky),
 * grad = M qacc - qfrc_smooth - J' force,  force_i = -D_i min(0, jar_i),
 * each followed by an exact line search on the (convex, piecewise quadratic) 1-D restriction.
 * Termination: scale * (cost decrease) < tolerance or scale * |grad| < tolerance, scale = 1/(meaninertia * max(1, nv)).
 * The minimiser of this strictly convex problem is unique, so the converged qacc does not depend on line-search
 * bookkeeping; iteration counts may differ from MuJoCo's by the details recalled here. */

/* mj_mulM, mujoco.h:381: res = M * vec with M in the sparse dof-ancestor format */
static void mul_M(const om_model* m, const double* qM, const double* v, double* res) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) res[i] = 0;
  for (int i = 0; i < nv; i++) {
    int adr = m->dof_Madr[i];
    res[i] += qM[adr] * v[i];
    adr++;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j], adr++) { res[i] += qM[adr] * v[j]; res[j] += qM[adr] * v[i]; }
  }
}

typedef struct { double alpha, cost, d0, d1; } om_lspt;
typedef struct { int n; const double *jar, *Jv, *D; double qg[3]; int nev; } om_lsctx;

/* PrimalEval [recall]: cost and its first two derivatives along the search direction at p->alpha */
static void ls_eval(om_lsctx* c, om_lspt* p) {
  double a = p->alpha;
  double cost = c->qg[0] + a * c->qg[1] + a * a * c->qg[2], d0 = c->qg[1] + 2 * a * c->qg[2], d1 = 2 * c->qg[2];
  for (int i = 0; i < c->n; i++) {
    double x = c->jar[i] + a * c->Jv[i];
    if (x < 0) { cost += 0.5 * c->D[i] * x * x; d0 += c->D[i] * x * c->Jv[i]; d1 += c->D[i] * c->Jv[i] * c->Jv[i]; }
  }
  if (d1 <= 0) d1 = MINVAL;
  p->cost = cost; p->d0 = d0; p->d1 = d1;
  c->nev++;
}

/* updateBracket [recall]: move p to the candidate on its side of the sign change that is closest to it */
static int ls_update_bracket(om_lsctx* c, om_lspt* p, const om_lspt cand[3], om_lspt* pnext) {
  int flag = 0;
  for (int i = 0; i < 3; i++) {
    if (p->d0 < 0 && cand[i].d0 < 0 && p->d0 < cand[i].d0) { *p = cand[i]; flag = 1; }
    else if (p->d0 > 0 && cand[i].d0 > 0 && p->d0 > cand[i].d0) { *p = cand[i]; flag = 2; }
  }
  if (flag) { pnext->alpha = p->alpha - p->d0 / p->d1; ls_eval(c, pnext); }
  return flag;
}

/* PrimalSearch [recall]: exact line search, Newton iterations in alpha, one-sided until the derivative changes
   sign, then bracketed */
static double ls_search(const om_model* m, om_lsctx* c, double snorm, double scale) {
  if (snorm < MINVAL) return 0;
  double gtol = m->tolerance * m->ls_tolerance * snorm / scale;
  int lsmax = m->ls_iterations, it = 0;
  om_lspt p0, p1, p2, pmid, p1next, p2next;
  p0.alpha = 0; ls_eval(c, &p0);
  p1.alpha = p0.alpha - p0.d0 / p0.d1; ls_eval(c, &p1);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d0) < gtol) return p1.alpha;
  int dir = p1.d0 < 0 ? 1 : -1, p2update = 0;
  p2 = p1;
  while (p1.d0 * dir <= -gtol && it < lsmax) {
    p2 = p1; p2update = 1;
    p1.alpha = p1.alpha - p1.d0 / p1.d1; ls_eval(c, &p1); it++;
    if (fabs(p1.d0) < gtol) return p1.alpha;
  }
  if (it >= lsmax || !p2update) return p1.alpha;
  p2next = p1;
  p1next.alpha = p1.alpha - p1.d0 / p1.d1; ls_eval(c, &p1next);
  while (it < lsmax) {
    pmid.alpha = 0.5 * (p1.alpha + p2.alpha); ls_eval(c, &pmid); it++;
    om_lspt cand[3] = {p1next, p2next, pmid};
    int best = -1;
    for (int i = 0; i < 3; i++) if (fabs(cand[i].d0) < gtol && (best < 0 || cand[i].cost < cand[best].cost)) best = i;
    if (best >= 0) return cand[best].alpha;
    int b1 = ls_update_bracket(c, &p1, cand, &p1next), b2 = ls_update_bracket(c, &p2, cand, &p2next);
    if (!b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : 0;
  }
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0;
}

/* dense Cholesky H = L L' in place (lower triangle), mju_cholFactor (mujoco.h:1211) incl. its diagonal floor */
static void chol_factor(double* H, int n) {
  for (int j = 0; j < n; j++) {
    double t = H[j * n + j];
    for (int k = 0; k < j; k++) t -= H[j * n + k] * H[j * n + k];
    if (t < MINVAL) t = MINVAL;
    t = sqrt(t);
    H[j * n + j] = t;
    for (int i = j + 1; i < n; i++) {
      double s = H[i * n + j];
      for (int k = 0; k < j; k++) s -= H[i * n + k] * H[j * n + k];
      H[i * n + j] = s / t;
    }
  }
}
static void chol_solve(const double* L, int n, double* x) { /* mju_cholSolve, mujoco.h:1214 */
  for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}

/* constraint part of the primal cost at jar (mj_constraintUpdate, mujoco.h:371) */
static double primal_constraint_cost(const om_data* d, const double* jar) {
  double c = 0;
  for (int i = 0; i < d->nefc; i++) if (jar[i] < 0) c += 0.5 * d->efc_D[i] * jar[i] * jar[i];
  return c;
}

static void sol_newton(const om_model* m, om_data* d) {
  int nv = m->nv, n = d->nefc;
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  double* buf = (double*)calloc((size_t)6 * nv + 2 * n + (size_t)nv * nv, sizeof(double));
  double *Ma = buf, *grad = Ma + nv, *search = grad + nv, *Mv = search + nv, *tmp = Mv + nv, *dq = tmp + nv, *jar = dq + nv, *Jv = jar + n, *H = Jv + n;
  int* state = (int*)calloc(2 * (size_t)n + 1, sizeof(int));
  double cost = 0;
  d->solver_nls = 0;
  /* initial point: d->qacc (set by the warm start) */
  mul_M(m, d->qM, d->qacc, Ma);
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc[k];
    jar[i] = s - d->efc_aref[i];
  }
  int need_factor = 1, iter = 0;
  for (;;) {
    /* PrimalUpdateConstraint: force, state, qfrc_constraint, cost */
    constraint_update(d, jar);
    int changed = need_factor;
    for (int i = 0; i < n; i++) { int st = jar[i] < 0; if (st != state[i]) changed = 1; state[i] = st; }
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    for (int i = 0; i < n; i++) if (state[i]) for (int k = 0; k < nv; k++) d->qfrc_constraint[k] += d->efc_J[(size_t)i * nv + k] * d->efc_force[i];
    double oldcost = cost;
    cost = primal_constraint_cost(d, jar);
    for (int k = 0; k < nv; k++) cost += 0.5 * (Ma[k] - d->qfrc_smooth[k]) * (d->qacc[k] - d->qacc_smooth[k]);
    /* Hessian (MakeHessian / HessianIncremental give the same matrix): H = M + J' diag(D active) J, Cholesky */
    if (changed) {
      memset(H, 0, sizeof(double) * nv * nv);
      for (int i = 0; i < nv; i++) {
        int adr = m->dof_Madr[i];
        for (int j = i; j >= 0; j = m->dof_parentid[j], adr++) { H[i * nv + j] = d->qM[adr]; H[j * nv + i] = d->qM[adr]; }
      }
      for (int r = 0; r < n; r++) if (state[r]) {
        const double* Jr = d->efc_J + (size_t)r * nv;
        for (int i = 0; i < nv; i++) if (Jr[i] != 0) for (int j = 0; j < nv; j++) H[i * nv + j] += d->efc_D[r] * Jr[i] * Jr[j];
      }
      chol_factor(H, nv);
      need_factor = 0;
    }
    /* PrimalUpdateGradient */
    double gnorm = 0;
    for (int k = 0; k < nv; k++) { grad[k] = Ma[k] - d->qfrc_smooth[k] - d->qfrc_constraint[k]; gnorm += grad[k] * grad[k]; tmp[k] = grad[k]; }
    gnorm = sqrt(gnorm);
    chol_solve(H, nv, tmp);
    if (iter > 0) {
      double improvement = scale * (oldcost - cost), gradient = scale * gnorm;
      if (improvement < m->tolerance || gradient < m->tolerance) break;
    }
    if (iter >= m->iterations) break;
    for (int k = 0; k < nv; k++) search[k] = -tmp[k];
    /* PrimalSearch */
    mul_M(m, d->qM, search, Mv);
    double snorm = 0, qg1 = 0, qg2 = 0;
    for (int k = 0; k < nv; k++) { snorm += search[k] * search[k]; qg1 += search[k] * (Ma[k] - d->qfrc_smooth[k]); qg2 += 0.5 * search[k] * Mv[k]; }
    snorm = sqrt(snorm);
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * search[k];
      Jv[i] = s;
    }
    double gauss = 0;
    for (int k = 0; k < nv; k++) gauss += 0.5 * (Ma[k] - d->qfrc_smooth[k]) * (d->qacc[k] - d->qacc_smooth[k]);
    om_lsctx c = {n, jar, Jv, d->efc_D, {gauss, qg1, qg2}, 0};
    double alpha = ls_search(m, &c, snorm, scale);
    d->solver_nls += c.nev;
    if (alpha == 0) break;
    for (int k = 0; k < nv; k++) { d->qacc[k] += alpha * search[k]; Ma[k] += alpha * Mv[k]; }
    for (int i = 0; i < n; i++) jar[i] += alpha * Jv[i];
    iter++;
  }
  d->solver_niter = iter;
  free(state);
  free(buf);
}

/* mj_fwdConstraint, mujoco.h:259: warm start, then mj_solPGS [recall] or the Newton solver above */
static void fwd_constraint(const om_model* m, om_data* d) {
  int nv = m->nv, n = d->nefc;
  d->solver_niter = 0;
  if (!n) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    return;
  }
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_smooth[k];
    d->efc_b[i] = s - d->efc_aref[i];
  }
  double* jar = (double*)malloc(sizeof(double) * n);
  if (m->solver == SOL_NEWTON) {
    /* warmstart() of mj_fwdConstraint for the primal solvers [recall]: start from qacc_warmstart unless
       qacc_smooth has the lower cost (its Gauss term is zero) */
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    if (!(m->disableflags & DSBL_WARMSTART)) {
      double* Ma = (double*)malloc(sizeof(double) * nv);
      for (int i = 0; i < n; i++) {
        double s = 0;
        for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_warmstart[k];
        jar[i] = s - d->efc_aref[i];
      }
      double cost_warm = primal_constraint_cost(d, jar);
      mul_M(m, d->qM, d->qacc_warmstart, Ma);
      for (int k = 0; k < nv; k++) cost_warm += 0.5 * (Ma[k] - d->qfrc_smooth[k]) * (d->qacc_warmstart[k] - d->qacc_smooth[k]);
      double cost_smooth = primal_constraint_cost(d, d->efc_b);
      if (cost_warm <= cost_smooth) memcpy(d->qacc, d->qacc_warmstart, sizeof(double) * nv);
      free(Ma);
    }
    sol_newton(m, d);
    free(jar);
    return;
  }
  /* warm start */
  if (!(m->disableflags & DSBL_WARMSTART)) {
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_warmstart[k];
      jar[i] = s - d->efc_aref[i];
    }
    constraint_update(d, jar);
    double cost = 0;
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int j = 0; j < n; j++) s += d->efc_AR[(size_t)i * n + j] * d->efc_force[j];
      cost += d->efc_force[i] * (0.5 * s + d->efc_b[i]);
    }
    if (cost > 0) memset(d->efc_force, 0, sizeof(double) * n);
  } else memset(d->efc_force, 0, sizeof(double) * n);
  /* PGS sweeps */
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  while (iter < m->iterations) {
    double improvement = 0;
    for (int i = 0; i < n; i++) {
      const double* row = d->efc_AR + (size_t)i * n;
      double res = d->efc_b[i];
      for (int j = 0; j < n; j++) res += row[j] * d->efc_force[j];
      double old = d->efc_force[i];
      double f = old - res / row[i];
      if (f < 0) f = 0;
      double delta = f - old;
      double change = 0.5 * delta * delta * row[i] + delta * res;
      if (change > 1e-10) { f = old; change = 0; __atomic_fetch_add(&g_pgs_reverts, 1, __ATOMIC_RELAXED); }
      d->efc_force[i] = f;
      improvement -= change;
    }
    improvement *= scale;
    iter++;
    if (improvement < m->tolerance) break;
  }
  d->solver_niter = iter;
  free(jar);
  /* dual finish: qfrc_constraint = J^T f, qacc = qacc_smooth + M^-1 qfrc_constraint */
  memset(d->qfrc_constraint, 0, sizeof(double) * nv);
  for (int i = 0; i < n; i++) for (int k = 0; k < nv; k++) d->qfrc_constraint[k] += d->efc_J[(size_t)i * nv + k] * d->efc_force[i];
  memcpy(d->qacc, d->qfrc_constraint, sizeof(double) * nv);
  solve_ld(m, d->qacc, d->qLD, d->qLDiagInv);
  for (int k = 0; k < nv; k++) d->qacc[k] += d->qacc_smooth[k];
}

/* number of PGS row updates undone by the cost-change test since load (test evidence that the
   device path may leave that test out: it never fires for scalar rows) */
long om_pgs_reverts(void) { return __atomic_load_n(&g_pgs_reverts, __ATOMIC_RELAXED); }

/* ------------------------------------------------------------------ top level ------------ */

static int bad(const double* x, int n) {
  for (int i = 0; i < n; i++) if (isnan(x[i]) || x[i] > MAXVAL || x[i] < -MAXVAL) return 1;
  return 0;
}

/* mj_forward, mujoco.h:129 — stage order per SURVEY.md Appendix B */
void om_forward(const om_model* m, om_data* d) {
  kinematics(m, d);
  com_pos(m, d);
  tendon(m, d);
  crb(m, d);
  factor_i(m, d->qM, d->qLD, d->qLDiagInv);
  collision(m, d);
  make_constraint(m, d);
  if (m->solver == SOL_PGS) project_constraint(m, d);
  com_vel(m, d);
  passive(m, d);
  reference_constraint(m, d);
  rne(m, d, d->qfrc_bias);
  fwd_actuation(m, d);
  fwd_acceleration(m, d);
  fwd_constraint(m, d);
}

/* mj_Euler, mujoco.h:262: semi-implicit Euler with implicit joint damping */
static void euler(const om_model* m, om_data* d) {
  int nv = m->nv;
  double* qacc = (double*)malloc(sizeof(double) * nv);
  int damp = 0;
  if (!(m->disableflags & DSBL_EULERDAMP)) for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) { damp = 1; break; }
  if (!damp) memcpy(qacc, d->qacc, sizeof(double) * nv);
  else {
    memcpy(d->qH, d->qM, sizeof(double) * m->nM);
    for (int i = 0; i < nv; i++) d->qH[m->dof_Madr[i]] += m->timestep * m->dof_damping[i];
    factor_i(m, d->qH, d->qH, d->qHDiagInv);
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    solve_ld(m, qacc, d->qH, d->qHDiagInv);
  }
  /* mj_advance */
  double h = m->timestep;
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  for (int j = 0; j < m->njnt; j++) { /* mj_integratePos, mujoco.h:466 */
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int i = 0; i < 3; i++) d->qpos[qa + i] += h * d->qvel[da + i];
      double v[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]}, qr[4];
      double ang = h * normalize3(v);
      axisangle2quat(qr, v, ang);
      normalize4(d->qpos + qa + 3);
      mulquat(d->qpos + qa + 3, d->qpos + qa + 3, qr);
    } else d->qpos[qa] += h * d->qvel[da];
  }
  d->time += h;
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
  free(qacc);
}

/* mj_step, mujoco.h:120 */
void om_step(const om_model* m, om_data* d) {
  if (bad(d->qpos, m->nq)) { d->warning[WARN_BADQPOS]++; int w = d->warning[WARN_BADQPOS]; om_reset(m, d, -1); d->warning[WARN_BADQPOS] = w; }
  if (bad(d->qvel, m->nv)) { d->warning[WARN_BADQVEL]++; int w = d->warning[WARN_BADQVEL]; om_reset(m, d, -1); d->warning[WARN_BADQVEL] = w; }
  om_forward(m, d);
  if (bad(d->qacc, m->nv)) { d->warning[WARN_BADQACC]++; int w = d->warning[WARN_BADQACC]; om_reset(m, d, -1); d->warning[WARN_BADQACC] = w; om_forward(m, d); }
  d->sum_ncon += d->ncon; d->sum_nefc += d->nefc; d->sum_iter += d->solver_niter; d->nstep++;
  if (d->ncon > d->max_ncon) d->max_ncon = d->ncon;
  if (d->nefc > d->max_nefc) d->max_nefc = d->nefc;
  euler(m, d);
}

/* ------------------------------------------------------------------ accessors for ctypes - */

#define FIELD(x, n) if (!strcmp(name, #x)) { *len = (n); return d->x; }
double* om_data_ptr(const om_model* m, om_data* d, const char* name, int* len) {
  int nb = m->nbody, nv = m->nv, nj = m->njnt, ng = m->ngeom, ne = d->nefc;
  FIELD(qpos, m->nq) FIELD(qvel, nv) FIELD(ctrl, m->nu) FIELD(qacc_warmstart, nv) FIELD(qfrc_applied, nv) FIELD(xfrc_applied, 6 * nb) FIELD(qacc, nv)
  FIELD(xpos, 3 * nb) FIELD(xquat, 4 * nb) FIELD(xmat, 9 * nb) FIELD(xipos, 3 * nb) FIELD(ximat, 9 * nb) FIELD(xanchor, 3 * nj) FIELD(xaxis, 3 * nj)
  FIELD(geom_xpos, 3 * ng) FIELD(geom_xmat, 9 * ng) FIELD(subtree_com, 3 * nb) FIELD(cinert, 10 * nb) FIELD(cdof, 6 * nv) FIELD(cdof_dot, 6 * nv)
  FIELD(crb, 10 * nb) FIELD(cvel, 6 * nb) FIELD(cacc, 6 * nb) FIELD(cfrc_body, 6 * nb) FIELD(qM, m->nM) FIELD(qLD, m->nM) FIELD(qLDiagInv, nv)
  FIELD(ten_length, m->ntendon) FIELD(ten_J, m->ntendon * nv) FIELD(actuator_force, m->nu) FIELD(qfrc_passive, nv) FIELD(qfrc_bias, nv)
  FIELD(qfrc_actuator, nv) FIELD(qfrc_smooth, nv) FIELD(qacc_smooth, nv) FIELD(qfrc_constraint, nv)
  FIELD(efc_J, ne * nv) FIELD(efc_pos, ne) FIELD(efc_margin, ne) FIELD(efc_diagApprox, ne) FIELD(efc_R, ne) FIELD(efc_D, ne) FIELD(efc_KBIP, 4 * ne)
  FIELD(efc_vel, ne) FIELD(efc_aref, ne) FIELD(efc_b, ne) FIELD(efc_force, ne) FIELD(efc_AR, ne * ne)
  *len = 0;
  return NULL;
}
double* om_model_ptr(om_model* m, const char* name, int* len) {
  om_model* d = m;
  int nb = m->nbody, nv = m->nv;
  FIELD(body_mass, nb) FIELD(body_subtreemass, nb) FIELD(body_inertia, 3 * nb) FIELD(body_invweight0, 2 * nb) FIELD(dof_invweight0, nv) FIELD(dof_M0, nv)
  FIELD(tendon_invweight0, m->ntendon) FIELD(qpos0, m->nq) FIELD(key_qpos, m->nkey * m->nq) FIELD(dof_damping, nv) FIELD(dof_armature, nv)
  FIELD(jnt_range, 2 * m->njnt) FIELD(gravity, 3) FIELD(body_ipos, 3 * nb) FIELD(jnt_stiffness, m->njnt)
  FIELD(jnt_margin, m->njnt) FIELD(geom_friction, 3 * m->ngeom) FIELD(actuator_gainprm, m->nu) FIELD(actuator_biasprm, 3 * m->nu)
  FIELD(actuator_forcerange, 2 * m->nu) FIELD(hfield_data, m->nhfielddata)
  *len = 0;
  return NULL;
}
int om_model_int(const om_model* m, const char* name) {
#define MI(x) if (!strcmp(name, #x)) return m->x;
  MI(nq) MI(nv) MI(nu) MI(nbody) MI(njnt) MI(ngeom) MI(ntendon) MI(nM) MI(nkey) MI(npair) MI(iterations) MI(disableflags) MI(solver) MI(nmesh) MI(nmeshvert)
  return -1;
}
void om_model_set_int(om_model* m, const char* name, int v) {
  if (!strcmp(name, "iterations")) m->iterations = v;
  else if (!strcmp(name, "disableflags")) m->disableflags = v;
  else if (!strcmp(name, "solver")) m->solver = v;
  else if (!strcmp(name, "ls_iterations")) m->ls_iterations = v;
  else if (!strcmp(name, "mpr_iterations")) m->mpr_iterations = v;
}
void om_model_set_dbl(om_model* m, const char* name, double v) {
  if (!strcmp(name, "timestep")) m->timestep = v;
  else if (!strcmp(name, "tolerance")) m->tolerance = v;
  else if (!strcmp(name, "impratio")) m->impratio = v;
  else if (!strcmp(name, "meaninertia")) m->meaninertia = v;
  else if (!strcmp(name, "ls_tolerance")) m->ls_tolerance = v;
}
double om_model_dbl(const om_model* m, const char* name) {
  if (!strcmp(name, "timestep")) return m->timestep;
  if (!strcmp(name, "tolerance")) return m->tolerance;
  if (!strcmp(name, "meaninertia")) return m->meaninertia;
  if (!strcmp(name, "impratio")) return m->impratio;
  return 0;
}
int om_data_int(const om_data* d, const char* name) {
  if (!strcmp(name, "ncon")) return d->ncon;
  if (!strcmp(name, "nefc")) return d->nefc;
  if (!strcmp(name, "nl")) return d->nl;
  if (!strcmp(name, "solver_niter")) return d->solver_niter;
  if (!strcmp(name, "solver_nls")) return d->solver_nls;
  if (!strcmp(name, "max_ncon")) return d->max_ncon;
  if (!strcmp(name, "max_nefc")) return d->max_nefc;
  if (!strcmp(name, "warn_contactfull")) return d->warning[WARN_CONTACTFULL];
  if (!strcmp(name, "warn_cnstrfull")) return d->warning[WARN_CNSTRFULL];
  if (!strcmp(name, "warn_badqpos")) return d->warning[WARN_BADQPOS];
  if (!strcmp(name, "warn_badqvel")) return d->warning[WARN_BADQVEL];
  if (!strcmp(name, "warn_badqacc")) return d->warning[WARN_BADQACC];
  return -1;
}
double om_data_time(const om_data* d) { return d->time; }
void om_data_set_time(om_data* d, double t) { d->time = t; }
/* contact k: out[0]=dist, [1..3]=pos, [4..12]=frame, [13]=dim, [14]=geom1, [15]=geom2, [16]=efc_address, [17]=friction0 */
void om_contact_get(const om_data* d, int k, double* out) {
  const om_contact* c = d->contact + k;
  out[0] = c->dist; memcpy(out + 1, c->pos, 3 * sizeof(double)); memcpy(out + 4, c->frame, 9 * sizeof(double));
  out[13] = c->dim; out[14] = c->geom1; out[15] = c->geom2; out[16] = c->efc_address; out[17] = c->friction[0];
  memcpy(out + 18, c->friction, 5 * sizeof(double));  /* out[18..22]: the full friction vector (sliding x2, torsional, rolling x2) */
}
void om_efc_types(const om_data* d, int* type, int* id) { for (int i = 0; i < d->nefc; i++) { type[i] = d->efc_type[i]; id[i] = d->efc_id[i]; } }
void om_stats(const om_data* d, double* out) {
  double n = d->nstep > 0 ? (double)d->nstep : 1.0;
  out[0] = d->sum_ncon / n; out[1] = d->sum_nefc / n; out[2] = d->sum_iter / n; out[3] = d->max_ncon; out[4] = d->max_nefc;
}

/* ------------------------------------------------------------------ batched CPU rollout -- */
/* Shape of simulation/mujoco/sample/testspeed.cc:84-103,203-210: shared read-only model, one
 * data block per env, a contiguous chunk of envs per thread.  Halton controls follow
 * testspeed.cc:64-80 (CtrlNoise) with the index convention of SURVEY.md §8(d). */

double om_halton(int index, int base) { /* mju_Halton, mujoco.h:1231 */
  int n0 = index;
  double b = (double)base, f = 1.0 / b, hn = 0;
  while (n0 > 0) {
    int n1 = n0 / base, r = n0 - n1 * base;
    hn += f * r;
    f /= b;
    n0 = n1;
  }
  return hn;
}

/* initial state of env e (global index) per SURVEY.md §8(d) config 2 */
void om_init_env(const om_model* m, om_data* d, int e) {
  om_reset(m, d, -1);
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j];
    if (m->jnt_type[j] == JNT_FREE) d->qpos[qa + 2] += 0.1 * om_halton(e + 1, 3);
    else d->qpos[qa] += 0.2 * (2 * om_halton(e + 1, 2 + j) - 1);
  }
}
void om_ctrl_env(const om_model* m, double* ctrl, int t, int e) {
  for (int i = 0; i < m->nu; i++) ctrl[i] = 2 * om_halton(1 + t + 1000 * e, i + 2) - 1;
}

typedef struct { const om_model* m; int e0, e1, nstep, env_offset, t_pre; double* qpos_out; double stats[5]; long long steps; double seconds; } om_job;

static double now_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* one chunk of envs on one thread (the loop shape of simulation/mujoco/sample/testspeed.cc:84-103): every env first runs
 * t_pre untimed steps of the same workload, then nstep steps whose wall time and statistics are what the job reports */
static void* rollout_worker(void* arg) {
  om_job* job = (om_job*)arg;
  const om_model* m = job->m;
  om_data* d = om_make_data(m);
  double sc = 0, se = 0, si = 0, mc = 0, me = 0;
  for (int e = job->e0; e < job->e1; e++) {
    om_init_env(m, d, e + job->env_offset);
    for (int t = 0; t < job->t_pre; t++) {
      om_ctrl_env(m, d->ctrl, t, e + job->env_offset);
      om_step(m, d);
    }
    const double c0 = (double)d->sum_ncon, e0 = (double)d->sum_nefc, i0 = (double)d->sum_iter;
    if (job->t_pre > 0) { d->max_ncon = 0; d->max_nefc = 0; }
    const double t0 = now_seconds();
    for (int t = job->t_pre; t < job->t_pre + job->nstep; t++) {
      om_ctrl_env(m, d->ctrl, t, e + job->env_offset);
      om_step(m, d);
    }
    job->seconds += now_seconds() - t0;
    if (job->qpos_out) memcpy(job->qpos_out + (size_t)e * m->nq, d->qpos, sizeof(double) * m->nq);
    sc += (double)d->sum_ncon - c0; se += (double)d->sum_nefc - e0; si += (double)d->sum_iter - i0;
    if (d->max_ncon > mc) mc = d->max_ncon;
    if (d->max_nefc > me) me = d->max_nefc;
    job->steps += job->nstep;
  }
  job->stats[0] = sc; job->stats[1] = se; job->stats[2] = si; job->stats[3] = mc; job->stats[4] = me;
  om_free_data(d);
  return NULL;
}

/* run n_env envs on nthread threads: t_pre untimed steps, then nstep timed ones, per env; qpos_out (nullable) [n_env x nq];
 * stats_out[5] = mean ncon, mean nefc, mean PGS iters, max ncon, max nefc over the timed steps; rate_out (nullable) = sum
 * over threads of (timed env-steps / seconds that thread spent in them): the throughput of the timed window with every
 * thread busy (a thread is in its pre-roll while its neighbours are timed and the other way round).  Returns the timed env-steps. */
long long om_rollout_window(const om_model* m, int n_env, int t_pre, int nstep, int nthread, int env_offset, double* qpos_out, double* stats_out,
                            double* rate_out) {
  if (nthread < 1) nthread = 1;
  if (nthread > n_env) nthread = n_env;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthread);
  om_job* jobs = (om_job*)calloc(nthread, sizeof(om_job));
  for (int i = 0; i < nthread; i++) {
    jobs[i].m = m; jobs[i].nstep = nstep; jobs[i].t_pre = t_pre; jobs[i].env_offset = env_offset; jobs[i].qpos_out = qpos_out;
    jobs[i].e0 = (int)((long long)n_env * i / nthread);
    jobs[i].e1 = (int)((long long)n_env * (i + 1) / nthread);
    pthread_create(th + i, NULL, rollout_worker, jobs + i);
  }
  long long total = 0;
  double s[5] = {0, 0, 0, 0, 0}, rate = 0;
  for (int i = 0; i < nthread; i++) {
    pthread_join(th[i], NULL);
    total += jobs[i].steps;
    if (jobs[i].seconds > 0) rate += (double)jobs[i].steps / jobs[i].seconds;
    s[0] += jobs[i].stats[0]; s[1] += jobs[i].stats[1]; s[2] += jobs[i].stats[2];
    if (jobs[i].stats[3] > s[3]) s[3] = jobs[i].stats[3];
    if (jobs[i].stats[4] > s[4]) s[4] = jobs[i].stats[4];
  }
  if (stats_out) {
    double n = total > 0 ? (double)total : 1.0;
    stats_out[0] = s[0] / n; stats_out[1] = s[1] / n; stats_out[2] = s[2] / n; stats_out[3] = s[3]; stats_out[4] = s[4];
  }
  if (rate_out) *rate_out = rate;
  free(th); free(jobs);
  return total;
}
long long om_rollout_threads(const om_model* m, int n_env, int nstep, int nthread, int env_offset, double* qpos_out, double* stats_out) {
  return om_rollout_window(m, n_env, 0, nstep, nthread, env_offset, qpos_out, stats_out, NULL);
}

void om_free_model(om_model* m) { free(m); /* arrays intentionally leaked at process end: test-only code */ }

/* test hook: MPR between two free-standing objects (type GEOM_SPHERE / GEOM_CAPSULE / GEOM_MESH; mat row-major; vert = nvert x 3 for
 * a mesh).  Returns ccdMPRPenetration's result (0: intersecting) and fills depth, dir[3], pos[3]. */
int om_mpr_test(int type1, const double* pos1, const double* mat1, const double* size1, const double* vert1, int nvert1,
                int type2, const double* pos2, const double* mat2, const double* size2, const double* vert2, int nvert2,
                double margin, double* depth, double* dir, double* pos) {
  ccd_obj a, b;
  memset(&a, 0, sizeof a); memset(&b, 0, sizeof b);
  a.type = type1; memcpy(a.pos, pos1, sizeof a.pos); memcpy(a.mat, mat1, sizeof a.mat); memcpy(a.size, size1, sizeof a.size); a.vert = vert1; a.nvert = nvert1; a.margin = 0.5 * margin;
  b.type = type2; memcpy(b.pos, pos2, sizeof b.pos); memcpy(b.mat, mat2, sizeof b.mat); memcpy(b.size, size2, sizeof b.size); b.vert = vert2; b.nvert = nvert2; b.margin = 0.5 * margin;
  return mpr_penetration(&a, &b, 50, 1e-6, depth, dir, pos);
}
