// hb_model.hpp — flat compiled model (host side, fp64), the analogue of mjModel
// (reference layout: simulation/mujoco/include/mujoco/mjmodel.h:558-1087) restricted to the
// features the 27-DoF humanoid path uses.  Field names follow the reference's so that a
// MuJoCo user can read them; storage is std::vector, serialised as the text ".hbm" format
// (model_io.cpp) which the oracle (oracle/mjstep_oracle.c) parses independently.
#pragma once
#include <string>
#include <vector>
#include <map>

namespace hb {

// enums mirror mjmodel.h values so integer codes stay familiar
enum JointType { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };          // mjmodel.h:86-91
enum GeomType { GEOM_PLANE = 0, GEOM_HFIELD = 1, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_ELLIPSOID = 4, GEOM_CYLINDER = 5, GEOM_BOX = 6, GEOM_MESH = 7 };   // mjmodel.h:94-103
enum Solver { SOL_PGS = 0, SOL_CG = 1, SOL_NEWTON = 2 };                                // mjmodel.h:159-163
enum DisableBit {                                                                      // mjmodel.h:50-68
  DSBL_CONSTRAINT = 1 << 0, DSBL_EQUALITY = 1 << 1, DSBL_FRICTIONLOSS = 1 << 2, DSBL_LIMIT = 1 << 3,
  DSBL_CONTACT = 1 << 4, DSBL_PASSIVE = 1 << 5, DSBL_GRAVITY = 1 << 6, DSBL_CLAMPCTRL = 1 << 7,
  DSBL_WARMSTART = 1 << 8, DSBL_FILTERPARENT = 1 << 9, DSBL_ACTUATION = 1 << 10,
  DSBL_REFSAFE = 1 << 11, DSBL_SENSOR = 1 << 12, DSBL_MIDPHASE = 1 << 13, DSBL_EULERDAMP = 1 << 14
};

typedef std::vector<double> vecd;
typedef std::vector<int> veci;

struct Model {
  // ---- sizes (mjmodel.h:560-620)
  int nq = 0, nv = 0, nu = 0, nbody = 0, njnt = 0, ngeom = 0, ntendon = 0, nwrap = 0, nM = 0,
      nkey = 0, nexclude = 0, npair = 0, nhfield = 0, nhfielddata = 0, nmesh = 0, nmeshvert = 0, nmeshnbr = 0;

  // ---- options (mjOption, mjmodel.h:403-445) — only the fields this path honours
  double timestep = 0.002, impratio = 1.0, tolerance = 1e-8;
  double gravity[3] = {0, 0, -9.81};
  int integrator = 0, cone = 0, solver = SOL_NEWTON, iterations = 100, disableflags = 0;  // mjOption defaults (mj_defaultOption)
  int ls_iterations = 50;      // Newton line search: evaluation cap (mjOption.ls_iterations, mjmodel.h:434)
  double ls_tolerance = 0.01;  // and its slope tolerance relative to the main one (mjmodel.h:411)
  double meaninertia = 1.0;  // mjModel.stat.meaninertia (mjmodel.h:547)

  // ---- bodies (mjmodel.h:660-690)
  veci body_parentid, body_rootid, body_weldid, body_jntnum, body_jntadr, body_dofnum, body_dofadr,
      body_geomnum, body_geomadr, body_depth;
  vecd body_pos, body_quat, body_ipos, body_iquat, body_mass, body_subtreemass, body_inertia,
      body_invweight0;
  // ---- joints (mjmodel.h:693-712)
  veci jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid, jnt_limited;
  vecd jnt_pos, jnt_axis, jnt_stiffness, jnt_range, jnt_margin, jnt_solref, jnt_solimp;
  // ---- dofs (mjmodel.h:715-728)
  veci dof_bodyid, dof_jntid, dof_parentid, dof_Madr;
  vecd dof_armature, dof_damping, dof_frictionloss, dof_invweight0, dof_M0;
  // ---- geoms (mjmodel.h:729-760)
  veci geom_type, geom_bodyid, geom_contype, geom_conaffinity, geom_condim, geom_priority,
      geom_dataid;
  vecd geom_size, geom_pos, geom_quat, geom_rbound, geom_friction, geom_solmix, geom_solref,
      geom_solimp, geom_margin, geom_gap;
  // ---- height fields (mjmodel.h:822-829)
  veci hfield_nrow, hfield_ncol, hfield_adr;
  vecd hfield_size, hfield_data;
  // ---- meshes (mjmodel.h:770-800): only what the collision path needs, the vertices of each mesh's CONVEX HULL in the
  // mesh geom's frame (MuJoCo collides mesh geoms through their hulls; mesh.cpp); geom_dataid = mesh id for mesh geoms
  veci mesh_vertadr, mesh_vertnum;
  vecd mesh_vert;
  // the hull's edge graph (mjModel.mesh_graph plays this role): per hull vertex (global index) the run of its neighbours in mesh_nbr,
  // neighbours as vertex indices local to the mesh, ascending: the support function climbs along it instead of sweeping every vertex
  veci mesh_nbradr, mesh_nbrnum, mesh_nbr;
  // ---- fixed tendons (mjmodel.h:950-985): wrap_objid = joint id, wrap_prm = coef
  veci tendon_adr, tendon_num, tendon_limited, wrap_objid;
  vecd tendon_range, tendon_margin, tendon_solref_lim, tendon_solimp_lim, tendon_invweight0,
      tendon_length0, wrap_prm;
  // ---- actuators (mjmodel.h:988-1012); joint transmission only
  veci actuator_trnid, actuator_ctrllimited, actuator_forcelimited;
  vecd actuator_gear, actuator_ctrlrange, actuator_forcerange, actuator_gainprm, actuator_biasprm;
  // ---- contact excludes (mjmodel.h:940-942), as body pairs
  veci exclude_body1, exclude_body2;
  // ---- static collision candidates: geom pairs (g1,g2) that survive the body-level filters of
  // mj_collision (same weld body, parent-child, exclude, contype/conaffinity), type(g1)<=type(g2)
  veci pair_geom1, pair_geom2;
  // ---- reference configurations and keyframes (mjmodel.h:626-627, 1046-1052)
  vecd qpos0, qpos_spring, key_qpos;

  // ---- names
  std::vector<std::string> body_name, jnt_name, geom_name, tendon_name, actuator_name, key_name, mesh_name;

  // Order of the candidate collision pairs = order in which mj_collision emits contacts (it changes nothing physical, but PGS cut at a
  // finite sweep count depends on it: DESIGN.md 2).  1 (what the MJCF compiler writes): body pairs ascending, inside a body pair the geoms of
  // the first body, then those of the second - the structure of MuJoCo's collision driver, which runs its broadphase over BODIES and then
  // walks the geoms of every body pair.  0: geom pairs ascending (rounds 1-3 of this engine; .hbm files without the field).
  int pair_order = 0;
  std::string pair_unsupported;  // compile time only: why a colliding geom pair cannot be simulated (empty: all pairs have colliders)

  // ---- field visitor used by serialisation (model_io.cpp): f(name, member) for every field
  template <class F> void visit(F& f) {
#define HB_F(x) f(#x, x)
    HB_F(nq); HB_F(nv); HB_F(nu); HB_F(nbody); HB_F(njnt); HB_F(ngeom); HB_F(ntendon); HB_F(nwrap);
    HB_F(nM); HB_F(nkey); HB_F(nexclude); HB_F(npair); HB_F(nhfield); HB_F(nhfielddata); HB_F(nmesh); HB_F(nmeshvert); HB_F(nmeshnbr);
    HB_F(timestep); HB_F(impratio); HB_F(tolerance);
    f("gravity", gravity, 3);
    HB_F(integrator); HB_F(cone); HB_F(solver); HB_F(iterations); HB_F(disableflags);
    HB_F(meaninertia); HB_F(ls_iterations); HB_F(ls_tolerance); HB_F(pair_order);
    HB_F(body_parentid); HB_F(body_rootid); HB_F(body_weldid); HB_F(body_jntnum); HB_F(body_jntadr);
    HB_F(body_dofnum); HB_F(body_dofadr); HB_F(body_geomnum); HB_F(body_geomadr); HB_F(body_depth);
    HB_F(body_pos); HB_F(body_quat); HB_F(body_ipos); HB_F(body_iquat); HB_F(body_mass);
    HB_F(body_subtreemass); HB_F(body_inertia); HB_F(body_invweight0);
    HB_F(jnt_type); HB_F(jnt_qposadr); HB_F(jnt_dofadr); HB_F(jnt_bodyid); HB_F(jnt_limited);
    HB_F(jnt_pos); HB_F(jnt_axis); HB_F(jnt_stiffness); HB_F(jnt_range); HB_F(jnt_margin);
    HB_F(jnt_solref); HB_F(jnt_solimp);
    HB_F(dof_bodyid); HB_F(dof_jntid); HB_F(dof_parentid); HB_F(dof_Madr);
    HB_F(dof_armature); HB_F(dof_damping); HB_F(dof_frictionloss); HB_F(dof_invweight0); HB_F(dof_M0);
    HB_F(geom_type); HB_F(geom_bodyid); HB_F(geom_contype); HB_F(geom_conaffinity); HB_F(geom_condim);
    HB_F(geom_priority); HB_F(geom_dataid);
    HB_F(geom_size); HB_F(geom_pos); HB_F(geom_quat); HB_F(geom_rbound); HB_F(geom_friction);
    HB_F(geom_solmix); HB_F(geom_solref); HB_F(geom_solimp); HB_F(geom_margin); HB_F(geom_gap);
    HB_F(hfield_nrow); HB_F(hfield_ncol); HB_F(hfield_adr); HB_F(hfield_size); HB_F(hfield_data);
    HB_F(mesh_vertadr); HB_F(mesh_vertnum); HB_F(mesh_vert); HB_F(mesh_nbradr); HB_F(mesh_nbrnum); HB_F(mesh_nbr);
    HB_F(tendon_adr); HB_F(tendon_num); HB_F(tendon_limited); HB_F(wrap_objid);
    HB_F(tendon_range); HB_F(tendon_margin); HB_F(tendon_solref_lim); HB_F(tendon_solimp_lim);
    HB_F(tendon_invweight0); HB_F(tendon_length0); HB_F(wrap_prm);
    HB_F(actuator_trnid); HB_F(actuator_ctrllimited); HB_F(actuator_forcelimited);
    HB_F(actuator_gear); HB_F(actuator_ctrlrange); HB_F(actuator_forcerange);
    HB_F(actuator_gainprm); HB_F(actuator_biasprm);
    HB_F(exclude_body1); HB_F(exclude_body2); HB_F(pair_geom1); HB_F(pair_geom2);
    HB_F(qpos0); HB_F(qpos_spring); HB_F(key_qpos);
    HB_F(body_name); HB_F(jnt_name); HB_F(geom_name); HB_F(tendon_name); HB_F(actuator_name);
    HB_F(key_name); HB_F(mesh_name);
#undef HB_F
  }
};

// mjcf.cpp — MJCF subset compiler (replaces mj_loadXML, mujoco.h:103)
bool compile_mjcf_file(const std::string& path, Model& m, std::string& err);
bool compile_mjcf_string(const std::string& xml, Model& m, std::string& err);
// mesh.cpp — STL reader and convex hull (the vertices MuJoCo's mesh collision uses)
bool read_stl_vertices(const std::string& path, std::vector<double>& pts, std::string& err);
bool convex_hull_vertices(const std::vector<double>& pts, std::vector<int>& hull, std::string& err, std::vector<int>* tris = nullptr);
// re-orders the candidate pair list in place (Model::pair_order)
void sort_pairs(Model& m, int order);
bool mesh_mass_properties(const std::vector<double>& verts, const std::vector<int>& tris, double& volume, double com[3], double inertia[6]);
// setconst.cpp — mj_setConst products (mujoco.h:221) computed in fp64 on the host
bool set_const(Model& m, std::string& err);
// model_io.cpp — ".hbm" text serialisation (replaces mj_saveModel/mj_loadModel, mujoco.h:159-163)
bool save_hbm(const Model& m, const std::string& path, std::string& err);
bool load_hbm(const std::string& path, Model& m, std::string& err);
bool load_hbm_string(const std::string& text, Model& m, std::string& err);
// every array length against its size field, every id / address against its range (run on every load and compile)
bool validate_model(const Model& m, std::string& err);

}  // namespace hb
