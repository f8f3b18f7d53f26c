"""Host-side model compiler (csrc/mjcf.cpp, setconst.cpp, model_io.cpp) — no GPU needed.

Replaces mj_loadXML / mj_setConst for the hot path's models; checked against hand-derived facts
about the reference's benchmark model and closed forms.
"""
import math
import os

import numpy as np
import pytest

from conftest import REF_HUMANOID_XML
from oracle_lib import HUMANOID_HBM, Oracle, parse_hbm

MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")


def test_humanoid_sizes_and_layout(humanoid_model):
    m = humanoid_model
    # simulation/mujoco/model/humanoid/README.md:4-5 (27 DoF, 21 actuators); SURVEY.md Appendix A.1
    assert (m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.ntendon, m.nM, m.nkey) == (28, 27, 21, 17, 22, 20, 2, 243, 2)
    assert m.nobs == 48
    info = parse_hbm(HUMANOID_HBM)
    assert info["body_name"][1] == "torso" and info["body_name"][6] == "shin_right"
    # dof order per Appendix A.1: 6 root, abdomen_z,y,x, right leg (6), left leg (6), arms
    assert info["jnt_name"][:4] == ["root", "abdomen_z", "abdomen_y", "abdomen_x"]
    assert info["jnt_name"][4:10] == ["hip_x_right", "hip_z_right", "hip_y_right", "knee_right", "ankle_y_right", "ankle_x_right"]
    # actuator order differs from dof order (humanoid.xml:203-223): abdomen_y first, gears 40/40/40/40/40/120/80/20/20
    assert info["actuator_name"][:3] == ["abdomen_y", "abdomen_z", "abdomen_x"]
    assert list(info["actuator_gear"][:9]) == [40, 40, 40, 40, 40, 120, 80, 20, 20]
    assert m.name2id("joint", "knee_left") == 13 and m.name2id("body", "nope") == -1
    # chain depths: feet 15 ancestors incl. self (Appendix A.1)
    madr = info["dof_Madr"]
    assert madr[15] - madr[14] == 15
    # degrees were converted: knee range -160..2 deg
    k = info["jnt_name"].index("knee_right")
    assert np.allclose(info["jnt_range"][2 * k:2 * k + 2], np.radians([-160, 2]))
    # keyframe squat: root z = 0.596 (humanoid.xml:236)
    assert abs(info["key_qpos"][2] - 0.596) < 1e-12
    # 8x floor contact params: condim 3 from the plane, friction max(1,.7)=1, solref mix (.02+.015)/2
    assert info["geom_condim"][0] == 3 and info["geom_condim"][1] == 1


def test_humanoid_mass_properties():
    info = parse_hbm(HUMANOID_HBM)
    mass = info["body_mass"]
    # head: sphere r=.09 at density 1000
    assert abs(mass[2] - 1000 * 4 / 3 * math.pi * 0.09 ** 3) < 1e-9
    # shin: capsule r=.049, half length .15
    r, h = 0.049, 0.15
    assert abs(mass[6] - 1000 * (math.pi * r * r * 2 * h + 4 / 3 * math.pi * r ** 3)) < 1e-9
    # left/right symmetry
    assert np.allclose(mass[5:8], mass[8:11]) and np.allclose(mass[11:14], mass[14:17])
    assert abs(mass.sum() - info["body_subtreemass"][1]) < 1e-9
    inertia = info["body_inertia"].reshape(-1, 3)
    assert (inertia[1:] > 0).all()
    # triangle inequality of principal inertias
    for I in inertia[1:]:
        assert I[0] <= I[1] + I[2] + 1e-12 and I[1] <= I[0] + I[2] + 1e-12 and I[2] <= I[0] + I[1] + 1e-12
    # capsule inertia against numerical integration (shin)
    n = 200000
    rng = np.random.default_rng(0)
    pts = rng.uniform([-r, -r, -h - r], [r, r, h + r], size=(n, 3))
    zc = np.clip(pts[:, 2], -h, h)
    inside = (pts[:, 0] ** 2 + pts[:, 1] ** 2 + (pts[:, 2] - zc) ** 2) <= r * r
    vol_box = (2 * r) * (2 * r) * (2 * h + 2 * r)
    dens = 1000 * vol_box / n
    p = pts[inside]
    Ixx = dens * (p[:, 1] ** 2 + p[:, 2] ** 2).sum()
    Izz = dens * (p[:, 0] ** 2 + p[:, 1] ** 2).sum()
    got = np.sort(inertia[6])
    assert abs(got[2] - Ixx) / Ixx < 0.02 and abs(got[0] - Izz) / Izz < 0.03


def test_setconst_products_match_oracle_recomputation(oracle):
    """dof_M0 / invweight0 / meaninertia (product, setconst.cpp) vs the oracle's own M at qpos0."""
    info = parse_hbm(HUMANOID_HBM)
    oracle.reset()
    oracle.forward()
    M = oracle.dense_M()
    assert np.allclose(np.diag(M), info["dof_M0"], rtol=1e-12)
    assert abs(np.diag(M).mean() - info["meaninertia"]) < 1e-12
    Minv = np.linalg.inv(M)
    dinv = np.diag(Minv)
    w = info["dof_invweight0"]
    assert np.allclose(w[:3], dinv[:3].mean()) and np.allclose(w[3:6], dinv[3:6].mean())
    assert np.allclose(w[6:], dinv[6:], rtol=1e-9)
    # tendon: J M^-1 J^T with J = 0.5 hip_y - 0.5 knee
    J = np.zeros(27)
    J[11], J[12] = 0.5, -0.5
    assert abs(J @ Minv @ J - info["tendon_invweight0"][0]) < 1e-9
    # body_invweight0 of the torso: translational block of J M^-1 J^T at its com
    bw = info["body_invweight0"].reshape(-1, 2)
    assert bw[0].tolist() == [0, 0]
    assert (bw[1:] > 0).all()
    # hands are welded to the lower arms: same rotational weight
    assert np.isclose(bw[12, 1], bw[13, 1])


@pytest.mark.skipif(not os.path.exists(REF_HUMANOID_XML), reason="reference tree not present (GPU box)")
def test_fixture_reproduced_from_reference_xml(hbmod, tmp_path):
    """The committed humanoid27.hbm is exactly what the compiler produces from the reference's MJCF."""
    p = str(tmp_path / "h.hbm")
    m = hbmod.Model.load(REF_HUMANOID_XML)
    assert m.opt.solver == 2 and m.opt.iterations == 100  # the MJCF sets no solver: mjOption's defaults (Newton, 100)
    m.set_opt(solver=0, iterations=50)                    # the benchmark configuration the committed model carries
    m.save(p)
    assert open(p).read() == open(HUMANOID_HBM).read()


def test_hbm_roundtrip_bit_exact(hbmod, tmp_path):
    for name in ("chain", "capsules"):
        a, b = str(tmp_path / (name + "_a.hbm")), str(tmp_path / (name + "_b.hbm"))
        hbmod.Model.load(os.path.join(MODELS, name + ".xml")).save(a)
        hbmod.Model.load(a).save(b)
        assert open(a).read() == open(b).read()


def test_defaults_fromto_and_pairs(hbmod, tmp_path):
    p = str(tmp_path / "chain.hbm")
    m = hbmod.Model.load(os.path.join(MODELS, "chain.xml"))
    m.save(p)
    info = parse_hbm(p)
    # class "link": capsule from fromto, half length .2, radius .03, z axis along -z..+z
    assert info["geom_type"][2] == 3 and np.allclose(info["geom_size"][6:8], [0.03, 0.2])
    assert np.allclose(info["geom_pos"][6:9], [0, 0, -0.2])
    # nested default inheritance: damping/armature from main, stiffness/range from class
    assert np.allclose(info["dof_damping"], 0.1) and np.allclose(info["dof_armature"], 0.01)
    assert np.allclose(info["jnt_stiffness"], [0, 0.5, 0.5, 0.5])
    assert np.allclose(info["jnt_range"][6:8], np.radians([-60, 60]))
    # slide joint range is not an angle
    assert np.allclose(info["jnt_range"][0:2], [-1, 1])
    # parent-child geoms are filtered; cart sphere is on the same body chain as l1 (parent-child)
    pairs = set(zip(info["pair_geom1"], info["pair_geom2"]))
    assert (1, 2) not in pairs and (2, 3) not in pairs and (2, 4) in pairs and (0, 4) in pairs
    # position actuator: gain kp, bias -kp*q
    assert np.allclose(info["actuator_gainprm"], [1, 5, 2])
    assert np.allclose(info["actuator_biasprm"].reshape(3, 3)[1], [0, -5, 0])
    opt = m.opt
    assert opt.timestep == 0.002 and opt.iterations == 50 and opt.solver == 0


def test_multi_tree_weld_and_options(hbmod, tmp_path):
    p = str(tmp_path / "c.hbm")
    m = hbmod.Model.load(os.path.join(MODELS, "capsules.xml"))
    m.save(p)
    info = parse_hbm(p)
    assert list(info["body_rootid"]) == [0, 1, 2, 3]
    assert info["npair"] == 6  # 3 with the plane + 3 among the bodies
    m.set_opt(timestep=0.001, iterations=10, disableflags=16)
    assert m.opt.timestep == 0.001 and m.opt.iterations == 10 and m.opt.disableflags == 16
    m.set_opt(solver=2, ls_iterations=20)  # Newton
    assert m.opt.solver == 2 and m.opt.ls_iterations == 20
    n = hbmod.Model.from_xml_string("<mujoco><option solver='Newton'/><worldbody><body><joint/><geom size='0.1'/></body></worldbody></mujoco>")
    assert n.opt.solver == 2 and n.opt.iterations == 100
    d = hbmod.Model.from_xml_string("<mujoco><worldbody><body><joint/><geom size='0.1'/></body></worldbody></mujoco>")
    assert d.opt.solver == 2 and d.opt.iterations == 100 and d.opt.ls_iterations == 50  # mjOption defaults
    with pytest.raises(hbmod.HbError):
        m.set_opt(solver=1)  # CG is not implemented: refused, not silently ignored


@pytest.mark.parametrize("xml,frag", [
    ("<mujoco><worldbody><body><geom type='box' size='1 1 1'/></body></worldbody></mujoco>", "not supported"),
    ("<mujoco><worldbody><body><joint type='ball'/><geom size='1'/></body></worldbody></mujoco>", "not supported"),
    ("<mujoco><worldbody><body><joint/><geom type='sphere'/></body></worldbody></mujoco>", "size"),
    ("<mujoco><worldbody><body><joint/></body></worldbody></mujoco>", "no mass"),
    ("<mujoco><worldbody><body></worldbody></mujoco>", "mismatched"),
    ("<mujoco><option solver='CG'/><worldbody/></mujoco>", "PGS"),
    ("<notmujoco/>", "root element"),
])
def test_compile_errors_are_reported_not_fatal(hbmod, xml, frag):
    with pytest.raises(hbmod.HbError) as e:
        hbmod.Model.from_xml_string(xml)
    assert frag in str(e.value)


def test_mesh_and_cylinder_geoms_in_the_inertia_of_a_body(hbmod):
    """A mesh geom beside an <inertial> (as every body of the reference's robot has) or with density 0; a cylinder contributes its
    closed-form mass and inertia like the other primitives (m = rho pi r^2 2h, Ixx = m (3 r^2 + (2h)^2) / 12, Izz = m r^2 / 2)."""
    mesh = "<asset><mesh name='m' vertex='0 0 0 1 0 0 0 1 0 0 0 1'/></asset>"
    a = hbmod.Model.from_xml_string("<mujoco>%s<worldbody><body><joint/><inertial pos='0 0 0' mass='1' diaginertia='1 1 1'/><geom type='mesh' mesh='m'/></body></worldbody></mujoco>" % mesh)
    assert a.array("body_mass")[1] == 1.0
    b = hbmod.Model.from_xml_string("<mujoco>%s<worldbody><body><joint/><geom size='0.1'/><geom type='mesh' mesh='m' density='0'/></body></worldbody></mujoco>" % mesh)
    assert abs(b.array("body_mass")[1] - 1000 * 4 / 3 * np.pi * 1e-3) < 1e-12
    c = hbmod.Model.from_xml_string("<mujoco><worldbody><body><joint/><geom type='cylinder' size='0.2 0.5' contype='0' conaffinity='0'/></body></worldbody></mujoco>")
    mass = 1000 * np.pi * 0.04 * 1.0
    assert abs(c.array("body_mass")[1] - mass) < 1e-9
    assert np.allclose(c.array("body_inertia").reshape(-1, 3)[1], [mass * (3 * 0.04 + 1.0) / 12, mass * (3 * 0.04 + 1.0) / 12, mass * 0.04 / 2], rtol=1e-12)


def _mesh_body(hbmod, verts, extra="", geom_attr=""):
    v = " ".join("%r" % float(x) for x in np.asarray(verts, float).ravel())
    xml = ("<mujoco><asset><mesh name='m' vertex='%s'/></asset><worldbody><body pos='0 0 1'><joint type='free'/>%s<geom type='mesh' mesh='m' %s/></body></worldbody></mujoco>"
           % (v, extra, geom_attr))
    return hbmod.Model.from_xml_string(xml)


def _box(lo, hi):
    return [[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])]


def test_mesh_mass_properties_closed_forms(hbmod):
    """MuJoCo derives a mesh geom's mass, inertial frame and inertia from the mesh volume, and stores the mesh about its centre of mass in
    its principal axes (the geoms of simulation/assets/humanoid.xml:22-93, defaults world.xml:18).  csrc/mesh.cpp: signed tetrahedra over
    the hull's faces; closed forms for a cube, an off-centre box, a tetrahedron - volume, centroid and principal moments to 1e-12 - and a
    body WITHOUT <inertial> gets them (it was an error before)."""
    rho = 1000.0
    # cube of edge 0.2 centred at (0.3, -0.1, 0.05) in the mesh file's coordinates
    c0 = np.array([0.3, -0.1, 0.05])
    m = _mesh_body(hbmod, _box(c0 - 0.1, c0 + 0.1))
    mass = rho * 0.2 ** 3
    assert abs(m.array("body_mass")[1] - mass) < 1e-12 * mass
    assert np.allclose(m.array("body_ipos")[3:6], c0, atol=1e-15) and np.allclose(m.array("geom_pos")[0:3], c0, atol=1e-15)
    assert np.allclose(m.array("body_inertia")[3:6], mass * 0.2 ** 2 / 6, rtol=1e-12)
    verts = m.array("mesh_vert").reshape(-1, 3)
    assert len(verts) == 8 and np.allclose(np.sort(np.abs(verts), axis=None), 0.1, atol=1e-15)  # stored about the centre of mass
    # off-centre box 0.4 x 0.2 x 0.1 with its corner at the file's origin: centroid at the half extents, principal moments m (b^2 + c^2) / 12
    # descending - i.e. about z, y, x - and the mesh stored in THAT frame (the long edge along the last principal axis)
    a_, b_, c_ = 0.4, 0.2, 0.1
    m = _mesh_body(hbmod, _box((0, 0, 0), (a_, b_, c_)))
    mass = rho * a_ * b_ * c_
    want = np.array([mass * (a_ ** 2 + b_ ** 2) / 12, mass * (a_ ** 2 + c_ ** 2) / 12, mass * (b_ ** 2 + c_ ** 2) / 12])
    assert abs(m.array("body_mass")[1] - mass) < 1e-12 * mass
    assert np.allclose(m.array("body_ipos")[3:6], [a_ / 2, b_ / 2, c_ / 2], atol=1e-15)
    assert np.allclose(m.array("body_inertia")[3:6], want, rtol=1e-12)
    verts = m.array("mesh_vert").reshape(-1, 3)
    assert np.allclose(np.abs(verts).max(axis=0), [c_ / 2, b_ / 2, a_ / 2], atol=1e-14)
    # the geom frame maps the stored vertices back onto the file's box
    q = m.array("geom_quat")[0:4]
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    back = verts @ R.T + m.array("geom_pos")[0:3]
    assert np.allclose(np.sort(back, axis=0), np.sort(np.array(_box((0, 0, 0), (a_, b_, c_))), axis=0), atol=1e-14)
    # the unit right tetrahedron: volume 1 / 6, centroid (1/4, 1/4, 1/4), inertia tensor about the centroid with eigenvalues 1/80 * (1, 1, 2/... )
    m = _mesh_body(hbmod, [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    mass = rho / 6
    assert abs(m.array("body_mass")[1] - mass) < 1e-12 * mass and np.allclose(m.array("body_ipos")[3:6], 0.25, atol=1e-15)
    # closed form: central second moments of the unit right tetrahedron are V * (3/80) on the diagonal and V * (-1/80) off it, so the
    # inertia tensor is V * [[6, 1, 1], [1, 6, 1], [1, 1, 6]] / 80: eigenvalues V * 8 / 80 (axis (1, 1, 1)) and V * 5 / 80 (twice)
    assert np.allclose(m.array("body_inertia")[3:6], [mass * 8 / 80, mass * 5 / 80, mass * 5 / 80], rtol=1e-12)
    # an explicit mass scales the moments; a second geom on the body composes as for primitives
    m = _mesh_body(hbmod, _box(c0 - 0.1, c0 + 0.1), geom_attr="mass='2'")
    assert m.array("body_mass")[1] == 2.0 and np.allclose(m.array("body_inertia")[3:6], 2.0 * 0.04 / 6, rtol=1e-12)
    m = _mesh_body(hbmod, _box(c0 - 0.1, c0 + 0.1), extra="<geom size='0.05' pos='-0.3 0 0'/>")
    ms = rho * 4 / 3 * np.pi * 0.05 ** 3
    mc = rho * 0.008
    assert abs(m.array("body_mass")[1] - (ms + mc)) < 1e-12
    assert np.allclose(m.array("body_ipos")[3:6], (mc * c0 + ms * np.array([-0.3, 0, 0])) / (ms + mc), atol=1e-14)
    # the free body's mass matrix from the oracle: diag(m, m, m, principal moments) in the inertial frame
    # (qM of a single free body: translation block m I, rotation block the inertia in the body's own frame)


def test_load_errors(hbmod, tmp_path):
    with pytest.raises(hbmod.HbError):
        hbmod.Model.load(str(tmp_path / "missing.xml"))
    bad = tmp_path / "bad.hbm"
    bad.write_text("HBM1\ni nq 3\n")  # truncated: no END
    with pytest.raises(hbmod.HbError):
        hbmod.Model.load(str(bad))


def test_heightfield_asset_and_config5_model(hbmod, tmp_path):
    p = str(tmp_path / "bh.hbm")
    hbmod.Model.load(os.path.join(MODELS, "ball_hfield.xml")).save(p)
    info = parse_hbm(p)
    assert info["nhfield"] == 1 and list(info["hfield_nrow"]) == [3] and list(info["hfield_ncol"]) == [4]
    d = info["hfield_data"].reshape(3, 4)
    assert d.min() == 0.0 and d.max() == 1.0          # normalised like mjModel.hfield_data
    assert np.allclose(d[1], [1.0, 0.7, 0.4, 0.1])   # middle row of the MJCF listing
    assert info["geom_type"][0] == 1 and info["geom_dataid"][0] == 0
    assert set(zip(info["pair_geom1"], info["pair_geom2"])) == {(0, 1), (0, 2), (1, 2)}
    # config-5 model: same humanoid, plane replaced by the 8x8 field of SURVEY.md 8(d), PGS exactly 50 sweeps
    from oracle_lib import ROOT
    h = parse_hbm(os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27_hfield.hbm"))
    base = parse_hbm(HUMANOID_HBM)
    assert h["geom_type"][0] == 1 and list(h["hfield_size"]) == [10, 10, 1, 1] and h["tolerance"] == 0 and h["iterations"] == 50
    assert np.array_equal(h["body_mass"], base["body_mass"]) and np.array_equal(h["pair_geom1"], base["pair_geom1"])
    assert abs((h["hfield_data"] * 1.0).mean() + h["geom_pos"][2]) < 1e-12  # mean terrain height 0
    assert h["hfield_data"].max() <= 0.1


def _mutate_hbm(tmp_path, fn):
    from oracle_lib import HUMANOID_HBM
    lines = open(HUMANOID_HBM).read().splitlines()
    out = fn(lines)
    p = tmp_path / "mutated.hbm"
    p.write_text("\n".join(out) + "\n")
    return str(p)


def _set_record(lines, name, values):
    kind = [ln.split()[0] for ln in lines if len(ln.split()) > 1 and ln.split()[1] == name][0]
    return [("%s %s %d %s" % (kind, name, len(values), " ".join(str(v) for v in values))) if (len(ln.split()) > 1 and ln.split()[1] == name) else ln for ln in lines]


@pytest.mark.parametrize("case", ["short_array", "bad_actuator_joint", "bad_pair_geom", "bad_geom_body", "bad_parent", "bad_madr", "huge_count", "bad_tendon_wrap", "bad_key"])
def test_edited_or_truncated_hbm_is_an_error_not_a_fault(hbmod, tmp_path, case):
    """A damaged compiled model must come back as an error string from hb_model_load: every array length is checked against
    its size field and every id / address against its range before build_device_model or a kernel indexes with it."""
    from oracle_lib import HUMANOID_HBM, parse_hbm
    info = parse_hbm(HUMANOID_HBM)

    def edit(lines):
        if case == "short_array":
            return _set_record(lines, "jnt_axis", list(info["jnt_axis"][:-3]))
        if case == "bad_actuator_joint":
            v = list(info["actuator_trnid"]); v[3] = 999
            return _set_record(lines, "actuator_trnid", v)
        if case == "bad_pair_geom":
            v = list(info["pair_geom2"]); v[-1] = 64
            return _set_record(lines, "pair_geom2", v)
        if case == "bad_geom_body":
            v = list(info["geom_bodyid"]); v[5] = -2
            return _set_record(lines, "geom_bodyid", v)
        if case == "bad_parent":
            v = list(info["body_parentid"]); v[4] = 9
            return _set_record(lines, "body_parentid", v)
        if case == "bad_madr":
            v = list(info["dof_Madr"]); v[10] += 1
            return _set_record(lines, "dof_Madr", v)
        if case == "huge_count":
            return [("I body_parentid 99999999999 0 0") if ln.startswith("I body_parentid ") else ln for ln in lines]
        if case == "bad_tendon_wrap":
            v = list(info["tendon_adr"]); v[1] = 3
            return _set_record(lines, "tendon_adr", v)
        if case == "bad_key":
            return _set_record(lines, "key_qpos", list(info["key_qpos"][:-1]))
        raise AssertionError(case)

    path = _mutate_hbm(tmp_path, edit)
    with pytest.raises(hbmod.HbError) as ei:
        hbmod.Model.load(path)
    assert len(str(ei.value)) > 10
