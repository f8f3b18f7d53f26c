"""Oracle KATs for the convex collision path of the reference's own robot (simulation/assets/world.xml:14-58,
humanoid.xml:4-93): mesh hulls, libccd-MPR restatement, MuJoCo's height-field prism scheme, condim-6 pyramids.

Nothing here can be checked against MuJoCo (parity unpinned); the checks are closed forms and invariants:
MPR against analytic penetrations, the hull against brute force / scipy, a sphere on a tilted planar field against the
sphere-plane formula, a resting hull carrying exactly its weight, pyramid-row structure for condim 4 and 6.
"""
import os

import numpy as np
import pytest

from conftest import REFERENCE
from oracle_lib import Oracle, mpr

TEAM_XML = os.path.join(REFERENCE, "simulation/assets/world.xml")
CUBE = np.array([[x, y, z] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)])


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def test_mpr_against_analytic_penetrations():
    # sphere - sphere
    r, depth, d, x = mpr(dict(type=2, size=[0.5]), dict(type=2, pos=[0.8, 0, 0], size=[0.5]))
    assert r == 0 and abs(depth - 0.2) < 1e-9 and np.allclose(d, [1, 0, 0], atol=1e-9) and np.allclose(x, [0.4, 0, 0], atol=1e-9)
    # any direction, any radii
    rng = np.random.default_rng(0)
    for _ in range(50):
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        r1, r2 = rng.uniform(0.1, 1.0, 2)
        gap = rng.uniform(0.01, 0.9) * min(r1, r2)
        c2 = u * (r1 + r2 - gap)
        r, depth, d, x = mpr(dict(type=2, size=[r1]), dict(type=2, pos=c2, size=[r2]))
        assert r == 0 and abs(depth - gap) < 1e-5 and np.allclose(d, u, atol=2e-3)
        assert np.allclose(x, u * (r1 - gap / 2), atol=2e-3)
    # separated objects: no intersection
    assert mpr(dict(type=2, size=[0.5]), dict(type=2, pos=[1.01, 0, 0], size=[0.5]))[0] == -1
    assert mpr(dict(type=7, vert=CUBE), dict(type=7, pos=[1.1, 0.1, 0.05], vert=CUBE))[0] == -1
    # cube - cube, axis-aligned face contact, and rotated about the contact normal
    for ang in (0.0, 0.3, 0.7):
        r, depth, d, x = mpr(dict(type=7, vert=CUBE), dict(type=7, pos=[0.9, 0.1, 0.05], mat=rot([1, 0, 0], ang), vert=CUBE))
        assert r == 0 and abs(depth - 0.1) < 1e-6 and np.allclose(d, [1, 0, 0], atol=1e-6) and abs(x[0] - 0.45) < 1e-6
    # cube - capsule (axis z, half length 0.3, radius 0.2) standing 0.1 deep in the cube's top face
    r, depth, d, x = mpr(dict(type=7, vert=CUBE), dict(type=3, pos=[0.1, -0.2, 0.9], size=[0.2, 0.3]))
    assert r == 0 and abs(depth - 0.1) < 1e-5 and np.allclose(d, [0, 0, 1], atol=1e-3) and np.allclose(x, [0.1, -0.2, 0.45], atol=1e-3)
    # vertex into face: the cube turned onto a corner, corner 0.05 inside a big slab
    slab = CUBE * np.array([4.0, 4.0, 1.0])
    R = rot([1, -1, 0], np.arccos(1 / np.sqrt(3)))  # body diagonal onto z
    corner = (R @ np.array([-.5, -.5, -.5]))
    assert np.allclose(corner[:2], 0, atol=1e-12)
    pos = np.array([0.3, -0.2, 0.5 - corner[2] - 0.05])
    r, depth, d, x = mpr(dict(type=7, vert=slab), dict(type=7, pos=pos, mat=R, vert=CUBE))
    assert r == 0 and abs(depth - 0.05) < 1e-5 and np.allclose(d, [0, 0, 1], atol=1e-3) and np.allclose(x[:2], [0.3, -0.2], atol=2e-3)


def _hull_of(hbmod, pts):
    xml = '<mujoco><asset><mesh name="m" vertex="%s"/></asset><worldbody><body><freejoint/><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/>' \
          '<geom type="mesh" mesh="m"/></body></worldbody></mujoco>' % " ".join("%.17g" % v for v in np.asarray(pts).reshape(-1))
    m = hbmod.Model.from_xml_string(xml)
    v = m.array("mesh_vert").reshape(-1, 3)
    gp = m.array("geom_pos").reshape(-1, 3)[0]
    w, x, y, z = m.array("geom_quat").reshape(-1, 4)[0]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    # back in the file's coordinates: the hull is stored about its centre of mass in its principal axes, the geom carries that frame
    return v @ R.T + gp, float(m.array("geom_rbound")[0]), gp


def _volume_about(_, cen, hull, pts):
    """the hull's volume as the sum of |tetrahedra| from the frame origin the compiler chose: equals the volume iff that point is inside"""
    tri = pts[hull.simplices]
    return np.abs(np.einsum("ij,ij->i", tri[:, 0] - cen, np.cross(tri[:, 1] - cen, tri[:, 2] - cen))).sum() / 6


def test_convex_hull_matches_scipy_and_brute_force_support(hbmod):
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(1)
    for trial in range(6):
        n = [8, 30, 200, 1000, 3000, 500][trial]
        pts = rng.normal(size=(n, 3)) * rng.uniform(0.05, 2.0, 3)
        if trial == 5:
            pts /= np.linalg.norm(pts, axis=1, keepdims=True)  # every point on the sphere: all of them are hull vertices
        hv, rb, cen = _hull_of(hbmod, pts)
        want = pts[np.sort(ConvexHull(pts).vertices)]
        assert len(hv) == len(want) and np.allclose(np.sort(hv, axis=0), np.sort(want, axis=0), atol=1e-12)
        dirs = rng.normal(size=(200, 3))
        assert np.allclose((hv @ dirs.T).max(0), (pts @ dirs.T).max(0), atol=1e-12)  # the support function is that of the point cloud
        assert abs(rb - np.linalg.norm(hv - cen, axis=1).max()) < 1e-12  # the bounding sphere about the frame origin: the hull's centre of mass
        hull = ConvexHull(pts)
        assert abs(hull.volume - _volume_about(hv[np.argsort(np.lexsort(hv.T))], cen, hull, pts)) < 1e-9 * hull.volume
    # interior and duplicate points are dropped; a cube keeps its 8 corners
    pts = np.vstack([CUBE, CUBE * 0.5, CUBE, [[0, 0, 0]]])
    hv, _, _ = _hull_of(hbmod, pts)
    assert len(hv) == 8 and np.allclose(np.sort(hv, axis=0), np.sort(CUBE, axis=0))


def _climb(verts, adr, num, nbr, d, start=0):
    """the support function's steepest ascent along the hull's edge graph (oracle ccd_support / device hb_mpr.hpp)"""
    tie = np.array([0.41421356237309503, 0.7320508075688772, 1.0])  # exact ties go to a second, generic direction
    cur, bd, bt = start, verts[start] @ d, verts[start] @ tie
    while True:
        best = cur
        for w in nbr[adr[cur]:adr[cur] + num[cur]]:
            v, t = verts[w] @ d, verts[w] @ tie
            if v > bd or (v == bd and t > bt):
                bd, bt, best = v, t, w
        if best == cur:
            return cur
        cur = best


def test_hull_edge_graph_climb_finds_the_support_vertex(hbmod):
    """Climbing along the compiled edge graph from ANY vertex ends on a maximiser of the direction: the hill-climbing support
    function returns what the exhaustive sweep returns (the support VALUE; on a flat facet several vertices tie)."""
    rng = np.random.default_rng(7)
    models = []
    for n in (8, 60, 400):
        pts = rng.normal(size=(n, 3)) * rng.uniform(0.05, 1.0, 3)
        xml = ('<mujoco><asset><mesh name="m" vertex="%s"/></asset><worldbody><body><freejoint/><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/>'
               '<geom type="mesh" mesh="m"/></body></worldbody></mujoco>' % " ".join("%.17g" % v for v in pts.reshape(-1)))
        models.append(hbmod.Model.from_xml_string(xml))
    if os.path.exists(TEAM_XML):
        models.append(hbmod.Model.load(TEAM_XML))
    for m in models:
        vadr, vnum = m.array("mesh_vertadr").astype(int), m.array("mesh_vertnum").astype(int)
        verts = m.array("mesh_vert").reshape(-1, 3)
        nadr, nnum, nbr = m.array("mesh_nbradr").astype(int), m.array("mesh_nbrnum").astype(int), m.array("mesh_nbr").astype(int)
        for k in range(len(vadr)):
            V = verts[vadr[k]:vadr[k] + vnum[k]]
            a, c = nadr[vadr[k]:vadr[k] + vnum[k]], nnum[vadr[k]:vadr[k] + vnum[k]]
            assert (c >= 3).all() and nbr.max() < max(vnum)
            # symmetric graph
            edges = {(u, w) for u in range(len(V)) for w in nbr[a[u]:a[u] + c[u]]}
            assert all((w, u) in edges for (u, w) in edges)
            for _ in range(40):
                d = rng.normal(size=3)
                got = _climb(V, a, c, nbr, d, start=int(rng.integers(len(V))))
                assert abs(V[got] @ d - (V @ d).max()) <= 1e-12 * max(1.0, np.abs(V).max())
            for d in np.vstack([np.eye(3), -np.eye(3)]):  # axis directions: flat facets and ties
                got = _climb(V, a, c, nbr, d)
                assert abs(V[got] @ d - (V @ d).max()) <= 1e-9


def test_degenerate_meshes_are_errors(hbmod):
    for pts in ([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0.3, 0.2, 0]], [[0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]]):
        xml = '<mujoco><asset><mesh name="m" vertex="%s"/></asset><worldbody/></mujoco>' % " ".join(str(v) for p in pts for v in p)
        with pytest.raises(hbmod.HbError):
            hbmod.Model.from_xml_string(xml)


@pytest.mark.skipif(not os.path.exists(TEAM_XML), reason="reference tree not present")
def test_team_robot_compiles_with_its_hulls(hbmod):
    """simulation/assets/world.xml + humanoid.xml: sizes of SURVEY.md Appendix A.2, hull support == raw STL support."""
    import struct
    m = hbmod.Model.load(TEAM_XML)
    assert (m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.nM) == (19, 18, 12, 15, 13, 13, 117)
    gt = m.array("geom_type").astype(int)
    assert list(gt) == [5, 5, 5, 1] + [7] * 9 and (m.array("geom_condim") == 6).all()
    assert abs(m.array("body_mass").sum() - 2.183804) < 1e-5
    # 9 hfield-mesh pairs + the mesh-mesh pairs that survive the 22 excludes and parent-child filtering
    p1, p2 = m.array("pair_geom1").astype(int), m.array("pair_geom2").astype(int)
    assert (p1 == 3).sum() == 9 and len(p1) == 37 and not ((gt[p1] == 5) | (gt[p2] == 5)).any()
    rng = np.random.default_rng(2)
    dirs = rng.normal(size=(300, 3))
    names = ["torso", "left_forearm_pitch_link", "left_knee_pitch_link"]
    vadr, vnum, verts = m.array("mesh_vertadr").astype(int), m.array("mesh_vertnum").astype(int), m.array("mesh_vert").reshape(-1, 3)
    for k, nm in [(0, names[0]), (3, names[1]), (6, names[2])]:
        path = os.path.join(REFERENCE, "simulation/assets/humanoid_urdf", nm + ".stl")
        b = open(path, "rb").read()
        n = struct.unpack("<I", b[80:84])[0]
        raw = np.frombuffer(b[84:], dtype=np.uint8).reshape(n, 50)[:, 12:48].copy().view("<f4").reshape(-1, 3).astype(np.float64)
        hv = verts[vadr[k]:vadr[k] + vnum[k]]
        # the mesh's frame (centre of mass, principal axes of its hull) as a geom at the body origin carries it
        one = hbmod.Model.from_xml_string('<mujoco><asset><mesh name="m" file="%s"/></asset><worldbody><body><freejoint/><inertial pos="0 0 0" mass="1" '
                                          'diaginertia="1 1 1"/><geom type="mesh" mesh="m"/></body></worldbody></mujoco>' % path)
        com = one.array("geom_pos")[0:3]
        w, x, y, z = one.array("geom_quat")[0:4]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        assert np.allclose(one.array("mesh_vert").reshape(-1, 3), hv, atol=0)
        assert np.allclose((hv @ dirs.T).max(0), (((raw - com) @ R) @ dirs.T).max(0), atol=1e-9)  # hull support == raw STL support
        from scipy.spatial import ConvexHull
        hull = ConvexHull(raw)
        tri = raw[hull.simplices]
        cen_vol = (np.einsum("i,ij->j", np.abs(np.einsum("ij,ij->i", tri[:, 0] - raw.mean(0), np.cross(tri[:, 1] - raw.mean(0), tri[:, 2] - raw.mean(0)))),
                             (tri.sum(1) + raw.mean(0)) / 4)) / np.abs(np.einsum("ij,ij->i", tri[:, 0] - raw.mean(0), np.cross(tri[:, 1] - raw.mean(0), tri[:, 2] - raw.mean(0)))).sum()
        assert np.allclose(com, cen_vol, atol=1e-9)  # ... about the centre of mass of the hull (scipy's triangulation of the same hull)
        assert vnum[k] < 400


def _hfield_xml(elev, body, nrow=4, ncol=4, size="2 2 1 0.5", extra=""):
    return ('<mujoco><option timestep="0.002"/><asset><hfield name="h" nrow="%d" ncol="%d" size="%s" elevation="%s"/>%s</asset>'
            '<worldbody><geom name="floor" type="hfield" hfield="h" condim="6" friction="1.5"/>%s</worldbody></mujoco>'
            % (nrow, ncol, size, " ".join("%.17g" % v for v in np.asarray(elev).reshape(-1)), extra, body))


def _oracle_from_xml(hbmod, xml, tmp_path, name="m.hbm"):
    m = hbmod.Model.from_xml_string(xml)
    p = str(tmp_path / name)
    m.save(p)
    return m, Oracle(p)


def test_sphere_on_tilted_planar_field_equals_sphere_on_plane(hbmod, tmp_path):
    """mjc_ConvexHField: a sphere over the interior of one surface triangle of a PLANAR (tilted) field must give exactly the
    sphere-plane contact: dist = signed distance - radius, normal = plane normal, position midway."""
    ncol = nrow = 4
    # elevation rises linearly with the column: MJCF normalises to [0, 1], size z = 1 scales it -> z = (x + 2) / 4 over x in [-2, 2]
    elev = np.tile(np.arange(ncol, dtype=float), (nrow, 1))
    slope = 1.0 / 4.0
    nrm = np.array([-slope, 0, 1.0]); nrm /= np.linalg.norm(nrm)
    radius = 0.1
    for (x, y, pen) in ((0.35, 0.9, 0.02), (-1.7, -0.9, 0.004), (0.9, 0.45, 0.05)):
        surf = np.array([x, y, (x + 2) * slope])
        centre = surf + nrm * (radius - pen)
        body = '<body pos="%.17g %.17g %.17g"><freejoint/><geom type="sphere" size="%g" condim="6"/></body>' % (*centre, radius)
        m, o = _oracle_from_xml(hbmod, _hfield_xml(elev, body), tmp_path)
        o.reset(); o.forward()
        cons = o.contacts()
        assert len(cons) >= 1
        c = min(cons, key=lambda c: c["dist"])
        assert abs(c["dist"] + pen) < 2e-6, (c["dist"], pen)
        assert np.allclose(c["frame"][0], nrm, atol=1e-5)
        # position: on the normal through the sphere's centre, between the two surfaces (MPR reads it off the portal's
        # barycentric coordinates, which include the interior point: it is not exactly the midpoint, in libccd either)
        off = centre - c["pos"]
        along = off @ nrm
        assert radius - pen - 1e-6 <= along <= radius + 1e-6 and np.linalg.norm(off - along * nrm) < 1e-4
        assert c["dim"] == 6 and (c["geom1"], c["geom2"]) == (0, 1)
        # every other prism contact of this sphere (across a diagonal or cell border) is shallower
        assert all(k["dist"] >= c["dist"] - 1e-9 for k in cons)
    # well above the surface: nothing
    body = '<body pos="0.35 0.9 %.17g"><freejoint/><geom type="sphere" size="0.1"/></body>' % ((0.35 + 2) * slope + 0.2)
    _, o = _oracle_from_xml(hbmod, _hfield_xml(elev, body), tmp_path)
    o.reset(); o.forward()
    assert o.ncon == 0


CUBE_MESH = '<mesh name="cube" vertex="%s"/>' % " ".join("%g" % (0.1 * v) for v in CUBE.reshape(-1))


def _ball_points(n=300, r=0.05):
    k = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * k / n)
    th = np.pi * (1 + 5 ** 0.5) * k
    return r * np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], axis=1)


BALL_MESH = '<mesh name="ball" vertex="%s"/>' % " ".join("%.9g" % v for v in _ball_points().reshape(-1))


def test_resting_hull_carries_its_weight_and_condim6_rows(hbmod, tmp_path):
    """A rounded mesh hull (300 vertices on a sphere: a polyhedron with a flat face rocks on the single contact MPR gives per
    prism, in MuJoCo too) dropped on a flat height field (the team robot's floor is one: world.xml:14,58) comes to rest with
    the normal components of its contact forces adding up to m g; every condim-6 contact is ten pyramid rows."""
    body = ('<body pos="0.3 0.2 0.06"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/>'
            '<geom type="mesh" mesh="ball" condim="6" friction="0.6 0.02 0.01"/></body>')
    m, o = _oracle_from_xml(hbmod, _hfield_xml(np.zeros((4, 4)), body, extra=BALL_MESH), tmp_path)
    o.reset()
    for t in range(1500):
        o.step()
    assert o.ncon >= 1 and np.abs(o.qvel[:3]).max() < 2e-3 and np.abs(o.qvel[3:]).max() < 0.05  # (it still rolls a little on its facets)
    assert abs(o.qpos[2] - 0.05) < 2e-3  # resting on the surface z = 0, slightly sunk into the soft contact
    types, ids = o.efc_types()
    assert o.nefc == 10 * o.ncon and (types == 6).all()
    f = o.efc_force[:o.nefc]
    assert (f >= 0).all()
    # every pyramid row is normal +- mu_k * direction_k: the normal force of a contact is the sum of its ten row forces
    assert abs(f.sum() - 0.5 * 9.81) < 0.5 * 9.81 * 1e-2
    for c in o.contacts():
        assert np.allclose(c["frame"][0], [0, 0, 1], atol=1e-6) and c["dim"] == 6
        assert np.allclose(c["friction5"], [1.5, 1.5, 0.02, 0.01, 0.01])  # element-wise maximum of the two geoms' coefficients
    # Jacobian rows: J[2k] + J[2k+1] = 2 J_normal for every direction k, and the three rotational directions differ from the normal row
    nv = o.nv
    J = o.efc_J[:o.nefc * nv].reshape(o.nefc, nv)
    for c in range(o.ncon):
        rows = J[10 * c:10 * c + 10]
        jn = 0.5 * (rows[0] + rows[1])
        for k in range(5):
            assert np.allclose(0.5 * (rows[2 * k] + rows[2 * k + 1]), jn, atol=1e-12)
        spin = 0.5 * (rows[4] - rows[5]) / 0.02
        assert np.allclose(spin[:3], 0, atol=1e-12) and abs(np.linalg.norm(spin[3:6]) - 1) < 1e-9  # pure rotation about the normal, free-joint angular dofs


def test_condim4_has_six_rows(hbmod, tmp_path):
    body = ('<body pos="0 0 0.049"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/>'
            '<geom type="mesh" mesh="cube" condim="4"/></body>')
    xml = _hfield_xml(np.zeros((4, 4)), body, extra=CUBE_MESH).replace('condim="6" friction="1.5"', 'condim="3"')
    _, o = _oracle_from_xml(hbmod, xml, tmp_path)
    o.reset(); o.forward()
    assert o.ncon >= 1 and o.nefc == 6 * o.ncon


def test_mesh_mesh_contact_between_two_bodies(hbmod, tmp_path):
    """mjc_Convex on two hulls: one cube resting on another (welded to the world): contact normal up, depth as placed."""
    xml = ('<mujoco><asset>%s</asset><worldbody><body pos="0 0 0.05"><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/><geom type="mesh" mesh="cube"/></body>'
           '<body pos="0.02 -0.01 0.148"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/><geom type="mesh" mesh="cube" condim="1"/></body>'
           '</worldbody></mujoco>' % CUBE_MESH)
    _, o = _oracle_from_xml(hbmod, xml, tmp_path)
    o.reset(); o.forward()
    assert o.ncon == 1
    c = o.contacts()[0]
    assert abs(c["dist"] + 0.002) < 1e-6 and np.allclose(c["frame"][0], [0, 0, 1], atol=1e-6) and abs(c["pos"][2] - 0.099) < 1e-6
    assert c["dim"] == 3 and o.nefc == 4


# ---- mjc_PlaneConvex: mesh hulls on a PLANE floor (the reference's green_screen_world.xml:30 and empty_world.xml:24 put the robot's
# meshes on one; rl/generate_policy_videos.py builds CPUEnv on it)
PLANE_CUBE_XML = ('<mujoco><option timestep="0.002"/><asset>%s</asset><worldbody><geom name="floor" type="plane" pos="0 0 0" size="0 0 .05" condim="3"/>'
                  '<body pos="%s" %s><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/><geom type="mesh" mesh="cube" condim="%d"/></body>'
                  '</worldbody></mujoco>')


def test_plane_mesh_contacts_at_the_lowest_corners(hbmod, tmp_path):
    """A cube hull (edge 0.1) lying flat, 1 mm into the plane: the support vertex and its graph neighbours in the bottom face, never a top
    vertex (outside the margin); contact i at its vertex moved half the penetration up, distance = the vertex's height, normal = the
    plane's; tilted onto an edge two contacts, onto a corner one."""
    _, o = _oracle_from_xml(hbmod, PLANE_CUBE_XML % (CUBE_MESH, "0.3 -0.2 0.049", "", 3), tmp_path)
    o.reset(); o.forward()
    assert 3 <= o.ncon <= 4 and o.nefc == 4 * o.ncon
    seen = set()
    for c in o.contacts():
        assert abs(c["dist"] + 0.001) < 1e-12 and np.allclose(c["frame"][0], [0, 0, 1], atol=1e-12)
        assert abs(c["pos"][2] + 0.0005) < 1e-12  # half way between the vertex (z = -0.001) and the plane
        corner = (round((c["pos"][0] - 0.3) / 0.05), round((c["pos"][1] + 0.2) / 0.05))
        assert corner in {(-1, -1), (-1, 1), (1, -1), (1, 1)} and corner not in seen
        seen.add(corner)
    # on an edge (rotated 45 degrees about x): the two vertices of that edge; on a corner: one
    d = -np.ones(3) / np.sqrt(3)  # the body diagonal turned onto -z: axis d x (-z), angle acos(d . (-z))
    ax = np.cross(d, [0, 0, -1]); ax /= np.linalg.norm(ax)
    half = 0.5 * np.arccos(1 / np.sqrt(3))
    corner_quat = 'quat="%.17g %.17g %.17g %.17g"' % (np.cos(half), *(np.sin(half) * ax))
    for pose, want in (('euler="45 0 0"', 2), (corner_quat, 1)):
        z = {2: 0.05 * np.sqrt(2), 1: 0.05 * np.sqrt(3)}[want] - 0.001
        _, o = _oracle_from_xml(hbmod, PLANE_CUBE_XML % (CUBE_MESH, "0 0 %.17g" % z, pose, 3), tmp_path)
        o.reset(); o.forward()
        assert o.ncon == want, (pose, o.ncon)
        for c in o.contacts():
            assert abs(c["dist"] + 0.001) < 1e-9
    # above the margin: nothing
    _, o = _oracle_from_xml(hbmod, PLANE_CUBE_XML % (CUBE_MESH, "0 0 0.06", "", 3), tmp_path)
    o.reset(); o.forward()
    assert o.ncon == 0


def test_hull_resting_on_a_plane_carries_its_weight(hbmod, tmp_path):
    """The cube dropped flat on the plane comes to rest level on its bottom corners: sum of the contacts' normal forces = m g (condim 6:
    ten pyramid rows per contact, all non-negative), no residual rocking."""
    _, o = _oracle_from_xml(hbmod, PLANE_CUBE_XML % (CUBE_MESH, "0.1 0.1 0.07", "", 6), tmp_path)
    o.reset()
    for _ in range(2000):
        o.step()
    assert 3 <= o.ncon <= 4 and o.nefc == 10 * o.ncon
    f = o.efc_force[:o.nefc]
    assert (f >= 0).all() and abs(f.sum() - 0.5 * 9.81) < 0.5 * 9.81 * 1e-3
    # (with three of the four bottom corners in the contact set the cube leans, by micrometres, towards the fourth until that one is the
    # lowest: a residual rocking of 5e-4 rad/s)
    assert np.abs(o.qvel[:3]).max() < 1e-4 and np.abs(o.qvel[3:]).max() < 2e-3 and abs(o.qpos[2] - 0.05) < 1e-3
    q = o.qpos[3:7] / np.linalg.norm(o.qpos[3:7])
    assert abs(abs(q[0]) - 1) < 1e-5  # still level


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "simulation/assets/green_screen_world.xml")), reason="reference tree not present")
def test_reference_green_screen_world_compiles_and_stands_on_its_plane(hbmod, tmp_path):
    """simulation/assets/green_screen_world.xml (the path simulation/__init__.py:7-9 exports, rl/generate_policy_videos.py:19): the
    reference's robot on a type="plane" floor.  It compiles, and the uncontrolled robot comes to rest on the plane (a dozen hull - plane
    contacts) with the floor carrying exactly its weight.  (empty_world.xml, the other plane file, holds no robot at all - a <body> outside <worldbody> - and no
    Python of the reference loads it: it has no degree of freedom to step, here or in MuJoCo.)"""
    m = hbmod.Model.load(os.path.join(REFERENCE, "simulation/assets/green_screen_world.xml"))
    assert (m.nq, m.nv, m.nu) == (19, 18, 12) and m.opt.solver == 2
    p = str(tmp_path / "green.hbm")
    m.save(p)
    o = Oracle(p)
    o.reset()
    for _ in range(2000):  # (a foot flips over once more at about step 1200 before everything is at rest)
        o.step()
    assert o.ncon >= 4 and np.abs(o.qvel).max() < 0.01
    mass = o.marr("body_mass").sum()
    normal = sum(o.efc_force[c["efc_address"]:c["efc_address"] + (1 if c["dim"] == 1 else 2 * (c["dim"] - 1))].sum() for c in o.contacts()
                 if o.info["geom_type"][c["geom1"]] == 0)
    assert abs(normal - mass * 9.81) < 0.01 * mass * 9.81, (normal, mass * 9.81)
    with pytest.raises(Exception):
        hbmod.Model.load(os.path.join(REFERENCE, "simulation/assets/empty_world.xml"))


def test_hull_of_a_subdivided_float32_box_is_closed(hbmod):
    """Many coplanar points (every face of a box subdivided 8 x 8, rotated, rounded to float32): the plain incremental build leaves an open
    hull on such input (found by review: a missing edge can stop the device's hill climb at a non-maximal vertex); the compiler verifies
    the hull (closed manifold, Euler's formula, every input point inside) and rebuilds it on joggled points when that fails.  The hull
    must support like the raw point set in every direction, to 2e-6 of the extent (0.6 um here: the tilt of float32-rounded facets), and its
    edge graph must climb to that support."""
    g = np.linspace(-0.5, 0.5, 9)
    pts = np.array([(x, y, z) for x in g for y in g for z in g if max(abs(x), abs(y), abs(z)) > 0.499]) * np.array([0.3, 0.2, 0.1])
    R = np.linalg.qr(np.random.default_rng(0).normal(size=(3, 3)))[0]
    pts = (pts.astype(np.float32).astype(np.float64) @ R.T).astype(np.float32).astype(np.float64)
    xml = ('<mujoco><asset><mesh name="m" vertex="%s"/></asset><worldbody><body><freejoint/><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/>'
           '<geom type="mesh" mesh="m"/></body></worldbody></mujoco>' % " ".join("%.9g" % v for v in pts.reshape(-1)))
    m = hbmod.Model.from_xml_string(xml)
    w, x, y, z = m.array("geom_quat").reshape(-1, 4)[0]
    Rg = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    hv = m.array("mesh_vert").reshape(-1, 3) @ Rg.T + m.array("geom_pos").reshape(-1, 3)[0]  # the geom's frame: centre of mass, principal axes
    assert 8 <= len(hv) <= len(pts)
    nbradr, nbrnum, nbr = m.array("mesh_nbradr").astype(int), m.array("mesh_nbrnum").astype(int), m.array("mesh_nbr").astype(int)
    assert (nbrnum >= 3).all()
    rng = np.random.default_rng(1)
    dirs = np.concatenate([rng.normal(size=(300, 3)), R.T, -R.T])  # random directions and the six face normals (exact ties on a facet)
    for d in dirs:
        d = d / np.linalg.norm(d)
        want = (pts @ d).max()
        assert abs((hv @ d).max() - want) < 2e-6 * 0.3
        cur = 0
        for _ in range(len(hv)):  # steepest ascent along the edge graph from vertex 0, as ccd_support does
            cand = nbr[nbradr[cur]:nbradr[cur] + nbrnum[cur]]
            best = cand[np.argmax(hv[cand] @ d)]
            if hv[best] @ d <= hv[cur] @ d:
                break
            cur = best
        assert abs(hv[cur] @ d - want) < 4e-6 * 0.3
