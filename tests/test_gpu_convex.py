"""GPU parity of the general collision / constraint path (the reference's own robot, simulation/assets/
world.xml + humanoid.xml: mesh hulls, condim 6, height-field floor, Newton): the HIP kernels through the C-ABI against the
fp64 oracle, teacher-forced one step from states along oracle trajectories.

Tolerances (at most 3x the maxima measured on the MI355X, gpurun_out/r02f/pytest.log): contact distance 5e-7 m, position
3e-6 m, normal 3e-5; qacc 2e-4 and forces 3e-4 relative to their scale, qvel 1.5e-5, qpos 1e-6.  Counts (ncon, nefc)
identical.  The portal search (MPR) runs in double precision on the device as in the oracle (an fp32 search proved
unstable against metre-sized prisms, hb_mpr.hpp), from fp32 poses: a state whose search still ends on another portal (hull against
hull, a handful per few hundred states) is not skipped but PROVED to sit on a rounding fence - the oracle must reproduce the device's
contacts from a state within fp32 rounding of it (oracle_lib.prove_rounding_fence) - and then compared in full; no proof, no pass.
"""
import os

import numpy as np
import pytest

from oracle_lib import ROOT, Oracle
from test_oracle_convex import BALL_MESH, CUBE_MESH, _hfield_xml

pytestmark = pytest.mark.gpu
TEAM_HBM = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "team_robot.hbm")


def _contacts_match(o, dev, nc_dev, ne_dev, tol):
    """oracle contacts (after o.forward()) against one env's device contact records: counts, geoms, dims, distance, position, normal"""
    if (o.ncon, o.nefc) != (nc_dev, ne_dev):
        return None
    w = [0.0, 0.0, 0.0]
    for i, c in enumerate(o.contacts()):
        if (int(dev[i, 14]), int(dev[i, 15]), int(dev[i, 13])) != (c["geom1"], c["geom2"], c["dim"]):
            return None
        dd, dp, dn = abs(dev[i, 0] - c["dist"]), np.abs(dev[i, 1:4] - c["pos"]).max(), np.abs(dev[i, 4:7] - c["frame"][0]).max()
        # A hull - hull contact shallower than 1e-4 m: the portal refinement of MPR stops when the portal is within the search tolerance
        # (1e-6) of the surface, and at such a depth that test can end the device's search (fp32 poses) one refinement away from the
        # oracle's (fp64 poses from the same state): depth within that tolerance, the same position, the normal that of the neighbouring
        # facet of the tessellated surface (to 3e-3).  Counted, and bounded by the caller; every other contact to tol.
        if abs(c["dist"]) < 1e-4 and c["geom1"] != 3 and dn > tol["nrm"] and dn <= 3e-3 and dd <= 1e-6 and dp <= tol["pos"]:
            _contacts_match.shallow_portals = getattr(_contacts_match, "shallow_portals", 0) + 1
            continue
        w = [max(w[0], dd), max(w[1], dp), max(w[2], dn)]
    return w if w[0] <= tol["dist"] and w[1] <= tol["pos"] and w[2] <= tol["nrm"] else None


def _teacher_forced(hbmod, gpu, path, states, ctrls, tol, min_contacts=1, max_divergent=0.02):
    """One device step from each state vs the oracle FROM THE SAME fp32-ROUNDED STATE; returns the worst deviations.  A state whose contact
    set differs (an MPR search ending on another portal, a pair on its margin boundary) is not skipped: it must be PROVED to sit on a
    rounding fence (oracle_lib.prove_rounding_fence: the oracle reproduces the device's contacts from a state within fp32 rounding), and is
    then compared in full against that neighbouring oracle state; without proof the test fails."""
    from oracle_lib import load_state, prove_rounding_fence
    m = hbmod.Model.load(path)
    o = Oracle(path)
    n = len(states)
    states = np.array(states).astype(np.float32).astype(np.float64)  # what hb_set_state leaves on the device
    ctrls = np.array(ctrls, dtype=np.float32).reshape(n, m.nu)
    b = hbmod.Batch(m, n, gpu)
    b.diag_enable(True)
    b.set_state(hbmod.STATE_INTEGRATION, states)
    b.step(ctrls)
    q, v, a = b.qpos.astype(np.float64), b.qvel.astype(np.float64), b.qacc().astype(np.float64)
    f = b.efc_force().astype(np.float64)
    con = b.contacts().astype(np.float64)
    nc, ne, ni = b.counts()
    assert not b.status().any(), b.status()
    worst = dict(qpos=0.0, qvel=0.0, qacc=0.0, force=0.0, dist=0.0, pos=0.0, nrm=0.0)
    seen = fences = 0
    max_nefc = int(ne.max())
    for k in range(n):
        load_state(o, states[k], ctrls[k])
        o.forward()
        sh0 = getattr(_contacts_match, "shallow_portals", 0)
        w = _contacts_match(o, con[k], nc[k], ne[k], tol)
        other_facet = getattr(_contacts_match, "shallow_portals", 0) > sh0  # (a contact of this state carries the neighbouring facet's normal)
        on_fence = w is None
        if on_fence:
            got = {}

            def accept(oo):
                oo.forward()
                got["w"] = _contacts_match(oo, con[k], nc[k], ne[k], tol)
                return got["w"] is not None
            mag = prove_rounding_fence(o, states[k], ctrls[k], accept, seed=k)
            assert mag is not None, ("state %d: device contacts differ from the oracle's and no state within fp32 rounding reproduces them" % k, nc[k], ne[k],
                                     o.ncon, o.nefc, con[k, :nc[k], :7])
            w = got["w"]
            fences += 1
        seen += o.ncon
        worst["dist"], worst["pos"], worst["nrm"] = max(worst["dist"], w[0]), max(worst["pos"], w[1]), max(worst["nrm"], w[2])
        if other_facet:  # its forces act along a normal 1e-3 rad off the oracle's: same step to that accuracy, not to the rounding-level bounds
            assert np.abs(a[k] - o.qacc).max() / max(1.0, np.abs(o.qacc).max()) <= 5e-3
            o.step()
            continue
        worst["qacc"] = max(worst["qacc"], np.abs(a[k] - o.qacc).max() / max(1.0, np.abs(o.qacc).max()))
        if o.nefc:
            fo = o.efc_force[:o.nefc]
            worst["force"] = max(worst["force"], np.abs(f[k, :o.nefc] - fo).max() / max(1.0, np.abs(fo).max()))
        o.step()
        if not on_fence:  # (the neighbouring state differs from the device's in qpos by construction)
            worst["qpos"] = max(worst["qpos"], (np.abs(q[k] - o.qpos) / np.maximum(1.0, np.abs(o.qpos))).max())
        worst["qvel"] = max(worst["qvel"], np.abs(v[k] - o.qvel).max() / max(1.0, np.abs(o.qvel).max()))
    shallow = getattr(_contacts_match, "shallow_portals", 0)
    _contacts_match.shallow_portals = 0
    print("\n%s: %d states, %d contacts, %d states on a PROVED rounding fence (compared against the neighbouring oracle state), %d shallow hull - hull "
          "contacts whose normal is the neighbouring facet's, worst %s" % (os.path.basename(path), n, seen, fences, shallow, {k: "%.2e" % x for k, x in worst.items()}), "max nefc", max_nefc)
    assert seen >= min_contacts and shallow <= max(1, 0.005 * seen), (shallow, seen)
    assert fences <= max(1, max_divergent * n), (fences, n)
    for k, x in worst.items():
        assert x <= tol[k], (k, x, tol[k])
    worst["max_nefc"] = max_nefc
    return worst


TOL = dict(qpos=1e-6, qvel=1.5e-5, qacc=2e-4, force=3e-4, dist=5e-7, pos=3e-6, nrm=3e-5)


def _oracle_states(path, envs, T, every, seed=0, ctrl_scale=1.0, init=None):
    o = Oracle(path)
    rng = np.random.default_rng(seed)
    states, ctrls = [], []
    for e in range(envs):
        o.reset()
        if init is not None:
            init(o, e, rng)
        c = np.zeros(o.nu)
        for t in range(T):
            if t % 25 == 0:
                c = rng.uniform(-1, 1, o.nu) * ctrl_scale
            o.ctrl[:] = c
            if t % every == every - 1:
                states.append(np.concatenate([[o.time], o.qpos, o.qvel, o.qacc_warmstart]))
                ctrls.append(c.copy())
            o.step()
    return states, ctrls


def test_team_robot_model_facts(hbmod):
    m = hbmod.Model.load(TEAM_HBM)
    assert (m.nq, m.nv, m.nu, m.nbody, m.ngeom) == (19, 18, 12, 15, 13)
    assert m.opt.solver == 2 and abs(m.opt.timestep - 0.002) < 1e-12  # Newton (mjOption default), the env's 500 Hz step
    assert (m.ncon_max, m.nefc_max) == (48, 256)


def test_team_robot_one_step_parity_along_oracle_trajectories(hbmod, gpu):
    """The reference's robot from its standing reset through its fall onto the flat height field, random motor commands:
    feet (condim 6, ten rows per contact), then limbs and torso on the floor and against each other."""
    def init(o, e, rng):
        o.qpos[7:] += rng.uniform(-0.2, 0.2, o.nq - 7)  # JOINT_INITIAL_OFFSET_MAX (simulation_parameters.py:22)
        if e % 2:  # the standup task's reset (cpu_env.py:291-328): lying on the floor, root quaternion (-.5, -.5, .5, .5) +- 0.1
            o.qpos[0:3] = [0, 0, -0.6 + 0.1 * rng.uniform()]
            q = np.array([-0.5, -0.5, 0.5, 0.5]) + rng.uniform(-0.1, 0.1, 4)
            o.qpos[3:7] = q / np.linalg.norm(q)
        else:      # standing reset, tipped a little so that it falls
            q = np.array([-0.7, 0, 0, 0.7]) + rng.uniform(-0.08, 0.08, 4)
            o.qpos[3:7] = q / np.linalg.norm(q)
    states, ctrls = _oracle_states(TEAM_HBM, envs=6, T=1200, every=40, seed=1, init=init)  # motors at full swing: flailing, few contacts
    calm = _oracle_states(TEAM_HBM, envs=6, T=800, every=25, seed=2, init=init, ctrl_scale=0.15)  # gentle commands: resting contacts
    states += calm[0]; ctrls += calm[1]
    _teacher_forced(hbmod, gpu, TEAM_HBM, states, ctrls, TOL, min_contacts=300, max_divergent=0.02)


def test_team_robot_staged_fast_pass_against_the_oracle(hbmod, gpu):
    """The same states through the path a training run takes (no diagnostics: pose kernel, narrowphase kernel, one-row-group Newton
    kernel, the four-group kernel only for what that defers): positions, velocities, contact and row counts against the oracle."""
    def init(o, e, rng):
        o.qpos[7:] += rng.uniform(-0.2, 0.2, o.nq - 7)
        o.qpos[0:3] = [0, 0, -0.6 + 0.1 * rng.uniform()]
        q = np.array([-0.5, -0.5, 0.5, 0.5]) + rng.uniform(-0.1, 0.1, 4)
        o.qpos[3:7] = q / np.linalg.norm(q)
    states, ctrls = _oracle_states(TEAM_HBM, envs=8, T=800, every=20, seed=7, init=init, ctrl_scale=0.3)
    from oracle_lib import load_state, prove_rounding_fence
    m = hbmod.Model.load(TEAM_HBM)
    o = Oracle(TEAM_HBM)
    n = len(states)
    states = np.array(states).astype(np.float32).astype(np.float64)
    ctrls = np.array(ctrls, dtype=np.float32).reshape(n, m.nu)
    b = hbmod.Batch(m, n, gpu)
    b.set_state(hbmod.STATE_INTEGRATION, states)
    b.step(ctrls)
    q, v = b.qpos.astype(np.float64), b.qvel.astype(np.float64)
    nc, ne, _ = b.counts()
    assert not b.status().any()
    worst_q = worst_v = 0.0
    fence = contacts = 0

    def dev_vs(oo, k):
        return ((np.abs(q[k] - oo.qpos) / np.maximum(1.0, np.abs(oo.qpos))).max(), np.abs(v[k] - oo.qvel).max() / max(1.0, np.abs(oo.qvel).max()))
    for k in range(n):
        load_state(o, states[k], ctrls[k])
        o.step()
        dq, dv = dev_vs(o, k)
        if (nc[k], ne[k]) != (o.ncon, o.nefc) or dq > TOL["qpos"] or dv > TOL["qvel"]:
            # not skipped: the oracle must reproduce the device's step from a state within fp32 rounding (another MPR portal, a pair on its
            # margin boundary), else this fails
            def accept(oo):
                oo.step()
                return (nc[k], ne[k]) == (oo.ncon, oo.nefc) and dev_vs(oo, k)[1] <= TOL["qvel"]
            mag = prove_rounding_fence(o, states[k], ctrls[k], accept, seed=k)
            assert mag is not None, ("state %d: no state within fp32 rounding reproduces the device's step" % k, nc[k], ne[k], dq, dv)
            fence += 1
            dv = dev_vs(o, k)[1]
            dq = 0.0
        contacts += o.ncon
        worst_q, worst_v = max(worst_q, dq), max(worst_v, dv)
    print("\nteam robot, staged fast pass: %d states, %d contacts, %d states on a PROVED rounding fence, worst qpos %.2e qvel %.2e" % (n, contacts, fence, worst_q, worst_v))
    assert contacts >= 300 and fence <= 0.03 * n


def test_team_robot_free_running_stays_finite(hbmod, gpu):
    """1500 steps of 256 robots from the standup task's reset (lying on the floor) under random motor commands: finite, no
    bad-state flags, unit quaternions, contact and row counts inside the capacity (six to seven contacts, about sixty rows).  (The root link's origin may go below the floor plane when the robot lies on its back - it does
    in the oracle too; free-running GPU and oracle trajectories part after a few hundred steps: tools/gpu_convex_freerun.py.)"""
    m = hbmod.Model.load(TEAM_HBM)
    n, T = 256, 1500
    b = hbmod.Batch(m, n, gpu)
    b.reset(perturb=True)
    st = b.get_state(hbmod.STATE_INTEGRATION)
    rng = np.random.default_rng(5)
    q0 = np.array([-0.5, -0.5, 0.5, 0.5]) + rng.uniform(-0.1, 0.1, (n, 4))
    st[:, 1:4] = [0, 0, -0.6]
    st[:, 3] += 0.1 * rng.uniform(size=n)
    st[:, 4:8] = q0 / np.linalg.norm(q0, axis=1, keepdims=True)
    b.set_state(hbmod.STATE_INTEGRATION, st)
    mx_c = mx_e = 0
    for t in range(T // 50):
        b.step((0.3 * rng.uniform(-1, 1, (n, m.nu))).astype(np.float32), n_substeps=50)
        nc, ne, ni = b.counts()
        mx_c, mx_e = max(mx_c, int(nc.max())), max(mx_e, int(ne.max()))
    q = b.qpos
    s = b.status()
    print("\nteam robot after %d steps: root z %.3f..%.3f, ncon max %d / %d, nefc max %d / %d, newton iterations mean %.2f max %d, flagged %d"
          % (T, q[:, 2].min(), q[:, 2].max(), mx_c, m.ncon_max, mx_e, m.nefc_max, ni.mean(), ni.max(), int((s != 0).sum())))
    assert np.isfinite(q).all() and np.isfinite(b.qvel).all()
    assert not (s & (hbmod.WARN_BADQPOS | hbmod.WARN_BADQVEL | hbmod.WARN_BADQACC)).any()
    assert q[:, 2].min() > -0.9 and q[:, 2].max() < 0.3
    assert np.abs(np.linalg.norm(q[:, 3:7], axis=1) - 1).max() < 1e-4
    assert mx_e >= 55 and mx_c <= m.ncon_max and mx_e <= m.nefc_max  # (rows beyond 64 in a parity test: bumpy_newton above)


def _save(hbmod, xml, tmp_path, name):
    m = hbmod.Model.from_xml_string(xml)
    p = str(tmp_path / name)
    m.save(p)
    return p


def test_primitives_and_hulls_on_a_bumpy_field(hbmod, gpu, tmp_path):
    """Sphere, capsule and two hulls (a cube and a 300-vertex ball) dropped on a bumpy height field, Newton and PGS: every
    collider of the general path against the oracle (mjc_ConvexHField for all four, mjc_Convex between the hulls)."""
    rng = np.random.default_rng(3)
    elev = rng.uniform(0, 1, (6, 6))
    body = ('<body pos="-0.5 0.3 0.45"><freejoint/><geom type="sphere" size="0.08" condim="6"/></body>'
            '<body pos="0.4 -0.4 0.5" euler="20 40 0"><freejoint/><geom type="capsule" size="0.05 0.12" condim="3"/></body>'
            '<body pos="0.1 0.5 0.5" euler="10 20 30"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/><geom type="mesh" mesh="cube" condim="4"/></body>'
            '<body pos="0.12 0.52 0.62"><freejoint/><inertial pos="0 0 0" mass="0.3" diaginertia="0.0005 0.0005 0.0005"/><geom type="mesh" mesh="ball" condim="6" friction="0.7 0.02 0.01"/></body>')
    xml = _hfield_xml(elev, body, nrow=6, ncol=6, size="1 1 0.3 0.2", extra=CUBE_MESH + BALL_MESH)
    for solver, name in ((2, "bumpy_newton.hbm"), (0, "bumpy_pgs.hbm")):
        if solver == 0:  # the one-group PGS instantiation holds 63 rows: friction cones of dimension 3 keep four bodies inside that
            xml = xml.replace('condim="6"', 'condim="3"').replace('condim="4"', 'condim="3"')
        m = hbmod.Model.from_xml_string(xml)
        m.set_opt(solver=solver, iterations=100 if solver == 2 else 50)
        p = str(tmp_path / name)
        m.save(p)
        states, ctrls = _oracle_states(p, envs=1, T=600, every=12)
        w = _teacher_forced(hbmod, gpu, p, states, ctrls, TOL, min_contacts=60, max_divergent=0.02)
        if solver == 2:
            assert w["max_nefc"] > 64  # rows beyond the first group of 64 took part (ten rows per condim-6 contact)


def test_mesh_mesh_stack(hbmod, gpu, tmp_path):
    xml = ('<mujoco><option timestep="0.002"/><asset>%s</asset><worldbody><body pos="0 0 0.05"><inertial pos="0 0 0" mass="1" diaginertia="1 1 1"/><geom type="mesh" mesh="cube"/></body>'
           '<body pos="0.02 -0.01 0.16" euler="5 8 0"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/><geom type="mesh" mesh="cube" condim="6"/></body>'
           '</worldbody></mujoco>' % CUBE_MESH)
    p = _save(hbmod, xml, tmp_path, "stack.hbm")
    states, ctrls = _oracle_states(p, envs=1, T=300, every=10)
    _teacher_forced(hbmod, gpu, p, states, ctrls, TOL, min_contacts=10)


def test_ball_and_capsules_on_a_sloped_field(hbmod, gpu, tmp_path):
    """tests/models/ball_hfield.xml (a ball and capsules on a sloped field, default solver): the prism scheme for primitives."""
    from test_gpu_parity import MODELS
    m = hbmod.Model.load(os.path.join(MODELS, "ball_hfield.xml"))
    p = str(tmp_path / "ball_hfield.hbm")
    m.save(p)
    states, ctrls = _oracle_states(p, envs=1, T=900, every=10)
    _teacher_forced(hbmod, gpu, p, states, ctrls, TOL, min_contacts=50, max_divergent=0.05)


def test_plane_mesh_hulls_on_a_plane(hbmod, gpu, tmp_path):
    """mjc_PlaneConvex on the device against the oracle, teacher-forced: a cube hull and a 300-vertex ball dropped, tilted, on a PLANE
    (up to four contacts per hull: the support vertex and its graph neighbours), Newton with condim 6 and PGS with condim 3; through the
    fused kernel (diagnostics on) and - the robot test below - the staged step."""
    from test_oracle_convex import PLANE_CUBE_XML  # noqa: F401
    body = ('<body pos="0.0 0.0 0.12" euler="20 30 10"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/><geom type="mesh" mesh="cube" condim="%d"/></body>'
            '<body pos="0.3 0.1 0.15"><freejoint/><inertial pos="0 0 0" mass="0.3" diaginertia="0.0005 0.0005 0.0005"/><geom type="mesh" mesh="ball" condim="%d" friction="0.7 0.02 0.01"/></body>')
    for solver, dim, name in ((2, 6, "plane_newton.hbm"), (0, 3, "plane_pgs.hbm")):
        xml = ('<mujoco><option timestep="0.002"/><asset>%s%s</asset><worldbody><geom name="floor" type="plane" pos="0 0 0" size="0 0 .05" condim="3"/>%s</worldbody></mujoco>'
               % (CUBE_MESH, BALL_MESH, body % (dim, dim)))
        m = hbmod.Model.from_xml_string(xml)
        m.set_opt(solver=solver, iterations=100 if solver == 2 else 50)
        p = str(tmp_path / name)
        m.save(p)
        states, ctrls = _oracle_states(p, envs=1, T=600, every=8)
        w = _teacher_forced(hbmod, gpu, p, states, ctrls, TOL, min_contacts=150, max_divergent=0.05)
        assert w["max_nefc"] >= (40 if solver == 2 else 16)  # a hull flat on the plane: four contacts


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(TEAM_HBM), "team_robot_plane.hbm")), reason="assets/team_robot_plane.hbm not built")
def test_team_robot_on_the_reference_plane_floor(hbmod, gpu):
    """The reference's robot on its type="plane" floor (simulation/assets/green_screen_world.xml, compiled to assets/team_robot_plane.hbm):
    the staged step (pose kernel evaluates the plane - hull items, no portal search for them) against the oracle, teacher-forced along the
    robot's fall and rest on the plane."""
    path = os.path.join(os.path.dirname(TEAM_HBM), "team_robot_plane.hbm")

    def init(o, e, rng):
        o.qpos[7:] += rng.uniform(-0.2, 0.2, o.nq - 7)
        o.qpos[2] += 0.05
        q = np.array([-0.7, 0, 0, 0.7]) + rng.uniform(-0.08, 0.08, 4)
        o.qpos[3:7] = q / np.linalg.norm(q)
    states, ctrls = _oracle_states(path, envs=4, T=900, every=15, seed=11, init=init, ctrl_scale=0.2)
    from oracle_lib import load_state, prove_rounding_fence
    m = hbmod.Model.load(path)
    o = Oracle(path)
    n = len(states)
    states = np.array(states).astype(np.float32).astype(np.float64)
    ctrls = np.array(ctrls, dtype=np.float32).reshape(n, m.nu)
    b = hbmod.Batch(m, n, gpu)
    b.set_state(hbmod.STATE_INTEGRATION, states)
    b.step(ctrls)
    q, v = b.qpos.astype(np.float64), b.qvel.astype(np.float64)
    nc, ne, _ = b.counts()
    assert not b.status().any()
    worst_q = worst_v = 0.0
    fence = contacts = 0

    def dev_vs(oo, k):
        return ((np.abs(q[k] - oo.qpos) / np.maximum(1.0, np.abs(oo.qpos))).max(), np.abs(v[k] - oo.qvel).max() / max(1.0, np.abs(oo.qvel).max()))
    for k in range(n):
        load_state(o, states[k], ctrls[k])
        o.step()
        dq, dv = dev_vs(o, k)
        if (nc[k], ne[k]) != (o.ncon, o.nefc) or dq > TOL["qpos"] or dv > TOL["qvel"]:
            def accept(oo):
                oo.step()
                return (nc[k], ne[k]) == (oo.ncon, oo.nefc) and dev_vs(oo, k)[1] <= TOL["qvel"]
            assert prove_rounding_fence(o, states[k], ctrls[k], accept, seed=k) is not None, ("state %d" % k, nc[k], ne[k], o.ncon, o.nefc, dq, dv)
            fence += 1
            dq, dv = 0.0, dev_vs(o, k)[1]
        contacts += o.ncon
        worst_q, worst_v = max(worst_q, dq), max(worst_v, dv)
    print("\nteam robot on the plane floor, staged step: %d states, %d contacts, %d on a PROVED rounding fence, worst qpos %.2e qvel %.2e" % (n, contacts, fence, worst_q, worst_v))
    assert contacts >= 300 and fence <= 0.05 * n


@pytest.mark.timeout(120)
def test_exact_ties_on_flat_facets_terminate_and_match(hbmod, gpu, tmp_path):
    """Regression for the hull-climb hang of round 2 (hb_mpr.hpp: hull_val / hull_tie, kClimbMax): a support direction exactly
    perpendicular to a flat facet makes all of the facet's vertices tie, and a vertex's neighbour list is padded with copies of the vertex
    itself - left to the compiler, the two sides of the tie comparison were once contracted differently, a copy "improved" on its own
    vertex in the last bit and the climb never ended.  Axis-aligned cubes, un-rotated, flat on a plane, flat on a flat height field and
    stacked on each other (every query direction of the first steps is an exact axis): the step returns, and contacts, counts and next
    state equal the oracle's."""
    cube = '<body pos="%s"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/><geom type="mesh" mesh="cube" condim="3"/></body>'
    worlds = {
        "tie_plane.hbm": '<mujoco><option timestep="0.002"/><asset>%s</asset><worldbody><geom type="plane" size="0 0 .05" condim="3"/>%s%s</worldbody></mujoco>'
                         % (CUBE_MESH, cube % "0 0 0.0495", cube % "0 0 0.149"),
        "tie_hfield.hbm": _hfield_xml(np.zeros((4, 4)), cube % "0.125 0.125 0.0495" + cube % "0.125 0.125 0.149", extra=CUBE_MESH),
    }
    for name, xml in worlds.items():
        p = _save(hbmod, xml, tmp_path, name)
        states, ctrls = _oracle_states(p, envs=1, T=40, every=2)
        w = _teacher_forced(hbmod, gpu, p, states, ctrls, TOL, min_contacts=40, max_divergent=0.1)
        assert w["max_nefc"] >= 8


def test_pgs_beyond_63_rows(hbmod, gpu, tmp_path):
    """A condim 4 / 6 model solved by PGS (six / ten pyramid rows per contact): the one-group kernel's 63 rows do not hold it; such a model
    compiles to the kPgsNefcMax-row instantiation (the matrix AR in LDS, lane l owning rows l and l + 64; the staged step runs the one-group
    kernel first and defers what overflows).  Teacher-forced against the oracle: counts, contacts, forces, next state; rows beyond 64 took
    part; no CNSTRFULL anywhere.  Both through the fused big kernel (diagnostics on) and through the staged step with deferral."""
    rng = np.random.default_rng(3)
    elev = rng.uniform(0, 1, (6, 6))
    body = ('<body pos="-0.5 0.3 0.45"><freejoint/><geom type="sphere" size="0.08" condim="6"/></body>'
            '<body pos="0.4 -0.4 0.5" euler="20 40 0"><freejoint/><geom type="capsule" size="0.05 0.12" condim="6"/></body>'
            '<body pos="0.1 0.5 0.5" euler="10 20 30"><freejoint/><inertial pos="0 0 0" mass="0.5" diaginertia="0.001 0.001 0.001"/><geom type="mesh" mesh="cube" condim="4"/></body>'
            '<body pos="0.12 0.52 0.62"><freejoint/><inertial pos="0 0 0" mass="0.3" diaginertia="0.0005 0.0005 0.0005"/><geom type="mesh" mesh="ball" condim="6" friction="0.7 0.02 0.01"/></body>')
    xml = _hfield_xml(elev, body, nrow=6, ncol=6, size="1 1 0.3 0.2", extra=CUBE_MESH + BALL_MESH)
    m = hbmod.Model.from_xml_string(xml)
    m.set_opt(solver=0, iterations=50)
    assert (m.ncon_max, m.nefc_max) == (48, 128)
    p = str(tmp_path / "bumpy_pgs_wide.hbm")
    m.save(p)
    states, ctrls = _oracle_states(p, envs=1, T=600, every=12)
    w = _teacher_forced(hbmod, gpu, p, states, ctrls, TOL, min_contacts=60, max_divergent=0.04)
    assert 64 < w["max_nefc"] <= 128
    # the staged step: one-group fast pass, deferral to the big kernel; against the oracle's next state
    from oracle_lib import load_state
    o = Oracle(p)
    n = len(states)
    st = np.array(states).astype(np.float32).astype(np.float64)
    cs = np.array(ctrls, dtype=np.float32).reshape(n, m.nu)
    b = hbmod.Batch(m, n, gpu)
    b.set_state(hbmod.STATE_INTEGRATION, st)
    b.step(cs)
    q, v = b.qpos.astype(np.float64), b.qvel.astype(np.float64)
    nc, ne, _ = b.counts()
    assert not b.status().any()
    big = 0
    for k in range(n):
        load_state(o, st[k], cs[k])
        o.step()
        if (nc[k], ne[k]) != (o.ncon, o.nefc):
            continue  # (a state on a rounding fence: covered, with proof, by the fused pass above)
        big += int(o.nefc > 63)
        assert (np.abs(q[k] - o.qpos) / np.maximum(1.0, np.abs(o.qpos))).max() <= TOL["qpos"] and np.abs(v[k] - o.qvel).max() / max(1.0, np.abs(o.qvel).max()) <= 3 * TOL["qvel"], k
    assert big >= 3  # env-steps the one-group kernel deferred
