"""First-principles checks of the oracle's rigid-body dynamics + contact (the part of the path whose
reference arithmetic lives in closed-source RaiSim -> "parity unpinned"; these invariants are what
pins the build's own formulation instead).  SURVEY section 4 lists them: total mass 8.88 kg,
M symmetric PD, tau = M qdd + b consistency (momentum / energy), static sum Fz = m g, friction cone,
no penetration."""
import numpy as np
import pytest

import oracle as O
import _np_robot as R
from conftest import load_env_cfg

S = O.S


def test_mass_matrix_equals_kinetic_energy_hessian():
    rng = np.random.RandomState(0)
    for _ in range(3):
        gc = R.random_config(rng)
        M = O.mass_matrix_world(gc)
        Mref = R.mass_matrix(gc)
        assert np.abs(M - M.T).max() < 1e-12
        assert np.linalg.eigvalsh(M).min() > 1e-4
        np.testing.assert_allclose(M, Mref, atol=2e-7)
        np.testing.assert_allclose(M[:3, :3], 8.88 * np.eye(3), atol=1e-12)


def test_toe_kinematics_match_independent_fk():
    rng = np.random.RandomState(1)
    gc = R.random_config(rng)
    gv = rng.normal(size=18)
    pos, vel = O.toe_kinematics(gc, gv)
    np.testing.assert_allclose(pos, R.toe_positions(gc), atol=1e-12)
    eps = 1e-6
    vnum = (R.toe_positions(R.advance(gc, gv, eps)) - R.toe_positions(R.advance(gc, gv, -eps))) / (2 * eps)
    np.testing.assert_allclose(vel, vnum, atol=1e-7)


def _free_flight_env(n=4, seed=5, sim_dt=0.00025):
    # no actuation: Kp = Kd = 0 -> tau = 0 (ENV:762-764 with torque_last = 0)
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=n, Stiffness=0.0, Damping=0.0, simulation_dt=sim_dt)
    env = O.OracleVecEnv(cfg)
    st = env.get_state()
    rng = np.random.RandomState(seed)
    for i in range(n):
        gc = R.random_config(rng, z=0.5)
        rv = rng.uniform(-0.4, 0.4, 3)  # keep the tilt well inside the termination cone (R22 > 0.5)
        gc[3:7] = np.concatenate([[np.cos(np.linalg.norm(rv) / 2)], np.sin(np.linalg.norm(rv) / 2) * rv / np.linalg.norm(rv)])
        st[i, S["GC"]:S["GC"] + 19] = gc
        gv = np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-2, 2, 3), rng.uniform(-6, 6, 12)])
        st[i, S["GV"]:S["GV"] + 18] = gv
        st[i, S["TQL"]:S["TQL"] + 12] = 0
        st[i, S["LAMW"]:S["LAMW"] + 12] = 0
        st[i, S["INCONTACT"]:S["INCONTACT"] + 4] = 0
    env.set_state(st)
    return env, cfg


def _momenta(gc, gv):
    M = O.mass_matrix_world(gc)
    p = M[0:3] @ gv
    LO = M[3:6] @ gv  # about the base origin, world components
    h = np.array([M[5, 1], M[3, 2], M[4, 0]])  # m * (com - base origin)
    Lc = LO - np.cross(h / M[0, 0], p)
    E = 0.5 * gv @ M @ gv + 9.81 * (M[0, 0] * gc[2] + h[2])
    return p, Lc, E


def _free_flight_errors(sim_dt, steps=15):
    env, cfg = _free_flight_env(sim_dt=sim_dt)
    n = env.n
    st0 = env.get_state()
    ref = [_momenta(st0[i, :19], st0[i, 19:37]) for i in range(n)]
    diss = np.zeros(n)
    for k in range(steps):
        st_a = env.get_state()
        _, _, done, _ = env.step(np.zeros((n, 12), np.float32))
        assert not done.any()
        st_b = env.get_state()
        qd2 = 0.5 * ((st_a[:, 25:37] ** 2).sum(1) + (st_b[:, 25:37] ** 2).sum(1))
        diss += 0.01 * qd2 * cfg["control_dt"]  # joint damping 0.01 N m s/rad (URDF:56), trapezoid rule
    st1 = env.get_state()
    T = steps * cfg["control_dt"]
    ep, eL, eE = [], [], []
    for i in range(n):
        assert np.all(st1[i, S["INCONTACT"]:S["INCONTACT"] + 4] == 0)
        p, Lc, E = _momenta(st1[i, :19], st1[i, 19:37])
        p0, L0, E0 = ref[i]
        ep.append(np.abs(p - (p0 + np.array([0, 0, -8.88 * 9.81 * T]))).max())
        eL.append(np.abs(Lc - L0).max())
        eE.append(abs(E - E0 + diss[i]))
    return np.array(ep), np.array(eL), np.array(eE)


def test_free_flight_momentum_and_energy_converge_first_order():
    # M(q) a + b(q,u) = tau must reproduce dp/dt = m g, dL_com/dt = 0, dE/dt = -sum d qd^2.
    # Semi-implicit Euler satisfies them up to O(dt): the residual must be small AND halve with dt,
    # which a wrong Coriolis/centrifugal term in b (an O(1) error) cannot do.
    e1 = _free_flight_errors(0.00025)
    e2 = _free_flight_errors(0.000125)
    for name, a, b_, lim in (("p", e1[0], e2[0], 1e-3), ("L", e1[1], e2[1], 1e-3), ("E", e1[2], e2[2], 2e-2)):
        assert a.max() < lim, (name, a)
        if name != "E":  # the energy residual also carries the trapezoid-rule error of the dissipation
            ratio = a / np.maximum(b_, 1e-12)
            assert np.all((ratio > 1.6) & (ratio < 2.5)), (name, ratio)


def test_bias_term_matches_lagrangian_gravity_at_rest():
    # at zero velocity h = -dV/dq: base z force = m g, joint torques = gradient of potential energy
    rng = np.random.RandomState(4)
    gc = R.random_config(rng)
    h = O.nonlinear_world(gc, np.zeros(18))
    assert h[2] == pytest.approx(8.88 * 9.81, abs=1e-9) and abs(h[0]) < 1e-9 and abs(h[1]) < 1e-9
    eps = 1e-6
    for j in range(12):
        gp, gm = gc.copy(), gc.copy()
        gp[7 + j] += eps
        gm[7 + j] -= eps
        (cp, m), (cm, _) = R.com_world(gp), R.com_world(gm)
        dV = m * 9.81 * (cp[2] - cm[2]) / (2 * eps)
        assert h[6 + j] == pytest.approx(dV, abs=1e-6)


def test_standing_force_balance_cone_and_penetration():
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=2, ContactIterations=8)
    env = O.OracleVecEnv(cfg)
    st = env.get_state()
    for i in range(2):  # drop from the nominal stance, zero velocity, no command
        st[i, S["GV"]:S["GV"] + 18] = 0
        st[i, S["GC"] + 7:S["GC"] + 19] = [0, -0.78, 1.57] * 4
        st[i, S["CMD"]:S["CMD"] + 6] = 0
    env.set_state(st)
    fz = []
    for k in range(400):
        _, _, done, _ = env.step(np.zeros((2, 12), np.float32))
        assert not done.any()
        s = env.get_state()
        lam = s[:, S["LAMW"]:S["LAMW"] + 12].reshape(2, 4, 3)
        mu = s[:, S["MATERIAL"]]
        ft = np.linalg.norm(lam[:, :, :2], axis=2)
        assert np.all(lam[:, :, 2] >= 0)
        assert np.all(ft <= mu[:, None] * lam[:, :, 2] * (1 + 1e-9) + 1e-12)
        if k >= 300:
            fz.append(lam[:, :, 2].sum(1) / cfg["simulation_dt"])
            for i in range(2):
                toes = R.toe_positions(s[i, :19])
                assert toes[:, 2].min() - 0.0275 > -2e-3
    fz = np.mean(fz, axis=0)
    np.testing.assert_allclose(fz, 8.88 * 9.81, rtol=0.03)
    assert np.all(np.abs(env.get_state()[:, 2] - 0.29) < 0.05)


def test_more_contact_sweeps_change_little():
    # the fixed sweep count is a build-defined truncation: 6 sweeps vs 40 sweeps over 50 control steps
    outs = []
    for iters in (6, 40):
        cfg = load_env_cfg("bp5_imitation.yaml", num_envs=4, ContactIterations=iters)
        env = O.OracleVecEnv(cfg)
        rng = np.random.RandomState(0)
        for k in range(50):
            env.step(np.clip(0.3 * rng.normal(size=(4, 12)), -1, 1).astype(np.float32))
        outs.append(env.get_state()[:, :37])
    assert np.abs(outs[0] - outs[1])[:, :19].max() < 5e-3


@pytest.mark.parametrize("solver", [3, 1, 2])
def test_contact_impulses_obey_momentum_balance_cone_and_stick_slip(solver):
    """One 0.25 ms substep with the feet on the ground, from sticking (robot at rest) and sliding (1.5 m/s sideways /
    forwards) starts: the stored world-frame contact impulses must (a) account for the change of the total linear
    momentum, p1 - p0 = sum(lambda) - m g dt z, up to the O(dt^2) change of configuration, (b) lie in the Coulomb cone,
    on its boundary and dissipative for sliding feet, (c) leave sticking feet without velocity.  Both per-contact rules: the
    published one (ContactSolver 3 / 1: maximum dissipation on the cone boundary -- its friction is tilted off the anti-slip
    direction through the normal-tangential coupling of the Delassus block, up to ~30 degrees on this robot) and the build's
    first rule (2: within 26 degrees)."""
    dt = 0.00025
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=6, control_dt=dt, simulation_dt=dt, ContactIterations=30, ContactTolerance=0.0,
                       ContactSolver=solver)
    env = O.OracleVecEnv(cfg)
    st = env.get_state()
    slide = [np.zeros(3), np.zeros(3), np.array([1.5, 0, 0]), np.array([0, 1.5, 0]), np.array([-1.0, 1.0, 0]), np.array([2.0, 0.3, 0])]
    for i in range(6):
        st[i, S["GC"]:S["GC"] + 19] = 0
        st[i, S["GC"] + 2] = 0.2890          # nominal stance: toes ~1 mm into the ground
        st[i, S["GC"] + 3] = 1.0
        st[i, S["GC"] + 7:S["GC"] + 19] = [0, -0.78, 1.57] * 4
        st[i, S["GV"]:S["GV"] + 18] = 0
        st[i, S["GV"]:S["GV"] + 3] = slide[i]
        st[i, S["LAMW"]:S["LAMW"] + 12] = 0
        st[i, S["INCONTACT"]:S["INCONTACT"] + 4] = 0
        st[i, S["TQL"]:S["TQL"] + 12] = 0
    env.set_state(st)
    s0 = env.get_state()
    env.step(np.zeros((6, 12), np.float32))
    s1 = env.get_state()
    for i in range(6):
        assert np.all(s1[i, S["INCONTACT"]:S["INCONTACT"] + 4] == 1)
        lam = s1[i, S["LAMW"]:S["LAMW"] + 12].reshape(4, 3)
        mu = s1[i, S["MATERIAL"]]
        p0, _, _ = _momenta(s0[i, :19], s0[i, 19:37])
        p1, _, _ = _momenta(s1[i, :19], s1[i, 19:37])
        resid = p1 - p0 - (lam.sum(0) + np.array([0, 0, -8.88 * 9.81 * dt]))
        assert np.abs(resid).max() < 2e-4 * max(1.0, np.abs(lam).max() / dt) * dt, (i, resid)   # O(dt^2) relative to the forces
        ln, lt = lam[:, 2], np.linalg.norm(lam[:, :2], axis=1)
        assert np.all(ln > 0) and np.all(lt <= mu * ln * (1 + 1e-6) + 1e-12)
        _, vel = O.toe_kinematics(s1[i, :19], s1[i, 19:37])
        vel = np.asarray(vel).reshape(4, 3)
        for f in range(4):
            vt = vel[f, :2]
            if i < 2:      # at rest: sticking, no residual foot velocity, impulse strictly inside the cone
                # (the probe reports the toe-sphere CENTRE: it may still roll about the resting contact point, r = 27.5 mm)
                assert np.linalg.norm(vt) < 5e-4 and abs(vel[f, 2]) < 1e-6 and lt[f] < 0.9 * mu * ln[f]
            else:          # sliding: on the cone boundary, opposing the slip, normal velocity removed
                assert abs(lt[f] - mu * ln[f]) < 1e-6 * ln[f] + 1e-12
                assert np.dot(lam[f, :2], vt) < 0 and np.dot(lam[f, :2], vt) < -(0.9 if solver == 2 else 0.85) * lt[f] * np.linalg.norm(vt)
                assert abs(vel[f, 2]) < 1e-6


def test_trunk_box_corner_contact_first_principles():
    """ENV:242 / URDF:26: the trunk's 0.3 x 0.2 x 0.1 collision box against the ground.  A robot with its legs folded up is put
    low, tilted 55 degrees onto one bottom corner and moving down: after ONE substep (a) the oracle reports corner contacts and
    no toe contact, (b) the touching corner no longer approaches the ground (hard contact at velocity level, restitution above
    the threshold), (c) the total momentum changed by exactly gravity plus an impulse that pushes up and lies inside the
    friction cone about the ground normal, (d) a robot hovering 5 cm higher in the same pose feels nothing."""
    dt = 0.00025
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=2, control_dt=dt, simulation_dt=dt, Stiffness=0.0, Damping=0.0,
                       ContactIterations=40, ContactTolerance=0.0)
    env = O.OracleVecEnv(cfg)
    st = env.get_state()
    d = np.array([0.15, 0.1, 0.0]) / np.hypot(0.15, 0.1)
    axis = np.cross([0.0, 0.0, 1.0], d)
    ang = np.radians(55.0)
    q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])
    for i, z in enumerate((0.165, 0.215)):
        st[i, S["GC"]:S["GC"] + 19] = 0
        st[i, S["GC"] + 2] = z
        st[i, S["GC"] + 3:S["GC"] + 7] = q
        st[i, S["GC"] + 7:S["GC"] + 19] = [-0.6, -2.25, 0.225, 0.6, -2.25, 0.225] * 2   # legs swung up over the back: toes 0.2 m above the ground
        st[i, S["GV"]:S["GV"] + 18] = 0
        st[i, S["GV"] + 2] = -0.8                                       # falling
        st[i, S["LAMW"]:S["LAMW"] + 12] = 0
        st[i, S["INCONTACT"]:S["INCONTACT"] + 4] = 0
        st[i, S["TQL"]:S["TQL"] + 12] = 0
    env.set_state(st)
    s0 = env.get_state()
    h0 = env.box_hits()
    env.step(np.zeros((2, 12), np.float32))
    s1 = env.get_state()
    hits = env.box_hits() - h0
    assert 1 <= hits <= 4                                               # only the low robot's lowest corner(s)
    assert np.all(s1[:, S["INCONTACT"]:S["INCONTACT"] + 4] == 0)        # no toe is involved
    Rm = R.quat_to_rot(s0[0, 3:7]) if hasattr(R, "quat_to_rot") else None
    if Rm is None:
        w, x, y, z = s0[0, 3:7]
        Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                       [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    corners = np.array([[sx * 0.15, sy * 0.1, sz * 0.05] for sx in (1, -1) for sy in (1, -1) for sz in (-1, 1)])
    cw = corners @ Rm.T
    low = int(np.argmin(cw[:, 2]))
    assert s0[0, 2] + cw[low, 2] < 0 and np.all(np.delete(s0[0, 2] + cw[:, 2], low) > -0.02)
    v1 = s1[0, 19:22] + np.cross(s1[0, 22:25], cw[low])
    v0 = s0[0, 19:22] + np.cross(s0[0, 22:25], cw[low])
    assert v0[2] < -0.5 and v1[2] >= -1e-9                              # (b) the corner stopped approaching (restitution 0.2 x 0.8 upwards)
    assert abs(v1[2] - 0.2 * 0.8) < 0.02
    for i in range(2):
        p0, _, _ = _momenta(s0[i, :19], s0[i, 19:37])
        p1, _, _ = _momenta(s1[i, :19], s1[i, 19:37])
        imp = p1 - p0 - np.array([0, 0, -8.88 * 9.81 * dt])              # what the ground gave
        if i == 0:
            mu = s1[i, S["MATERIAL"]]
            assert imp[2] > 0.5 and np.hypot(imp[0], imp[1]) <= mu * imp[2] * (1 + 1e-6) + 1e-4   # (c)
        else:
            assert np.abs(imp).max() < 1e-6                              # (d) free flight


def _sphere_state(st, i, p, v, rad, mass, dyn=1):
    k = S["SPHERE"]
    st[i, k:k + 3] = p
    st[i, k + 3:k + 6] = v
    st[i, k + 6:k + 9] = (rad, mass, dyn)


def test_meteorite_hits_the_trunk_first_principles():
    """Crutial: True (ENV:273-284, 815-861).  A robot in free flight with a steel sphere dropping onto the top face of its trunk
    box, ONE substep: (a) the oracle reports the sphere-trunk contact, (b) robot + sphere momentum (linear, and angular about
    the world origin) changes by gravity only -- the contact impulse is internal, (c) the contact points separate at 0.95 of the
    approach speed ("steel"-"steel": e 0.95, ENV:244), (d) nothing tangential is exchanged (mu 0), (e) a sphere passing 1 cm
    beside the box exchanges nothing."""
    dt = 0.00025
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=2, control_dt=dt, simulation_dt=dt, Stiffness=0.0, Damping=0.0, Crutial=True, CubeNum=1)
    env = O.OracleVecEnv(cfg)
    st = env.get_state()
    rad, mass = 0.09, 1.3
    for i in range(2):
        st[i, S["GC"]:S["GC"] + 19] = 0
        st[i, S["GC"] + 2] = 0.5
        st[i, S["GC"] + 3] = 1.0
        st[i, S["GC"] + 7:S["GC"] + 19] = [-0.1, -0.78, 1.57, 0.1, -0.78, 1.57] * 2
        st[i, S["GV"]:S["GV"] + 18] = 0
        st[i, S["GV"]:S["GV"] + 6] = (0.4, -0.2, -0.3, 0.5, -0.4, 0.3)
        st[i, S["LAMW"]:S["LAMW"] + 12] = 0
        st[i, S["INCONTACT"]:S["INCONTACT"] + 4] = 0
        st[i, S["TQL"]:S["TQL"] + 12] = 0
    # env 0: centre above the top face at (0.08, -0.03), 1 cm of overlap; env 1: beside the +y face, 1 cm clear of it
    _sphere_state(st, 0, (0.08, -0.03, 0.5 + 0.05 + rad - 0.01), (0.7, 0.3, -6.0), rad, mass)
    _sphere_state(st, 1, (0.0, 0.1 + rad + 0.01, 0.5), (0.0, -0.05, -6.0), rad, mass)
    env.set_state(st)
    s0 = env.get_state()
    env.step(np.zeros((2, 12), np.float32))
    s1 = env.get_state()
    assert env.sphere_hits() == 1                                       # (a), (e)
    k = S["SPHERE"]
    g = np.array([0, 0, -9.81])
    # env 1 is the control: the same robot in the same state with the sphere passing beside it -- what differs between the two
    # robots after the substep is the contact impulse alone (the integrator's own O(dt) momentum drift cancels)
    assert np.abs(s0[0, :37] - s0[1, :37]).max() == 0
    assert np.abs(mass * (s1[1, k + 3:k + 6] - s0[1, k + 3:k + 6]) - mass * g * dt).max() < 1e-12     # (e) free fall
    # momenta of the new velocities in the configuration the impulse was applied in (velocity-level solve, positions follow)
    dP_robot = _momenta(s0[0, :19], s1[0, 19:37])[0] - _momenta(s0[1, :19], s1[1, 19:37])[0]
    dP_sphere = mass * (s1[0, k + 3:k + 6] - s0[0, k + 3:k + 6]) - mass * g * dt
    assert np.abs(dP_robot + dP_sphere).max() < 1e-9 and dP_robot[2] < -1.0               # (b) equal and opposite, robot pushed down
    assert np.abs(dP_sphere[:2]).max() < 1e-9                                             # (d) frictionless, normal to the top face (n = z)
    # angular: the impulse acts at the closest box point q (start positions: the solve is at velocity level)
    q_w = np.array([0.08, -0.03, 0.05])                                 # base axes = world axes here
    M0 = O.mass_matrix_world(s0[0, :19])
    dL = M0[3:6] @ (s1[0, 19:37] - s1[1, 19:37])                        # about the base origin
    assert np.abs(dL - np.cross(q_w, dP_robot)).max() < 1e-9, (dL, np.cross(q_w, dP_robot))
    # (c) restitution: normal relative speed of the contact points after = -0.95 x before
    def rel_n(s):
        vr = s[19:22] + np.cross(s[22:25], q_w) - s[k + 3:k + 6]
        return vr[2]
    before, after = rel_n(s0[0]), rel_n(s1[0])
    assert before > 5.0                                                 # trunk point moves up relative to the sphere = approach
    assert abs(after + 0.95 * before) < 1e-6 * before + 1e-9


def test_meteorite_ground_bounce_and_free_fall():
    """the sphere alone: free fall is semi-implicit Euler under g; on the ground it bounces with the env's default material
    (restitution above the threshold speed) and Coulomb friction slows its slide"""
    dt = 0.00025
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=1, control_dt=dt, simulation_dt=dt, Crutial=True, CubeNum=2)
    env = O.OracleVecEnv(cfg)
    st = env.get_state()
    mu, rest, thr = st[0, S["MATERIAL"]:S["MATERIAL"] + 3]
    rad, mass = 0.1, 0.8
    _sphere_state(st, 0, (3.0, 1.0, 0.5), (0.5, 0.0, -3.0), rad, mass)
    env.set_state(st)
    k = S["SPHERE"]
    z, vz = 0.5, -3.0
    bounced = False
    for n in range(900):
        env.step(np.zeros((1, 12), np.float32))
        s = env.get_state()[0]
        if not bounced and z - rad > 0:
            vz -= 9.81 * dt
            z += vz * dt
            assert abs(s[k + 2] - z) < 1e-9 and abs(s[k + 5] - vz) < 1e-9 and abs(s[k + 3] - 0.5) < 1e-12
            v_in = vz
        elif not bounced:
            bounced = True
            assert abs(s[k + 5] - rest * abs(v_in)) < 1e-6 and rest > 0 and abs(v_in) > thr
            # friction impulse mu * normal impulse against the slide
            jn = mass * (s[k + 5] - (v_in - 9.81 * dt))
            assert abs(mass * (0.5 - s[k + 3]) - min(mu * jn, mass * 0.5)) < 1e-6
            break
    assert bounced
    info = env.sphere_info()[0]
    assert np.allclose(info[:3], s[k:k + 3], atol=1e-6) and abs(info[3] - rad) < 1e-7


def test_meteorite_schedule_follows_the_reference():
    """ENV:608-612 (reset parks the sphere above the PREVIOUS base position, sized by the new start time), ENV:731-740 (parked
    every int(5 period / control_dt) frames above the current base, released one control step later with (gv0, gv1, -5)),
    ENV:1423-1436 GetSphereInfo."""
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=3, Crutial=True, CubeNum=6, max_time=10.0)
    env = O.OracleVecEnv(cfg)
    k = S["SPHERE"]
    K = int(5 * cfg["period"] / cfg["control_dt"])
    first = env.get_state()
    # the reset inside init(): the state before it was all zeros
    assert np.allclose(first[:, k:k + 3], [0.05, 0.0, 1.0]) and np.all(first[:, k + 8] == 0)
    env.reset()
    st = env.get_state()
    t0 = st[:, S["T0"]]
    # a later reset: above where the base was BEFORE it, not above the new random start position
    assert np.allclose(st[:, k:k + 3], first[:, 0:3] + [0.05, 0.0, 1.0]) and np.all(st[:, k + 8] == 0)
    assert np.abs(st[:, 0:2] - first[:, 0:2]).min() > 1e-3
    assert np.allclose(st[:, k + 6], (t0 / 5 + 1) * 0.08) and np.allclose(st[:, k + 7], 6 * (t0 / 5 + 0.2))
    info = env.sphere_info()
    assert np.allclose(info[:, :3], st[:, k:k + 3], atol=1e-6) and np.allclose(info[:, 3], st[:, k + 6], atol=1e-7)
    a = np.zeros((3, 12), np.float32)
    prev = st
    env.step(a)                                                        # frame 1: released with the base's horizontal velocity
    s = env.get_state()
    assert np.all(s[:, k + 8] == 1)
    sub = int(round(cfg["control_dt"] / cfg["simulation_dt"]))
    assert np.allclose(s[:, k + 3:k + 5], prev[:, 19:21], atol=1e-12) and np.allclose(s[:, k + 5], -5 - 9.81 * cfg["simulation_dt"] * sub)
    # walk one env to the next parking frame by hand (frames count control steps since the reset, resets restart them)
    st = env.get_state()
    st[:, S["FRAME"]] = K - 1
    env.set_state(st)
    env.step(a)                                                        # frame K-1: still flying
    s1 = env.get_state()
    assert np.all(s1[:, k + 8] == 1)
    base_before = s1[:, 0:3].copy()
    tnow = s1[:, S["T0"]] + K * cfg["control_dt"]
    env.step(a)                                                        # frame K: parked above the base as it was at the start of this step
    s2 = env.get_state()
    live = s2[:, S["FRAME"]] == K + 1                                  # envs that did not terminate inside this step
    assert live.any()
    assert np.all(s2[live, k + 8] == 0) and np.allclose(s2[live, k:k + 3], base_before[live] + [0.05, 0, 1.0], atol=1e-12)
    assert np.allclose(s2[live, k + 6], (tnow[live] / 5 + 1) * 0.08) and np.allclose(s2[live, k + 3:k + 6], 0)
