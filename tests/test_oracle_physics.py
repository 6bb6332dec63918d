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
