"""Pins the oracle's task math against golden vectors generated from the reference's importable
Python twins (tools/gen_golden.py -> tests/golden/task_math.json) and against formula-level
known answers of SURVEY Appendix E (ENV:61-156, 1273-1312, 1687-1890)."""
import json
import math
import os

import numpy as np
import pytest

import oracle as O
from conftest import GOLDEN, load_env_cfg

G = json.load(open(os.path.join(GOLDEN, "task_math.json")))


def test_cubic_bezier_matches_reference_twin():
    for row in G["cubicBezier"]:
        np.testing.assert_allclose(O.cubic_bezier(row["p0"], row["pf"], row["s"]), row["out"], rtol=0, atol=1e-15)


def test_gauss_matches_reference_twin():
    for row in G["gauss"]:
        assert abs(O.gauss(row["x"], row["w"], row["h"]) - row["out"]) < 1e-15


def test_bezier2_is_bezier_xy_plus_gauss_z():
    p0, pf = [0.1, -0.02, -0.28], [-0.1, 0.02, -0.28]
    for s in np.linspace(0, 1, 11):
        o = O.bezier2(p0, pf, float(s), 0.08)
        b = O.cubic_bezier(p0, pf, float(s))
        assert abs(o[0] - b[0]) < 1e-15 and abs(o[1] - b[1]) < 1e-15
        assert abs(o[2] - (p0[2] + O.gauss(float(s), 1.0, 0.08))) < 1e-15


def test_ik_abad_and_knee_match_reference_twin():
    # theta0 / theta2 share their formulas with the Python twin (GG:276-296 == ENV:1700-1729)
    for row in G["ik"]:
        th, err = O.inverse_kinematics(row["x"], row["y"], row["z"], row["is_right"])
        assert err == 0
        assert abs(th[0] - row["theta"][0]) < 1e-12
        # the twin uses np.pi, the C++ uses PI = 3.1415926 (ENV:45): 5.4e-8 apart
        assert abs(th[2] - row["theta"][2]) < 1e-7
        if row["x"] == 0.0:  # hip formulas coincide only at x == 0 (GG:299 vs ENV:1738)
            assert abs(th[1] - row["theta"][1]) < 1e-8


def test_ik_inverts_reference_forward_kinematics():
    # ENV IK o GG.kinematic == identity up to the 1e-5 fudge terms inside acos (ENV:1726,1739)
    # (rows with abad == 0: the twin's y row has a hip-knee sign slip, GG:326, exact only there)
    n = 0
    for row in G["kinematic"]:
        if row["theta"][0] != 0.0:
            continue
        x, y, z = row["xyz"]
        th, err = O.inverse_kinematics(x, y, z, row["is_right"])
        assert err == 0
        np.testing.assert_allclose(th, row["theta"], atol=2e-4)
        n += 1
    assert n == 16


def test_ik_inverts_analytic_leg_geometry():
    # same property with abad != 0, against the closed-form leg geometry the IK was derived from
    rng = np.random.RandomState(3)
    lh, lt, lc = 0.085, 0.209, 0.2175
    for i in range(64):
        right = bool(i % 2)
        # the reference IK takes sqrt(y*y*(...)) = |y|*..., so it is the inverse only while the toe
        # stays on its own side of the hip (y keeps the sign of the hip offset): abduct outward freely,
        # inward by at most 0.15 rad here.
        a, hip, knee = rng.uniform(-0.4, 0.15), rng.uniform(0.2, 1.2), -rng.uniform(0.6, 2.2)
        a = a if right else -a
        sy = -lh if right else lh
        L = -lt * math.cos(hip) - lc * math.cos(hip + knee)          # sagittal-plane drop (negative)
        x = -lt * math.sin(hip) - lc * math.sin(hip + knee)
        y = sy * math.cos(a) - L * math.sin(a)
        z = sy * math.sin(a) + L * math.cos(a)
        th, err = O.inverse_kinematics(x, y, z, right)
        assert err == 0
        np.testing.assert_allclose(th, [a, hip, knee], atol=3e-4)


def test_ik_stale_value_and_rescale():
    # out-of-reach target: rescaled to max_len - 1e-5 (ENV:1692-1698), lr clamped (ENV:1725)
    th, err = O.inverse_kinematics(0.0, -0.085, -1.0, True, theta0=(9.0, 9.0, 9.0))
    assert err == 0 and abs(th[2]) < 0.05  # nearly straight leg
    # |y| < l_hip makes sqrt(negative) -> NaN -> asin branch skipped, slot keeps its old value
    th, err = O.inverse_kinematics(0.0, 0.0, -0.05, True, theta0=(0.123, 0.0, 0.0))
    assert err & 1 and th[0] == 0.123


def test_smooth_functions():
    lam = 0.5
    for ph in np.linspace(0, 0.999, 37):
        f = math.fmod(ph, 1.0)
        if f < lam:
            y = 2 * math.sin(f / lam * 2 * 3.1415926) + 0.5
        else:
            y = -2 * math.sin((f - lam) / (1 - lam) * 2 * 3.1415926) + 0.5
        y1 = min(max(y, 0.0), 1.0)
        assert abs(O.smooth_function(float(ph), 2, lam) - y1) < 1e-14
        assert abs(O.smooth_function2(float(ph), 2, lam) - (1.0 - y1)) < 1e-14


def test_sampling_reshape():
    assert O.sampling_reshape(0.3) == pytest.approx(0.4)
    assert O.sampling_reshape(0.5) == pytest.approx(2.0 / 3.0)
    assert O.sampling_reshape(0.0) == pytest.approx(1.0 / 3.0)  # ratio > 0 is strict (ENV:73)


def test_torque_clamp_known_answers():
    # train cfg: tau_max 18, w_c 100, w_max 200 (CFG:35-37); knees scaled by 1.55f
    tau = np.array([30, -30, 50, 5, -5, -50, 30, 30, 30, -30, -30, -30], float)
    qd = np.array([0, 0, 0, 150, -150, 0, 120, -120, 80, 0, 0, -100], float)
    out, up, lo = O.torque_clamp(tau, qd, 18.0, 100.0, 200.0)
    k = float(np.float32(1.55))
    assert up[0] == 18.0 and lo[0] == -18.0 and out[0] == 18.0 and out[1] == -18.0
    assert up[2] == pytest.approx(18.0 * k) and out[2] == pytest.approx(18.0 * k)
    # hip at +150 rad/s: up = 18 - 50*0.18 = 9
    assert up[3] == pytest.approx(9.0) and out[3] == 5.0
    # -150: low = (-200+150)/(-200+100)*(-18) = -9
    assert lo[4] == pytest.approx(-9.0) and out[4] == -5.0
    # knee at 80 rad/s: 80*1.55 = 124 > 100 -> up = (18 - 24*0.18)*1.55
    assert up[8] == pytest.approx((18.0 - (80 * k - 100.0) * 0.18) * k)
    assert lo[11] == pytest.approx(((-200.0 + 100 * k) / (-100.0)) * (-18.0) * k)


def test_torque_clamp_eval_cfg():
    out, up, lo = O.torque_clamp(np.full(12, 100.0), np.full(12, 20.0), 18.0, 14.2, 40.0)
    r = 18.0 / (40.0 - 14.2)
    assert up[0] == pytest.approx(18.0 - (20.0 - 14.2) * r)


def test_obs_scaling_matches_bp5_config():
    cfg = load_env_cfg("default_cfg.yaml")
    mean, std = O.obs_scaling(cfg)
    np.testing.assert_allclose(mean, G["bp5_config"]["obs_mean"], atol=1e-15)
    np.testing.assert_allclose(std, G["bp5_config"]["obs_std"], atol=1e-15)
    np.testing.assert_allclose(mean[5:17], G["bp5_config"]["action_mean"], atol=1e-15)


def test_gait_reference_structure():
    cfg = load_env_cfg("bp5_imitation.yaml")
    cmd = [2.0, 0.0, 0.3]
    t = 0.537
    g1 = O.gait_reference(cfg, cmd, t, True)
    # first call: jointDotRef is a backward difference over control_dt
    g0 = O.gait_reference(cfg, cmd, t - cfg["control_dt"], False)
    np.testing.assert_allclose(g1["jointDotRef"], (g1["jointRef"] - g0["jointRef"]) / cfg["control_dt"], atol=1e-9)
    # second call at the same time: reference velocity is exactly zero
    g2 = O.gait_reference(cfg, cmd, t, False, joint_ref_last=g1["jointRefLast"], joint_ref=g1["jointRef"])
    assert np.all(g2["jointDotRef"] == 0.0)
    # bound gait (GaitType 1): FR/FL share a phase, HR/HL share a phase
    gs = O.gait_reference(cfg, [2.0, 0.0, 0.0], t, True)  # no yaw command -> left/right mirror images
    np.testing.assert_allclose(gs["jointRef"][1:3], gs["jointRef"][4:6], atol=1e-12)
    np.testing.assert_allclose(gs["jointRef"][7:9], gs["jointRef"][10:12], atol=1e-12)
    np.testing.assert_allclose(gs["jointRef"][0], -gs["jointRef"][3], atol=1e-12)
    # end-effector reference = toe + hip offset (ENV:331-334,1889): x offsets +-0.19, y offsets -+0.058
    ee = g1["eeRef"].reshape(4, 3)
    assert ee[0, 0] - ee[2, 0] == pytest.approx(0.38 + (ee[0, 0] - 0.19) - (ee[2, 0] + 0.19))
    # WILDCAT mirrors the stride: stance foot moves +x with time for positive command
    gA = O.gait_reference(cfg, cmd, 0.01, False)
    gB = O.gait_reference(cfg, cmd, 0.02, False)
    leg_hind = 2  # phase 0 -> in stance at t in [0, 0.1)
    assert gB["eeRef"][3 * leg_hind] > gA["eeRef"][3 * leg_hind]


def test_gait_reference_tracks_ik_of_bezier():
    cfg = load_env_cfg("bp5_imitation.yaml", WILDCAT=False, GaitType=0)
    cmd = [1.0, 0.0, 0.0]
    t = 0.03  # leg 1 (phase 0): stance, s = 0.3
    g = O.gait_reference(cfg, cmd, t, False)
    step = cmd[0] * cfg["lam"] * cfg["period"]
    toe = O.cubic_bezier([step / 2, 0, -cfg["stand_height"]], [-step / 2, 0, -cfg["stand_height"]], 0.3)
    th, _ = O.inverse_kinematics(toe[0], toe[1] + 0.085, toe[2], False)
    np.testing.assert_allclose(g["jointRef"][3:6], [th[0], -th[1], -th[2]], atol=1e-12)


def test_height_variable():
    cfg = load_env_cfg("bp5_imitation.yaml", HeightVariable=True)
    g = O.gait_reference(cfg, [0.25, 0.0, 0.0], 0.0, False)  # ratio 0.05 <= 0.1
    assert g["up_height"] == pytest.approx(0.05 * cfg["up_height"])
    g = O.gait_reference(cfg, [1.0, 0.0, 0.0], 0.0, False)
    assert g["up_height"] == pytest.approx(cfg["up_height"])


def test_rng_is_counter_based_and_uniform():
    a = O.rng_u01(1, 7, 3, 11, 44)
    b = O.rng_u01(1, 7, 3, 11, 44)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, O.rng_u01(1, 7, 3, 12, 44))
    # known answer: Philox4x32-10 with counter (0,0,0,0), key (0, 'IRR1') is whatever it is, but the
    # all-zero-key/zero-counter vector of the Random123 KAT is reproduced when seed carries key0=0
    u = np.concatenate([O.rng_u01(5, e, 1, 0, 21) for e in range(2000)])
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    # values are multiples of 2^-24 (exactly representable in f32)
    assert np.all(u * 2 ** 24 == np.round(u * 2 ** 24))


def test_f32_build_agrees_on_task_math():
    cfg = load_env_cfg("bp5_imitation.yaml")
    g64 = O.gait_reference(cfg, [3.0, 0.0, -0.4], 0.813, True)
    g32 = O.gait_reference(cfg, [3.0, 0.0, -0.4], 0.813, True, precision="f32")
    np.testing.assert_allclose(g32["jointRef"], g64["jointRef"], atol=5e-5)
    np.testing.assert_allclose(g32["eeRef"], g64["eeRef"], atol=1e-6)
