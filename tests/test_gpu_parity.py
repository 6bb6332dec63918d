"""Parity tests proper: the HIP kernels, called through the C-ABI (libirrl_env.so) by the reference-style
FlexibleGymEnv shim, against the f64 oracle on the same seeded inputs -- on a real MI355X.
Tolerances are stated in parity_lib.py.  Full-size (4096-env) runs use size-independent properties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle as O
import parity_lib as PL
from conftest import load_env_cfg


def _hip(cfg):
    from hip_env import HipVecEnv
    return HipVecEnv(cfg)


def _pair(cfg):
    return O.OracleVecEnv(cfg), _hip(cfg)


def test_init_matches_oracle_train_cfg():
    PL.check_init(*_pair(load_env_cfg("default_cfg.yaml", num_envs=64)))


def test_init_matches_oracle_imitation_cfg_ragged_grid():
    # 37 envs: the last wave has idle quads (store masks), the reference's own 1-env eval case is below
    PL.check_init(*_pair(load_env_cfg("bp5_imitation.yaml", num_envs=37)))


def test_single_env():
    orc, cand = _pair(load_env_cfg("bp5_imitation.yaml", num_envs=1))
    PL.check_init(orc, cand)
    PL.check_teacher_forced(orc, cand, steps=20)


def test_dynamics_probe_matches_oracle():
    orc, cand = _pair(load_env_cfg("default_cfg.yaml", num_envs=48))
    rng = np.random.RandomState(0)
    for _ in range(5):
        orc.step(PL.random_actions(rng, 48))
    st = PL.f32_round_state(orc.get_state())
    orc.set_state(st)
    cand.set_state(st)
    PL.check_probe(orc, cand)


def test_teacher_forced_train_cfg_with_noise_randomisation_and_resets():
    orc, cand = _pair(load_env_cfg("default_cfg.yaml", num_envs=64))
    worst, n_done = PL.check_teacher_forced(orc, cand, steps=150, force_terminal_every=5)
    assert n_done >= 20
    print("teacher-forced worst errors:", worst)


def test_teacher_forced_eval_style_cfg():
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=16, GaitType=0, WILDCAT=False, HeightVariable=True,
                       MotorCriticalSpeed=14.2, MotorMaxSpeed=40, stand_height=0.30, ObsFilter=True, Filter=True,
                       TimeBasedContact=True, ActionNoise=0.1, SharedNoiseScalar=False)
    orc, cand = _pair(cfg)
    PL.check_teacher_forced(orc, cand, steps=60, seed=2, force_terminal_every=9)


def test_free_running_horizons_1_8_400_substeps():
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=64)
    out = PL.check_free_running(O.OracleVecEnv, _hip, cfg)
    print("free-running (pos, vel) errors:", out)


def test_full_size_invariants_and_determinism():
    cfg = load_env_cfg("default_cfg.yaml", num_envs=4096)
    a = _hip(cfg)
    st_a = PL.check_invariants(a, steps=40)
    b = _hip(cfg)
    st_b = PL.check_invariants(b, steps=40)
    assert np.array_equal(st_a, st_b)  # same seed -> bit-identical (counter RNG, no atomics, fixed reduction order)


def test_env_results_do_not_depend_on_batch_size():
    # env i is keyed by (seed, i): the first 64 envs of a 4096-env pool == a 64-env pool, bit for bit
    rng = np.random.RandomState(5)
    acts = [PL.random_actions(rng, 4096) for _ in range(10)]
    big = _hip(load_env_cfg("default_cfg.yaml", num_envs=4096))
    small = _hip(load_env_cfg("default_cfg.yaml", num_envs=64))
    for a in acts:
        ob_b, r_b, d_b, _ = big.step(a)
        ob_s, r_s, d_s, _ = small.step(a[:64].copy())
        assert np.array_equal(ob_b[:64], ob_s) and np.array_equal(r_b[:64], r_s) and np.array_equal(d_b[:64], d_s)


def test_two_shards_with_env_id_offsets_are_the_big_pool_bit_for_bit():
    """SURVEY 8e / VEC:273: rank r of an N-GPU job owns the global env ids r * n .. (r + 1) * n - 1 (`EnvIdOffset`).  Two pools of
    2048 with offsets 0 / 2048 == one pool of 4096, bit for bit over 50 steps (training config: noise, randomised dynamics, command
    process, in-step resets) -- the multi-GPU partitioning is the single-GPU job, by construction."""
    rng = np.random.RandomState(7)
    acts = [PL.random_actions(rng, 4096, 0.5) for _ in range(50)]
    big = _hip(load_env_cfg("default_cfg.yaml", num_envs=4096))
    lo = _hip(load_env_cfg("default_cfg.yaml", num_envs=2048, EnvIdOffset=0))
    hi = _hip(load_env_cfg("default_cfg.yaml", num_envs=2048, EnvIdOffset=2048))
    assert np.array_equal(big.observe()[:2048], lo.observe()) and np.array_equal(big.observe()[2048:], hi.observe())
    n_done = 0
    for k, a in enumerate(acts):
        if k == 10:        # some robots below the termination height: the in-step resets draw from the global-id streams too
            st = big.get_state()
            st[::97, PL.S["GC"] + 2] = 0.14
            big.set_state(st); lo.set_state(st[:2048]); hi.set_state(st[2048:])
        ob_b, r_b, d_b, x_b = big.step(a)
        ob_l, r_l, d_l, x_l = lo.step(a[:2048].copy())
        ob_h, r_h, d_h, x_h = hi.step(a[2048:].copy())
        assert np.array_equal(ob_b[:2048], ob_l) and np.array_equal(ob_b[2048:], ob_h)
        assert np.array_equal(r_b[:2048], r_l) and np.array_equal(r_b[2048:], r_h)
        assert np.array_equal(d_b[:2048], d_l) and np.array_equal(d_b[2048:], d_h)
        assert np.array_equal(x_b[:2048], x_l) and np.array_equal(x_b[2048:], x_h)
        n_done += int(d_b.sum())
    st = big.get_state()
    assert np.array_equal(st[:2048], lo.get_state()) and np.array_equal(st[2048:], hi.get_state())
    assert n_done >= 40


def test_control_dt_setter_moves_the_frame_cadences_like_the_reference():
    """VEC:328-331 setControlTimeStep after construction: the reference evaluates `frame_idx % int(5 * period_ / control_dt_)`
    (meteorite, ENV:733) and `int(period_ / control_dt_ * 10)` (state_disturbance, ENV:747) with the LIVE control_dt_.  A Crutial
    pool (period 0.05: parked every 125 frames at 0.002 s) switched to 0.004 s must park at frame 62 -- engine (C-ABI setter) and
    oracle, teacher-forced across that frame; same for the state_disturbance kick (period 0.02: every 100 -> 50 frames)."""
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=16, Crutial=True, CubeNum=6, period=0.05)
    orc, cand = _pair(cfg)
    orc.set_control_dt(0.004)
    cand.impl.setControlTimeStep(0.004)
    st = orc.get_state()
    st[:, PL.S["FRAME"]] = 55
    orc.set_state(st)
    k = PL.S["SPHERE"]
    parked_at = []
    rng = np.random.RandomState(3)
    for step in range(12):
        s0 = PL.f32_round_state(orc.get_state())
        orc.set_state(s0); cand.set_state(s0)
        a = PL.random_actions(rng, 16, 0.3)
        orc.step(a); cand.step(a)
        so, sc = orc.get_state(), cand.get_state()
        assert np.array_equal(so[:, k + 8], sc[:, k + 8]) and np.abs(so[:, k:k + 8] - sc[:, k:k + 8]).max() < 1e-4
        if (so[:, k + 8] == 0).all() and (s0[:, k + 8] == 1).any():
            parked_at.append(int(s0[0, PL.S["FRAME"]]))
    assert parked_at == [62], parked_at
    cfg = load_env_cfg("default_cfg.yaml", num_envs=8, Manual=True, ForceDisturbance=True, period=0.02, ObsNoise=0.0, ActionNoise=0.0)
    orc, cand = _pair(cfg)
    orc.set_control_dt(0.004)
    cand.impl.setControlTimeStep(0.004)
    st = orc.get_state()
    st[:, PL.S["FRAME"]] = 44
    orc.set_state(st)
    PL.check_teacher_forced(orc, cand, steps=12, seed=9, action_scale=0.1)      # crosses frame 50: both kick, or the states differ


def test_statistical_agreement_over_a_rollout():
    # free-running trajectories diverge (chaos), distributions must not: 256 envs x 300 steps
    cfg = load_env_cfg("default_cfg.yaml", num_envs=256)
    orc, cand = _pair(cfg)
    rng = np.random.RandomState(9)
    R_o, R_c, D_o, D_c = [], [], 0, 0
    for k in range(300):
        a = PL.random_actions(rng, 256, 0.5)
        _, r_o, d_o, _ = orc.step(a)
        _, r_c, d_c, _ = cand.step(a)
        R_o.append(r_o.mean()); R_c.append(r_c.mean()); D_o += d_o.sum(); D_c += d_c.sum()
    assert abs(np.mean(R_o) - np.mean(R_c)) < 0.02 * max(abs(np.mean(R_o)), 0.1)
    assert abs(D_o - D_c) <= max(8, 0.25 * max(D_o, D_c))


def test_device_tensor_path_matches_host_path():
    import torch
    from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import TorchVecEnv
    cfg = load_env_cfg("default_cfg.yaml", num_envs=128)
    host = _hip(cfg)
    dev = TorchVecEnv(_hip(cfg).impl, init=False)
    rng = np.random.RandomState(1)
    for _ in range(12):
        a = PL.random_actions(rng, 128)
        ob_h, r_h, d_h, _ = host.step(a)
        ob_d, r_d, d_d = dev.step(torch.from_numpy(a).cuda())
        assert np.array_equal(ob_h, ob_d.cpu().numpy()) and np.array_equal(r_h, r_d.cpu().numpy())
        assert np.array_equal(d_h, d_d.cpu().numpy())


def test_reference_call_surface():
    from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import RaisimGymVecEnv
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
    cfg = load_env_cfg("default_cfg.yaml", num_envs=32)
    env = RaisimGymVecEnv(FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)))
    assert env.num_envs == 32 and env.num_obs == 35 and env.num_acts == 12 and len(env.extra_info_names) == 6
    ob = env.reset()
    assert ob.shape == (32, 35) and ob.dtype == np.float32
    ob, rew, done, info = env.step(np.zeros((32, 12), np.float32))
    assert rew.shape == (32,) and done.dtype == np.bool_ and len(info) == 32
    assert env.OriginState().shape == (32, 41) and env.GetInverseMassMatrix().shape == (32, 324)
    assert env.GetJointEffort().shape == (32, 12) and env.GetGeneralizedForce().shape == (32, 18)
    # ReferenceState reproduces the reference's dispatch bug (VEC:223-226): first 24 origin-state entries
    np.testing.assert_array_equal(env.ReferenceState(), env.OriginState()[:, :24])
    with pytest.raises(TypeError):
        env.wrapper.step(np.zeros((32, 12), np.float64), env._observation, env._reward, env._done, env._extraInfo)
    ob2, info2 = env.reset_and_update_info()
    assert len(info2) == 32 and "episode" in info2[0]


def test_compiled_pybind_boundary_matches_the_ctypes_shim():
    """The reference's own boundary, compiled (native/_flexible_robot: pybind11 over the C-ABI, raisim_gym.cpp:14-46): numpy
    arrays by reference and filled in place, TypeError on a wrong dtype / layout (no silent copy), ValueError on a wrong shape,
    bit-identical results to the ctypes class over 30 steps with resets, usable as the `impl` of RaisimGymVecEnv."""
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import RaisimGymVecEnv
    from test_abi_surface import REF_METHODS, load_native_module
    mod = load_native_module()
    n = 48
    cfg = load_env_cfg("default_cfg.yaml", num_envs=n)
    text = yaml.safe_dump(cfg)
    nat = mod.FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, text)
    nat.init()
    ref = _hip(cfg)
    assert all(callable(getattr(nat, m)) for m in REF_METHODS)
    assert (nat.getNumOfEnvs(), nat.getObDim(), nat.getActionDim(), nat.getExtraInfoDim(), nat.GetOriginStateDim()) == (n, 35, 12, 6, 41)
    assert nat.getExtraInfoNames() == ref.impl.getExtraInfoNames()
    ob, rew, done, extra = np.zeros((n, 35), np.float32), np.zeros(n, np.float32), np.zeros(n, np.bool_), np.zeros((n, 6), np.float32)
    nat.observe(ob)
    assert np.array_equal(ob, ref.observe()) and ob.any()                       # filled in place
    rng = np.random.RandomState(11)
    n_done = 0
    for k in range(30):
        a = PL.random_actions(rng, n, 0.5)
        if k == 5:
            st = nat.get_state()
            st[::7, PL.S["GC"] + 2] = 0.14
            nat.set_state(st); ref.set_state(st)
        nat.step(a, ob, rew, done, extra)
        ob_r, rew_r, done_r, extra_r = ref.step(a)
        assert np.array_equal(ob, ob_r) and np.array_equal(rew, rew_r) and np.array_equal(done, done_r) and np.array_equal(extra, extra_r)
        n_done += int(done.sum())
    assert n_done >= 6
    out_n, out_r = np.zeros((n, 41), np.float32), np.zeros((n, 41), np.float32)
    nat.OriginState(out_n); ref.impl.OriginState(out_r)
    assert np.array_equal(out_n, out_r)
    r24, r24r = np.zeros((n, 24), np.float32), np.zeros((n, 24), np.float32)
    nat.ReferenceState(r24); ref.impl.ReferenceState(r24r)
    assert np.array_equal(r24, r24r)                                              # the VEC:223-226 dispatch quirk, both
    minv, minv_r = np.zeros((n, 324), np.float32), np.zeros((n, 324), np.float32)
    nat.GetInverseMassMatrix(minv); ref.impl.GetInverseMassMatrix(minv_r)
    assert np.array_equal(minv, minv_r)
    # the Eigen::Ref contract: no conversion, no copy
    with pytest.raises(TypeError):
        nat.observe(np.zeros((n, 35), np.float64))
    with pytest.raises(TypeError):
        nat.observe(np.zeros((35, n), np.float32).T)                              # not C-contiguous
    with pytest.raises(TypeError):
        nat.step(a, ob, rew, np.zeros(n, np.float32), extra)                      # done must be bool
    with pytest.raises(ValueError):
        nat.observe(np.zeros((n, 34), np.float32))
    with pytest.raises(RuntimeError, match="Flag_Crucial"):
        nat.GetSphereInfo(np.zeros((n, 4), np.float32))
    # the reference's Python adapter on top of the compiled class
    venv = RaisimGymVecEnv(mod.FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, text))
    o0 = venv.reset()
    o1, r1, d1, info = venv.step(PL.random_actions(rng, n, 0.3))
    assert o0.shape == (n, 35) and o1.dtype == np.float32 and r1.shape == (n,) and d1.dtype == np.bool_ and len(info) == n


def test_terrain_and_per_episode_randomisation_config5():
    """BASELINE config 5 ingredients: Perlin height field shared by the pool + friction/mass/COM/thigh randomisation
    redrawn at every reset."""
    cfg = load_env_cfg("bp5_terrain.yaml", num_envs=64)
    orc, cand = _pair(cfg)
    Ho, Hc = orc.heightfield(), cand.impl.heightfield()
    assert Ho.shape == (5000, 500) and np.abs(Ho - Hc).max() < 1e-7
    PL.check_init(orc, cand)
    worst, n_done = PL.check_teacher_forced(orc, cand, steps=120, force_terminal_every=4, max_factor=PL.TERRAIN_MAX_FACTOR)
    assert n_done >= 20
    print("terrain teacher-forced:", worst)
    # model parameters were redrawn by the in-kernel resets and still agree with the oracle's
    so, sc = orc.get_state(), cand.get_state()
    np.testing.assert_allclose(sc[:, PL.S["MATERIAL"]:PL.S["OB"]], so[:, PL.S["MATERIAL"]:PL.S["OB"]], atol=2e-7)
    assert np.unique(np.round(so[:, PL.S["MATERIAL"]], 5)).size > 32


@pytest.mark.parametrize("lanes", [4, 16])
def test_both_lane_layouts_match_oracle_on_gpu(lanes, monkeypatch):
    """The library carries the env kernels in two lane layouts (16 lanes per robot while the waves fit the SIMDs: <= 4096 robots on an MI355X, 4 beyond);
    IRRL_LANES_PER_ROBOT forces one so that both are checked on a small pool."""
    monkeypatch.setenv("IRRL_LANES_PER_ROBOT", str(lanes))
    cfg = load_env_cfg("default_cfg.yaml", num_envs=48)
    orc, cand = _pair(cfg)
    assert cand.impl.lanes_per_robot == lanes
    PL.check_init(orc, cand)
    PL.check_teacher_forced(orc, cand, steps=60, force_terminal_every=5)
    PL.check_probe(orc, cand)


def test_layout_is_chosen_by_pool_size(monkeypatch):
    monkeypatch.delenv("IRRL_LANES_PER_ROBOT", raising=False)
    monkeypatch.delenv("IRRL_L4_WAVES", raising=False)
    small = _hip(load_env_cfg("default_cfg.yaml", num_envs=4096)).impl
    assert small.lanes_per_robot == 16 and small.waves_per_simd == 1                # 1024 waves of four robots = one per SIMD of the MI355X
    mid = _hip(load_env_cfg("default_cfg.yaml", num_envs=4100)).impl
    assert mid.lanes_per_robot == 4 and mid.waves_per_simd == 1                     # a second round of 16-lane waves would cost a whole step more
    big = _hip(load_env_cfg("default_cfg.yaml", num_envs=16384))
    assert big.impl.lanes_per_robot == 4 and big.impl.waves_per_simd == 1      # 1024 waves of 16 robots = one per SIMD of the MI355X
    PL.check_invariants(big, steps=20)
    bigger = _hip(load_env_cfg("default_cfg.yaml", num_envs=16400))
    assert bigger.impl.lanes_per_robot == 4 and bigger.impl.waves_per_simd == 2  # more waves than SIMDs: the two-waves-per-SIMD build
    PL.check_invariants(bigger, steps=20)


@pytest.mark.parametrize("name,extra", [("default_cfg.yaml", {}), ("bp5_terrain.yaml", {}), ("bp5_imitation.yaml", {"Crutial": True, "CubeNum": 3, "period": 0.05}),
                                        ("bp5_imitation.yaml", {"ContactSolver": 0})])
def test_the_two_waves_per_simd_build_of_the_4_lane_kernels_equals_the_one_wave_build_bit_for_bit(name, extra, monkeypatch):
    """Pools of more than 16 384 robots launch the 4-lane kernels compiled for two resident waves per SIMD (`_l4w2`: 256 registers per wave,
    130-220 values of the step kernels in scratch).  Same source, same arithmetic (-ffp-contract=on): forced onto a small pool
    (IRRL_L4_WAVES), every step's outputs and the final pool equal the one-wave build's bit for bit -- one launch per step and the
    multi-step launch, noise, forced resets through falls, rough ground with per-episode randomisation, the meteorite, the other solver."""
    import torch
    monkeypatch.setenv("IRRL_LANES_PER_ROBOT", "4")
    n, steps = 80, 90
    rng = np.random.default_rng(11)
    acts = np.clip(0.6 * rng.standard_normal((steps, n, 12)), -1, 1).astype(np.float32)
    res = {}
    for waves in (1, 2):
        monkeypatch.setenv("IRRL_L4_WAVES", str(waves))
        env = _hip(load_env_cfg(name, num_envs=n, **extra))
        assert env.impl.lanes_per_robot == 4 and env.impl.waves_per_simd == waves
        env.reset()
        outs = []
        for k, a in enumerate(acts[:steps // 2]):                              # one launch per step (host boundary)
            if k % 6 == 5:                                                     # some robots below the termination height (ENV:1560): in-step resets
                st = env.get_state()
                st[k % n::9, PL.S["GC"] + 2] = 0.14
                env.set_state(st)
            outs.append(env.step(a))
        st = env.get_state()
        st[3::11, PL.S["GC"] + 2] = 0.14                                       # ... and inside the multi-step launch
        env.set_state(st)
        ob = torch.zeros(steps - steps // 2, n, 35, device="cuda"); rew = torch.zeros(steps - steps // 2, n, device="cuda")
        done = torch.zeros(steps - steps // 2, n, dtype=torch.bool, device="cuda"); ext = torch.zeros(steps - steps // 2, n, 6, device="cuda")
        table = torch.from_numpy(acts[steps // 2:]).cuda()
        env.impl.step_rows(steps - steps // 2, table, 0, ob, rew, done, ext, persistent=True)   # the multi-step kernel
        torch.cuda.synchronize()
        res[waves] = (outs, ob.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), ext.cpu().numpy(), env.get_state())
    a, b = res[1], res[2]
    assert sum(int(o[2].sum()) for o in a[0]) + int(a[3].sum()) > 0            # episodes did end on the way
    for (o1, o2) in zip(a[0], b[0]):
        for x, y in zip(o1, o2):
            assert np.array_equal(x, y)
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)


def test_reference_trajectory_mode_from_csv(tmp_path):
    """ManualTraj: False: the library reads the CSV named by cfg["RefTraj"] itself
    (VectorizedEnvironment.hpp:33-76, 158-176); the oracle gets the same table directly."""
    tab = PL.ref_table()
    path = tmp_path / "2020 ref_traj.csv"                      # the reference's file names contain spaces
    np.savetxt(str(path), tab, delimiter=",", fmt="%.9g")
    tab = np.loadtxt(str(path), delimiter=",", dtype=np.float32)  # what a reader of the file sees
    cfg = load_env_cfg("default_cfg.yaml", num_envs=48, ManualTraj=False, max_time=0.4, RefTraj=str(path))
    cand = _hip(cfg)
    ocfg = dict(cfg)
    ocfg["_ref_table"] = tab
    orc = O.OracleVecEnv(ocfg)
    PL.check_init(orc, cand)
    PL.check_teacher_forced(orc, cand, steps=40, seed=2)
    # without a readable file the pool is created but init() refuses; set_ref() then supplies the table
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
    cfg2 = dict(cfg, RefTraj="/nonexistent/ref.csv")
    env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg2))
    with pytest.raises(RuntimeError):
        env.init()
    env.set_ref(tab)
    env.init()


def test_manual_eval_mode_with_state_disturbance():
    """Evaluation configuration (Manual: True, bp5_test.yaml) + ForceDisturbance -> state_disturbance (ENV:912-940), through
    the C-ABI; period 0.02 s puts a kick every 100 steps inside the teacher-forced window."""
    cfg = load_env_cfg("default_cfg.yaml", num_envs=32, Manual=True, ForceDisturbance=True, period=0.02, ObsNoise=0.0, ActionNoise=0.0)
    orc, cand = _pair(cfg)
    PL.check_init(orc, cand)
    PL.check_teacher_forced(orc, cand, steps=130, seed=9, action_scale=0.1)
    # testStep (PYB:24 -> VEC:280-290): env 0 only
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
    env = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(dict(cfg, num_envs=4)))
    env.init()
    ob = np.zeros((4, 35), np.float32); rew = np.full(4, -7.0, np.float32); done = np.zeros(4, bool); extra = np.zeros((4, 6), np.float32)
    env.reset(ob)
    ob0 = ob.copy()
    env.testStep(np.zeros((4, 12), np.float32), ob, rew, done, extra)
    assert rew[0] != -7.0 and np.all(rew[1:] == -7.0)            # rows 1.. untouched
    assert np.array_equal(ob[1:], ob0[1:]) and not np.array_equal(ob[0], ob0[0])


# sim-to-sim table of the RaiSim-trained bp5_155 controller in THIS physics (profiles/r03_sim2sim_reference_policy_by_contact_rule.log,
# oracle): command -> mean v_x over the last 2 s.  With the PUBLISHED per-contact rule of RaiSim's solver (ContactSolver 3, the
# default since round 3) the controller tracks its commands up to 5 m/s: 0.511 / 1.023 / 2.047 / 2.936 / 4.086 / 4.737 / -0.697.
# With the build's first sliding rule (ContactSolver 2, rounds 1-2) the 4-5 m/s commands saturated at 3.6-3.8 m/s
# (0.511 / 1.022 / 2.035 / 2.819 / 3.758 / 3.564 / -0.700) -- which round 2 had attributed to the motor torque clamp.
SIM2SIM_BANDS = {3: ((0.5, 0.45, 0.58), (1.0, 0.93, 1.12), (2.0, 1.9, 2.2), (3.0, 2.7, 3.15), (4.0, 3.75, 4.4), (5.0, 4.3, 5.1), (-1.0, -0.85, -0.55)),
                 2: ((1.0, 0.9, 1.15), (3.0, 2.5, 3.2), (5.0, 3.2, 4.2))}


@pytest.mark.parametrize("solver", [3, 2])
def test_raisim_trained_policy_trots_in_the_hip_kernels(solver):
    """Sim-to-sim through the C-ABI: the reference's RaiSim-trained bp5_155 actor drives one Manual-mode env of the HIP engine
    (evaluation config rsc/bp5_manual_eval.yaml): no fall in 4 s at any command from -1 to 5 m/s, speed inside the regression
    band of the table above -- under the published contact rule the 5 m/s command is tracked to better than 10 %."""
    cfg = load_env_cfg("bp5_manual_eval.yaml", ContactSolver=solver)
    for cmd, lo, hi in SIM2SIM_BANDS[solver]:
        vx, falls = PL.closed_loop_reference_policy(_hip(cfg), cfg, cmd, 2000)
        assert falls == 0, cmd
        assert lo < vx[1000:].mean() < hi, (cmd, vx[1000:].mean())
        print("sim2sim solver %d cmd %+.1f -> %.3f m/s" % (solver, cmd, vx[1000:].mean()))


@pytest.mark.parametrize("lanes", [4, 16])
@pytest.mark.parametrize("solver", [0, 1, 2])
def test_every_contact_solver_matches_the_oracle_on_gpu(solver, lanes, monkeypatch):
    """ContactSolver 0 (Gauss-Seidel + the build's first rule), 1 (Gauss-Seidel + the published rule = the published method
    literally) and 2 (simultaneous sweeps + first rule) through the C-ABI on the MI355X, both lane layouts -- the shipped default 3
    is what every other test in this file runs.  Training config (noise, randomised friction, forced resets), robots tilted onto
    a trunk-box corner (the corners use the same per-contact rule), rough ground."""
    monkeypatch.setenv("IRRL_LANES_PER_ROBOT", str(lanes))
    orc, cand = _pair(load_env_cfg("default_cfg.yaml", num_envs=48, ContactSolver=solver))
    assert cand.impl.lanes_per_robot == lanes
    PL.check_init(orc, cand)
    worst, n_done = PL.check_teacher_forced(orc, cand, steps=60, force_terminal_every=5)
    print("solver", solver, "lanes", lanes, worst)
    orc, cand = _pair(load_env_cfg("bp5_imitation.yaml", num_envs=32, ContactSolver=solver))
    h0 = orc.box_hits()
    PL.check_teacher_forced(orc, cand, steps=30, seed=3, perturb=PL.tilt_onto_box_corner, max_factor=PL.CORNER_MAX_FACTOR, cap_factor=PL.CORNER_CAP_FACTOR)
    assert orc.box_hits() - h0 > 32 * 30 * 4
    orc, cand = _pair(load_env_cfg("bp5_terrain.yaml", num_envs=32, ContactSolver=solver))
    PL.check_teacher_forced(orc, cand, steps=40, force_terminal_every=9, max_factor=PL.TERRAIN_MAX_FACTOR)


@pytest.mark.parametrize("name", ["bp5_imitation.yaml", "bp5_terrain.yaml"])
def test_full_size_invariants_and_determinism_benchmark_and_terrain_configs(name):
    """4096 envs (the benchmarked pool size) of BASELINE config 2 (bp5_imitation.yaml) and of the config-5 ingredients
    (bp5_terrain.yaml: height field + per-episode randomisation): invariants over 40 steps and a bit-identical rerun."""
    cfg = load_env_cfg(name, num_envs=4096)
    flat = not cfg["Terrain"]
    st_a = PL.check_invariants(_hip(cfg), steps=40, flat_ground=flat)
    st_b = PL.check_invariants(_hip(cfg), steps=40, flat_ground=flat)
    assert np.array_equal(st_a, st_b)


def test_trained_policy_closed_loop_statistics_match_the_oracle():
    """Closed loop, training mode (commands up to 5 m/s, observation / action noise, randomised dynamics), 64 envs x 400 steps:
    the actor trained on this engine gives the same mean reward and speed, and no more terminations, in the HIP kernels
    (through the C-ABI) as in the f64 oracle.  Trajectories diverge (chaos), the statistics must not."""
    cfg = load_env_cfg("default_cfg.yaml", num_envs=64)
    so = PL.closed_loop_training_mode(O.OracleVecEnv(cfg), "actor_trained_on_hip_engine.npz", 400)
    sc = PL.closed_loop_training_mode(_hip(cfg), "actor_trained_on_hip_engine.npz", 400)
    print("closed-loop statistics oracle", so, "hip", sc)
    assert abs(sc["reward"] - so["reward"]) < 0.01 * abs(so["reward"])       # measured: 0.733741 vs 0.733722
    assert abs(sc["speed"] - so["speed"]) < 0.01 * so["speed"]               # measured: 1.282898 vs 1.282887 m/s
    assert so["terminations"] <= 2 and sc["terminations"] <= 2


def test_diagnostic_getters_match_the_oracle_values():
    """(f)3: OriginState [N,41] = gc | gv | contact flags (ENV:1317-1334), GetJointEffort / GetGeneralizedForce (ENV:1351-1371)
    after teacher-forced steps from common states -- values, not shapes."""
    n = 48
    orc, cand = _pair(load_env_cfg("default_cfg.yaml", num_envs=n))
    rng = np.random.RandomState(4)
    for k in range(60):                      # land first (free flight until ~step 46), then compare with feet on the ground
        orc.step(PL.random_actions(rng, n, 0.4))
    for k in range(6):
        st = PL.f32_round_state(orc.get_state())
        orc.set_state(st)
        cand.set_state(st)
        a = PL.random_actions(rng, n, 0.4)
        orc.step(a)
        cand.step(a)
        os_o = orc.origin_state()
        os_c = np.zeros((n, 41), np.float32)
        cand.impl.OriginState(os_c)
        same = np.all(os_o[:, 37:] == os_c[:, 37:], axis=1)                  # contact flags: exact except for threshold envs
        assert same.mean() > 0.95 and os_o[:, 37:].sum() > n                    # and the feet really are on the ground
        np.testing.assert_allclose(os_c[same, 0:19], os_o[same, 0:19], atol=10 * PL.TOL_STEP["pos"])
        np.testing.assert_allclose(os_c[same, 19:37], os_o[same, 19:37], atol=10 * PL.TOL_STEP["vel"])
        je_c, gf_c = np.zeros((n, 12), np.float32), np.zeros((n, 18), np.float32)
        cand.impl.GetJointEffort(je_c)
        cand.impl.GetGeneralizedForce(gf_c)
        je_o, gf_o = orc.joint_effort(), orc.generalized_force()
        assert np.abs(je_o).max() > 1.0                                         # real torques (N m), not zeros
        np.testing.assert_allclose(je_c[same], je_o[same], atol=2e-2)            # kp 40 x joint angle error 2e-5 + kd x 5e-3
        np.testing.assert_allclose(gf_c[same], gf_o[same], atol=2e-2)
        np.testing.assert_array_equal(gf_c[:, 6:], je_c)


def test_set_contact_coefficient_changes_the_cone_and_follows_the_oracle():
    """(f)3: SetContactCoefficient (ENV:1407-1418) installs (mu, restitution, threshold) per env: the same material in the
    oracle gives the same trajectories, the stored impulses obey the NEW cone, and slippery robots slide further."""
    n = 32
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=n)
    orc, cand = _pair(cfg)
    coeff = np.zeros((n, 3), np.float32)
    coeff[:, 0] = np.where(np.arange(n) % 2 == 0, 0.05, 0.9)   # ice under the even robots
    coeff[:, 1] = 0.1
    coeff[:, 2] = 0.5
    orc.set_contact_coeff(coeff)
    cand.impl.SetContactCoefficient(coeff)
    np.testing.assert_allclose(cand.get_state()[:, PL.S["MATERIAL"]:PL.S["MATERIAL"] + 3], coeff, atol=0)
    rng = np.random.RandomState(8)
    for _ in range(70):
        orc.step(PL.random_actions(rng, n, 0.5))
    st = PL.f32_round_state(orc.get_state())
    cand.set_state(st)
    orc.set_state(st)
    worst, _ = PL.check_teacher_forced(orc, cand, steps=30, seed=12)
    # the stored impulses obey the NEW cone, per env, over 40 more steps of the candidate alone
    touching, grippy_ratio = 0, 0.0
    for _ in range(40):
        cand.step(PL.random_actions(rng, n, 0.5))
        st = cand.get_state()
        lam = st[:, PL.S["LAMW"]:PL.S["LAMW"] + 12].reshape(n, 4, 3)
        ft = np.linalg.norm(lam[:, :, :2], axis=2)
        touching += int((lam[:, :, 2] > 0).sum())
        assert np.all(ft <= coeff[:, 0:1] * lam[:, :, 2] * (1 + 1e-4) + 1e-6)
        on = lam[1::2, :, 2] > 1e-4
        if on.any():
            grippy_ratio = max(grippy_ratio, float((ft[1::2][on] / lam[1::2, :, 2][on]).max()))
    assert touching > 4 * n                       # feet were on the ground
    assert grippy_ratio > 0.05 * 1.5              # the grippy robots do use more friction than the icy ones may


@pytest.mark.parametrize("lanes", [4, 16])
def test_trunk_box_corner_contacts_match_the_oracle_on_gpu(lanes, monkeypatch):
    """ENV:242 / URDF:26: robots tilted 52-58 degrees onto a bottom corner of the trunk's collision box at base heights of
    0.15-0.17 m (still inside the episode), teacher-forced through the C-ABI in both lane layouts; then the same on rough ground."""
    monkeypatch.setenv("IRRL_LANES_PER_ROBOT", str(lanes))
    n = 64
    orc, cand = _pair(load_env_cfg("bp5_imitation.yaml", num_envs=n))
    assert cand.impl.lanes_per_robot == lanes
    h0 = orc.box_hits()
    worst, n_done = PL.check_teacher_forced(orc, cand, steps=40, seed=3, perturb=PL.tilt_onto_box_corner, max_factor=PL.CORNER_MAX_FACTOR, cap_factor=PL.CORNER_CAP_FACTOR)
    assert orc.box_hits() - h0 > n * 40 * 4 and n_done < n * 40 // 4
    print("box-corner teacher-forced worst errors (lanes %d):" % lanes, worst)
    orc, cand = _pair(load_env_cfg("bp5_terrain.yaml", num_envs=32))
    h0 = orc.box_hits()
    PL.check_teacher_forced(orc, cand, steps=30, seed=5, perturb=lambda st, k, rng: PL.tilt_onto_box_corner(st, k, rng, 0.16, 0.30, 20.0, 55.0),
                            max_factor=PL.TERRAIN_MAX_FACTOR)
    assert orc.box_hits() > h0


@pytest.mark.parametrize("lanes", [4, 16])
def test_crutial_meteorite_matches_the_oracle_on_gpu(lanes, monkeypatch):
    """Crutial: True (ENV:273-284, 731-740, 815-861) through the C-ABI in both lane layouts: the park / release schedule,
    teacher-forced steps with a sphere dropped onto the trunk or the ground before every step, GetSphereInfo (ENV:1423-1436),
    and a free-running pool in which the scheduled spheres do reach robots."""
    monkeypatch.setenv("IRRL_LANES_PER_ROBOT", str(lanes))
    n = 48
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=n, Crutial=True, CubeNum=6, period=0.05)   # parked every 125 control steps
    orc, cand = _pair(cfg)
    assert cand.impl.lanes_per_robot == lanes
    k = PL.S["SPHERE"]
    assert np.abs(orc.get_state()[:, k:k + 9] - cand.get_state()[:, k:k + 9]).max() < 1e-6
    np.testing.assert_allclose(cand.sphere_info(), orc.sphere_info(), atol=1e-6)
    st = orc.get_state()
    st[:, PL.S["FRAME"]] = 110
    orc.set_state(st)
    PL.check_teacher_forced(orc, cand, steps=30, seed=2, force_terminal_every=11)       # crosses frame 125: park + release
    assert (orc.get_state()[:, k + 8] == 1).any()
    h0 = orc.sphere_hits()
    worst, n_done = PL.check_teacher_forced(orc, cand, steps=40, seed=4, perturb=PL.drop_meteorite, max_factor=PL.CORNER_MAX_FACTOR, cap_factor=PL.CORNER_CAP_FACTOR)
    assert orc.sphere_hits() - h0 > 5 * n
    print("meteorite teacher-forced worst errors (lanes %d):" % lanes, worst)
    np.testing.assert_allclose(cand.sphere_info(), orc.sphere_info(), atol=2e-4)
    orc, cand = _pair(load_env_cfg("bp5_terrain.yaml", num_envs=24, Crutial=True, CubeNum=2))
    PL.check_teacher_forced(orc, cand, steps=24, seed=6, perturb=PL.drop_meteorite, max_factor=PL.TERRAIN_MAX_FACTOR)
    # free running with the default gait period: released at frame 1, 1 m above the base at -5 m/s -> it arrives ~0.16 s later;
    # a standing robot (zero actions keep the nominal pose) is hit on the trunk and pushed down
    cfg = load_env_cfg("bp5_imitation.yaml", num_envs=n, Crutial=True, CubeNum=6, Manual=True, Vx=0.0, max_time=10.0)
    hip = _hip(cfg)
    hip.reset()
    a = np.zeros((n, 12), np.float32)
    zmin, vzmin, bounced = 1e9, 0.0, False
    for t in range(140):
        hip.step(a)
        s = hip.get_state()
        vzmin = min(vzmin, s[:, 21].min())
        bounced |= bool((s[:, k + 5] > 1.0).any())
    assert vzmin < -0.6 and bounced          # (a free-standing robot never drops faster than ~0.45 m/s while it settles)
    # without Crutial the getter refuses, like the reference's message
    plain = _hip(load_env_cfg("bp5_imitation.yaml", num_envs=4))
    with pytest.raises(RuntimeError, match="Flag_Crucial"):
        plain.sphere_info()


# ------------------------------------------------------------------------------------------------------------------------
# Round 4: the benchmarked configurations AT their benchmarked sizes against the oracle, and BASELINE configs 4 / 5 (32 768 envs
# = 8 shards of 4096) at their real size on one device.
# ------------------------------------------------------------------------------------------------------------------------
def _landed_oracle(cfg, preroll=70, seed=21):
    """oracle pool after `preroll` free-running steps: the robots have landed (46 steps of free flight after a reset), contact
    sets differ from robot to robot, the first falls / in-step resets have happened"""
    orc = O.OracleVecEnv(cfg)
    rng = np.random.RandomState(seed)
    for _ in range(preroll):
        orc.step(PL.random_actions(rng, orc.n, 0.4))
    return orc


@pytest.mark.parametrize("name,lanes,n", [("bp5_imitation.yaml", 16, 4096), ("default_cfg.yaml", 16, 4096), ("bp5_terrain.yaml", 16, 4096),
                                          ("bp5_imitation.yaml", 4, 16384), ("bp5_terrain.yaml", 4, 16384)])
def test_teacher_forced_at_the_benchmarked_pool_sizes(name, lanes, n, monkeypatch):
    """The kernel exactly as the bench and the PPO runs launch it -- 4096 envs = the full 1024-wave grid with the XCD-renumbered
    block mapping in the 16-lane layout, 16 384 envs in the 4-lane layout (the pool-size rule's own choice) -- against the f64
    oracle: 20 teacher-forced control steps from landed states with forced terminations, stated tolerances of parity_lib."""
    monkeypatch.delenv("IRRL_LANES_PER_ROBOT", raising=False)
    cfg = load_env_cfg(name, num_envs=n)
    cand = _hip(cfg)
    assert cand.impl.lanes_per_robot == lanes
    orc = _landed_oracle(cfg)
    rough = bool(cfg["Terrain"])
    worst, n_done = PL.check_teacher_forced(orc, cand, steps=20, seed=4, force_terminal_every=3, cap_factor=PL.FULL_SIZE_CAP_FACTOR,
                                            event_budget=PL.TERRAIN_EVENT_BUDGET if rough else 0.005)
    inc = orc.get_state()[:, PL.S["INCONTACT"]:PL.S["INCONTACT"] + 4]
    assert inc.sum() > n            # more than one foot on the ground per robot on average: the contact solve is what is compared
    assert n_done >= 6
    print("full-size teacher-forced %s lanes %d n %d:" % (name, lanes, n), worst)


def _step_all(pools, acts):
    outs = [p.step(np.ascontiguousarray(a)) for p, a in zip(pools, acts)]
    return [np.concatenate([o[i] for o in outs]) for i in range(4)]


@pytest.mark.parametrize("name", ["default_cfg.yaml", "bp5_terrain.yaml"])
def test_eight_shards_of_4096_are_the_32768_pool_bit_for_bit(name, monkeypatch):
    """BASELINE configs 4 / 5 at their real size on one device: eight pools of 4096 with EnvIdOffset k * 4096 (what rank k of the
    8-GPU job creates) against ONE pool of 32 768 in the same lane layout -- observations, rewards, dones, extraInfo and the
    whole state bit-identical over 30 steps with in-step resets (VEC:268-278: envs are independent; every random draw is addressed
    by the global env id).  The same big pool in the OTHER lane layout (what the pool-size rule would pick for 32 768 on one GPU)
    differs in the last bits -- different reduction trees -- and is held to the stated one-step tolerance instead."""
    monkeypatch.setenv("IRRL_LANES_PER_ROBOT", "16")
    n, k = 4096, 8
    shards = [_hip(load_env_cfg(name, num_envs=n, EnvIdOffset=r * n)) for r in range(k)]
    big = _hip(load_env_cfg(name, num_envs=n * k))
    assert big.impl.lanes_per_robot == 16 and all(s.impl.lanes_per_robot == 16 for s in shards)
    assert np.array_equal(big.observe(), np.concatenate([s.observe() for s in shards]))
    rng = np.random.RandomState(13)
    n_done = 0
    for step in range(30):
        a = PL.random_actions(rng, n * k, 0.5)
        if step == 8:       # robots below the termination height in every shard: in-step resets draw from the global-id streams
            st = big.get_state()
            st[::61, PL.S["GC"] + 2] = 0.14
            big.set_state(st)
            for r, s in enumerate(shards):
                s.set_state(st[r * n:(r + 1) * n])
        ob_b, r_b, d_b, x_b = big.step(a)
        ob_s, r_s, d_s, x_s = _step_all(shards, [a[r * n:(r + 1) * n] for r in range(k)])
        assert np.array_equal(ob_b, ob_s) and np.array_equal(r_b, r_s) and np.array_equal(d_b, d_s) and np.array_equal(x_b, x_s), step
        n_done += int(d_b.sum())
    assert n_done >= 400
    st_big = big.get_state()
    assert np.array_equal(st_big, np.concatenate([s.get_state() for s in shards]))
    # the other layout: teacher-forced against the 16-lane pool (f32 both: nothing to round), stated one-step tolerance
    monkeypatch.setenv("IRRL_LANES_PER_ROBOT", "4")
    other = _hip(load_env_cfg(name, num_envs=n * k))
    assert other.impl.lanes_per_robot == 4
    rough = name == "bp5_terrain.yaml"
    worst, _ = PL.check_teacher_forced(big, other, steps=10, seed=6, force_terminal_every=3, cap_factor=PL.FULL_SIZE_CAP_FACTOR,
                                       event_budget=PL.TERRAIN_EVENT_BUDGET if rough else 0.005)
    assert worst["ob_p99"] < 1e-4 and worst["pos_p99"] < 1e-5      # two f32 evaluation orders: far inside the f32-vs-f64 tolerance
    print("32768-pool, 4-lane against 16-lane layout:", worst)


def test_compiled_pybind_boundary_takes_device_buffers_zero_copy():
    """Round 4: the compiled boundary (native/_flexible_robot) with DEVICE buffers -- torch CUDA tensors through `__cuda_array_interface__`
    in step / reset / observe / isTerminalState (raisim_gym.cpp:19-21, 26; RaisimGymEnv.hpp:46-49 array contract: same shapes, in
    place, no conversion): bit-identical to the ctypes class's device path over 40 steps with resets, stream-ordered on torch's
    current stream (a side stream here), TypeError on a wrong dtype / a non-contiguous view / a host-device mix, ValueError on a
    wrong shape."""
    import torch
    import yaml
    import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
    from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
    from test_abi_surface import load_native_module
    mod = load_native_module()
    n = 256
    text = yaml.safe_dump(load_env_cfg("default_cfg.yaml", num_envs=n))
    nat = mod.FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, text)
    nat.init()
    ref = FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, text)
    ref.init()
    dev = torch.device("cuda")
    mk = lambda: (torch.zeros(n, 35, device=dev), torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.bool, device=dev), torch.zeros(n, 6, device=dev))
    ob_n, rew_n, done_n, ex_n = mk()
    ob_r, rew_r, done_r, ex_r = mk()
    nat.reset(ob_n); ref.reset(ob_r)
    assert torch.equal(ob_n, ob_r) and bool(ob_n.abs().sum() > 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    side = torch.cuda.Stream(device=dev)
    n_done = 0
    for k in range(40):
        a = torch.clamp(0.5 * torch.randn(n, 12, device=dev, generator=g), -1, 1)
        if k == 6:
            st = nat.get_state()
            st[::9, PL.S["GC"] + 2] = 0.14
            nat.set_state(st); ref.set_state(st)
        torch.cuda.synchronize()
        with torch.cuda.stream(side if k % 2 else torch.cuda.current_stream(dev)):
            nat.step(a, ob_n, rew_n, done_n, ex_n)
        ref.step(a, ob_r, rew_r, done_r, ex_r)
        torch.cuda.synchronize()
        assert torch.equal(ob_n, ob_r) and torch.equal(rew_n, rew_r) and torch.equal(done_n, done_r) and torch.equal(ex_n, ex_r), k
        n_done += int(done_n.sum())
    assert n_done >= 20
    nat.observe(ob_n); ref.observe(ob_r)
    assert torch.equal(ob_n, ob_r)
    t_n, t_r = torch.zeros(n, dtype=torch.bool, device=dev), torch.zeros(n, dtype=torch.bool, device=dev)
    nat.isTerminalState(t_n); ref.isTerminalState(t_r)
    assert torch.equal(t_n, t_r)
    a = torch.zeros(n, 12, device=dev)
    with pytest.raises(TypeError):
        nat.step(a.double(), ob_n, rew_n, done_n, ex_n)
    with pytest.raises(TypeError):
        nat.step(a, ob_n, rew_n, done_n.float(), ex_n)                      # done must be bool / uint8
    with pytest.raises(TypeError):
        nat.observe(torch.zeros(35, n, device=dev).t())                     # not C-contiguous
    with pytest.raises(TypeError):
        nat.step(a, np.zeros((n, 35), np.float32), rew_n, done_n, ex_n)     # host / device mix
    with pytest.raises(ValueError):
        nat.observe(torch.zeros(n, 34, device=dev))
    # numpy still goes through the host entry points
    ob_h = np.zeros((n, 35), np.float32)
    nat.observe(ob_h)
    assert np.array_equal(ob_h, ob_n.cpu().numpy())
