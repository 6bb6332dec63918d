#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's importable Python twins.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes are plain
input/output vectors (a few KB) and are committed, so the tests never read /root/reference.

Sources (all under /root/reference/IRRL/script):
  utils/GaitGenerator.py   ik (GG:268), kinematic (GG:319), cubicBezier (GG:313), gauss (GG:309)
  bp5_config.py            obs_mean / obs_std / action_mean / action_std (bp5_config.py:19-55)
  utils/CustomerLstmNN.py  numpy LSTM actor forward (NN:112-135) with model/bp5_155/*.csv weights
  pkl/bp5_155.pkl          stable-baselines checkpoint: (data dict, 19 parameter arrays)
"""
import io
import json
import os
import pickle
import sys
import types
import contextlib

import numpy as np

REF = "/root/reference/IRRL/script"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def gen_task_math():
    sys.path.insert(0, os.path.join(REF, "utils"))
    sys.path.insert(0, REF)
    from GaitGenerator import GaitGenerator as GG  # noqa
    with contextlib.redirect_stdout(io.StringIO()):
        import bp5_config  # prints one line

    rng = np.random.RandomState(20211002)
    out = {}

    # cubicBezier / gauss tables
    bez = []
    for _ in range(24):
        p0 = rng.uniform(-0.3, 0.3, 3)
        pf = rng.uniform(-0.3, 0.3, 3)
        s = float(rng.uniform(0, 1))
        bez.append(dict(p0=p0.tolist(), pf=pf.tolist(), s=s, out=GG.cubicBezier(p0, pf, s).tolist()))
    out["cubicBezier"] = bez
    out["gauss"] = [dict(x=float(x), w=float(w), h=float(h), out=float(GG.gauss(x, w, h)))
                    for x, w, h in zip(rng.uniform(0, 1, 24), rng.uniform(0.5, 2, 24), rng.uniform(0.01, 0.2, 24))]

    # IK: the Python twin shares the abad (theta0) and knee (theta2) formulas with the C++ IK
    # (ENV:1702-1729) when the target is inside the workspace; the hip formula differs in one term
    # (GG:299 vs ENV:1738) except at x == 0.
    ik = []
    for i in range(48):
        is_right = bool(i % 2)
        x = 0.0 if i < 16 else float(rng.uniform(-0.15, 0.15))
        y = float((-1 if is_right else 1) * rng.uniform(0.05, 0.14))
        z = float(-rng.uniform(0.2, 0.36))
        th = GG.ik(x, y, z, 0.085, 0.209, 0.2175, is_right)
        ik.append(dict(x=x, y=y, z=z, is_right=is_right, theta=th.tolist()))
    out["ik"] = ik

    # forward kinematics twin: toe position for joint angles (IK convention: theta, not -theta).
    # GG:326 has cos(theta_hip - theta_knee) in its y row where x and z use (hip + knee), so the
    # twin is an exact inverse of the IK only in the sagittal plane: first 16 rows use abad = 0.
    fk = []
    for i in range(32):
        is_right = bool(i % 2)
        th = [0.0 if i < 16 else float(rng.uniform(-0.3, 0.3)), float(rng.uniform(0.2, 1.2)),
              float(-rng.uniform(0.6, 2.2))]
        x, y, z = GG.kinematic(th[0], th[1], th[2], is_right=is_right)
        fk.append(dict(theta=th, is_right=is_right, xyz=[float(x), float(y), float(z)]))
    out["kinematic"] = fk

    out["bp5_config"] = dict(obs_mean=bp5_config.obs_mean.tolist(), obs_std=bp5_config.obs_std.tolist(),
                             action_mean=bp5_config.action_mean.tolist(), action_std=bp5_config.action_std.tolist())
    with open(os.path.join(OUT, "task_math.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote task_math.json")


class _Stub(object):
    """Stands in for any tensorflow / stable_baselines / gym / cloudpickle object inside the pickle."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Stub()

    def __setstate__(self, s):
        self.state = s

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Stub()


def _stub_fn(*a, **k):
    return _Stub()


class _StubUnpickler(pickle.Unpickler):
    """Decode the SB pickle without tensorflow / stable_baselines / gym installed."""
    _ROOTS = ("cloudpickle", "gym", "tensorflow", "stable_baselines", "raisim_gym", "flex_gym", "types")
    _FNS = ("CodeType", "code", "_make_skel_func", "_fill_function", "_builtin_type", "_make_cell",
            "_make_empty_cell", "_rehydrate_skeleton_class", "_make_skeleton_class", "subimport")

    def find_class(self, module, name):
        if module.split(".")[0] in self._ROOTS:
            return _stub_fn if name in self._FNS else _Stub
        return super().find_class(module, name)


def load_bp5_pickle():
    with open(os.path.join(REF, "pkl", "bp5_155.pkl"), "rb") as f:
        data, params = _StubUnpickler(f).load()
    return data, params


def gen_lstm():
    """Known answers for the LSTM actor/critic from the trained bp5_155 weights."""
    # (1) the reference's own numpy actor (CSV branch, NN:73-88), needs cwd = IRRL/script
    sys.modules.setdefault("raisim_gym", types.ModuleType("raisim_gym"))
    sys.modules.setdefault("raisim_gym.algo", types.ModuleType("raisim_gym.algo"))
    m = types.ModuleType("raisim_gym.algo.ppo2")
    m.PPO2 = object
    sys.modules.setdefault("raisim_gym.algo.ppo2", m)
    sys.path.insert(0, os.path.join(REF, "utils"))
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            from CustomerLstmNN import CustomerLstmNN
            nn = CustomerLstmNN("./pkl/bp5_155.pkl", n_lstm=[48, 48])
        rng = np.random.RandomState(155)
        obs_seq = rng.uniform(-1.5, 1.5, (6, 35))
        obs_seq[0] = 0.0
        acts = []
        for t in range(obs_seq.shape[0]):
            nn.predict(obs_seq[t])
            acts.append(np.array(nn.output, np.float64).tolist())
        csv_w = dict(wx0=np.array(nn.lstm_wx[0]), pi_b=np.array(nn.pi_b))
    finally:
        os.chdir(cwd)

    # (2) checkpoint contents: shapes, hyper-parameters, per-tensor checksums (not the weights)
    data, params = load_bp5_pickle()
    if isinstance(params, dict):
        plist = list(params.values())
        pnames = list(params.keys())
    else:
        plist = list(params)
        pnames = [str(i) for i in range(len(plist))]
    keep = {}
    for k in ("n_envs", "n_steps", "gamma", "lam", "noptepochs", "nminibatches", "ent_coef", "vf_coef",
              "max_grad_norm", "learning_rate", "cliprange", "policy_kwargs"):
        if k in data:
            v = data[k]
            keep[k] = v if isinstance(v, (int, float, dict, list, str)) else str(v)
    meta = dict(hparams=keep, names=pnames, shapes=[list(np.shape(p)) for p in plist],
                n_params=int(sum(np.size(p) for p in plist)),
                abs_sums=[float(np.abs(np.asarray(p, np.float64)).sum()) for p in plist],
                csv_vs_pkl_wx0_maxdiff=float(np.abs(csv_w["wx0"] - np.asarray(plist[0], np.float64)).max()))

    # (3) critic value + neglogp of a fixed action from the pickle weights, by a literal numpy
    #     transcription of the stable-baselines lstm recurrence (gate order i,f,o,g; NN:119-129)
    P = [np.asarray(p, np.float64) for p in plist]

    def lstm_stack(x, layers, state):
        h_in = x
        new = []
        for (wx, wh, b), (c, h) in zip(layers, state):
            z = h_in @ wx + h @ wh + b
            n = wh.shape[0]
            i, f, o, g = z[:n], z[n:2 * n], z[2 * n:3 * n], z[3 * n:]
            sig = lambda v: 1.0 / (1.0 + np.exp(-v))
            c = sig(f) * c + sig(i) * np.tanh(g)
            h = sig(o) * np.tanh(c)
            new.append((c, h))
            h_in = h
        return h_in, new

    pi_layers = [(P[0], P[1], P[2]), (P[3], P[4], P[5])]
    v_layers = [(P[6], P[7], P[8]), (P[9], P[10], P[11])]
    vf_w, vf_b, pi_w, pi_b, logstd = P[12], P[13], P[14], P[15], P[16]
    st_pi = [(np.zeros(48), np.zeros(48)) for _ in range(2)]
    st_v = [(np.zeros(48), np.zeros(48)) for _ in range(2)]
    means, values, nlps = [], [], []
    act = np.linspace(-0.5, 0.5, 12)
    for t in range(obs_seq.shape[0]):
        hp, st_pi = lstm_stack(obs_seq[t], pi_layers, st_pi)
        hv, st_v = lstm_stack(obs_seq[t], v_layers, st_v)
        mean = hp @ pi_w + pi_b
        val = float((hv @ vf_w + vf_b).reshape(-1)[0])
        std = np.exp(logstd.reshape(-1))
        nlp = float(0.5 * np.sum(((act - mean) / std) ** 2) + 0.5 * np.log(2 * np.pi) * 12 + np.sum(logstd))
        means.append(mean.tolist())
        values.append(val)
        nlps.append(nlp)
    out = dict(obs_seq=obs_seq.tolist(), actor_clipped_csv=acts, actor_mean_pkl=means, value_pkl=values,
               neglogp_pkl=nlps, action_for_neglogp=act.tolist(), checkpoint=meta)
    with open(os.path.join(OUT, "lstm_bp5_155.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote lstm_bp5_155.json; n_params", meta["n_params"], "csv-vs-pkl", meta["csv_vs_pkl_wx0_maxdiff"])


def gen_lstm_synthetic():
    """The reference's numpy LSTM actor (CustomerLstmNN.predict, NN:112-135) driven with SEEDED RANDOM weights,
    so the fixture pins the recurrence (gate order, state handling, clipping) without shipping trained weights:
    the test regenerates the same weights from the seed."""
    sys.modules.setdefault("raisim_gym", types.ModuleType("raisim_gym"))
    sys.modules.setdefault("raisim_gym.algo", types.ModuleType("raisim_gym.algo"))
    m = types.ModuleType("raisim_gym.algo.ppo2")
    m.PPO2 = object
    sys.modules.setdefault("raisim_gym.algo.ppo2", m)
    sys.path.insert(0, os.path.join(REF, "utils"))
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            from CustomerLstmNN import CustomerLstmNN
            nn = CustomerLstmNN("./pkl/bp5_155.pkl", n_lstm=[48, 48])
    finally:
        os.chdir(cwd)
    seed = 424242
    rng = np.random.RandomState(seed)
    shapes = [(35, 192), (48, 192), (192,), (48, 192), (48, 192), (192,), (48, 12), (12,)]
    w = [rng.uniform(-0.4, 0.4, sh) for sh in shapes]          # wx0 wh0 b0 wx1 wh1 b1 pi_w pi_b
    nn.lstm_wx, nn.lstm_wh, nn.lstm_b = [w[0], w[3]], [w[1], w[4]], [w[2], w[5]]
    nn.pi_w, nn.pi_b = w[6], w[7]
    nn.reset()
    obs_seq = rng.uniform(-2, 2, (8, 35))
    outs, hid = [], []
    for t in range(8):
        nn.predict(obs_seq[t])
        outs.append(np.array(nn.output).tolist())
        hid.append([np.array(nn.cell_state[1]).tolist(), np.array(nn.hidden[1]).tolist()])
    with open(os.path.join(OUT, "lstm_synthetic.json"), "w") as f:
        json.dump(dict(seed=seed, shapes=[list(sh) for sh in shapes], weight_range=[-0.4, 0.4], obs_range=[-2, 2],
                       steps=8, actor_out=outs, layer1_c_h=hid), f, indent=1)
    print("wrote lstm_synthetic.json")


def gen_ppo_math():
    """GAE and PPO2 loss known answers from a literal numpy transcription of ppo2.py:554-568 and 152-175
    (tensorflow is absent, so the TF graph itself cannot be run)."""
    rng = np.random.RandomState(7)
    T, N = 12, 5
    rewards = rng.uniform(-1, 1, (T, N)).astype(np.float32)
    values = rng.uniform(-2, 2, (T, N)).astype(np.float32)
    dones = rng.uniform(size=(T, N)) < 0.2
    last_values = rng.uniform(-2, 2, N).astype(np.float32)
    last_dones = rng.uniform(size=N) < 0.3
    gamma, lam = 0.99, 0.998
    # --- ppo2.py:554-568 ---
    mb_advs = np.zeros_like(rewards)
    last_gae_lam = 0
    for step in reversed(range(T)):
        if step == T - 1:
            nextnonterminal = 1.0 - last_dones
            nextvalues = last_values
        else:
            nextnonterminal = 1.0 - dones[step + 1]
            nextvalues = values[step + 1]
        delta = rewards[step] + gamma * nextvalues * nextnonterminal - values[step]
        mb_advs[step] = last_gae_lam = delta + gamma * lam * nextnonterminal * last_gae_lam
    mb_returns = mb_advs + values
    # --- ppo2.py:152-175 + 262-263 ---
    B = 64
    neglogpac = rng.uniform(5, 15, B)
    old_neglogpac = neglogpac + rng.normal(0, 0.3, B)
    vpred = rng.uniform(-2, 2, B)
    old_vpred = vpred + rng.normal(0, 0.4, B)
    returns = rng.uniform(-2, 2, B)
    entropy = np.full(B, 17.03)
    clip, ent_coef, vf_coef = 0.2, 0.01, 0.5
    advs = returns - old_vpred
    advs = (advs - advs.mean()) / (advs.std() + 1e-8)
    vpredclipped = old_vpred + np.clip(vpred - old_vpred, -clip, clip)
    vf_loss = .5 * np.mean(np.maximum(np.square(vpred - returns), np.square(vpredclipped - returns)))
    ratio = np.exp(old_neglogpac - neglogpac)
    pg_loss = np.mean(np.maximum(-advs * ratio, -advs * np.clip(ratio, 1.0 - clip, 1.0 + clip)))
    approxkl = .5 * np.mean(np.square(neglogpac - old_neglogpac))
    clipfrac = np.mean((np.abs(ratio - 1.0) > clip).astype(np.float32))
    loss = pg_loss - np.mean(entropy) * ent_coef + vf_loss * vf_coef
    out = dict(gae=dict(rewards=rewards.tolist(), values=values.tolist(), dones=dones.tolist(), last_values=last_values.tolist(),
                        last_dones=last_dones.tolist(), gamma=gamma, lam=lam, advs=mb_advs.tolist(), returns=mb_returns.tolist()),
               loss=dict(neglogpac=neglogpac.tolist(), old_neglogpac=old_neglogpac.tolist(), vpred=vpred.tolist(),
                         old_vpred=old_vpred.tolist(), returns=returns.tolist(), entropy=entropy.tolist(), cliprange=clip,
                         ent_coef=ent_coef, vf_coef=vf_coef, normalized_advs=advs.tolist(), pg_loss=float(pg_loss),
                         vf_loss=float(vf_loss), approxkl=float(approxkl), clipfrac=float(clipfrac), loss=float(loss)))
    with open(os.path.join(OUT, "ppo_math.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote ppo_math.json")


def gen_actor_weights():
    """The eight actor tensors of the trained bp5_155 checkpoint (a data file of the reference: 35 340 f32 numbers) as a
    fixture for the closed-loop sim-to-sim tests: the RaiSim-trained controller must trot in this build's physics."""
    _, params = load_bp5_pickle()
    params = [np.asarray(p, np.float32) for p in params]
    # SB order (run_bp_v5.py:143-176): pi-lstm0 (wx, wh, b), pi-lstm1, vf-lstm0, vf-lstm1, vf (w, b), pi (w, b), logstd, q (w, b)
    names = ["wx0", "wh0", "b0", "wx1", "wh1", "b1"]
    out = {n: params[i] for i, n in enumerate(names)}
    out["pi_w"], out["pi_b"] = params[14], params[15]
    assert out["wx0"].shape == (35, 192) and out["wh1"].shape == (48, 192) and out["pi_w"].shape == (48, 12)
    np.savez_compressed(os.path.join(OUT, "actor_bp5_155.npz"), **out)
    print("wrote actor_bp5_155.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["task", "lstm", "lstm_syn", "ppo", "actor"]
    if "actor" in which:
        gen_actor_weights()
    if "task" in which:
        gen_task_math()
    if "lstm" in which:
        gen_lstm()
    if "lstm_syn" in which:
        gen_lstm_synthetic()
    if "ppo" in which:
        gen_ppo_math()
