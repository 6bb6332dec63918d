#!/usr/bin/env python3
"""run_bp_v5.py -- train branch of the reference's experiment script (IRRL/script/run_bp_v5.py:196-259) on the
MI355X engine.  Same construction order and hyper-parameters (run_bp_v5.py:227-242); the import block is the
reference's own (run_bp_v5.py:8-13), resolved by the compat packages at the repo root.

    python scripts/run_bp_v5.py --train --max_iter 6144000                 # imitation stage
    python scripts/run_bp_v5.py --train --load data/..._Iteration_100.pkl  # relaxation stage (new reward coeffs in --cfg)
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_bp_v5.py --train   # 8 x 4096 envs, RCCL grads

    python scripts/run_bp_v5.py --test --model data/..._final.pkl --cmd 1.5 --steps 2000   # headless evaluation

`--test` keeps the evaluation LOOP of the reference (run_bp_v5.py:353-470: Manual-mode env, externally supplied command in
obs[0:3], observation delay line, velocity / action low-pass filters, numpy LSTM actor, per-step records of joint state,
posture, effort and the true simulator state) with a fixed command (`--cmd`, the reference's `flag_fix_cmd`) instead of
the gamepad, and writes the records to an .npz instead of the ~700 lines of matplotlib analysis (run_bp_v5.py:471-1121),
which stay out of scope.  `--o` (CSV export of the actor) is kept.
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import yaml  # noqa: E402

from flex_gym.env.RaisimGymVecEnv import TorchVecEnv as Environment  # noqa: E402  (device-resident VecEnv)
from flex_gym.env.env.BlackPanther_V55 import __BLACKPANTHER_V55_RESOURCE_DIRECTORY__ as __RSCDIR__  # noqa: E402
from flex_gym.algo.ppo2 import PPO2  # noqa: E402
from flex_gym.archi.policies import CustomLSTMPolicy, MlpPolicy  # noqa: E402
from flex_gym.helper.raisim_gym_helper import ConfigurationSaver, TensorboardLauncher  # noqa: E402
from _flexible_robot import FlexibleGymEnv  # noqa: E402

N_LSTM = [48, 48]  # run_bp_v5.py:111


def parse_args(argv):
    p = argparse.ArgumentParser(description="Train control policies (MI355X engine).")
    p.add_argument("--train", dest="train", action="store_true", default=True)
    p.add_argument("--test", dest="train", action="store_false")
    p.add_argument("--cfg", type=str, default=os.path.abspath(__RSCDIR__ + "/default_cfg.yaml"), help="configuration file")
    p.add_argument("--max_iter", dest="max_iter", type=int, default=200000000, help="total timesteps (all GPUs)")
    p.add_argument("--save", dest="save_flag", type=lambda s: str(s).lower() not in ("0", "false", "no"), default=True)
    p.add_argument("--l", dest="learn_rate", type=float, default=1e-3)
    p.add_argument("--load", dest="pre_trained_model", type=str, default=None, help="warm start (IRRL relaxation stage)")
    p.add_argument("--model", dest="trained_model", type=str, default=None)
    p.add_argument("--o", dest="flag_output", action="store_true", default=False, help="export the actor to CSV")
    p.add_argument("--policy", choices=["lstm", "mlp"], default="lstm", help="lstm = CustomLSTMPolicy (bp5), mlp = MlpPolicy")
    p.add_argument("--num_envs", type=int, default=None, help="override environment.num_envs (per GPU)")
    p.add_argument("--eval_every_n", type=int, default=100)
    p.add_argument("--seed", type=int, default=None, help="override the yaml's seeds: `seed` (policy init, sampling noise) and environment.seedd (env RNG)")
    # --test (run_bp_v5.py:61-108 flag_fix_cmd, delay, vel_filter_freq, act_filter_freq)
    p.add_argument("--cmd", dest="flag_fix_cmd", type=float, default=1.0, help="fixed forward-velocity command of --test [m/s]")
    p.add_argument("--steps", type=int, default=1500, help="control steps of --test")
    p.add_argument("--delay", type=int, default=0, help="observation delay in control steps (DelayTool)")
    p.add_argument("--vel_filter_freq", type=float, default=1.0e6, help="low-pass on the joint-rate / omega observations [Hz]")
    p.add_argument("--act_filter_freq", type=float, default=1.0e6, help="low-pass on the action [Hz]")
    p.add_argument("--cmd_filter_freq", type=float, default=1.0, help="low-pass on the command [Hz] (GaitGenerator command_filter_freq)")
    p.add_argument("--out", type=str, default=None, help="--test: .npz with the per-step records")
    return p.parse_args(argv)


def run_test(args, cfg):
    """Headless evaluation loop (run_bp_v5.py:300-470 without gamepad, window and plots).  One Manual-mode env on the
    numpy boundary (testStep-style single env), the trained actor as a numpy LSTM (CustomerLstmNN twin)."""
    import numpy as np
    from flex_gym.env.RaisimGymVecEnv import RaisimGymVecEnv
    from high_speed_quadrupedal_locomotion_by_irrl_amd.checkpoint import NumpyLstmActor, read_checkpoint
    from high_speed_quadrupedal_locomotion_by_irrl_amd.helper import DelayTool, obs_normalisation
    if args.trained_model is None:
        raise SystemExit("model path can't be ignored during test mode (--model)")
    ecfg = dict(cfg["environment"])
    ecfg["num_envs"] = 1
    if not ecfg.get("Manual"):
        print("*" * 50 + "\nMake sure the Manual flag is opened!!! (forcing Manual: True for this run)\n" + "*" * 50)
        ecfg["Manual"] = True
    env = RaisimGymVecEnv(FlexibleGymEnv(__RSCDIR__, yaml.safe_dump(ecfg)))
    _, params = read_checkpoint(args.trained_model)
    ctrl = NumpyLstmActor.from_parameter_list(params, n_layers=len(N_LSTM))
    obs_mean, obs_std, action_mean, action_std = obs_normalisation(ecfg)
    dt = float(ecfg["control_dt"])
    alpha = lambda f: 2 * math.pi * dt * f / (2 * math.pi * dt * f + 1.0)
    a_vel, a_act = alpha(args.vel_filter_freq), alpha(args.act_filter_freq)
    env.SetContactCoefficient(np.array([[0.8, 0.2, 0.01]], dtype=np.float32))      # run_bp_v5.py:317-318
    obs = env.reset()
    d_tool = DelayTool(dt, dt * args.delay)
    vel_his, act_his = np.zeros(35), np.zeros(12)
    action_total = np.zeros([env.num_envs, env.num_acts], dtype=np.float32)
    rec = {k: [] for k in ("joint", "joint_dot", "posture", "omega", "phase", "act", "oss", "contact", "joint_effort", "cmd", "reward")}
    cmd_target = np.array([args.flag_fix_cmd, 0.0, 0.0])
    a_cmd = alpha(args.cmd_filter_freq)          # the reference's GaitGenerator low-passes the gamepad / fixed command at 1 Hz
    cmd = np.zeros(3)
    n_done = 0
    for t in range(args.steps):
        cmd = (1 - a_cmd) * cmd + a_cmd * cmd_target
        o = np.array(d_tool.input_output(obs[0, :].copy()), dtype=np.float64)
        o[32:35] = (1 - a_vel) * vel_his[32:35] + a_vel * o[32:35]
        o[17:29] = (1 - a_vel) * vel_his[17:29] + a_vel * o[17:29]
        vel_his = o.copy()
        o[0:3] = (cmd - obs_mean[0:3]) / obs_std[0:3]                                # Manual: the script owns obs[0:3]
        action = ctrl.predict(o)
        action = (1 - a_act) * act_his + a_act * action
        act_his = action
        action_total[0, :] = action
        ob_double = o * obs_std + obs_mean
        state = env.OriginState()[0, :]
        rec["joint"].append(ob_double[5:17]); rec["joint_dot"].append(ob_double[17:29]); rec["posture"].append(ob_double[29:32])
        rec["omega"].append(ob_double[32:35]); rec["phase"].append(ob_double[3:5]); rec["act"].append(action * action_std + action_mean)
        rec["oss"].append(state[0:37]); rec["contact"].append(state[37:41]); rec["joint_effort"].append(env.GetJointEffort()[0, :])
        rec["cmd"].append(cmd.copy())
        obs, reward, done, _ = env.step(action_total, visualize=False)
        rec["reward"].append(float(reward[0]))
        if done[0]:
            n_done += 1
            ctrl.reset()
            cmd = np.zeros(3)                   # the env restarted from rest
    out = {k: np.asarray(v) for k, v in rec.items()}
    vx = out["oss"][:, 19]
    print("test: %d steps (%.2f s), command %.2f m/s, mean forward velocity %.3f m/s (last half %.3f), falls %d, mean reward %.4f"
          % (args.steps, args.steps * dt, args.flag_fix_cmd, float(vx.mean()), float(vx[len(vx) // 2:].mean()), n_done,
             float(np.mean(out["reward"]))))
    if args.out:
        np.savez_compressed(args.out, **out)
        print("records written to", args.out)
    return out


def main(argv=None):
    import torch
    args = parse_args(argv)
    cfg = yaml.safe_load(open(args.cfg, "r"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under a launcher (complete env:// rendezvous: WORLD_SIZE, RANK, MASTER_PORT), also with one rank: the collectives run over RCCL.
    # A shell that merely exports WORLD_SIZE=1 stays single-process.
    launched = all(k in os.environ for k in ("WORLD_SIZE", "RANK", "MASTER_PORT"))
    if world > 1 and not launched:
        raise SystemExit("WORLD_SIZE=%d but RANK / MASTER_PORT are not set: start the ranks with torch.distributed.run" % world)
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("IRRL_BACKEND", "nccl")          # "gloo": CPU-side collectives (tests, one-device multi-rank runs)
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
        else:
            torch.distributed.init_process_group(backend)
    if not args.train and not args.flag_output:
        return run_test(args, cfg)
    if args.num_envs:
        cfg["environment"]["num_envs"] = args.num_envs
    if args.seed is not None:
        cfg["seed"] = int(args.seed)
        cfg["environment"]["seedd"] = int(args.seed)
    # rank r owns the global env ids r * num_envs .. of the one big pool: same seed everywhere, results independent of the GPU count
    cfg["environment"]["EnvIdOffset"] = rank * int(cfg["environment"]["num_envs"])
    # run_bp_v5.py:205-207: the environment sub-tree is dumped to a string and parsed again on the native side
    env = Environment(FlexibleGymEnv(__RSCDIR__, yaml.safe_dump(cfg["environment"]), device=local_rank))

    if args.flag_output:
        model = PPO2.load(args.trained_model, env=env)
        from high_speed_quadrupedal_locomotion_by_irrl_amd.checkpoint import export_actor_csv
        name = os.path.splitext(os.path.basename(args.trained_model))[0]
        print("exported to", export_actor_csv(model, os.path.join(os.getcwd(), "model", name)))
        return


    saver = None
    if args.save_flag and rank == 0:
        saver = ConfigurationSaver(log_dir=os.path.join(ROOT, "data", "black_panther_v5_test"), save_items=[args.cfg])
    log_path = saver.data_dir if saver else ""
    policy = CustomLSTMPolicy if args.policy == "lstm" else MlpPolicy
    kwargs = dict(n_lstm=N_LSTM) if args.policy == "lstm" else {}
    n_steps = math.floor(cfg["environment"]["max_time"] / cfg["environment"]["control_dt"])      # 750
    if args.pre_trained_model is None:
        model = PPO2(tensorboard_log=log_path, policy=policy, policy_kwargs=kwargs, env=env, gamma=0.99, n_steps=n_steps,
                     ent_coef=0.000, learning_rate=args.learn_rate, vf_coef=0.5, max_grad_norm=0.5, lam=0.998,
                     nminibatches=1 if args.policy == "lstm" else 4, noptepochs=10, cliprange=0.2, verbose=1,
                     seed=int(cfg.get("seed", 1)))
    else:
        model = PPO2.load(args.pre_trained_model, env=env, verbose=1)      # run_bp_v5.py:244-248
        model.tensorboard_log = log_path
        model.learning_rate = args.learn_rate
    if saver:
        TensorboardLauncher(saver.data_dir + "/PPO2_1")
    model.learn(total_timesteps=args.max_iter, eval_every_n=args.eval_every_n, log_dir=log_path,
                record_video=bool(cfg.get("record_video", False)))
    if rank == 0 and log_path:
        print("final checkpoint:", model.save(log_path + "_final"))
    if launched:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
