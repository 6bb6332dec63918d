#!/usr/bin/env python3
"""run_bp_v5.py -- train branch of the reference's experiment script (IRRL/script/run_bp_v5.py:196-259) on the
MI355X engine.  Same construction order and hyper-parameters (run_bp_v5.py:227-242); the import block is the
reference's own (run_bp_v5.py:8-13), resolved by the compat packages at the repo root.

    python scripts/run_bp_v5.py --train --max_iter 6144000                 # imitation stage
    python scripts/run_bp_v5.py --train --load data/..._Iteration_100.pkl  # relaxation stage (new reward coeffs in --cfg)
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_bp_v5.py --train   # 8 x 4096 envs, RCCL grads

The `--test` branch of the reference (gamepad-driven evaluation + ~700 lines of matplotlib analysis,
run_bp_v5.py:261-1121) is out of scope; `--o` (CSV export of the actor) is kept.
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
    return p.parse_args(argv)


def main(argv=None):
    import torch
    args = parse_args(argv)
    cfg = yaml.safe_load(open(args.cfg, "r"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
    if args.num_envs:
        cfg["environment"]["num_envs"] = args.num_envs
    cfg["environment"]["seedd"] = int(cfg["environment"]["seedd"]) + 7919 * rank   # each rank owns different robots
    # run_bp_v5.py:205-207: the environment sub-tree is dumped to a string and parsed again on the native side
    env = Environment(FlexibleGymEnv(__RSCDIR__, yaml.safe_dump(cfg["environment"]), device=local_rank))

    if args.flag_output:
        model = PPO2.load(args.trained_model, env=env)
        from high_speed_quadrupedal_locomotion_by_irrl_amd.checkpoint import export_actor_csv
        name = os.path.splitext(os.path.basename(args.trained_model))[0]
        print("exported to", export_actor_csv(model, os.path.join(os.getcwd(), "model", name)))
        return

    if not args.train:
        raise SystemExit("--test (gamepad evaluation + plotting) is outside this engine's scope; see the module docstring")

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
        model = PPO2.load(args.pre_trained_model, env=env)      # run_bp_v5.py:244-248
        model.tensorboard_log = log_path
        model.learning_rate = args.learn_rate
    if saver:
        TensorboardLauncher(saver.data_dir + "/PPO2_1")
    model.learn(total_timesteps=args.max_iter, eval_every_n=args.eval_every_n, log_dir=log_path,
                record_video=bool(cfg.get("record_video", False)))
    if rank == 0 and log_path:
        print("final checkpoint:", model.save(log_path + "_final"))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
