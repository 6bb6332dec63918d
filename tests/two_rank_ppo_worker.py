"""Worker of the N-rank == 1-rank tests (not a test module): one PPO iteration -- rollout through the product's Runner, one update
-- on this rank's env shard, saved to <out>/rank<r>of<w>.npz.  RANK / WORLD_SIZE / MASTER_* from the environment (gloo; every rank
on cuda:0 when `--device cuda`, the oracle-backed CPU double with `--device cpu`).  The shard of rank r is the pool with
EnvIdOffset = r * envs: the global env ids r * envs .. (r + 1) * envs - 1."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--envs", type=int, default=64)        # per rank
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--policy", default="lstm")
    ap.add_argument("--cfg", default="default_cfg.yaml")
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="gloo", help="gloo, or nccl (= RCCL: one rank per GPU; with WORLD_SIZE=1 a one-rank communicator)")
    ap.add_argument("--nminibatches", type=int, default=1)
    a = ap.parse_args()
    import torch
    import yaml
    from conftest import load_env_cfg
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    launched = world > 1 or a.backend == "nccl"
    if launched:
        if a.backend == "nccl":
            torch.cuda.set_device(0)
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        else:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = load_env_cfg(a.cfg, num_envs=a.envs, EnvIdOffset=rank * a.envs)
    from high_speed_quadrupedal_locomotion_by_irrl_amd.policies import CustomLSTMPolicy, MlpPolicy
    from high_speed_quadrupedal_locomotion_by_irrl_amd.ppo2 import PPO2, Runner
    if a.device == "cuda":
        import high_speed_quadrupedal_locomotion_by_irrl_amd as pkg
        from high_speed_quadrupedal_locomotion_by_irrl_amd.flexible_robot import FlexibleGymEnv
        from high_speed_quadrupedal_locomotion_by_irrl_amd.vec_env import TorchVecEnv
        env = TorchVecEnv(FlexibleGymEnv(pkg.__BLACKPANTHER_V55_RESOURCE_DIRECTORY__, yaml.safe_dump(cfg)))
    else:
        from oracle_torch_env import OracleTorchEnv
        env = OracleTorchEnv(cfg)
    lstm = a.policy == "lstm"
    model = PPO2(policy=CustomLSTMPolicy if lstm else MlpPolicy, env=env, gamma=0.99, n_steps=a.steps, ent_coef=0.0, learning_rate=1e-3,
                 vf_coef=0.5, max_grad_norm=0.5, lam=0.998, nminibatches=a.nminibatches, noptepochs=2, cliprange=0.2, verbose=0, seed=7)
    runner = Runner(env, model, a.steps, 0.99, 0.998)
    batch = runner.run()
    losses = model.update(batch, 1e-3, 0.2)
    out = {k: v.detach().cpu().numpy() for k, v in batch.items() if hasattr(v, "detach")}
    out["params"] = np.concatenate([p.reshape(-1) for p in model.get_parameter_list()])
    out["losses"] = losses.detach().cpu().numpy() if hasattr(losses, "detach") else np.asarray(losses)
    out["fused_rollout"] = np.array(int(bool(getattr(runner, "_fused", False))))
    out["collective"] = np.array(int(bool(model.collective)))
    out["backend"] = np.array(torch.distributed.get_backend() if launched else "none")
    np.savez(os.path.join(a.out, "rank%dof%d.npz" % (rank, world)), **out)
    if launched:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
