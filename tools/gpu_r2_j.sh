# pre-drawn step noise + rollout variants: GPU tests, training-cfg bench, PPO bench (default two launches, and the one-launch option)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/j
python -m pytest tests -x -q -m gpu > gpurun_out/j/pytest_gpu.log 2>&1; tail -2 gpurun_out/j/pytest_gpu.log
python bench.py --cpu-seconds 0 --ppo-iters 0 --steps 2000 --cfg default_cfg.yaml 2>/dev/null | grep metric > gpurun_out/j/bench_train_cfg.json
python bench.py --cpu-seconds 0 --ppo-iters 0 --steps 2000 2>/dev/null | grep metric > gpurun_out/j/bench_imitation.json
for r in 1 2; do
python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 > gpurun_out/j/ppo_$r.json
done
IRRL_ROLLOUT_FUSED=1 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 > gpurun_out/j/ppo_onelaunch.json
python3 - <<'PY'
import json
for n in ("bench_train_cfg","bench_imitation"):
    d=json.loads(open("gpurun_out/j/%s.json"%n).read()); print(n, round(d["ms_per_step"]*1e3,2), "us/step", round(d["value"]/1e6,1), "M")
for n in ("ppo_1","ppo_2","ppo_onelaunch"):
    d=json.loads(open("gpurun_out/j/%s.json"%n).read())
    print(n, "rollout %.2f ms update %.2f ms  %.3f it/s  %.1f M env-steps/s in rollout"%(d["rollout_s"]*1e3,d["update_s"]*1e3,d["ppo_iters_per_sec"],d["env_steps_per_sec_in_rollout"]/1e6))
PY
