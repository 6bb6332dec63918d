# direct-launch rollout (irrl_lstm_rollout) vs the hipGraph one: tests, then the PPO bench both ways
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ppo.py tests/test_gpu_bench.py -x -q > gpurun_out/rollout_direct_pytest.log 2>&1; tail -3 gpurun_out/rollout_direct_pytest.log
for r in 1 2; do
python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 > gpurun_out/rollout_direct_$r.json
IRRL_ROLLOUT_LAUNCH=graph python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 > gpurun_out/rollout_graph_$r.json
done
python3 - <<'PY'
import json
for n in ("direct_1","graph_1","direct_2","graph_2"):
    d=json.loads(open("gpurun_out/rollout_%s.json"%n).read())
    print(n, "rollout %.2f ms update %.2f ms  %.3f it/s  %.1f M env-steps/s in rollout"%(d["rollout_s"]*1e3,d["update_s"]*1e3,d["ppo_iters_per_sec"],d["env_steps_per_sec_in_rollout"]/1e6))
PY
