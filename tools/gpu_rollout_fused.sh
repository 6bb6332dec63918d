# one launch per rollout step (env.step + policy step fused) against the two-launch sequence: tests, then the PPO bench both ways
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_ppo.py -x -q > gpurun_out/rollout_fused_pytest.log 2>&1; tail -3 gpurun_out/rollout_fused_pytest.log
for r in 1 2; do
IRRL_ROLLOUT_FUSED=1 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 > gpurun_out/rollout_fused_$r.json
python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 > gpurun_out/rollout_two_$r.json
done
python3 - <<'PY'
import json
for n in ("fused_1","two_1","fused_2","two_2"):
    d=json.loads(open("gpurun_out/rollout_%s.json"%n).read())
    print(n, "rollout %.2f ms update %.2f ms  %.3f it/s  %.1f M env-steps/s in rollout"%(d["rollout_s"]*1e3,d["update_s"]*1e3,d["ppo_iters_per_sec"],d["env_steps_per_sec_in_rollout"]/1e6))
PY
