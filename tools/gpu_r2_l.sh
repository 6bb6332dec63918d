cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/l
python -m pytest tests/test_gpu_ppo.py -x -q > gpurun_out/l/pytest_ppo.log 2>&1; tail -2 gpurun_out/l/pytest_ppo.log
for r in 1 2; do
python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 > gpurun_out/l/ppo_new_$r.json
IRRL_LSTM_FWD_SPLIT=0 IRRL_LSTM_BWD_SHARE=0 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 > gpurun_out/l/ppo_old_$r.json
done
python3 - <<'PY'
import json
for n in ("ppo_new_1","ppo_old_1","ppo_new_2","ppo_old_2"):
    d=json.loads(open("gpurun_out/l/%s.json"%n).read())
    print(n, "rollout %.2f ms update %.2f ms  %.3f it/s  %.1f M env-steps/s in rollout"%(d["rollout_s"]*1e3,d["update_s"]*1e3,d["ppo_iters_per_sec"],d["env_steps_per_sec_in_rollout"]/1e6))
PY
