cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/k
python -m pytest tests/test_gpu_ppo.py -x -q > gpurun_out/k/pytest_ppo.log 2>&1; tail -2 gpurun_out/k/pytest_ppo.log
for r in 1 2; do
python tools/ppo_bench.py --policy lstm --envs 4096 --iters 4 2>/dev/null | tail -1 > gpurun_out/k/ppo_$r.json
done
L=$PWD/high_speed_quadrupedal_locomotion_by_irrl_amd/csrc/_variants/libirrl_env_pol.so
IRRL_ENV_LIB=$L python tools/rollout_phases.py 2>/dev/null > gpurun_out/k/phases.log
python3 - <<'PY'
import json
for n in ("ppo_1","ppo_2"):
    d=json.loads(open("gpurun_out/k/%s.json"%n).read())
    print(n, "rollout %.2f ms update %.2f ms  %.3f it/s  %.1f M env-steps/s in rollout"%(d["rollout_s"]*1e3,d["update_s"]*1e3,d["ppo_iters_per_sec"],d["env_steps_per_sec_in_rollout"]/1e6))
PY
cat gpurun_out/k/phases.log
