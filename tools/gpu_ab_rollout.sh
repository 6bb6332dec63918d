cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do
for ov in "ContactSolver=2" "ContactSolver=0"; do
  IRRL_CFG_OVERRIDE=$ov timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ov ppo rollout', round(d['rollout_s']*1e3,2), 'ms update', round(d['update_s']*1e3,2), 'ms')" >> gpurun_out/ab_rollout.log
done
done
echo done
