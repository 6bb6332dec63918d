cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/n
for r in 1 2 3; do
python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('update %.2f ms rollout %.2f ms'%(d['update_s']*1e3, d['rollout_s']*1e3))"
done
