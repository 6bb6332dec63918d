cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s
python -m pytest tests/test_gpu_ppo.py -x -q > gpurun_out/s/pytest.log 2>&1; tail -2 gpurun_out/s/pytest.log
for v in 1 2 3; do
python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('update %.2f ms'%(d['update_s']*1e3))"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/s/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s/prof -- python3 $GRAFT_REPO_ROOT/tools/ppo_bench.py --policy lstm --envs 4096 --iters 2 --epochs 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/s/prof/*/*_kernel_stats.csv | head -1)
grep -E "heads_loss" $f | cut -c1-200
rm -f gpurun_out/s/prof/*/*_kernel_trace.csv
