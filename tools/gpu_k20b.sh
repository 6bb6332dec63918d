cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/k20b.log
for r in 1 2 3 4 5; do
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 --check-steps 0 2>/dev/null | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows prepared', round(d['ms_per_step']*1e3,2), 'us/step wall,', round(d['roofline']['avg_launch_us'],2), 'us/step events,', round(d['value']/1e6,1), 'M')" >> gpurun_out/k20b.log
done
python -m pytest tests/test_gpu_bench.py -x -q 2>&1 | tail -2 >> gpurun_out/k20b.log
cat gpurun_out/k20b.log
