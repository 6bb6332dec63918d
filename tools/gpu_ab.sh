cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 1 1; do
  timeout 300 python bench.py --cpu-seconds 0 --ppo-iters 0 --steps 3000 2>/dev/null | grep metric | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M env-steps/s', round(d['roofline']['avg_launch_us'],2), 'us')" >> gpurun_out/ab.log
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/pytest_ab.log 2>&1
echo done
