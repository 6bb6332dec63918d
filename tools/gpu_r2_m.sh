cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/m; rm -f gpurun_out/m/share.log
for sh in 00 10 20 01 11 02 00; do
IRRL_LSTM_BWD_SHARE=$sh python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share $sh update %.2f ms'%(d['update_s']*1e3))" >> gpurun_out/m/share.log
done
cat gpurun_out/m/share.log
