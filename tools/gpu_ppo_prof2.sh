cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ppo_r2b
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ppo_r2b -- python3 $R/tools/ppo_bench.py --policy lstm --envs 4096 --iters 2 --epochs 2 > $R/gpurun_out/rocprof_ppo2.log 2>&1
cd $R
f=$(ls -t gpurun_out/prof_ppo_r2b/*/*_kernel_trace.csv | head -1)
python3 tools/ppo_timeline.py $f --epoch -1 > gpurun_out/ppo_timeline.log 2>&1
s=$(ls -t gpurun_out/prof_ppo_r2b/*/*_kernel_stats.csv | head -1)
cp $s gpurun_out/ppo_kernel_stats_latest.csv
# keep the merge small: drop the big trace
rm -f gpurun_out/prof_ppo_r2b/*/*_kernel_trace.csv
cat gpurun_out/ppo_timeline.log
