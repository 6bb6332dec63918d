cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ppo
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ppo -- python3 $GRAFT_REPO_ROOT/tools/ppo_bench.py --policy lstm --envs 4096 --iters 2 --epochs 2 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_ppo.log 2>&1
echo done
