# round-2 evidence run: tests, bench lines (3 configs + driver style), rocprof stats (bench + PPO), PMC passes, sweeps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_* gpurun_out/prof_r2_bench gpurun_out/prof_ppo_r2
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; lscpu | grep "Model name") > gpurun_out/box.log 2>&1
timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_style.log 2>&1
timeout 600 python bench.py --cfg bp5_terrain.yaml --cpu-seconds 0 --ppo-iters 0 --steps 2000 > gpurun_out/bench_terrain.log 2>&1
timeout 600 python bench.py --cfg default_cfg.yaml --cpu-seconds 0 --ppo-iters 0 --steps 2000 > gpurun_out/bench_train.log 2>&1
bash tools/gpu_sweep.sh
bash tools/gpu_pmc.sh
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --ppo-iters 0 --check-steps 500 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ppo_r2 -- python3 $GRAFT_REPO_ROOT/tools/ppo_bench.py --policy lstm --envs 4096 --iters 2 --epochs 2 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_ppo.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python tools/ppo_bench.py --policy mlp --envs 4096 --iters 3 > gpurun_out/ppo_mlp.log 2>&1
timeout 300 python tools/ppo_bench.py --policy lstm --envs 4096 --iters 3 > gpurun_out/ppo_lstm.log 2>&1
echo done
