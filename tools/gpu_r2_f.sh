cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 900 python tools/cpu_config1.py --seconds 4 > gpurun_out/config1_gpu_box.json 2> gpurun_out/config1_gpu_box.err
echo done
