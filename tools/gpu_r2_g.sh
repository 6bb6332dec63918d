cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
bash tools/gpu_variants.sh
