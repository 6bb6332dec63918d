#!/bin/bash
# round-2: meteorite (Crutial) parity on the GPU + regression check of the default bench line
mkdir -p gpurun_out/crutial
python -m pytest tests/test_gpu_parity.py -x -q -k "crutial or box_corner or layouts" -s > gpurun_out/crutial/pytest_crutial.log 2>&1
echo "pytest crutial rc=$?" >> gpurun_out/crutial/pytest_crutial.log
python bench.py --ppo-iters 0 > gpurun_out/crutial/bench_default.json 2> gpurun_out/crutial/bench_default.err
python bench.py --ppo-iters 0 --steps 2000 > gpurun_out/crutial/bench_2000.json 2>> gpurun_out/crutial/bench_default.err
python bench.py --ppo-iters 0 --steps 2000 --set Crutial=true > gpurun_out/crutial/bench_crutial.json 2>> gpurun_out/crutial/bench_default.err
python -m pytest tests -x -q -m gpu > gpurun_out/crutial/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" >> gpurun_out/crutial/pytest_gpu.log
tail -3 gpurun_out/crutial/pytest_crutial.log; tail -3 gpurun_out/crutial/pytest_gpu.log; cat gpurun_out/crutial/bench_*.json | cut -c1-400
