#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd.json 2> gpurun_out/r05_bench_driver_cmd.err
python bench.py --workload c1 --no-others > gpurun_out/r05_bench_c1.json 2>/dev/null
python -m pytest tests/test_bench_gpu.py -q 2>&1 | tail -2 > gpurun_out/r05_bench_gpu_tests.txt
