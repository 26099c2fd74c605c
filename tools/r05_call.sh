#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_bunny.py -q 2>&1 | tail -4 > gpurun_out/r05_bunny_tests.txt
python bench.py --workload c1 --steps 30 --warmup 5 --no-others > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.err
