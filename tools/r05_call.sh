#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r05_bench_gpu_tests.txt
for w in c3 c4; do bash tools/profile.sh $w --workload $w > gpurun_out/prof_$w.log 2>&1; done
