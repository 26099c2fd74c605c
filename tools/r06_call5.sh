#!/bin/bash
# Round 6, fifth pass: GPU suite after the ADVICE / hardening changes, latency tool (Release pool), the driver's bench command with the launch curve in the line.
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_suite5.txt 2>&1; tail -5 $O/gpu_suite5.txt
timeout 200 ./tensor-stream_amd/lib/vpp_latency > $O/latency5.json 2>&1; cat $O/latency5.json
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd5.json 2> $O/bench_driver_cmd5.err ) 2>&1 | tail -3; tail -c 400 $O/bench_driver_cmd5.err
