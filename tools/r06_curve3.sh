#!/bin/bash
# Round 6, third pass: replay cache + pinned issuing threads + size rule; A/B of the option's values (1 both, 2 barrier-free only, 3 second stream only).
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
C=./tensor-stream_amd/lib/vpp_curve
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite3.txt 2>&1; tail -3 $O/gpu_suite3.txt
TSVPP_REPLAY=0 timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite3_noreplay.txt 2>&1; tail -3 $O/gpu_suite3_noreplay.txt
HL="1920 1080 2048 0 0 0 0 1280 720 1 2 0 1 14169600"
{
echo "# headline through the consumer pool (1xc, 2xc, 4xc), option values 0 / 1 / 2 / 3; replay on"
for v in 0 1 2 3; do timeout 300 $C $HL 1,2,4,8,16,32,64 1xc,2xc,4xc 25 $v; done
echo "# the same, option 0 / 1, TSVPP_REPLAY=0"
for v in 0 1; do TSVPP_REPLAY=0 timeout 300 $C $HL 1,2,4,8 1xc,4xc 25 $v; done
echo "# unpinned, option 0 / 1"
for v in 0 1; do VPP_CURVE_NO_PIN=1 timeout 300 $C $HL 1,2,4,8 1xc,4xc 25 $v; done
} > $O/curve_values.txt 2>&1
timeout 1200 python bench.py --curve-only headline,c3,c4 > $O/curve_third.json 2> $O/curve_third.err; tail -c 600 $O/curve_third.err
timeout 120 ./tensor-stream_amd/lib/vpp_latency > $O/latency3.json 2>&1
