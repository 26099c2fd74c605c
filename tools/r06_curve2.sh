#!/bin/bash
# Round 6, second pass on the small-launch regime: host cost of a launch, honest pools (reads > 3x the Infinity Cache), the consumer pool with and without
# TSVPP_OPT_INPUTS_READY, and the two effects of the option (second stream / no barrier bit) apart.  gpurun --timeout 1500 -- bash tools/r06_curve2.sh
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
C=./tensor-stream_amd/lib/vpp_curve
nproc > $O/launch_cost.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $O/launch_cost.txt
timeout 300 tools/bin/launch_cost >> $O/launch_cost.txt 2>&1; cat $O/launch_cost.txt
timeout 1200 python bench.py --curve-only headline,c3,c4 > $O/curve_second.json 2> $O/curve_second.err; tail -c 600 $O/curve_second.err
HL="1920 1080 2048 0 0 0 0 1280 720 1 2 0 1 14169600"
{
echo "# headline: the option's two effects apart (explicit streams): second stream alone (1x2, any_order 0), no barrier bit alone (1x1, any_order 1), both (1x2, any_order 1)"
timeout 300 $C $HL 1,2,4,8,16,64 1x1,1x2,1x3 25 0
timeout 300 $C $HL 1,2,4,8,16,64 1x1,1x2,1x3 25 1
} > $O/curve_effects.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > $O/gpu_suite2.txt 2>&1; tail -2 $O/gpu_suite2.txt
