#!/bin/bash
# Round 6, first contact with the small-launch regime: the GPU suite, the launch curve (1 .. 64 frames per launch) of the headline / C3 / C4 with and without
# TSVPP_OPT_INPUTS_READY, other consumer shapes, and workgroup shapes at small n.  gpurun --timeout 1500 -- bash tools/r06_curve.sh
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
C=./tensor-stream_amd/lib/vpp_curve
timeout 600 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; tail -2 $O/gpu_suite.txt
timeout 900 python bench.py --curve-only headline,c3,c4 > $O/curve_first.json 2> $O/curve_first.err; tail -c 600 $O/curve_first.err
HL="1920 1080 2048 0 0 0 0 1280 720 1 2 0 1 14169600"
{
echo "# headline, consumer shapes (threads x streams per thread), in-order then inputs-ready"
timeout 300 $C $HL 1,2,4,8,16 1x1,1x2,1x4,2x1,2x2,8x1 20 0
timeout 300 $C $HL 1,2,4,8,16 1x1,1x2,1x4,2x1,2x2,8x1 20 1
} > $O/curve_modes.txt 2>&1
{
echo "# headline, workgroup shapes at small n (TSVPP_SHAPE=tx,ty; default 64,4 = 256 x 8 pixel tiles)"
for sh in 64,4 32,4 64,2 32,8 16,4 64,1 32,2; do echo "## shape $sh"; TSVPP_SHAPE=$sh timeout 200 $C $HL 1,2,4,8 1x1,4x1 15 0; TSVPP_SHAPE=$sh timeout 200 $C $HL 1,2,4,8 1x1 15 1; done
} > $O/curve_shapes.txt 2>&1
timeout 120 ./tensor-stream_amd/lib/vpp_latency > $O/latency.json 2>&1
ls -la $O
