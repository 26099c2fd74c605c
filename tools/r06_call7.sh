#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
{
for e in TSVPP_BILINEAR_ROWS_WAVES=9 X=1; do
python tools/nn_matrix.py --src 3840x2160 --sizes 224,160 --types NEAREST,BILINEAR --batches 64,256 --env $e --pmc 256
done
} > $O/nn_wide_segments_ab.txt 2>&1
cut -c1-330 $O/nn_wide_segments_ab.txt
