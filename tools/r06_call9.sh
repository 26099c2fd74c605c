#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
{
echo "# vpp_area_cols_kernel with tap row a + 1 in flight while tap row a is accumulated (round 6); before: 1080p -> 300^2 543 us / -> 416^2 639 us per 512 frames (profiles/r06_nn_matrix.txt)"
python tools/nn_matrix.py --src 1920x1080 --sizes 300,416 --types AREA --batches 64,512 --pmc 256
python tools/nn_matrix.py --src 1920x1080 --sizes 300,416 --types AREA --batches 64,512 --env TSVPP_AREA_COLS_ROWS=8
python tools/nn_matrix.py --src 1920x1080 --sizes 300,416 --types AREA --batches 64,512 --env TSVPP_AREA_COLS_ROWS=32
python tools/nn_matrix.py --src 3840x2160 --sizes 640,800 --types AREA --batches 64,256
} > $O/area_cols_ab.txt 2>&1
cut -c1-330 $O/area_cols_ab.txt
