#!/bin/bash
# Round 6: knob A/Bs on the NN-input cells below 0.60, and the 4K rows
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
{
python tools/nn_matrix.py --src 1920x1080 --sizes 416,300 --types BILINEAR --batches 64,512 --env TSVPP_BILINEAR_ROWS=2
python tools/nn_matrix.py --src 1920x1080 --sizes 300,416,256 --types AREA --batches 64,512 --env TSVPP_AREA_STREAM=2
python tools/nn_matrix.py --src 1920x1080 --sizes 300,416 --types AREA --batches 64,512 --env TSVPP_AREA_COLS_ROWS=8
python tools/nn_matrix.py --src 1920x1080 --sizes 256,300 --types BICUBIC --batches 64,512 --env TSVPP_BICUBIC_ROWS=16
python tools/nn_matrix.py --src 1920x1080 --sizes 256,300 --types BICUBIC --batches 64,512 --env TSVPP_BICUBIC_DMA=0
python tools/nn_matrix.py --src 3840x2160 --pmc 256 --batches 64,256
} > $O/nn_matrix_ab.txt 2> $O/nn_matrix_ab.err
cat $O/nn_matrix_ab.txt; tail -3 $O/nn_matrix_ab.err
