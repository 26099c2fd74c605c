#!/bin/bash
# Round 6: point samplers on the row-segment kernel (A/B against vpp_point_kernel), and where BILINEAR should leave the LDS-staged kernel for it
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bilinear_rows.py tests/test_gpu_edges.py tests/test_reference_crcs.py -m gpu -x -q 2>&1 | tail -3
TSVPP_BILINEAR_ROWS=2 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
{
for e in TSVPP_BILINEAR_ROWS=3 TSVPP_BILINEAR_ROWS=1 TSVPP_BILINEAR_ROWS=2; do
python tools/nn_matrix.py --src 1920x1080 --sizes 224,256,300,416,512,640 --types NEAREST --batches 64,512 --env $e
python tools/nn_matrix.py --src 3840x2160 --sizes 256,300,416,640 --types NEAREST --batches 64,256 --env $e
done
for e in TSVPP_BILINEAR_ROWS=1 TSVPP_BILINEAR_ROWS=2; do
python tools/nn_matrix.py --src 1920x1080 --sizes 416,480,540,640,800 --types BILINEAR --batches 64,512 --env $e
python tools/nn_matrix.py --src 3840x2160 --sizes 1024,1280 --types BILINEAR --batches 64 --env $e
done
} > $O/nn_point_rows_ab.txt 2> $O/nn_point_rows_ab.err
cat $O/nn_point_rows_ab.txt | cut -c1-330; tail -3 $O/nn_point_rows_ab.err
