#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r05_gpu_suite.txt
KNOBS="TSVPP_BILINEAR_ROWS=0
TSVPP_BILINEAR_ROWS=2
TSVPP_BILINEAR_ROWS_WAVES=1
TSVPP_POINT_RN=0
TSVPP_POINT_RN=2" bash tools/knob_matrix.sh > gpurun_out/r05_knob_matrix_new.txt 2>&1
