#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
bash tools/matrix.sh > gpurun_out/r05_perf_matrix.txt 2>&1; bash tools/outmatrix.sh > gpurun_out/r05_output_matrix.txt 2>&1
