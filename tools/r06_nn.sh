#!/bin/bash
# Round 6: the NN-input-size cells (1080p -> 224 / 256 / 300 / 416 squared, every resize type) at 64 / 256 / 512 frames per launch, with the PMC bytes of each kernel
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
mkdir -p gpurun_out/r06
python tools/nn_matrix.py --src 1920x1080 --pmc 256 > gpurun_out/r06/nn_matrix_1080p.txt 2> gpurun_out/r06/nn_matrix_1080p.err
cat gpurun_out/r06/nn_matrix_1080p.txt; tail -5 gpurun_out/r06/nn_matrix_1080p.err
