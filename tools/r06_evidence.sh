#!/bin/bash
# What produces profiles/r06_* of the round's sources (each block is one gpurun call; run from the repo root on the GPU box).  Afterwards, here:
#   for w in headline c1 c2 c3 c4 c5 area bicubic nearest; do bash tools/save_profile.sh r06 $w; done   and copy gpurun_out/r06_* into profiles/.
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
export TMPDIR=/tmp
case "$1" in
1)  # the suite, the rocprofv3 passes of the final kernels (kernel trace + FETCH_SIZE / WRITE_SIZE in their own passes: QUICK=1 skips the SQ / LDS passes)
    python -m pytest tests -m gpu -q 2>&1 | tail -2 > gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_gpu_suite.txt
    for w in headline c1 c2 c3 c4 c5; do QUICK=$QUICK bash tools/profile.sh $w --workload $w > /dev/null 2>&1; done
    QUICK=$QUICK bash tools/profile.sh area --resize AREA > /dev/null 2>&1; QUICK=$QUICK bash tools/profile.sh bicubic --resize BICUBIC > /dev/null 2>&1
    QUICK=$QUICK bash tools/profile.sh nearest --resize NEAREST > /dev/null 2>&1
    du -sh gpurun_out/prof_*
    ;;
2)  # bench lines with the traffic stamps of block 1 in place (profiles/traffic_latest.json), matrices
    python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
    python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.json 2> gpurun_out/r06_bench_driver_cmd.err
    for w in c1 c2 c3 c4 c5; do python bench.py --workload $w --no-others > gpurun_out/r06_bench_$w.json 2>/dev/null; done
    python bench.py --workload c5 --consumers 64 --steps 50 --warmup 10 > gpurun_out/r06_bench_c5_consumers64.json 2>/dev/null
    bash tools/matrix.sh > gpurun_out/r06_perf_matrix.txt 2>&1; bash tools/outmatrix.sh > gpurun_out/r06_output_matrix.txt 2>&1
    python tools/nn_matrix.py --src 1920x1080 --pmc 256 > gpurun_out/r06_nn_matrix.txt 2>/dev/null
    python tools/nn_matrix.py --src 3840x2160 --pmc 256 --batches 64,256 > gpurun_out/r06_nn_matrix_4k.txt 2>/dev/null
    ;;
3)  # every test under the knob settings that select another kernel family or another code path
    KNOBS="${KNOBS:-TSVPP_FORCE_GATHER=1
TSVPP_REPLAY=0
TSVPP_BILINEAR_ROWS=0
TSVPP_BILINEAR_ROWS=2
TSVPP_BILINEAR_ROWS=3
TSVPP_POINT_RN=0
TSVPP_POINT_RN=2
TSVPP_BICUBIC_U8X=0
TSVPP_BICUBIC_COLS=2 TSVPP_BICUBIC_U8X=2
TSVPP_R32=0
TSVPP_R32=2
TSVPP_NT=0
TSVPP_NT=2
TSVPP_DMA=0
TSVPP_BILINEAR_INT=0
TSVPP_BICUBIC_INT=0
TSVPP_BICUBIC_COLS=0
TSVPP_AREA_BOX=0
TSVPP_AREA_STREAM=2
TSVPP_GEO=0
TSVPP_TAIL_SHIFT=0
TSVPP_TILE_ORDER=1
TSVPP_SHAPE=32,8
TSVPP_RPT=2}" bash tools/knob_matrix.sh > gpurun_out/r06_knob_matrix.txt 2>&1
    cat gpurun_out/r06_knob_matrix.txt
    ;;
esac
