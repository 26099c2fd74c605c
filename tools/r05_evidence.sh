#!/bin/bash
# What produced profiles/r05_* of the FINAL sources (each block is one gpurun call; run from the repo root on the GPU box; afterwards, here:
#   for w in headline c1 c2 c3 c4 c5 area bicubic bicubic_u8m up2_u8m; do bash tools/save_profile.sh r05 $w; done   and copy gpurun_out/r05_* into profiles/).
# The same-box A/B files of the round (r05_rows_ab, r05_c3_diag, r05_table_ab, r05_point_rn_ab, r05_c4_shapes, r05_c4_diag, r05_rep2_ab, r05_up2_shapes, r05_bicubic_cols_u8_ab,
# r05_prn_nt_variants, r05_u8_nt_ab, r05_u8_sc1_ab, r05_st1_nt_variants, r05_knob_matrix_streaming) were written by the tools/r05_call.sh of their commit (see git log).
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
case "$1" in
1)  # the suite, the rocprofv3 passes of the final kernels, the bench lines
    python -m pytest tests -m gpu -q 2>&1 | tail -2 > gpurun_out/r05_gpu_suite.txt
    for w in headline c1 c2 c3 c4 c5; do bash tools/profile.sh $w --workload $w > /dev/null 2>&1; done
    bash tools/profile.sh area --resize AREA > /dev/null 2>&1; bash tools/profile.sh bicubic --resize BICUBIC > /dev/null 2>&1
    bash tools/profile.sh bicubic_u8m --custom 1920x1080:1280x720:BICUBIC:RGB24:MERGED:0 > /dev/null 2>&1
    bash tools/profile.sh up2_u8m --custom 960x540:1920x1080:BILINEAR:RGB24:MERGED:0 > /dev/null 2>&1
    ;;
2)  # bench lines with the traffic stamps of block 1 in place (copy profiles/traffic_latest.json first), matrices
    python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
    python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd.json 2> gpurun_out/r05_bench_driver_cmd.err
    for w in c1 c2 c3 c4 c5; do python bench.py --workload $w --no-others > gpurun_out/r05_bench_$w.json 2>/dev/null; done
    python bench.py --workload c3 --batch 64 --no-others --no-cpu-baseline > gpurun_out/r05_bench_c3_batch64.json 2>/dev/null
    python bench.py --workload c4 --batch 64 --no-others --no-cpu-baseline > gpurun_out/r05_bench_c4_batch64.json 2>/dev/null
    bash tools/matrix.sh > gpurun_out/r05_perf_matrix.txt 2>&1; bash tools/outmatrix.sh > gpurun_out/r05_output_matrix.txt 2>&1
    ;;
3)  # every test under the knob settings that select another kernel family or another code path of the round's new kernels
    KNOBS="TSVPP_FORCE_GATHER=1
TSVPP_BILINEAR_ROWS=0
TSVPP_BILINEAR_ROWS=2
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
TSVPP_RPT=2" bash tools/knob_matrix.sh > gpurun_out/r05_knob_matrix.txt 2>&1
    ;;
esac
