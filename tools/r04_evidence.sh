#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# What produced profiles/r04_* (each block is one gpurun call; run from the repo root on the GPU box, copy gpurun_out/* into profiles/ afterwards).
# 1. the suite, the rocprofv3 passes of the final kernels, the bench lines, the matrices
python -m pytest tests -m gpu -q 2>&1 | tail -2 > gpurun_out/r04_gpu_suite.txt
for w in headline c2 c3 c4 c5; do bash tools/profile.sh $w --workload $w; done
bash tools/profile.sh area --resize AREA; bash tools/profile.sh bicubic --resize BICUBIC
bash tools/profile.sh bicubic_u8m --custom 1920x1080:1280x720:BICUBIC:RGB24:MERGED:0
#    (here, not on the box: for w in headline c2 c3 c4 c5 area bicubic bicubic_u8m; do bash tools/save_profile.sh r04 $w; done)
python bench.py > gpurun_out/bench_default.json; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json
for w in c2 c3 c4 c5; do python bench.py --workload $w --no-others > gpurun_out/bench_$w.json; done
bash tools/matrix.sh > gpurun_out/matrix.txt; bash tools/outmatrix.sh > gpurun_out/outmatrix.txt
# 2. same-box A/B files
bash tools/bicubic_ab.sh "TSVPP_BICUBIC_INT=1" "TSVPP_BICUBIC_INT=2" > gpurun_out/bicubic_r32_ab.txt   # streaming vs LDS integer BICUBIC kernel
bash tools/abc.sh "1920x1080:1280x720:AREA:BGR24:PLANAR:1 3840x2160:1920x1080:AREA:RGB24:PLANAR:1" "TSVPP_BILINEAR_INT=2" "TSVPP_BILINEAR_INT=1" > gpurun_out/tap22_ab.txt
bash tools/pmc.sh bcr32_f32 "TSVPP_X=0" --resize BICUBIC                                                 # PMC passes of one workload (dispatched kernel only)
# 3. every test under every knob
bash tools/knob_matrix.sh > gpurun_out/knob_matrix.txt
# 4. the last day: outputs 4 k + 2 columns wide (the shifted tile column against the row-tail launch), its tests under every knob
bash tools/tail_probe.sh > gpurun_out/tail_probe3.txt
TESTS="tests/test_gpu_tail_shift.py tests/test_gpu_geo.py" bash tools/knob_matrix.sh > gpurun_out/knob_matrix_tail.txt
KNOBS="TSVPP_TAIL_SHIFT=0
TSVPP_FORCE_GATHER=1
TSVPP_DMA=0" bash tools/knob_matrix.sh > gpurun_out/knob_matrix_tail2.txt
#    (blocks 1 was run again afterwards with QUICK unset: every kernel's tile origin changed, so every PMC entry was re-taken)
# 5. the last hours: launches on several streams (diagnostic), up-scales with uint8 outputs before / after the streaming BILINEAR 1 : 2 kernel, its BICUBIC twin (commit ed4f57e)
bash tools/streams_probe.sh > gpurun_out/streams_probe.txt
bash tools/abc.sh "960x540:1920x1080:BILINEAR:RGB24:MERGED:0 960x540:1920x1080:BILINEAR:RGB24:PLANAR:0 1920x1080:3840x2160:BILINEAR:RGB24:MERGED:0" "TSVPP_R32=0" "TSVPP_X=0" > gpurun_out/up2_ab.txt
python -m pytest tests -m gpu -q 2>&1 | tail -2 > gpurun_out/r04_gpu_suite.txt   # 1143 passed: the state of the final commit
