#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
timeout 300 python -m pytest tests/test_gpu_bilinear_rows.py -m gpu -x -q 2>&1 | tail -3
bash tools/pmc.sh area300_cols "X=1" --custom 1920x1080:300x300:AREA:RGB24:PLANAR:1 --batch 256 --table > /dev/null 2>&1
bash tools/pmc.sh area300_stream "TSVPP_AREA_STREAM=2" --custom 1920x1080:300x300:AREA:RGB24:PLANAR:1 --batch 256 --table > /dev/null 2>&1
bash tools/pmc.sh bicubic300 "X=1" --custom 1920x1080:300x300:BICUBIC:RGB24:PLANAR:1 --batch 256 --table > /dev/null 2>&1
bash tools/pmc.sh bicubic_720_1080_u8 "X=1" --custom 1280x720:1920x1080:BICUBIC:RGB24:MERGED:0 --batch 64 > /dev/null 2>&1
cat gpurun_out/pmc_area300_cols.txt gpurun_out/pmc_area300_stream.txt gpurun_out/pmc_bicubic300.txt gpurun_out/pmc_bicubic_720_1080_u8.txt
