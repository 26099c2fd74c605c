#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_gpu_suite.txt
bash tools/outmatrix.sh > gpurun_out/r06_output_matrix.txt 2>&1
QUICK=1 bash tools/profile.sh uyvy720 --custom 1920x1080:1280x720:BILINEAR:UYVY:MERGED:0 > /dev/null 2>&1
QUICK=1 bash tools/profile.sh yuv444_720 --custom 1920x1080:1280x720:BILINEAR:YUV444:MERGED:0 > /dev/null 2>&1
QUICK=1 bash tools/profile.sh bicubic_u8m --custom 1920x1080:1280x720:BICUBIC:RGB24:MERGED:0 > /dev/null 2>&1
QUICK=1 bash tools/profile.sh up2_u8m --custom 960x540:1920x1080:BILINEAR:RGB24:MERGED:0 > /dev/null 2>&1
QUICK=1 bash tools/profile.sh bicubicup --custom 1280x720:1920x1080:BICUBIC:RGB24:PLANAR:1 > /dev/null 2>&1
du -sh gpurun_out/prof_*; grep -c . gpurun_out/r06_output_matrix.txt
