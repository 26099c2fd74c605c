#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
bash tools/r06_evidence.sh 2
QUICK=1 bash tools/profile.sh area300 --custom 1920x1080:300x300:AREA:RGB24:PLANAR:1 > /dev/null 2>&1
QUICK=1 bash tools/profile.sh bil224_4k --custom 3840x2160:224x224:BILINEAR:RGB24:PLANAR:1 > /dev/null 2>&1
QUICK=1 bash tools/profile.sh area224 --custom 1920x1080:224x224:AREA:RGB24:PLANAR:1 > /dev/null 2>&1
QUICK=1 bash tools/profile.sh uyvy720 --custom 1920x1080:1280x720:BILINEAR:UYVY:MERGED:0 > /dev/null 2>&1
QUICK=1 bash tools/profile.sh bicubic480 --custom 1080x608:480x360:BICUBIC:RGB24:PLANAR:1 > /dev/null 2>&1
tail -c 600 gpurun_out/r06_bench_driver_cmd.json; du -sh gpurun_out/prof_*
