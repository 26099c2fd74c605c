#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# Round-3 evidence: rocprofv3 kernel-trace + PMC passes of every bench workload on ONE box, one commit (run through gpurun):
#   tools/r03_evidence.sh        -> gpurun_out/prof_<tag>/, then tools/save_profile.sh r03 <tag> copies the summaries into profiles/
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tools/profile.sh headline > /dev/null 2>&1
tools/profile.sh c2 --workload c2 > /dev/null 2>&1
tools/profile.sh c3 --workload c3 > /dev/null 2>&1
tools/profile.sh c4 --workload c4 > /dev/null 2>&1
tools/profile.sh c5 --workload c5 > /dev/null 2>&1
tools/profile.sh area --resize AREA > /dev/null 2>&1
tools/profile.sh bicubic --resize BICUBIC > /dev/null 2>&1
tools/profile.sh bicubic480 --custom 1080x608:480x360:BICUBIC:RGB24:PLANAR:1 > /dev/null 2>&1
tools/profile.sh bicubicup --custom 1280x720:1920x1080:BICUBIC:RGB24:PLANAR:1 > /dev/null 2>&1
tools/profile.sh area224 --custom 1920x1080:224x224:AREA:RGB24:PLANAR:1 > /dev/null 2>&1
tools/profile.sh uyvy720 --custom 1920x1080:1280x720:BILINEAR:UYVY:MERGED:0 > /dev/null 2>&1
for t in headline c2 c3 c4 c5 area bicubic bicubic480 bicubicup area224 uyvy720; do head -3 gpurun_out/prof_$t/kt/kt_kernel_stats.csv | tail -2 | cut -c1-160; done
