#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
C="--custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0"
bash tools/pmc.sh u8p_geo0 "TSVPP_GEO=0" $C
bash tools/pmc.sh u8p_geo1 "TSVPP_GEO=1" $C
bash tools/pmc.sh f32_geo1 "TSVPP_GEO=1" --workload headline
