#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-26s %-22s %-7s %-7s %s " "$1" $2 $3 $4 $5
  env $1 python bench.py --custom $2:AREA:$3:$4:$5 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for g in 1920x1080:854x480 3840x2160:1600x900 3840x2160:1366x768 1920x1080:600x400 1920x1080:720x480 1920x1080:800x450 3840x2160:1440x810 1920x1080:640x480 1280x720:480x270; do for e in X=1 TSVPP_AREA_DIRECT_FMIN=99 TSVPP_AREA_STREAM=2 "TSVPP_AREA_COLS=2" "TSVPP_AREA_DIRECT_FMIN=99 TSVPP_AREA_STREAM=0"; do row "$e" $g RGB24 PLANAR 1; done; done; } > $O/area_float_hunt.txt 2>&1; cat $O/area_float_hunt.txt
