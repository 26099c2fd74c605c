#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-12s %-22s %-7s %-7s %s " "$1" $2 $3 $4 $5
  env $1 python bench.py --custom $2:NEAREST:$3:$4:$5 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for e in X=1 TSVPP_NT=5; do for g in 1366x768:1366x768 1918x1080:1918x1080 854x480:854x480 1360x768:1360x768 1280x720:1280x720 1920x1080:1920x1080 1600x900:1600x900 960x540:960x540; do row $e $g RGB24 PLANAR 0; done; done; } > $O/color_u8_planar_after.txt 2>&1; sort -k2,5 -s $O/color_u8_planar_after.txt
