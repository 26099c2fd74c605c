#!/bin/bash
# anomaly hunt 2: merged uint8 (display-typical) and merged fp32, NV12 uint8, over output geometries x resize types, 64 frames per launch
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-20s %-9s %-6s %-7s %s " $1 $2 $3 $4 $5
  python bench.py --custom $1:$2:$3:$4:$5 --steps 6 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:]))"; }
{ for s in 1920x1080 3840x2160; do for d in 640x360 854x480 960x540 1280x720 1366x768 1600x900 1920x1080 2560x1440; do [ $s = $d ] && continue; for rt in NEAREST BILINEAR BICUBIC AREA; do row $s:$d $rt RGB24 MERGED 0; row $s:$d $rt RGB24 MERGED 1; done; done; done; } > $O/merged_hunt.txt 2>&1; cat $O/merged_hunt.txt
