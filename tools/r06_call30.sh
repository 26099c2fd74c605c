#!/bin/bash
# anomaly hunt: planar fp32 (the NN-typical output) over output geometries x resize types, 64 frames per launch
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-20s %-9s " $1 $2
  python bench.py --custom $1:$2:BGR24:PLANAR:1 --steps 8 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f touched %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf.get(\"touched_frac\") or 0, rf[\"kernel\"][7:]))"; }
{ for s in 1920x1080 3840x2160 1280x720; do for d in 640x360 854x480 960x540 1024x576 1280x720 1366x768 1600x900 1920x1080 2560x1440; do [ $s = $d ] && continue; for rt in NEAREST BILINEAR BICUBIC AREA; do row $s:$d $rt; done; done; done; } > $O/planar_f32_hunt.txt 2>&1; cat $O/planar_f32_hunt.txt
