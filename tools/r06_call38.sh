#!/bin/bash
# anomaly hunt 3: SOURCE geometry -- tight pitches (pitch = width: multiples of 16 or not) against pitches rounded up to 256 bytes, planar fp32 and merged uint8
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-14s %-22s %-9s %-6s %-7s %s " "$1" $2 $3 $4 $5 $6
  python bench.py --custom $2:$3:$4:$5:$6 $1 --steps 6 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:]))"; }
{ for s in 1366x768 1360x768 854x480 1920x1080 1918x1080; do for d in 640x360 1280x720 $s; do for rt in NEAREST BILINEAR BICUBIC AREA; do [ $s = $d ] && [ $rt != NEAREST ] && continue; for p in "" "--tight-pitch"; do row "$p" $s:$d $rt BGR24 PLANAR 1; row "$p" $s:$d $rt BGR24 MERGED 0; done; done; done; done; } > $O/pitch_hunt.txt 2>&1; cat $O/pitch_hunt.txt
