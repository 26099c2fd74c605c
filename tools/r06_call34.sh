#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-12s %-22s %-9s %-7s %-7s %s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for e in X=1 TSVPP_NT=1; do
  for g in 1920x1080:1366x768 1920x1080:854x480 3840x2160:1366x768 3840x2160:854x480 1280x720:1366x768 1280x720:854x480 1920x1080:1600x900 3840x2160:1600x900; do for rt in NEAREST BILINEAR BICUBIC AREA; do row $e $g $rt BGR24 PLANAR 1; done; done
  row $e 1366x768:1366x768 NEAREST BGR24 PLANAR 1; row $e 854x480:854x480 NEAREST BGR24 PLANAR 1; row $e 1600x900:1600x900 NEAREST BGR24 PLANAR 1
done; } > $O/row_alignment_after.txt 2>&1; sort -k2,4 -s $O/row_alignment_after.txt
