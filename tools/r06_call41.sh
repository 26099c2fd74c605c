#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-12s %-22s %-7s %-7s %s " "$1" $2 $3 $4 $5
  env $1 python bench.py --custom $2:NEAREST:$3:$4:$5 --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for e in X=1 TSVPP_NT=1 TSVPP_NT=2; do for g in 1366x768:1366x768 1364x768:1364x768 1368x768:1368x768 854x480:854x480 1918x1080:1918x1080 1360x768:1360x768; do row $e $g BGR24 MERGED 1; row $e $g HSV MERGED 1; row $e $g NV12 MERGED 1; row $e $g Y800 MERGED 1; done; done; } > $O/row_alignment_color.txt 2>&1; sort -k2,5 -s $O/row_alignment_color.txt
