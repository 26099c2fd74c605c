#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-12s %-22s %-9s %-7s %-7s %s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 8 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:]))"; }
{ for e in X=1 TSVPP_NT=0; do for g in 1920x1080:1366x768 1920x1080:854x480 1366x768:1366x768; do row $e $g BILINEAR NV12 MERGED 1; row $e $g BILINEAR Y800 MERGED 1; row $e $g BILINEAR BGR24 MERGED 1; row $e $g BILINEAR HSV MERGED 1; row $e $g BILINEAR YUV444 MERGED 1; row $e $g BILINEAR UYVY MERGED 1; done; done; } > $O/row_alignment_flavours.txt 2>&1; sort -k2,6 -s $O/row_alignment_flavours.txt
