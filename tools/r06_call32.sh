#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-14s %-22s %-9s %-7s %-7s %s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 8 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:]))"; }
{ for e in X=1 TSVPP_NT=0 TSVPP_NT=1 TSVPP_NT=2; do for w in 1366 1364 1368 1376; do row $e 1920x1080:${w}x768 BILINEAR BGR24 PLANAR 1; done; row $e 1366x768:1366x768 NEAREST BGR24 PLANAR 1; row $e 1920x1080:854x480 BILINEAR BGR24 PLANAR 1; row $e 1920x1080:1366x768 BICUBIC BGR24 PLANAR 1; row $e 1920x1080:1366x768 AREA BGR24 PLANAR 1; row $e 1920x1080:300x300 BILINEAR BGR24 PLANAR 1; row $e 1920x1080:224x224 BILINEAR BGR24 PLANAR 1; done
} > $O/width_nt.txt 2>&1; sort -k2,6 -s $O/width_nt.txt
