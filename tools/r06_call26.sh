#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-18s %-20s %-9s %-7s %-7s norm=%s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:]))"; }
{ for g in 1280x720:1280x720 1920x1080:1920x1080 2048x1152:2048x1152 2560x1440:2560x1440 3840x2160:3840x2160 1920x1088:1920x1088 1920x1024:1920x1024; do
  row X=1 $g NEAREST BGR24 PLANAR 1; row X=1 $g NEAREST BGR24 MERGED 1; row X=1 $g NEAREST BGR24 MERGED 0; done
  for e in TSVPP_NT=0 TSVPP_NT=2 TSVPP_SHAPE=64,4 TSVPP_SHAPE=32,8 TSVPP_SHAPE=64,2 TSVPP_TILE_ORDER=1 TSVPP_TILE_ORDER=2 TSVPP_RPT=2 TSVPP_RPT=1; do row $e 1920x1080:1920x1080 NEAREST BGR24 PLANAR 1; done
} > $O/color_geoms.txt 2>&1; cat $O/color_geoms.txt
