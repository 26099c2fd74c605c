#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-18s %-20s %-9s %-7s %-7s norm=%s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:]))"; }
{ for g in 1280x720:1280x720 2560x1440:2560x1440 3840x2160:3840x2160 2048x1152:2048x1152 1024x576:1024x576 640x360:640x360; do for e in X=1 TSVPP_SHAPE=64,4 TSVPP_SHAPE=64,2 TSVPP_SHAPE=32,4 TSVPP_SHAPE=16,16; do
  row $e $g NEAREST BGR24 PLANAR 1; done; done
  for g in 1280x720:1280x720 3840x2160:3840x2160; do for e in X=1 TSVPP_SHAPE=64,4; do row $e $g NEAREST BGR24 PLANAR 0; row $e $g NEAREST NV12 MERGED 1; row $e $g NEAREST Y800 MERGED 1; done; done
} > $O/color_shapes.txt 2>&1; cat $O/color_shapes.txt
