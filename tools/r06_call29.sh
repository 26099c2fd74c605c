#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-18s %-20s %-9s %-7s %-7s norm=%s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; c=r[\"config\"]; print(\"%9.0f fps %8.1f us frac %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:]))"; }
{ for e in X=1 TSVPP_SHAPE=64,4; do
  row $e 3840x2160:1280x720 AREA BGR24 PLANAR 1; row $e 3840x2160:1280x720 NEAREST BGR24 PLANAR 1; row $e 3840x2160:1280x720 BICUBIC BGR24 PLANAR 1; row $e 3840x2160:1280x720 BILINEAR BGR24 PLANAR 1
  row $e 2560x1440:1280x720 AREA BGR24 PLANAR 1; row $e 1920x1080:1280x720 NEAREST BGR24 PLANAR 1; row $e 1920x1080:2560x1440 BILINEAR BGR24 PLANAR 1; row $e 1920x1080:2560x1440 NEAREST BGR24 PLANAR 1; row $e 1920x1080:3840x2160 BILINEAR BGR24 PLANAR 1
  row $e 3840x2160:1280x720 AREA BGR24 PLANAR 0; row $e 1920x1080:1280x720 AREA NV12 MERGED 1; row $e 1280x720:1280x720 NEAREST HSV MERGED 1
done; } > $O/planar1280_shapes.txt 2>&1; sort -k2,7 -s $O/planar1280_shapes.txt
