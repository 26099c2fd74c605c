#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-22s %-20s %-9s %-7s %-7s norm=%s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for e in X=1 TSVPP_TILE_ORDER=3 TSVPP_TILE_ORDER=4 TSVPP_TILE_ORDER=5 TSVPP_TILE_ORDER=1 TSVPP_TILE_ORDER=2; do
  row $e 1920x1080:1280x720 BICUBIC RGB24 PLANAR 1; row $e 3840x2160:1920x1080 BICUBIC RGB24 PLANAR 1; row $e 1920x1080:1280x720 BICUBIC RGB24 MERGED 0
  row $e 1920x1080:1280x720 BILINEAR RGB24 MERGED 0; row $e 1920x1080:1280x720 BILINEAR YUV444 MERGED 0
done; } > $O/r32_tile_order.txt 2>&1; cat $O/r32_tile_order.txt
