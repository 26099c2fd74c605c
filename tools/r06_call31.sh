#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-22s %-9s %-7s %-7s %s " $1 $2 $3 $4 $5
  python bench.py --custom $1:$2:$3:$4:$5 --steps 8 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:]))"; }
{ for w in 1360 1362 1364 1366 1368 1370 1376; do row 1920x1080:${w}x768 BILINEAR BGR24 PLANAR 1; done
  for w in 1360 1362 1364 1366 1368 1376; do row 1920x1080:${w}x768 BILINEAR BGR24 MERGED 1; done
  for w in 1360 1362 1364 1366 1368 1376; do row 1920x1080:${w}x768 BILINEAR BGR24 PLANAR 0; done
  for w in 1364 1366 1368; do row 1920x1080:${w}x768 BILINEAR BGR24 MERGED 0; row 1920x1080:${w}x768 NEAREST BGR24 PLANAR 1; row 1920x1080:${w}x768 AREA BGR24 PLANAR 1; row ${w}x768:${w}x768 NEAREST BGR24 PLANAR 1; done
} > $O/width_4k2.txt 2>&1; cat $O/width_4k2.txt
