#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-12s %-22s %-9s %-7s %-7s %s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 8 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:]))"; }
{ for e in X=1 TSVPP_NT=0; do
  for w in 1360 1368 1376; do row $e 1920x1080:${w}x768 BILINEAR BGR24 PLANAR 1; row $e 1920x1080:${w}x768 NEAREST BGR24 PLANAR 1; row $e 1920x1080:${w}x768 BILINEAR NV12 MERGED 1; row $e 1920x1080:${w}x768 BILINEAR Y800 MERGED 1; row $e 1920x1080:${w}x768 BILINEAR BGR24 MERGED 1; row $e 1920x1080:${w}x768 BILINEAR HSV MERGED 1; done
  row $e 2040x1152:1360x768 BICUBIC BGR24 PLANAR 1; row $e 2052x1152:1368x768 BICUBIC BGR24 PLANAR 1; row $e 2064x1152:1376x768 BICUBIC BGR24 PLANAR 1
  row $e 4080x2304:1360x768 AREA BGR24 PLANAR 1; row $e 4104x2304:1368x768 AREA BGR24 PLANAR 1; row $e 1920x1080:1366x768 BILINEAR BGR24 PLANAR 0; row $e 1920x1080:1366x768 BILINEAR BGR24 MERGED 0
  row $e 1920x1080:600x400 AREA BGR24 PLANAR 1; row $e 1920x1080:600x400 BILINEAR BGR24 PLANAR 1; row $e 1920x1080:416x416 BILINEAR BGR24 PLANAR 1; row $e 1920x1080:416x416 AREA BGR24 PLANAR 1; row $e 1920x1080:300x300 AREA BGR24 PLANAR 1; row $e 1920x1080:300x300 BICUBIC BGR24 PLANAR 1; row $e 1920x1080:300x300 NEAREST BGR24 PLANAR 1
done; } > $O/width_nt2.txt 2>&1; sort -k2,6 -s $O/width_nt2.txt
