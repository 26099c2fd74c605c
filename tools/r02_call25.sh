#!/bin/bash
# round-2 GPU call 25: workgroup shape x rows-per-thread sweep of the uint8 2x2-tap kernel with geometry tables; persistent variant
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
one() { env $1 timeout 120 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); r=json.loads(l); print(l[:300]) if 'value' not in r else print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in "1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0" "1920x1080:1280x720:BILINEAR:RGB24:MERGED:0"; do
  for e in "TSVPP_SHAPE=32,8 TSVPP_RPT=2" "TSVPP_SHAPE=32,4 TSVPP_RPT=2" "TSVPP_SHAPE=32,4 TSVPP_RPT=4" "TSVPP_SHAPE=16,8 TSVPP_RPT=2" "TSVPP_SHAPE=16,4 TSVPP_RPT=4" "TSVPP_SHAPE=64,2 TSVPP_RPT=2" "TSVPP_SHAPE=64,2 TSVPP_RPT=4" "TSVPP_SHAPE=64,4 TSVPP_RPT=2" "TSVPP_SHAPE=32,2 TSVPP_RPT=4" "TSVPP_GEO=0 TSVPP_PERSIST=2" "TSVPP_GEO=0 TSVPP_PERSIST=3"; do
    echo -n "$c $e: "; one "$e" --custom $c
  done
done
echo -n "alias=3 GEO=1: "; one "TSVPP_GEO=1" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0 --alias 3
echo -n "alias=3 GEO=0: "; one "TSVPP_GEO=0" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0 --alias 3
} 2>&1 | tee $O/call25.txt
