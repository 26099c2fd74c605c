#!/bin/bash
# round-2 GPU call 22: geometry-table variant of the 2x2-tap kernel (TSVPP_GEO=0/1): parity suite, then same-box A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in "1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0" "1920x1080:1280x720:BILINEAR:RGB24:MERGED:0" "1920x1080:1280x720:BILINEAR:BGR24:PLANAR:1" "1920x1080:1280x720:BILINEAR:RGB24:MERGED:1" \
           "3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:0" "1280x720:1920x1080:AREA:RGB24:MERGED:0" \
           "1920x1080:1366x768:BILINEAR:RGB24:PLANAR:0" "1920x1080:1600x900:BILINEAR:RGB24:PLANAR:1"; do
  for e in "TSVPP_GEO=0" "TSVPP_GEO=1" "TSVPP_GEO=1 TSVPP_RPT=4"; do
    echo -n "$c $e: "; one "$e" --custom $c
  done
done
for c in "1280x720:1920x1080:BILINEAR:RGB24:PLANAR:0" "1920x1080:1000x562:BILINEAR:RGB24:PLANAR:0" "1920x1080:1152x648:BILINEAR:RGB24:PLANAR:1"; do
  for e in "TSVPP_BILINEAR_WIN=1" "TSVPP_BILINEAR_WIN=2" "TSVPP_BILINEAR_WIN=2 TSVPP_GEO=0"; do
    echo -n "$c $e: "; one "$e" --custom $c
  done
done
} 2>&1 | tee $O/call22.txt
