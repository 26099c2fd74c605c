#!/bin/bash
# round-2 GPU call 26: geometry tables with scalar row records (workgroups >= 64 thread tiles wide)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_geo.py tests/test_gpu_bilinear_int.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
one() { env $1 timeout 120 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); r=json.loads(l); print(l[:300]) if 'value' not in r else print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in "1920x1080:1280x720:BILINEAR:BGR24:PLANAR:1" "1920x1080:1280x720:BILINEAR:RGB24:MERGED:1" "3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:1"; do
  for e in "TSVPP_GEO=0" "TSVPP_GEO=2" "TSVPP_GEO=0" "TSVPP_GEO=2"; do echo -n "$c $e: "; one "$e" --custom $c; done
done
for c in "1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0" "1920x1080:1280x720:BILINEAR:RGB24:MERGED:0" "3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:0"; do
  for e in "TSVPP_GEO=0" "TSVPP_GEO=1" "TSVPP_GEO=1 TSVPP_SHAPE=64,4" "TSVPP_GEO=1 TSVPP_SHAPE=64,2 TSVPP_RPT=4"; do echo -n "$c $e: "; one "$e" --custom $c; done
done
} 2>&1 | tee $O/call26.txt
