#!/bin/bash
# round-2 GPU call 19: four-row tiles for the LDS-staged column-per-lane AREA kernel (20 KiB of LDS: 7 workgroups per CU)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
TSVPP_AREA_COLS_ROWS=4 TSVPP_AREA_COLS_LDS=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_reference_crcs.py -m gpu -x -q > $O/pytest_gpu_rows4.log 2>&1; echo "pytest ROWS=4 LDS=2 rc=$?"; tail -2 $O/pytest_gpu_rows4.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in 1920x1080:224x224 3840x2160:384x384 1920x1080:300x300 3840x2160:224x224 1920x1080:192x192; do
  for e in "TSVPP_AREA_COLS_LDS=2 TSVPP_AREA_COLS_ROWS=8" "TSVPP_AREA_COLS_LDS=2 TSVPP_AREA_COLS_ROWS=4"; do
    echo -n "$c AREA f32 planar $e: "; one "$e" --custom $c:AREA:RGB24:PLANAR:1
    echo -n "$c AREA f32 planar $e alias=3: "; one "$e" --alias 3 --custom $c:AREA:RGB24:PLANAR:1
    echo -n "$c AREA u8 merged $e: "; one "$e" --custom $c:AREA:RGB24:MERGED:0
  done
done
} 2>&1 | tee $O/call19.txt
