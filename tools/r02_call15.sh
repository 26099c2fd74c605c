#!/bin/bash
# round-2 GPU call 15: host-built divisor table for the LDS-staged column-per-lane AREA kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
TSVPP_AREA_COLS_LDS=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_reference_crcs.py -m gpu -x -q > $O/pytest_gpu_lds2.log 2>&1; echo "pytest LDS=2 rc=$?"; tail -2 $O/pytest_gpu_lds2.log
TSVPP_AREA_DIVTAB=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_reference_crcs.py -m gpu -x -q > $O/pytest_gpu_div0.log 2>&1; echo "pytest DIVTAB=0 rc=$?"; tail -2 $O/pytest_gpu_div0.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in 1920x1080:224x224 3840x2160:384x384 1920x1080:300x300 1920x1080:416x416 3840x2160:608x342 1280x720:224x224; do
  for e in "TSVPP_AREA_COLS_LDS=0" "TSVPP_AREA_COLS_LDS=2 TSVPP_AREA_DIVTAB=0" "TSVPP_AREA_COLS_LDS=2"; do
    echo -n "$c AREA f32 $e: "; one "$e" --custom $c:AREA:RGB24:PLANAR:1
  done
  echo -n "$c AREA f32 LDS=2 alias=3: "; one "TSVPP_AREA_COLS_LDS=2" --alias 3 --custom $c:AREA:RGB24:PLANAR:1
  echo -n "$c AREA u8 merged default: "; one "X=1" --custom $c:AREA:RGB24:MERGED:0
done
} 2>&1 | tee $O/call15.txt
