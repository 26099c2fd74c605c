#!/bin/bash
# round-2 GPU call 13: LDS-staged column-per-lane float AREA kernel vs the global-memory one
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in 1920x1080:224x224 1920x1080:300x300 1920x1080:416x416 3840x2160:608x342 1280x720:224x224 1920x1080:384x384 3840x2160:640x640; do
  for e in "TSVPP_AREA_COLS_LDS=0" "TSVPP_AREA_COLS_LDS=1" "TSVPP_AREA_COLS_LDS=1 TSVPP_AREA_COLS_ROWS=8" "TSVPP_AREA_COLS_LDS=1 TSVPP_AREA_COLS_ROWS=32"; do
    echo -n "$c AREA f32 $e: "; one "$e" --custom $c:AREA:RGB24:PLANAR:1
  done
  echo -n "$c AREA u8 merged default: "; one "X=1" --custom $c:AREA:RGB24:MERGED:0
  echo -n "$c AREA f32 default alias=3: "; one "X=1" --alias 3 --custom $c:AREA:RGB24:PLANAR:1
done
} 2>&1 | tee $O/call13.txt
