#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in 3840x2160:160x90 3840x2160:150x84 3840x2160:128x72 3840x2160:96x54; do
  for e in "TSVPP_AREA_COLS=0" "TSVPP_AREA_COLS=1"; do
    echo -n "$c AREA f32 planar $e: "; one "$e" --custom $c:AREA:RGB24:PLANAR:1
  done
done
} 2>&1 | tee $O/call21.txt
