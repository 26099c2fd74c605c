#!/bin/bash
# round-2 GPU call 17: window form of the float 2x2-tap thread tile (non-dyadic ratios <= 2), TSVPP_BILINEAR_WIN=0/1
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
TSVPP_BILINEAR_INT=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_bilinear_int.py tests/test_reference_crcs.py -m gpu -x -q > $O/pytest_gpu_int0.log 2>&1; echo "pytest INT=0 (float window everywhere) rc=$?"; tail -2 $O/pytest_gpu_int0.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in 1280x720:1920x1080:BILINEAR 1920x1080:1000x562:BILINEAR 1920x1080:1366x768:BILINEAR 1920x1080:1600x900:BILINEAR 1280x720:1920x1080:AREA 1920x1080:2560x1440:BILINEAR 1080x608:720x480:BILINEAR; do
  for o in RGB24:PLANAR:1 RGB24:MERGED:0 RGB24:PLANAR:0; do
    for e in "TSVPP_BILINEAR_WIN=0" "TSVPP_BILINEAR_WIN=1"; do
      echo -n "$c $o $e: "; one "$e" --custom $c:$o
    done
  done
done
echo -n "headline: "; one "X=1"
echo -n "c3: "; one "X=1" --workload c3
} 2>&1 | tee $O/call17.txt
