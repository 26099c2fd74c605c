#!/bin/bash
# round-2 GPU call 11: persistent (double-buffered) 2x2-tap kernel with the window tile, uint8 outputs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bilinear_int.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_gpu_default.log 2>&1; echo "pytest default rc=$?"; tail -1 $O/pytest_gpu_default.log
TSVPP_PERSIST=4 timeout 600 python -m pytest tests/test_gpu_bilinear_int.py tests/test_gpu_fuzz.py tests/test_reference_crcs.py -m gpu -x -q > $O/pytest_gpu_persist.log 2>&1; echo "pytest persist rc=$?"; tail -1 $O/pytest_gpu_persist.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for e in "TSVPP_PERSIST=0" "TSVPP_PERSIST=2" "TSVPP_PERSIST=3" "TSVPP_PERSIST=4" "TSVPP_PERSIST=6" "TSVPP_PERSIST=8" "TSVPP_PERSIST=4 TSVPP_SHAPE=64,4" "TSVPP_PERSIST=6 TSVPP_SHAPE=64,4" "TSVPP_PERSIST=3 TSVPP_SHAPE=16,16" "TSVPP_PERSIST=4 TSVPP_SHAPE=16,16"; do
  echo -n "u8 planar 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
  echo -n "u8 merged 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0
  echo -n "headline $e: "; one "$e"
done
} 2>&1 | tee $O/call11.txt
