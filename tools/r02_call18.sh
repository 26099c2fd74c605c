#!/bin/bash
# round-2 GPU call 18: divisor table in the global-memory column-per-lane AREA kernel (TSVPP_AREA_DIVTAB=0/1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in 1920x1080:300x300 1920x1080:416x416 3840x2160:608x342 1280x720:224x224 1920x1080:384x216 1920x1080:224x224; do
  for e in "TSVPP_AREA_DIVTAB=0" "TSVPP_AREA_DIVTAB=1"; do
    echo -n "$c AREA f32 planar $e: "; one "$e" --custom $c:AREA:RGB24:PLANAR:1
    echo -n "$c AREA u8 merged $e: "; one "$e" --custom $c:AREA:RGB24:MERGED:0
  done
done
} 2>&1 | tee $O/call18.txt
