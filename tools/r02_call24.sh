#!/bin/bash
# round-2 GPU call 24: row-pair UYVY / YUV444 kernels (TSVPP_FMT_RP=0/1): parity suite, then same-box A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
one() { env $1 timeout 120 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); r=json.loads(l); print(l[:300]) if 'value' not in r else print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in "1920x1080:1920x1080" "1920x1080:1280x720"; do for f in UYVY YUV444; do for n in 0 1; do
  for e in "TSVPP_FMT_RP=0" "TSVPP_FMT_RP=1"; do
    echo -n "$c $f norm=$n $e: "; one "$e" --custom $c:BILINEAR:$f:MERGED:$n
  done
done; done; done
} 2>&1 | tee $O/call24.txt
