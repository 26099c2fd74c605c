#!/bin/bash
# round-2 GPU call 16: uint8 merged outputs -- in-wave LDS exchange to 16-byte stores (TSVPP_U8_XCHG=0/1); row pairs per thread for uint8 2x2-tap
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in 1920x1080:1920x1080:NEAREST 1920x1080:1280x720:BILINEAR 1920x1080:1280x720:NEAREST 1920x1080:1280x720:BICUBIC 1920x1080:1280x720:AREA 1920x1080:960x540:AREA 3840x2160:1920x1080:BILINEAR 1920x1080:224x224:AREA 1920x1080:320x180:AREA 1280x720:1920x1080:BILINEAR 1920x1080:1000x562:BILINEAR; do
  for e in "TSVPP_U8_XCHG=0" "TSVPP_U8_XCHG=1"; do
    echo -n "$c u8 merged $e: "; one "$e" --custom $c:RGB24:MERGED:0
  done
done
for r in 2 3 4; do
  echo -n "u8 planar 1080p->720p BILINEAR RPT=$r: "; one "TSVPP_RPT=$r" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
  echo -n "u8 merged 1080p->720p BILINEAR RPT=$r: "; one "TSVPP_RPT=$r" --custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0
  echo -n "u8 planar 4k->1080p BILINEAR RPT=$r: "; one "TSVPP_RPT=$r" --custom 3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:0
done
} 2>&1 | tee $O/call16.txt
