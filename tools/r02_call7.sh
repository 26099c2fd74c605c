#!/bin/bash
# round-2 GPU call 7: compact LDS-DMA layout: parity, A/B against the power-of-two layout
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for e in "X=1" "TSVPP_DMA_POW2=1" "TSVPP_DMA=0"; do
  echo -n "headline $e: "; one "$e"
  echo -n "u8 planar $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
  echo -n "u8 merged $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0
  echo -n "bicubic $e: "; one "$e" --resize BICUBIC
  echo -n "area $e: "; one "$e" --resize AREA
  echo -n "area 960x540 $e: "; one "$e" --custom 1920x1080:960x540:AREA:RGB24:PLANAR:1
  echo -n "4k->1080p $e: "; one "$e" --custom 3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:1
  echo -n "720p->1080p bicubic $e: "; one "$e" --custom 1280x720:1920x1080:BICUBIC:RGB24:PLANAR:1
  echo -n "1080x608->480x360 area $e: "; one "$e" --custom 1080x608:480x360:AREA:RGB24:PLANAR:1
done
for rpt in 1 2 3 4; do for lds in 40 64; do echo -n "u8 planar RPT=$rpt LDS=$lds: "; one "TSVPP_RPT=$rpt TSVPP_LDS_KB=$lds" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0; done; done
for rpt in 1 2 3; do echo -n "bicubic RPT=$rpt: "; one "TSVPP_RPT=$rpt" --resize BICUBIC; done
} 2>&1 | tee $O/call7.txt
