#!/bin/bash
# round-2 GPU call 3: integer BICUBIC kernel parity + speed, new default shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:20]))"; }
{ for e in "X=1" "TSVPP_BICUBIC_INT=0"; do
  echo -n "bicubic 1080p->720p $e: "; one "$e" --resize BICUBIC
  echo -n "bicubic 1080p->960x540 $e: "; one "$e" --custom 1920x1080:960x540:BICUBIC:RGB24:PLANAR:1
  echo -n "bicubic 4k->1080p $e: "; one "$e" --custom 3840x2160:1920x1080:BICUBIC:RGB24:PLANAR:1
  echo -n "bicubic 540p->1080p $e: "; one "$e" --custom 960x540:1920x1080:BICUBIC:RGB24:PLANAR:1
  echo -n "bicubic 1080p->720p u8 merged $e: "; one "$e" --custom 1920x1080:1280x720:BICUBIC:BGR24:MERGED:0
done
for sh in "32,8" "64,4" "32,4"; do for rpt in 1 2; do for lds in 40 64; do echo -n "bicubic int SHAPE=$sh RPT=$rpt LDS=$lds: "; one "TSVPP_SHAPE=$sh TSVPP_RPT=$rpt TSVPP_LDS_KB=$lds" --resize BICUBIC; done; done; done
echo -n "headline: "; one "X=1"
echo -n "up 720p->1080p: "; one "X=1" --custom 1280x720:1920x1080:BILINEAR:RGB24:PLANAR:1
echo -n "4k->1080p: "; one "X=1" --custom 3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:1
} 2>&1 | tee $O/call3.txt
