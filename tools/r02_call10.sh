#!/bin/bash
# round-2 GPU call 10: round-toward-zero saturating pack (no v_trunc), address-as-shift v_alignbyte, incremental LDS-DMA slot walk
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ e="X=1"
  echo -n "headline $e: "; one "$e"
  echo -n "u8 planar 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
  echo -n "u8 merged 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0
  echo -n "f32 merged 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:1
  echo -n "y800 u8 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:Y800:MERGED:0
  echo -n "4k->1080p u8 planar $e: "; one "$e" --custom 3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:0
  echo -n "540p->1080p u8 planar $e: "; one "$e" --custom 960x540:1920x1080:BILINEAR:RGB24:PLANAR:0
  echo -n "720p->1080p AREA-up u8 merged $e: "; one "$e" --custom 1280x720:1920x1080:AREA:RGB24:MERGED:0
  echo -n "u8 planar 1080p->720p NEAREST: "; one "$e" --custom 1920x1080:1280x720:NEAREST:RGB24:PLANAR:0
  echo -n "u8 planar 1080p->720p BICUBIC: "; one "$e" --custom 1920x1080:1280x720:BICUBIC:RGB24:PLANAR:0
  echo -n "u8 planar 1080p->720p AREA: "; one "$e" --custom 1920x1080:1280x720:AREA:RGB24:PLANAR:0
  echo -n "u8 merged 1080p->1080p (colour only): "; one "$e" --custom 1920x1080:1920x1080:NEAREST:RGB24:MERGED:0
  echo -n "u8 planar 1080p->1080p (colour only): "; one "$e" --custom 1920x1080:1920x1080:NEAREST:RGB24:PLANAR:0
  for s in 32,8 64,4; do for r in 1 2 4; do
    echo -n "u8 planar window SHAPE=$s RPT=$r: "; one "TSVPP_SHAPE=$s TSVPP_RPT=$r" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
  done; done
  for w in c2 c3 c5; do echo -n "$w: "; one "X=1" --workload $w; done
  echo -n "bicubic: "; one "X=1" --resize BICUBIC
  echo -n "area: "; one "X=1" --resize AREA
} 2>&1 | tee $O/call10.txt
tools/profile.sh u8win2 --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0 > $O/prof_u8win2.log 2>&1
python tools/pmc_summary.py $(find $O/prof_u8win2 -name "*counter_collection.csv") 2>&1 | tee $O/u8win2_pmc.txt
