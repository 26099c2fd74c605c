#!/bin/bash
# round-2 GPU call 9: window form of the integer 2x2-tap thread tile (aligned 12-byte row reads) vs the byte form
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for e in "TSVPP_BILINEAR_INT=1" "TSVPP_BILINEAR_INT=2" "TSVPP_BILINEAR_INT=0"; do
  echo -n "headline $e: "; one "$e"
  echo -n "u8 planar 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
  echo -n "u8 merged 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0
  echo -n "f32 merged 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:1
  echo -n "y800 u8 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:Y800:MERGED:0
  echo -n "4k->1080p u8 planar $e: "; one "$e" --custom 3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:0
  echo -n "540p->1080p u8 planar $e: "; one "$e" --custom 960x540:1920x1080:BILINEAR:RGB24:PLANAR:0
  echo -n "720p->1080p AREA-up u8 merged $e: "; one "$e" --custom 1280x720:1920x1080:AREA:RGB24:MERGED:0
done
for s in 32,8 64,4 16,16; do for r in 1 2 4; do
  echo -n "u8 planar window SHAPE=$s RPT=$r: "; one "TSVPP_SHAPE=$s TSVPP_RPT=$r" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
done; done
echo -n "c2: "; one "X=1" --workload c2
} 2>&1 | tee $O/call9.txt
tools/profile.sh u8win --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0 > $O/prof_u8win.log 2>&1
python tools/pmc_summary.py $(find $O/prof_u8win -name "*counter_collection.csv") 2>&1 | tee $O/u8win_pmc.txt
