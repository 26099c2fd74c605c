#!/bin/bash
# round-2 GPU call 5: integer bilinear tile + staging diet: parity, A/B on fp32 and uint8 outputs, output matrix
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:20]))"; }
{ for e in "X=1" "TSVPP_BILINEAR_INT=0"; do
  echo -n "headline $e: "; one "$e"
  echo -n "u8 planar 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
  echo -n "u8 merged 1080p->720p $e: "; one "$e" --custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0
  echo -n "4k->1080p f32 $e: "; one "$e" --custom 3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:1
  echo -n "4k->1080p u8 $e: "; one "$e" --custom 3840x2160:1920x1080:BILINEAR:RGB24:MERGED:0
  echo -n "540p->1080p u8 $e: "; one "$e" --custom 960x540:1920x1080:BILINEAR:RGB24:MERGED:0
done
for sh in "32,8" "64,4"; do for rpt in 1 2 4; do echo -n "u8 planar int SHAPE=$sh RPT=$rpt: "; one "TSVPP_SHAPE=$sh TSVPP_RPT=$rpt" --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0; done; done
for sh in "32,8" "64,4"; do for rpt in 1 2; do echo -n "headline int SHAPE=$sh RPT=$rpt: "; one "TSVPP_SHAPE=$sh TSVPP_RPT=$rpt"; done; done
echo -n "bicubic: "; one "X=1" --resize BICUBIC
echo -n "area: "; one "X=1" --resize AREA
echo -n "c2: "; one "X=1" --workload c2
echo -n "c3: "; one "X=1" --workload c3
echo -n "c4: "; one "X=1" --workload c4
} 2>&1 | tee $O/call5.txt
