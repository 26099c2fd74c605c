#!/bin/bash
# round-2 GPU call 2: new API tests, HBM mix floor with rotating sets, workgroup-shape sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 300 tools/bin/membench2 > $O/membench2.txt 2>&1; tail -14 $O/membench2.txt
one() { env $1 python bench.py --steps 30 --no-cpu-baseline --no-parity --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms']))"; }
sweep() { # $1 = label, rest = bench args
  for sh in "32,8" "64,4" "128,2" "64,2" "32,4"; do for rpt in 1 2 4; do echo -n "$1 SHAPE=$sh RPT=$rpt: "; one "TSVPP_SHAPE=$sh TSVPP_RPT=$rpt" "${@:2}"; done; done; }
{ sweep headline
  sweep area --resize AREA
  sweep bicubic --resize BICUBIC
  sweep u8planar --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
  sweep 960x540 --custom 1920x1080:960x540:BILINEAR:RGB24:PLANAR:1
  sweep 4k1080 --custom 3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:1
  sweep up --custom 1280x720:1920x1080:BILINEAR:RGB24:PLANAR:1
  echo -n "headline default: "; one "X=1"; } 2>&1 | tee $O/shape_sweep.txt
