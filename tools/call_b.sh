#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
TSVPP_AREA_BOX=2 timeout 120 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
one() { env $1 timeout 60 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); r=json.loads(l); print(l[:300]) if 'value' not in r else print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in "1920x1080:960x540:AREA:RGB24:PLANAR:1" "1920x1080:640x360:AREA:RGB24:PLANAR:1" "1920x1080:640x360:AREA:RGB24:PLANAR:0" "3840x2160:1280x720:AREA:RGB24:PLANAR:1" "3840x2160:1280x720:AREA:RGB24:MERGED:0"; do
  for e in "TSVPP_AREA_BOX=1" "TSVPP_AREA_BOX=2"; do echo -n "$c $e: "; one "$e" --custom $c; done
done; } 2>&1 | tee $O/call_b.txt
