#!/bin/bash
# round-2 GPU call 29: streaming 3:2 BILINEAR kernel for uint8 outputs (TSVPP_R32=0/1): parity suite, then same-box A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
one() { env $1 timeout 120 python bench.py --steps 30 --repeats 7 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; l=sys.stdin.read(); r=json.loads(l); print(l[:300]) if 'value' not in r else print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for c in "1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0" "3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:0"; do
  for e in "TSVPP_SHAPE=64,4" "TSVPP_SHAPE=64,2" "TSVPP_SHAPE=64,1" "TSVPP_SHAPE=128,2" "TSVPP_SHAPE=32,8" "TSVPP_TILE_ORDER=1" "TSVPP_TILE_ORDER=2" "TSVPP_NT=0"; do echo -n "$c $e: "; one "$e" --custom $c; done
done
} 2>&1 | tee $O/call29d.txt
