#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  launch %7.1f us  frac %.3f  fpl %d  %s  %s' % (r['value'], ro['avg_launch_ms']*1e3, ro['frac'], r['config']['frames_per_launch'], ro['kernel'].split('::')[-1], r['config']['parity'][:9]))
except Exception as e:
    print('ERROR', e)"; }
{
for args in "--workload c3" "--workload c3 --batch 128" "--custom 1920x1080:224x224:BILINEAR:RGB24:PLANAR:1"  "--custom 3840x2160:640x360:BILINEAR:BGR24:PLANAR:1"; do
  for e in TSVPP_X=0 TSVPP_BILINEAR_ROWS_WAVES=1 TSVPP_BILINEAR_ROWS_WAVES=2 TSVPP_BILINEAR_ROWS_WAVES=4 TSVPP_TILE_ORDER=1 TSVPP_NT=0 TSVPP_NT=2; do
    printf "%-58s %-28s " "$args" "$e"; env $e python bench.py $args --steps 50 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line
  done
done
} > gpurun_out/r05_rows_waves_ab.txt 2>&1
