#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  launch %7.1f us  frac %.3f  %s  %s' % (r['value'], ro['avg_launch_ms']*1e3, ro['frac'], ro['kernel'].split('::')[-1], r['config']['parity'][:9]))
except Exception as e:
    print('ERROR', e)"; }
{
echo "# tile order: 0 = one tile row per XCD (default); 3 / 4 / 5 = 2 / 4 / 8 consecutive tile rows per XCD, same box"
for args in "--workload c2" "--workload c1" "--custom 1280x720:1920x1080:BILINEAR:RGB24:PLANAR:1" "--custom 960x540:1920x1080:BILINEAR:RGB24:PLANAR:1" "--workload headline" "--resize NEAREST" "--resize BICUBIC" "--workload c5" "--custom 1920x1080:1920x1080:NEAREST:RGB24:MERGED:0" "--custom 960x540:1920x1080:AREA:RGB24:MERGED:0" "--custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0" "--custom 3840x2160:1920x1080:BILINEAR:RGB24:PLANAR:1"; do
  for e in TSVPP_X=0 TSVPP_TILE_ORDER=3 TSVPP_TILE_ORDER=4 TSVPP_TILE_ORDER=5; do
    printf "%-58s %-20s " "$args" "$e"; env $e python bench.py $args --steps 30 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line
  done
done
} > gpurun_out/r05_tile_groups_ab.txt 2>&1
