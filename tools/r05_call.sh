#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  launch %7.1f us  frac %.3f  %s  %s' % (r['value'], ro['avg_launch_ms']*1e3, ro['frac'], ro['kernel'].split('::')[-1], r['config']['parity'][:9]))
except Exception as e:
    print('ERROR', e)"; }
{
echo "# uint8 outputs, store policy of the 8- / 16-byte stores: default (non-temporal) vs TSVPP_NT=2 (sc1), same box"
for c in 3840x2160:1280x720:BICUBIC:BGR24:MERGED:0 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0 1920x1080:1280x720:BICUBIC:RGB24:MERGED:0 960x540:1920x1080:BILINEAR:RGB24:MERGED:0 960x540:1920x1080:AREA:RGB24:PLANAR:0 1920x1080:1920x1080:NEAREST:RGB24:MERGED:0 1920x1080:1920x1080:NEAREST:RGB24:PLANAR:0 1920x1080:1920x1080:NEAREST:Y800:MERGED:0 1920x1080:640x360:AREA:RGB24:MERGED:0 3840x2160:1920x1080:BILINEAR:RGB24:MERGED:0; do
  for e in TSVPP_X=0 TSVPP_NT=2; do
    printf "%-48s %-12s " "$c" "$e"; env $e python bench.py --custom $c --steps 30 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line
  done
done
} > gpurun_out/r05_u8_sc1_ab.txt 2>&1
