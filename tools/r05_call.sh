#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r05_gpu_suite.txt
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  launch %7.1f us  frac %.3f  %s  %s' % (r['value'], ro['avg_launch_ms']*1e3, ro['frac'], ro['kernel'].split('::')[-1], r['config']['parity'][:9]))
except Exception as e:
    print('ERROR', e)"; }
{
echo "# uint8 outputs: TSVPP_NT=0 (plain stores -- what every 8- / 16-byte uint8 store REALLY was until round 5: the compiler merged the 'if (nt) non-temporal else plain' pair into one plain store) vs default (non-temporal, inline asm), same box; frac on algorithmic bytes"
for c in 3840x2160:1280x720:BICUBIC:BGR24:MERGED:0 3840x2160:1280x720:BICUBIC:BGR24:PLANAR:0 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0 1920x1080:1280x720:BILINEAR:NV12:MERGED:0 1920x1080:1280x720:BILINEAR:UYVY:MERGED:0 1920x1080:1280x720:BICUBIC:RGB24:MERGED:0 1920x1080:1280x720:BICUBIC:RGB24:PLANAR:0 1920x1080:1280x720:AREA:RGB24:MERGED:0 1920x1080:1280x720:NEAREST:RGB24:MERGED:0 3840x2160:1920x1080:BILINEAR:RGB24:MERGED:0 3840x2160:1920x1080:BICUBIC:RGB24:MERGED:0 960x540:1920x1080:BILINEAR:RGB24:MERGED:0 960x540:1920x1080:AREA:RGB24:MERGED:0 960x540:1920x1080:AREA:RGB24:PLANAR:0 1920x1080:1920x1080:NEAREST:RGB24:MERGED:0 1920x1080:1920x1080:NEAREST:Y800:MERGED:0 1920x1080:1920x1080:NEAREST:NV12:MERGED:0 1920x1080:1600x900:BILINEAR:RGB24:MERGED:0 1920x1080:960x544:AREA:BGR24:MERGED:0 1280x720:1920x1080:BICUBIC:RGB24:MERGED:0 1920x1080:640x360:AREA:RGB24:MERGED:0 3840x2160:1280x720:AREA:BGR24:MERGED:0; do
  for e in TSVPP_NT=0 TSVPP_X=0; do
    printf "%-48s %-12s " "$c" "$e"; env $e python bench.py --custom $c --steps 30 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line
  done
done
} > gpurun_out/r05_u8_merged_nt_ab.txt 2>&1
