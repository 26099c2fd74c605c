#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_rep2.py tests/test_gpu_up2.py -x -q 2>&1 | tail -6 > gpurun_out/r05_rep2_tests.txt
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  launch %7.1f us  frac %.3f  fpl %d  %s  %s' % (r['value'], ro['avg_launch_ms']*1e3, ro['frac'], r['config']['frames_per_launch'], ro['kernel'].split('::')[-1], r['config']['parity'][:9]))
except Exception as e:
    print('ERROR', e)"; }
{
echo "# same-box A/B: TSVPP_R32=0 (LDS kernels) vs default (streaming 1 : 2 kernels: vpp_rep2_kernel for NEAREST / AREA, vpp_bilinear_up2_kernel for BILINEAR) vs TSVPP_R32=2 (fp32 flavours too)"
for c in 960x540:1920x1080:AREA:RGB24:MERGED:0 960x540:1920x1080:AREA:RGB24:PLANAR:0 960x540:1920x1080:NEAREST:BGR24:MERGED:0 960x540:1920x1080:AREA:NV12:MERGED:0 960x540:1920x1080:NEAREST:Y800:MERGED:0 1920x1080:3840x2160:AREA:RGB24:MERGED:0 960x540:1920x1080:AREA:RGB24:PLANAR:1 960x540:1920x1080:NEAREST:RGB24:MERGED:1 960x540:1920x1080:BILINEAR:RGB24:MERGED:0 960x540:1920x1080:BILINEAR:RGB24:PLANAR:1; do
  for e in TSVPP_R32=0 TSVPP_X=0 TSVPP_R32=2; do
    printf "%-48s %-12s " "$c" "$e"; env $e python bench.py --custom $c --steps 30 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line
  done
done
} > gpurun_out/r05_rep2_ab.txt 2>&1
