#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_bicubic_cols.py tests/test_gpu_formats.py tests/test_gpu_fuzz.py tests/test_reference_crcs.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_bc_tests.txt
TSVPP_BICUBIC_COLS=2 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 >> gpurun_out/r05_bc_tests.txt
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  launch %7.1f us  frac %.3f  %s  %s' % (r['value'], ro['avg_launch_ms']*1e3, ro['frac'], ro['kernel'].split('::')[-1], r['config']['parity'][:9]))
except Exception as e:
    print('ERROR', e)"; }
{
echo "# same-box A/B: TSVPP_BICUBIC_U8X=0 (4 x 2 thread tiles, rounds 3-4) vs default (uint8 outputs of vpp_bicubic_cols_kernel through the 8 x 4 output side of the streaming kernels)"
for c in 1280x720:1920x1080:BICUBIC:RGB24:MERGED:0 1280x720:1920x1080:BICUBIC:RGB24:PLANAR:0 1280x720:1920x1080:BICUBIC:NV12:MERGED:0 1280x720:1920x1080:BICUBIC:Y800:MERGED:0 1920x1080:1440x816:BICUBIC:RGB24:MERGED:0 1920x1080:640x640:BICUBIC:BGR24:PLANAR:0 1080x608:480x360:BICUBIC:RGB24:MERGED:0 1920x1080:224x224:BICUBIC:RGB24:MERGED:0 1280x720:1920x1080:BICUBIC:UYVY:MERGED:0; do
  for e in TSVPP_BICUBIC_U8X=0 TSVPP_X=0 TSVPP_BICUBIC_ROWS=32 TSVPP_BICUBIC_ROWS=16; do
    printf "%-48s %-22s " "$c" "$e"; env $e python bench.py --custom $c --steps 30 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line
  done
done
} > gpurun_out/r05_bicubic_cols_u8_ab.txt 2>&1
