#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_bicubic_cols.py tests/test_gpu_fuzz.py tests/test_gpu_tail_shift.py tests/test_gpu_bunny.py -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_bc_wide_tests.txt
TSVPP_BICUBIC_COLS=2 python -m pytest tests/test_gpu_bicubic_cols.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_edges.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/r05_bc_wide_tests.txt
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  launch %7.1f us  frac %.3f  %s  %s' % (r['value'], ro['avg_launch_ms']*1e3, ro.get('roi_frac', ro['frac']), ro['kernel'].split('::')[-1], r['config']['parity'][:9]))
except Exception as e:
    print('ERROR', e)"; }
{
echo "# vpp_bicubic_cols_kernel, dense rows at 17 .. 32 chunks per row segment (horizontal ratios 3.7 .. 7.4): TSVPP_BICUBIC_DMA=3 (per-lane loads, rounds 3-4) vs default (LDS-DMA, two instructions per group of four rows); frac on ROI bytes"
for c in 1920x1080:300x300:BICUBIC:RGB24:PLANAR:1 1920x1080:300x300:BICUBIC:RGB24:MERGED:0 3840x2160:640x360:BICUBIC:RGB24:PLANAR:1 1920x1080:416x416:BICUBIC:RGB24:PLANAR:1 1920x1080:384x288:BICUBIC:BGR24:PLANAR:1 3840x2160:854x480:BICUBIC:RGB24:PLANAR:1 1920x1080:512x288:BICUBIC:RGB24:PLANAR:1; do
  for e in TSVPP_BICUBIC_DMA=3 TSVPP_X=0 TSVPP_BICUBIC_ROWS=16; do
    printf "%-48s %-22s " "$c" "$e"; env $e python bench.py --custom $c --steps 30 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line
  done
done
} > gpurun_out/r05_bicubic_wide_ab.txt 2>&1
