#!/bin/bash
# round 5, call 1: the row-segment BILINEAR kernel (vpp_bilinear_rows.hip) -- its tests, then same-box A/B against the byte-gather kernel it replaces
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
python -m pytest tests/test_gpu_bilinear_rows.py -x -q 2>&1 | tail -15 > gpurun_out/r05_rows_tests.txt
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  launch %7.1f us  frac %.3f  fpl %d  %s  %s' % (r['value'], ro['avg_launch_ms']*1e3, ro['frac'], r['config']['frames_per_launch'], ro['kernel'].split('::')[-1], 'ok' if r['config']['parity'].startswith('bit-exact') else 'PARITY-FAIL'))
except Exception as e:
    print('ERROR', e)"; }
{
echo "# same-box A/B: TSVPP_BILINEAR_ROWS=0 (byte gathers, rounds 1-4) vs default (row segments by LDS-DMA); bench.py --steps 50, frac = moved/PMC bytes where stamped else algorithmic"
for args in "--workload c3" "--workload c3 --batch 128" "--custom 1920x1080:300x300:BILINEAR:RGB24:PLANAR:1" "--custom 1920x1080:224x224:BILINEAR:RGB24:PLANAR:1" "--custom 3840x2160:640x360:BILINEAR:BGR24:PLANAR:1" "--custom 3840x2160:256x256:BILINEAR:RGB24:MERGED:0" "--custom 1920x1080:300x300:BILINEAR:RGB24:MERGED:0"; do
  for e in "TSVPP_BILINEAR_ROWS=0" "TSVPP_X=0" "TSVPP_RPT=2"; do
    printf "%-58s %-22s " "$args" "$e"; env $e python bench.py $args --steps 50 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line
  done
done
} > gpurun_out/r05_rows_ab.txt 2>&1
