#!/bin/bash
# round 5: where does C3's launch spend its time?  --alias 1 = reads from cache, 2 = writes stay in cache, 3 = both
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
python -m pytest tests/test_gpu_bilinear_rows.py -x -q 2>&1 | tail -5 > gpurun_out/r05_rows_tests.txt
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  launch %7.1f us  frac %.3f  fpl %d  %s  %s' % (r['value'], ro['avg_launch_ms']*1e3, ro['frac'], r['config']['frames_per_launch'], ro['kernel'].split('::')[-1], r['config']['parity'][:9]))
except Exception as e:
    print('ERROR', e)"; }
{
for args in "--workload c3" "--workload c3 --alias 1" "--workload c3 --alias 2" "--workload c3 --alias 3" "--workload c3 --batch 128" "--custom 1920x1080:300x300:BILINEAR:RGB24:PLANAR:1" "--custom 1920x1080:224x224:BILINEAR:RGB24:PLANAR:1" "--custom 3840x2160:640x360:BILINEAR:BGR24:PLANAR:1" "--custom 3840x2160:256x256:BILINEAR:RGB24:MERGED:0" "--custom 1920x1080:300x300:BILINEAR:RGB24:MERGED:0"; do
  for e in ${ENVS:-TSVPP_X=0}; do
    printf "%-58s %-22s " "$args" "$e"; env $e python bench.py $args --steps 50 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line
  done
done
} > gpurun_out/r05_c3_diag.txt 2>&1
