#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
line() { python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); ro=r['roofline']; print('%9.0f fps  step %7.1f us  fpl %d %s' % (r['value'], r['ms_per_step']*1e3, r['config']['frames_per_launch'], r['config']['parity'][:9]))
except Exception as e:
    print('ERROR', e)"; }
{
echo "# DIAGNOSTIC (bench.py --streams N: the steps of a timed region issued round-robin on N HIP streams), final sources of round 5"
for args in "--workload c4" "--workload c4 --streams 2" "--workload c3" "--workload c3 --streams 2" "--workload c1 --streams 2" "--workload headline --streams 2"; do printf "%-40s " "$args"; python bench.py $args --steps 30 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | tail -1 | line; done
} > gpurun_out/r05_streams_probe.txt 2>&1
