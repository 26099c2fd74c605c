#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# same-box A/B of environment settings over ad-hoc workloads:
#   tools/abc.sh "SRC:DST:RESIZE:FOURCC:PLANES:NORM ..." "ENV_A" "ENV_B" ...     (an empty ENV = defaults)
# prints one row per workload, one column per setting: frames/s, fraction of the 8 TB/s roofline, parity
one() { env $1 python bench.py --custom "$2" --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); print('%8.0f fps %.3f %s' % (r['value'], r['roofline']['frac'], 'ok' if r['config']['parity'].startswith('bit-exact') else 'PARITY-FAIL'), end='')
except Exception as e:
    print('ERROR', end='')"; }
for c in $1; do
  printf "%-44s" "$c"; for e in "${@:2}"; do echo -n " | ${e:-default}: "; one "$e" "$c"; done; echo
done
