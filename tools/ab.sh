#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# same-box A/B of one environment knob over the bench workloads: tools/ab.sh "TSVPP_DMA=0" "TSVPP_DMA=1"
one() { env $1 python bench.py --steps 30 --no-cpu-baseline --no-parity "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps %6.0f GB/s' % (r['value'], r['roofline']['achieved']), end='')"; }
for args in "--workload headline" "--resize NEAREST" "--resize AREA" "--resize BICUBIC" "--workload c2" "--workload c3" "--workload c4" "--workload c5"; do
  printf "%-20s" "$args"; for e in "$@"; do echo -n " | $e: "; one "$e" $args; done; echo
done
