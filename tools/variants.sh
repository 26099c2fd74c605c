#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# same-box A/B of library BUILDS (variants/<name>.so, built here with different -D flags) over ad-hoc workloads:
#   tools/variants.sh "SRC:DST:RESIZE:FOURCC:PLANES:NORM ..." "ENV" A B C ...
# each variant is copied over tensor-stream_amd/lib/libtsvpp.so on the GPU box (a scratch copy of the tree) before its runs
one() { env $1 python bench.py --custom "$2" --steps 20 --warmup 3 --no-cpu-baseline $BENCH_FLAGS 2>&1 | tail -1 | python -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); print('%8.0f fps %.3f %s' % (r['value'], r['roofline']['frac'], 'ok' if r['config']['parity'].startswith('bit-exact') else r['config']['parity'][:8]), end='')
except Exception as e:
    print('ERROR', end='')"; }
cp tensor-stream_amd/lib/libtsvpp.so /tmp/libtsvpp.keep
for c in $1; do
  printf "%-44s" "$c"
  for v in "${@:3}"; do cp variants/$v.so tensor-stream_amd/lib/libtsvpp.so; echo -n " | $v: "; one "$2" "$c"; done; echo
done
cp /tmp/libtsvpp.keep tensor-stream_amd/lib/libtsvpp.so
