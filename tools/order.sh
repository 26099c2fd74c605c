#!/bin/bash
run() { echo -n "$1 $2 $3 :: "; env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['achieved'], r['roofline']['avg_launch_ms'], r['config']['parity'])"; }
for o in 0 1 2; do run "TSVPP_TILE_ORDER=$o"; run "TSVPP_TILE_ORDER=$o" --workload c2; run "TSVPP_TILE_ORDER=$o" --resize NEAREST; done
