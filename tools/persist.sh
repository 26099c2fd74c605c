#!/bin/bash
run() { echo -n "$1 :: "; env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['achieved'], r['roofline']['avg_launch_ms'], r['config']['parity'])"; }
for s in "TSVPP_PERSIST=0" "TSVPP_PERSIST=4" "TSVPP_PERSIST=6" "TSVPP_PERSIST=7" "TSVPP_PERSIST=0" "TSVPP_PERSIST=8" "TSVPP_PERSIST=5" "TSVPP_PERSIST=3"; do run "$s" "$@"; done
