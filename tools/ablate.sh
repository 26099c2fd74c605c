#!/bin/bash
run() { echo -n "$1 :: "; env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['achieved'], r['roofline']['avg_launch_ms'])"; }
for s in "TSVPP_ABLATE=0" "TSVPP_ABLATE=1" "TSVPP_ABLATE=2" "TSVPP_ABLATE=3" "TSVPP_ABLATE=4" "TSVPP_ABLATE=5" "TSVPP_ABLATE=6" "TSVPP_ABLATE=7"; do run "$s" "$@"; done
