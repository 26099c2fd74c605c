#!/bin/bash
# A/B sweep of the tuning knobs on the GPU box (one box, interleaved): tools/sweep.sh [bench args]
run() { echo -n "$1 :: "; env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['achieved'], r['roofline']['avg_launch_ms'])"; }
for s in "X=0" "TSVPP_NT=0" "TSVPP_NT=2" "TSVPP_TILE_ORDER=1" "TSVPP_TILE_ORDER=2" "TSVPP_SHAPE=64,4" "TSVPP_SHAPE=16,16" "TSVPP_DMA=0" "TSVPP_RPT=1" "X=1"; do run "$s" "$@"; done
