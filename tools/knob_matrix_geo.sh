#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# The GPU parity suite under the knobs added after profiles/r02_knob_matrix.txt (geometry tables, row-pair format kernels, streaming
# 3:2 / 2:1 kernel)
for e in "TSVPP_GEO=0" "TSVPP_GEO=2" "TSVPP_GEO=2 TSVPP_SHAPE=64,4" "TSVPP_GEO=2 TSVPP_SHAPE=16,4" "TSVPP_GEO=2 TSVPP_RPT=3" "TSVPP_GEO=2 TSVPP_DMA=0" \
         "TSVPP_R32=0" "TSVPP_R32=2" "TSVPP_R32=0 TSVPP_GEO=0" "TSVPP_R32=2 TSVPP_SHAPE=16,4" "TSVPP_R32=2 TSVPP_SHAPE=128,2" "TSVPP_R32=2 TSVPP_TILE_ORDER=2"; do
  [ -n "$ONLY_R32" ] && [[ "$e" != *R32* ]] && continue
  printf "%-45s" "$e"; env $e timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
