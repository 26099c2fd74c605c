#!/bin/bash
# The GPU parity suite under the knobs added after profiles/r02_knob_matrix.txt (geometry tables, row-pair format kernels)
for e in "TSVPP_GEO=0" "TSVPP_GEO=2" "TSVPP_GEO=2 TSVPP_SHAPE=64,4" "TSVPP_GEO=2 TSVPP_SHAPE=16,4" "TSVPP_GEO=2 TSVPP_RPT=3" "TSVPP_GEO=2 TSVPP_BILINEAR_WIN=2" "TSVPP_GEO=2 TSVPP_DMA=0" "TSVPP_FMT_RP=0"; do
  printf "%-45s" "$e"; env $e timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
done
