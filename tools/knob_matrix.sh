#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# Runs the whole GPU parity suite once per tuning-knob setting: every alternative code path (gather fallbacks, compact vs LDS-DMA staging, the
# BICUBIC kernels (integer / wave-per-tile with every staging mode and tile height / gathers), generic vs table vs streaming AREA kernels, thread-tile
# heights, direct-kernel thresholds, workgroup shapes, tile orders, store policies, geometry tables, the streaming 3:2 / 2:1 kernel incl. its
# single-pass UYVY / YUV444) must stay bit-exact.  KNOBS="..." (newline-separated) overrides the list, TESTS="files" the suite (default: tests).
DEFAULT="TSVPP_FORCE_GATHER=1
TSVPP_BILINEAR_ROWS=0
TSVPP_BILINEAR_ROWS=2
TSVPP_BILINEAR_ROWS_WAVES=1
TSVPP_POINT_RN=0
TSVPP_POINT_RN=2
TSVPP_BICUBIC_U8X=0
TSVPP_BICUBIC_U8X=2
TSVPP_BICUBIC_COLS=2 TSVPP_BICUBIC_U8X=2
TSVPP_DMA=0
TSVPP_BILINEAR_INT=0
TSVPP_BILINEAR_INT=2
TSVPP_BICUBIC_INT=0
TSVPP_BICUBIC_INT=2
TSVPP_BICUBIC_COLS=2
TSVPP_BICUBIC_COLS=0
TSVPP_BICUBIC_COLS=2 TSVPP_BICUBIC_DMA=0
TSVPP_BICUBIC_COLS=2 TSVPP_BICUBIC_DMA=2
TSVPP_BICUBIC_COLS=2 TSVPP_BICUBIC_DMA=3
TSVPP_BICUBIC_COLS=2 TSVPP_BICUBIC_ROWS=8
TSVPP_BICUBIC_COLS=2 TSVPP_BICUBIC_ROWS=32
TSVPP_RPT=1
TSVPP_RPT=2
TSVPP_RPT=3
TSVPP_AREA_DIRECT_MIN=2
TSVPP_AREA_DIRECT_MIN=100
TSVPP_SHAPE=16,4
TSVPP_SHAPE=32,8
TSVPP_SHAPE=64,4
TSVPP_SHAPE=128,2
TSVPP_AREA_BOX=0
TSVPP_AREA_COLS=0
TSVPP_AREA_COLS=2
TSVPP_AREA_COLS_ROWS=8
TSVPP_AREA_COLS_ROWS=32
TSVPP_TAIL_SHIFT=0
TSVPP_AREA_DIVTAB=0
TSVPP_AREA_STREAM=0
TSVPP_AREA_STREAM=2
TSVPP_TILE_ORDER=1
TSVPP_TILE_ORDER=2
TSVPP_TILE_ORDER=3
TSVPP_TILE_ORDER=5
TSVPP_NT=0
TSVPP_NT=2
TSVPP_R32=0
TSVPP_R32=2
TSVPP_GEO=0
TSVPP_GEO=2"
echo "${KNOBS:-$DEFAULT}" | while read -r e; do
  [ -z "$e" ] && continue
  printf "%-55s" "$e"; timeout 200 env $e python -m pytest ${TESTS:-tests} -m gpu -q --timeout 100 -p no:cacheprovider 2>&1 | tail -1
done
