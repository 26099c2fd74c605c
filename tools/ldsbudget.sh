#!/bin/bash
# A/B of the per-workgroup LDS budget (TSVPP_LDS_KB) over requests whose footprint sits near it
one() { env TSVPP_LDS_KB=$1 python bench.py --custom $2 --steps 10 --warmup 2 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%8.0f' % r['value'], end='')"; }
for c in "1920x1080:1280x720:BILINEAR" "1920x1080:960x540:BILINEAR" "3840x2160:1920x1080:BILINEAR" "1920x1080:416x416:BILINEAR" "1920x1080:1280x720:BICUBIC" "1920x1080:960x540:BICUBIC" "1280x720:1920x1080:BICUBIC" "1920x1080:960x540:AREA" "1920x1080:640x360:AREA" "1920x1080:640x640:AREA" "1080x608:480x360:AREA"; do
  printf "%-32s" $c; for kb in 24 32 40 48 64; do printf " | %s KiB:" $kb; one $kb $c:BGR24:PLANAR:1; done; echo
done
