#!/bin/bash
# A/B of the BICUBIC kernels: table kernel (SEP=0) vs separable kernel with 2-row / 4-row thread tiles
one() { env $1 python bench.py --custom $2:BICUBIC:${3:-BGR24}:PLANAR:${4:-1} --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps %6.0f GB/s %s' % (r['value'], r['roofline']['achieved'], r['config']['parity'][:9]), end='')"; }
for g in "1920x1080:1280x720" "1280x720:1920x1080" "1920x1080:640x640" "3840x2160:1920x1080" "1080x608:480x360" "1920x1080:300x300"; do
  printf "%-22s" $g; for e in "TSVPP_BICUBIC_SEP=0" "TSVPP_BICUBIC_SEP=1 TSVPP_RPT=1" "TSVPP_BICUBIC_SEP=1 TSVPP_RPT=2"; do echo -n " | "; one "$e" $g; done; echo
done
printf "%-22s" "u8 1080p->720p"; for e in "TSVPP_BICUBIC_SEP=0" "TSVPP_BICUBIC_SEP=1 TSVPP_RPT=1" "TSVPP_BICUBIC_SEP=1 TSVPP_RPT=2"; do echo -n " | "; one "$e" 1920x1080:1280x720 BGR24 0; done; echo
