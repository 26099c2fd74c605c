#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# same-box A/B over BICUBIC workloads and output flavours: tools/bicubic_ab.sh "TSVPP_BICUBIC_INT=2" "TSVPP_BICUBIC_INT=1"
one() { env $1 python bench.py --custom $2 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps %.3f %s' % (r['value'], r['roofline']['frac'], 'ok' if r['config']['parity'].startswith('bit-exact') else r['config']['parity'][:20]), end='')"; }
for c in "1920x1080:1280x720:BICUBIC:BGR24:PLANAR:1" "1920x1080:1280x720:BICUBIC:RGB24:MERGED:0" "1920x1080:1280x720:BICUBIC:RGB24:PLANAR:0" "1920x1080:1280x720:BICUBIC:RGB24:MERGED:1" \
         "1920x1080:1280x720:BICUBIC:NV12:MERGED:0" "1920x1080:1280x720:BICUBIC:HSV:MERGED:1" "3840x2160:1920x1080:BICUBIC:RGB24:PLANAR:1" "3840x2160:1920x1080:BICUBIC:RGB24:MERGED:0" \
         "1920x1080:960x540:BICUBIC:RGB24:PLANAR:1" "1920x1080:960x540:BICUBIC:RGB24:MERGED:0" ${EXTRA_CASES}; do
  printf "%-46s" "$c"; for e in "$@"; do echo -n " | $e: "; one "$e" $c; done; echo
done
