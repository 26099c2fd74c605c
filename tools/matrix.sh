#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# perf matrix over common conversions (looking for cliffs): tools/matrix.sh
for c in "1920x1080:224x224" "1920x1080:640x640" "1920x1080:960x540" "1920x1080:1920x1080" "1280x720:1920x1080" "3840x2160:1920x1080" "1080x608:480x360" "1920x1080:300x300" "1920x1080:1280x720" "1920x1080:1440x810"; do
  for r in NEAREST BILINEAR BICUBIC AREA; do
    [ "$c" = "1920x1080:1920x1080" ] && [ "$r" != "NEAREST" ] && continue
    printf "%-22s %-9s" $c $r
    python bench.py --custom $c:$r:RGB24:PLANAR:1 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r['roofline']; print('%9.0f fps %7.1f GB/s  frac %.3f%s  %s  %s' % (r['value'], rf.get('roi_achieved', rf['achieved']), rf.get('roi_frac', rf['frac']), (' (moved bytes: %.3f)' % rf['frac']) if 'roi_frac' in rf else '', rf['kernel'][7:], r['config']['parity'][:60]))"
  done
done
