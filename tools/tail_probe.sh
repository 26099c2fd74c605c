#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# Outputs 4 k + 2 columns wide (854x480, 1366x768, 270x270) next to their 4 k neighbours, per resize type: shift=0 -- the two-column tail launch behind the
# main launch (rounds 1-3, TSVPP_TAIL_SHIFT=0); shift=1 -- the launch's last tile column shifted to the frame's right edge, no second launch (default)
# (ms_per_step covers every launch of a step; frac(step) = algorithmic bytes of a step / ms_per_step / 8 TB/s)
one() { env $1 python bench.py --custom $2 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r['roofline']; print('%9.0f fps  step %.4f ms  frac(step) %.3f  %s  %s' % (r['value'], r['ms_per_step'], rf['bytes_per_frame']*r['config']['frames_per_step']/(r['ms_per_step']*1e-3)/8e12, rf['kernel'][7:], r['config']['parity'][:9]))"; }
for c in "1920x1080:852x480" "1920x1080:854x480" "1920x1080:1364x768" "1920x1080:1366x768" "1920x1080:268x268" "1920x1080:270x270"; do
  for r in NEAREST BILINEAR BICUBIC AREA; do
    for o in "RGB24:PLANAR:1" "RGB24:MERGED:0"; do
      w=${c##*:}; w=${w%x*}
      if [ $((w % 4)) -eq 2 ]; then
        for k in 0 1; do printf "%-20s %-9s %-15s shift=%d " $c $r $o $k; one "TSVPP_TAIL_SHIFT=$k" $c:$r:$o; done
      else
        printf "%-20s %-9s %-15s         " $c $r $o; one "TSVPP_X=0" $c:$r:$o
      fi
    done
  done
done
