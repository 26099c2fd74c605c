#!/bin/bash
# what the two-column tail launch costs (outputs 4 k + 2 columns wide: 854x480, 1366x768, 270x270): each width next to its 4 k neighbour, per resize type
# (ms_per_step covers both launches of a step; frac(step) = algorithmic bytes of a step / ms_per_step / 8 TB/s).  profiles/r04_tail_probe.txt also holds
# the columns of an experiment that is not in the tree (the tail on a side stream, "fork=1").
one() { python bench.py --custom $1 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r['roofline']; print('%9.0f fps  step %.4f ms  frac(step) %.3f  %s  %s' % (r['value'], r['ms_per_step'], rf['bytes_per_frame']*r['config']['frames_per_step']/(r['ms_per_step']*1e-3)/8e12, rf['kernel'][7:], r['config']['parity'][:9]))"; }
for c in "1920x1080:852x480" "1920x1080:854x480" "1920x1080:1364x768" "1920x1080:1366x768" "1920x1080:268x268" "1920x1080:270x270"; do
  for r in NEAREST BILINEAR BICUBIC AREA; do
    for o in "RGB24:PLANAR:1" "RGB24:MERGED:0"; do printf "%-20s %-9s %-15s " $c $r $o; one $c:$r:$o; done
  done
done
