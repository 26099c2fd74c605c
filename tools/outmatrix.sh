#!/bin/bash
# Output-flavour matrix on the headline geometry (1080p -> 720p BILINEAR) and colour-only 1080p.
for c in "1920x1080:1280x720" "1920x1080:1920x1080"; do
for f in RGB24 Y800 NV12 UYVY YUV444 HSV; do for pl in PLANAR MERGED; do for n in 0 1; do
  if [ $f != RGB24 ] && [ $pl = PLANAR ]; then continue; fi
  if [ $f = HSV ] && [ $n = 0 ]; then continue; fi
  printf "%-20s %-7s %-7s norm=%s " $c $f $pl $n
  python bench.py --custom $c:BILINEAR:$f:$pl:$n --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(\"%9.0f fps %7.1f GB/s frac %.3f %s\" % (r[\"value\"], r[\"roofline\"][\"achieved\"], r[\"roofline\"][\"frac\"], r[\"config\"][\"parity\"]))"
done; done; done; done
