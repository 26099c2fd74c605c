#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# Output-flavour matrix on the headline geometry (1080p -> 720p BILINEAR), colour-only 1080p, and -- round 4 -- BICUBIC at 1080p -> 720p (the streaming
# kernel) and 720p -> 1080p (the column kernel).  Sparse samplers print the ROI-formula fraction (roi) next to the fraction on the bytes they move.
row() { # geometry resize fourcc planes norm
  printf "%-20s %-9s %-7s %-7s norm=%s " $1 $2 $3 $4 $5
  python bench.py --custom $1:$2:$3:$4:$5 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %7.1f GB/s frac %.3f%s %s %s\" % (r[\"value\"], rf.get(\"roi_achieved\", rf[\"achieved\"]), rf.get(\"roi_frac\", rf[\"frac\"]), (\" (moved bytes: %.3f)\" % rf[\"frac\"]) if \"roi_frac\" in rf else \"\", rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:60]))"
}
for c in "1920x1080:1280x720" "1920x1080:1920x1080"; do
for f in RGB24 Y800 NV12 UYVY YUV444 HSV; do for pl in PLANAR MERGED; do for n in 0 1; do
  if [ $f != RGB24 ] && [ $pl = PLANAR ]; then continue; fi
  if [ $f = HSV ] && [ $n = 0 ]; then continue; fi
  row $c BILINEAR $f $pl $n
done; done; done; done
for c in "1920x1080:1280x720" "1280x720:1920x1080" "3840x2160:1920x1080"; do
for f in RGB24 NV12 HSV; do for pl in PLANAR MERGED; do for n in 0 1; do
  if [ $f != RGB24 ] && [ $pl = PLANAR ]; then continue; fi
  if [ $f = HSV ] && [ $n = 0 ]; then continue; fi
  row $c BICUBIC $f $pl $n
done; done; done; done
