#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# DIAGNOSTIC: what launches of independent batches on 1 / 2 / 3 consumer streams do to throughput (bench.py --streams N: steps issued round-robin, one
# launch's drain overlaps the next one's fill).  The bench line itself times ONE stream; this is the caller-side lever of DESIGN.md section 8 "Launch size".
one() { python bench.py $1 --streams $2 --steps 24 --warmup 3 --no-cpu-baseline --no-others 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r['roofline']; print('  streams=$2 %9.0f fps  step %.4f ms  %s' % (r['value'], r['ms_per_step'], r['config']['parity'][:9]), end='')"; }
for w in "--workload headline" "--workload c2" "--workload c3" "--workload c4" "--workload c5" "--custom 1920x1080:1920x1080:BILINEAR:Y800:MERGED:0" "--custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0" "--custom 1920x1080:300x300:AREA:RGB24:PLANAR:1"; do
  printf "%-58s" "$w"; for n in 1 2 3; do one "$w" $n; done; echo
done
