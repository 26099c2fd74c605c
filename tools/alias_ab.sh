one() { python bench.py --custom "$1" --steps 20 --warmup 3 --no-cpu-baseline --no-parity --alias $2 2>&1 | tail -1 | python -c "
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
import sys,json
r=json.loads(sys.stdin.read()); print('%8.0f fps %.3f' % (r['value'], r['roofline']['frac']), end='')"; }
for c in 1920x1080:224x224:AREA:RGB24:PLANAR:1 3840x2160:608x342:AREA:RGB24:PLANAR:1 1920x1080:224x224:BICUBIC:RGB24:PLANAR:1 1280x720:1920x1080:BICUBIC:RGB24:PLANAR:1 1920x1080:300x300:AREA:RGB24:PLANAR:1; do
  printf "%-44s" $c; for a in 0 1 2 3; do echo -n " | alias $a: "; one $c $a; done; echo
done
