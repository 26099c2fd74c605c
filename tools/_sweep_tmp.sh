one() { env $1 python bench.py $2 --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-others 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps %.3f' % (r['value'], r['roofline'].get('roi_frac', r['roofline']['frac'])), end='')"; }
for c in "--resize NEAREST" "--workload c4" "--workload c5" "--workload c2" "--custom 3840x2160:1920x1080:NEAREST:RGB24:PLANAR:1" "--custom 1920x1080:640x360:AREA:RGB24:PLANAR:1"; do
  printf "%-60s" "$c"; for e in "TSVPP_X=0" "TSVPP_SHAPE=64,4" "TSVPP_SHAPE=64,2" "TSVPP_SHAPE=32,4" "TSVPP_SHAPE=32,2"; do echo -n " | $e: "; one "$e" "$c"; done; echo
done
