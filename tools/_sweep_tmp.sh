one() { env $1 python bench.py --custom $2 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps %.3f %s' % (r['value'], r['roofline']['frac'], 'ok' if r['config']['parity'].startswith('bit-exact') else 'BAD'), end='')"; }
for c in "1920x1080:1280x720:AREA:BGR24:PLANAR:1" "1920x1080:1280x720:NEAREST:BGR24:PLANAR:1" "1920x1080:1280x720:AREA:RGB24:MERGED:1" "1920x1080:1280x720:NEAREST:RGB24:MERGED:1" "1920x1080:1280x720:AREA:NV12:MERGED:1" "1920x1080:1280x720:AREA:Y800:MERGED:1" \
         "3840x2160:1920x1080:AREA:RGB24:PLANAR:1" "1920x1080:960x540:AREA:RGB24:PLANAR:1" "3840x2160:1920x1080:AREA:RGB24:MERGED:1" "3840x2160:2560x1440:AREA:RGB24:PLANAR:1" "3840x2160:2560x1440:NEAREST:RGB24:PLANAR:1"; do
  printf "%-46s" "$c"; for e in "TSVPP_BILINEAR_INT=2" "TSVPP_BILINEAR_INT=1"; do echo -n " | $e: "; one "$e" $c; done; echo
done
