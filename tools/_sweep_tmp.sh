one() { env $1 python bench.py --custom $2 --steps 20 --warmup 3 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps %.3f' % (r['value'], r['roofline']['frac']), end='')"; }
for c in "1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0" "1920x1080:1280x720:BILINEAR:RGB24:MERGED:0" "1920x1080:1280x720:AREA:RGB24:MERGED:0" "1920x1080:1280x720:NEAREST:RGB24:PLANAR:0" "1920x1080:1280x720:BILINEAR:UYVY:MERGED:0" "1920x1080:1280x720:BILINEAR:YUV444:MERGED:0" "1920x1080:1280x720:BILINEAR:Y800:MERGED:0"; do
  printf "%-46s" "$c"; for e in "TSVPP_SHAPE=64,4" "TSVPP_SHAPE=32,8" "TSVPP_SHAPE=32,4"; do echo -n " | $e: "; one "$e" $c; done; echo
done
