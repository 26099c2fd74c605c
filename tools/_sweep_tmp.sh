one() { env $1 python bench.py --custom $2 --steps 20 --warmup 3 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps %.3f' % (r['value'], r['roofline']['frac']), end='')"; }
c="1920x1080:1280x720:BICUBIC:BGR24:PLANAR:1"
for e in "TSVPP_SHAPE=32,2" "TSVPP_SHAPE=64,2" "TSVPP_SHAPE=64,1" "TSVPP_SHAPE=32,4" "TSVPP_SHAPE=32,2 TSVPP_NT=0" "TSVPP_SHAPE=32,2 TSVPP_NT=2" "TSVPP_SHAPE=64,2 TSVPP_NT=0" "TSVPP_SHAPE=64,2 TSVPP_NT=2" "TSVPP_SHAPE=32,2 TSVPP_TILE_ORDER=1" "TSVPP_SHAPE=32,2 TSVPP_TILE_ORDER=2" "TSVPP_SHAPE=64,2 TSVPP_TILE_ORDER=1" "TSVPP_BICUBIC_INT=2"; do
  printf "%-44s" "$e"; one "$e" $c; echo
done
