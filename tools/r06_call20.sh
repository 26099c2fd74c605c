#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_formats.py -m gpu -x -q 2>&1 | tail -3
row() { printf "%-18s %-20s %-9s %-7s %-7s norm=%s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for e in TSVPP_FMT_RPW=1 TSVPP_FMT_RPW=2; do for n in 1 0; do
  row $e 1920x1080:1920x1080 BILINEAR UYVY MERGED $n; row $e 1280x720:1280x720 BILINEAR UYVY MERGED $n; row $e 1920x1080:1280x720 BILINEAR UYVY MERGED $n; row $e 3840x2160:3840x2160 BILINEAR UYVY MERGED $n
done; done; } > $O/uyvy_rpw.txt 2>&1; cat $O/uyvy_rpw.txt
