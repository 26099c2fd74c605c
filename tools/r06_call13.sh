#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-20s %-9s %-7s %-7s norm=%s " $1 $2 $3 $4 $5
  python bench.py --custom $1:$2:$3:$4:$5 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for n in 0 1; do
  row 1920x1080:1366x768 BILINEAR UYVY MERGED $n; row 1920x1080:1366x768 BILINEAR NV12 MERGED $n; row 1920x1080:1366x768 BILINEAR YUV444 MERGED $n
  row 1920x1080:1080x608 BILINEAR YUV444 MERGED $n; row 1920x1080:1080x608 BILINEAR UYVY MERGED $n; row 1920x1080:1080x608 BILINEAR NV12 MERGED $n
  row 1280x720:1920x1080 BILINEAR NV12 MERGED $n; row 1280x720:1920x1080 BILINEAR UYVY MERGED $n; row 1280x720:1920x1080 BILINEAR Y800 MERGED $n; row 1280x720:1920x1080 BILINEAR RGB24 MERGED $n
  row 1366x768:1366x768 BILINEAR YUV444 MERGED $n; row 1366x768:1366x768 BILINEAR UYVY MERGED $n
done; } > $O/formats_odd.txt 2>&1; cat $O/formats_odd.txt
