#!/bin/bash
# same-box A/B of two builds of the library: lib/ab/libtsvpp_old.so (vpp_bicubic_r32.hip / vpp_bilinear_up2.hip of the previous commit) against lib/ab/libtsvpp_new.so, alternating
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
L=tensor-stream_amd/lib
row() { printf "%-4s %-20s %-9s %-7s %-7s norm=%s " $1 $2 $3 $4 $5 $6
  python bench.py --custom $2:$3:$4:$5:$6 --steps 20 --warmup 3 --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:]))"; }
{ for rep in 1 2 3; do for v in old new; do cp $L/ab/libtsvpp_$v.so $L/libtsvpp.so
  row $v 1920x1080:1280x720 BICUBIC RGB24 PLANAR 1; row $v 3840x2160:1920x1080 BICUBIC RGB24 PLANAR 1; row $v 1920x1080:1280x720 BICUBIC RGB24 MERGED 0; row $v 1920x1080:960x540 BICUBIC RGB24 PLANAR 1
  row $v 960x540:1920x1080 BILINEAR RGB24 MERGED 0; row $v 960x540:1920x1080 BILINEAR RGB24 PLANAR 0; row $v 960x540:1920x1080 BILINEAR NV12 MERGED 0
done; done; } > $O/ab_merge_waits.txt 2>&1; sort -k2,7 -s $O/ab_merge_waits.txt
