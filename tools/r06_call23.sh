#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_up2.py tests/test_gpu_rep2.py tests/test_gpu_bicubic_r32.py tests/test_gpu_fuzz.py tests/test_gpu_formats.py -m gpu -x -q 2>&1 | tail -4
row() { printf "%-20s %-9s %-7s %-7s norm=%s " $1 $2 $3 $4 $5
  python bench.py --custom $1:$2:$3:$4:$5 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ row 960x540:1920x1080 BILINEAR RGB24 MERGED 0; row 960x540:1920x1080 BILINEAR RGB24 PLANAR 0; row 960x540:1920x1080 BILINEAR NV12 MERGED 0; row 1280x720:2560x1440 BILINEAR RGB24 MERGED 0; row 960x540:1920x1080 BILINEAR Y800 MERGED 0
  row 1920x1080:1280x720 BICUBIC RGB24 PLANAR 1; row 3840x2160:1920x1080 BICUBIC RGB24 PLANAR 1; } > $O/up2_loads.txt 2>&1; cat $O/up2_loads.txt
