#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_bicubic_r32.py tests/test_gpu_formats.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
row() { printf "%-20s %-9s %-7s %-7s norm=%s " $1 $2 $3 $4 $5
  python bench.py --custom $1:$2:$3:$4:$5 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ row 1920x1080:1280x720 BICUBIC RGB24 PLANAR 1; row 3840x2160:1920x1080 BICUBIC RGB24 PLANAR 1; row 1920x1080:1280x720 BICUBIC RGB24 MERGED 1; row 1920x1080:1280x720 BICUBIC NV12 MERGED 1; row 1920x1080:1280x720 BICUBIC HSV MERGED 1
  row 1920x1080:1280x720 BICUBIC RGB24 MERGED 0; row 1920x1080:1280x720 BICUBIC RGB24 PLANAR 0; row 3840x2160:1920x1080 BICUBIC RGB24 MERGED 0; row 1920x1080:960x540 BICUBIC RGB24 PLANAR 1
  row 1366x768:1366x768 BILINEAR YUV444 MERGED 0; row 1366x768:1366x768 BILINEAR YUV444 MERGED 1; } > $O/bicubic_r32_loads.txt 2>&1; cat $O/bicubic_r32_loads.txt
