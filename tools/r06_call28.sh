#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_api.py tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -4
row() { printf "%-20s %-9s %-7s %-7s norm=%s " $1 $2 $3 $4 $5
  python bench.py --custom $1:$2:$3:$4:$5 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for g in 1280x720:1280x720 2560x1440:2560x1440 3840x2160:3840x2160 2048x1152:2048x1152 1920x1080:1920x1080; do row $g NEAREST BGR24 PLANAR 1; done; } > $O/color_shapes_after.txt 2>&1; cat $O/color_shapes_after.txt
