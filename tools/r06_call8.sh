#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
row() { printf "%-24s %-20s %-9s %-7s %-7s norm=%s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us frac %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{
python tools/nn_matrix.py --src 1920x1080 --sizes 300,416 --types AREA --batches 64,512
python tools/nn_matrix.py --src 1920x1080 --sizes 300,416 --types AREA --batches 64,512 --env TSVPP_AREA_DIVTAB=0
python tools/nn_matrix.py --src 1920x1080 --sizes 224,256,300,416 --types BICUBIC --batches 64,512
row X=1 1280x720:1920x1080 BICUBIC RGB24 MERGED 0
row X=1 1280x720:1920x1080 BICUBIC RGB24 PLANAR 1
row X=1 1080x608:480x360 BICUBIC RGB24 PLANAR 1
row X=1 1920x1080:1440x810 BICUBIC RGB24 PLANAR 1
row X=1 1920x1080:640x640 BICUBIC RGB24 PLANAR 1
row X=1 1080x608:480x360 AREA RGB24 PLANAR 1
} > $O/cols_ab.txt 2>&1
cut -c1-330 $O/cols_ab.txt
