#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-20s %-22s %-7s %-7s %s " "$1" $2 $3 $4 $5
  env $1 python bench.py --custom $2:BICUBIC:$3:$4:$5 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for g in 3840x2160:960x540 3840x2160:1024x576 1920x1080:1024x576 3840x2160:1536x864 1920x1080:768x432 1920x1080:480x270 3840x2160:768x432 1920x1080:640x480 1920x1080:2560x1440 1280x720:1024x576 1920x1080:1536x864; do for e in X=1 TSVPP_BICUBIC_INT=0; do row $e $g RGB24 PLANAR 1; row $e $g RGB24 MERGED 0; done; done; } > $O/bicubic_int_vs_cols.txt 2>&1; cat $O/bicubic_int_vs_cols.txt
