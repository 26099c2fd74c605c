#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-28s %-20s %-9s %-7s %-7s norm=%s " "$1" $2 $3 $4 $5 $6
  env $1 python bench.py --custom $2:$3:$4:$5:$6 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps frac %.3f %s %s\" % (r[\"value\"], rf.get(\"roi_frac\", rf[\"frac\"]), rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for e in X=1 TSVPP_BILINEAR_INT=0; do row "$e" 1920x1080:1280x720 AREA Y800 MERGED 1; row "$e" 3840x2160:1920x1080 AREA Y800 MERGED 1; row "$e" 1920x1080:1280x720 BILINEAR Y800 MERGED 1; row "$e" 3840x2160:1920x1080 BILINEAR Y800 MERGED 1; done; } > $O/fmt_ab2.txt 2>&1
cat $O/fmt_ab2.txt
QUICK=1 bash tools/r06_evidence.sh 1
