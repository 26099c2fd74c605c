#!/bin/bash
# round-2 GPU call 4: AREA box kernel parity + speed, PMC of the integer BICUBIC kernel and of C5
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
one() { env $1 python bench.py --steps 30 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:20]))"; }
{ for e in "X=1" "TSVPP_AREA_BOX=0"; do
  echo -n "c5 $e: "; one "$e" --workload c5
  echo -n "4k->960x540 AREA $e: "; one "$e" --custom 3840x2160:960x540:AREA:BGR24:PLANAR:1
  echo -n "4k->640x480 AREA $e: "; one "$e" --custom 3840x2160:640x480:AREA:BGR24:PLANAR:1
  echo -n "1080p->384x216 AREA u8 $e: "; one "$e" --custom 1920x1080:384x216:AREA:RGB24:MERGED:0
done
for sh in "32,8" "64,4" "128,2" "16,16" "32,4" "64,2"; do echo -n "c5 box SHAPE=$sh: "; one "TSVPP_SHAPE=$sh" --workload c5; done
for sh in "32,8" "64,4" "128,2"; do echo -n "c5 direct SHAPE=$sh: "; one "TSVPP_SHAPE=$sh TSVPP_AREA_BOX=0" --workload c5; done
} 2>&1 | tee $O/call4.txt
timeout 600 bash tools/profile.sh bicubic --resize BICUBIC > $O/prof_bicubic.log 2>&1
python tools/pmc_summary.py $O/prof_bicubic/pmc_sq/sq_counter_collection.csv $O/prof_bicubic/pmc_lds/lds_counter_collection.csv $O/prof_bicubic/pmc_fetch/fetch_counter_collection.csv $O/prof_bicubic/pmc_write/write_counter_collection.csv | tee $O/prof_bicubic_summary.txt
head -2 $O/prof_bicubic/kt/kt_kernel_stats.csv | cut -c1-200
timeout 600 bash tools/profile.sh c5 --workload c5 > $O/prof_c5.log 2>&1
python tools/pmc_summary.py $O/prof_c5/pmc_sq/sq_counter_collection.csv $O/prof_c5/pmc_lds/lds_counter_collection.csv $O/prof_c5/pmc_fetch/fetch_counter_collection.csv $O/prof_c5/pmc_write/write_counter_collection.csv | tee $O/prof_c5_summary.txt
grep tsvpp $O/prof_c5/kt/kt_kernel_stats.csv | cut -c1-200
