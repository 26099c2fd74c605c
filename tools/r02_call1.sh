#!/bin/bash
# round-2 GPU call 1: parity suite, the driver's literal bench command, HBM floor, u8 PMC, cheap headline A/Bs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; cut -c1-600 $O/bench_driver.json
timeout 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2.json 2>&1; echo "bench --gpus 2 rc=$?"; cut -c1-300 $O/bench_gpus2.json
timeout 300 tools/bin/membench2 > $O/membench2.txt 2>&1; cat $O/membench2.txt
one() { env $1 python bench.py --steps 30 --no-cpu-baseline --no-parity --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms']))"; }
for e in "X=0" "TSVPP_RPT=3" "TSVPP_RPT=1" "TSVPP_SHAPE=64,4" "TSVPP_SHAPE=64,4 TSVPP_RPT=4" "TSVPP_LDS_KB=64 TSVPP_RPT=3" "X=1"; do echo -n "headline $e: "; one "$e"; done 2>&1 | tee $O/ab_headline.txt
timeout 600 bash tools/profile.sh u8p --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0 > $O/prof_u8p.log 2>&1
python tools/pmc_summary.py $O/prof_u8p/pmc_sq/sq_counter_collection.csv $O/prof_u8p/pmc_lds/lds_counter_collection.csv | tee $O/prof_u8p_summary.txt
head -3 $O/prof_u8p/kt/kt_kernel_stats.csv | cut -c1-160
