#!/bin/bash
# round-2 GPU call 6: evidence refresh -- driver bench line, rocprofv3 + PMC for headline / c2 / c3 / c4 / area, matrices, knob matrix
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; cut -c1-400 $O/bench_driver.json
timeout 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
for w in headline c2 c3 c4; do timeout 600 bash tools/profile.sh $w --workload $w > $O/prof_$w.log 2>&1; grep tsvpp $O/prof_$w/kt/kt_kernel_stats.csv | cut -c1-160; done
timeout 600 bash tools/profile.sh area --resize AREA > $O/prof_area.log 2>&1; grep tsvpp $O/prof_area/kt/kt_kernel_stats.csv | cut -c1-160
timeout 900 bash tools/matrix.sh > $O/matrix.txt 2>&1; cat $O/matrix.txt
timeout 900 bash tools/outmatrix.sh > $O/outmatrix.txt 2>&1; cat $O/outmatrix.txt
timeout 1500 bash tools/knob_matrix.sh > $O/knob_matrix.txt 2>&1; cat $O/knob_matrix.txt
