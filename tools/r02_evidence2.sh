#!/bin/bash
# round-2 end-of-round evidence (one box, one commit): the driver's literal bench command, the default run, rocprofv3 kernel-trace
# + PMC passes for the headline, C2..C5, AREA and BICUBIC on the headline geometry, resize and output-flavour matrices.
# Every step has its own timeout.  Raw output -> gpurun_out/, summaries are copied into profiles/ by tools/save_profile.sh.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
timeout 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
timeout 120 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?"
for w in headline c2 c3 c4 c5; do timeout 400 bash tools/profile.sh $w --workload $w > $O/prof_$w.log 2>&1; grep tsvpp $O/prof_$w/kt/kt_kernel_stats.csv | cut -c1-160; done
timeout 400 bash tools/profile.sh area --resize AREA > $O/prof_area.log 2>&1; grep tsvpp $O/prof_area/kt/kt_kernel_stats.csv | cut -c1-160
timeout 400 bash tools/profile.sh bicubic --resize BICUBIC > $O/prof_bicubic.log 2>&1; grep tsvpp $O/prof_bicubic/kt/kt_kernel_stats.csv | cut -c1-160
timeout 400 bash tools/matrix.sh > $O/matrix.txt 2>&1
timeout 400 bash tools/outmatrix.sh > $O/outmatrix.txt 2>&1
tail -3 $O/matrix.txt $O/outmatrix.txt
