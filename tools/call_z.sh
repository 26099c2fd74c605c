#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 100 python -m pytest tests/test_gpu_area_box.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -2
OUT=$O/prof_headline; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-others --workload headline"
timeout 60 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py $ARGS > $OUT/kt.log 2>&1
timeout 60 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 60 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
cd $R; grep tsvpp $OUT/kt/kt_kernel_stats.csv | cut -c1-140
