#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# Profiles one bench.py workload with rocprofv3 on the GPU box (run through gpurun).
#   tools/profile.sh <tag> [bench.py args...]
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2-4: PMC counters, each in its own run
# (never combined with tracing domains other than kernel-trace).  Raw output -> gpurun_out/prof_<tag>/.
# QUICK=1: the kernel trace and the two traffic passes only (no SQ / LDS counters).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-others: ONLY the named workload's kernel runs (round 3 profiled the default line with its side legs: VERDICT r03 weak #1)
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-others $@"
timeout 200 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python $R/bench.py $ARGS > $OUT/kt.log 2>&1
[ -n "$QUICK" ] || timeout 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq -o sq -- python $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
[ -n "$QUICK" ] || timeout 200 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_lds -o lds -- python $R/bench.py $ARGS > $OUT/pmc_lds.log 2>&1
timeout 200 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
# drop bulky per-dispatch traces of non-vpp kernels: keep CSVs only
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
du -sh $OUT
