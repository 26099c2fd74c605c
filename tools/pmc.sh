#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# PMC passes (each in its own run, kernel-trace only) of one bench.py workload under an environment setting:
#   tools/pmc.sh <tag> "<ENV=...>" [bench.py args...]   -> gpurun_out/pmc_<tag>.txt (means per dispatch of the tsvpp kernel)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; ENVS=$2; shift 2
OUT=$R/gpurun_out/pmcraw_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-parity --no-others $@"
pass() { timeout 150 env $ENVS rocprofv3 --output-format csv --pmc $2 --kernel-trace -d $OUT/$1 -o $1 -- python $R/bench.py $ARGS > $OUT/$1.log 2>&1; }
pass sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
pass lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
pass tcp "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
pass sq2 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_DEP_WAIT SQ_INSTS_WAVE32_VALU"
pass tcc "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RD_UNCACHED_32B_sum"
pass tcc2 "TCC_TAG_STALL_sum TCC_EA_RDREQ_DRAM_sum TCC_BUBBLE_sum TCC_READ_sum"
KERNEL=$(grep -h '"metric"' $OUT/sq.log | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["roofline"]["kernel"])' 2>/dev/null)
{ echo "# $TAG: $ENVS bench.py $@ -- dispatches of $KERNEL only"; for p in sq lds tcp sq2 tcc tcc2; do python $R/tools/pmc_summary.py --kernel "$KERNEL" $(find $OUT/$p -name "*counter_collection.csv"); done; } > $R/gpurun_out/pmc_$TAG.txt 2>&1
find $OUT -type f ! -name "*.log" -delete
cat $R/gpurun_out/pmc_$TAG.txt
