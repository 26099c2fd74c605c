#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}; cd $R; export TSVPP_DEBUG_KNOBS=1
O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in 1366x768:1366x768:BILINEAR:UYVY:MERGED:0 1366x768:1366x768:BILINEAR:YUV444:MERGED:0 1366x768:1366x768:BILINEAR:UYVY:MERGED:1; do
  t=$(echo $c | tr ':' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$t -o kt -- python $R/bench.py --custom $c --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-parity > /dev/null 2>&1
  echo "== $c"; head -4 $(find /tmp/kt_$t -name "*kernel_stats.csv") | cut -c1-200
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d /tmp/pm_$t -o pm -- python $R/bench.py --custom $c --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-parity > /dev/null 2>&1
  python $R/tools/pmc_summary.py --kernel fmt_ $(find /tmp/pm_$t -name "*counter_collection.csv") 2>&1 | head -20
  rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -d /tmp/pm2_$t -o pm -- python $R/bench.py --custom $c --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-parity > /dev/null 2>&1
  python $R/tools/pmc_summary.py --kernel fmt_ $(find /tmp/pm2_$t -name "*counter_collection.csv") 2>&1 | head -20
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE WRITE_SIZE TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_REQ_sum -d /tmp/pm3_$t -o pm -- python $R/bench.py --custom $c --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-parity > /dev/null 2>&1
  python $R/tools/pmc_summary.py --kernel fmt_ $(find /tmp/pm3_$t -name "*counter_collection.csv") 2>&1 | head -20
done > $O/fmt_onepair_prof.txt 2>&1
cat $O/fmt_onepair_prof.txt
