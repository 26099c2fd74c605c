#!/bin/bash
export TSVPP_DEBUG_KNOBS=1  # the A/B knobs are honoured only under this gate (round 6)
# Copies the judged summaries of gpurun_out/prof_<tag>/ into profiles/<round>_<tag>_*: tools/save_profile.sh r01 headline
R=$1; T=$2; S=gpurun_out/prof_$T; D=profiles
head -4 $S/kt/kt_kernel_stats.csv > $D/${R}_${T}_kernel_stats.csv
K=$(grep -h '"metric"' $S/kt.log | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["roofline"]["kernel"])')
{ echo "# rocprofv3 PMC means per dispatch of $K ONLY (separate passes), workload: $T";
  python tools/pmc_summary.py --kernel "$K" $S/pmc_sq/sq_counter_collection.csv $S/pmc_lds/lds_counter_collection.csv $S/pmc_fetch/fetch_counter_collection.csv $S/pmc_write/write_counter_collection.csv; } > $D/${R}_${T}_pmc.txt
grep -h '"metric"' $S/kt.log | tail -1 > $D/${R}_${T}_bench_under_rocprof.json
python tools/traffic_json.py $R $T $S
ls -la $D
