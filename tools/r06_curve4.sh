#!/bin/bash
# Round 6, fourth pass: final thresholds, facade batch reads, vpp_latency on a rotating ring + graph8, rocprofv3 kernel traces of the n = 1 and n = 8 launches.
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
R=$PWD
O=$R/gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
C=$R/tensor-stream_amd/lib/vpp_curve
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_suite4.txt 2>&1; tail -3 $O/gpu_suite4.txt
timeout 200 $R/tensor-stream_amd/lib/vpp_latency > $O/latency4.json 2>&1; cat $O/latency4.json
timeout 1200 python bench.py --curve-only headline,c3,c4 > $O/curve_fourth.json 2> $O/curve_fourth.err; tail -c 600 $O/curve_fourth.err
HL="1920 1080 2048 0 0 0 0 1280 720 1 2 0 1 14169600"
C4="3840 2160 3840 0 0 0 0 1280 720 2 2 1 0 6912000"
cd /tmp
for n in 1 8; do
  for v in 0 1; do
    timeout 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_hl_n${n}_v$v -o kt -- $C $HL $n 1xc 10 $v > $O/prof_hl_n${n}_v$v.log 2>&1
    timeout 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_c4_n${n}_v$v -o kt -- $C $C4 $n 1xc 10 $v > $O/prof_c4_n${n}_v$v.log 2>&1
  done
done
find $O -name "*kernel_stats.csv" | head -20
# keep the stats and a slice of the trace only
for d in $O/prof_*; do [ -d $d ] || continue; f=$(find $d -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && head -400 $f > $d/kernel_trace_head.csv; find $d -type f ! -name "*kernel_stats.csv" ! -name "kernel_trace_head.csv" -delete; done
du -sh $O
