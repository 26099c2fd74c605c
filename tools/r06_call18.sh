#!/bin/bash
# Round 6: where do the 6.5 us of a one-frame launch go?  Kernel arguments in DEVICE memory (HIP_FORCE_DEV_KERNARG=1: the runtime writes them through the BAR, the waves' first
# scalar loads stay on the GPU) against the default (host memory: the first s_load of every launch crosses the host link).
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O
C=$R/tensor-stream_amd/lib/vpp_curve
HL="1920 1080 2048 0 0 0 0 1280 720 1 2 0 1 14169600"
C3="1920 1080 2048 0 0 1280 720 256 256 1 1 0 1 1770244"
C4="3840 2160 3840 0 0 0 0 1280 720 2 2 1 0 6912924"
show() { python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
from collections import defaultdict
g=defaultdict(list)
for p in r['points']: g[(p['threads'],p['consumer_pool'])].append(p)
for k,pts in g.items():
    print('  threads=%d' % k[0], ' '.join('n=%d:%.2fus(host %.2f)' % (p['n'],p['us_per_launch'],p['host_issue_us_per_launch']) for p in pts))"; }
{ for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "ROC_USE_FGS_KERNARG=0"; do for v in 0 1; do
  echo "== $e option $v headline"; env $e timeout 200 $C $HL 1,2,4,8,64 1xc,4xc 300 $v | show
  echo "== $e option $v c3"; env $e timeout 200 $C $C3 1,2,4,8,64 1xc 300 $v | show
  echo "== $e option $v c4"; env $e timeout 200 $C $C4 1,2,4,8,64 1xc 300 $v | show
done; done; hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_cost tools/launch_cost.hip 2>/dev/null; /tmp/launch_cost 2>/dev/null | head -12; echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 /tmp/launch_cost 2>/dev/null | head -12; } > $O/curve_dev_kernarg.txt 2>&1
cat $O/curve_dev_kernarg.txt
