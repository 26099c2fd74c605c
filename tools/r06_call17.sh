#!/bin/bash
# Round 6: does the streaming kernel (no LDS staging, no workgroup barrier) live shorter than the LDS kernel for ONE frame?  headline at 1 .. 16 frames per launch, TSVPP_R32=2 against the default.
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O
C=$R/tensor-stream_amd/lib/vpp_curve
HL="1920 1080 2048 0 0 0 0 1280 720 1 2 0 1 14169600"
{ for e in "X=1" "TSVPP_R32=2"; do for v in 0 1; do echo "== $e option $v"; env $e timeout 200 $C $HL 1,2,4,8,16,64 1xc,4xc 300 $v | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
mb=14169600
from collections import defaultdict
g=defaultdict(list)
for p in r['points']: g[(p['threads'],p['consumer_pool'])].append(p)
for k,pts in g.items():
    print('  threads=%d pool=%d' % k, ' '.join('n=%d:%.2fus/%.3f' % (p['n'],p['us_per_launch'],p['n']*mb*p['threads']/(p['us_per_launch']*1e-6)/8e12) for p in pts))
" ; done; done; } > $O/curve_r32_small_n.txt 2>&1
cat $O/curve_r32_small_n.txt
