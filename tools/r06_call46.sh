#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O
C=$R/tensor-stream_amd/lib/vpp_curve
HL="1920 1080 2048 0 0 0 0 1280 720 1 2 0 1 14169600"
C3="1920 1080 2048 0 0 1280 720 256 256 1 1 0 1 1770244"
C4="3840 2160 3840 0 0 0 0 1280 720 2 2 1 0 6912924"
show() { python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', ' '.join('n=%d:%.2fus' % (p['n'],p['us_per_launch']) for p in r['points']))"; }
{ for e in X=1 TSVPP_NT=0 TSVPP_NT=2; do for v in 0 1; do
  echo "== $e option $v headline"; env $e timeout 200 $C $HL 1,2,4,8,16,64 1xc 300 $v 2>/dev/null | show
  echo "== $e option $v c3"; env $e timeout 200 $C $C3 1,2,4,8,16,64 1xc 300 $v 2>/dev/null | show
  echo "== $e option $v c4"; env $e timeout 200 $C $C4 1,2,4,8,16,64 1xc 300 $v 2>/dev/null | show
done; done; } > $O/curve_nt.txt 2>&1; cat $O/curve_nt.txt
