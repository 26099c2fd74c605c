#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
timeout 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python3 -c "
import json
for f in ('bench_driver','bench_default'):
    r=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, r['value'], r['roofline']['frac'], r['roofline']['traffic'], r['timing']['repeats'], r['ms_per_step'])
"
