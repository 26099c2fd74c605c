#!/bin/bash
# anomaly hunt 4: crops (ROI -> NN input; crop only), even and odd origins, planar fp32, 256 frames per launch out of a frame table for the small outputs
cd ${GRAFT_REPO_ROOT:-.}; export TSVPP_DEBUG_KNOBS=1
O=gpurun_out/r06; mkdir -p $O
row() { printf "%-10s %-22s %-9s %-20s " $1 $2 $3 $4
  python bench.py --custom $1:$2:$3:RGB24:PLANAR:1:$4 --batch 256 --table --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); rf=r[\"roofline\"]; print(\"%9.0f fps %8.1f us roi %.3f touched %.3f %s %s\" % (r[\"value\"], rf[\"avg_launch_ms\"]*1e3, rf[\"frac\"], rf.get(\"touched_frac\") or 0, rf[\"kernel\"][7:], r[\"config\"][\"parity\"][:9]))"; }
{ for c in 0,0,0,0 100,100,740,580 101,101,741,581 102,100,742,580 1000,500,1500,1000 1001,501,1501,1001; do for d in 224x224 416x416 0x0; do for rt in NEAREST BILINEAR BICUBIC AREA; do [ $d = 0x0 ] && [ $rt != NEAREST ] && continue; [ $d = 0x0 ] && [ $c = 0,0,0,0 ] && continue; row 1920x1080 $d $rt $c; done; done; done
  for c in 200,200,1800,1400 201,201,1801,1401 2000,1000,3000,2000; do for d in 224x224 640x640 0x0; do for rt in NEAREST BILINEAR BICUBIC AREA; do [ $d = 0x0 ] && [ $rt != NEAREST ] && continue; row 3840x2160 $d $rt $c; done; done; done
} > $O/crop_hunt.txt 2>&1; cat $O/crop_hunt.txt
