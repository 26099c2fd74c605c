#!/bin/bash
# round-2 GPU call 12: where does the uint8 2x2-tap kernel spend its time?  Aliased buffers keep reads (1) / writes (2) / both (3) in cache.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
one() { env $1 python bench.py --steps 30 --repeats 5 --no-cpu-baseline --no-others "${@:2}" 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%9.0f fps frac %.4f launch %.5f ms %s' % (r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['parity'][:9]))"; }
{ for a in 0 1 2 3; do
  echo -n "u8 planar 1080p->720p BILINEAR alias=$a: "; one "X=1" --alias $a --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0
  echo -n "u8 merged 1080p->720p BILINEAR alias=$a: "; one "X=1" --alias $a --custom 1920x1080:1280x720:BILINEAR:RGB24:MERGED:0
  echo -n "headline alias=$a: "; one "X=1" --alias $a
  echo -n "u8 planar 1080p->720p NEAREST alias=$a: "; one "X=1" --alias $a --custom 1920x1080:1280x720:NEAREST:RGB24:PLANAR:0
  echo -n "u8 planar 1080p->720p BICUBIC alias=$a: "; one "X=1" --alias $a --custom 1920x1080:1280x720:BICUBIC:RGB24:PLANAR:0
  echo -n "f32 planar 1080p->720p BICUBIC alias=$a: "; one "X=1" --alias $a --resize BICUBIC
  echo -n "u8 planar 1080p->720p AREA alias=$a: "; one "X=1" --alias $a --custom 1920x1080:1280x720:AREA:RGB24:PLANAR:0
  echo -n "f32 planar 1080p->224x224 AREA alias=$a: "; one "X=1" --alias $a --custom 1920x1080:224x224:AREA:RGB24:PLANAR:1
  echo -n "c5 alias=$a: "; one "X=1" --alias $a --workload c5
done
for r in 1 2 4; do echo -n "u8 planar BILINEAR alias=3 RPT=$r: "; one "TSVPP_RPT=$r" --alias 3 --custom 1920x1080:1280x720:BILINEAR:RGB24:PLANAR:0; done
} 2>&1 | tee $O/call12.txt
