#!/usr/bin/env python3
"""Reads frames through TensorStreamConverter on ROCm and optionally dumps them -- the workflow of the reference's
python_examples/simple.py on this repository's VPP path.  Without a hardware decoder in the image the input is a
`synthetic://WxH?seed=..&frames=..&fps=..` generator or a raw `.nv12` file (see tensor_stream/sources.py):

    python tensor-stream_amd/examples/simple.py -i "synthetic://1920x1080?frames=50&fps=0" -w 1280 --height 720 \\
        --fourcc BGR24 --planes PLANAR --resize_type BILINEAR --normalize -n 50 -o /tmp/dump
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tensor_stream import FourCC, FrameRate, LogsLevel, LogsType, Planes, ResizeType, TensorStreamConverter  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-i", "--input", default="synthetic://1920x1080?frames=100&fps=0")
    ap.add_argument("-o", "--output", default=None, help="dump every frame read to <output>.yuv")
    ap.add_argument("-w", "--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--fourcc", default="RGB24", choices=[f.name for f in FourCC])
    ap.add_argument("--planes", default="MERGED", choices=[p.name for p in Planes])
    ap.add_argument("--resize_type", default="NEAREST", choices=[r.name for r in ResizeType])
    ap.add_argument("--crop", default="0,0,0,0", help="left,top,right,bottom")
    ap.add_argument("--normalize", action="store_true")
    ap.add_argument("-n", "--number", type=int, default=0, help="stop after this frame index")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--buffer_size", type=int, default=5)
    ap.add_argument("--framerate_mode", default="NATIVE", choices=[m.name for m in FrameRate])
    ap.add_argument("-v", "--verbose", default="LOW", choices=[lv.name for lv in LogsLevel])
    args = ap.parse_args()

    reader = TensorStreamConverter(args.input, cuda_device=args.device, buffer_size=args.buffer_size,
                                   framerate_mode=FrameRate[args.framerate_mode])
    reader.enable_logs(LogsLevel[args.verbose], LogsType.CONSOLE)
    reader.initialize()
    reader.start()
    if args.output and os.path.exists(args.output + ".yuv"):
        os.remove(args.output + ".yuv")
    params = dict(pixel_format=FourCC[args.fourcc], width=args.width, height=args.height,
                  crop_coords=tuple(int(v) for v in args.crop.split(",")), normalization=args.normalize,
                  planes_pos=Planes[args.planes], resize_type=ResizeType[args.resize_type])
    tensor, frames, t0 = None, 0, time.perf_counter()
    try:
        while True:
            tensor, index = reader.read(**params, return_index=True)
            frames += 1
            if args.output:
                reader.dump(tensor, args.output, **params)
            if args.number and index >= args.number:
                break
    except RuntimeError as e:  # "Decoding finished" at the end of a finite source, like the reference
        print(f"stopped: {e}")
    finally:
        dt = time.perf_counter() - t0
        print("Frame size:", reader.frame_size, "FPS of the source:", reader.fps)
        if tensor is not None:
            print("Tensor shape:", tuple(tensor.shape), "dtype:", tensor.dtype, "device:", tensor.device)
        print(f"{frames} frames read in {dt:.3f} s ({frames / dt:.0f} per second through the facade, one Convert per read)")
        reader.stop()


if __name__ == "__main__":
    main()
