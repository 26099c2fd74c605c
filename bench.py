#!/usr/bin/env python3
"""bench.py -- throughput of the fused VPP hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic NV12 frames: ONE launch of the fused
crop+resize+colour kernel converting `--batch` (default 64) independent frames
(tsvpp_convert_batch).  Default workload = BASELINE.json's metric: 1920x1080 NV12 -> 1280x720
BILINEAR -> BGR24 PLANAR fp32 (normalised).  Inputs are resident in HBM before the timed region;
the batch rotates over enough buffer sets that the working set is > 512 MiB (Infinity Cache is 256 MiB).

Prints ONE JSON line (rank 0).  N>1: launched by torch.distributed.run, one rank per GPU; frames
shard by rank with no data-path collective (weak scaling); the only collective is a one-off RCCL
broadcast of the 8 colour coefficients, verified against the compiled-in defaults.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tensor-stream_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (src_w, src_h, pitch, crop, dst, resize, fourcc, planes, norm)
    "headline": (1920, 1080, 2048, (0, 0, 0, 0), (1280, 720), "BILINEAR", "BGR24", "PLANAR", True),
    "c2": (1920, 1080, 2048, (0, 0, 0, 0), (0, 0), "NEAREST", "BGR24", "PLANAR", True),
    "c3": (1920, 1080, 2048, (0, 0, 1280, 720), (256, 256), "BILINEAR", "RGB24", "PLANAR", True),
    "c4": (3840, 2160, 3840, (0, 0, 0, 0), (1280, 720), "BICUBIC", "BGR24", "MERGED", False),
    "c5": (3840, 2160, 3840, (0, 0, 0, 0), (640, 360), "AREA", "BGR24", "PLANAR", True),
}
RESIZE = {"NEAREST": 0, "BILINEAR": 1, "BICUBIC": 2, "AREA": 3}
FOURCC = {"Y800": 0, "RGB24": 1, "BGR24": 2, "NV12": 3, "UYVY": 4, "YUV444": 5, "HSV": 6}
PLANES = {"PLANAR": 0, "MERGED": 1}


def algorithmic_bytes(src_w, src_h, crop, dst, norm, channels=3.0, luma_only=False):
    """ROI_w*ROI_h*3/2 + dst_w*dst_h*channels*sizeof(T) (Y800 does not read the chroma plane)."""
    cw, ch = crop[2] - crop[0], crop[3] - crop[1]
    roi_w, roi_h = (cw, ch) if (0 < cw < src_w and 0 < ch < src_h) else (src_w, src_h)
    dw, dh = dst if (dst[0] and dst[1]) else (roi_w, roi_h)
    return roi_w * roi_h * (2 if luma_only else 3) // 2 + int(dw * dh * channels) * (4 if norm else 1)


def cpu_baseline(spec, budget_s=12.0):
    """The CPU oracle (same arithmetic, bit-comparable with the GPU output) on all host cores:
    one frame per thread (frames are independent, exactly as they shard across GPUs), bounded sample."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    src_w, src_h, pitch, crop, dst, rt, fcc, planes, norm = spec
    if args.tight_pitch:
        pitch = src_w
    cores = O.host_cores()
    rng = np.random.default_rng(1)
    y = rng.integers(0, 256, (src_h, src_w), dtype=np.uint8)
    uv = rng.integers(0, 256, (src_h // 2, src_w), dtype=np.uint8)
    kw = dict(crop=crop, dst=dst, resize_type=RESIZE[rt], fourcc=FOURCC[fcc], planes=PLANES[planes],
              normalization=norm, nthreads=1)
    t1 = time.perf_counter()
    O.convert(y, uv, **kw)  # warm-up + single-core time per frame
    per_frame = time.perf_counter() - t1
    deadline = time.perf_counter() + budget_s

    def work(_):
        k = 0
        while True:
            O.convert(y, uv, **kw)  # ctypes releases the GIL during the call
            k += 1
            if time.perf_counter() > deadline:
                return k

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        n = sum(ex.map(work, range(cores)))
    el = time.perf_counter() - t0
    return {"value": round(n / el, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} frames of the same workload in {el:.1f} s: oracle/vpp_oracle.c, {cores} threads, "
                      f"1 frame per thread ({per_frame * 1e3:.1f} ms per frame on one idle core)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="frames per step (one launch per 64)")
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--resize", default=None, choices=sorted(RESIZE), help="override the resize type")
    ap.add_argument("--sets", type=int, default=3, help="rotating buffer sets")
    ap.add_argument("--custom", default=None, help="ad-hoc workload SRCWxSRCH:DSTWxDSTH:RESIZE:FOURCC:PLANES:NORM, e.g. 1920x1080:224x224:BILINEAR:RGB24:PLANAR:1")
    ap.add_argument("--per-call", type=int, default=0, help="frames per C-ABI call (default: the whole batch); 1 = the reference's one Convert per frame")
    ap.add_argument("--graph", action="store_true", help="capture a step's calls in a hipGraph and replay it (launch-bound small calls)")
    ap.add_argument("--tight-pitch", action="store_true", help="source pitch = width instead of width rounded up to 256 bytes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("TSVPP_BENCH_FORCE_DIST") == "1":  # the latter: exercise the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = local if dist is not None else 0
    torch.cuda.set_device(dev)

    import tensor_stream as ts
    from tensor_stream import parallel

    spec = list(WORKLOADS[args.workload])
    if args.custom:
        a = args.custom.split(":")
        sw, sh = (int(x) for x in a[0].split("x"))
        dw, dh = (int(x) for x in a[1].split("x"))
        spec = [sw, sh, (sw + 255) // 256 * 256, (0, 0, 0, 0), (dw, dh), a[2], a[3], a[4], a[5] == "1"]
        args.workload = "custom"
    if args.resize:
        spec[5] = args.resize
    spec = tuple(spec)
    src_w, src_h, pitch, crop, dst, rt, fcc, planes, norm = spec
    if args.tight_pitch:
        pitch = src_w
    vpp = ts.VideoProcessor(device=dev, max_consumers=8)
    # the one collective of the path: rank 0's colour coefficient block -> every rank (RCCL over xGMI)
    parallel.broadcast_coeffs(vpp, dist)

    fp = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=RESIZE[rt],
                            pixel_format=FOURCC[fcc], planes_pos=PLANES[planes], normalization=norm)
    vpp.prepare(fp, src_w, src_h)
    B = args.batch
    # synthetic full-range NV12, distinct per frame / set / rank
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    sets = []
    for _ in range(args.sets):
        ys = torch.randint(0, 256, (B, src_h, pitch), dtype=torch.uint8, device="cuda", generator=g)
        uvs = torch.randint(0, 256, (B, src_h // 2, pitch), dtype=torch.uint8, device="cuda", generator=g)
        out = vpp._alloc(fp.parameters, src_w, src_h, B)
        sets.append((ys, uvs, out))
    chans = {0: 1.0, 3: 1.5, 4: 2.0}.get(FOURCC[fcc], 3.0)
    bytes_per_frame = algorithmic_bytes(src_w, src_h, crop, dst, norm or fcc == "HSV", chans, luma_only=(fcc == "Y800"))
    ws_mib = sum(a.numel() * a.element_size() for s in sets for a in s) / 2**20

    parity = "skipped"
    if not args.no_parity and rank == 0:
        from oracle import oracle as O
        ys, uvs, out = sets[0]
        vpp.convert_batch(ys[:2], uvs[:2], fp, out=out[:2], width=src_w)
        torch.cuda.synchronize()
        ref, _, _ = O.convert(ys[1].cpu().numpy(), uvs[1].cpu().numpy(), crop=crop, dst=dst, resize_type=RESIZE[rt],
                              fourcc=FOURCC[fcc], planes=PLANES[planes], normalization=norm, nthreads=min(16, O.host_cores()), width=src_w)
        got = out[1].cpu().numpy().ravel()
        same = np.array_equal(got.view(np.uint8), ref.view(np.uint8))
        parity = "bit-exact vs oracle" if same else "MISMATCH vs oracle"
        if not same:
            print(json.dumps({"error": "parity gate failed", "workload": args.workload}), flush=True)
            sys.exit(2)

    # descriptor arrays are built once per buffer set; a step is then a single C-ABI call
    F = args.per_call if 0 < args.per_call < B else B
    batches = [[vpp.make_batch(ys[k:k + F], uvs[k:k + F], fp, out=out[k:k + F], width=src_w) for k in range(0, B, F)] for (ys, uvs, out) in sets]
    cur_stream = torch.cuda.current_stream(dev).cuda_stream

    def issue(i, stream):
        for b in batches[i % len(batches)]:
            vpp.run_batch(b, stream)

    graphs = []
    if args.graph:  # one graph per buffer set, captured on a side stream, replayed on the current one
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for i in range(len(batches)):
                issue(i, side.cuda_stream)
        torch.cuda.synchronize()
        for i in range(len(batches)):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                issue(i, side.cuda_stream)
            graphs.append(gr)

    def step(i):
        if graphs:
            graphs[i % len(graphs)].replay()
        else:
            issue(i, cur_stream)

    for i in range(args.warmup):
        step(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # on torch's current stream == the stream convert_batch launches on
    for i in range(args.steps):
        step(i)
    ev1.record()
    host_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        tt = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, dev_ms = tt[0].item(), tt[1].item()

    if rank == 0:
        launches_per_step = ((F + 63) // 64) * (B // F) + ((B % F + 63) // 64)
        frames = B * args.steps * world
        kernel_ms = dev_ms / (args.steps * launches_per_step)  # avg launch duration from HIP events
        frames_per_launch = B / launches_per_step
        achieved = bytes_per_frame * frames_per_launch / (kernel_ms * 1e-3) / 1e9
        res = {
            "metric": "1080p NV12\u2192720p BGR24 planar fp32 frames/sec per GPU; achieved HBM GB/s vs roofline"
            if args.workload == "headline" else f"{args.workload} frames/sec",
            "value": round(frames / wall, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if norm else "u8", "data": "synthetic",
            "config": {"workload": f"{src_w}x{src_h} NV12 (pitch {pitch}) crop{list(crop)} -> {dst[0] or src_w}x{dst[1] or src_h} "
                                   f"{rt if dst[0] else 'no-resize'} -> {fcc} {planes} {'fp32 /255' if norm else 'uint8'}",
                       "name": args.workload, "frames_per_step": B, "frames_per_launch": frames_per_launch, "hip_graph": bool(graphs),
                       "buffer_sets": len(sets), "working_set_MiB": round(ws_mib, 1), "sharding": f"frames/{world} ranks, no data collective",
                       "parity": parity},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "tsvpp::vpp_*_kernel (one fused launch)", "bytes_per_frame": bytes_per_frame,
                         "avg_launch_ms": round(kernel_ms, 5), "host_issue_ms_per_step": round(host_issue * 1e3 / args.steps, 4)},
        }
        # HBM traffic per launch from the committed PMC passes (tools/profile.sh -> tools/traffic_json.py); the counters
        # need their own rocprofv3 runs, so this is the last profiled value for this workload, not a live one
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json"))).get(args.workload if not args.resize else "")
            if tr and tr["frames_per_launch"] == frames_per_launch:
                res["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                res["roofline"]["traffic_source"] = f"profiles/traffic_latest.json ({tr['round']}): 2*FETCH_SIZE+WRITE_SIZE, KiB"
        except (OSError, ValueError, KeyError):
            pass
        # BASELINE.json's metric names no resize type ("report all four"): the timed region above is
        # BILINEAR; the other three on the same buffers, 20 launches each, outside the timed region
        if world == 1 and args.workload == "headline" and not args.resize and not args.no_cpu_baseline:
            others = {}
            for name in ("NEAREST", "BICUBIC", "AREA"):
                fp2 = ts.FrameParameters(width=dst[0], height=dst[1], crop_coords=crop, resize_type=RESIZE[name],
                                         pixel_format=FOURCC[fcc], planes_pos=PLANES[planes], normalization=norm)
                vpp.prepare(fp2, src_w, src_h)
                bs = [vpp.make_batch(ys, uvs, fp2, out=out, width=src_w) for (ys, uvs, out) in sets]
                for i in range(3):
                    vpp.run_batch(bs[i % len(bs)], cur_stream)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(20):
                    vpp.run_batch(bs[i % len(bs)], cur_stream)
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 20
                others[name] = {"frames_per_s": round(B / (ms * 1e-3), 1), "hbm_frac": round(bytes_per_frame * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            res["config"]["other_resize_types"] = others
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(spec)
        print(json.dumps(res), flush=True)
    vpp.Close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
