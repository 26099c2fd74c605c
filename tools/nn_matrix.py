#!/usr/bin/env python3
"""NN-input sizes: 1080p (and 4K) -> 224^2 / 256^2 / 300^2 / 416^2, every resize type, RGB24 planar fp32, at several launch sizes, each cell
priced three ways: ROI formula, touched bytes (oracle: a lower bound of what must move), and -- with --pmc -- the HBM bytes the dispatched kernel really moved
(rocprofv3 FETCH_SIZE / WRITE_SIZE in their own passes, 2 * FETCH_SIZE + WRITE_SIZE: the guide's gfx950 correction), per frame.

  python tools/nn_matrix.py [--src 1920x1080] [--sizes 224,256,300,416] [--batches 64,256,512] [--pmc 256] [--types NEAREST,BILINEAR,BICUBIC,AREA] [--env K=V ...]
Runs bench.py once per (cell, launch size); on the GPU box (gpurun).  One line per cell on stdout."""
import argparse
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_lib  # noqa: E402


def bench(custom, batch, env, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--custom", custom, "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--batch", str(batch)] + list(extra)
    if batch > 128:
        cmd.append("--table")
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    if p.returncode != 0 or not lines:
        return None, (p.stderr or p.stdout)[-300:]
    return json.loads(lines[-1]), None


def pmc_bytes(custom, batch, env, kernel):
    """(read bytes, write bytes) per launch of `kernel` from two rocprofv3 passes, or None."""
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="nnpmc_", dir="/tmp")
        cmd = ["rocprofv3", "--output-format", "csv", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--custom", custom, "--steps", "6", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline", "--no-parity", "--batch", str(batch)] + (["--table"] if batch > 128 else [])
        try:
            subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=400, cwd="/tmp")
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None
            means, _name = pmc_lib.kernel_means(pmc_lib.load(files[0]), kernel)
            if counter not in means:
                return None
            out[counter] = means[counter][0] * 1024.0
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2.0 * out["FETCH_SIZE"], out["WRITE_SIZE"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default="1920x1080")
    ap.add_argument("--sizes", default="224,256,300,416")
    ap.add_argument("--batches", default="64,256,512")
    ap.add_argument("--types", default="NEAREST,BILINEAR,BICUBIC,AREA")
    ap.add_argument("--pmc", type=int, default=0, help="launch size at which the PMC passes run (0 = none)")
    ap.add_argument("--fmt", default="RGB24:PLANAR:1")
    ap.add_argument("--env", nargs="*", default=[])
    a = ap.parse_args()
    env = dict(os.environ, TMPDIR="/tmp")
    for kv in a.env:
        k, _, v = kv.partition("=")
        env[k] = v
    batches = [int(b) for b in a.batches.split(",")]
    print(f"# {a.src} -> N x N, {a.fmt}; frac = fraction of 8 TB/s per launch: roi = ROI formula, touched = oracle's touched bytes, moved = PMC bytes of the dispatched kernel "
          f"(at {a.pmc} frames per launch); env {a.env}", flush=True)
    for size in a.sizes.split(","):
        for rt in a.types.split(","):
            custom = f"{a.src}:{size}x{size}:{rt}:{a.fmt}"
            cells, kernel, tb, bpf = [], None, None, None
            for b in batches:
                r, err = bench(custom, b, env)
                if r is None:
                    cells.append(f"n={b}: ERROR {err!r}")
                    continue
                rf = r["roofline"]
                kernel = rf["kernel"]
                bpf, tb = rf["bytes_per_frame"], rf.get("touched_bytes")
                ms = rf["avg_launch_ms"]
                fpl = r["config"]["frames_per_launch"]
                roi = bpf * fpl / (ms * 1e-3) / 8e12
                tch = (tb * fpl / (ms * 1e-3) / 8e12) if isinstance(tb, int) else float("nan")
                cells.append((b, fpl, ms, roi, tch, r["config"]["parity"][:9]))
            moved = None
            if a.pmc and kernel:
                pm = pmc_bytes(custom, a.pmc, env, kernel)
                if pm:
                    moved = (pm[0] + pm[1]) / float(min(a.pmc, 1024))
            txt = []
            for c in cells:
                if isinstance(c, str):
                    txt.append(c)
                    continue
                b, fpl, ms, roi, tch, par = c
                mv = ("  moved %.3f" % (moved * fpl / (ms * 1e-3) / 8e12)) if moved else ""
                txt.append(f"n={b}: {ms * 1e3:7.1f} us roi {roi:.3f} touched {tch:.3f}{mv} {par}")
            print(f"{a.src}:{size}x{size} {rt:9s} {kernel and kernel[7:]:44s} bytes/frame roi {bpf} touched {tb} moved {moved and int(moved)} | " + " | ".join(txt), flush=True)


if __name__ == "__main__":
    main()
