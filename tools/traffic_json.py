"""Derives HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh and records them in
profiles/traffic_latest.json (read by bench.py for roofline.traffic).

Only the dispatches of the kernel the profiled bench line names (roofline.kernel) are averaged (tools/pmc_lib.py); the
entry records that kernel's exact name in the CSV, its dispatch count, and the read / write split next to the algorithmic
split, so that a blend of several kernels cannot pass for one kernel's traffic again.

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and reports exactly
half of the bytes of a wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE (KiB) is checked against a known
byte count in the same run: the kernel's output is written exactly once, and WRITE_SIZE*1024 equals it to the byte.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import pmc_lib


def entry(rnd, tag, fetch_csv, write_csv, bench_line, src_hash):
    """The traffic_latest.json entry of one profiled workload (pure function of the two CSVs and the bench line)."""
    kernel = bench_line["roofline"].get("kernel")
    fm, fname = pmc_lib.kernel_means(pmc_lib.load(fetch_csv), kernel)
    wm, wname = pmc_lib.kernel_means(pmc_lib.load(write_csv), kernel)
    if fname is None or wname is None or "FETCH_SIZE" not in fm or "WRITE_SIZE" not in wm:
        raise SystemExit(f"traffic_json: kernel {kernel!r} has no FETCH_SIZE / WRITE_SIZE rows in the PMC passes")
    if fname != wname:
        raise SystemExit(f"traffic_json: the two passes chose different kernels: {fname!r} vs {wname!r}")
    (f, nf), (w, nw) = fm["FETCH_SIZE"], wm["WRITE_SIZE"]
    fpl = bench_line["config"]["frames_per_launch"]
    bpf = bench_line["roofline"]["bytes_per_frame"]
    alg_w = bench_line["roofline"].get("write_bytes_per_frame")
    e = {"round": rnd, "kernel": kernel, "kernel_csv_name": fname.split("(")[0].replace("void ", ""), "kernel_src_sha": src_hash(kernel),
         "fetch_size_kib": f, "write_size_kib": w, "dispatches": min(nf, nw),
         "hbm_bytes_per_launch": int((2 * f + w) * 1024), "read_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024),
         "frames_per_launch": fpl, "algorithmic_bytes_per_launch": int(bpf * fpl),
         "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/profile.sh {tag}, dispatches of {base(fname)} only; FETCH_SIZE doubled (gfx950)"}
    if alg_w is not None:
        e["algorithmic_write_bytes_per_launch"] = int(alg_w * fpl)
        e["algorithmic_read_bytes_per_launch"] = int((bpf - alg_w) * fpl)
    return e


def base(name):
    return pmc_lib.base_name(name)


if __name__ == "__main__":
    from bench import kernel_src_hash  # stamps the entry: bench.py drops it once the kernel sources change
    rnd, tag, src = sys.argv[1:4]
    bench_line = json.loads([l for l in open(os.path.join(src, "kt.log")) if l.startswith('{"metric"')][-1])
    e = entry(rnd, tag, os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), os.path.join(src, "pmc_write", "write_counter_collection.csv"),
              bench_line, kernel_src_hash)
    out_path = os.path.join(HERE, "..", "profiles", "traffic_latest.json")
    db = json.load(open(out_path)) if os.path.exists(out_path) else {}
    db[tag] = e
    json.dump(db, open(out_path, "w"), indent=1)
    print(json.dumps(e))
