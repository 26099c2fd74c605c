"""Derives HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh and records them in
profiles/traffic_latest.json (read by bench.py for roofline.traffic).

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and reports exactly
half of the bytes of a wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE (KiB) is checked against a known
byte count in the same run: the kernel's output is written exactly once, and WRITE_SIZE*1024 equals it to the byte.
"""
import csv, json, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_src_hash  # stamps the entry: bench.py drops it once the kernel sources change

rnd, tag, src = sys.argv[1:4]


def mean(path, name):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "tsvpp::" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v)


f, nf = mean(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
w, nw = mean(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
bench = json.loads([l for l in open(os.path.join(src, "kt.log")) if l.startswith('{"metric"')][-1])
out_path = "profiles/traffic_latest.json"
db = json.load(open(out_path)) if os.path.exists(out_path) else {}
kernel = bench["roofline"].get("kernel")  # per-kernel stamp: a change to another kernel's translation unit does not invalidate this entry
db[tag] = {"round": rnd, "kernel": kernel, "kernel_src_sha": kernel_src_hash(kernel), "fetch_size_kib": f, "write_size_kib": w, "dispatches": min(nf, nw),
           "hbm_bytes_per_launch": int((2 * f + w) * 1024), "read_bytes": int(2 * f * 1024), "write_bytes": int(w * 1024),
           "frames_per_launch": bench["config"]["frames_per_launch"], "algorithmic_bytes_per_launch":
           int(bench["roofline"]["bytes_per_frame"] * bench["config"]["frames_per_launch"]),
           "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/profile.sh {tag}; FETCH_SIZE doubled (gfx950)"}
json.dump(db, open(out_path, "w"), indent=1)
print(json.dumps(db[tag]))
