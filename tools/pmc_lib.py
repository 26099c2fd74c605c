"""Shared by tools/traffic_json.py and tools/pmc_summary.py: per-dispatch means of rocprofv3 counter CSVs for ONE kernel.

Round 3's versions averaged every row whose name contained "tsvpp::" -- with side legs in the profiled command that is a
blend of several kernels.  Here a CSV is reduced to the dispatches of the kernel that was asked for:
`kernel` is the name bench.py prints in roofline.kernel ("tsvpp::vpp_bilinear_kernel<bilinear,OUT>": tsvpp_describe's
spelling, template arguments symbolic), matched on its base name "tsvpp::vpp_bilinear_kernel"; when several template
instances of that base were dispatched the one with the most dispatches is taken (the timed region's), and when no kernel
is named the tsvpp kernel with the most dispatches.  The exact CSV name chosen is returned so that it can be recorded.
"""
import collections
import csv


def base_name(kernel):
    """'tsvpp::vpp_bilinear_kernel<bilinear,OUT>' / 'void tsvpp::vpp_bilinear_kernel<false, 2>(...)' -> 'tsvpp::vpp_bilinear_kernel'."""
    k = kernel.strip()
    if k.startswith("void "):
        k = k[5:]
    k = k.split("(")[0].split("<")[0].strip()
    if "tsvpp::" not in k:
        k = "tsvpp::" + k
    return k[k.index("tsvpp::"):]


def load(path):
    """rows of a *_counter_collection.csv as (kernel name, counter name, value)."""
    with open(path) as f:
        return [(r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])) for r in csv.DictReader(f)]


def pick_kernel(rows, kernel=None):
    """Exact CSV name of the kernel to reduce on (see module docstring); None if no tsvpp kernel was dispatched."""
    counts = collections.Counter()
    first_counter = None
    for name, counter, _ in rows:
        if "tsvpp::" not in name:
            continue
        if first_counter is None:
            first_counter = counter
        if counter == first_counter:  # count dispatches once, not once per counter
            counts[name] += 1
    if kernel:
        want = base_name(kernel)
        counts = collections.Counter({n: c for n, c in counts.items() if base_name(n) == want})
    if not counts:
        return None
    return counts.most_common(1)[0][0]


def kernel_means(rows, kernel=None):
    """{counter: (mean per dispatch, dispatches)} for the chosen kernel, and its exact name."""
    name = pick_kernel(rows, kernel)
    acc = collections.defaultdict(list)
    for n, counter, v in rows:
        if n == name:
            acc[counter].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}, name
