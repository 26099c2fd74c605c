"""Per-dispatch counter means of ONE kernel from rocprofv3 counter CSVs:
    python tools/pmc_summary.py [--kernel 'tsvpp::vpp_bilinear_kernel<...>'] a_counter_collection.csv ...
Without --kernel: the tsvpp kernel with the most dispatches in each file (tools/pmc_lib.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_lib

args = sys.argv[1:]
kernel = None
if args and args[0] == "--kernel":
    kernel = args[1] or None
    args = args[2:]
for f in args:
    if not os.path.exists(f):  # (tools/profile.sh with QUICK=1 skips the SQ / LDS passes)
        print(f.split('/')[-1], "not collected")
        continue
    means, name = pmc_lib.kernel_means(pmc_lib.load(f), kernel)
    short = (name or "no tsvpp kernel").split("(")[0].replace("void ", "")
    for k, (m, n) in means.items():
        print(f.split('/')[-1], k, 'n=%d' % n, 'mean=%.4g' % m, '[%s]' % short)
