#!/usr/bin/env python3
"""Sum the counters of profiles/pmc_sq.sh per kernel (and per dispatch for the kernels matching the pattern)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main(root, pat):
    per_kernel = defaultdict(lambda: defaultdict(float))
    per_dispatch = defaultdict(lambda: defaultdict(float))
    order = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("pcseg::", "").replace("void ", "")
                c, v = r["Counter_Name"], float(r["Counter_Value"])
                per_kernel[name][c] += v
                if re.search(pat, name):
                    # dispatch ids differ between passes; the launch ORDER of the matching kernels does not
                    key = (f, int(r["Dispatch_Id"]))
                    per_dispatch[key][c] += v
                    order[key] = name
    counters = sorted({c for d in per_kernel.values() for c in d})
    print("kernel," + ",".join(counters))
    for k, d in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        print(k[:60] + "," + ",".join("%.4g" % d.get(c, 0) for c in counters))
    print()
    by_file = defaultdict(list)
    for (f, did), d in per_dispatch.items():
        by_file[f].append((did, d))
    for f, lst in by_file.items():
        lst.sort()
        cs = sorted({c for _, d in lst for c in d})
        print("# dispatches matching %s in %s" % (pat, os.path.relpath(f, root)))
        print("n," + ",".join(cs))
        for i, (_, d) in enumerate(lst[-14:]):
            print("%d," % i + ",".join("%.4g" % d.get(c, 0) for c in cs))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "ws_relax_kernel")
