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
    # durations of the same dispatches (the --kernel-trace CSV of the FIRST counter group: every group runs the same command)
    dur_ns, launches = defaultdict(float), defaultdict(int)
    traces = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
    if traces:
        first_dir = os.path.dirname(traces[0])
        for f in traces:
            if os.path.dirname(f) != first_dir:
                continue
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("pcseg::", "").replace("void ", "")
                    dur_ns[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    launches[name] += 1
    counters = sorted({c for d in per_kernel.values() for c in d})
    # valu_busy = SQ_ACTIVE_INST_VALU (quad-cycles: 4 cycles of one SIMD each) against the SIMD-cycles of the kernel's own
    # durations: 1 024 SIMDs x 2.4 cycles per ns (the clock under the profiler is not pinned: an estimate)
    print("kernel,launches,dur_ms,valu_busy,lds_busy," + ",".join(counters))
    tot = defaultdict(float)
    for k, d in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        simd_cycles = dur_ns.get(k, 0.0) * 2.4 * 1024
        vb = 4.0 * d.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles if simd_cycles else 0.0
        lb = 4.0 * d.get("SQ_ACTIVE_INST_LDS", 0) / simd_cycles if simd_cycles else 0.0
        print("%s,%d,%.3f,%.3f,%.3f," % (k[:60], launches.get(k, 0), dur_ns.get(k, 0.0) / 1e6, vb, lb) +
              ",".join("%.4g" % d.get(c, 0) for c in counters))
        if not k.startswith("at::") and "rocclr" not in k:
            tot["dur"] += dur_ns.get(k, 0.0)
            tot["valu"] += d.get("SQ_ACTIVE_INST_VALU", 0)
            tot["lds"] += d.get("SQ_ACTIVE_INST_LDS", 0)
    if tot["dur"]:
        print("# library kernels together: %.3f ms of kernel time, VALU busy %.3f, LDS busy %.3f of their SIMD-cycles"
              % (tot["dur"] / 1e6, 4.0 * tot["valu"] / (tot["dur"] * 2.4 * 1024), 4.0 * tot["lds"] / (tot["dur"] * 2.4 * 1024)))
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
