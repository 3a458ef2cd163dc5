#!/usr/bin/env python3
"""Per-launch durations of ws_relax_kernel inside one watershed call, from a rocprofv3 --kernel-trace CSV:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline
    python profiles/relax_rounds.py gpurun_out/trace

prints, for the last step, the duration of every relaxation round in launch order."""
import csv
import glob
import os
import sys


def main(root):
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under %s" % root)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    relax = [(s, e) for s, e, n in rows if "ws_relax_kernel" in n]
    if not relax:
        raise SystemExit("no ws_relax_kernel dispatches found")
    # split into watershed calls: a gap of more than 1 ms between two relax launches starts a new call
    calls, cur = [], [relax[0]]
    for a, b in zip(relax, relax[1:]):
        if b[0] - a[1] > 1_000_000:
            calls.append(cur)
            cur = []
        cur.append(b)
    calls.append(cur)
    last = calls[-1]
    print("calls: %d, launches in the last call: %d" % (len(calls), len(last)))
    for i, (s, e) in enumerate(last):
        print("round %2d  %9.1f us" % (i, (e - s) / 1e3))
    print("total     %9.1f us" % (sum(e - s for s, e in last) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace")
