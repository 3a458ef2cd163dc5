#!/usr/bin/env python3
"""GPU idle gaps inside the timed region of a bench.py run traced with `rocprofv3 --kernel-trace`:
    python profiles/idle_gaps.py <dir with */*_kernel_trace.csv> [min_gap_us]
Prints, for the last complete step, the wall time, the time with at least one kernel running, and the gaps."""
import csv
import glob
import os
import sys


def main(src, min_gap=15.0):
    trace = glob.glob(os.path.join(src, "*", "*_kernel_trace.csv"))[0]
    rows = []
    for r in csv.DictReader(open(trace)):
        if "pcseg::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("pcseg::", 1)[1].split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    # steps start with argmax_kernel
    starts = [i for i, r in enumerate(rows) if r[2].startswith("argmax_kernel")]
    if len(starts) < 3:
        raise SystemExit("need at least 3 steps in the trace")
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    t0, t1 = step[0][0], rows[b][0]
    busy = 0
    cur_end = t0
    gaps = []
    last_name = None
    for s, e, n in step:
        if s > cur_end:
            gaps.append(((s - cur_end) / 1e3, (cur_end - t0) / 1e3, last_name, n))
        else:
            pass
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
            last_name = n
    print("step wall %.1f us, busy %.1f us, idle %.1f us over %d kernels" % ((t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, len(step)))
    tot = 0.0
    for g, at, prev, nxt in gaps:
        if g >= min_gap:
            tot += g
            print("  gap %7.1f us at +%8.1f us  after %-40s before %s" % (g, at, prev, nxt))
    print("gaps >= %.0f us: %.1f us" % (min_gap, tot))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 15.0)
