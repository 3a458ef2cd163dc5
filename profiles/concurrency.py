#!/usr/bin/env python3
"""How many pcseg kernels run at the same time in the steady state of a traced bench.py run:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py ...
    python profiles/concurrency.py DIR
Takes the middle half of the traced span, prints the share of time at each concurrency level (0 = GPU idle) and, for
the time with exactly one kernel running, which kernels those were."""
import collections
import csv
import glob
import os
import sys


def main(src):
    trace = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(trace)):
        if "pcseg::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("pcseg::", 1)[1].split("(")[0][:40]))
    rows.sort()
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    a, b = t_lo + (t_hi - t_lo) // 4, t_hi - (t_hi - t_lo) // 4
    events = []
    for s, e, n in rows:
        s2, e2 = max(s, a), min(e, b)
        if e2 > s2:
            events.append((s2, 1, n))
            events.append((e2, -1, n))
    events.sort()
    level = collections.Counter()
    alone = collections.Counter()
    running = collections.Counter()
    cur, last = 0, a
    for t, d, n in events:
        level[cur] += t - last
        if cur == 1:
            alone[next(k for k, v in running.items() if v > 0)] += t - last
        last = t
        cur += d
        running[n] += d
    level[cur] += b - last
    total = b - a
    print("steady-state window %.1f ms" % (total / 1e6))
    for k in sorted(level):
        print("  %2d kernels running: %5.1f %%" % (k, 100.0 * level[k] / total))
    print("average concurrency %.2f" % (sum(k * v for k, v in level.items()) / total))
    print("running alone (share of the window):")
    for n, v in alone.most_common(8):
        print("  %-42s %5.1f %%" % (n, 100.0 * v / total))


if __name__ == "__main__":
    main(sys.argv[1])
