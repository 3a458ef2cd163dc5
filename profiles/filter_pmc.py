#!/usr/bin/env python3
"""Filter a rocprofv3 counter_collection CSV down to the dispatches of kernels whose name contains a pattern:
    python profiles/filter_pmc.py <counter_collection.csv> <pattern> <out.csv>
one row per dispatch: dispatch id, kernel (short), grid, duration_us, then one column per counter."""
import collections
import csv
import sys

src, pat, dst = sys.argv[1:4]
disp = collections.OrderedDict()
for r in csv.DictReader(open(src)):
    if pat not in r["Kernel_Name"]:
        continue
    k = int(r["Dispatch_Id"])
    d = disp.setdefault(k, {"kernel": r["Kernel_Name"].split("pcseg::")[-1].split("(")[0], "grid": r["Grid_Size"],
                            "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for d in disp.values() for c in d} - {"kernel", "grid", "dur_us"})
with open(dst, "w") as f:
    w = csv.writer(f)
    w.writerow(["dispatch", "kernel", "grid", "dur_us"] + names)
    for k, d in disp.items():
        w.writerow([k, d["kernel"], d["grid"], round(d["dur_us"], 1)] + [d.get(c, "") for c in names])
