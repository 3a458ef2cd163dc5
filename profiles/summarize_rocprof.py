#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_rNN/{trace,fetch,write}) into the small, tracked summaries under
profiles/: per-kernel duration stats of the pcseg kernels, per-kernel HBM traffic from the PMC passes, and
profiles/traffic.json (bytes per launch, which bench.py reports as roofline.traffic).

PMC units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE reads exactly half of the bytes of a wide coalesced stream, so it is doubled
(checked here on argmax_kernel: 5 planes x 64 frames x 4 MiB = 1.34 GB algorithmic vs 0.67 GB raw FETCH_SIZE);
WRITE_SIZE is taken as is.  The two counters come from separate passes.

    python profiles/summarize_rocprof.py <dir with trace/ fetch/ write/> r01 [output dir]
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    n = name.split("pcseg::", 1)[1]
    depth = 0
    for i, ch in enumerate(n):  # cut the argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i]
    return n


def main(src, tag, here=None):
    here = here or os.path.dirname(os.path.abspath(__file__))
    os.makedirs(here, exist_ok=True)
    trace = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True))[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "pcseg::" in r["Kernel_Name"]:
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in dur.values())
    with open(os.path.join(here, "%s_kernel_stats.csv" % tag), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_pcseg_time"])
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(sum(v), 1), round(sum(v) / len(v), 2), round(min(v), 2), round(max(v), 2),
                        round(100 * sum(v) / total, 2)])
    pmc = {}
    for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        files = sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True))
        if not files:
            continue
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(files[0])):
            if "pcseg::" in r["Kernel_Name"] and r["Counter_Name"] == cname:
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        pmc[cname] = agg
    traffic = {}
    if pmc:
        with open(os.path.join(here, "%s_pmc_hbm.csv" % tag), "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_per_launch_raw", "WRITE_SIZE_KiB_per_launch",
                        "hbm_bytes_per_launch_corrected(2*FETCH+WRITE)*1024"])
            names = sorted(set(pmc.get("FETCH_SIZE", {})) | set(pmc.get("WRITE_SIZE", {})))
            for k in names:
                fc, fv = pmc.get("FETCH_SIZE", {}).get(k, [0, 0.0])
                wc, wv = pmc.get("WRITE_SIZE", {}).get(k, [0, 0.0])
                fpl = fv / fc if fc else 0.0
                wpl = wv / wc if wc else 0.0
                b = (2 * fpl + wpl) * 1024
                w.writerow([k, max(fc, wc), round(fpl, 1), round(wpl, 1), int(b)])
                traffic[k.split("<")[0]] = traffic.get(k.split("<")[0], 0) + 0  # placeholder to keep key order
            # bench.py keys by bare kernel name; for templated kernels keep the instantiation that moves the most bytes in all
            best = {}
            for k in names:
                base = k.split("<")[0]
                fc, fv = pmc.get("FETCH_SIZE", {}).get(k, [0, 0.0])
                wc, wv = pmc.get("WRITE_SIZE", {}).get(k, [0, 0.0])
                per_launch = int((2 * (fv / fc if fc else 0) + (wv / wc if wc else 0)) * 1024)
                if per_launch * max(fc, wc) >= best.get(base, (0, 0))[0]:
                    best[base] = (per_launch * max(fc, wc), per_launch)
            traffic = {k: v[1] for k, v in best.items()}
        json.dump(traffic, open(os.path.join(here, "traffic.json"), "w"), indent=1, sort_keys=True)
        # whole chain per step: every kernel's launches x bytes; the number of steps in the PMC run = launches of the
        # front-end kernel (one per step)
        steps = 0
        for k in names:
            if k.startswith("classmap_median_ccl_kernel") or k.startswith("argmax_kernel"):
                steps = max(steps, pmc.get("FETCH_SIZE", {}).get(k, [0, 0.0])[0])
        if steps:
            fetch = sum(v[1] for v in pmc.get("FETCH_SIZE", {}).values()) * 1024 * 2 / steps
            write = sum(v[1] for v in pmc.get("WRITE_SIZE", {}).values()) * 1024 / steps
            json.dump({"steps_in_pmc_run": steps, "fetch_bytes_per_step_corrected": int(fetch), "write_bytes_per_step": int(write),
                       "hbm_bytes_per_step": int(fetch + write)},
                      open(os.path.join(here, "%s_chain_traffic.json" % tag), "w"), indent=1)
    print("wrote", tag, "summaries:", len(dur), "kernels,", len(traffic), "traffic entries")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
