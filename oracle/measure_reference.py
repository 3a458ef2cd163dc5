#!/usr/bin/env python3
"""Time the REAL reference (scikit-image / scipy under /opt/conda/bin/python3.9) against the CPU oracle on the same
synthetic frames, in the build container (the reference cannot travel to the GPU box).  Writes
oracle/reference_ratio.json, which bench.py's cpu_baseline leg quotes next to the oracle's throughput so that a
reader can map the 'port' baseline back to the Python reference.

    python oracle/measure_reference.py            # system python: runs the oracle part, spawns the reference part

TEST / MEASUREMENT INFRASTRUCTURE ONLY; never imported by the product."""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_PY = "/opt/conda/bin/python3.9"
SEEDS = (900, 901)
SIZE = 1024


def _frames():
    import importlib.util
    spec = importlib.util.spec_from_file_location("pcseg_synth", os.path.join(ROOT, "particle_col_image_segmentation_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    return synth, [synth.gen_frame(s, SIZE, SIZE) for s in SEEDS]


def reference_part():
    """runs under the conda interpreter: the reference's own functions plus the four library calls of
    refine_boundaries.py:60-73 (a script, not importable)"""
    import warnings
    warnings.filterwarnings("ignore")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, "/root/reference")
    import numpy as np
    from scipy import ndimage as ndi
    from scipy.ndimage import median_filter
    from skimage import measure, morphology
    from skimage.segmentation import watershed
    import tiff_analysis as ta
    synth, frames = _frames()
    ct = dict(synth.CELL_TYPES_5)
    out = {"no_merge_s": [], "merge_s": [], "merge_regions": [], "refine_s": []}
    for st in frames:
        cm = (np.argmax(st, axis=0) + 1).astype(np.uint8)
        t0 = time.perf_counter()
        den = median_filter(cm, size=5)
        cell_pos, cell_clusters, pa, _ = ta.get_cell_positions_and_areas(den, ct, False)
        ta.get_cell_counts_and_densities(cell_pos, cell_clusters, pa)
        ta.recreate_particle_area(den, ct, pa)
        t1 = time.perf_counter()
        mask = st[3] < 0.5
        dist = ndi.distance_transform_edt(mask)
        lm = morphology.local_maxima(dist)
        mk = measure.label(lm)
        watershed(st[3], mk, mask=mask)
        t2 = time.perf_counter()
        out["no_merge_s"].append(t1 - t0 + (t2 - t1))
        out["refine_s"].append(t2 - t1)
    # the O(R^2) merge as written (tiff_analysis.py:850-852) on ONE frame: minutes
    st = frames[0]
    den = median_filter((np.argmax(st, axis=0) + 1).astype(np.uint8), size=5)
    cell_pos, cell_clusters, pa, _ = ta.get_cell_positions_and_areas(den, ct, False)
    t0 = time.perf_counter()
    merged, _ = ta.get_cell_clusters_from_distances(den, cell_pos, cell_clusters, ct)
    out["merge_s"].append(time.perf_counter() - t0)
    out["merge_regions"].append(sum(len(v) for v in cell_pos.values()) + sum(len(v) for v in cell_clusters.values()))
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--reference-part":
        return reference_part()
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    synth, frames = _frames()
    ct = dict(synth.CELL_TYPES_5)
    orc.build()
    o_nomerge, o_merge = [], []
    for st in frames:
        t0 = time.perf_counter()
        try:
            orc.segment_frame(st, ct, merged=False)
        except ValueError:
            pass
        o_nomerge.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        try:
            orc.segment_frame(st, ct, merged=True)
        except ValueError:
            pass
        o_merge.append(time.perf_counter() - t0)
    env = dict(os.environ, MPLBACKEND="Agg")
    ref = json.loads(subprocess.check_output([REF_PY, "-B", os.path.abspath(__file__), "--reference-part"], env=env, cwd="/tmp").decode().strip().splitlines()[-1])
    mean = lambda v: sum(v) / len(v)
    rec = {
        "where": "build container, 1 core, %d synthetic %dx%dx5 frames (seeds %s) of particle_col_image_segmentation_amd/synth.py" % (len(SEEDS), SIZE, SIZE, list(SEEDS)),
        "reference_versions": "scikit-image 0.18.3, scipy 1.7.1, numpy 1.26.4 under /opt/conda/bin/python3.9",
        "reference_chain_no_merge_s_per_frame": round(mean(ref["no_merge_s"]), 3),
        "reference_refine_s_per_frame": round(mean(ref["refine_s"]), 3),
        "reference_merge_as_written_s": round(ref["merge_s"][0], 1),
        "reference_merge_as_written_regions": ref["merge_regions"][0],
        "oracle_chain_no_merge_s_per_frame": round(mean(o_nomerge), 3),
        "oracle_chain_with_OR_merge_s_per_frame": round(mean(o_merge), 3),
    }
    rec["reference_over_oracle_no_merge"] = round(rec["reference_chain_no_merge_s_per_frame"] / rec["oracle_chain_no_merge_s_per_frame"], 2)
    rec["reference_with_merge_over_oracle_with_merge"] = round(
        (rec["reference_chain_no_merge_s_per_frame"] + rec["reference_merge_as_written_s"]) / rec["oracle_chain_with_OR_merge_s_per_frame"], 1)
    with open(os.path.join(HERE, "reference_ratio.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
