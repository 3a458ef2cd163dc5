"""Frame-by-frame parity of FramePipeline results against the CPU oracle, in worker processes.

TEST INFRASTRUCTURE ONLY (like everything under ``oracle/``): imported by ``tests/``, by
``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg, which times the oracle on frames of the
batch it has just benchmarked and compares what the GPU produced for exactly those frames.

A worker runs ``oracle.segment_frame`` on one frame (a ``.npy`` file, so that a 20 MB stack is not pickled through a
pipe) and returns digests of the integer images plus the small float tables; the parent digests the GPU arrays the
same way.  Equal digests <=> bit-exact images.
"""
import hashlib
import os
import tempfile
import time

import numpy as np

IMAGE_KEYS = ("denoised", "labels", "recreated", "ws_labels")


def _digest(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return hashlib.blake2b(a.tobytes(), digest_size=16).hexdigest() + ":%s" % (a.shape,)


def oracle_frame_job(job):
    """job = (path of the (C,H,W) float32 stack, cell_types dict, merged) -> dict (picklable, small)."""
    path, cell_types, merged = job
    from oracle import oracle as orc
    stack = np.load(path)
    t0 = time.perf_counter()
    nan = False
    try:
        ref = orc.segment_frame(stack, cell_types, merged=merged)
    except ValueError:
        # the reference's int(NaN) crash (tiff_analysis.py:776-781): everything that does not depend on the
        # cluster statistic is still defined -- recompute those parts directly
        nan = True
        cm = (np.argmax(stack, axis=0) + 1).astype(np.uint8)
        den = orc.median_filter(cm)
        lab = orc.label(den)
        rf = orc.refine_boundaries(stack[3])
        ref = {"denoised": den, "label_im": lab, "recreated": orc.recreate_particle_area(den, cell_types, 0)[0], "refine": rf,
               "roi_sums": orc.channel_sums(rf["labels"], stack)}
    seconds = time.perf_counter() - t0
    out = describe(ref, cell_types, merged and not nan, nan)
    out["seconds"] = seconds
    return out


def describe(ref, cell_types, merged=True, nan=False):
    """What ``compare`` / ``compare_tables`` need of one ``oracle.segment_frame`` result: digests of the integer images,
    the small float tables, the classification vectors and (``merged``) the merged groups."""
    return {"nan": nan,
            "classes": None if nan else _classification(ref, cell_types),
            "groups": _groups(ref, cell_types) if (merged and not nan) else None,
            "denoised": _digest(ref["denoised"], np.uint8),
            "labels": _digest(ref["label_im"], np.int32),
            "recreated": None if ref["recreated"] is None else _digest(ref["recreated"], np.uint8),
            "ws_labels": _digest(ref["refine"]["labels"], np.int32),
            "n_labels": int(ref["label_im"].max()),
            "n_markers": int(ref["refine"]["markers"].max()),
            "roi_sums": np.asarray(ref["roi_sums"], np.float64),
            "roi_area": np.bincount(ref["refine"]["labels"].ravel(), minlength=int(ref["refine"]["markers"].max()) + 1)[1:]}


def slot_names(cell_types):
    """Cell-type names in the order FramePipeline numbers its type slots (first appearance in ``cell_types``)."""
    from oracle import oracle as orc
    names = []
    for name in cell_types.values():
        if name in orc.CELL_TYPES and name not in names:
            names.append(name)
    return names


def _classification(ref, cell_types):
    """Per class-map component (label l -> index l - 1) what the reference's region loop decides
    (tiff_analysis.py:754-781): kind (0 none, 1 cell, 2 cluster), cells (1 / cluster.cells / 0) and the type slot of
    the component's class (255: not a cell type)."""
    from oracle import oracle as orc
    names = slot_names(cell_types)
    lab, den = ref["label_im"], ref["denoised"]
    n = int(lab.max())
    tab = orc.region_table(lab, n)
    kind = np.zeros(n, np.uint8)
    cells = np.zeros(n, np.int32)
    slot_of = np.full(n, 255, np.uint8)
    first = tab[:, 7]
    cls = den.ravel()[first]
    for v, t in cell_types.items():
        if t in names:
            slot_of[cls == v] = names.index(t)
    for name in names:
        for r in ref["cell_pos"].get(name, []):
            kind[r.label - 1] = 1
            cells[r.label - 1] = 1
        for r in ref["cell_clusters"].get(name, []):
            kind[r.label - 1] = 2
            cells[r.label - 1] = r.cells
    return {"kind": kind, "cells": cells, "slot_of": slot_of}


def _groups(ref, cell_types):
    """merged_clusters of get_cell_positions_and_areas(merged=True) (tiff_analysis.py:843-878) as plain arrays per
    slot (type slots 0..3, 4 = "combined"): member labels (flattened, groups in order) + offsets, area, bbox, centroid."""
    names = slot_names(cell_types)
    out = {}
    for key, groups in ref["merged_clusters"].items():
        s = 4 if key == "combined" else names.index(key)
        members, offsets = [], [0]
        for g in groups:
            members += [r.label for r in g["regions"]]
            offsets.append(len(members))
        out[s] = {"members": np.array(members, np.int32), "offsets": np.array(offsets, np.int64),
                  "area": np.array([g["area"] for g in groups], np.int64),
                  "bbox": np.array([g["bbox"] for g in groups], np.int64).reshape(len(groups), 4),
                  "centroid": np.array([g["centroid"] for g in groups], np.float64).reshape(len(groups), 2)}
    return out


def run_oracle(stacks, cell_types, merged=True, processes=None):
    """Oracle results for the frames of ``stacks`` ((n,C,H,W) float32 numpy), one frame per worker process.
    Returns (list of per-frame dicts, wall seconds of the pool run, processes used)."""
    import multiprocessing as mp
    from oracle import oracle as orc
    orc.build()
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    procs = max(1, min(processes or cores, len(stacks)))
    with tempfile.TemporaryDirectory(prefix="pcseg_parity_") as tmp:
        jobs = []
        for i, st in enumerate(stacks):
            p = os.path.join(tmp, "f%04d.npy" % i)
            np.save(p, np.ascontiguousarray(st, dtype=np.float32))
            jobs.append((p, dict(cell_types), merged))
        ctx = mp.get_context("spawn")  # the parent may hold a HIP context: never fork it
        with ctx.Pool(procs) as pool:
            pool.map(_noop, range(procs))  # start-up (interpreter + numpy import) is not the oracle's time
            t0 = time.perf_counter()
            out = pool.map(oracle_frame_job, jobs, chunksize=1)
            wall = time.perf_counter() - t0
    return out, wall, procs


def _noop(i):
    import numpy  # noqa: F401
    from oracle import oracle as orc
    orc.lib()
    return i


def compare(res, frame_indices, refs, sums_rtol=1e-6):
    """``res``: a FramePipeline result (device tensors), ``refs[k]`` the oracle dict of batch frame
    ``frame_indices[k]``.  Raises AssertionError naming the first difference; returns the number of frames checked."""
    for b, ref in zip(frame_indices, refs):
        got = {"denoised": (res["denoised"][b], np.uint8), "labels": (res["labels"][b], np.int32),
               "recreated": (res["recreated"][b], np.uint8), "ws_labels": (res["ws_labels"][b], np.int32)}
        for key in IMAGE_KEYS:
            if ref[key] is None:
                continue
            t, dt = got[key]
            d = _digest(t.cpu().numpy(), dt)
            assert d == ref[key], "frame %d: %s differs from the oracle (%s vs %s)" % (b, key, d, ref[key])
        assert int(res["counts"][b]) == ref["n_labels"], "frame %d: label count" % b
        m = ref["n_markers"]
        assert int(res["n_markers"][b]) == m, "frame %d: marker count" % b
        assert int(res["nan_flag"][b]) == int(ref["nan"]), "frame %d: int(NaN) flag" % b
        np.testing.assert_array_equal(res["ws_stats"][b, :m, 0].cpu().numpy(), ref["roi_area"],
                                      err_msg="frame %d: ROI areas" % b)
        np.testing.assert_allclose(res["ws_sums"][b, :m].cpu().numpy(), ref["roi_sums"], rtol=sums_rtol, atol=0,
                                   err_msg="frame %d: ROI plane sums" % b)
        if ref.get("classes") is not None:
            n = ref["n_labels"]
            for key in ("kind", "cells", "slot_of"):
                np.testing.assert_array_equal(res[key][b, :n].cpu().numpy(), ref["classes"][key],
                                              err_msg="frame %d: %s per class-map component" % (b, key))
        if ref.get("groups") is not None and res.get("groups"):
            _compare_groups(res, b, ref["groups"])
    return len(refs)


def _compare_groups(res, b, ref_groups):
    """merged groups of frame ``b`` (per type slot and combined): member label lists in the reference's order, area,
    bbox bit-exact; area-weighted centroid within 1e-12 relative (tiff_analysis.py:843-878)."""
    for s, g in res["groups"].items():
        exp = ref_groups.get(s)
        ng = int(g["n_groups"][b])
        if exp is None:
            assert ng == 0, "frame %d slot %d: %d groups, the oracle has none" % (b, s, ng)
            continue
        assert ng == len(exp["area"]), "frame %d slot %d: %d groups, oracle %d" % (b, s, ng, len(exp["area"]))
        k = int(res["n_list"][b, s])
        lst = res["region_list"][b, s, :k].cpu().numpy()
        gof = g["group_of"][b, :k].cpu().numpy()
        gst = g["group_stats"][b, :ng].cpu().numpy()
        # members of group i in list order == the oracle's "regions" of group i
        order = np.argsort(gof, kind="stable")
        order = order[gof[order] > 0]
        np.testing.assert_array_equal(lst[order] + 1, exp["members"], err_msg="frame %d slot %d: group members" % (b, s))
        np.testing.assert_array_equal(np.bincount(gof[gof > 0] - 1, minlength=ng), np.diff(exp["offsets"]),
                                      err_msg="frame %d slot %d: group sizes" % (b, s))
        np.testing.assert_array_equal(gst[:, 0], exp["area"], err_msg="frame %d slot %d: group area" % (b, s))
        np.testing.assert_array_equal(gst[:, 3:7], exp["bbox"], err_msg="frame %d slot %d: group bbox" % (b, s))
        np.testing.assert_array_equal(gst[:, 7], np.diff(exp["offsets"]), err_msg="frame %d slot %d: member counts" % (b, s))
        if ng:
            cen = gst[:, 1:3].astype(np.float64) / gst[:, 0:1].astype(np.float64)
            np.testing.assert_allclose(cen, exp["centroid"], rtol=1e-12, atol=0, err_msg="frame %d slot %d: centroid" % (b, s))


def compare_tables(tabs, frame_ids, refs):
    """The dense output tables of ``FramePipeline.tables`` (``groups`` rows and the ``group`` / ``group_combined``
    columns of ``cells``) against the oracle's merged groups of the same frames.  ``frame_ids[k]`` is the id column
    value of the frame ``refs[k]`` describes.  Returns the number of frames checked (frames on which the reference
    raises int(NaN) have no merged groups and are skipped)."""
    groups, cells = np.asarray(tabs["groups"]), np.asarray(tabs["cells"])
    checked = 0
    for fid, ref in zip(frame_ids, refs):
        if ref.get("groups") is None:
            continue
        checked += 1
        grow = groups[groups[:, 0] == fid]
        crow = cells[cells[:, 0] == fid]
        exp_rows = []
        own = {}   # label -> group id inside its own type
        comb = {}  # label -> group id in "combined"
        for s in sorted(ref["groups"]):
            e = ref["groups"][s]
            for gi in range(len(e["area"])):
                mem = e["members"][e["offsets"][gi]:e["offsets"][gi + 1]]
                exp_rows.append([fid, s, gi + 1, e["area"][gi], e["centroid"][gi, 0], e["centroid"][gi, 1],
                                 e["bbox"][gi, 0], e["bbox"][gi, 1], e["bbox"][gi, 2], e["bbox"][gi, 3], len(mem)])
                for l in mem:
                    (comb if s == 4 else own)[int(l)] = gi + 1
        exp_rows = np.array(exp_rows, np.float64).reshape(len(exp_rows), 11)
        assert grow.shape == exp_rows.shape, "frame %s: %d group rows, oracle %d" % (fid, grow.shape[0], exp_rows.shape[0])
        np.testing.assert_array_equal(grow[:, [0, 1, 2, 3, 6, 7, 8, 9, 10]], exp_rows[:, [0, 1, 2, 3, 6, 7, 8, 9, 10]],
                                      err_msg="frame %s: groups table" % fid)
        np.testing.assert_allclose(grow[:, 4:6], exp_rows[:, 4:6], rtol=1e-12, atol=0, err_msg="frame %s: group centroids" % fid)
        labels = crow[:, 1].astype(np.int64)
        np.testing.assert_array_equal(crow[:, 12], [own.get(int(l), 0) for l in labels], err_msg="frame %s: cells.group" % fid)
        np.testing.assert_array_equal(crow[:, 13], [comb.get(int(l), 0) for l in labels],
                                      err_msg="frame %s: cells.group_combined" % fid)
    return checked
