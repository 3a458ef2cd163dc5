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
    out = {"seconds": seconds, "nan": nan,
           "denoised": _digest(ref["denoised"], np.uint8),
           "labels": _digest(ref["label_im"], np.int32),
           "recreated": None if ref["recreated"] is None else _digest(ref["recreated"], np.uint8),
           "ws_labels": _digest(ref["refine"]["labels"], np.int32),
           "n_labels": int(ref["label_im"].max()),
           "n_markers": int(ref["refine"]["markers"].max()),
           "roi_sums": np.asarray(ref["roi_sums"], np.float64),
           "roi_area": np.bincount(ref["refine"]["labels"].ravel(), minlength=int(ref["refine"]["markers"].max()) + 1)[1:]}
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
    return len(refs)
