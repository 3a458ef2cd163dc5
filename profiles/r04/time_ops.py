#!/usr/bin/env python3
"""Per-kernel times of ONE stage on the benchmark batch, for ablation builds whose wrong results must not reach the rest of the
chain (a front end that skips its median would hand the region passes a noise image):

    PCSEG_LIB=ab/<name>/libpcseg.so python profiles/r04/time_ops.py classmap|refine|locmax|fill [reps]

classmap: ops.classmap_label on the 64 x 1024^2 x 5 batch; refine: EDT, local maxima and the watershed (mode 2: frames the proof
fails on are reported, not recomputed); fill: the two particle fills on the denoised class map.  Prints the library's event table (us per launch) of the timed repetitions."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from particle_col_image_segmentation_amd import _lib, ops, synth


def table(lib, fn, reps):
    fn()
    torch.cuda.synchronize()
    lib.pcseg_timing_enable(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    n = lib.pcseg_timing_report(None, 0)
    buf = ctypes.create_string_buffer(n + 16)
    lib.pcseg_timing_report(buf, n + 16)
    lib.pcseg_timing_enable(0)
    rows = []
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.split("\t")
        rows.append((float(ms), int(calls), name))
    for ms, calls, name in sorted(rows, reverse=True):
        print("%10.3f ms %6d launches %9.2f us/launch  %s" % (ms, calls, 1e3 * ms / calls, name))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "classmap"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda:0")
    lib = _lib.load()
    stack = synth.gen_batch_torch(10_000, 64, 1024, 1024, dev)
    if what == "classmap":
        table(lib, lambda: ops.classmap_label(stack), reps)
    elif what == "refine":
        bm = stack[:, synth.BOUNDARY_PLANE]

        def fn():
            d2, mask = ops.edt_sq_lt(bm, 0.5)
            _, markers, _ = ops.local_maxima(d2, want_mask=False)
            ops.watershed(bm, markers, mask, mode=2)

        table(lib, fn, reps)
    elif what == "locmax":
        d2, _ = ops.edt_sq_lt(stack[:, synth.BOUNDARY_PLANE], 0.5)
        table(lib, lambda: ops.local_maxima(d2, want_mask=False), reps)
    elif what == "fill":
        ds, _, _ = ops.classmap_label(stack)

        def fn():
            out = ds
            for v in (1, 2):  # the two cell classes of synth.CELL_TYPES_5, particle = 3 (tiff_analysis.py:982-1015)
                out, _ = ops.fill_particle(out, 3, v, 3, 20, 2)

        table(lib, fn, reps)
    else:
        raise SystemExit("classmap | refine | locmax | fill")


if __name__ == "__main__":
    main()
