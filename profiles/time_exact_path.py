#!/usr/bin/env python3
"""Times the watershed's exact (heap-emulation) path alone: mode 1 on synthetic 1024x1024 boundary maps.
Usage: python profiles/time_exact_path.py [frames] [size]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from particle_col_image_segmentation_amd import ops, synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
stack = synth.gen_batch_torch(4242, frames, size, size, dev)
bm = stack[:, 3].contiguous()
mask = ops.threshold_lt(bm, 0.5)
d2 = ops.edt_sq(mask)
_, markers, counts = ops.local_maxima(d2)
for mode, name in ((0, "auto"), (1, "exact only")):
    ops.watershed(bm, markers, mask, mode=mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, flags = ops.watershed(bm, markers, mask, mode=mode)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-10s %d frames %dx%d: %.3f s (%.3f s/frame if serial, %.2f Mpx/s, %d seeds/frame, flagged %d)" % (
        name, frames, size, size, dt, dt, frames * size * size / dt / 1e6, int(counts.float().mean()), int(flags.sum())), flush=True)
