#!/usr/bin/env python3
"""Round-3 diagnostic (GPU box): how much work does the watershed's second level see on the benchmark batch?
Reads the level-1 frame flags, the level-2 flags and the active 64x64 tiles straight out of the call's workspace
(layout of pcseg_watershed4_f32's Carver, csrc/watershed.hip)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from particle_col_image_segmentation_amd import _lib, ops, synth


def al(x, a=256):
    return (x + a - 1) // a * a


def main():
    dev = torch.device("cuda", 0)
    B, H, W = 64, 1024, 1024
    stack = synth.gen_batch_torch(10_000, B, H, W, dev)
    for levels in (0, 100):
        if levels:
            stack[:, 3] = torch.round(stack[:, 3] * levels) / levels
        bm = stack[:, 3]
        d2, mask = ops.edt_sq_lt(bm, 0.5)
        _, markers, n_markers = ops.local_maxima(d2, want_mask=False)
        lib = _lib.load()
        img, fstride = ops._plane_view(bm)
        out = torch.empty((B, H, W), dtype=torch.int32, device=dev)
        flags_out = torch.zeros((B,), dtype=torch.int32, device=dev)
        nbytes = lib.pcseg_watershed_workspace_bytes(B, H, W)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.pcseg_watershed4_f32(ops._ptr(img), fstride, ops._ptr(markers), ops._ptr(mask), ops._ptr(out), ops._ptr(flags_out),
                                            B, H, W, 2, ops._ptr(ws), nbytes, ops._stream()), "watershed")
        torch.cuda.synchronize()
        n = B * H * W
        tx = ty = (W + 63) // 64
        ntiles = B * tx * ty
        ntiles_max = B * (tx + 1) * (ty + 1)
        off = 0
        off += al(n * 4) * 2            # val, L
        off += al(ntiles_max) * 2       # dirtyA, dirtyB
        o_active = off
        off += al(ntiles)
        off += al(4 * (32 + 16 * 32))   # changed
        o_flags = off
        off += al(4 * B)
        o_flags2 = off
        h = ws.cpu().numpy()
        flags = h[o_flags:o_flags + 4 * B].view(np.int32)
        flags2 = h[o_flags2:o_flags2 + 4 * B].view(np.int32)
        active = h[o_active:o_active + ntiles].reshape(B, ty, tx)
        per = active.reshape(B, -1).sum(1)
        print("levels=%d: level-1 flagged frames %d / %d, level-2 flagged %d (tie_flags %d)" %
              (levels, int((flags != 0).sum()), B, int((flags2 != 0).sum()), int(flags_out.sum())))
        print("  active tiles: total %d of %d; per flagged frame min %d median %d max %d" %
              (int(per.sum()), ntiles, int(per[flags != 0].min()) if (flags != 0).any() else 0,
               int(np.median(per[flags != 0])) if (flags != 0).any() else 0, int(per.max())))
        print("  markers per frame: mean %.0f; labelled px mean %.0f" % (float(n_markers.float().mean()), float((markers != 0).sum()) / B))


if __name__ == "__main__":
    main()
