"""Boundary refinement: the reference's ``refine_boundaries.py`` as a function.

The reference file is an unguarded script (refine_boundaries.py:27-79): it loads
an ilastik probabilities ``.h5``, takes channel 3 as the boundary map,
thresholds at 0.5, and runs EDT -> local maxima -> label -> marker-controlled
watershed.  Here the same chain is a function over the HIP kernels, plus a
script-compatible ``__main__``.
"""
import numpy as np
import torch

from . import ops
from .tiff_analysis import _device

DEFAULT_FILE = "working_folder/Tp_C3M10_1_120h_60X_RFP_GFP_1_MIP_probabilities.h5"  # refine_boundaries.py:28
BOUNDARY_CHANNEL = 3  # refine_boundaries.py:34
THRESHOLD = 0.5  # refine_boundaries.py:44


def refine_boundaries_batch(boundary_maps, threshold=THRESHOLD, mode=0):
    """(B,H,W) float32 CUDA tensor -> dict of device stages (refine_boundaries.py:44-73)."""
    d2, mask = ops.edt_sq_lt(boundary_maps, threshold)          # :44-45, :60
    local_max, markers, n_markers = ops.local_maxima(d2)        # :63-64
    labels, tie_flags = ops.watershed(boundary_maps, markers, mask, mode=mode)  # :73
    return {"binary_mask": mask, "distance_sq": d2, "local_max": local_max, "markers": markers,
            "n_markers": n_markers, "labels": labels, "tie_flags": tie_flags}


def refine_boundaries(boundary_map, threshold=THRESHOLD, return_stages=False):
    """2-D boundary probability map -> int32 label image (the script's ``labels``).

    ``return_stages=True`` returns the script's globals instead: ``binary_mask``, ``distance`` (float64 =
    sqrt of the exact integer squared distance), ``local_max``, ``markers``, ``labels``."""
    was_tensor = isinstance(boundary_map, torch.Tensor)
    t = boundary_map if was_tensor else torch.from_numpy(np.ascontiguousarray(np.asarray(boundary_map, dtype=np.float32)))
    if t.dim() != 2:
        raise ValueError("expected a 2-D boundary map, got shape %s" % (tuple(t.shape),))
    t = t.to(device=_device(), dtype=torch.float32).contiguous()[None]
    st = refine_boundaries_batch(t, threshold)
    conv = (lambda x: x) if was_tensor else (lambda x: x.cpu().numpy())
    labels = conv(st["labels"][0])
    if not return_stages:
        return labels
    d2 = st["distance_sq"][0]
    return {"binary_mask": conv(st["binary_mask"][0].bool()),
            "distance": conv(torch.sqrt(d2.to(torch.float64))),
            "local_max": conv(st["local_max"][0].bool()),
            "markers": conv(st["markers"][0]),
            "labels": labels}


def refine_from_h5(file_path=DEFAULT_FILE, channel=BOUNDARY_CHANNEL, threshold=THRESHOLD, return_stages=False):
    """refine_boundaries.py:28-34 + the chain; ``.npy`` probability stacks are accepted without h5py."""
    if file_path.endswith(".npy"):
        probabilities = np.load(file_path, allow_pickle=False)
    else:
        try:
            import h5py
        except ImportError as e:  # pragma: no cover
            raise ImportError("reading .h5 probabilities needs h5py; save the stack as .npy instead") from e
        with h5py.File(file_path, "r") as f:
            probabilities = np.array(f["exported_data"])
    return refine_boundaries(probabilities[channel], threshold, return_stages)


if __name__ == "__main__":  # script-compatible entry: same default path, prints the number of segments
    import sys
    out = refine_from_h5(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_FILE)
    print("segments:", int(out.max()))
