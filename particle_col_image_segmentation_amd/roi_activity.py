"""Per-ROI isotope activity / distance table: the reference's MATLAB script
``HCN_nanosims_rois_activity_distance_5iso_YG.m`` restated over the HIP kernels.

PARITY UNPINNED: the script cannot run here (no MATLAB / Octave) and the reference holds no output of it; the
functions follow the script line by line (cited) and are checked against the tests' CPU restatement only.

ROI numbering follows MATLAB ``regionprops`` (column-major first pixel, 8-connectivity): components are labelled
on the transposed mask.  Positions are MATLAB-style ``(x, y) = (col + 1, row + 1)`` (.m:164-165).
"""
import numpy as np
import torch

from . import ops
from .tiff_analysis import _device

ISOTOPES_7 = ("12C", "13C", "14N12C", "15N12C", "16O", "17O", "18O")
RATIOS_7 = (("C13act", 1, (1, 0)), ("N15act", 3, (2, 3)), ("O17act", 5, (6, 5, 4)), ("O18act", 6, (6, 5, 4)))  # .m:136-139
ISOTOPES_5 = ("12C", "13C", "14N12C", "15N12C", "32S")
RATIOS_5 = (("C13act", 1, (1, 0)), ("N15act", 3, (2, 3)))


def crop_border(im):
    """.m:18-28: drop the 1-pixel frame of an ion image."""
    return im[1:-1, 1:-1]


def roi_masks_from_rgb(rgb):
    """.m:82-102: red = (R - B) == 255, green = (G - B) == 255 with uint8 saturating subtraction."""
    rgb = np.asarray(rgb, dtype=np.uint8)
    r, g, b = rgb[..., 0].astype(np.int16), rgb[..., 1].astype(np.int16), rgb[..., 2].astype(np.int16)
    return np.clip(r - b, 0, 255) == 255, np.clip(g - b, 0, 255) == 255


def label_matlab_order(mask):
    """bwconncomp / regionprops order: 8-connected components numbered by their column-major first pixel."""
    m = torch.from_numpy(np.ascontiguousarray(np.asarray(mask).T.astype(np.uint8)))[None].to(_device())
    lab, cnt = ops.label_bool8(m)
    return lab[0].t().contiguous(), int(cnt[0].item())


def activity_table(labels, n_rois, planes, roi_class, ratios=RATIOS_7):
    """.m:122-170 / 186-234 for one ROI class.  labels: (H,W) int32 CUDA tensor (1..n_rois); planes: (C,H,W) float32.
    Returns (rows [n, 2 + C + 2*len(ratios)], positions_xy [n, 2]) as numpy float64."""
    planes = planes if isinstance(planes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(planes, dtype=np.float32))
    planes = planes.to(device=_device(), dtype=torch.float32).contiguous()
    stats, _, sums, _ = ops.region_reduce(labels[None].contiguous(), planes=planes[None], cap=max(n_rois, 1))
    st = stats[0, :n_rois].cpu().numpy()
    sm = sums[0, :n_rois].cpu().numpy()
    rows, xy = [], []
    for i in range(n_rois):
        s = sm[i]
        acts = []
        for _, num, den in ratios:
            d = 0.0
            for k in den:
                d = d + s[k]
            with np.errstate(all="ignore"):
                acts.append(np.float64(s[num]) / np.float64(d))
        rows.append([roi_class, i + 1] + list(s) + acts + [a * 100 for a in acts])        # .m:154
        xy.append([st[i, 2] / st[i, 0] + 1.0, st[i, 1] / st[i, 0] + 1.0])                 # .m:164-165
    ncol = 2 + planes.shape[0] + 2 * len(ratios)
    return np.array(rows, np.float64).reshape(n_rois, ncol), np.array(xy, np.float64).reshape(n_rois, 2)


def nearest_neighbour_distances(a_positions, b_positions, raster=19.0, size=512.0):
    """.m:260-268: nearest ROI of the other class for every ROI, in micrometres (/(512/raster))."""
    dev = _device()
    a = torch.from_numpy(np.ascontiguousarray(a_positions, dtype=np.float64)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(b_positions, dtype=np.float64)).to(dev)
    near = torch.cat([ops.nearest_dist(a, b), ops.nearest_dist(b, a)])
    return near.cpu().numpy() / (size / raster)


def boundary_points(mask):
    """bwboundaries(red) as a point SET (.m:271-292): mask pixels with a 4-neighbour outside the mask, (row, col)
    1-based like cell2mat(bwboundaries(...)); the order is irrelevant for the minimum taken at .m:302-305."""
    m = torch.from_numpy(np.ascontiguousarray(np.asarray(mask).astype(np.uint8)))[None].to(_device())
    er = torch.nn.functional.pad(m, (1, 1, 1, 1))  # plumbing: 4-neighbour test on a tiny aggregate mask
    inner = er[:, 1:-1, 1:-1] & er[:, :-2, 1:-1] & er[:, 2:, 1:-1] & er[:, 1:-1, :-2] & er[:, 1:-1, 2:]
    pts = torch.nonzero((m != 0) & (inner == 0))[:, 1:].to(torch.float64) + 1.0
    return pts


def boundary_distances(a_positions, b_positions, aggregate_mask, raster=19.0, size=512.0):
    """.m:301-309.  NOTE the script compares (x, y) ROI centroids with (row, col) boundary points as they are
    (pdist2(a_positions, bd_position)); that latent axis mix is reproduced, not fixed."""
    dev = _device()
    bd = boundary_points(aggregate_mask)
    a = torch.from_numpy(np.ascontiguousarray(a_positions, dtype=np.float64)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(b_positions, dtype=np.float64)).to(dev)
    d = torch.cat([ops.nearest_dist(a, bd), ops.nearest_dist(b, bd)])
    return d.cpu().numpy() / (size / raster)


def activity_distance_table(red_mask, green_mask, planes, aggregate_mask=None, ratios=RATIOS_7, raster=19.0, size=512.0):
    """The script end to end: data.csv columns, + x, y (data_xy.csv), + nearest (data_dist_nearest.csv), + boundary
    distance (data_dist_nearest_bound.csv) when an aggregate mask is given (.m:237, 254-256, 268, 309)."""
    lr, nr = label_matlab_order(red_mask)
    lg, ng = label_matlab_order(green_mask)
    ta, xa = activity_table(lr, nr, planes, 1, ratios)
    tb, xb = activity_table(lg, ng, planes, 2, ratios)
    all_data = np.concatenate([ta, tb], axis=0)
    out = {"data": all_data, "data_xy": np.concatenate([all_data, np.concatenate([xa, xb])], axis=1)}
    if nr and ng:
        near = nearest_neighbour_distances(xa, xb, raster, size)
        out["data_dist_nearest"] = np.concatenate([all_data, near[:, None]], axis=1)
        if aggregate_mask is not None:
            bd = boundary_distances(xa, xb, aggregate_mask, raster, size)
            out["data_dist_nearest_bound"] = np.concatenate([out["data_dist_nearest"], bd[:, None]], axis=1)
    return out
