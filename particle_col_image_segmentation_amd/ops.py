"""Batched device operators: torch tensors in, torch tensors out, HIP inside.

Every function takes ``(B, H, W)`` (or ``(B, C, H, W)``) CUDA tensors, allocates
its outputs and workspace through torch's caching allocator and enqueues the
libpcseg kernels on torch's current stream.  PyTorch is plumbing here (device
memory + streams); all arithmetic is in ``csrc/*.hip``.
"""
import ctypes

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _req(t, dtype, ndim):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("expected a CUDA tensor (the HIP path has no CPU fallback)")
    if t.dtype == torch.bool and dtype == torch.uint8:
        t = t.view(torch.uint8)
    if t.dtype != dtype:
        raise TypeError("expected dtype %s, got %s" % (dtype, t.dtype))
    if t.dim() != ndim:
        raise ValueError("expected %d dims, got shape %s" % (ndim, tuple(t.shape)))
    return t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def argmax_planes(stack):
    """class map = argmax over planes + 1 (tiff_analysis.py:639-642 reads this from ilastik)."""
    stack = _req(stack, torch.float32, 4)
    B, C, H, W = stack.shape
    out = torch.empty((B, H, W), dtype=torch.uint8, device=stack.device)
    lib = _lib.load()
    _lib.check(lib.pcseg_argmax_planes_f32(_ptr(stack), _ptr(out), B, C, H, W, _stream()), "argmax_planes")
    return out


def median5(x):
    """scipy.ndimage.median_filter(x, size=5) (tiff_analysis.py:122, 643)."""
    x = _req(x, torch.uint8, 3)
    B, H, W = x.shape
    out = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.pcseg_median5_u8(_ptr(x), _ptr(out), B, H, W, _stream()), "median5")
    return out


def classmap_label(stack):
    """class map (argmax + 1) -> median_filter(size=5) -> label (tiff_analysis.py:639-643, 743) as one fused front end:
    returns (denoised uint8 (B,H,W), labels int32 (B,H,W), counts int32 (B,))."""
    stack = _req(stack, torch.float32, 4)
    B, C, H, W = stack.shape
    dev = stack.device
    z = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    labels = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    nbytes = lib.pcseg_classmap_label_workspace_bytes(B, H, W)
    ws = _ws(nbytes, dev)
    _lib.check(lib.pcseg_classmap_label_f32(_ptr(stack), C, _ptr(z), _ptr(labels), _ptr(counts), B, H, W, _ptr(ws), nbytes, _stream()),
               "classmap_label")
    return z, labels, counts


def _ccl(fn_name, x):
    x = _req(x, torch.uint8, 3)
    B, H, W = x.shape
    lib = _lib.load()
    labels = torch.empty((B, H, W), dtype=torch.int32, device=x.device)
    counts = torch.empty((B,), dtype=torch.int32, device=x.device)
    nbytes = lib.pcseg_ccl_workspace_bytes(B, H, W)
    ws = _ws(nbytes, x.device)
    _lib.check(getattr(lib, fn_name)(_ptr(x), _ptr(labels), _ptr(counts), B, H, W, _ptr(ws), nbytes, _stream()), fn_name)
    return labels, counts


def label_equal8(x):
    """skimage.measure.label(int image): equal-valued 8-connected components (tiff_analysis.py:743)."""
    return _ccl("pcseg_ccl8_equal_u8", x)


def label_bool8(x):
    """skimage.measure.label(bool image) (tiff_analysis.py:260, 829; refine_boundaries.py:64)."""
    return _ccl("pcseg_ccl8_bool", x)


def label_bool4(x):
    """scipy.ndimage.label(bool image), 4-connectivity."""
    return _ccl("pcseg_ccl4_bool", x)


def compact_labels(roots):
    roots = _req(roots, torch.int32, 3)
    B, H, W = roots.shape
    lib = _lib.load()
    labels = torch.empty_like(roots)
    counts = torch.empty((B,), dtype=torch.int32, device=roots.device)
    nbytes = lib.pcseg_ccl_workspace_bytes(B, H, W)
    ws = _ws(nbytes, roots.device)
    _lib.check(lib.pcseg_compact_labels(_ptr(roots), _ptr(labels), _ptr(counts), B, H, W, _ptr(ws), nbytes, _stream()),
               "compact_labels")
    return labels, counts


def region_init(counts, cap, C, shape, device):
    """Initialised (not yet filled) region tables of a (B, H, W) label batch: stats int64 (B, cap, 8) with the
    reduction's neutral rows, sums float64 (B, cap, C) zeroed (rows below counts[b]), overflow int32 (B,) cleared."""
    B, H, W = shape
    counts = _req(counts, torch.int32, 1)
    stats = torch.empty((B, cap, 8), dtype=torch.int64, device=device)
    sums = torch.empty((B, cap, C), dtype=torch.float64, device=device)
    overflow = torch.empty((B,), dtype=torch.int32, device=device)
    lib = _lib.load()
    _lib.check(lib.pcseg_region_init(_ptr(counts), cap, C, B, H, W, _ptr(stats), _ptr(sums), _ptr(overflow), _stream()), "region_init")
    return stats, sums, overflow


def region_sums2(labels_a, cls, sum_classes, sums_a, labels_b, sums_b, planes, stats_b=None, overflow_b=None):
    """Per-label plane sums of two label images in ONE pass over the planes (csrc/reduce.hip region_sums2_col_kernel):
    ``sums_a`` / ``sums_b`` (float64 (B, cap, C), zeroed by ``region_reduce(..., zero_sums=C)`` or ``region_init``) are
    added to in place; image A only under the class values in ``sum_classes`` (bit v = value v, 0 = everywhere).
    ``stats_b`` / ``overflow_b`` (from ``region_init``): image B's integer columns are accumulated in the same walk."""
    labels_a = _req(labels_a, torch.int32, 3)
    labels_b = _req(labels_b, torch.int32, 3)
    planes = _req(planes, torch.float32, 4)
    sums_a = _req(sums_a, torch.float64, 3)
    sums_b = _req(sums_b, torch.float64, 3)
    cls = _req(cls, torch.uint8, 3) if cls is not None else None
    B, C, H, W = planes.shape
    if tuple(labels_a.shape) != (B, H, W) or tuple(labels_b.shape) != (B, H, W) or sums_a.shape[2] != C or sums_b.shape[2] != C:
        raise ValueError("label images / sums tables do not match the planes")
    if stats_b is not None:
        stats_b = _req(stats_b, torch.int64, 3)
        if tuple(stats_b.shape) != (B, sums_b.shape[1], 8):
            raise ValueError("stats_b must be (B, cap_b, 8)")
    lib = _lib.load()
    _lib.check(lib.pcseg_region_sums2(_ptr(labels_a), _ptr(cls), int(sum_classes), sums_a.shape[1], _ptr(sums_a), _ptr(labels_b),
                                      sums_b.shape[1], _ptr(sums_b), _ptr(stats_b), _ptr(overflow_b), _ptr(planes), C, B, H, W,
                                      _stream()), "region_sums2")


def region_reduce(labels, counts=None, cls=None, planes=None, cap=None, sum_classes=0, zero_sums=0):
    """regionprops sums + optional class at first pixel + optional per-label plane sums.  ``sum_classes`` (bit v = class
    value v; 0 = all): plane sums only over pixels of those classes -- the other regions keep 0 and cost no plane reads.
    ``zero_sums`` = C (without ``planes``): also return a (B, cap, C) sums table with the rows below ``counts`` zeroed, for
    :func:`region_sums2`.

    Returns (stats int64 (B,cap,8), cls_out uint8 (B,cap) | None, sums float64 (B,cap,C) | None,
    overflow int32 (B,))."""
    labels = _req(labels, torch.int32, 3)
    B, H, W = labels.shape
    dev = labels.device
    if cap is None:
        if counts is None:
            raise ValueError("cap or counts is required")
        cap = max(1, int(counts.max().item()))
    # no zero fill: the library initialises the rows it fills (all of them, or the first counts[b] of frame b); rows
    # beyond counts[b] are never read by anything that takes `counts`
    stats = torch.empty((B, cap, 8), dtype=torch.int64, device=dev)
    cls_out = None
    sums = None
    C = 0
    if cls is not None:
        cls = _req(cls, torch.uint8, 3)
        cls_out = torch.empty((B, cap), dtype=torch.uint8, device=dev)
    if planes is not None:
        planes = _req(planes, torch.float32, 4)
        C = planes.shape[1]
        sums = torch.empty((B, cap, C), dtype=torch.float64, device=dev)
    elif zero_sums:
        C = int(zero_sums)
        sums = torch.empty((B, cap, C), dtype=torch.float64, device=dev)
    if counts is not None:
        counts = _req(counts, torch.int32, 1)
    overflow = torch.empty((B,), dtype=torch.int32, device=dev)
    lib = _lib.load()
    if sum_classes and (cls is None or planes is None):
        raise ValueError("sum_classes needs cls and planes")
    _lib.check(lib.pcseg_region_reduce_sel(_ptr(labels), _ptr(counts), _ptr(cls), int(sum_classes), _ptr(planes), C, B, H, W,
                                           cap, _ptr(stats), _ptr(cls_out), _ptr(sums), _ptr(overflow), _stream()),
               "region_reduce")
    return stats, cls_out, sums, overflow


def threshold_lt(img, threshold):
    """binary_mask = boundary_map < threshold (refine_boundaries.py:44-45)."""
    img = _req(img, torch.float32, 3)
    B, H, W = img.shape
    out = torch.empty((B, H, W), dtype=torch.uint8, device=img.device)
    lib = _lib.load()
    _lib.check(lib.pcseg_threshold_lt_f32(_ptr(img), float(threshold), _ptr(out), B, H, W, _stream()), "threshold_lt")
    return out


def edt_sq(mask, cap=-1):
    """exact squared EDT of a 0/1 mask (refine_boundaries.py:60, tiff_analysis.py:996)."""
    mask = _req(mask, torch.uint8, 3)
    B, H, W = mask.shape
    lib = _lib.load()
    d2 = torch.empty((B, H, W), dtype=torch.int32, device=mask.device)
    nbytes = lib.pcseg_edt_workspace_bytes(B, H, W)
    ws = _ws(nbytes, mask.device)
    _lib.check(lib.pcseg_edt_sq_u8(_ptr(mask), _ptr(d2), B, H, W, int(cap), _ptr(ws), nbytes, _stream()), "edt_sq")
    return d2


def _plane_view(img):
    """(B,H,W) float32 view whose frames may be strided (a plane of a (B,C,H,W) stack): (tensor, frame stride)."""
    if not isinstance(img, torch.Tensor) or not img.is_cuda or img.dtype != torch.float32 or img.dim() != 3:
        raise TypeError("expected a (B,H,W) float32 CUDA tensor")
    B, H, W = img.shape
    if img.stride(2) == 1 and img.stride(1) == W and (B == 1 or img.stride(0) >= H * W):
        return img, (img.stride(0) if B > 1 else H * W)
    return img.contiguous(), H * W


def edt_sq_lt(img, threshold, want_mask=True):
    """fused threshold + squared EDT (refine_boundaries.py:44-45, 60); img may be a plane view of a stack."""
    img, fstride = _plane_view(img)
    B, H, W = img.shape
    lib = _lib.load()
    d2 = torch.empty((B, H, W), dtype=torch.int32, device=img.device)
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=img.device) if want_mask else None
    nbytes = lib.pcseg_edt_workspace_bytes(B, H, W)
    ws = _ws(nbytes, img.device)
    _lib.check(lib.pcseg_edt_sq_lt_f32(_ptr(img), fstride, float(threshold), _ptr(d2), _ptr(mask), B, H, W, _ptr(ws),
                                       nbytes, _stream()), "edt_sq_lt")
    return d2, mask


def dilate_disk(x, value_bits, radius):
    """binary_dilation(((value_bits >> x) & 1), disk(radius)) (tiff_analysis.py:827-828, 990)."""
    x = _req(x, torch.uint8, 3)
    B, H, W = x.shape
    lib = _lib.load()
    out = torch.empty_like(x)
    nbytes = lib.pcseg_edt_workspace_bytes(B, H, W)
    ws = _ws(nbytes, x.device)
    _lib.check(lib.pcseg_dilate_disk_u8(_ptr(x), ctypes.c_uint64(int(value_bits)), int(radius), _ptr(out), B, H, W,
                                        _ptr(ws), nbytes, _stream()), "dilate_disk")
    return out


def fill_particle(ds, particle_label, cell_label, overlap_label, dilation_radius, dist_threshold, overlap_area=None):
    """fill_particle_area (tiff_analysis.py:982-1015); overlap_area int64 (B,) is accumulated in place."""
    ds = _req(ds, torch.uint8, 3)
    B, H, W = ds.shape
    lib = _lib.load()
    out = torch.empty_like(ds)
    if overlap_area is None:
        overlap_area = torch.zeros((B,), dtype=torch.int64, device=ds.device)
    nbytes = lib.pcseg_edt_workspace_bytes(B, H, W)
    ws = _ws(nbytes, ds.device)
    _lib.check(lib.pcseg_fill_particle(_ptr(ds), _ptr(out), int(particle_label), int(cell_label), int(overlap_label),
                                       int(dilation_radius), int(dist_threshold), _ptr(overlap_area), B, H, W,
                                       _ptr(ws), nbytes, _stream()), "fill_particle")
    return out, overlap_area


def fill_holes(mask):
    """scipy.ndimage.binary_fill_holes (tiff_analysis.py:880)."""
    mask = _req(mask, torch.uint8, 3)
    B, H, W = mask.shape
    lib = _lib.load()
    out = torch.empty_like(mask)
    nbytes = lib.pcseg_fill_holes_workspace_bytes(B, H, W)
    ws = _ws(nbytes, mask.device)
    _lib.check(lib.pcseg_fill_holes(_ptr(mask), _ptr(out), B, H, W, _ptr(ws), nbytes, _stream()), "fill_holes")
    return out


def local_maxima(img, want_mask=True, want_markers=True):
    """skimage.morphology.local_maxima + measure.label on an int32 image (refine_boundaries.py:63-64)."""
    img = _req(img, torch.int32, 3)
    B, H, W = img.shape
    lib = _lib.load()
    is_max = torch.empty((B, H, W), dtype=torch.uint8, device=img.device) if want_mask else None
    markers = torch.empty((B, H, W), dtype=torch.int32, device=img.device) if want_markers else None
    counts = torch.empty((B,), dtype=torch.int32, device=img.device)
    nbytes = lib.pcseg_local_maxima_workspace_bytes(B, H, W)
    ws = _ws(nbytes, img.device)
    _lib.check(lib.pcseg_local_maxima_i32(_ptr(img), _ptr(is_max), _ptr(markers), _ptr(counts), B, H, W, _ptr(ws),
                                          nbytes, _stream()), "local_maxima")
    return is_max, markers, counts


def watershed(img, markers, mask, mode=0):
    """skimage.segmentation.watershed(img, markers, mask=mask) (refine_boundaries.py:73).

    Returns (labels int32, tie_flags int32 (B,)): tie_flags[b] = 1 where the parallel flood could not be
    proven exact (mode 0 re-runs those frames with the exact sequential flood)."""
    img, fstride = _plane_view(img)
    markers = _req(markers, torch.int32, 3)
    mask = _req(mask, torch.uint8, 3)
    B, H, W = img.shape
    lib = _lib.load()
    out = torch.empty((B, H, W), dtype=torch.int32, device=img.device)
    flags = torch.zeros((B,), dtype=torch.int32, device=img.device)
    nbytes = lib.pcseg_watershed_workspace_bytes(B, H, W)
    ws = _ws(nbytes, img.device)
    _lib.check(lib.pcseg_watershed4_f32(_ptr(img), fstride, _ptr(markers), _ptr(mask), _ptr(out), _ptr(flags), B, H, W, int(mode),
                                        _ptr(ws), nbytes, _stream()), "watershed")
    return out, flags


def dilated_roots(x, value_bits, radius):
    """components of binary_dilation(((value_bits >> x) & 1), disk(radius)) as a union-find parent image (A6)."""
    x = _req(x, torch.uint8, 3)
    B, H, W = x.shape
    lib = _lib.load()
    roots = torch.empty((B, H, W), dtype=torch.int32, device=x.device)
    nbytes = lib.pcseg_dilate_ccl_workspace_bytes(B, H, W)
    ws = _ws(nbytes, x.device)
    _lib.check(lib.pcseg_dilate_ccl_roots_u8(_ptr(x), ctypes.c_uint64(int(value_bits)), int(radius), _ptr(roots), B, H, W,
                                             _ptr(ws), nbytes, _stream()), "dilated_roots")
    return roots


def dilated_runs(x, value_bits, radius, run_parent=None):
    """components of binary_dilation(((value_bits >> x) & 1), disk(radius)) WITHOUT a label image (A6): the dilated
    mask as 32-row column words int32 (B, ceil(H/32), W) and a union-find over its vertical runs (int32 (B,H,W) scratch
    of which only the run-head entries are written; pass ``run_parent`` to reuse one).  For merge_groups_runs."""
    x = _req(x, torch.uint8, 3)
    B, H, W = x.shape
    lib = _lib.load()
    bits = torch.empty((B, (H + 31) // 32, W), dtype=torch.int32, device=x.device)
    if run_parent is None:
        run_parent = torch.empty((B, H, W), dtype=torch.int32, device=x.device)
    nbytes = lib.pcseg_dilate_ccl_runs_workspace_bytes(B, H, W)
    ws = _ws(nbytes, x.device)
    _lib.check(lib.pcseg_dilate_ccl_runs_u8(_ptr(x), ctypes.c_uint64(int(value_bits)), int(radius), _ptr(bits), _ptr(run_parent),
                                            B, H, W, _ptr(ws), nbytes, _stream()), "dilated_runs")
    return bits, run_parent


def merge_groups_runs(bits, run_parent, stats, region_list, n_list):
    """get_merged_regions grouping (tiff_analysis.py:843-878) on the run components of dilated_runs()."""
    bits = _req(bits, torch.int32, 3)
    run_parent = _req(run_parent, torch.int32, 3)
    stats = _req(stats, torch.int64, 3)
    region_list = _req(region_list, torch.int32, 2)
    n_list = _req(n_list, torch.int32, 1)
    B, H, W = run_parent.shape
    cap = stats.shape[1]
    list_cap = region_list.shape[1]
    lib = _lib.load()
    group_of = torch.zeros((B, list_cap), dtype=torch.int32, device=stats.device)
    n_groups = torch.zeros((B,), dtype=torch.int32, device=stats.device)
    nbytes = lib.pcseg_merge_groups_workspace_bytes(B, list_cap)
    ws = _ws(nbytes, stats.device)
    _lib.check(lib.pcseg_merge_groups_runs(_ptr(bits), _ptr(run_parent), _ptr(stats), _ptr(region_list), _ptr(n_list), _ptr(group_of),
                                           _ptr(n_groups), B, H, W, cap, list_cap, _ptr(ws), nbytes, _stream()), "merge_groups_runs")
    return group_of, n_groups


def dilated_runs_multi(x, value_bits_list, radius):
    """dilated_runs for several masks of one class map in one go (the class map is read once; every later pass is ONE launch
    over all masks): returns (bits int32 (M, B, ceil(H/32), W), run_parent int32 (M, B, H, W))."""
    import numpy as np
    x = _req(x, torch.uint8, 3)
    B, H, W = x.shape
    M = len(value_bits_list)
    lib = _lib.load()
    bits = torch.empty((M, B, (H + 31) // 32, W), dtype=torch.int32, device=x.device)
    run_parent = torch.empty((M, B, H, W), dtype=torch.int32, device=x.device)
    nbytes = lib.pcseg_dilate_ccl_runs_workspace_bytes(B * M, H, W)
    ws = _ws(nbytes, x.device)
    vb = np.array([int(v) for v in value_bits_list], np.uint64)
    _lib.check(lib.pcseg_dilate_ccl_runs_multi_u8(_ptr(x), ctypes.c_void_p(vb.ctypes.data), M, int(radius), _ptr(bits), _ptr(run_parent),
                                                  B, H, W, _ptr(ws), nbytes, _stream()), "dilated_runs_multi")
    return bits, run_parent


def merge_groups_fused_multi(bits, run_parent, stats, region_lists, n_lists, slots):
    """merge_groups_fused for the M masks of dilated_runs_multi in one launch; mask m is grouped over the list of type slot
    ``slots[m]``.  Returns (group_of (M,B,cap), n_groups (M,B), group_stats (M,B,cap,8))."""
    import numpy as np
    bits = _req(bits, torch.int32, 4)
    run_parent = _req(run_parent, torch.int32, 4)
    stats = _req(stats, torch.int64, 3)
    region_lists = _req(region_lists, torch.int32, 3)
    n_lists = _req(n_lists, torch.int32, 2)
    M, B, H, W = run_parent.shape
    cap = stats.shape[1]
    n_slots = region_lists.shape[1]
    if len(slots) != M or bits.shape[0] != M or region_lists.shape[2] != cap or tuple(n_lists.shape) != (B, n_slots):
        raise ValueError("masks / lists / region table do not match")
    dev = stats.device
    lib = _lib.load()
    group_of = torch.empty((M, B, cap), dtype=torch.int32, device=dev)
    n_groups = torch.empty((M, B), dtype=torch.int32, device=dev)
    gstats = torch.empty((M, B, cap, 8), dtype=torch.int64, device=dev)
    nbytes = lib.pcseg_merge_groups_workspace_bytes(B * M, cap)
    ws = _ws(nbytes, dev)
    sl = np.array([int(v) for v in slots], np.int32)
    _lib.check(lib.pcseg_merge_groups_fused_multi(_ptr(bits), _ptr(run_parent), _ptr(stats), _ptr(region_lists), _ptr(n_lists),
                                                  ctypes.c_void_p(sl.ctypes.data), M, n_slots, _ptr(group_of), _ptr(n_groups),
                                                  _ptr(gstats), B, H, W, cap, _ptr(ws), nbytes, _stream()), "merge_groups_fused_multi")
    return group_of, n_groups, gstats


def merge_groups_fused(bits, run_parent, stats, region_lists, n_lists, slot):
    """get_merged_regions grouping + the member sums of the groups (tiff_analysis.py:843-878) for type slot ``slot`` in
    one launch: ``region_lists`` int32 (B, n_slots, cap) / ``n_lists`` int32 (B, n_slots) as classify_regions returns
    them.  Returns (group_of int32 (B,cap), n_groups int32 (B,), group_stats int64 (B,cap,8)); entries beyond a frame's
    list length / group count are not initialised."""
    bits = _req(bits, torch.int32, 3)
    run_parent = _req(run_parent, torch.int32, 3)
    stats = _req(stats, torch.int64, 3)
    region_lists = _req(region_lists, torch.int32, 3)
    n_lists = _req(n_lists, torch.int32, 2)
    B, H, W = run_parent.shape
    cap = stats.shape[1]
    n_slots = region_lists.shape[1]
    if region_lists.shape[2] != cap or tuple(n_lists.shape) != (B, n_slots):
        raise ValueError("region lists of shape %s / %s do not match the region table" % (tuple(region_lists.shape), tuple(n_lists.shape)))
    dev = stats.device
    lib = _lib.load()
    group_of = torch.empty((B, cap), dtype=torch.int32, device=dev)
    n_groups = torch.empty((B,), dtype=torch.int32, device=dev)
    gstats = torch.empty((B, cap, 8), dtype=torch.int64, device=dev)
    nbytes = lib.pcseg_merge_groups_workspace_bytes(B, cap)
    ws = _ws(nbytes, dev)
    _lib.check(lib.pcseg_merge_groups_fused(_ptr(bits), _ptr(run_parent), _ptr(stats), _ptr(region_lists), _ptr(n_lists), int(slot),
                                            n_slots, _ptr(group_of), _ptr(n_groups), _ptr(gstats), B, H, W, cap, _ptr(ws), nbytes,
                                            _stream()), "merge_groups_fused")
    return group_of, n_groups, gstats


def merge_groups(dilated_labels, stats, region_list, n_list, roots=False):
    """get_merged_regions grouping (tiff_analysis.py:843-878): group id per list entry, 0 = dropped.
    roots=True: `dilated_labels` is the parent image of dilated_roots()."""
    dl = _req(dilated_labels, torch.int32, 3)
    stats = _req(stats, torch.int64, 3)
    region_list = _req(region_list, torch.int32, 2)
    n_list = _req(n_list, torch.int32, 1)
    B, H, W = dl.shape
    cap = stats.shape[1]
    list_cap = region_list.shape[1]
    lib = _lib.load()
    group_of = torch.zeros((B, list_cap), dtype=torch.int32, device=dl.device)
    n_groups = torch.zeros((B,), dtype=torch.int32, device=dl.device)
    nbytes = lib.pcseg_merge_groups_workspace_bytes(B, list_cap)
    ws = _ws(nbytes, dl.device)
    _lib.check(lib.pcseg_merge_groups(_ptr(dl), int(bool(roots)), _ptr(stats), _ptr(region_list), _ptr(n_list), _ptr(group_of),
                                      _ptr(n_groups), B, H, W, cap, list_cap, _ptr(ws), nbytes, _stream()), "merge_groups")
    return group_of, n_groups


def group_reduce(stats, region_list, n_list, group_of, n_groups, H, W):
    """member sums of merged groups (tiff_analysis.py:855-872): int64 (B, list_cap, 8)."""
    B, cap = stats.shape[0], stats.shape[1]
    list_cap = region_list.shape[1]
    lib = _lib.load()
    gstats = torch.zeros((B, list_cap, 8), dtype=torch.int64, device=stats.device)
    _lib.check(lib.pcseg_group_reduce(_ptr(stats), _ptr(region_list), _ptr(n_list), _ptr(group_of), _ptr(n_groups),
                                      _ptr(gstats), B, H, W, cap, list_cap, _stream()), "group_reduce")
    return gstats


class ClassTables:
    """Host-side class tables of pcseg_classify_regions built from the reference's ``cell_types`` dict and its
    MIN_CELL_AREA / MIN_CLUSTER_AREA constants (tiff_analysis.py:54-60, 754-773)."""

    def __init__(self, cell_types, cell_type_names, min_cell_area, min_cluster_area):
        import numpy as np
        self.slot_names = []
        slot = np.full(256, 255, np.uint8)
        particle = np.zeros(256, np.uint8)
        for val, name in cell_types.items():
            if name in cell_type_names:
                if name not in self.slot_names:
                    self.slot_names.append(name)
                slot[val] = self.slot_names.index(name)
            elif name == "Particle":
                particle[val] = 1
        if len(self.slot_names) > 4:
            raise ValueError("at most 4 cell types")
        if any(not (0 <= int(v) < 64) for v, t in cell_types.items() if t in cell_type_names or t == "Particle"):
            raise ValueError("cell / particle class values must be below 64 (they index a 64-bit class set)")
        self.slot = slot
        self.particle = particle
        self.min_cell = np.array([min_cell_area[n] for n in self.slot_names] or [0], np.int32)
        self.min_cluster = np.array([min_cluster_area[n] for n in self.slot_names] or [0], np.int32)
        # class value that get_cell_clusters_from_distances looks up per type: the FIRST key with that name (:806-810)
        self.slot_value = []
        for n in self.slot_names:
            self.slot_value.append([v for v, t in cell_types.items() if t == n][0])
        self.cell_values = [v for v, t in cell_types.items() if t in cell_type_names]
        self.particle_value = None
        for v, t in cell_types.items():
            if t == "Particle":
                self.particle_value = v


def classify_regions(stats, cls_out, counts, tables):
    """per-region loop of get_cell_positions_and_areas + region lists (tiff_analysis.py:754-781, 794-796)."""
    stats = _req(stats, torch.int64, 3)
    cls_out = _req(cls_out, torch.uint8, 2)
    counts = _req(counts, torch.int32, 1)
    B, cap = stats.shape[0], stats.shape[1]
    dev = stats.device
    # no fills: the kernel writes kind / slot_of / cells for every region below counts[b], the lists up to their lengths and
    # every per-frame scalar; nothing reads beyond those (eight fill launches per batch, 26 MB of them, for nothing)
    out = {
        "kind": torch.empty((B, cap), dtype=torch.uint8, device=dev),
        "slot_of": torch.empty((B, cap), dtype=torch.uint8, device=dev),
        "cells": torch.empty((B, cap), dtype=torch.int32, device=dev),
        "particle_area": torch.empty((B,), dtype=torch.int64, device=dev),
        "type_stats": torch.empty((B, 4, 4), dtype=torch.int64, device=dev),
        "region_list": torch.empty((B, 5, cap), dtype=torch.int32, device=dev),
        "n_list": torch.empty((B, 5), dtype=torch.int32, device=dev),
        "nan_flag": torch.empty((B,), dtype=torch.int32, device=dev),
    }
    lib = _lib.load()
    hp = lambda a: ctypes.c_void_p(a.ctypes.data)
    _lib.check(lib.pcseg_classify_regions(_ptr(stats), _ptr(cls_out), _ptr(counts), hp(tables.slot), hp(tables.particle),
                                          hp(tables.min_cell), hp(tables.min_cluster), len(tables.slot_names),
                                          _ptr(out["kind"]), _ptr(out["slot_of"]), _ptr(out["cells"]),
                                          _ptr(out["particle_area"]), _ptr(out["type_stats"]), _ptr(out["region_list"]),
                                          _ptr(out["n_list"]), _ptr(out["nan_flag"]), B, cap, _stream()), "classify_regions")
    return out


def build_tables(res, groups, frame_ids, C, ratios, check=False, distance_slots=None, raster=19.0):
    """csrc/tables.hip: dense row tables of one batch (see FramePipeline.tables_device).  ``distance_slots``: the
    class value -> type slot table (uint8[256] numpy); with it the result carries ``cell_dist`` (one value per row of
    ``cells``, NaN = no entry) of pcseg_cell_distances."""
    lib = _lib.load()
    B, cap = res["stats"].shape[0], res["stats"].shape[1]
    dev = res["stats"].device
    if len(ratios) > 8 or any(len(den) > 4 for _, _, den in ratios):
        raise ValueError("at most 8 ratios of at most 4 denominator planes")
    ti = _lib.TableInputs()
    ti.B, ti.cap, ti.C, ti.n_ratios = B, cap, C, len(ratios)
    keep = []  # tensors referenced by raw pointers until the kernels are enqueued

    def ptr(t, dtype, shape):
        t = _req(t, dtype, len(shape))
        if tuple(t.shape) != tuple(shape):
            raise ValueError("table input of shape %s, expected %s" % (tuple(t.shape), tuple(shape)))
        keep.append(t)
        return t.data_ptr()

    ti.frame_ids = ptr(frame_ids, torch.int64, (B,))
    ti.counts = ptr(res["counts"], torch.int32, (B,))
    ti.stats = ptr(res["stats"], torch.int64, (B, cap, 8))
    ti.cls_out = ptr(res["cls_out"], torch.uint8, (B, cap))
    ti.cc_sums = ptr(res["cc_sums"], torch.float64, (B, cap, C))
    ti.kind = ptr(res["kind"], torch.uint8, (B, cap))
    ti.slot_of = ptr(res["slot_of"], torch.uint8, (B, cap))
    ti.cells = ptr(res["cells"], torch.int32, (B, cap))
    ti.particle_area = ptr(res["particle_area"], torch.int64, (B,))
    ti.overlap_area = ptr(res["overlap_area"], torch.int64, (B,))
    ti.type_stats = ptr(res["type_stats"], torch.int64, (B, 4, 4))
    ti.tie_flags = ptr(res["tie_flags"], torch.int32, (B,))
    ti.region_list = ptr(res["region_list"], torch.int32, (B, 5, cap))
    ti.n_list = ptr(res["n_list"], torch.int32, (B, 5))
    for s, g in groups.items():
        ti.group_of[s] = ptr(g["group_of"], torch.int32, (B, cap))
        ti.n_groups[s] = ptr(g["n_groups"], torch.int32, (B,))
        ti.group_stats[s] = ptr(g["group_stats"], torch.int64, (B, cap, 8))
    ti.n_markers = ptr(res["n_markers"], torch.int32, (B,))
    ti.ws_stats = ptr(res["ws_stats"], torch.int64, (B, cap, 8))
    ti.ws_sums = ptr(res["ws_sums"], torch.float64, (B, cap, C))
    for k, (_, num, den) in enumerate(ratios):
        ti.ratio_num[k] = int(num)
        for j in range(4):
            ti.ratio_den[k][j] = int(den[j]) if j < len(den) else -1
    ti.overflow = ptr(res["overflow"], torch.int32, (B,))
    ti.ws_overflow = ptr(res["ws_overflow"], torch.int32, (B,))
    ti.nan_flag = ptr(res["nan_flag"], torch.int32, (B,))
    nbytes = lib.pcseg_table_workspace_bytes(B, cap)
    ws = _ws(nbytes, dev)
    totals = torch.empty((6,), dtype=torch.int64, device=dev)
    _lib.check(lib.pcseg_table_layout(ctypes.byref(ti), _ptr(totals), _ptr(ws), nbytes, _stream()), "table_layout")
    # the one host read: sizes of the outputs, plus the flags BatchResult.check() would otherwise fetch one by one
    n_roi, n_cell, n_group, n_overflow, n_ws_overflow, n_nan = (int(v) for v in totals.cpu())
    if check and (n_overflow or n_ws_overflow):
        raise RuntimeError("region table capacity exceeded: raise FramePipeline(cap=...)")
    if check and n_nan:
        raise ValueError("cannot convert float NaN to integer")  # tiff_analysis.py:776-781
    nr = len(ratios)
    # (one spare row each: an empty table still needs a non-null pointer for the library's argument check)
    rois = torch.empty((n_roi + 1, 5 + C + nr), dtype=torch.float64, device=dev)
    cells = torch.empty((n_cell + 1, 14 + C + nr), dtype=torch.float64, device=dev)
    grp = torch.empty((n_group + 1, 11), dtype=torch.float64, device=dev)
    frames = torch.empty((B, 17), dtype=torch.int64, device=dev)
    _lib.check(lib.pcseg_table_write(ctypes.byref(ti), _ptr(rois), _ptr(cells), _ptr(grp), _ptr(frames), _ptr(ws), nbytes,
                                     _stream()), "table_write")
    out = {"rois": rois[:n_roi], "cells": cells[:n_cell], "groups": grp[:n_group], "frames": frames, "frame_ids": frame_ids}
    if distance_slots is not None:
        dist = torch.empty((n_cell + 1,), dtype=torch.float64, device=dev)
        _lib.check(lib.pcseg_cell_distances(_ptr(cells), n_cell, cells.shape[1], ctypes.c_void_p(distance_slots.ctypes.data),
                                            float(raster), 512.0, _ptr(dist), B, _ptr(ws), nbytes, _stream()), "cell_distances")
        out["cell_dist"] = dist[:n_cell]
    return out


def remove_overlapping(dapi, other, threshold):
    """combine_cell_positions_and_clusters (tiff_analysis.py:252-287)."""
    dapi = _req(dapi, torch.uint8, 3)
    other = _req(other, torch.uint8, 3)
    B, H, W = dapi.shape
    lib = _lib.load()
    out = torch.empty_like(dapi)
    nbytes = lib.pcseg_overlap_workspace_bytes(B, H, W)
    ws = _ws(nbytes, dapi.device)
    _lib.check(lib.pcseg_remove_overlapping(_ptr(dapi), _ptr(other), float(threshold), _ptr(out), B, H, W, _ptr(ws),
                                            nbytes, _stream()), "remove_overlapping")
    return out


def otsu_hist(img):
    """256-bin histogram of each frame over its own [min, max] (north_star extension X1)."""
    img = _req(img, torch.float32, 3)
    B, H, W = img.shape
    lib = _lib.load()
    hist = torch.zeros((B, 256), dtype=torch.int64, device=img.device)
    lohi = torch.zeros((B, 2), dtype=torch.float32, device=img.device)
    _lib.check(lib.pcseg_otsu_hist_f32(_ptr(img), _ptr(hist), _ptr(lohi), B, H, W, _stream()), "otsu_hist")
    return hist, lohi


def morph3x3(mask, erode):
    """3x3 binary erosion / dilation (north_star extension X2)."""
    mask = _req(mask, torch.uint8, 3)
    B, H, W = mask.shape
    lib = _lib.load()
    out = torch.empty_like(mask)
    _lib.check(lib.pcseg_morph3x3(_ptr(mask), _ptr(out), int(bool(erode)), B, H, W, _stream()), "morph3x3")
    return out


def nearest_dist(a, b):
    """min over b of the Euclidean distance, for every row of a: (na,2), (nb,2) float64 CUDA tensors (.m:260-263)."""
    a = _req(a, torch.float64, 2)
    b = _req(b, torch.float64, 2)
    out = torch.empty((a.shape[0],), dtype=torch.float64, device=a.device)
    if a.shape[0] == 0:
        return out
    if b.shape[0] == 0:
        return out.fill_(float("inf"))
    lib = _lib.load()
    _lib.check(lib.pcseg_nearest_dist_f64(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out), _stream()), "nearest_dist")
    return out


def threshold_otsu(img, return_hist=False):
    """skimage.filters.threshold_otsu per frame, entirely on the device (north_star extension X1; the library itself
    is the pin: tests/golden/extensions.npz): float64 (B,) CUDA tensor holding the float32 bin centre the
    library returns.  ``return_hist``: also the (B,256) histogram and the (B,2) [min, max] it was taken from."""
    img = _req(img, torch.float32, 3)
    B, H, W = img.shape
    lib = _lib.load()
    thr = torch.empty((B,), dtype=torch.float64, device=img.device)
    hist = torch.empty((B, 256), dtype=torch.int64, device=img.device)
    lohi = torch.empty((B, 2), dtype=torch.float32, device=img.device)
    _lib.check(lib.pcseg_otsu_f32(_ptr(img), _ptr(thr), _ptr(hist), _ptr(lohi), B, H, W, _stream()), "otsu")
    return (thr, hist, lohi) if return_hist else thr
