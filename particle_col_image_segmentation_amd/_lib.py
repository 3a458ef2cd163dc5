"""ctypes binding of libpcseg.so (include/pcseg.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible
every compute call raises -- loudly, by design.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpcseg.so")
_lib = None


class PcsegError(RuntimeError):
    pass


_P = c_void_p
_I = c_int
# name -> (restype, argtypes); mirrors include/pcseg.h one to one
SIGNATURES = {
    "pcseg_version": (c_int, []),
    "pcseg_last_error": (c_char_p, []),
    "pcseg_device_count": (c_int, []),
    "pcseg_timing_enable": (None, [c_int]),
    "pcseg_timing_report": (c_int, [c_char_p, c_size_t]),
    "pcseg_argmax_planes_f32": (c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "pcseg_median5_u8": (c_int, [_P, _P, _I, _I, _I, _P]),
    "pcseg_classmap_label_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "pcseg_classmap_label_f32": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_ccl_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "pcseg_ccl8_equal_u8": (c_int, [_P, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_ccl8_bool": (c_int, [_P, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_ccl4_bool": (c_int, [_P, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_compact_labels": (c_int, [_P, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_region_reduce": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "pcseg_region_reduce_n": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "pcseg_region_reduce_sel": (c_int, [_P, _P, _P, ctypes.c_uint64, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "pcseg_region_init": (c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "pcseg_region_sums2": (c_int, [_P, _P, ctypes.c_uint64, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "pcseg_threshold_lt_f32": (c_int, [_P, c_float, _P, _I, _I, _I, _P]),
    "pcseg_edt_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "pcseg_edt_sq_u8": (c_int, [_P, _P, _I, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_edt_sq_lt_f32": (c_int, [_P, c_int64, c_float, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_dilate_disk_u8": (c_int, [_P, c_uint64, _I, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_fill_particle": (c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_fill_holes_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "pcseg_fill_holes": (c_int, [_P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_local_maxima_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "pcseg_local_maxima_i32": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_watershed_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "pcseg_watershed_counters": (None, [_P, _I]),
    "pcseg_watershed4_f32": (c_int, [_P, c_int64, _P, _P, _P, _P, _I, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_merge_groups_workspace_bytes": (c_size_t, [_I, _I]),
    "pcseg_merge_groups": (c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_dilate_ccl_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "pcseg_dilate_ccl_roots_u8": (c_int, [_P, c_uint64, _I, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_dilate_ccl_runs_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "pcseg_dilate_ccl_runs_u8": (c_int, [_P, c_uint64, _I, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_merge_groups_runs": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_dilate_ccl_runs_multi_u8": (c_int, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_merge_groups_fused_multi": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_merge_groups_fused": (c_int, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_group_reduce": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "pcseg_classify_regions": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "pcseg_nearest_dist_f64": (c_int, [_P, _I, _P, _I, _P, _P]),
    "pcseg_overlap_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "pcseg_remove_overlapping": (c_int, [_P, _P, c_double, _P, _I, _I, _I, _P, c_size_t, _P]),
    "pcseg_table_workspace_bytes": (c_size_t, [_I, _I]),
    "pcseg_table_layout": (c_int, [_P, _P, _P, c_size_t, _P]),
    "pcseg_table_write": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "pcseg_cell_distances": (c_int, [_P, c_int64, _I, _P, c_double, c_double, _P, _I, _P, c_size_t, _P]),
    "pcseg_otsu_hist_f32": (c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "pcseg_otsu_f32": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "pcseg_morph3x3": (c_int, [_P, _P, _I, _I, _I, _I, _P]),
}


class TableInputs(ctypes.Structure):
    """struct pcseg_table_inputs of include/pcseg.h, field for field."""
    _fields_ = [("B", ctypes.c_int32), ("cap", ctypes.c_int32), ("C", ctypes.c_int32), ("n_ratios", ctypes.c_int32),
                ("frame_ids", _P),
                ("counts", _P), ("stats", _P), ("cls_out", _P), ("cc_sums", _P),
                ("kind", _P), ("slot_of", _P), ("cells", _P),
                ("particle_area", _P), ("overlap_area", _P), ("type_stats", _P), ("tie_flags", _P),
                ("region_list", _P), ("n_list", _P),
                ("group_of", _P * 5), ("n_groups", _P * 5), ("group_stats", _P * 5),
                ("n_markers", _P), ("ws_stats", _P), ("ws_sums", _P),
                ("ratio_num", ctypes.c_int32 * 8), ("ratio_den", (ctypes.c_int32 * 4) * 8),
                ("overflow", _P), ("ws_overflow", _P), ("nan_flag", _P)]


def load():
    """Load libpcseg.so and declare every prototype of include/pcseg.h."""
    global _lib
    if _lib is not None:
        return _lib
    # PCSEG_LIB: an A/B build of the SAME library (particle_col_image_segmentation_amd.build.build(out_dir=...)), for
    # same-box kernel comparisons; never a different implementation -- the prototypes below must all resolve
    path = os.environ.get("PCSEG_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise PcsegError(
            "libpcseg.so is not built (%s). Run `python -m particle_col_image_segmentation_amd.build` "
            "or __graft_entry__.build(); there is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pcseg_last_error()
        raise PcsegError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
