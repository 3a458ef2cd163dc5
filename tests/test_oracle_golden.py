"""The CPU oracle (oracle/) against the golden vectors captured from the real
reference (tests/golden/make_golden.py).  This is what PINS the oracle."""
import numpy as np
import pytest

from conftest import FUNC_CASES, golden_cases, load_golden
from oracle import oracle as orc


def test_median(primitives):
    cases = golden_cases(primitives, "median")
    assert len(cases) >= 8
    for c in cases:
        np.testing.assert_array_equal(orc.median_filter(c["inp"]), c["out"])


@pytest.mark.parametrize("kind,conn", [("label_eq8", None), ("label_bool8", None), ("label_bool4", 1)])
def test_label(primitives, kind, conn):
    cases = golden_cases(primitives, kind)
    assert cases
    for c in cases:
        out, n = orc.label(c["inp"], connectivity=conn, return_num=True)
        np.testing.assert_array_equal(out, c["out"])
        assert n == int(c["n"])


def test_regionprops(primitives):
    for c in golden_cases(primitives, "props"):
        lab = orc.label(c["inp"])
        np.testing.assert_array_equal(lab, c["labels"])
        regs = orc.regionprops(lab)
        assert [r.label for r in regs] == list(c["label"])
        assert [r.area for r in regs] == list(c["area"])
        np.testing.assert_array_equal(np.array([r.centroid for r in regs]).reshape(-1, 2), c["centroid"])
        np.testing.assert_array_equal(np.array([r.bbox for r in regs]).reshape(-1, 4), c["bbox"])
        np.testing.assert_array_equal(np.array([r.first for r in regs]).reshape(-1, 2), c["first"])
        np.testing.assert_array_equal(np.array([r.coords[0] for r in regs]).reshape(-1, 2), c["first"])
        assert [int(orc.get_type(r, c["inp"])) for r in regs] == list(c["typ"])


def test_dilate_and_edt_identity(primitives):
    for c in golden_cases(primitives, "dilate"):
        for rad in (2, 5, 20):
            ref = c["r%d" % rad]
            np.testing.assert_array_equal(orc.binary_dilation_disk(c["inp"], rad), ref)
            # identity used by the product: dilate(m, disk(r)) == EDT^2(~m) <= r^2
            if c["inp"].any():
                np.testing.assert_array_equal(orc.edt_sq(~c["inp"]) <= rad * rad, ref)


def test_edt(primitives):
    cases = golden_cases(primitives, "edt")
    assert len(cases) >= 10
    for c in cases:
        d2 = orc.edt_sq(c["inp"])
        np.testing.assert_array_equal(np.sqrt(d2.astype(np.float64)), c["out"])
        if c["inp"].size <= 64 * 64:
            np.testing.assert_array_equal(d2, orc.edt_sq(c["inp"], brute=True))


def test_fill_holes(primitives):
    for c in golden_cases(primitives, "fill"):
        np.testing.assert_array_equal(orc.binary_fill_holes(c["inp"]), c["out"])


def test_local_maxima(primitives):
    for c in golden_cases(primitives, "locmax"):
        np.testing.assert_array_equal(orc.distance_transform_edt(c["mask"]), c["dist"])
        lm = orc.local_maxima(c["dist"])
        np.testing.assert_array_equal(lm, c["out"])
        np.testing.assert_array_equal(orc.label(lm), c["markers"])
        # the product works on integer d^2 (order-isomorphic to the float64 distance)
        np.testing.assert_array_equal(orc.local_maxima(orc.edt_sq(c["mask"])), c["out"])
    for c in golden_cases(primitives, "locmax_int"):
        np.testing.assert_array_equal(orc.local_maxima(c["inp"]), c["out"])


def test_watershed(primitives):
    cases = golden_cases(primitives, "ws")
    assert len(cases) >= 40
    for c in cases:
        np.testing.assert_array_equal(orc.watershed(c["img"], c["markers"], c["mask"]), c["out"])


def _cell_types(g):
    return {int(k): str(v) for k, v in zip(g["ct_keys"], g["ct_vals"])}


def _check_regions(regs, g, prefix):
    assert [r.label for r in regs] == list(g[prefix + "_label"])
    assert [r.area for r in regs] == list(g[prefix + "_area"])
    np.testing.assert_array_equal(np.array([r.centroid for r in regs]).reshape(-1, 2), g[prefix + "_centroid"])
    np.testing.assert_array_equal(np.array([r.bbox for r in regs]).reshape(-1, 4), g[prefix + "_bbox"])
    assert [getattr(r, "cells", -1) for r in regs] == list(g[prefix + "_cells"])


def check_merged(groups, g, t):
    """Per-type groups must match in order; the reference builds 'combined' in
    the (hash-randomised) set-iteration order of the type names
    (tiff_analysis.py:794), so that one is compared order-free."""
    got = [(m["area"], tuple(m["bbox"]), [r.label for r in m["regions"]], np.asarray(m["centroid"])) for m in groups]
    off = g["mrg_%s_offsets" % t]
    exp = [(int(g["mrg_%s_area" % t][i]), tuple(int(v) for v in g["mrg_%s_bbox" % t][i]),
            [int(v) for v in g["mrg_%s_members" % t][off[i]:off[i + 1]]], g["mrg_%s_centroid" % t][i])
           for i in range(len(off) - 1)]
    if t == "combined":
        canon = lambda rows: sorted(((a, b, sorted(m), c) for a, b, m, c in rows), key=lambda x: x[2][0])
        got, exp = canon(got), canon(exp)
    assert len(got) == len(exp)
    for a, b in zip(got, exp):
        assert a[:3] == b[:3]
        np.testing.assert_allclose(a[3], b[3], rtol=1e-13, atol=0)


@pytest.mark.parametrize("name", FUNC_CASES)
def test_function_level(name):
    g = load_golden(name)
    ct = _cell_types(g)
    cm = orc.median_filter(g["class_map"])
    np.testing.assert_array_equal(cm, g["denoised"])
    if "crash" in g.files:
        with pytest.raises(ValueError, match="cannot convert float NaN to integer"):
            orc.get_cell_positions_and_areas(cm, ct, merged=True)
        return
    np.testing.assert_array_equal(orc.label(cm), g["label_im"])
    cell_pos, cell_clusters, pa, merged = orc.get_cell_positions_and_areas(cm, ct, merged=True)
    assert pa == int(g["particle_area"])
    assert sorted(cell_pos) == sorted(str(t) for t in g["types_pos"])
    for t in cell_pos:
        _check_regions(cell_pos[t], g, "pos_" + t)
        _check_regions(cell_clusters[t], g, "clu_" + t)
    assert sorted(merged) == [str(t) for t in g["types_merged"]]
    for t, groups in merged.items():
        check_merged(groups, g, t)
    _, images = orc.get_cell_clusters_from_distances(cm, cell_pos, cell_clusters, ct)
    for t, img in images.items():
        np.testing.assert_array_equal(img, g["mimg_" + t])
    cnt, dens, ratio = orc.get_cell_counts_and_densities(cell_pos, cell_clusters, pa)
    for i, k in enumerate(g["cnt_keys"]):
        assert cnt[str(k)] == int(g["cnt"][i])
        assert dens[str(k)] == float(g["dens"][i])
        assert ratio[str(k)] == float(g["ratio"][i])
    rec, pa2 = orc.recreate_particle_area(cm, ct, pa)
    np.testing.assert_array_equal(rec, g["recreated"])
    assert pa2 == int(g["particle_area2"])
    plabel = [k for k, v in ct.items() if v == "Particle"][0]
    upd, ov = orc.fill_particle_area(cm, plabel, 1, plabel)
    np.testing.assert_array_equal(upd, g["fill1"])
    assert ov == int(g["fill1_area"])
    rf = orc.refine_boundaries(g["stack"][3])
    np.testing.assert_array_equal(rf["binary_mask"], g["rf_mask"])
    np.testing.assert_array_equal(rf["distance"], g["rf_distance"])
    np.testing.assert_array_equal(rf["local_max"], g["rf_local_max"])
    np.testing.assert_array_equal(rf["markers"], g["rf_markers"])
    np.testing.assert_array_equal(rf["labels"], g["rf_labels"])


def test_overlap_removal():
    g = load_golden("overlap")
    for i in range(2):
        out = orc.combine_cell_positions_and_clusters(g["ov_%d_dapi" % i], g["ov_%d_other" % i])
        np.testing.assert_array_equal(out, g["ov_%d_out" % i])


def test_extensions_otsu_and_morph3x3_against_the_libraries():
    """X1 / X2 (north_star extensions): skimage.filters.threshold_otsu and the 3x3 binary erosion / dilation of
    skimage.morphology (= scipy.ndimage with the same border values), captured by make_golden.py (extensions.npz)."""
    g = load_golden("extensions")
    i = 0
    while "otsu_%02d_inp" % i in g.files:
        thr, hist = orc.threshold_otsu(g["otsu_%02d_inp" % i])
        assert thr == float(g["otsu_%02d_thr" % i]), i  # bit-exact: the library's float32 bin centre
        if "otsu_%02d_hist" % i in g.files:
            np.testing.assert_array_equal(hist, g["otsu_%02d_hist" % i])
        i += 1
    assert i >= 6
    i = 0
    while "morph_%02d_inp" % i in g.files:
        m = g["morph_%02d_inp" % i]
        np.testing.assert_array_equal(orc.morph3x3(m, 1), g["morph_%02d_erode" % i])
        np.testing.assert_array_equal(orc.morph3x3(m, 0), g["morph_%02d_dilate" % i])
        i += 1
    assert i >= 6
