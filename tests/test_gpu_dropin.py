"""The drop-in Python entry points (same names / arguments / returns as the
reference's tiff_analysis.py and refine_boundaries.py) on the HIP path, against
the golden vectors captured from the real reference."""
import os

import numpy as np
import pytest

from conftest import FUNC_CASES, load_golden
from test_oracle_golden import check_merged

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path has no CPU fallback")
    from particle_col_image_segmentation_amd import tiff_analysis
    return tiff_analysis


def _cell_types(g):
    return {int(k): str(v) for k, v in zip(g["ct_keys"], g["ct_vals"])}


def _check_regions(regs, g, prefix):
    assert [r.label for r in regs] == list(g[prefix + "_label"])
    assert [r.area for r in regs] == list(g[prefix + "_area"])
    np.testing.assert_array_equal(np.array([r.centroid for r in regs]).reshape(-1, 2), g[prefix + "_centroid"])
    np.testing.assert_array_equal(np.array([r.bbox for r in regs]).reshape(-1, 4), g[prefix + "_bbox"])
    assert [getattr(r, "cells", -1) for r in regs] == list(g[prefix + "_cells"])


@pytest.mark.parametrize("name", FUNC_CASES)
def test_function_level(ta, name, tmp_path):
    from particle_col_image_segmentation_amd.refine_boundaries import refine_boundaries
    g = load_golden(name)
    ct = _cell_types(g)
    cm = ta.median_filter(g["class_map"], size=ta.DENOISE_SIZE)
    np.testing.assert_array_equal(cm, g["denoised"])
    if "crash" in g.files:
        with pytest.raises(ValueError, match="cannot convert float NaN to integer"):
            ta.get_cell_positions_and_areas(cm, ct, merged=True)
        return
    np.testing.assert_array_equal(ta.label(cm), g["label_im"])
    cell_pos, cell_clusters, pa, merged = ta.get_cell_positions_and_areas(cm, ct, merged=True)
    assert pa == int(g["particle_area"])
    assert sorted(cell_pos) == sorted(str(t) for t in g["types_pos"])
    for t in cell_pos:
        _check_regions(cell_pos[t], g, "pos_" + t)
        _check_regions(cell_clusters[t], g, "clu_" + t)
        for r in cell_pos[t][:2]:
            assert int(ta.get_type(r, cm)) == [k for k, v in ct.items() if v == t][0]
            assert tuple(r.coords[0]) == r.first
    assert sorted(merged) == [str(t) for t in g["types_merged"]]
    for t, groups in merged.items():
        check_merged(groups, g, t)
    _, _, _, none = ta.get_cell_positions_and_areas(cm, ct)
    assert none == {}
    mr, images = ta.get_cell_clusters_from_distances(cm, cell_pos, cell_clusters, ct)
    for t, img in images.items():
        np.testing.assert_array_equal(img, g["mimg_" + t])
        check_merged(mr[t], g, t)
    first_type = next(iter(cell_pos))
    val = [k for k, v in ct.items() if v == first_type][0]
    groups, img = ta.get_merged_regions(cm == val, cell_pos[first_type] + cell_clusters[first_type])
    np.testing.assert_array_equal(img, g["mimg_" + first_type])
    check_merged(groups, g, first_type)
    cnt, dens, ratio = ta.get_cell_counts_and_densities(cell_pos, cell_clusters, pa)
    for i, k in enumerate(g["cnt_keys"]):
        assert cnt[str(k)] == int(g["cnt"][i])
        assert dens[str(k)] == float(g["dens"][i])
        assert ratio[str(k)] == float(g["ratio"][i])
    before = cm.copy()
    rec, pa2 = ta.recreate_particle_area(cm, ct, pa)
    np.testing.assert_array_equal(cm, before)  # input untouched (:1010)
    np.testing.assert_array_equal(rec, g["recreated"])
    assert pa2 == int(g["particle_area2"])
    plabel = [k for k, v in ct.items() if v == "Particle"][0]
    upd, ov = ta.fill_particle_area(cm, plabel, 1, plabel)
    np.testing.assert_array_equal(upd, g["fill1"])
    assert ov == int(g["fill1_area"])
    # CSV text, byte for byte
    p1, p2, p3 = (str(tmp_path / n) for n in ("a_cell_pos.csv", "a_merged_cell_pos.csv", "a_density.csv"))
    ta.write_cell_position_info(cell_pos, cell_clusters, p1, pa2)
    exp_pos = g["csv_pos"].tobytes()
    got_pos = open(p1, "rb").read()
    assert sorted(got_pos.split(b"\r\n")) == sorted(exp_pos.split(b"\r\n"))  # type order is insertion order in both
    assert got_pos == exp_pos
    ta.write_merged_cell_position_info({k: merged[k] for k in sorted(merged) if k != "combined"}, p2, pa2)
    exp_rows = [r for r in g["csv_merged"].tobytes().split(b"\r\n") if not r.startswith(b"combined")]
    assert open(p2, "rb").read().split(b"\r\n") == exp_rows
    ta.write_merged_cell_position_info({"combined": merged["combined"]}, p2, pa2)
    exp_comb = sorted(r for r in g["csv_merged"].tobytes().split(b"\r\n") if r.startswith(b"combined"))
    assert sorted(r for r in open(p2, "rb").read().split(b"\r\n") if r.startswith(b"combined")) == exp_comb
    ta.write_density_info(p3, "folderA", dens, ratio, cnt)
    ta.write_density_info(p3, "folderB", dens, ratio, cnt)
    ta.write_density_info(p3, "folderA", dens, ratio, cnt)
    assert open(p3, "rb").read() == g["csv_density"].tobytes()
    # refine_boundaries.py stages
    rf = refine_boundaries(g["stack"][3], return_stages=True)
    np.testing.assert_array_equal(rf["binary_mask"], g["rf_mask"])
    np.testing.assert_array_equal(rf["distance"], g["rf_distance"])
    np.testing.assert_array_equal(rf["local_max"], g["rf_local_max"])
    np.testing.assert_array_equal(rf["markers"], g["rf_markers"])
    np.testing.assert_array_equal(rf["labels"], g["rf_labels"])
    np.testing.assert_array_equal(refine_boundaries(g["stack"][3]), g["rf_labels"])


def test_unmapped_class_raises_keyerror(ta):
    g = load_golden("func_64_s1")
    with pytest.raises(KeyError):
        ta.get_cell_positions_and_areas(g["denoised"], {1: "3D05", 2: "Particle", 3: "Background"})


def test_overlap_removal(ta):
    g = load_golden("overlap")
    for i in range(2):
        out = ta.combine_cell_positions_and_clusters(g["ov_%d_dapi" % i], g["ov_%d_other" % i])
        np.testing.assert_array_equal(out, g["ov_%d_out" % i])


def test_e2e_single_folder(ta, tmp_path):
    """process_single_h5_file on a one-file folder: the three CSVs byte for byte (class map stored as .npy)."""
    g = load_golden("e2e_single")
    folder = tmp_path / "3D05" / "24h" / "Tp_3D05_1_24h_60X_1"
    folder.mkdir(parents=True)
    fname = "Tp_3D05_1_24h_60X_1_Simple Segmentation.npy"
    np.save(str(folder / fname), g["class_map"][:, :, None])
    ta.process_single_h5_file(str(folder), fname)
    for k in g.files:
        if not k.startswith("csv:"):
            continue
        rel = k[4:]
        got = open(os.path.join(str(tmp_path), rel), "rb").read()
        assert got == g[k].tobytes(), rel


def test_e2e_multi_channel_folder(ta, tmp_path):
    """process_multiple_h5_files (DAPI + RFP channels of one sample): every CSV the reference wrote, byte for byte
    (the merged CSV per type, because the reference orders its types by a hash-randomised set)."""
    g = load_golden("e2e_multi")
    folder = tmp_path / "3D05_6B07" / "24h" / "Tp_3D05_6B07_1_24h_60X_1"
    folder.mkdir(parents=True)
    files = []
    for fn, key in zip(g["files"], ("dapi", "rfp")):
        name = str(fn).replace(".h5", ".npy")
        np.save(str(folder / name), g[key][None, :, :])
        files.append(name)
    ta.process_h5_folder(str(folder), files)
    n = 0
    for k in g.files:
        if not k.startswith("csv:"):
            continue
        got = open(os.path.join(str(tmp_path), k[4:]), "rb").read()
        exp = g[k].tobytes()
        if k.endswith("_merged_cell_pos.csv"):
            assert sorted(got.split(b"\r\n")) == sorted(exp.split(b"\r\n")), k
            for t in (b"3D05", b"6B07"):
                pick = lambda txt: [r for r in txt.split(b"\r\n") if r.startswith(t)]
                assert pick(got) == pick(exp)
        else:
            assert got == exp, k
        n += 1
    assert n == 4
