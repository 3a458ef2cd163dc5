"""split_zstack drop-in (host I/O in front of the GPU path) against the tree and pixels the reference produced
(tests/golden/split_zstack.npz: input TIFF bytes written by tifffile, output listing, output pixels)."""
import os

import numpy as np

from conftest import load_golden
from particle_col_image_segmentation_amd import split_zstack as sz
from particle_col_image_segmentation_amd import tiffio


def _listing(root):
    out = []
    for r, dirs, files in os.walk(root):
        for d in dirs:
            out.append(os.path.relpath(os.path.join(r, d), root) + "/")
        for f in files:
            out.append(os.path.relpath(os.path.join(r, f), root))
    return sorted(out)


def test_reader_matches_tifffile_series_shape(tmp_path):
    g = load_golden("split_zstack")
    for ci in range(3):
        p = tmp_path / ("in%d.tif" % ci)
        p.write_bytes(g["sp_%d_tif" % ci].tobytes())
        arr = tiffio.imread(str(p))
        np.testing.assert_array_equal(arr, g["sp_%d_arr" % ci])
        assert arr.dtype == g["sp_%d_arr" % ci].dtype


def test_writer_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for dt in (np.uint8, np.uint16, np.float32, np.int32):
        for shape in ((7, 5), (3, 6, 4), (2, 3, 5, 4)):
            a = (rng.random(shape) * 200).astype(dt)
            p = str(tmp_path / "x.tif")
            tiffio.imwrite(p, a)
            b = tiffio.imread(p)
            np.testing.assert_array_equal(a, b)
            assert b.dtype == a.dtype


def test_process_folder_matches_reference(tmp_path):
    g = load_golden("split_zstack")
    for ci in range(3):
        root = tmp_path / ("case%d" % ci)
        sub = root / "top" / "day1"
        sub.mkdir(parents=True)
        name = str(g["sp_%d_name" % ci])
        (sub / name).write_bytes(g["sp_%d_tif" % ci].tobytes())
        sz.process_folder(str(root / "top"), [1, 2])
        assert _listing(str(root)) == [str(s) for s in g["sp_%d_listing" % ci]]
        for k in g.files:
            pre = "sp_%d_px:" % ci
            if k.startswith(pre) and "_z" in os.path.basename(k):
                got = tiffio.imread(os.path.join(str(root), k[len(pre):]))
                np.testing.assert_array_equal(got, g[k])


def test_five_channel_option(tmp_path):
    """additive: a (Z,5,H,W) isotope stack split with an explicit 5-entry channel map (the reference would treat it
    as 2-channel, split_zstack.py:53-55)."""
    sub = tmp_path / "top" / "d"
    sub.mkdir(parents=True)
    a = np.arange(2 * 5 * 4 * 6, dtype=np.uint16).reshape(2, 5, 4, 6)
    tiffio.imwrite(str(sub / "S_1_zstack.tif"), a)
    cmap = {0: "12C", 1: "13C", 2: "14N12C", 3: "15N12C", 4: "32S"}
    files = sz.process_tif(str(sub / "S_1_zstack.tif"), [0, 1, 2, 3, 4], channel_map=cmap)
    assert len(files) == 10
    frames = sz.load_frames(files).reshape(2, 5, 4, 6)
    np.testing.assert_array_equal(frames, a)
