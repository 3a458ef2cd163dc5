#!/opt/conda/bin/python3.9
"""Generate tests/golden/*.npz by running the REAL reference.

Run only in the build container (never on the GPU box):

    cd /tmp && MPLBACKEND=Agg /opt/conda/bin/python3.9 -B \
        /root/repo/tests/golden/make_golden.py

It imports ``/root/reference/tiff_analysis.py`` (and executes
``refine_boundaries.py`` verbatim through ``runpy``) under the oracle
interpreter (numpy 1.26.4 / scipy 1.7.1 / scikit-image 0.18.3) and stores the
inputs and every returned value as plain arrays (``allow_pickle=False`` loads
them).  Nothing of the reference's source is written anywhere: the fixtures are
data only.
"""
import importlib.util
import io
import json
import os
import runpy
import shutil
import sys
import tempfile
import warnings

import numpy as np

warnings.filterwarnings("ignore")
os.environ.setdefault("MPLBACKEND", "Agg")

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)

import scipy  # noqa: E402
import skimage  # noqa: E402
from scipy import ndimage as ndi  # noqa: E402
from skimage import measure, morphology  # noqa: E402
from skimage.morphology import binary_dilation, disk  # noqa: E402
from skimage.segmentation import watershed  # noqa: E402

import tiff_analysis as ta  # noqa: E402  (the reference)

spec = importlib.util.spec_from_file_location(
    "pcseg_synth", os.path.join(REPO, "particle_col_image_segmentation_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synth)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    clean = {}
    for k, v in arrays.items():
        v = np.asarray(v)
        assert v.dtype != object, (name, k)
        clean[k] = v
    np.savez_compressed(path, **clean)
    print("wrote", path, os.path.getsize(path))


# ---------------------------------------------------------------------------
# primitive-level goldens
# ---------------------------------------------------------------------------
def prim_cases():
    rng = np.random.default_rng(1234)
    out = {}
    n = 0

    def add(kind, **kw):
        nonlocal n
        for k, v in kw.items():
            out["%s_%02d_%s" % (kind, n, k)] = v
        n += 1

    # ---- A1: median_filter(size=5) (tiff_analysis.py:122,643)
    for (h, w, lo, hi) in [(17, 23, 1, 6), (64, 64, 1, 6), (5, 5, 1, 4), (3, 7, 1, 6),
                           (1, 9, 1, 6), (33, 65, 0, 256), (40, 31, 0, 9)]:
        a = rng.integers(lo, hi, (h, w)).astype(np.uint8)
        add("median", inp=a, out=ndi.median_filter(a, size=5))
    # structured class map with salt noise
    st = synth.gen_frame(7, 64, 64)
    cm = synth.class_map_from_stack(st)
    add("median", inp=cm, out=ndi.median_filter(cm, size=5))

    # ---- A2: label (equal-valued 8-conn; bool 8-conn) + ndimage 4-conn
    for (h, w, k, p) in [(17, 23, 3, 0.5), (64, 64, 4, 0.6), (32, 32, 2, 0.3),
                         (9, 40, 5, 0.9), (1, 1, 2, 1.0), (8, 8, 1, 0.0)]:
        a = (rng.integers(1, k + 1, (h, w)) * (rng.random((h, w)) < p)).astype(np.uint8)
        lab = measure.label(a)
        add("label_eq8", inp=a, out=lab.astype(np.int32), n=lab.max())
        b = a > 0
        lb = measure.label(b)
        add("label_bool8", inp=b, out=lb.astype(np.int32), n=lb.max())
        l4, n4 = ndi.label(b)
        add("label_bool4", inp=b, out=l4.astype(np.int32), n=n4)
    lab = measure.label(cm)
    add("label_eq8", inp=cm, out=lab.astype(np.int32), n=lab.max())
    # serpentine / spiral worst cases for union-find
    sp = np.zeros((33, 33), bool)
    for r in range(0, 33, 2):
        sp[r, :] = True
        if (r // 2) % 2 == 0 and r + 1 < 33:
            sp[r + 1, -1] = True
        elif r + 1 < 33:
            sp[r + 1, 0] = True
    lb = measure.label(sp)
    add("label_bool8", inp=sp, out=lb.astype(np.int32), n=lb.max())

    # ---- A3: regionprops (area, centroid, bbox, coords[0])
    for (h, w, k, p) in [(24, 31, 3, 0.55), (64, 64, 3, 0.7)]:
        a = (rng.integers(1, k + 1, (h, w)) * (rng.random((h, w)) < p)).astype(np.uint8)
        lab = measure.label(a)
        props = measure.regionprops(lab)
        add("props", inp=a, labels=lab.astype(np.int32),
            label=np.array([r.label for r in props], np.int64),
            area=np.array([r.area for r in props], np.int64),
            centroid=np.array([r.centroid for r in props], np.float64).reshape(-1, 2),
            bbox=np.array([r.bbox for r in props], np.int64).reshape(-1, 4),
            first=np.array([r.coords[0] for r in props], np.int64).reshape(-1, 2),
            typ=np.array([ta.get_type(r, a) for r in props], np.int64))

    # ---- A6/A8: binary_dilation(disk(r)) and distance_transform_edt
    for (h, w, p) in [(17, 23, 0.1), (64, 64, 0.02), (50, 70, 0.5), (30, 30, 0.0),
                      (45, 45, 0.002), (12, 80, 0.05)]:
        m = rng.random((h, w)) < p
        add("dilate", inp=m, r2=binary_dilation(m, disk(2)), r20=binary_dilation(m, disk(20)),
            r5=binary_dilation(m, disk(5)))
    for (h, w, p) in [(17, 23, 0.9), (64, 64, 0.97), (50, 70, 0.5), (4, 5, 1.0), (30, 30, 1.0),
                      (30, 30, 0.0), (1, 17, 0.8), (19, 1, 0.8), (80, 90, 0.995), (70, 64, 0.9995)]:
        m = rng.random((h, w)) < p
        add("edt", inp=m, out=ndi.distance_transform_edt(m))
    blob = synth.gen_frame(11, 64, 64)[3] < 0.5
    add("edt", inp=blob, out=ndi.distance_transform_edt(blob))

    # ---- A7: binary_fill_holes
    for (h, w, p) in [(17, 23, 0.6), (64, 64, 0.55), (30, 30, 0.8), (20, 20, 0.0), (20, 20, 1.0)]:
        m = rng.random((h, w)) < p
        add("fill", inp=m, out=ndi.binary_fill_holes(m))
    ring = np.zeros((20, 20), bool)
    ring[3:15, 4:16] = True
    ring[5:13, 6:14] = False
    ring[8:10, 9:11] = True
    ring[14, 10] = False  # 4-connected leak? (diagonal-only gaps must NOT leak)
    add("fill", inp=ring, out=ndi.binary_fill_holes(ring))
    ring2 = ring.copy()
    ring2[14, 10] = True
    ring2[14, 9] = False
    ring2[13, 10] = False  # diagonal gap only
    add("fill", inp=ring2, out=ndi.binary_fill_holes(ring2))

    # ---- R3/R4: local_maxima + label on EDT maps and on small-int plateaus
    for (h, w, p) in [(17, 23, 0.8), (64, 64, 0.9), (40, 50, 0.97), (30, 30, 1.0), (30, 30, 0.0)]:
        m = rng.random((h, w)) < p
        d = ndi.distance_transform_edt(m)
        lm = morphology.local_maxima(d)
        add("locmax", mask=m, dist=d, out=lm, markers=measure.label(lm).astype(np.int32))
    d = ndi.distance_transform_edt(blob)
    lm = morphology.local_maxima(d)
    add("locmax", mask=blob, dist=d, out=lm, markers=measure.label(lm).astype(np.int32))
    for (h, w, k) in [(17, 23, 3), (40, 40, 4), (25, 60, 2), (12, 12, 1)]:
        a = rng.integers(0, k, (h, w)).astype(np.int32)
        lm = morphology.local_maxima(a)
        add("locmax_int", inp=a, out=lm)

    # ---- W1: watershed (4-conn, mask, markers)
    def ws_case(h, w, q, pm, pk, tag):
        img = rng.random((h, w)).astype(np.float32)
        if q:
            img = (np.round(img * q) / q).astype(np.float32)
        mask = rng.random((h, w)) < pm
        mk = np.zeros((h, w), np.int32)
        sel = rng.random((h, w)) < pk
        mk[sel] = rng.permutation(int(sel.sum())).astype(np.int32) + 1
        # a few multi-pixel markers
        if h > 8 and w > 8:
            mk[2:4, 2:5] = 999
        add("ws", img=img, markers=mk, mask=mask, q=q,
            out=watershed(img, mk, mask=mask).astype(np.int32))

    for (h, w) in [(17, 23), (32, 32), (48, 40)]:
        for q in (0, 0, 5, 20, 100):
            for pm, pk in [(0.85, 0.02), (1.0, 0.01), (0.6, 0.05)]:
                ws_case(h, w, q, pm, pk, "")
    # the refine chain on synthetic frames, tie-free and quantised
    for seed, ties in [(3, False), (4, True)]:
        st = synth.gen_frame(seed, 64, 64, ties=ties)
        bm = st[3]
        mask = bm < 0.5
        d = ndi.distance_transform_edt(mask)
        lm = morphology.local_maxima(d)
        mk = measure.label(lm)
        add("ws", img=bm, markers=mk.astype(np.int32), mask=mask, q=100 if ties else 0,
            out=watershed(bm, mk, mask=mask).astype(np.int32))
    return out


# ---------------------------------------------------------------------------
# function-level goldens (the reference's own functions)
# ---------------------------------------------------------------------------
def regions_to_arrays(prefix, regs, out):
    out[prefix + "_label"] = np.array([r.label for r in regs], np.int64)
    out[prefix + "_area"] = np.array([r.area for r in regs], np.int64)
    out[prefix + "_centroid"] = np.array([r.centroid for r in regs], np.float64).reshape(-1, 2)
    out[prefix + "_bbox"] = np.array([r.bbox for r in regs], np.int64).reshape(-1, 4)
    out[prefix + "_cells"] = np.array([getattr(r, "cells", -1) for r in regs], np.int64)


def merged_to_arrays(prefix, groups, out):
    out[prefix + "_area"] = np.array([g["area"] for g in groups], np.int64)
    out[prefix + "_centroid"] = np.array([g["centroid"] for g in groups], np.float64).reshape(-1, 2)
    out[prefix + "_bbox"] = np.array([g["bbox"] for g in groups], np.int64).reshape(-1, 4)
    off = [0]
    mem = []
    for g in groups:
        mem.extend(r.label for r in g["regions"])
        off.append(len(mem))
    out[prefix + "_members"] = np.array(mem, np.int64)
    out[prefix + "_offsets"] = np.array(off, np.int64)


def func_case(name, seed, H, W, ties, cell_types):
    out = {}
    stack = synth.gen_frame(seed, H, W, ties=ties)
    cm = synth.class_map_from_stack(stack)
    if max(cell_types) == 4:  # 4-class variant: boundary joins background
        cm = np.where(cm >= 4, 4, cm).astype(np.uint8)
    if max(cell_types) == 3:  # single strain: both cell planes -> 1, particle 2, rest 3
        cm = np.where(cm <= 2, 1, np.where(cm == 3, 2, 3)).astype(np.uint8)
    out["stack"] = stack
    out["class_map"] = cm
    out["ct_keys"] = np.array(list(cell_types.keys()), np.int64)
    out["ct_vals"] = np.array(list(cell_types.values()))
    den = ndi.median_filter(cm, size=ta.DENOISE_SIZE)
    out["denoised"] = den
    out["label_im"] = measure.label(den).astype(np.int32)
    try:
        cell_pos, cell_clusters, particle_area, merged = ta.get_cell_positions_and_areas(den, cell_types, merged=True)
    except ValueError as e:  # the reference's latent int(NaN) crash (tiff_analysis.py:776-781)
        out["crash"] = np.array(str(e))
        save(name, **out)
        return
    out["particle_area"] = np.int64(particle_area)
    out["types_pos"] = np.array(list(cell_pos.keys()))
    for t, regs in cell_pos.items():
        regions_to_arrays("pos_" + t, regs, out)
    for t, regs in cell_clusters.items():
        regions_to_arrays("clu_" + t, regs, out)
    out["types_merged"] = np.array(sorted(merged.keys()))
    for t, groups in merged.items():
        merged_to_arrays("mrg_" + t, groups, out)
    mr, mi = ta.get_cell_clusters_from_distances(den, cell_pos, cell_clusters, cell_types)
    for t, img in mi.items():
        out["mimg_" + t] = img
    cnt, dens, ratio = ta.get_cell_counts_and_densities(cell_pos, cell_clusters, particle_area)
    out["cnt_keys"] = np.array(list(cnt.keys()))
    out["cnt"] = np.array([cnt[k] for k in cnt], np.int64)
    out["dens"] = np.array([dens[k] for k in cnt], np.float64)
    out["ratio"] = np.array([ratio[k] for k in cnt], np.float64)
    rec, pa2 = ta.recreate_particle_area(den, cell_types, particle_area)
    out["recreated"] = rec
    out["particle_area2"] = np.int64(pa2)
    # single fill_particle_area call
    plabel = [k for k, v in cell_types.items() if v == "Particle"][0]
    upd, ov = ta.fill_particle_area(den, plabel, 1, plabel)
    out["fill1"] = upd
    out["fill1_area"] = np.int64(ov)
    # CSV texts through the reference's writers
    tmp = tempfile.mkdtemp()
    p1 = os.path.join(tmp, "a_cell_pos.csv")
    p2 = os.path.join(tmp, "a_merged_cell_pos.csv")
    p3 = os.path.join(tmp, "a_density.csv")
    ta.write_cell_position_info(cell_pos, cell_clusters, p1, pa2)
    merged_sorted = {k: merged[k] for k in sorted(merged)}
    ta.write_merged_cell_position_info(merged_sorted, p2, pa2)
    ta.write_density_info(p3, "folderA", dens, ratio, cnt)
    ta.write_density_info(p3, "folderB", dens, ratio, cnt)
    ta.write_density_info(p3, "folderA", dens, ratio, cnt)  # replace-rows semantics
    for key, p in (("csv_pos", p1), ("csv_merged", p2), ("csv_density", p3)):
        with open(p, "rb") as f:
            out[key] = np.frombuffer(f.read(), np.uint8)
    shutil.rmtree(tmp)

    # refine_boundaries.py executed verbatim on probabilities.h5 (channel 3)
    import h5py
    import matplotlib.pyplot as plt
    plt.show = lambda *a, **k: None
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "working_folder"))
    with h5py.File(os.path.join(tmp, "working_folder",
                                "Tp_C3M10_1_120h_60X_RFP_GFP_1_MIP_probabilities.h5"), "w") as f:
        f["exported_data"] = stack
    cwd = os.getcwd()
    os.chdir(tmp)
    sink = io.StringIO()
    so = sys.stdout
    sys.stdout = sink
    try:
        g = runpy.run_path(os.path.join(REF, "refine_boundaries.py"))
    finally:
        sys.stdout = so
        os.chdir(cwd)
        plt.close("all")
    shutil.rmtree(tmp)
    out["rf_mask"] = g["binary_mask"]
    out["rf_distance"] = g["distance"]
    out["rf_local_max"] = g["local_max"]
    out["rf_markers"] = g["markers"].astype(np.int32)
    out["rf_labels"] = g["labels"].astype(np.int32)
    save(name, **out)


def overlap_case():
    """combine_cell_positions_and_clusters (tiff_analysis.py:252-287)."""
    rng = np.random.default_rng(77)
    out = {}
    for i, (h, w) in enumerate([(48, 48), (64, 80)]):
        a = synth.class_map_from_stack(synth.gen_frame(100 + i, h, w))
        b = synth.class_map_from_stack(synth.gen_frame(200 + i, h, w))
        dapi = np.where(a <= 2, 1, np.where(a == 3, 2, 3)).astype(np.uint8)
        other = np.where((b == 1) | ((a <= 2) & (rng.random((h, w)) < 0.3)), 1,
                         np.where(b == 3, 2, 3)).astype(np.uint8)
        dapi = ndi.median_filter(dapi, size=5)
        other = ndi.median_filter(other, size=5)
        so = sys.stdout
        sys.stdout = io.StringIO()
        try:
            res = ta.combine_cell_positions_and_clusters(dapi, other)
        finally:
            sys.stdout = so
        out["ov_%d_dapi" % i] = dapi
        out["ov_%d_other" % i] = other
        out["ov_%d_out" % i] = res
    save("overlap", **out)


def e2e_case():
    """process_h5_folder on a one-file folder (tiff_analysis.py:85-90, 627-671)."""
    import h5py
    H = W = 96
    for seed in range(21, 80):  # first seed the reference does not crash on (int(NaN), :776-781)
        stack = synth.gen_frame(seed, H, W)
        cm = synth.class_map_from_stack(stack)
        cm = np.where(cm == 2, 1, cm)          # single strain 3D05
        cm = np.where(cm == 3, 2, np.where(cm >= 4, 3, cm)).astype(np.uint8)
        den = ndi.median_filter(cm, size=5)
        try:
            ta.get_cell_positions_and_areas(den, {1: "3D05", 2: "Particle", 3: "Background"})
        except ValueError:
            continue
        break
    tmp = tempfile.mkdtemp()
    folder = os.path.join(tmp, "3D05", "24h", "Tp_3D05_1_24h_60X_1")
    os.makedirs(folder)
    fname = "Tp_3D05_1_24h_60X_1_Simple Segmentation.h5"
    with h5py.File(os.path.join(folder, fname), "w") as f:
        f["exported_data"] = cm[:, :, None]
    so = sys.stdout
    sys.stdout = io.StringIO()
    try:
        ta.process_h5_folder(folder, [fname])
    finally:
        sys.stdout = so
    out = {"class_map": cm, "h5_shape": np.array(cm[:, :, None].shape)}
    listing = []
    for root, _, files in os.walk(tmp):
        for fn in sorted(files):
            rel = os.path.relpath(os.path.join(root, fn), tmp)
            listing.append(rel)
            if fn.endswith(".csv"):
                with open(os.path.join(root, fn), "rb") as f:
                    out["csv:" + rel] = np.frombuffer(f.read(), np.uint8)
    out["listing"] = np.array(sorted(listing))
    shutil.rmtree(tmp)
    save("e2e_single", **out)


def e2e_multi_case():
    """process_h5_folder on a two-file folder -> process_multiple_h5_files (tiff_analysis.py:92-222): DAPI (6B07) + RFP
    (3D05) channels, overlap removal, channel recombination, merged clusters.  The matplotlib figure builders are
    stubbed out at run time (they only write PNGs); every CSV comes from the reference's own code."""
    import h5py
    for fn in ("create_channel_plots", "visualize_dapi_overlap_results", "plot_original_vs_merged", "create_plot"):
        setattr(ta, fn, lambda *a, **k: None)
    H = W = 128
    rng = np.random.default_rng(99)
    for seed in range(400, 480):
        stack = synth.gen_frame(seed, H, W)
        cm = synth.class_map_from_stack(stack)
        rfp = np.where(cm == 1, 1, np.where(cm == 3, 2, 3)).astype(np.uint8)
        extra = (cm == 1) & (ndi.label(cm == 1)[0] % 3 == 0)      # a third of the 3D05 cells also light up in DAPI
        dapi = np.where((cm == 2) | extra, 1, np.where(cm == 3, 2, 3)).astype(np.uint8)
        tmp = tempfile.mkdtemp()
        folder = os.path.join(tmp, "3D05_6B07", "24h", "Tp_3D05_6B07_1_24h_60X_1")
        os.makedirs(folder)
        files = ["Tp_3D05_6B07_1_24h_60X_1_DAPI_Simple Segmentation.h5", "Tp_3D05_6B07_1_24h_60X_1_RFP_Simple Segmentation.h5"]
        for fn, arr in zip(files, (dapi, rfp)):
            with h5py.File(os.path.join(folder, fn), "w") as f:
                f["exported_data"] = arr[None, :, :]
        so = sys.stdout
        sys.stdout = io.StringIO()
        try:
            ta.process_h5_folder(folder, files)
            ok = True
        except ValueError:
            ok = False
        finally:
            sys.stdout = so
        if not ok:
            shutil.rmtree(tmp)
            continue
        out = {"dapi": dapi, "rfp": rfp, "files": np.array(files), "seed": np.int64(seed)}
        for root, _, fs in os.walk(tmp):
            for fn in sorted(fs):
                if fn.endswith(".csv"):
                    rel = os.path.relpath(os.path.join(root, fn), tmp)
                    with open(os.path.join(root, fn), "rb") as f:
                        out["csv:" + rel] = np.frombuffer(f.read(), np.uint8)
        shutil.rmtree(tmp)
        save("e2e_multi", **out)
        return
    raise RuntimeError("no seed worked")


def split_case():
    """split_zstack.process_tif / process_folder (split_zstack.py:38-89)."""
    import tifffile
    import split_zstack as sz
    out = {}
    rng = np.random.default_rng(5)
    for ci, (shape, name) in enumerate([((3, 4, 8, 9), "Tp_3D05_CY5_RFP_GFP_DAPI_1_zstack.tif"),
                                        ((2, 5, 8, 9), "Tp_3D05_RFP_GFP_2_zstack.tif"),
                                        ((2, 2, 6, 7), "Tp_6B07_RFP_GFP_3_mip.tif")]):
        tmp = tempfile.mkdtemp()
        sub = os.path.join(tmp, "top", "day1")
        os.makedirs(sub)
        arr = rng.integers(0, 65535, shape).astype(np.uint16)
        tifffile.imwrite(os.path.join(sub, name), arr)
        with open(os.path.join(sub, name), "rb") as f:
            out["sp_%d_tif" % ci] = np.frombuffer(f.read(), np.uint8)
        so = sys.stdout
        sys.stdout = io.StringIO()
        try:
            sz.process_folder(os.path.join(tmp, "top"), [1, 2])
        finally:
            sys.stdout = so
        listing = []
        for root, dirs, files in os.walk(tmp):
            for d in dirs:
                listing.append(os.path.relpath(os.path.join(root, d), tmp) + "/")
            for fn in files:
                rel = os.path.relpath(os.path.join(root, fn), tmp)
                listing.append(rel)
                if "_z" in fn and fn.endswith(".tif"):
                    out["sp_%d_px:%s" % (ci, rel)] = tifffile.imread(os.path.join(root, fn))
        out["sp_%d_arr" % ci] = arr
        out["sp_%d_name" % ci] = np.array(name)
        out["sp_%d_listing" % ci] = np.array(sorted(listing))
        shutil.rmtree(tmp)
    save("split_zstack", **out)


# ---------------------------------------------------------------------------
# north_star extensions without a reference call site (refine_boundaries.py:22 imports ``filters`` and never calls
# it): pinned against the libraries SURVEY.md 8a names as their oracle -- skimage.filters.threshold_otsu (X1) and the
# 3x3 binary erosion / dilation of skimage.morphology / scipy.ndimage (X2).
# ---------------------------------------------------------------------------
def ext_cases():
    from skimage import filters
    from skimage.morphology import binary_erosion, square
    rng = np.random.default_rng(4321)
    out = {}
    images = []
    images.append(rng.random((100, 90)).astype(np.float32))                                   # uniform noise
    images.append(np.concatenate([rng.normal(0.2, 0.05, 4000), rng.normal(0.7, 0.1, 2000)]).astype(np.float32).reshape(60, 100))
    images.append(np.full((17, 9), 0.37, np.float32))                                         # constant image
    two = np.where(rng.random((33, 47)) < 0.3, np.float32(0.125), np.float32(0.875)).astype(np.float32)
    images.append(two)                                                                        # two values
    # values that sit exactly on / next to the float32 bin edges of a [0, 1] histogram (1024 candidates for 256 bins)
    edge = (np.arange(1024, dtype=np.float64) / 1024.0).astype(np.float32)
    adv = np.concatenate([edge, np.nextafter(edge, np.float32(2)), np.nextafter(edge, np.float32(-1)), [np.float32(1.0)]])
    images.append(np.resize(adv, (64, 49)).astype(np.float32))
    images.append((rng.random((40, 40)) * 2000.0 - 700.0).astype(np.float32))                 # negative values, wide range
    images.append(synth.gen_frame(11, 128, 128)[3])                                           # a boundary-probability plane
    images.append(synth.gen_frame(12, 96, 160, ties=True)[3])                                 # quantised to k/100
    for i, img in enumerate(images):
        thr = filters.threshold_otsu(img)
        out["otsu_%02d_inp" % i] = img
        out["otsu_%02d_thr" % i] = np.float64(thr)
        if not np.all(img == img.ravel()[0]):  # (a constant image returns before any histogram is made)
            from skimage.exposure import histogram as sk_histogram
            out["otsu_%02d_hist" % i] = sk_histogram(img, 256)[0].astype(np.int64)
        out["otsu_%02d_thr_is_f32" % i] = np.bool_(np.asarray(thr).dtype == np.float32)
    masks = [rng.random((50, 70)) < 0.5, rng.random((31, 33)) < 0.8, np.ones((9, 12), bool), np.zeros((7, 5), bool),
             rng.random((1, 40)) < 0.6, rng.random((40, 1)) < 0.6, rng.random((64, 64)) < 0.2]
    se = np.ones((3, 3), bool)
    for i, m in enumerate(masks):
        out["morph_%02d_inp" % i] = m
        # scikit-image's own conventions (binary.py): erosion treats the outside as True, dilation as False
        out["morph_%02d_erode" % i] = binary_erosion(m, square(3))
        out["morph_%02d_dilate" % i] = binary_dilation(m, square(3))
        assert np.array_equal(out["morph_%02d_erode" % i], ndi.binary_erosion(m, structure=se, border_value=True))
        assert np.array_equal(out["morph_%02d_dilate" % i], ndi.binary_dilation(m, structure=se))
    return out


def main():
    meta = {
        "python": sys.version.split()[0],
        "numpy": np.__version__,
        "scipy": scipy.__version__,
        "skimage": skimage.__version__,
        "reference": "ssilverman16/particle_col_image_segmentation @ /root/reference",
    }
    if sys.argv[1:] == ["extensions"]:  # only the X1 / X2 fixtures (added after the others; leaves them untouched)
        save("extensions", **ext_cases())
        return
    with open(os.path.join(HERE, "VERSIONS.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    save("extensions", **ext_cases())
    save("primitives", **prim_cases())
    ct3 = {1: "3D05", 2: "Particle", 3: "Background"}
    ct5 = dict(synth.CELL_TYPES_5)
    ct4 = {1: "3D05", 2: "6B07", 3: "Particle", 4: "Background"}
    func_case("func_64_s1", 1, 64, 64, False, ct5)
    func_case("func_64_s2_ties", 2, 64, 64, True, ct5)
    func_case("func_96x80_s5", 5, 96, 80, False, ct4)
    func_case("func_96x80_s6", 6, 96, 80, False, ct4)
    func_case("func_128_s7_ct3", 7, 128, 128, False, ct3)
    func_case("func_256_s9", 9, 256, 256, False, ct5)
    func_case("func_256_s10_ties", 10, 256, 256, True, ct5)
    overlap_case()
    e2e_case()
    split_case()
    e2e_multi_case()


if __name__ == "__main__":
    main()
