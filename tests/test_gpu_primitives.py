"""HIP kernels (through the C ABI) against the CPU oracle and the golden vectors.
Bit-exact for every integer / label / mask result."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden
from oracle import oracle as orc

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path has no CPU fallback")
    from particle_col_image_segmentation_amd import ops as _ops
    return _ops


def dev(a, dtype=None):
    a = np.ascontiguousarray(a)
    if a.dtype == bool:
        a = a.astype(np.uint8)
    t = torch.from_numpy(a)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def one(a, dtype=None):
    return dev(a, dtype)[None]


def host(t):
    return t.cpu().numpy()


RNG = np.random.default_rng(99)


def test_argmax(ops):
    from particle_col_image_segmentation_amd import synth
    for shape in [(3, 5, 33, 47), (2, 5, 64, 64), (1, 7, 17, 23)]:
        st = RNG.random(shape).astype(np.float32)
        got = host(ops.argmax_planes(dev(st)))
        np.testing.assert_array_equal(got, np.argmax(st, axis=1) + 1)
    st = synth.gen_batch(5, 2, 128, 128)
    np.testing.assert_array_equal(host(ops.argmax_planes(dev(st))), synth.class_map_from_stack(st))


def test_median(ops, primitives):
    for c in golden_cases(primitives, "median"):
        np.testing.assert_array_equal(host(ops.median5(one(c["inp"])))[0], c["out"])
    for (b, h, w, hi) in [(3, 100, 131, 6), (2, 257, 64, 256), (1, 1024, 1024, 6), (2, 70, 300, 17)]:
        a = RNG.integers(0, hi, (b, h, w)).astype(np.uint8)
        got = host(ops.median5(dev(a)))
        for i in range(b):
            np.testing.assert_array_equal(got[i], orc.median_filter(a[i]))


@pytest.mark.parametrize("kind,fn,conn", [("label_eq8", "label_equal8", None), ("label_bool8", "label_bool8", None),
                                          ("label_bool4", "label_bool4", 1)])
def test_label_golden(ops, primitives, kind, fn, conn):
    for c in golden_cases(primitives, kind):
        lab, cnt = getattr(ops, fn)(one(c["inp"]))
        np.testing.assert_array_equal(host(lab)[0], c["out"])
        assert int(cnt[0]) == int(c["n"])


def test_label_large(ops):
    from particle_col_image_segmentation_amd import synth
    for (b, h, w, k, p) in [(3, 200, 333, 3, 0.6), (2, 512, 512, 2, 0.45), (1, 97, 1030, 4, 0.8), (2, 65, 129, 1, 0.55)]:
        a = (RNG.integers(1, k + 1, (b, h, w)) * (RNG.random((b, h, w)) < p)).astype(np.uint8)
        lab, cnt = ops.label_equal8(dev(a))
        lb8, cb8 = ops.label_bool8(dev(a))
        lb4, cb4 = ops.label_bool4(dev(a))
        for i in range(b):
            exp, n = orc.label(a[i], return_num=True)
            np.testing.assert_array_equal(host(lab)[i], exp)
            assert int(cnt[i]) == n
            exp, n = orc.label(a[i] > 0, return_num=True)
            np.testing.assert_array_equal(host(lb8)[i], exp)
            assert int(cb8[i]) == n
            exp, n = orc.label(a[i] > 0, connectivity=1, return_num=True)
            np.testing.assert_array_equal(host(lb4)[i], exp)
            assert int(cb4[i]) == n
    st = synth.gen_batch(31, 2, 512, 512)
    cm = np.stack([orc.median_filter(c) for c in synth.class_map_from_stack(st)])
    lab, cnt = ops.label_equal8(dev(cm))
    for i in range(2):
        exp, n = orc.label(cm[i], return_num=True)
        np.testing.assert_array_equal(host(lab)[i], exp)
        assert int(cnt[i]) == n
    # one huge spiral component + empty + full frames
    sp = np.zeros((3, 200, 200), np.uint8)
    sp[1] = 1
    r = np.arange(200)
    sp[2][::2, :] = 1
    sp[2][1::4, -1] = 1
    sp[2][3::4, 0] = 1
    lab, cnt = ops.label_bool4(dev(sp))
    for i in range(3):
        exp, n = orc.label(sp[i] > 0, connectivity=1, return_num=True)
        np.testing.assert_array_equal(host(lab)[i], exp)
        assert int(cnt[i]) == n


def test_region_reduce(ops):
    from particle_col_image_segmentation_amd import synth
    for (b, h, w) in [(2, 64, 64), (2, 200, 333), (1, 512, 512)]:
        st = synth.gen_batch(40, b, h, w)
        cm = np.stack([orc.median_filter(c) for c in synth.class_map_from_stack(st)])
        labs = np.stack([orc.label(c) for c in cm])
        counts = torch.tensor([int(l.max()) for l in labs], dtype=torch.int32).cuda()
        stats, cls_out, sums, ovf = ops.region_reduce(dev(labs), counts, dev(cm), dev(st))
        assert int(ovf.sum()) == 0
        for i in range(b):
            n = int(labs[i].max())
            tab = orc.region_table(labs[i])
            np.testing.assert_array_equal(host(stats)[i, :n], tab)
            firsts = tab[:, 7]
            np.testing.assert_array_equal(host(cls_out)[i, :n], cm[i].ravel()[firsts])
            np.testing.assert_allclose(host(sums)[i, :n], orc.channel_sums(labs[i], st[i]), rtol=1e-12, atol=0)
    # capacity overflow is reported, rows below cap stay correct
    labs = np.stack([orc.label(RNG.random((50, 60)) < 0.3)])
    stats, _, _, ovf = ops.region_reduce(dev(labs), cap=5)
    assert int(ovf[0]) == 1
    np.testing.assert_array_equal(host(stats)[0], orc.region_table(labs[0])[:5])


def test_region_sums2(ops):
    """The fused plane pass: per-label float64 sums of two label images (class components under a class selection, and a
    second, unrestricted labelling) in one read of the planes, against the oracle's per-image sums."""
    from particle_col_image_segmentation_amd import synth
    for (b, h, w) in [(2, 64, 64), (2, 200, 332), (1, 512, 512), (3, 33, 68)]:
        st = synth.gen_batch(40, b, h, w)
        cm = np.stack([orc.median_filter(c) for c in synth.class_map_from_stack(st)])
        la = np.stack([orc.label(c) for c in cm])
        lb = np.stack([orc.label(RNG.random((h, w)) < 0.55) for _ in range(b)])
        ca = torch.tensor([int(l.max()) for l in la], dtype=torch.int32).cuda()
        cb = torch.tensor([int(l.max()) for l in lb], dtype=torch.int32).cuda()
        sel = (1 << 1) | (1 << 2)
        stats_a, _, sums_a, _ = ops.region_reduce(dev(la), ca, dev(cm), zero_sums=5)
        # image B: once with its integer columns made by the plane-free pass, once accumulated by the fused pass itself
        stats_b, _, sums_b, _ = ops.region_reduce(dev(lb), cb, zero_sums=5)
        ops.region_sums2(dev(la), dev(cm), sel, sums_a, dev(lb), sums_b, dev(st))
        stats_b2, sums_b2, ovf_b2 = ops.region_init(cb, stats_b.shape[1], 5, (b, h, w), stats_b.device)
        sums_a2 = torch.zeros_like(sums_a)
        ops.region_sums2(dev(la), dev(cm), sel, sums_a2, dev(lb), sums_b2, dev(st), stats_b=stats_b2, overflow_b=ovf_b2)
        assert int(ovf_b2.sum()) == 0
        for i in range(b):
            nb = int(lb[i].max())
            np.testing.assert_array_equal(host(stats_b2)[i, :nb], host(stats_b)[i, :nb])
            np.testing.assert_allclose(host(sums_b2)[i, :nb], host(sums_b)[i, :nb], rtol=1e-12, atol=0)
        for i in range(b):
            na, nb = int(la[i].max()), int(lb[i].max())
            np.testing.assert_array_equal(host(stats_a)[i, :na], orc.region_table(la[i]))
            np.testing.assert_array_equal(host(stats_b)[i, :nb], orc.region_table(lb[i]))
            first = orc.region_table(la[i])[:, 7]
            is_sel = np.isin(cm[i].ravel()[first], [1, 2])
            exp_a = orc.channel_sums(la[i], st[i])
            np.testing.assert_allclose(host(sums_a)[i, :na][is_sel], exp_a[is_sel], rtol=1e-12, atol=0)
            assert not host(sums_a)[i, :na][~is_sel].any()
            np.testing.assert_allclose(host(sums_b)[i, :nb], orc.channel_sums(lb[i], st[i]), rtol=1e-12, atol=0)


def test_edt(ops, primitives):
    for c in golden_cases(primitives, "edt"):
        d2 = host(ops.edt_sq(one(c["inp"])))[0]
        np.testing.assert_array_equal(np.sqrt(d2.astype(np.float64)), c["out"])
    for (b, h, w, p) in [(3, 100, 131, 0.97), (2, 257, 300, 0.999), (2, 33, 1025, 0.9), (1, 1024, 1024, 0.9995),
                         (2, 40, 40, 1.0), (2, 31, 65, 0.5), (1, 9, 12001, 0.9998), (1, 5, 7, 0.8), (1, 3, 2, 0.5)]:
        # (12001 columns: a staged row no longer fits four to a block -- the one-row form of the horizontal pass; 7 and 2
        # columns: every read of the search is a clamped one)
        m = RNG.random((b, h, w)) < p
        d2 = host(ops.edt_sq(dev(m)))
        dc = host(ops.edt_sq(dev(m), cap=50))
        for i in range(b):
            exp = orc.edt_sq(m[i])
            np.testing.assert_array_equal(d2[i], exp)
            np.testing.assert_array_equal(dc[i], np.minimum(exp, 51))


def test_edt_fused_threshold(ops):
    img = RNG.random((2, 120, 77)).astype(np.float32)
    img[0, 5, 5] = 0.5
    d2, mask = ops.edt_sq_lt(dev(img), 0.5)
    np.testing.assert_array_equal(host(mask).astype(bool), img < 0.5)
    np.testing.assert_array_equal(host(ops.threshold_lt(dev(img), 0.5)).astype(bool), img < 0.5)
    for i in range(2):
        np.testing.assert_array_equal(host(d2)[i], orc.edt_sq(img[i] < 0.5))


def test_dilate(ops, primitives):
    for c in golden_cases(primitives, "dilate"):
        x = c["inp"].astype(np.uint8) * 3
        for rad in (2, 5, 20):
            got = host(ops.dilate_disk(one(x), 1 << 3, rad))[0]
            np.testing.assert_array_equal(got.astype(bool), c["r%d" % rad])
    z = RNG.integers(1, 6, (2, 90, 140)).astype(np.uint8) * (RNG.random((2, 90, 140)) < 0.02)
    got = host(ops.dilate_disk(dev(z.astype(np.uint8)), (1 << 1) | (1 << 2), 2))
    for i in range(2):
        np.testing.assert_array_equal(got[i].astype(bool), orc.binary_dilation_disk((z[i] == 1) | (z[i] == 2), 2))


def test_fill_particle(ops):
    from particle_col_image_segmentation_amd import synth
    st = synth.gen_batch(60, 3, 128, 160)
    cm = np.stack([orc.median_filter(c) for c in synth.class_map_from_stack(st)])
    cm[2][cm[2] == 3] = 5  # a frame without any particle pixel (scipy's degenerate EDT)
    cm[2][0, 0] = 1
    cm[2][0, 1] = 1
    cm[2][1, 0] = 1
    out, area = ops.fill_particle(dev(cm), 3, 1, 3, 20, 2)
    for i in range(3):
        exp, ov = orc.fill_particle_area(cm[i], 3, 1, 3)
        np.testing.assert_array_equal(host(out)[i], exp)
        assert int(area[i]) == ov
    assert int(area[2]) == 2


def test_fill_holes(ops, primitives):
    for c in golden_cases(primitives, "fill"):
        np.testing.assert_array_equal(host(ops.fill_holes(one(c["inp"])))[0].astype(bool), c["out"])
    m = RNG.random((2, 150, 210)) < 0.58
    got = host(ops.fill_holes(dev(m)))
    for i in range(2):
        np.testing.assert_array_equal(got[i].astype(bool), orc.binary_fill_holes(m[i]))


def test_local_maxima(ops, primitives):
    for c in golden_cases(primitives, "locmax"):
        d2 = orc.edt_sq(c["mask"])
        is_max, markers, cnt = ops.local_maxima(one(d2))
        np.testing.assert_array_equal(host(is_max)[0].astype(bool), c["out"])
        np.testing.assert_array_equal(host(markers)[0], c["markers"])
        assert int(cnt[0]) == int(c["markers"].max())
    for c in golden_cases(primitives, "locmax_int"):
        is_max, markers, cnt = ops.local_maxima(one(c["inp"]))
        np.testing.assert_array_equal(host(is_max)[0].astype(bool), c["out"])
        np.testing.assert_array_equal(host(markers)[0], orc.label(c["out"]))
    a = RNG.integers(0, 3, (2, 130, 200)).astype(np.int32)
    is_max, markers, cnt = ops.local_maxima(dev(a))
    for i in range(2):
        lm = orc.local_maxima(a[i])
        np.testing.assert_array_equal(host(is_max)[i].astype(bool), lm)
        np.testing.assert_array_equal(host(markers)[i], orc.label(lm))


@pytest.mark.parametrize("mode", [0, 1, 4])
def test_watershed_golden(ops, primitives, mode):
    n_flagged = 0
    for c in golden_cases(primitives, "ws"):
        out, flags = ops.watershed(one(c["img"]), one(c["markers"]), one(c["mask"]), mode=mode)
        np.testing.assert_array_equal(host(out)[0], c["out"])
        n_flagged += int(flags[0])
    if mode in (0, 4):
        assert n_flagged > 0  # quantised cases must have gone through the exact path


@pytest.mark.parametrize("shape", [(64, 64), (65, 129), (130, 257), (193, 64), (300, 500), (31, 700)])
def test_watershed_odd_shapes(ops, shape):
    """Frame sizes around the tile edges: the relaxation alternates between the 64-aligned tiling and the one shifted by
    half a tile (one more tile per axis, partial tiles on every side); labels must match the oracle bit for bit."""
    from particle_col_image_segmentation_amd import synth
    H, W = shape
    st = synth.gen_batch(300 + H + W, 2, H, W)
    bm = np.ascontiguousarray(st[:, 3])
    refs = [orc.refine_boundaries(b) for b in bm]
    mk = np.stack([r["markers"] for r in refs])
    ms = np.stack([r["binary_mask"] for r in refs])
    out, _ = ops.watershed(dev(bm), dev(mk), dev(ms), mode=0)
    for i in range(2):
        np.testing.assert_array_equal(host(out)[i], refs[i]["labels"])


def test_watershed_many_seeds(ops):
    """A wider net for the concurrent parts (sweeps racing inside a tile, alternating tilings, speculative rounds):
    48 independent frames in one batch, every one compared with the oracle."""
    from particle_col_image_segmentation_amd import synth
    st = synth.gen_batch(9000, 48, 200, 264)
    bm = np.ascontiguousarray(st[:, 3])
    refs = [orc.refine_boundaries(b) for b in bm]
    mk = np.stack([r["markers"] for r in refs])
    ms = np.stack([r["binary_mask"] for r in refs])
    for _ in range(2):  # twice: scheduling differs from run to run, the result must not
        out, _ = ops.watershed(dev(bm), dev(mk), dev(ms), mode=0)
        got = host(out)
        for i in range(len(refs)):
            np.testing.assert_array_equal(got[i], refs[i]["labels"])


def test_watershed_exact_path_deep_heap(ops):
    """Mode 1 = the heap emulation alone, on quantised frames large enough for the heap to outgrow its LDS levels
    (4095 slots) so that the workspace levels, the 5-level descents and long sift-ups are all exercised."""
    from particle_col_image_segmentation_amd import synth
    st = synth.gen_batch(81, 2, 448, 576, ties=True)
    bm = np.ascontiguousarray(st[:, 3])
    bm[1] = np.round(bm[1] * 6) / 6  # very few levels: huge plateaus, heap of tens of thousands of entries
    refs = [orc.refine_boundaries(b) for b in bm]
    mk = np.stack([r["markers"] for r in refs])
    ms = np.stack([r["binary_mask"] for r in refs])
    out, _ = ops.watershed(dev(bm), dev(mk), dev(ms), mode=1)
    for i in range(2):
        np.testing.assert_array_equal(host(out)[i], refs[i]["labels"])
    # a flood from a handful of seeds over a flat image: the queue holds whole BFS fronts of one value
    flat = np.full((1, 300, 700), 0.25, np.float32)
    mk1 = np.zeros((1, 300, 700), np.int32)
    for k, (r, c) in enumerate([(5, 5), (150, 350), (299, 699), (20, 600), (280, 30)]):
        mk1[0, r, c] = k + 1
    ms1 = np.ones((1, 300, 700), bool)
    ms1[0, 100:200, 340] = False
    out, _ = ops.watershed(dev(flat), dev(mk1), dev(ms1), mode=1)
    np.testing.assert_array_equal(host(out)[0], orc.watershed(flat[0], mk1[0], ms1[0]))


def test_watershed_parallel_path_is_proven(ops):
    """Tie-free frames: the parallel flood alone (mode 2) must already be exact wherever it says so."""
    from particle_col_image_segmentation_amd import synth
    st = synth.gen_batch(70, 3, 192, 256)
    bm = np.ascontiguousarray(st[:, 3])
    refs = [orc.refine_boundaries(b) for b in bm]
    mk = np.stack([r["markers"] for r in refs])
    ms = np.stack([r["binary_mask"] for r in refs])
    out, flags = ops.watershed(dev(bm), dev(mk), dev(ms), mode=2)
    out_v, flags_v = ops.watershed(dev(bm), dev(mk), dev(ms), mode=6)  # + explicit per-pixel proof check
    assert torch.equal(flags, flags_v) and torch.equal(out, out_v)
    proven = 0
    for i in range(3):
        if int(flags[i]) == 0:
            np.testing.assert_array_equal(host(out)[i], refs[i]["labels"])
            proven += 1
    assert proven >= 2
    out, _ = ops.watershed(dev(bm), dev(mk), dev(ms), mode=0)
    for i in range(3):
        np.testing.assert_array_equal(host(out)[i], refs[i]["labels"])


def test_refine_chain_batch(ops):
    from particle_col_image_segmentation_amd import synth
    for ties in (False, True):
        st = synth.gen_batch(80, 2, 128, 128, ties=ties)
        bm = np.ascontiguousarray(st[:, 3])
        d2, mask = ops.edt_sq_lt(dev(bm), 0.5)
        is_max, markers, cnt = ops.local_maxima(d2)
        labels, flags = ops.watershed(dev(bm), markers, mask)
        for i in range(2):
            ref = orc.refine_boundaries(bm[i])
            np.testing.assert_array_equal(host(mask)[i].astype(bool), ref["binary_mask"])
            np.testing.assert_array_equal(np.sqrt(host(d2)[i].astype(np.float64)), ref["distance"])
            np.testing.assert_array_equal(host(is_max)[i].astype(bool), ref["local_max"])
            np.testing.assert_array_equal(host(markers)[i], ref["markers"])
            np.testing.assert_array_equal(host(labels)[i], ref["labels"])


def test_classmap_label_fused_equals_separate_kernels(ops):
    """pcseg_classmap_label_f32 (argmax + median + union-find tile pass in one kernel) against the three separate calls
    and the oracle: full-width tiles, ragged widths / heights, 3-, 4-, 5- and 7-plane stacks (7: the unfused route)."""
    from particle_col_image_segmentation_amd import synth
    for (H, W, C) in ((160, 192, 5), (97, 132, 5), (64, 130, 5), (33, 64, 4), (70, 257, 3), (40, 72, 7), (5, 3, 5), (130, 64, 2)):
        st = synth.gen_batch(70, 2, H, W)
        if C <= 5:
            st = np.ascontiguousarray(st[:, :C])
        else:
            st = np.ascontiguousarray(np.concatenate([st, st[:, :C - 5] * 0.5], axis=1))
        cls = ops.argmax_planes(dev(st))
        z_ref = ops.median5(cls)
        lab_ref, cnt_ref = ops.label_equal8(z_ref)
        z, lab, cnt = ops.classmap_label(dev(st))
        assert torch.equal(z, z_ref) and torch.equal(lab, lab_ref) and torch.equal(cnt, cnt_ref), (H, W, C)
        for b in range(2):
            cm = (np.argmax(st[b], axis=0) + 1).astype(np.uint8)
            den = orc.median_filter(cm)
            np.testing.assert_array_equal(host(z)[b], den)
            np.testing.assert_array_equal(host(lab)[b], orc.label(den))


def test_merge_groups(ops):
    from particle_col_image_segmentation_amd import synth
    st = synth.gen_batch(90, 2, 160, 160)
    cm = np.stack([orc.median_filter(c) for c in synth.class_map_from_stack(st)])
    labs = np.stack([orc.label(c) for c in cm])
    counts = torch.tensor([int(l.max()) for l in labs], dtype=torch.int32).cuda()
    stats, cls_out, _, _ = ops.region_reduce(dev(labs), counts, dev(cm))
    dil = ops.dilate_disk(dev(cm), (1 << 1) | (1 << 2), 2)
    dl, _ = ops.label_bool8(dil)
    lists, regs_all = [], []
    for i in range(2):
        regs = [r for r in orc.regionprops(labs[i]) if cm[i][r.first] in (1, 2) and r.area >= 20]
        regs = [r for r in regs if r.area < 200] + [r for r in regs if r.area >= 200]  # cells then clusters (:796)
        regs_all.append(regs)
        lists.append([r.label - 1 for r in regs])
    cap = max(len(l) for l in lists) + 3
    rl = np.full((2, cap), -1, np.int32)
    for i, l in enumerate(lists):
        rl[i, :len(l)] = l
    n_list = torch.tensor([len(l) for l in lists], dtype=torch.int32).cuda()
    group_of, n_groups = ops.merge_groups(dl, stats, dev(rl), n_list)
    # fused path: bit dilation + union-find roots, no numbering pass
    roots = ops.dilated_roots(dev(cm), (1 << 1) | (1 << 2), 2)
    np.testing.assert_array_equal(host(roots) >= 0, host(dil).astype(bool))
    g2, n2 = ops.merge_groups(roots, stats, dev(rl), n_list, roots=True)
    assert torch.equal(g2, group_of) and torch.equal(n2, n_groups)
    # run-based path (what the pipeline uses): no label image at all, components looked up through the bit words
    dbits, run_par = ops.dilated_runs(dev(cm), (1 << 1) | (1 << 2), 2)
    words = host(dbits).view(np.uint32)
    for i in range(2):
        H_, W_ = cm[i].shape
        unpacked = ((words[i][:, None, :] >> np.arange(32, dtype=np.uint32)[None, :, None]) & 1).reshape(-1, W_)[:H_]
        np.testing.assert_array_equal(unpacked.astype(bool), host(dil)[i].astype(bool))
    g3, n3 = ops.merge_groups_runs(dbits, run_par, stats, dev(rl), n_list)
    assert torch.equal(g3, group_of) and torch.equal(n3, n_groups)
    # the one-launch form the pipeline runs (keys + grouping + member sums, lists read in place from the (B, slots, cap)
    # arrays of classify_regions): same groups, and the group rows equal pcseg_group_reduce's
    tab_cap = stats.shape[1]
    rls = np.full((2, 3, tab_cap), -7, np.int32)   # the list sits in slot 1 of 3; the other slots hold garbage
    nls = np.array([[5, 0, 9], [5, 0, 9]], np.int32)
    for i, l in enumerate(lists):
        rls[i, 1, :len(l)] = l
        nls[i, 1] = len(l)
    gf, nf, gsf = ops.merge_groups_fused(dbits, run_par, stats, dev(rls), dev(nls), 1)
    rl_full = np.full((2, tab_cap), -1, np.int32)
    rl_full[:, :min(cap, tab_cap)] = rl[:, :min(cap, tab_cap)]
    g4, n4 = ops.merge_groups_runs(dbits, run_par, stats, dev(rl_full), n_list)
    gs4 = ops.group_reduce(stats, dev(rl_full), n_list, g4, n4, 160, 160)
    assert torch.equal(nf, n_groups)
    for i in range(2):
        k, ng = len(lists[i]), int(n_groups[i])
        assert torch.equal(gf[i, :k], group_of[i, :k])
        assert torch.equal(gsf[i, :ng], gs4[i, :ng])
    for rad in (0, 1, 3, 5):
        rr = ops.dilated_roots(dev(cm), 1 << 1, rad)
        for i in range(2):
            np.testing.assert_array_equal(host(rr)[i] >= 0, orc.binary_dilation_disk(cm[i] == 1, rad))
    for i in range(2):
        groups, _ = orc.get_merged_regions((cm[i] == 1) | (cm[i] == 2), regs_all[i])
        exp = np.zeros(len(lists[i]), np.int32)
        pos = {r.label: k for k, r in enumerate(regs_all[i])}
        for gi, g in enumerate(groups):
            for r in g["regions"]:
                exp[pos[r.label]] = gi + 1
        np.testing.assert_array_equal(host(group_of)[i, :len(lists[i])], exp)
        assert int(n_groups[i]) == len(groups)


def test_remove_overlapping(ops):
    g = load_golden("overlap")
    for i in range(2):
        out = ops.remove_overlapping(one(g["ov_%d_dapi" % i]), one(g["ov_%d_other" % i]), 0.1)
        np.testing.assert_array_equal(host(out)[0], g["ov_%d_out" % i])


def test_extensions_otsu_morph(ops):
    """north_star extensions X1 / X2: no reference call site, so the libraries SURVEY.md 8a names are the pin
    (skimage.filters.threshold_otsu, skimage / scipy 3x3 binary erosion and dilation; tests/golden/extensions.npz),
    plus the oracle on a batch (frames of one launch keep their own [min, max])."""
    g = load_golden("extensions")
    i = 0
    while "otsu_%02d_inp" % i in g.files:
        img = g["otsu_%02d_inp" % i]
        thr, hist, lohi = ops.threshold_otsu(dev(img[None]), return_hist=True)
        assert float(thr[0]) == float(g["otsu_%02d_thr" % i]), i  # bit-exact float32 bin centre
        if "otsu_%02d_hist" % i in g.files:
            np.testing.assert_array_equal(host(hist)[0], g["otsu_%02d_hist" % i])
        assert float(lohi[0, 0]) == img.min() and float(lohi[0, 1]) == img.max()
        i += 1
    assert i >= 6
    i = 0
    while "morph_%02d_inp" % i in g.files:
        m = g["morph_%02d_inp" % i]
        np.testing.assert_array_equal(host(ops.morph3x3(dev(m[None]), 1))[0].astype(bool), g["morph_%02d_erode" % i])
        np.testing.assert_array_equal(host(ops.morph3x3(dev(m[None]), 0))[0].astype(bool), g["morph_%02d_dilate" % i])
        i += 1
    assert i >= 6
    m = RNG.random((2, 50, 70)) < 0.5
    for erode in (0, 1):
        got = host(ops.morph3x3(dev(m), erode))
        for k in range(2):
            np.testing.assert_array_equal(got[k].astype(bool), orc.morph3x3(m[k], erode))
    img = RNG.random((3, 100, 90)).astype(np.float32)
    img[1] *= 37.0
    img[2] = 0.25
    thr, hist, lohi = ops.threshold_otsu(dev(img), return_hist=True)
    h2, lohi2 = ops.otsu_hist(dev(img))
    assert torch.equal(hist, h2) and torch.equal(lohi, lohi2)
    for k in range(3):
        t, h = orc.threshold_otsu(img[k])
        assert float(thr[k]) == t
        np.testing.assert_array_equal(host(hist)[k], h)


def test_watershed_proof_holds_on_adversarial_ties(ops):
    """Property behind the parallel path: whenever it reports a frame as proven (tie flag 0, mode 2 = no exact
    fallback) the labels equal the reference's sequential flood -- on images built to be full of ties, plateaus, lakes
    and equal-valued seeds.  Batches of frames so that flagged and proven frames mix inside one launch."""
    rng = np.random.default_rng(2024)
    proven = flagged = 0
    for (B, H, W, levels, pm, pk) in [(48, 24, 31, 3, 0.9, 0.03), (48, 33, 33, 6, 1.0, 0.02), (32, 40, 70, 12, 0.8, 0.01),
                                      (32, 64, 64, 50, 0.95, 0.004), (24, 70, 130, 0, 0.9, 0.003), (16, 96, 96, 200, 1.0, 0.002)]:
        img = rng.random((B, H, W)).astype(np.float32)
        if levels:
            img = (np.floor(img * levels) / levels).astype(np.float32)
        # smooth a little so that lakes (pits without seeds) exist at every level
        img = ((img + np.roll(img, 1, 1) + np.roll(img, 1, 2)) / 3).astype(np.float32) if levels != 3 else img
        mask = rng.random((B, H, W)) < pm
        mk = np.zeros((B, H, W), np.int32)
        for b in range(B):
            sel = rng.random((H, W)) < pk
            mk[b][sel] = rng.permutation(int(sel.sum())).astype(np.int32) + 1
            if b % 3 == 0:
                mk[b, 1:3, 1:4] = 500  # a multi-pixel marker
        out, flags = ops.watershed(dev(img), dev(mk), dev(mask), mode=2)
        out_v, flags_v = ops.watershed(dev(img), dev(mk), dev(mask), mode=6)
        full, _ = ops.watershed(dev(img), dev(mk), dev(mask), mode=0)
        out, flags, out_v, flags_v, full = host(out), host(flags), host(out_v), host(flags_v), host(full)
        np.testing.assert_array_equal(flags, flags_v)  # component test == explicit per-pixel proof check
        for b in range(B):
            exp = orc.watershed(img[b], mk[b], mask[b])
            np.testing.assert_array_equal(full[b], exp)
            if flags[b] == 0:
                np.testing.assert_array_equal(out[b], exp)
                np.testing.assert_array_equal(out_v[b], exp)
                proven += 1
            else:
                flagged += 1
    assert proven > 20 and flagged > 20, (proven, flagged)


def test_threshold_rows_skip_chunks_out_of_reach(ops):
    """Wide rows (18 chunks of 64 columns, a ragged last one): the threshold pass of the EDT skips chunks that no
    interval starts in or reaches -- a compact particle, isolated points at chunk borders and an empty frame."""
    H, W = 70, 1100
    cm = np.full((4, H, W), 1, np.uint8)
    yy, xx = np.mgrid[:H, :W]
    cm[0][(yy - 30) ** 2 + (xx - 500) ** 2 <= 15 ** 2] = 3      # one compact particle
    for x in (0, 63, 64, 127, 128, 640, 1087, 1099):             # single particle pixels at chunk borders
        cm[1][(7 * x) % H, x] = 3
    cm[2][:, 1090:] = 3                                          # a particle in the ragged last chunk only
    cm[3][5, 700] = 2                                            # no particle pixel at all
    cm[:, ::9, ::5] = 4                                          # pixels that are neither cell nor particle
    out, area = ops.fill_particle(dev(cm), 3, 1, 3, 20, 2)
    for i in range(4):
        exp, ov = orc.fill_particle_area(cm[i], 3, 1, 3)
        np.testing.assert_array_equal(host(out)[i], exp)
        assert int(area[i]) == ov
    for rad in (3, 40, 70):
        got = host(ops.dilate_disk(dev(cm), 1 << 3, rad))
        for i in range(4):
            np.testing.assert_array_equal(got[i].astype(bool), orc.binary_dilation_disk(cm[i] == 3, rad))


def test_threshold_on_bit_words_shapes_and_radii(ops):
    """The threshold epilogues with a reach of at most 31 pixels run on the bit words (reach_bits_kernel): ragged widths
    (byte path), widths of one, two and three column blocks, heights that end inside a bit word, radii 0 .. 31 and the
    first radius that takes the row-block pass again (32), zero pixels at block and word borders, a frame without any."""
    for (h, w, p) in [(33, 45, 0.02), (95, 450, 0.004), (64, 1347, 0.001), (40, 900, 0.002), (130, 452, 0.0005), (5, 3, 0.3)]:
        z = (RNG.random((3, h, w)) < p).astype(np.uint8) * 3
        z[1] = 0
        for (r, c) in [(0, 0), (h - 1, w - 1), (31 % h, 447 % w), (32 % h, 448 % w), (h // 2, (w // 2) & ~3)]:
            z[1][r, c] = 3
        z[2] = 1  # no zero pixel of the set at all
        for rad in (0, 1, 2, 7, 20, 31, 32):
            got = host(ops.dilate_disk(dev(z), 1 << 3, rad))
            for i in range(3):
                np.testing.assert_array_equal(got[i].astype(bool), orc.binary_dilation_disk(z[i] == 3, rad), err_msg="%s r=%d f=%d" % ((h, w), rad, i))
        cm = np.where(z == 3, 3, RNG.integers(1, 3, z.shape)).astype(np.uint8)
        cm[2, 0, :2] = 1
        out, area = ops.fill_particle(dev(cm), 3, 1, 3, 20, 2)
        for i in range(3):
            exp, ov = orc.fill_particle_area(cm[i], 3, 1, 3)
            np.testing.assert_array_equal(host(out)[i], exp)
            assert int(area[i]) == ov
