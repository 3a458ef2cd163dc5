"""Batched per-frame hot path on one GPU: frames (B,5,H,W) -> label masks + ROI tables.

One ``FramePipeline.run`` is the full chain of SURVEY.md section 8a for a batch
that is already resident in HBM:

    class map (argmax+1) -> median5 (A1) -> label (A2) -> region table + isotope sums (A3, M1)
    -> classification / cluster cell counts (A3 tail, A4) -> proximity merge per type + combined (A5, A6)
    -> particle-area reconstruction per cell class (A8) -> [counts / densities: host epilogue (A9)]
    -> boundary refinement: threshold + EDT + local maxima + markers + watershed (R1-R4, W1)
    -> per-ROI isotope sums of the refined ROIs (M1)

Everything stays on the device; the only host synchronisations are the
watershed's convergence polls.  ``tables()`` downloads the result as plain numpy
tables (the per-ROI table format of this build; the reference's CSV writers take
the per-frame drop-in objects instead, see ``tiff_analysis``).
"""
import numpy as np
import torch

from . import ops
from . import tiff_analysis as ta
from .synth import BOUNDARY_PLANE, CELL_TYPES_5

# ratio presets for the isotope planes: (name, numerator plane, denominator planes) -- .m:136-139
RATIOS_7 = (("C13act", 1, (1, 0)), ("N15act", 3, (2, 3)), ("O17act", 5, (6, 5, 4)), ("O18act", 6, (6, 5, 4)))
RATIOS_5 = (("C13act", 1, (1, 0)), ("N15act", 3, (2, 3)))


class BatchResult(dict):
    """Device tensors of one batch (see FramePipeline.run for the keys).

    ``FramePipeline.run`` returns as soon as both kernel chains are enqueued; reading any entry waits (once) for the two
    events that close them, so callers never see unfinished tensors, while a loop that only calls ``run`` keeps the
    next batch's class-map chain running under this batch's watershed tail."""

    _pending = None

    def synchronize(self):
        pending, self._pending = self._pending, None
        if pending is None:
            return self
        if hasattr(pending, "result"):  # a lane's future: (entries, closing events); re-raises what the lane raised
            entries, pending = pending.result()
            dict.update(self, entries)
        for ev in pending:
            ev.synchronize()
        return self

    def __getitem__(self, key):
        if self._pending:
            self.synchronize()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if self._pending:
            self.synchronize()
        return dict.get(self, key, default)

    def items(self):
        if self._pending:
            self.synchronize()
        return dict.items(self)

    def values(self):
        if self._pending:
            self.synchronize()
        return dict.values(self)

    def keys(self):
        if self._pending:
            self.synchronize()
        return dict.keys(self)

    def __iter__(self):
        if self._pending:
            self.synchronize()
        return dict.__iter__(self)

    def __contains__(self, key):
        if self._pending:
            self.synchronize()
        return dict.__contains__(self, key)

    def __len__(self):
        if self._pending:
            self.synchronize()
        return dict.__len__(self)

    def check(self):
        """Raise for the conditions the reference raises for / the tables cannot hold."""
        if int(self["overflow"].sum().item()) or int(self["ws_overflow"].sum().item()):
            raise RuntimeError("region table capacity exceeded: raise FramePipeline(cap=...)")
        if int(self["nan_flag"].sum().item()):
            # tiff_analysis.py:776-781: clusters of a type without any single cell -> int(NaN)
            raise ValueError("cannot convert float NaN to integer")
        return self


_LANES = {}  # (device index, lanes) -> (executors, stream pairs): shared by every pipeline of the process


def _lanes_for(device, lanes):
    key = (device.index, lanes)
    if key not in _LANES:
        from concurrent.futures import ThreadPoolExecutor
        pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix="pcseg-lane%d" % i) for i in range(lanes)]
        streams = [(torch.cuda.Stream(device=device, priority=0), torch.cuda.Stream(device=device, priority=-1))
                   for _ in range(lanes)]
        _LANES[key] = (pools, streams)
    return _LANES[key]


class FramePipeline:
    def __init__(self, cell_types=None, threshold=0.5, boundary_plane=BOUNDARY_PLANE, cap=None, merged=True,
                 watershed_mode=0, overlap=True, lanes=2):
        self.cell_types = dict(cell_types or CELL_TYPES_5)
        self.tables_ = ops.ClassTables(self.cell_types, ta.CELL_TYPES, ta.MIN_CELL_AREA, ta.MIN_CLUSTER_AREA)
        self.threshold = float(threshold)
        self.boundary_plane = int(boundary_plane)
        self.cap = cap
        self.merged = merged
        self.watershed_mode = watershed_mode
        self.overlap = overlap
        self.lanes = max(1, int(lanes))
        self._lane_pool = None  # one single-thread executor + stream pair per lane, made on first use
        self._lane_streams = None
        self._step = 0

    def run(self, stack):
        if stack.dim() != 4 or stack.dtype != torch.float32 or not stack.is_cuda:
            raise TypeError("stack must be a (B, C, H, W) float32 CUDA tensor")
        stack = stack.contiguous()
        res = BatchResult()
        res["shape"] = tuple(stack.shape)
        if not self.overlap:
            self._class_chain(stack, res)
            self._refine_chain(stack, res)
            return res
        # The class-map chain and the boundary-refinement chain only share the input: each gets its own HIP stream (the
        # refinement chain, whose fixed points poll the host, the higher priority).  Consecutive batches alternate between
        # `lanes` host threads with a stream pair each, and `run` returns at once: while one batch sits in the
        # latency-bound tail of its watershed (small launches, host round trips) the next batch's dense kernels keep the
        # GPU busy.  The result object waits for its lane the first time an entry is read.
        if self._lane_pool is None or self._lane_streams[0][0].device != stack.device:
            self._lane_pool, self._lane_streams = _lanes_for(stack.device, self.lanes)
        lane = self._step % self.lanes
        self._step += 1
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        res._pending = self._lane_pool[lane].submit(self._run_lane, lane, stack, ready, dict(res))
        return res

    def synchronize(self):
        """Wait until every batch handed to ``run`` so far is finished (lanes drained, device idle)."""
        for pool in self._lane_pool or ():
            pool.submit(lambda: None).result()
        torch.cuda.synchronize()

    def _run_lane(self, lane, stack, ready, entries):
        torch.cuda.set_device(stack.device)
        out = BatchResult(entries)  # built privately: the caller's object only receives it on synchronize()
        s1, s2 = self._lane_streams[lane]
        s1.wait_event(ready)
        s2.wait_event(ready)
        with torch.cuda.stream(s1):
            self._class_chain(stack, out)
        with torch.cuda.stream(s2):
            self._refine_chain(stack, out)
        done = (torch.cuda.Event(), torch.cuda.Event())
        done[0].record(s1)
        done[1].record(s2)
        stack.record_stream(s1)
        stack.record_stream(s2)
        return dict(out), done

    def _class_chain(self, stack, res):
        B, C, H, W = stack.shape
        cap = self.cap or max(1024, (H * W) // 64)
        tb = self.tables_
        # ---- class map + denoise (A1)
        cls = ops.argmax_planes(stack)
        z = ops.median5(cls)
        res["denoised"] = z
        # ---- label + region table (+ isotope sums of the class components) (A2, A3, M1)
        labels, counts = ops.label_equal8(z)
        # isotope sums are only ever reported for cell / cluster regions: the planes are read under those classes only
        cell_bits = 0
        for v in tb.cell_values:
            cell_bits |= 1 << int(v)
        stats, cls_out, cc_sums, overflow = ops.region_reduce(labels, counts, cls=z, planes=stack, cap=cap, sum_classes=cell_bits)
        res.update(labels=labels, counts=counts, stats=stats, cls_out=cls_out, cc_sums=cc_sums, overflow=overflow)
        # ---- classification, cluster cell counts, region lists (A3 tail, A4)
        res.update(ops.classify_regions(stats, cls_out, counts, tb))
        # ---- proximity merge per cell type and combined (A5, A6)
        if self.merged:
            n_slots = len(tb.slot_names)
            bits_all = 0
            groups = {}
            for s in list(range(n_slots)) + [4]:
                if s < n_slots:
                    bits = 1 << tb.slot_value[s]
                    bits_all |= bits
                else:
                    bits = bits_all
                if bits == 0:
                    continue
                # dilated components as union-find roots on the 1-bit image: grouping only needs "same component"
                dl = ops.dilated_roots(z, bits, ta.CELL_CLUSTER_DISTANCE_THRESHOLD // 2)
                lst = res["region_list"][:, s].contiguous()
                nl = res["n_list"][:, s].contiguous()
                gof, ng = ops.merge_groups(dl, stats, lst, nl, roots=True)
                gst = ops.group_reduce(stats, lst, nl, gof, ng, H, W)
                groups[s] = {"group_of": gof, "n_groups": ng, "group_stats": gst}
            res["groups"] = groups
        # ---- particle-area reconstruction, one fill per cell class on the previous output (A8)
        ds = z
        overlap = torch.zeros((B,), dtype=torch.int64, device=stack.device)
        if tb.particle_value is not None:
            for v in tb.cell_values:
                ds, overlap = ops.fill_particle(ds, tb.particle_value, v, tb.particle_value, ta.DILATION_RADIUS,
                                                ta.DISTANCE_THRESHOLD, overlap)
        res["recreated"] = ds
        res["overlap_area"] = overlap

    def _refine_chain(self, stack, res):
        B, C, H, W = stack.shape
        cap = self.cap or max(1024, (H * W) // 64)
        # ---- boundary refinement (R1-R4, W1) on the boundary plane, read in place
        bm = stack[:, self.boundary_plane]
        d2, mask = ops.edt_sq_lt(bm, self.threshold)
        _, markers, n_markers = ops.local_maxima(d2, want_mask=False)
        ws_labels, tie_flags = ops.watershed(bm, markers, mask, mode=self.watershed_mode)
        res.update(mask=mask, markers=markers, n_markers=n_markers, ws_labels=ws_labels, tie_flags=tie_flags)
        # ---- isotope sums of the refined ROIs (M1)
        ws_stats, _, ws_sums, ws_overflow = ops.region_reduce(ws_labels, n_markers, planes=stack, cap=cap)
        res.update(ws_stats=ws_stats, ws_sums=ws_sums, ws_overflow=ws_overflow)

    # ------------------------------------------------------------------ host epilogue
    def tables(self, res, frame_ids=None, ratios=RATIOS_5, distances=False, raster=19.0, check=True):
        """Download one batch as numpy tables: ``cells`` (one row per cell / cluster region), ``rois`` (one row per
        refined ROI), ``frames`` (one row per frame) and ``groups`` (one row per merged group).  ``check=False`` skips
        ``BatchResult.check`` (a caller that has looked at the flags itself, e.g. to keep the ROI rows of a batch in
        which the reference would have raised on one frame's cluster statistics)."""
        if check:
            res.check()
        else:
            res.synchronize()
        B, C, H, W = res["shape"]
        frame_ids = np.arange(B) if frame_ids is None else np.asarray(frame_ids)
        tb = self.tables_
        h = lambda k: res[k].cpu().numpy()
        counts, stats, cls_out, cc_sums = h("counts"), h("stats"), h("cls_out"), h("cc_sums")
        kind, slot_of, cells = h("kind"), h("slot_of"), h("cells")
        pa, ovl, tstats = h("particle_area"), h("overlap_area"), h("type_stats")
        n_ws, ws_stats, ws_sums, ties = h("n_markers"), h("ws_stats"), h("ws_sums"), h("tie_flags")
        group_of = {}
        if self.merged:
            for s, g in res["groups"].items():
                group_of[s] = (g["group_of"].cpu().numpy(), g["n_groups"].cpu().numpy(), g["group_stats"].cpu().numpy(),
                               res["region_list"][:, s].cpu().numpy(), res["n_list"][:, s].cpu().numpy())
        nr = len(ratios)
        cell_rows, roi_rows, frame_rows, group_rows, dist_rows = [], [], [], [], []
        for b in range(B):
            n = int(counts[b])
            st, sums = stats[b, :n], cc_sums[b, :n]
            own = np.zeros(n, np.int64)
            comb = np.zeros(n, np.int64)
            for s, (gof, ng, gst, lst, nl) in group_of.items():
                k = int(nl[b])
                tgt = comb if s == 4 else own
                tgt[lst[b, :k]] = gof[b, :k]
                for gi in range(int(ng[b])):
                    t = gst[b, gi]
                    group_rows.append([frame_ids[b], s, gi + 1, t[0], t[1] / t[0], t[2] / t[0], t[3], t[4], t[5], t[6], t[7]])
            sel = np.nonzero(kind[b, :n] > 0)[0]
            for r in sel:
                a = float(st[r, 0])
                cell_rows.append([frame_ids[b], r + 1, cls_out[b, r], kind[b, r], st[r, 0], st[r, 1] / a, st[r, 2] / a,
                                  st[r, 3], st[r, 4], st[r, 5], st[r, 6], cells[b, r], own[r], comb[r]]
                                 + list(sums[r]) + _ratios(sums[r], ratios))
            m = int(n_ws[b])
            for r in range(m):
                a = float(ws_stats[b, r, 0])
                if a == 0:
                    continue
                roi_rows.append([frame_ids[b], r + 1, ws_stats[b, r, 0], ws_stats[b, r, 1] / a, ws_stats[b, r, 2] / a]
                                + list(ws_sums[b, r]) + _ratios(ws_sums[b, r], ratios))
            row = [frame_ids[b], n, m, pa[b], pa[b] + ovl[b], ties[b]]
            pa_um = pa[b] / (ta.PX_TO_UM_CONV ** 2)
            for s in range(len(tb.slot_names)):
                ncell, nclu, sumcell, first = tstats[b, s]
                present = first != 0x7FFFFFFF
                clu = (kind[b, :n] == 2) & (slot_of[b, :n] == s)
                count = int(ncell + cells[b, :n][clu].sum())
                area = (sumcell + st[clu, 0].sum()) / (ta.PX_TO_UM_CONV ** 2)
                with np.errstate(all="ignore"):
                    dens = round(count / pa_um, 5) if present and pa_um else float("nan")
                    ratio = round(area / pa_um, 5) if present and pa_um else float("nan")
                row += [int(present), count, dens, ratio]
            if distances:
                # .m:260-268 per frame: nearest ROI of the other cell type for the cells / clusters of slots 0 and 1
                a = [[st[r, 2] / st[r, 0] + 1.0, st[r, 1] / st[r, 0] + 1.0] for r in sel if slot_of[b, r] == 0]
                c = [[st[r, 2] / st[r, 0] + 1.0, st[r, 1] / st[r, 0] + 1.0] for r in sel if slot_of[b, r] == 1]
                if a and c:
                    dev = res["stats"].device
                    ta_ = torch.tensor(a, dtype=torch.float64, device=dev)
                    tc_ = torch.tensor(c, dtype=torch.float64, device=dev)
                    da, dc = ops.nearest_dist(ta_, tc_).cpu().numpy(), ops.nearest_dist(tc_, ta_).cpu().numpy()
                    ia = [r for r in sel if slot_of[b, r] == 0]
                    ic = [r for r in sel if slot_of[b, r] == 1]
                    for r, d in list(zip(ia, da)) + list(zip(ic, dc)):
                        dist_rows.append([frame_ids[b], r + 1, d / (512.0 / raster)])
            frame_rows.append(row)
        ncols_cell = 14 + C + nr
        ncols_roi = 5 + C + nr
        return {
            "cells": np.array(cell_rows, np.float64).reshape(-1, ncols_cell),
            "cells_columns": ["frame", "label", "class", "kind", "area", "centroid_row", "centroid_col", "min_row", "min_col",
                              "max_row1", "max_col1", "cells", "group", "group_combined"]
                             + ["S%d" % k for k in range(C)] + [r[0] for r in ratios],
            "rois": np.array(roi_rows, np.float64).reshape(-1, ncols_roi),
            "rois_columns": ["frame", "label", "area", "centroid_row", "centroid_col"]
                            + ["S%d" % k for k in range(C)] + [r[0] for r in ratios],
            "frames": np.array(frame_rows, np.float64).reshape(B, -1),
            "frames_columns": ["frame", "n_labels", "n_rois", "particle_area", "particle_area_recreated", "tie_flag"]
                              + [c % n for n in tb.slot_names for c in ("%s_present", "%s_count", "%s_density", "%s_area_ratio")],
            "distances": np.array(dist_rows, np.float64).reshape(-1, 3),
            "distances_columns": ["frame", "label", "nearest_other_type_um"],
            "groups": np.array(group_rows, np.float64).reshape(-1, 11),
            "groups_columns": ["frame", "slot", "group", "area", "centroid_row", "centroid_col", "min_row", "min_col",
                               "max_row1", "max_col1", "members"],
        }


def _ratios(s, ratios):
    out = []
    for _, num, den in ratios:
        d = 0.0
        for k in den:
            d = d + s[k]
        with np.errstate(all="ignore"):
            out.append(float(np.float64(s[num]) / np.float64(d)) if len(s) > max(den + (num,)) else float("nan"))
    return out
