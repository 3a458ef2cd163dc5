"""Batched per-frame hot path on one GPU: frames (B,5,H,W) -> label masks + ROI tables.

One ``FramePipeline.run`` is the full chain of SURVEY.md section 8a for a batch
that is already resident in HBM:

    class map (argmax+1) -> median5 (A1) -> label (A2) -> region table + isotope sums (A3, M1)
    -> classification / cluster cell counts (A3 tail, A4) -> proximity merge per type + combined (A5, A6)
    -> particle-area reconstruction per cell class (A8) -> [counts / densities: host epilogue (A9)]
    -> boundary refinement: threshold + EDT + local maxima + markers + watershed (R1-R4, W1)
    -> per-ROI isotope sums of the refined ROIs (M1)

Everything stays on the device and no kernel chain waits for the host (the watershed's fixed points finish in
device-side tail kernels), so the whole chain of a batch can be captured once as a hipGraph and replayed
(``FramePipeline(graph=True)``).  ``tables()`` downloads the result as plain numpy tables (the per-ROI table format of
this build; the reference's CSV writers take the per-frame drop-in objects instead, see ``tiff_analysis``).
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
    _slot = None   # graph mode: the capture slot whose static tensors this result aliases ...
    _gen = 0       # ... and the slot's replay count when this result was handed out

    def _check_alive(self):
        if self._slot is not None and self._slot.gen != self._gen:
            raise RuntimeError("this graph-mode result has been overwritten: its lane has replayed a later batch "
                               "(results stay valid until `lanes` further batches have been handed to run())")

    def done_events(self):
        """The events that close this batch's kernel chains (waits for the lane thread to have enqueued them)."""
        pending = self._pending
        if pending is None:
            return []
        if hasattr(pending, "result"):
            return list(pending.result()[1])
        return list(pending)

    def ready(self):
        """True once the batch's kernel chains have finished (never blocks: a batch whose lane has not even enqueued it
        yet is not ready)."""
        pending = self._pending
        if pending is None:
            return True
        if hasattr(pending, "result"):
            if not pending.done():
                return False
            events = pending.result()[1]
        else:
            events = pending
        return all(ev.query() for ev in events)

    def synchronize(self):
        self._check_alive()
        pending = self._pending
        if pending is None:
            return self
        events = pending
        if hasattr(pending, "result"):  # a lane's future: (entries, closing events)
            # a lane that raised keeps raising here on every read: _pending is only cleared after a clean hand-over
            entries, events = pending.result()
            dict.update(self, entries)
        for ev in events:
            ev.synchronize()
        # the tensors were allocated on the lane's streams: tell the caching allocator that the caller's stream uses
        # them too, so that dropping this result cannot hand their blocks to the lane's next batch while a kernel the
        # caller enqueued is still reading them
        if self._slot is None:  # (a graph slot's tensors live in the graph's private pool: nothing to announce)
            cur = torch.cuda.current_stream()
            for t in _tensors(dict.values(self)):
                if t.is_cuda:
                    t.record_stream(cur)
        self._pending = None
        return self

    def __getitem__(self, key):
        self._settle()
        return dict.__getitem__(self, key)

    def _settle(self):
        """every accessor: wait for the batch once, and refuse a graph-mode result whose lane has replayed since"""
        if self._pending:
            self.synchronize()
        elif self._slot is not None:
            self._check_alive()

    def get(self, key, default=None):
        self._settle()
        return dict.get(self, key, default)

    def items(self):
        self._settle()
        return dict.items(self)

    def values(self):
        self._settle()
        return dict.values(self)

    def keys(self):
        self._settle()
        return dict.keys(self)

    def __iter__(self):
        self._settle()
        return dict.__iter__(self)

    def __contains__(self, key):
        self._settle()
        return dict.__contains__(self, key)

    def __len__(self):
        self._settle()
        return dict.__len__(self)

    def check(self):
        """Raise for the conditions the reference raises for / the tables cannot hold."""
        if int(self["overflow"].sum().item()) or int(self["ws_overflow"].sum().item()):
            raise RuntimeError("region table capacity exceeded: raise FramePipeline(cap=...)")
        if int((self["counts"] < 0).sum().item()):
            # a label pass found a union-find entry outside its fence (csrc/common.h, walk_ok) and reported -1 components:
            # the workspace was overwritten while the chain ran (e.g. one captured graph replayed on two streams at once)
            raise RuntimeError("corrupt union-find image in the class-map labelling (counts < 0)")
        if int(self["nan_flag"].sum().item()):
            # tiff_analysis.py:776-781: clusters of a type without any single cell -> int(NaN)
            raise ValueError("cannot convert float NaN to integer")
        return self


def _tensors(values):
    for v in values:
        if isinstance(v, torch.Tensor):
            yield v
        elif isinstance(v, dict):
            yield from _tensors(v.values())


_LANES = {}  # (device index, lanes) -> (executors, stream sets): shared by every pipeline of the process
N_STREAMS = 5  # per lane: class chain, refinement chain (high priority), particle fill, two more merge slots


def _lanes_for(device, lanes):
    key = (device.index, lanes)
    if key not in _LANES:
        from concurrent.futures import ThreadPoolExecutor
        pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix="pcseg-lane%d" % i) for i in range(lanes)]
        streams = [tuple(torch.cuda.Stream(device=device, priority=-1 if k == 1 else 0) for k in range(N_STREAMS))
                   for _ in range(lanes)]
        _LANES[key] = (pools, streams)
    return _LANES[key]


def _download(tables):
    """device tensors -> numpy through PINNED staging buffers (torch caches them): all copies are enqueued, then one
    wait.  A pageable ``.cpu()`` of the ~100 MB ROI table of a dataset costs more than the kernels of a batch."""
    out, staged = {}, []
    for k, v in tables.items():
        if isinstance(v, torch.Tensor) and v.is_cuda:
            buf = torch.empty(tuple(v.shape), dtype=v.dtype, pin_memory=True)
            buf.copy_(v, non_blocking=True)
            staged.append((k, buf))
        elif isinstance(v, torch.Tensor):
            out[k] = v.numpy()
        else:
            out[k] = np.asarray(v)
    if staged:
        torch.cuda.current_stream().synchronize()
        for k, buf in staged:
            out[k] = buf.numpy()
    return out


def _on(stream, *tensors):
    """Tensors made on another stream are about to be read by kernels on ``stream``: tell the caching allocator, so
    that freeing them cannot hand their memory to new work of the producing stream while ``stream`` still reads."""
    for t in tensors:
        if t is not None:
            t.record_stream(stream)


class _GraphSlot:
    """One captured chain: the hipGraph of a lane for one input buffer, the static result tensors it writes (they live in
    the graph's private memory pool) and the stream it is replayed on."""
    __slots__ = ("graph", "entries", "stream", "release", "gen", "stack")

    def __init__(self, stream):
        self.graph, self.entries, self.stream, self.release, self.gen, self.stack = None, None, stream, None, 0, None


class FramePipeline:
    def __init__(self, cell_types=None, threshold=0.5, boundary_plane=BOUNDARY_PLANE, cap=None, merged=True,
                 watershed_mode=0, overlap=True, lanes=None, multi_stream=True, graph=False, max_graphs=None):
        """``graph=True``: the chain of a batch (about a hundred launches on five streams) is captured ONCE per input
        buffer as a hipGraph and replayed for every later batch in that buffer -- one host call per batch instead of a
        hundred, no host thread per lane.  Inputs must then live in a small set of reused device buffers (the graph is
        keyed on the buffer's address and shape; ``max_graphs`` bounds the cache), and a result's tensors are the
        graph's static outputs: they stay valid until ``lanes`` further batches have been handed to :meth:`run`.
        ``lanes``: batches in flight (default 8 host threads in eager mode, 2 replay streams in graph mode)."""
        if lanes is None:
            lanes = 2 if graph else 8
        self.cell_types = dict(cell_types or CELL_TYPES_5)
        self.tables_ = ops.ClassTables(self.cell_types, ta.CELL_TYPES, ta.MIN_CELL_AREA, ta.MIN_CLUSTER_AREA)
        self.threshold = float(threshold)
        self.boundary_plane = int(boundary_plane)
        self.cap = cap
        self.merged = merged
        self.watershed_mode = watershed_mode
        self.overlap = overlap
        self.lanes = max(1, int(lanes))
        self.multi_stream = bool(multi_stream)  # False: the class-map chain (incl. merges and fill) on ONE stream
        self.graph = bool(graph)
        self.max_graphs = int(max_graphs or 4 * self.lanes)
        self._graphs = {}  # (lane, input address, shape) -> _GraphSlot, in insertion order (oldest first)
        self._table_stream = None
        self._lane_pool = None  # one single-thread executor + stream set per lane, made on first use
        self._lane_streams = None
        self._step = 0

    def run(self, stack):
        """Enqueue the full chain for one batch and return at once; reading any entry of the result waits for it.
        ``stack`` must not be modified (or refilled in place) before the result has been read or
        ``result.synchronize()`` has returned: the kernels read it asynchronously."""
        if stack.dim() != 4 or stack.dtype != torch.float32 or not stack.is_cuda:
            raise TypeError("stack must be a (B, C, H, W) float32 CUDA tensor")
        if self.graph:
            return self._run_graph(stack)
        stack = stack.contiguous()
        res = BatchResult()
        res["shape"] = tuple(stack.shape)
        if not self.overlap:
            self._class_stage(stack, res)
            self._merge_all(stack, res)
            self._fill_stage(stack, res)
            self._refine_chain(stack, res)
            self._sums_stage(stack, res)
            return res
        # The chain is a small graph, not a line: the class-map stage (front end, region table, classification), the
        # merge of each cell type and of all types together (they only need the classification), the particle fill (it
        # only needs the denoised map) and the boundary refinement (it only needs the input) run on streams of their
        # own, ordered by events.  Nothing in the chain waits for the host, so one host thread can keep them all fed;
        # consecutive batches still alternate between `lanes` host threads (a stream set each) so that the launch
        # overhead of one batch runs under the kernels of the other.  `run` returns at once; the result object waits
        # for its lane and its closing events the first time an entry is read.
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

    # ------------------------------------------------------------------ hipGraph capture / replay
    def _run_graph(self, stack):
        if not stack.is_contiguous():
            raise ValueError("graph mode needs a contiguous input buffer (the graph is keyed on its address)")
        dev = stack.device
        lane = self._step % self.lanes
        self._step += 1
        key = (lane, stack.data_ptr(), tuple(stack.shape))
        slot = self._graphs.get(key)
        if slot is None:
            slot = self._capture(lane, stack, key)
        cur = torch.cuda.current_stream(dev)
        slot.stream.wait_stream(cur)  # whatever the caller's stream wrote into the buffer
        if slot.release is not None:  # ... and whoever still reads the slot's previous result (tables_device)
            slot.stream.wait_event(slot.release)
            slot.release = None
        done = torch.cuda.Event()
        with torch.cuda.stream(slot.stream):
            # ONE instance of a captured chain at a time: a slot owns one workspace (union-find parents, overflow tables,
            # dirty-tile lists) and one set of static outputs, so two replays of it must never overlap.  Replaying on
            # slot.stream -- and nowhere else -- serialises them; profiles/r03/exp_graph_r3a.log is what the other way
            # round looks like (one graph replayed on 4 streams at once: a memory-access fault).
            if torch.cuda.current_stream(dev) != slot.stream:
                raise RuntimeError("a graph slot may only be replayed on its own stream")
            slot.graph.replay()
            done.record(slot.stream)
        slot.gen += 1
        res = BatchResult(slot.entries)
        res._pending, res._slot, res._gen = [done], slot, slot.gen
        return res

    def _capture(self, lane, stack, key):
        """Capture the lane's five-stream chain for this input buffer (once).  A plain run comes first: one-time work
        inside the library (function attributes, the tile counter's allocation) must not fall into the capture."""
        dev = stack.device
        while len(self._graphs) >= self.max_graphs:
            old = self._graphs.pop(next(iter(self._graphs)))  # oldest first; its memory pool goes with it ...
            old.stream.synchronize()  # ... so no replay of it may still be running (a caller may have dropped its result unread)
            old.gen += 1              # results that still alias its static tensors refuse to be read from now on
        _, lane_streams = _lanes_for(dev, self.lanes)
        streams = lane_streams[lane]
        slot = _GraphSlot(torch.cuda.Stream(device=dev))
        entries = {"shape": tuple(stack.shape)}

        def chain(s_main):
            ready = torch.cuda.Event()
            ready.record(s_main)
            if self.overlap:
                out, done = self._run_streams(streams, stack, ready, dict(entries))
                for ev in done:
                    s_main.wait_event(ev)
                return out
            out = BatchResult(entries)
            self._class_stage(stack, out)
            self._merge_all(stack, out)
            self._fill_stage(stack, out)
            self._refine_chain(stack, out)
            self._sums_stage(stack, out)
            return dict(out)

        slot.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(slot.stream):
            chain(slot.stream)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=slot.stream, capture_error_mode="thread_local"):
            slot.entries = chain(slot.stream)
        slot.graph, slot.stack = graph, stack  # (the buffer stays alive as long as a graph reads it)
        self._graphs[key] = slot
        return slot

    def _merge_slots(self):
        if not self.merged:
            return []
        return [s for s in range(len(self.tables_.slot_names))] + [4]

    def _run_lane(self, lane, stack, ready, entries):
        torch.cuda.set_device(stack.device)
        return self._run_streams(self._lane_streams[lane], stack, ready, entries)

    def _run_streams(self, streams, stack, ready, entries):
        out = BatchResult(entries)  # built privately: the caller's object only receives it on synchronize()
        s_class, s_refine, s_fill, s_m1, s_m2 = streams
        if not self.multi_stream:
            s_fill = s_m1 = s_m2 = s_class
        for st in streams:
            st.wait_event(ready)
            stack.record_stream(st)
        with torch.cuda.stream(s_refine):
            self._refine_chain(stack, out)
        with torch.cuda.stream(s_class):
            self._class_stage(stack, out, denoised_ready=(ev_z := torch.cuda.Event()))
            ev_c = torch.cuda.Event()
            ev_c.record(s_class)
        with torch.cuda.stream(s_refine):
            # the isotope sums of BOTH label images in one pass over the planes, once both are final
            s_refine.wait_event(ev_c)
            _on(s_refine, out["labels"], out["denoised"], out["cc_sums"])
            self._sums_stage(stack, out)
        with torch.cuda.stream(s_fill):
            if s_fill is not s_class:
                s_fill.wait_event(ev_z)
                _on(s_fill, out["denoised"])
            self._fill_stage(stack, out)
        with torch.cuda.stream(s_class):  # (the class-map stream is idle after the classification)
            self._merge_all(stack, out)
        done = []
        for st in streams:
            ev = torch.cuda.Event()
            ev.record(st)
            done.append(ev)
        return dict(out), done

    def _class_stage(self, stack, res, denoised_ready=None):
        B, C, H, W = stack.shape
        cap = self.cap or max(1024, (H * W) // 64)
        tb = self.tables_
        # ---- class map + denoise + label in one fused front end (ingest, A1, A2)
        z, labels, counts = ops.classmap_label(stack)
        res["denoised"] = z
        if denoised_ready is not None:
            denoised_ready.record(torch.cuda.current_stream())
        # ---- region table (A3): the integer columns only (4 bytes per pixel); the isotope sums of the class components
        # come from the fused plane pass at the end of the batch (_sums_stage), into the zeroed table made here
        stats, cls_out, cc_sums, overflow = ops.region_reduce(labels, counts, cls=z, cap=cap, zero_sums=C)
        res.update(labels=labels, counts=counts, stats=stats, cls_out=cls_out, cc_sums=cc_sums, overflow=overflow)
        # ---- classification, cluster cell counts, region lists (A3 tail, A4)
        res.update(ops.classify_regions(stats, cls_out, counts, tb))
        res["groups"] = {}

    def _merge_all(self, stack, res):
        """every proximity merge of the batch (one mask per cell type + the union of all types, A5 / A6) in five launches:
        the class map is read once for all masks, dilation / run components / cross-tile links / grouping each run once over
        masks x frames"""
        slots = self._merge_slots()
        if not slots:
            return
        B, C, H, W = stack.shape
        tb = self.tables_
        if W % 4 or res["denoised"].data_ptr() % 4:  # the batched bit pass reads the class map four bytes at a time
            for s in slots:
                self._merge_stage(stack, res, s)
            return
        masks = []
        for s in slots:
            bits = 0
            for v in ([tb.slot_value[s]] if s < 4 else tb.slot_value):
                bits |= 1 << v
            masks.append(bits)
        keep = [k for k, m in enumerate(masks) if m]
        if not keep:
            return
        # the batched kernels take up to four masks a call (MaskSet / MergeSlots are four wide): a four-type table has five
        for lo in range(0, len(keep), 4):
            part = keep[lo:lo + 4]
            dbits, run_par = ops.dilated_runs_multi(res["denoised"], [masks[k] for k in part], ta.CELL_CLUSTER_DISTANCE_THRESHOLD // 2)
            gof, ng, gst = ops.merge_groups_fused_multi(dbits, run_par, res["stats"], res["region_list"], res["n_list"], [slots[k] for k in part])
            for m, k in enumerate(part):
                res["groups"][slots[k]] = {"group_of": gof[m], "n_groups": ng[m], "group_stats": gst[m]}

    def _merge_stage(self, stack, res, s):
        """proximity merge of one cell type (s < 4) or of all types together (s = 4) (A5, A6)"""
        B, C, H, W = stack.shape
        tb = self.tables_
        if s < 4:
            bits = 1 << tb.slot_value[s]
        else:
            bits = 0
            for v in tb.slot_value:
                bits |= 1 << v
        if bits == 0:
            return
        # dilated components as a union-find over the vertical runs of the 1-bit image: grouping only needs "same
        # component" at the centroid pixels, so no label image is ever written
        dbits, run_par = ops.dilated_runs(res["denoised"], bits, ta.CELL_CLUSTER_DISTANCE_THRESHOLD // 2)
        gof, ng, gst = ops.merge_groups_fused(dbits, run_par, res["stats"], res["region_list"], res["n_list"], s)
        res["groups"][s] = {"group_of": gof, "n_groups": ng, "group_stats": gst}

    def _fill_stage(self, stack, res):
        """particle-area reconstruction, one fill per cell class on the previous output (A8)"""
        B = stack.shape[0]
        tb = self.tables_
        ds = res["denoised"]
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
        # ---- area / centroid sums of the refined ROIs (plane-free pass); their isotope sums: _sums_stage.  (The fused pass can
        # take the integer columns along -- region_sums2(stats_b=...) -- but the refined ROIs are small, every lane ends a
        # run or two per 32-row block, and eight 64-bit LDS atomics per run end cost more there (614 us against 407 + 191).)
        ws_stats, _, ws_sums, ws_overflow = ops.region_reduce(ws_labels, n_markers, cap=cap, zero_sums=C)
        res.update(ws_stats=ws_stats, ws_sums=ws_sums, ws_overflow=ws_overflow)

    def _sums_stage(self, stack, res):
        """per-ROI isotope sums (M1, .m:122-135) of the class-map components AND of the refined ROIs in one pass over
        the planes.  Sums of class components are only ever reported for cell / cluster regions (tiff_analysis.py:1041-1044):
        that image is summed under the cell classes only."""
        cell_bits = 0
        for v in self.tables_.cell_values:
            cell_bits |= 1 << int(v)
        B, C, H, W = stack.shape
        # the fused pass reads labels and planes as 16-byte quads: a contiguous view at an odd storage offset (legal for run())
        # takes the per-image kernels below like a ragged width does
        aligned = all(t.data_ptr() % 16 == 0 for t in (res["labels"], res["ws_labels"], stack)) and res["denoised"].data_ptr() % 4 == 0
        if W % 4 == 0 and aligned:
            ops.region_sums2(res["labels"], res["denoised"], cell_bits, res["cc_sums"], res["ws_labels"], res["ws_sums"], stack)
            return
        # ragged widths: the per-image kernels (their plane pass adds nothing to the already counted integer columns'
        # cost worth a special path)
        cap = res["stats"].shape[1]
        _, _, res["cc_sums"], _ = ops.region_reduce(res["labels"], res["counts"], cls=res["denoised"], planes=stack, cap=cap,
                                                    sum_classes=cell_bits)
        _, _, res["ws_sums"], _ = ops.region_reduce(res["ws_labels"], res["n_markers"], planes=stack, cap=cap)

    # ------------------------------------------------------------------ table output
    def table_columns(self, C, ratios=RATIOS_5):
        """Column names of every table of :meth:`tables` (known without any data: ranks that own no frame of a
        dataset still agree on the schema, see ``distributed.run_sharded``)."""
        tb = self.tables_
        rn = [r[0] for r in ratios]
        return {
            "cells": ["frame", "label", "class", "kind", "area", "centroid_row", "centroid_col", "min_row", "min_col",
                      "max_row1", "max_col1", "cells", "group", "group_combined"] + ["S%d" % k for k in range(C)] + rn,
            "rois": ["frame", "label", "area", "centroid_row", "centroid_col"] + ["S%d" % k for k in range(C)] + rn,
            "frames": ["frame", "n_labels", "n_rois", "particle_area", "particle_area_recreated", "tie_flag"]
                      + [c % n for n in tb.slot_names for c in ("%s_present", "%s_count", "%s_density", "%s_area_ratio")],
            "distances": ["frame", "label", "nearest_other_type_um"],
            "groups": ["frame", "slot", "group", "area", "centroid_row", "centroid_col", "min_row", "min_col",
                       "max_row1", "max_col1", "members"],
        }

    def tables_device(self, res, frame_ids=None, ratios=RATIOS_5, check=True, distances=False, raster=19.0):
        """The batch as dense row tables, assembled ON THE DEVICE (``csrc/tables.hip``): float64 CUDA tensors ``rois``,
        ``cells``, ``groups`` and ``frames_rec`` (one row per frame: frame id + the int64 record of
        ``pcseg_table_write``, see include/pcseg.h).  One small device-to-host copy (three row totals) sizes the
        outputs; nothing else leaves the GPU, so the tables can go straight into the all-gather.  ``distances``: also the
        nearest-other-type table of BASELINE config 5 (.m:260-268) as ``distances`` = ``[frame, label, distance]`` -- one
        batched launch over the cell rows of every frame (rows of type slot 0, then of slot 1, inside each frame); always
        present (empty unless requested) so that every rank gathers the same set of tables."""
        res.synchronize()
        B, C, H, W = res["shape"]
        dev = res["stats"].device
        # a stream of its own, at high priority: the handful of small kernels here must not queue behind the next
        # batch's big ones on a saturated GPU
        if self._table_stream is None or self._table_stream.device != dev:
            self._table_stream = torch.cuda.Stream(device=dev, priority=-1)
        ts = self._table_stream
        ts.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(ts):
            ids = list(range(B)) if frame_ids is None else [int(v) for v in frame_ids]
            step = ids[1] - ids[0] if B > 1 else 1
            if all(ids[k] == ids[0] + k * step for k in range(B)):  # arithmetic progression (shards are): no upload
                fid = torch.arange(B, dtype=torch.int64, device=dev) * step + ids[0]
            else:
                fid = torch.tensor(ids, dtype=torch.int64).to(dev)
            groups = (res.get("groups") or {}) if self.merged else {}
            dt = ops.build_tables(res, groups, fid, C, ratios, check=check,  # raises what BatchResult.check() raises
                                  distance_slots=self.tables_.slot if distances else None, raster=raster)
            dt["frames_rec"] = torch.cat([fid[:, None].to(torch.float64), dt.pop("frames").to(torch.float64)], dim=1)
            del dt["frame_ids"]
            dt["distances"] = self._distance_rows(dt["cells"], dt.pop("cell_dist", None))
            if res._slot is not None:  # graph mode: the lane may overwrite this result once the tables are out
                res._check_alive()
                res._slot.release = torch.cuda.Event()
                res._slot.release.record(ts)
        torch.cuda.current_stream(dev).wait_stream(ts)
        for t in dt.values():
            t.record_stream(torch.cuda.current_stream(dev))
        return dt

    def _distance_rows(self, cells, cell_dist):
        """``[frame, label, distance]`` for the rows of ``cells`` that have a distance, frame by frame the rows of type
        slot 0, then those of slot 1 (the order in which the script concatenates its two ROI classes, .m:264-268)."""
        if cell_dist is None or cells.shape[0] == 0:
            return torch.zeros((0, 3), dtype=torch.float64, device=cells.device)
        keep = torch.nonzero(~torch.isnan(cell_dist))[:, 0]
        if keep.numel() == 0:
            return torch.zeros((0, 3), dtype=torch.float64, device=cells.device)
        rows = cells[keep]
        lut = torch.from_numpy(self.tables_.slot.astype("int64")).to(cells.device)
        slot = lut[rows[:, 2].to(torch.int64)]
        # rows are in (frame position, label) order already: a STABLE sort by slot inside each frame keeps labels ascending
        pos = torch.cumsum(torch.cat([rows.new_zeros(1), (rows[1:, 0] != rows[:-1, 0]).to(rows.dtype)]), 0).to(torch.int64)
        order = torch.sort(pos * 2 + slot, stable=True)[1]
        return torch.stack([rows[order, 0], rows[order, 1], cell_dist[keep][order]], dim=1)

    def empty_device_tables(self, C, ratios=RATIOS_5, device=None):
        """What :meth:`tables_device` returns for zero frames (a rank that owns no frame of a dataset)."""
        cols = self.table_columns(C, ratios)
        mk = lambda n: torch.zeros((0, n), dtype=torch.float64, device=device)
        return {"rois": mk(len(cols["rois"])), "cells": mk(len(cols["cells"])), "groups": mk(len(cols["groups"])),
                "frames_rec": mk(18), "distances": mk(3)}

    def host_tables(self, dt, C, ratios=RATIOS_5, distances=False, raster=19.0):
        """numpy tables from (downloaded or gathered) :meth:`tables_device` output: ``cells`` / ``rois`` / ``groups`` /
        ``distances`` as they are, ``frames`` after the two ``round(x, 5)`` of get_cell_counts_and_densities
        (tiff_analysis.py:1018-1038; Python's decimal rounding, a handful of numbers per frame).  Every table's width
        must be the one :meth:`table_columns` names for ``C`` planes and these ``ratios``."""
        host = _download(dt)
        cols = self.table_columns(C, ratios)
        tb = self.tables_
        out = {k: host[k] for k in ("cells", "rois", "groups")}
        for k in out:
            if out[k].shape[1] != len(cols[k]):
                raise ValueError("table %r has %d columns, the schema for %d planes names %d" % (k, out[k].shape[1], C, len(cols[k])))
        rec = host["frames_rec"]
        px2 = ta.PX_TO_UM_CONV ** 2
        frame_rows = []
        n_slots = len(tb.slot_names)
        nan = float("nan")
        for row_rec in rec.tolist():  # plain Python floats: indexing numpy scalars one by one cost 50 us per frame
            row = row_rec[:6]
            pa_um = row_rec[3] / px2
            for s in range(n_slots):
                present, count, area_px = int(row_rec[6 + 3 * s]), int(row_rec[7 + 3 * s]), int(row_rec[8 + 3 * s])
                ok = present and pa_um
                row += [float(present), float(count), round(count / pa_um, 5) if ok else nan,
                        round((area_px / px2) / pa_um, 5) if ok else nan]
            frame_rows.append(row)
        out["frames"] = np.array(frame_rows, np.float64).reshape(len(frame_rows), len(cols["frames"]))
        # the nearest-other-type table comes from the device (tables_device(distances=True)); without it an empty table
        if distances and "distances" not in host:
            raise ValueError("host_tables(distances=True) needs tables made by tables_device(..., distances=True)")
        out["distances"] = host["distances"].reshape(-1, 3) if distances and "distances" in host else np.zeros((0, 3), np.float64)
        for k, v in cols.items():
            out[k + "_columns"] = v
        return out

    def tables(self, res, frame_ids=None, ratios=RATIOS_5, distances=False, raster=19.0, check=True):
        """Download one batch as numpy tables: ``cells`` (one row per cell / cluster region), ``rois`` (one row per
        refined ROI), ``frames`` (one row per frame) and ``groups`` (one row per merged group): :meth:`tables_device`
        followed by :meth:`host_tables`.  ``check=False`` skips ``BatchResult.check`` (a caller that has looked at the
        flags itself, e.g. to keep the ROI rows of a batch in which the reference would have raised on one frame's
        cluster statistics)."""
        C = res["shape"][1]
        return self.host_tables(self.tables_device(res, frame_ids, ratios, check, distances, raster), C, ratios, distances, raster)
