"""Frame-level data parallelism: one process per GPU, frames are independent units.

The reference processes one image at a time in one process (tiff_analysis.py:1130-1132); frames never exchange
state, so the path shards with NO data-path collective.  The only exchange is the all-gather of the final per-ROI
table: row counts first, then one padded ``all_gather`` -- ``nccl`` (= RCCL over xGMI) for
CUDA tensors, ``gloo`` for CPU tensors (tests).  Label masks stay on the rank that made them.
"""
import collections

import numpy as np
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world_size):
    """Round-robin frame ownership: frame i -> rank i mod world (BASELINE config 3)."""
    return list(range(rank, n_frames, world_size))


def _dist_on(group=None):
    return dist.is_available() and dist.is_initialized()


def _lexsort_rows(table, sort_cols):
    """Rows ordered by the key columns, first key most significant, as a chain of STABLE sorts (exact for any float64
    key values; rows with equal keys keep their order -- the `distances` rows of a frame are slot 0 first, then slot 1,
    .m:264-268, and must stay that way whatever the number of ranks)."""
    order = torch.arange(table.shape[0], device=table.device)
    for c in reversed(tuple(sort_cols)):
        order = order[torch.sort(table[order, c], stable=True)[1]]
    return table[order]


GATHER_CHUNK_BYTES = 64 << 20  # per table and rank: BASELINE config 5's 460 MB per rank never needs world x max_rows at once


def _gather_rows(tables, group=None, chunk_bytes=None):
    """all-gather a list of (rows_r, cols) tables of one dtype/device: ONE exchange of all the row counts, then every
    table in chunks of at most ``chunk_bytes`` per rank through ``all_gather_into_tensor`` on one preallocated buffer,
    each rank's valid rows copied straight to their place (rank-major) in the output."""
    world = dist.get_world_size(group)
    dev = tables[0].device
    chunk_bytes = int(chunk_bytes or GATHER_CHUNK_BYTES)
    mine = torch.tensor([int(t.shape[0]) for t in tables], dtype=torch.int64, device=dev)
    counts = torch.zeros((world, len(tables)), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts.view(-1), mine, group=group)
    counts = counts.cpu().tolist()  # the one host sync of the gather
    out = []
    for k, t in enumerate(tables):
        cols = int(t.shape[1])
        per_rank = [int(c[k]) for c in counts]
        starts = [0]
        for c in per_rank:
            starts.append(starts[-1] + c)
        res = torch.empty((starts[-1], cols), dtype=t.dtype, device=dev)
        most = max(per_rank)
        if most and cols:
            chunk_rows = max(1, min(most, chunk_bytes // (cols * t.element_size())))
            send = torch.zeros((chunk_rows, cols), dtype=t.dtype, device=dev)
            recv = torch.empty((world, chunk_rows, cols), dtype=t.dtype, device=dev)
            for lo in range(0, most, chunk_rows):
                n = max(0, min(int(t.shape[0]) - lo, chunk_rows))
                if n:
                    send[:n] = t[lo:lo + n]
                dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
                for r in range(world):
                    m = max(0, min(per_rank[r] - lo, chunk_rows))
                    if m:
                        res[starts[r] + lo:starts[r] + lo + m] = recv[r, :m]
        out.append(res)
    return out


def all_gather_table(table, group=None, sort_cols=(0, 1), presorted=False, chunk_bytes=None):
    """Gather (rows_r, cols) float64 tables from every rank into one table, sorted by the key columns
    (frame, label) so that the result does not depend on the number of ranks."""
    single = not _dist_on() or dist.get_world_size(group) == 1
    out = table if not _dist_on() else _gather_rows([table], group, chunk_bytes)[0]
    if presorted and single:
        return out  # one rank, rows already in (frame, label) order: nothing to merge
    if out.shape[0] and sort_cols:
        out = _lexsort_rows(out, sort_cols)
    return out


TABLE_KEYS = ("cells", "rois", "frames", "groups", "distances")
# key columns per table.  `distances`: frame only -- a frame lives on one rank and the sort is stable, so the rows of a
# frame keep the order their rank made them in (type slot 0 first, then slot 1), with any number of ranks
_SORT_COLS = {"frames": (0,), "frames_rec": (0,), "groups": (0, 1, 2), "distances": (0,)}


def gather_tables(tables, device=None, group=None, presorted=False, chunk_bytes=None):
    """all-gather every 2-D table of a dict (numpy arrays or tensors in, same kind out), each sorted by its key columns.
    Every rank must pass the same set of tables with the same column counts -- also a rank without a single row
    (``(0, ncols)`` arrays).  One exchange of row counts for all tables, then chunked gathers (``_gather_rows``)."""
    out = dict(tables)
    names, tens, was_np = [], [], []
    for name, value in tables.items():
        is_np = isinstance(value, np.ndarray)
        if not (is_np or isinstance(value, torch.Tensor)) or value.ndim != 2:
            continue
        t = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float64)) if is_np else value
        if device is not None:
            t = t.to(device)
        names.append(name)
        tens.append(t.contiguous())
        was_np.append(is_np)
    if not names:
        return out
    single = not _dist_on() or dist.get_world_size(group) == 1
    if _dist_on():
        same = all(t.dtype == tens[0].dtype and t.device == tens[0].device for t in tens)
        gathered = _gather_rows(tens, group, chunk_bytes) if same else [_gather_rows([t], group, chunk_bytes)[0] for t in tens]
    else:
        gathered = tens
    for name, g, is_np in zip(names, gathered, was_np):
        if not (presorted and single) and g.shape[0]:
            g = _lexsort_rows(g, _SORT_COLS.get(name, (0, 1)))
        out[name] = g.cpu().numpy() if is_np else g
    return out


def _agree_planes(local_planes, group, device):
    """Plane count of the dataset as every rank must see it: the maximum over the ranks (a rank that owns no frame
    knows none and contributes 0).  One tiny all-reduce; without it a rank with an empty shard would build its empty
    tables from a default width and enter the padded all-gather with a different column count than its peers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(local_planes)
    if device is None and dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([int(local_planes)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


class _HostRows:
    """A rank's table rows in pinned host memory, filled batch by batch on a copy stream of its own while newer batches
    compute (one rank: there is nothing to gather, and downloading the whole dataset's tables after the last batch -- 0.5 GB
    for 1 024 frames -- cost as much as two batches' kernels).  The buffers are sized from the first batch and the number
    of batches to come; a table that outgrows its buffer moves to one twice the size."""

    def __init__(self, device, expected_batches):
        self.device = device
        self.expected = max(1, int(expected_batches))
        self.stream = torch.cuda.Stream(device=device)
        self.buf, self.rows = {}, {}

    def append(self, part):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        for k, t in part.items():
            n, cols = int(t.shape[0]), int(t.shape[1])
            if k not in self.buf:
                self.buf[k] = torch.empty((int(n * self.expected * 1.15) + 4096, cols), dtype=t.dtype, pin_memory=True)
                self.rows[k] = 0
            used = self.rows[k]
            if used + n > self.buf[k].shape[0]:
                self.stream.synchronize()  # the copies into the old buffer
                grown = torch.empty((2 * (used + n), cols), dtype=t.dtype, pin_memory=True)
                grown[:used] = self.buf[k][:used]
                self.buf[k] = grown
            if n:
                with torch.cuda.stream(self.stream):
                    self.buf[k][used:used + n].copy_(t, non_blocking=True)
                t.record_stream(self.stream)
            self.rows[k] = used + n

    def finish(self):
        self.stream.synchronize()
        return {k: self.buf[k][:self.rows[k]].numpy() for k in self.buf}


def run_sharded(n_frames, make_batch, pipe, batch=64, group=None, device=None, planes=None, check=True, force_gather=False,
                chunk_bytes=None, **table_kwargs):
    """BASELINE configs 3 / 5: frames ``rank, rank + world, ...`` of a dataset go through ``pipe`` in batches of
    ``batch`` and the tables of all ranks are gathered once at the end.  ``make_batch(frame_ids)`` returns the
    ``(len(frame_ids), planes, H, W)`` float32 CUDA stack of those frames (e.g. ``synth.gen_batch_torch`` per seed, or
    frames written by ``split_zstack.process_tif`` through ``ingest.FrameUploader``).

    Per batch only the device-side table assembly runs (``pipe.tables_device``: three kernels and one 24-byte read),
    and only after ``pipe.lanes`` NEWER batches have been handed to the pipeline, so the wait for the batch is short and
    the assembly sits under the newer batches' kernels.  The per-batch
    device tables are concatenated, all-gathered as device tensors (RCCL when the group is NCCL) and downloaded ONCE;
    the host epilogue (``pipe.host_tables``) then runs on the gathered rows.  A rank that owns no frame
    (``n_frames < world``) contributes ``pipe.empty_device_tables``, so that every rank enters the same collectives with
    the same column counts.  The plane count comes from the data (the first batch a rank makes; ranks agree on it with one
    all-reduce), ``planes`` only overrides it.  ``table_kwargs`` (ratios, distances, raster) go to ``host_tables``.

    ``force_gather``: one rank normally streams its rows to pinned host memory batch by batch (``_HostRows``: there is
    nothing to gather); with this switch it takes the route every rank of a larger world takes -- device tables
    concatenated, ``gather_tables`` through the process group (RCCL for CUDA tensors), one download -- so that the N > 1
    path can be exercised on one GPU.  ``chunk_bytes``: per-rank chunk of the gather (default 64 MB).

    Graph-mode pipelines (``FramePipeline(graph=True)``): a lane's replay reads the input buffer it was captured on, so
    ``make_batch`` may only refill a buffer whose previous batch has finished.  The oldest pending batch's tables are
    therefore taken (which waits for that batch) BEFORE ``make_batch`` is called for the batch that reuses its lane:
    with ``lanes`` in flight a caller needs ``lanes`` rotating input buffers, not ``lanes + 1``."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    mine = shard_frames(n_frames, rank, world)
    ratio_kw = {k: table_kwargs[k] for k in ("ratios", "distances", "raster") if k in table_kwargs}
    parts = []
    n_batches = (len(mine) + batch - 1) // batch
    host_rows = None  # one rank: the rows go to the host as the batches finish (see _HostRows)

    def take(part):
        nonlocal host_rows
        if world == 1 and not force_gather and all(isinstance(v, torch.Tensor) and v.is_cuda and v.ndim == 2 for v in part.values()):
            if host_rows is None:
                host_rows = _HostRows(next(iter(part.values())).device, n_batches)
            host_rows.append(part)
        else:
            parts.append(part)

    pending = collections.deque()
    depth = max(1, int(getattr(pipe, "lanes", 1)))  # batches the pipeline keeps in flight: a table is assembled (which
    seen_planes = 0
    for i in range(0, len(mine), batch):            # waits for its batch) only once that many newer batches are queued
        ids = mine[i:i + batch]
        # a graph-mode pipeline keeps `lanes` results alive (a lane's next replay overwrites its previous result, and reads
        # the input buffer it was captured on): the oldest batch's tables are taken BEFORE the feeder refills that buffer
        # and before the batch that reuses its lane is handed over -- the wait for it runs under the other lanes' kernels
        while getattr(pipe, "graph", False) and len(pending) >= depth:
            res, rid = pending.popleft()
            take(pipe.tables_device(res, frame_ids=rid, check=check, **ratio_kw))
        frames = make_batch(ids)
        if not seen_planes and hasattr(frames, "shape") and len(frames.shape) == 4:
            seen_planes = int(frames.shape[1])
        pending.append((pipe.run(frames), ids))
        # ... and the tables of every batch that has finished meanwhile are taken right away (in order): what is left
        # to drain after the last batch is then only what is still running
        while pending and (len(pending) > depth or (len(pending) > 1 and getattr(pending[0][0], "ready", lambda: False)())):
            res, rid = pending.popleft()
            take(pipe.tables_device(res, frame_ids=rid, check=check, **ratio_kw))
    while pending:
        res, rid = pending.popleft()
        take(pipe.tables_device(res, frame_ids=rid, check=check, **ratio_kw))
    if planes is None:
        planes = _agree_planes(seen_planes, group, device) or 5
    if host_rows is not None:
        # (one rank's rows come out of the batches in frame order, labels ascending: that IS the gathered order)
        return pipe.host_tables(host_rows.finish(), planes, **table_kwargs)
    if parts:
        merged = {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
    else:
        merged = pipe.empty_device_tables(planes, device=device, **({"ratios": ratio_kw["ratios"]} if "ratios" in ratio_kw else {}))
    # (a rank's own rows come out of the batches in frame order, labels ascending: with one rank that IS the gathered
    # order and the sort is skipped)
    gathered = gather_tables(merged, device=device, group=group, presorted=world == 1 and mine == sorted(mine),
                             chunk_bytes=chunk_bytes)
    return pipe.host_tables(gathered, planes, **table_kwargs)
