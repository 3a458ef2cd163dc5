"""Frame-level data parallelism: one process per GPU, frames are independent units.

The reference processes one image at a time in one process (tiff_analysis.py:1130-1132); frames never exchange
state, so the path shards with NO data-path collective.  The only exchange is the all-gather of the final per-ROI
table: row counts first, then one padded ``all_gather`` -- ``nccl`` (= RCCL over xGMI) for
CUDA tensors, ``gloo`` for CPU tensors (tests).  Label masks stay on the rank that made them.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world_size):
    """Round-robin frame ownership: frame i -> rank i mod world (BASELINE config 3)."""
    return list(range(rank, n_frames, world_size))


def all_gather_table(table, group=None, sort_cols=(0, 1)):
    """Gather (rows_r, cols) float64 tables from every rank into one table, sorted by the key columns
    (frame, label) so that the result does not depend on the number of ranks."""
    if not dist.is_available() or not dist.is_initialized():
        out = table
    else:
        world = dist.get_world_size(group)
        dev = table.device
        n = torch.tensor([table.shape[0]], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n, group=group)
        counts = [int(c.item()) for c in counts]
        cols = table.shape[1]
        pad = torch.zeros((max(max(counts), 1), cols), dtype=table.dtype, device=dev)
        pad[: table.shape[0]] = table
        parts = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    if out.shape[0] and sort_cols:
        key = out[:, sort_cols[0]].to(torch.float64)
        for c in sort_cols[1:]:
            key = key * (float(out[:, c].max().item()) + 1.0) + out[:, c].to(torch.float64)
        out = out[torch.argsort(key)]
    return out


TABLE_KEYS = ("cells", "rois", "frames", "groups", "distances")


def gather_tables(tables, device=None, group=None):
    """all-gather every table of FramePipeline.tables() (numpy in, numpy out).  Every rank must pass the same set of
    2-D tables with the same column counts -- also a rank without a single row (``(0, ncols)`` arrays)."""
    out = dict(tables)
    for name in [k for k in TABLE_KEYS if isinstance(tables.get(k), np.ndarray)]:
        t = torch.from_numpy(np.ascontiguousarray(tables[name], dtype=np.float64))
        if device is not None:
            t = t.to(device)
        sort_cols = (0,) if name == "frames" else ((0, 1, 2) if name == "groups" else (0, 1))
        out[name] = all_gather_table(t, group, sort_cols).cpu().numpy()
    return out


def run_sharded(n_frames, make_batch, pipe, batch=64, group=None, device=None, planes=5, **table_kwargs):
    """BASELINE configs 3 / 5: frames ``rank, rank + world, ...`` of a dataset go through ``pipe`` in batches of
    ``batch`` and the per-ROI tables of all ranks are gathered once at the end.  ``make_batch(frame_ids)`` returns the
    ``(len(frame_ids), planes, H, W)`` float32 CUDA stack of those frames (e.g. ``synth.gen_batch_torch`` per seed, or
    frames written by ``split_zstack.process_tif``).

    Batch k's tables are taken only after batch k + 1 has been handed to the pipeline, so the table assembly and its
    download run under the next batch's kernels.  A rank that owns no frame (``n_frames < world``) contributes empty
    tables of the pipeline's schema (``pipe.table_columns``), so that every rank enters the same collectives with the
    same column counts."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    mine = shard_frames(n_frames, rank, world)
    parts = []
    pending = None
    for i in range(0, len(mine), batch):
        ids = mine[i:i + batch]
        res = pipe.run(make_batch(ids))
        if pending is not None:
            parts.append(pipe.tables(pending[0], frame_ids=pending[1], **table_kwargs))
        pending = (res, ids)
    if pending is not None:
        parts.append(pipe.tables(pending[0], frame_ids=pending[1], **table_kwargs))
    columns = pipe.table_columns(planes, **({"ratios": table_kwargs["ratios"]} if "ratios" in table_kwargs else {}))
    merged = {}
    for k in TABLE_KEYS:
        rows = [p[k] for p in parts if k in p]
        merged[k] = np.concatenate(rows) if rows else np.zeros((0, len(columns[k])), np.float64)
        if merged[k].shape[1] != len(columns[k]):
            raise ValueError("table %r has %d columns, the schema says %d" % (k, merged[k].shape[1], len(columns[k])))
        merged[k + "_columns"] = columns[k]
    return gather_tables(merged, device=device, group=group)
