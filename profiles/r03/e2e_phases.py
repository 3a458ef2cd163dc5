"""Where a dataset run's time goes beyond the kernels (GPU box): the phases of distributed.run_sharded, timed one by one
on the bench's own 16-batch dataset.  python profiles/r03/e2e_phases.py"""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from particle_col_image_segmentation_amd import distributed, synth
from particle_col_image_segmentation_amd.pipeline import FramePipeline


def main():
    dev = torch.device("cuda:0")
    B, H, W = 64, 1024, 1024
    stack = synth.gen_batch_torch(10_000, B, H, W, dev)
    cell_types = dict(synth.CELL_TYPES_5)
    pipe = FramePipeline(cell_types)
    n_batches = 16
    ids_all = [list(range(i * B, (i + 1) * B)) for i in range(n_batches)]

    def one(verbose):
        t = [time.perf_counter()]
        parts, pending = [], collections.deque()
        for ids in ids_all:
            pending.append((pipe.run(stack), ids))
            if len(pending) > pipe.lanes:
                res, rid = pending.popleft()
                parts.append(pipe.tables_device(res, frame_ids=rid, check=False))
        t.append(time.perf_counter())          # all batches handed over
        while pending:
            res, rid = pending.popleft()
            parts.append(pipe.tables_device(res, frame_ids=rid, check=False))
        torch.cuda.synchronize()
        t.append(time.perf_counter())          # drained
        merged = {k: torch.cat([p[k] for p in parts]) for k in parts[0]}
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        gathered = distributed.gather_tables(merged, device=dev, group=None, presorted=True)
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        tabs = pipe.host_tables(gathered, 5)
        t.append(time.perf_counter())
        if verbose:
            names = ["hand over 16 batches (tables of the first 8 inside)", "drain: tables of the last 8", "concatenate",
                     "gather + download", "host epilogue"]
            for n, a, b in zip(names, t[:-1], t[1:]):
                print("%-55s %8.2f ms" % (n, (b - a) * 1e3))
            print("%-55s %8.2f ms   rows %d, bytes %.1f MB" % ("total", (t[-1] - t[0]) * 1e3, tabs["rois"].shape[0],
                                                               sum(v.numel() * v.element_size() for v in merged.values()) / 1e6))
            for k, v in merged.items():
                print("   ", k, tuple(v.shape), v.dtype)

    one(False)
    one(True)
    # the product path (one rank: rows streamed to pinned host memory batch by batch)
    for rep in range(3):
        pipe.synchronize()
        t0 = time.perf_counter()
        tabs = distributed.run_sharded(n_batches * B, lambda ids: stack[:len(ids)], pipe, batch=B, device=dev, check=False)
        torch.cuda.synchronize()
        print("run_sharded: %.2f ms, %d roi rows" % ((time.perf_counter() - t0) * 1e3, tabs["rois"].shape[0]))
    # kernel-only for comparison
    pipe.synchronize()
    t0 = time.perf_counter()
    rs = [pipe.run(stack) for _ in range(n_batches)]
    pipe.synchronize()
    print("kernel only, 16 batches: %.2f ms" % ((time.perf_counter() - t0) * 1e3))


if __name__ == "__main__":
    main()
