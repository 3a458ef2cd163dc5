#!/usr/bin/env python3
"""Round-3 experiment (GPU box): hipGraph replays on different streams overlap less than eager launches from many
streams do (bench: 2 graphs in flight 5.99 ms, 3 in flight 6.17 ms, eager 8 lanes 5.37 ms per step).  Does ONE graph that
holds the chains of several batches (parallelism inside the graph) do better?  ms per 64-frame batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from particle_col_image_segmentation_amd import synth
from particle_col_image_segmentation_amd.pipeline import FramePipeline, _lanes_for


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, H, W = 64, 1024, 1024
    stack = synth.gen_batch_torch(10_000, B, H, W, dev)
    ct = dict(synth.CELL_TYPES_5)
    for width in (1, 2, 3, 4):
        pipe = FramePipeline(ct, lanes=width)
        pipe.run(stack).synchronize()
        pipe.synchronize()
        _, lane_streams = _lanes_for(dev, width)
        main_s = torch.cuda.Stream()

        def chains():
            ready = torch.cuda.Event()
            ready.record(main_s)
            outs = []
            for lane in range(width):
                out, done = pipe._run_streams(lane_streams[lane], stack, ready, {"shape": tuple(stack.shape)})
                outs.append(out)
                for ev in done:
                    main_s.wait_event(ev)
            return outs

        with torch.cuda.stream(main_s):
            chains()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main_s, capture_error_mode="thread_local"):
            outs = chains()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        n = 12
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("one graph holding %d chains: %.3f ms per batch" % (width, 1e3 * dt / n / width), flush=True)
        ok = all(torch.equal(outs[0]["ws_labels"], o["ws_labels"]) and torch.equal(outs[0]["labels"], o["labels"]) for o in outs[1:])
        print("  chains agree:", ok, flush=True)
        del g, outs, pipe
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
