#!/usr/bin/env python3
"""Round-3 experiment (GPU box): does hipGraph capture of the chain work through torch.cuda.graph, and what do
sub-batches (Infinity-Cache-sized) buy with and without it?  Prints one line per configuration: ms per 64-frame step."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from particle_col_image_segmentation_amd import synth
from particle_col_image_segmentation_amd.pipeline import FramePipeline, BatchResult


def timeit(fn, steps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, H, W = 64, 1024, 1024
    stack = synth.gen_batch_torch(10_000, B, H, W, dev)
    ct = dict(synth.CELL_TYPES_5)
    print("built", flush=True)

    # --- eager, one stream, sub-batches back to back (pure cache effect + launch overhead)
    solo = FramePipeline(ct, overlap=False)
    for sb in (64, 32, 16, 8):
        def step():
            for i in range(0, B, sb):
                solo.run(stack[i:i + sb])
        print("eager serial  sub=%2d : %.3f ms/step" % (sb, timeit(step, 6, 2)), flush=True)

    # --- eager, lanes=8 as shipped
    pipe = FramePipeline(ct, lanes=8)
    def step8():
        pipe.run(stack)
    for _ in range(10):
        step8()
    pipe.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step8()
    pipe.synchronize()
    print("eager lanes=8 sub=64 : %.3f ms/step" % (1e3 * (time.perf_counter() - t0) / 20), flush=True)

    # --- graphs, one stream per graph
    def capture_serial(view):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            solo.run(view)  # warm on this stream (allocator, hipMalloc of counters)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            res = solo.run(view)
        return g, res

    for sb in (64, 32, 16, 8):
        try:
            graphs = [capture_serial(stack[i:i + sb]) for i in range(0, B, sb)]
            print("captured %d serial graphs of %d frames" % (len(graphs), sb), flush=True)
            # replay back to back on ONE stream
            def step():
                for g, _ in graphs:
                    g.replay()
            print("graph 1 stream sub=%2d : %.3f ms/step" % (sb, timeit(step, 10, 3)), flush=True)
            # replay on k streams round-robin (k sub-batches in flight)
            for k in (2, 4):
                if len(graphs) < k and sb != 64:
                    continue
                streams = [torch.cuda.Stream() for _ in range(k)]
                def stepk():
                    for j, (g, _) in enumerate(graphs):
                        with torch.cuda.stream(streams[j % k]):
                            g.replay()
                print("graph %d streams sub=%2d : %.3f ms/step" % (k, sb, timeit(stepk, 10, 3)), flush=True)
            if sb == 64:
                # two whole-batch graphs alternating on two streams
                g2 = capture_serial(stack)
                streams = [torch.cuda.Stream() for _ in range(2)]
                gs = [graphs[0][0], g2[0]]
                cnt = [0]
                def step2():
                    j = cnt[0] % 2
                    cnt[0] += 1
                    with torch.cuda.stream(streams[j]):
                        gs[j].replay()
                print("graph 2 alternating whole-batch graphs : %.3f ms/step" % timeit(step2, 20, 4), flush=True)
                del g2, gs
            # correctness of a replay against eager
            ref = solo.run(stack[:sb])
            torch.cuda.synchronize()
            graphs[0][0].replay()
            torch.cuda.synchronize()
            ok = torch.equal(ref["ws_labels"], graphs[0][1]["ws_labels"]) and torch.equal(ref["labels"], graphs[0][1]["labels"])
            print("replay == eager:", ok, flush=True)
            del graphs
            torch.cuda.empty_cache()
        except Exception:
            traceback.print_exc()
            print("graph serial sub=%d FAILED" % sb, flush=True)

    # --- graphs, the five-stream lane captured into one graph
    try:
        from particle_col_image_segmentation_amd import pipeline as pl
        mp = FramePipeline(ct, lanes=1)
        mp.run(stack).synchronize()  # creates lane streams, warms
        mp.synchronize()
        def capture_lane(view):
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                ready = torch.cuda.Event()
                ready.record(s)
                entries, done = mp._run_lane(0, view, ready, {"shape": tuple(view.shape)})
                for ev in done:
                    s.wait_event(ev)
            return g, entries
        for sb in (64, 16):
            graphs = [capture_lane(stack[i:i + sb]) for i in range(0, B, sb)]
            print("captured %d five-stream graphs of %d frames" % (len(graphs), sb), flush=True)
            def step():
                for g, _ in graphs:
                    g.replay()
            print("graph5 1 stream sub=%2d : %.3f ms/step" % (sb, timeit(step, 10, 3)), flush=True)
            for k in (2, 4):
                # ONE GRAPH PER (stream, view).  The round-3 version of this leg replayed the SAME captured graph on k streams
                # in turn (with sb = 64 there is one graph): k instances of one chain then ran at once on one private
                # workspace -- union-find parents, overflow tables, dirty-tile lists -- and the run ended in "Memory access
                # fault by GPU" (exp_graph_r3a.log, right after "graph5 2 streams sub=64": a parent entry read while another
                # instance rewrote it walked out of its frame).  A captured chain owns its workspace: it may be replayed
                # again only after its previous replay, i.e. on ONE stream; concurrency needs a graph per stream.
                streams = [torch.cuda.Stream() for _ in range(k)]
                per_stream = [graphs] + [[capture_lane(stack[i:i + sb]) for i in range(0, B, sb)] for _ in range(k - 1)]
                cnt = [0]
                def stepk():
                    for idx in range(len(graphs)):
                        j = cnt[0] % k
                        with torch.cuda.stream(streams[j]):
                            per_stream[j][idx][0].replay()
                        cnt[0] += 1
                print("graph5 %d streams sub=%2d : %.3f ms/step" % (k, sb, timeit(stepk, 12, 4)), flush=True)
                del per_stream
            ref = solo.run(stack[:sb])
            torch.cuda.synchronize()
            graphs[0][0].replay()
            torch.cuda.synchronize()
            print("replay5 == eager:", torch.equal(ref["ws_labels"], graphs[0][1]["ws_labels"]) and
                  torch.equal(ref["labels"], graphs[0][1]["labels"]), flush=True)
            del graphs
            torch.cuda.empty_cache()
    except Exception:
        traceback.print_exc()
        print("graph5 FAILED", flush=True)


if __name__ == "__main__":
    main()
