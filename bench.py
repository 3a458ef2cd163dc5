#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s segmented on synthetic 5-channel 1024x1024 frames (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            # N = 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the full per-frame chain (FramePipeline.run: class map -> median -> label -> region table
+ isotope sums -> classification -> proximity merge -> particle fill -> refine (EDT, maxima, watershed) -> ROI sums)
over one batch of 64 frames of 1024x1024x5 float32 that is already resident in HBM (BASELINE config 2).  Frames are
independent, so N ranks each process their own batch (weak scaling, no data-path collective); the only exchange is
the RCCL all-gather of the ROI table, done once after the timed region.

The timed region carries no instrumentation (no event records, no counter reads).  How its steps are LAUNCHED is settled in the
setup phase (`--launch auto`, the default): eight untimed steps, twice, as eager launches from 8 host threads with five streams
each and as hipGraph replays with four in flight from one host thread; the timed region runs the faster (`config.launch`,
`config.launch_probe_ms_per_step`).  On a quiet host the two are within 3 %; on a busy one the eight launch threads fall behind
and replays win by 15 %.  `--launch eager|graph` fixes the mode.  The JSON line carries `roofline` (dominant kernel by
serial time; HIP events on the launch stream in two extra single-stream steps right after the timed region -- the
reproducible figure -- with the same kernel's duration while 8 batches share the chip as `in_flight`, and `stage_frac` =
the watershed stage's compulsory 13 B/px over the serial time of all its kernels) and `cpu_baseline` (the CPU oracle timed
on this box's host cores on a bounded sample; reported, not the target).  `end_to_end` (dataset run incl. table assembly,
download and gather), `mosaic4096` / `frames2048` (BASELINE configs 4 / 5 shapes), `batch64` / `secondary` (quantised
input) are further legs of the same run, never `value`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
CHAIN_BYTES_PER_PIXEL = 28  # 5 x f32 in + int32 class-CC mask + int32 ROI mask

# algorithmic (compulsory) bytes per pixel of one launch of each kernel, stated in DESIGN.md
# (kernels of the default chain that can come out on top of the per-kernel table)
KERNEL_BYTES_PER_PIXEL = {
    "classmap_median_ccl_kernel": 26.0, "ccl_flatten_count_kernel": 4.0, "ccl_relabel_kernel": 8.0,
    "ccl_relabel_quads_kernel": 8.0, "region_sums2_col_kernel": 29.0, "region_stats_col_kernel": 4.0,
    "edt_bits_kernel": 5.125, "edt_row_kernel": 4.0, "edt_reach_kernel": 2.0, "locmax_candidates_kernel": 8.25,
    "ws_relax_kernel": 12.0, "ws_uf_tile_kernel": 13.0, "ws_uf_label4_kernel": 9.0, "ws_exact_kernel": 21.0,
}
# the watershed STAGE's compulsory traffic: boundary plane 4 + markers 4 + mask 1 in, labels 4 out (once per pixel,
# however often the relaxation revisits a tile): what `roofline.stage_frac` prices the stage's serial kernel time against
WS_STAGE_BYTES_PER_PIXEL = 13.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU-baseline sample (0 = 2 per core)")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel event table to stderr")
    ap.add_argument("--lanes", type=int, default=0,
                    help="batches in flight (0 = the pipeline's default: 8 host threads, or 2 graph replays with --graph)")
    ap.add_argument("--graph", action="store_true",
                    help="time hipGraph replays (FramePipeline(graph=True)) instead of eager launches")
    ap.add_argument("--graph-leg-steps", type=int, default=10,
                    help="steps of the `graph_replay` leg of a default run (0 = off): the same chain as one hipGraph replay per step")
    ap.add_argument("--no-merge", action="store_true",
                    help="A/B aid, NOT the headline workload: skip the proximity merges (what do their kernels cost the batch?)")
    ap.add_argument("--eager", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--batch64-frames", type=int, default=64,
                    help="frames of the `batch64` leg (0 = off): the quantised boundary plane at BASELINE config 2's own batch size")
    ap.add_argument("--single-class-stream", action="store_true",
                    help="A/B aid: merges and particle fill on the class-map stream instead of streams of their own")
    ap.add_argument("--serial", action="store_true",
                    help="profiling aid, not the headline: both kernel chains on one stream, so that per-kernel durations "
                         "are not inflated by the other chain's kernels sharing the CUs")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the dataset / PCIe legs")
    ap.add_argument("--secondary-batch", type=int, default=2560,
                    help="frames of the SECONDARY leg (0 = off): one batch with the boundary plane quantised to k/100, the "
                         "realistic ilastik-random-forest input on which every frame floods through ties; reported as "
                         "\"secondary\" next to the headline, never as `value`")
    ap.add_argument("--no-shape-legs", action="store_true",
                    help="skip the `mosaic4096` (BASELINE config 4) and `frames2048` (config 5's frame shape) legs")
    ap.add_argument("--levels", type=int, default=0,
                    help="NOT the headline workload: quantise the boundary plane to k/LEVELS (random-forest-like vote "
                         "fractions); every frame then floods through ties and takes the watershed's exact path")
    ap.add_argument("--launch", choices=["auto", "eager", "graph"], default="auto",
                    help="how the timed steps are launched.  auto (default): the SETUP phase, before warm-up, runs a few untimed "
                         "steps both ways -- eager launches from eight host threads, and hipGraph replays with four in flight from "
                         "one host thread -- and the timed region uses the faster; the same chain either way.  (Eager launches "
                         "overlap a little better on a quiet host; on a busy one the eight launch threads fall behind and "
                         "replays win by 15 %%: profiles/r04/ab_logs/r5d_*.)")
    ap.add_argument("--watershed-mode", type=int, default=0,
                    help="profiling aid for ABLATION builds (libraries that skip a phase: wrong labels, right timing): 2 = frames "
                         "the proof fails on are reported, not recomputed by the exact flood.  The headline runs mode 0")
    return ap.parse_args()


def cpu_baseline(res, stack, cell_types, n_frames, tables=None):
    """CPU oracle (bit-exact restatement of the reference chain, 'port') on this box's host cores, one frame per
    process, on the FIRST n frames of the batch that was just timed -- and, because the oracle's results for those
    frames exist anyway, the parity check of the benchmark itself: the GPU's masks for exactly these frames must be
    bit-exact, the ROI plane sums within 1e-6 relative.  A mismatch raises (non-zero exit)."""
    from oracle import parity
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    n = min(int(stack.shape[0]), n_frames or 2 * cores)
    H, W = int(stack.shape[2]), int(stack.shape[3])
    refs, wall, procs = parity.run_oracle(stack[:n].cpu().numpy(), cell_types, merged=True, processes=cores)
    checked = parity.compare(res, range(n), refs, sums_rtol=1e-6)  # images, counts, classification, merged groups, ROI sums
    if tables is not None:
        # ... and the device-assembled `groups` table / group columns of `cells` of the same frames
        host = {k: tables[k].cpu().numpy() for k in ("groups", "cells")}
        parity.compare_tables(host, list(range(n)), refs)  # (rank 0 numbers its frames 0 .. B-1)
    per = [r["seconds"] for r in refs]
    block = {"value": round(n * H * W / wall / 1e6, 4), "unit": "Mpixels/s", "cores": procs, "kind": "port",
             "sample": "the first %d frames (%dx%dx5) of the timed batch, full chain incl. O(R) merge, %d processes, "
                       "wall %.1f s (%.2f s/frame/core)" % (n, H, W, procs, wall, sum(per) / len(per))}
    # how the port relates to the real reference (Python + scikit-image, which cannot travel to this box): measured in
    # the build container by oracle/measure_reference.py on this generator -- incl. the reference's merge AS WRITTEN,
    # whose O(R^2) list comprehension (tiff_analysis.py:850-852) dominates its CPU path
    ratio_path = os.path.join(ROOT, "oracle", "reference_ratio.json")
    if os.path.exists(ratio_path):
        ratio = json.load(open(ratio_path))
        block["reference_vs_port"] = {
            "measured": ratio["where"] + "; " + ratio["reference_versions"],
            "reference_s_per_frame_no_merge": ratio["reference_chain_no_merge_s_per_frame"],
            "port_s_per_frame_no_merge": ratio["oracle_chain_no_merge_s_per_frame"],
            "reference_merge_as_written_s_per_frame": ratio["reference_merge_as_written_s"],
            "reference_merge_regions": ratio["reference_merge_as_written_regions"],
            "port_s_per_frame_with_OR_merge": ratio["oracle_chain_with_OR_merge_s_per_frame"],
            "reference_over_port_no_merge": ratio["reference_over_oracle_no_merge"],
            "reference_with_merge_as_written_over_port": ratio["reference_with_merge_over_oracle_with_merge"],
            "reference_with_merge_Mpixels_per_s_per_core_estimate": round(
                H * W / 1e6 / (ratio["reference_chain_no_merge_s_per_frame"] + ratio["reference_merge_as_written_s"]), 4)}
    return block, checked, refs


def end_to_end_leg(args, stack, cell_types, pipe, dev, world, kernel_only_mpx, refs=None):
    """What a dataset run costs beyond the kernels (BASELINE configs 3 / 5 code path, per rank): `distributed.run_sharded`
    over a dataset of 2 x lanes batches per rank (the resident batch stands in for every batch: generation is not what is
    measured) = kernel chain + device-side table assembly + table download + the all-gather of every table; and the
    same chain fed over PCIe from pinned host memory through `ingest.FrameUploader` (double-buffered copies on a
    stream of their own).  Neither figure is `value` (inputs resident in HBM, by contract)."""
    import torch
    from particle_col_image_segmentation_amd.distributed import run_sharded
    from particle_col_image_segmentation_amd.ingest import FrameUploader
    B, C, H, W = stack.shape
    n_batches = max(6, 2 * max(1, getattr(pipe, "lanes", 1)))  # twice the pipeline's depth: steady state, not ramp
    n_frames = n_batches * B * world
    make_batch = lambda ids: stack[:len(ids)]
    # warm with a dataset of the same size: sort kernels, the table buffers of every lane's streams (the caching allocator
    # keeps a pool per stream) and the pinned staging buffers of the download (page-locking 270 MB costs 30 ms once)
    run_sharded(n_frames, make_batch, pipe, batch=B, device=dev, check=False)
    pipe.synchronize()
    t0 = time.perf_counter()
    tabs = run_sharded(n_frames, make_batch, pipe, batch=B, device=dev, check=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e2e = n_frames * H * W / dt / 1e6
    checked = None
    if refs:
        # the gathered tables themselves against the oracle (the CPU baseline's frames are the first frames of every batch of
        # this dataset: the resident batch stands in for all of them) -- groups rows and the group columns of `cells` of the
        # first AND the last batch, so the 1 024-frame run is not only timed
        from oracle import parity
        rank0 = int(os.environ.get("RANK", "0"))
        if rank0 == 0:
            ids0 = list(range(len(refs)))
            last = (n_batches - 1) * B * world  # rank 0 owns frames i with i % world == 0: its k-th frame has id k * world
            checked = parity.compare_tables(tabs, [i * world for i in ids0], refs)
            checked += parity.compare_tables(tabs, [last + i * world for i in ids0], refs)
    block = {"dataset_frames": n_frames, "batches_per_rank": n_batches, "value": round(e2e, 1), "unit": "Mpixels/s",
             "parity_checked_table_frames": checked,
             "includes": "kernel chain + device table assembly + table download + all-gather of rois/cells/groups/frames",
             "fraction_of_kernel_only": round(e2e / kernel_only_mpx, 3), "gathered_roi_rows": int(tabs["rois"].shape[0])}
    if world == 1:
        host = [stack.cpu().pin_memory() for _ in range(2)]
        dbuf = [torch.empty_like(stack) for _ in range(2)]  # two device buffers, refilled in turn (graph mode keys on them)
        up = FrameUploader((C, H, W), batch=B, device=dev)
        events, results = [None, None], [None, None]

        def feed(k):
            j = k % 2
            if results[j] is not None:  # the kernels that still read device buffer j: the copy waits for them on the GPU
                for ev in results[j].done_events():
                    up.copy_stream.wait_event(ev)
            if events[j] is not None:
                events[j].synchronize()  # staging buffer j is free again
            d, events[j] = up.upload_staged(host[j], out=dbuf[j])
            results[j] = pipe.run(d)

        for k in range(max(2, pipe.lanes)):  # warm: both staging slots, both device buffers, the graphs keyed on them
            feed(k)
        pipe.synchronize()
        up.bytes_uploaded = 0
        t0 = time.perf_counter()
        for k in range(n_batches):
            feed(k)
        pipe.synchronize()
        dt = time.perf_counter() - t0
        block["pcie_inclusive"] = {"value": round(n_batches * B * H * W / dt / 1e6, 1), "unit": "Mpixels/s",
                                   "host_to_device_GBps": round(up.bytes_uploaded / dt / 1e9, 2),
                                   "how": "%d batches from pinned host memory, copies on their own stream under the previous "
                                          "batch's kernels" % n_batches}
    return block


def secondary_leg(args, stack, cell_types, pipe):
    """The realistic worst case, on the record every round: the boundary plane as k/100 vote fractions (what a 100-tree
    random forest emits).  Equal-valued seeds and plateaus are then everywhere, no frame can be proven by the parallel
    flood and every frame runs the exact emulation of the reference's binary heap -- one wave per frame, sequential by
    definition, so throughput comes from frames in flight: ONE batch of `--secondary-batch` frames (the headline batch
    repeated), i.e. several frames per CU.  Two of its frames are checked against the oracle."""
    import torch
    from oracle import parity
    import gc
    B0 = stack.shape[0]
    reps = max(1, args.secondary_batch // B0)
    # everything the earlier legs still hold goes back to the driver BEFORE the 160 GB of this leg are allocated: the flood
    # is a latency-bound random walk over its heaps, and with the card's memory carved up by what earlier legs left behind
    # (captured graphs' pools, cached blocks) the same batch measured 215-250 Mpixels/s instead of 490 -- same kernels,
    # same results; with the other legs switched off it is 488-490 every time
    pipe.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    big = stack.repeat(reps, 1, 1, 1)
    n, H, W = big.shape[0], big.shape[2], big.shape[3]
    solo = type(pipe)(cell_types, lanes=1)
    res = solo.run(big)        # allocator priming on the tie-free frames (same sizes, milliseconds)
    res.synchronize()
    del res
    big[:, 3] = torch.round(big[:, 3] * 100) / 100
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    res = solo.run(big)
    res.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if int(res["overflow"].sum().item()) or int(res["ws_overflow"].sum().item()):
        raise SystemExit("bench.py: region table capacity exceeded (secondary leg)")
    ties = int(res["tie_flags"].sum().item())
    check = [0, n - 1]
    refs, _, _ = parity.run_oracle(big[check].cpu().numpy(), cell_types, merged=True, processes=2)
    checked = parity.compare(res, check, refs, sums_rtol=1e-6)
    return {"workload": "boundary plane quantised to k/100 (random-forest vote fractions): ONE batch of %d frames %dx%dx5 "
                        "(the headline batch repeated %d times), full kernel chain, inputs resident in HBM" % (n, H, W, reps),
            "value": round(n * H * W / dt / 1e6, 3), "unit": "Mpixels/s", "steps": 1, "ms_per_step": round(1e3 * dt, 1),
            "frames": n, "tie_fallback_frames": ties, "parity_checked_frames": checked,
            "peak_device_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}


class _StdoutToStderr:
    """Library banners (RCCL prints its host / library path on stdout at init) must not pollute the one JSON line:
    route file descriptor 1 to stderr while the benchmark runs, restore it for the final print."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def main():
    args = parse_args()
    with _StdoutToStderr():
        line = _run(args)
    if line is not None:
        print(line, flush=True)


def _timing_pass(lib, fn, sync):
    """Run ``fn`` with the library's per-launch HIP events on: returns ({kernel: (launches, total ms)},
    relaxation tiles processed, relaxation launches)."""
    lib.pcseg_watershed_counters(None, 1)
    lib.pcseg_timing_enable(1)
    fn()
    sync()
    nbytes = lib.pcseg_timing_report(None, 0)
    buf = ctypes.create_string_buffer(nbytes + 16)
    lib.pcseg_timing_report(buf, nbytes + 16)
    lib.pcseg_timing_enable(0)
    tiles = (ctypes.c_int64 * 4)()
    lib.pcseg_watershed_counters(tiles, 0)
    kernels = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.split("\t")
        kernels[name] = (int(calls), float(ms))
    return kernels, int(tiles[0]), int(tiles[1])


def graph_leg(args, stack, cell_types, eager_ms):
    """The same chain as ONE hipGraph replay per step (FramePipeline(graph=True) as the library ships it: two graphs in flight):
    what a host that cannot afford eight launch threads gets.  `fraction_of_headline` compares it with the timed region,
    whichever way that was launched (config.launch)."""
    import torch
    from particle_col_image_segmentation_amd.pipeline import FramePipeline
    pipe = FramePipeline(cell_types, graph=True)
    for _ in range(pipe.lanes + 2):  # the first pass through a lane captures its graph
        res = pipe.run(stack)
    pipe.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.graph_leg_steps):
        res = pipe.run(stack)
    pipe.synchronize()
    dt = time.perf_counter() - t0
    ms = 1e3 * dt / args.graph_leg_steps
    B, H, W = int(stack.shape[0]), int(stack.shape[2]), int(stack.shape[3])
    out = {"launch": "one hipGraph replay per step, %d in flight, one host thread" % pipe.lanes, "steps": args.graph_leg_steps,
           "ms_per_step": round(ms, 3), "value": round(B * H * W / ms / 1e3, 1), "unit": "Mpixels/s",
           "fraction_of_headline": round(eager_ms / ms, 3), "tie_fallback_frames": int(res["tie_flags"].sum().item())}
    del pipe, res
    torch.cuda.empty_cache()
    return out


def batch64_leg(args, stack, cell_types):
    """The quantised boundary plane (k/100 vote fractions) at BASELINE config 2's OWN batch size: every frame floods
    through equal-valued seeds, i.e. through the exact emulation of the reference's heap, one wave per frame -- the
    latency-bound operating point (the `secondary` leg is the throughput-bound one).  Bit-exact check of two frames."""
    import torch
    from oracle import parity
    from particle_col_image_segmentation_amd.pipeline import FramePipeline
    n = min(int(args.batch64_frames), int(stack.shape[0]))
    q = stack[:n].clone()
    q[:, 3] = torch.round(q[:, 3] * 100) / 100
    H, W = int(q.shape[2]), int(q.shape[3])
    solo = FramePipeline(cell_types, lanes=1)
    solo.run(stack[:n]).synchronize()  # allocator priming on the tie-free frames (same sizes, milliseconds)
    solo.synchronize()
    t0 = time.perf_counter()
    res = solo.run(q)
    res.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    check = [0, n - 1] if n > 1 else [0]
    refs, _, _ = parity.run_oracle(q[check].cpu().numpy(), cell_types, merged=True, processes=2)
    checked = parity.compare(res, check, refs, sums_rtol=1e-6)
    return {"workload": "boundary plane quantised to k/100: ONE batch of %d frames %dx%dx5, full kernel chain, inputs resident "
                        "in HBM (BASELINE config 2's shape on random-forest-like input)" % (n, H, W),
            "value": round(n * H * W / dt / 1e6, 3), "unit": "Mpixels/s", "steps": 1, "ms_per_step": round(1e3 * dt, 1),
            "frames": n, "tie_fallback_frames": int(res["tie_flags"].sum().item()), "parity_checked_frames": checked}


def shape_leg(name, B, H, W, cell_types, lib, dev, steps, seed):
    """BASELINE config 4 (one 4096 x 4096 x 5 mosaic) / config 5's frame shape (2048 x 2048 x 5): the full chain on that
    shape -- throughput with the default pipeline (8 batches in flight), the latency of ONE batch with nothing beside it,
    the serial kernel table's top five, and frame 0 against the oracle.  Never `value`."""
    import torch
    from oracle import parity
    from particle_col_image_segmentation_amd import synth
    from particle_col_image_segmentation_amd.pipeline import FramePipeline
    stack = synth.gen_batch_torch(seed, B, H, W, dev)
    cap = max(1024, (H * W) // 64)
    pipe = FramePipeline(cell_types)
    for _ in range(pipe.lanes + 2):
        res = pipe.run(stack)
    pipe.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = pipe.run(stack)
    pipe.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    if int(res["overflow"].sum().item()) or int(res["ws_overflow"].sum().item()):
        raise SystemExit("bench.py: region table capacity exceeded (%s leg)" % name)
    solo = FramePipeline(cell_types, overlap=False)
    solo.run(stack).synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        one = solo.run(stack)
    torch.cuda.synchronize()
    lat = 1e3 * (time.perf_counter() - t0) / 3
    kernels, _, _ = _timing_pass(lib, lambda: [solo.run(stack) for _ in range(2)], torch.cuda.synchronize)
    top = sorted(kernels.items(), key=lambda kv: -kv[1][1])[:5]
    total = sum(v[1] for v in kernels.values()) / 2
    refs, _, _ = parity.run_oracle(stack[:1].cpu().numpy(), cell_types, merged=True, processes=1)
    checked = parity.compare(one, [0], refs, sums_rtol=1e-6)
    out = {"workload": "%d x %dx%dx5 float32 per step, full kernel chain, inputs resident in HBM, region-table capacity %d"
                       % (B, H, W, cap),
           "value": round(B * H * W / ms / 1e3, 1), "unit": "Mpixels/s", "steps": steps, "ms_per_step": round(ms, 3),
           "launch": "eager launches from %d host threads (that many batches in flight)" % pipe.lanes,
           "one_batch_alone_ms": round(lat, 3), "one_batch_alone_Mpixels_per_s": round(B * H * W / lat / 1e3, 1),
           "serial_kernel_ms_per_step": round(total, 3),
           "top_kernels_serial": [{"kernel": k.lstrip("(").split(")")[0].split(" [")[0], "launches_per_step": c / 2, "ms_per_step": round(t / 2, 3)}
                                  for k, (c, t) in top],
           "tie_fallback_frames": int(one["tie_flags"].sum().item()), "parity_checked_frames": checked}
    del pipe, solo, res, one, stack
    torch.cuda.empty_cache()
    return out


def _run(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)  # torchrun, also with one rank
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from particle_col_image_segmentation_amd import _lib, synth
    from particle_col_image_segmentation_amd.distributed import all_gather_table
    from particle_col_image_segmentation_amd.pipeline import FramePipeline

    B, H, W = args.batch, args.size, args.size
    lib = _lib.load()
    ct = dict(synth.CELL_TYPES_5)
    stack = synth.gen_batch_torch(10_000 + rank * B, B, H, W, dev)
    if args.levels > 0:
        stack[:, 3] = torch.round(stack[:, 3] * args.levels) / args.levels
    graph = bool(args.graph) or args.launch == "graph"
    launch_probe = None
    auto = args.launch == "auto" and not args.graph and not args.serial and not args.lanes and not args.single_class_stream
    if auto:
        # SETUP: both launch modes, primed, then timed over a few untimed steps each (alternating, the better of two); the timed
        # region below runs the faster one.  With several ranks the slowest rank's figures decide, so every rank chooses alike.
        pipe_e = FramePipeline(ct, merged=not args.no_merge, watershed_mode=args.watershed_mode)
        pipe_g = FramePipeline(ct, graph=True, lanes=4, merged=not args.no_merge, watershed_mode=args.watershed_mode)
        for p_, n_ in ((pipe_e, max(2, pipe_e.lanes)), (pipe_g, pipe_g.lanes + 2)):
            for _ in range(n_):
                p_.run(stack)
            p_.synchronize()

        def probe_ms(p_, n_=8):
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(n_):
                p_.run(stack)
            p_.synchronize()
            return 1e3 * (time.perf_counter() - t_) / n_

        te, tg = probe_ms(pipe_e), probe_ms(pipe_g)
        te, tg = min(te, probe_ms(pipe_e)), min(tg, probe_ms(pipe_g))
        if use_dist:
            t = torch.tensor([te, tg], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            te, tg = float(t[0].item()), float(t[1].item())
        graph = tg < te
        launch_probe = {"eager_8_host_threads": round(te, 3), "graph_replays_4_in_flight": round(tg, 3)}
        pipe = pipe_g if graph else pipe_e
        del pipe_e, pipe_g, p_
        torch.cuda.empty_cache()
    else:
        pipe = FramePipeline(ct, overlap=not args.serial, lanes=args.lanes or (4 if (graph and args.launch == "graph") else None),
                             multi_stream=not args.single_class_stream, graph=graph, merged=not args.no_merge,
                             watershed_mode=args.watershed_mode)
    res = None
    # setup (not warmup): in graph mode the first pass through each lane captures its graph (one plain run + the capture);
    # in eager mode two priming passes fill torch's caching allocator with every workspace block.  Then W untimed warmup
    # steps and the K timed steps.
    for _ in range(max(2, pipe.lanes) + args.warmup):
        res = pipe.run(stack)
    pipe.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    # ---- the timed region: K steps, nothing but the chain (no event records, no counters read)
    lib.pcseg_timing_enable(0)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = pipe.run(stack)
    pipe.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations (rank 0): HIP events on the launch streams.  (a) the chain as the timed region runs it --
    # same streams, same number of batches in flight -- launched eagerly, because the kernels of a graph replay cannot be
    # bracketed one by one; (b) the same kernels with nothing beside them (one stream): a launch's duration under (a) says
    # how the CUs were shared, under (b) how good the kernel is.
    insitu = alone = None
    if rank == 0:
        steps_i = max(2, min(args.steps, 8))
        probe = FramePipeline(ct, overlap=not args.serial, lanes=pipe.lanes, multi_stream=not args.single_class_stream)
        for _ in range(pipe.lanes + 1):
            probe.run(stack)
        probe.synchronize()
        insitu = _timing_pass(lib, lambda: [probe.run(stack) for _ in range(steps_i)], probe.synchronize) + (steps_i,)
        if not args.serial:
            solo = FramePipeline(ct, overlap=False)
            solo.run(stack)
            torch.cuda.synchronize()
            alone = _timing_pass(lib, lambda: [solo.run(stack) for _ in range(2)], torch.cuda.synchronize) + (2,)
            del solo
        del probe

    # a table overflow would invalidate the measurement; frames on which the reference itself raises (clusters of a type
    # without a single cell -> int(NaN), tiff_analysis.py:776-781) are only counted: they cost the same work
    if int(res["overflow"].sum().item()) or int(res["ws_overflow"].sum().item()):
        raise SystemExit("bench.py: region table capacity exceeded")
    nan_frames = int(res["nan_flag"].sum().item())
    tie_frames = int(res["tie_flags"].sum().item())

    # the one exchange step of the path: all-gather of the per-ROI table (outside the timed region)
    t_tab = time.perf_counter()
    tables = pipe.tables_device(res, frame_ids=[rank * B + i for i in range(B)], check=nan_frames == 0)
    torch.cuda.synchronize()
    table_ms = 1e3 * (time.perf_counter() - t_tab)
    all_gather_table(tables["rois"])  # first call: communicator set-up / torch's sort kernels are loaded
    torch.cuda.synchronize()
    t_gat = time.perf_counter()
    gathered = all_gather_table(tables["rois"])  # device tensor straight into the collective (RCCL when world > 1)
    torch.cuda.synchronize()
    gather_ms = 1e3 * (time.perf_counter() - t_gat)
    n_rois = int(gathered.shape[0])

    # parity evidence of THIS run, before anything replays the pipeline's graphs again (a replay overwrites `res`):
    # world == 1: the CPU baseline's frames double as the check (32 by default); world > 1: no CPU baseline (rank 0 at
    # N = 1 only, by contract), but rank 0 still checks two frames of its timed batch against the oracle
    cpu_block, checked, cpu_refs = None, None, None
    if rank == 0 and not args.no_cpu_baseline:
        if world == 1:
            cpu_block, checked, cpu_refs = cpu_baseline(res, stack, ct, args.cpu_frames, tables)
        else:
            from oracle import parity
            refs, _, _ = parity.run_oracle(stack[[0, B - 1]].cpu().numpy(), ct, merged=True, processes=2)
            checked = parity.compare(res, [0, B - 1], refs, sums_rtol=1e-6)
    e2e_block = None
    legs_pipe = None
    if not args.serial and not args.no_end_to_end:
        # (the dataset legs refill rotating input buffers: they run the eager pipeline whichever way the timed region was launched)
        legs_pipe = pipe if not graph else FramePipeline(ct, merged=not args.no_merge, watershed_mode=args.watershed_mode)
        e2e_block = end_to_end_leg(args, stack, ct, legs_pipe, dev, world, world * B * H * W * args.steps / elapsed / 1e6,
                                   refs=cpu_refs if world == 1 else None)
    if rank == 0:
        kernels, tiles_i, launches_i, steps_i = insitu
        if args.kernel_table:
            for name, (calls, ms) in sorted(kernels.items(), key=lambda kv: -kv[1][1]):
                print("%10.3f ms %7d launches %9.2f us/launch  %s" % (ms, calls, 1e3 * ms / calls, name), file=sys.stderr)
        # the dominant kernel: by its time with nothing beside it (the serial table) when that table exists
        rank_table = alone[0] if alone is not None else kernels
        dom_name = max(rank_table.items(), key=lambda kv: kv[1][1])[0]
        if dom_name not in kernels:
            dom_name = max(kernels.items(), key=lambda kv: kv[1][1])[0]
        dom_calls, dom_ms = kernels[dom_name]
        short = dom_name.split("<")[0].split(" ")[0].strip("()")
        bpp = KERNEL_BYTES_PER_PIXEL.get(short, 0.0)

        # the relaxation is TWO kernels since round 4 (full grids for the first rounds, list walks for the late ones): its tile
        # counter and its launch counter cover both, so the roofline block prices them together
        def joint(kern):
            if short != "ws_relax_kernel":
                return kern[dom_name]
            parts = [v for k, v in kern.items() if k.lstrip("(").startswith(("ws_relax_kernel", "ws_relax_list_kernel"))]
            return sum(c for c, _ in parts), sum(m for _, m in parts)

        def launch_block(kern, tiles, launches, steps):
            calls, ms = joint(kern)
            avg_s = ms / calls / 1e3
            units = float(B * H * W)  # pixels one launch processes
            nbytes = bpp * units
            if short == "ws_relax_kernel" and launches:
                # the relaxation only touches marked tiles: count the 64x64 tiles it really processed; one launch per step
                # is the set-up round, which reads the three inputs (9 B/px) and writes value keys, seed labels and levels
                # (12 B/px) for every pixel instead of the 12 B/px of a plain round
                units = 4096.0 * tiles / launches
                nbytes = 12.0 * 4096.0 * tiles / launches + 9.0 * B * H * W * steps / launches
            return {"launches_per_step": calls / steps, "avg_launch_us": round(1e6 * avg_s, 2),
                    "algorithmic_bytes_per_launch": round(nbytes), "pixels_per_launch": round(units),
                    "achieved": round(nbytes / avg_s / 1e9, 2), "frac": round(nbytes / avg_s / 1e9 / HBM_PEAK_GBS, 5)}

        insitu_block = launch_block(kernels, tiles_i, launches_i, steps_i)
        insitu_block["how"] = ("%d steps of the same chain with %d batches in flight (the launches of a step then share the CUs "
                               "with seven other batches: a launch's duration here is a concurrency figure, not a kernel time -- "
                               "launches_per_step x avg_launch_us may exceed ms_per_step)" % (steps_i, pipe.lanes))
        main_block, stage_frac, ws_serial_ms = insitu_block, None, None
        if alone is not None and dom_name in alone[0]:
            # THE roofline figure: the kernel with nothing beside it (one stream), reproducible from profiles/
            main_block = launch_block(alone[0], alone[1], alone[2], alone[3])
            main_block["serial_kernel_ms_per_step"] = round(sum(ms for _, ms in alone[0].values()) / alone[3], 3)
            # the watershed stage as a whole: its compulsory bytes (13 B/px, once) over the serial time of all its kernels
            ws_serial_ms = sum(ms for name, (_, ms) in alone[0].items() if name.lstrip("(").startswith("ws_")) / alone[3]
            if ws_serial_ms > 0:
                stage_frac = WS_STAGE_BYTES_PER_PIXEL * B * H * W / (ws_serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        total_kernel_ms = sum(ms for _, ms in kernels.values())
        value = world * B * H * W * args.steps / elapsed / 1e6
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = tj.get(short)
            if short == "ws_relax_kernel" and tj.get("ws_relax_list_kernel") is not None and traffic is not None:
                # per launch of the pair, weighted by their launch counts in this run
                src_table = alone[0] if alone is not None else kernels
                n_full = sum(c for k, (c, _) in src_table.items() if k.lstrip("(").startswith("ws_relax_kernel"))
                n_list = sum(c for k, (c, _) in src_table.items() if k.lstrip("(").startswith("ws_relax_list_kernel"))
                if n_full + n_list:
                    traffic = round((traffic * n_full + tj["ws_relax_list_kernel"] * n_list) / (n_full + n_list))
        out = {
            "metric": "Mpixels/sec segmented (5-ch 1024x1024 TIFF), ROI mask IoU=1.0",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 planes / u8 class maps / int32 labels / f64 ROI sums", "data": "synthetic",
            "config": {"workload": ("BASELINE config 2" if not (args.levels or args.no_merge) else
                                    "NOT THE HEADLINE (levels=%d, merges %s)" % (args.levels, "off" if args.no_merge else "on"))
                                   + ": batch of %d frames %dx%dx5 float32 per GPU, full kernel chain, "
                                   "inputs resident in HBM" % (B, H, W),
                       "frames_per_gpu": B, "height": H, "width": W, "planes": 5, "parallelism": "frames x%d" % world,
                       "launch": ("one hipGraph replay per step, %d in flight" if graph else "eager launches from %d host threads") % pipe.lanes,
                       "tie_fallback_frames_last_step": tie_frames, "reference_nan_frames_rank0": nan_frames,
                       "gathered_roi_rows": n_rois, "table_assembly_ms_last_batch": round(table_ms, 3),
                       "roi_table_all_gather_ms": round(gather_ms, 3),
                       "launch_probe_ms_per_step": launch_probe},
            "roofline": dict({"bound": "hbm", "kernel": short if short != "ws_relax_kernel" else "ws_relax_kernel + ws_relax_list_kernel",
                              "algorithmic_bytes_per_pixel": bpp},
                             **{k: v for k, v in main_block.items() if k != "how"},
                             **{"peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic,
                                "how": ("HIP events on the launch stream, 2 extra steps of the same chain on ONE stream right after "
                                        "the timed region (nothing beside the kernel); `in_flight` = the same kernel while 8 batches "
                                        "share the chip") if main_block is not insitu_block else insitu_block["how"],
                                "in_flight": insitu_block if main_block is not insitu_block else None,
                                "stage": "watershed (ws_* kernels): %.1f B/px compulsory, once" % WS_STAGE_BYTES_PER_PIXEL,
                                "stage_serial_ms_per_step": None if ws_serial_ms is None else round(ws_serial_ms, 3),
                                "stage_frac": None if stage_frac is None else round(stage_frac, 5),
                                "share_of_kernel_time": round(joint(kernels)[1] / total_kernel_ms, 4),
                                "chain_bytes_per_pixel": CHAIN_BYTES_PER_PIXEL,
                                "chain_achieved_GBps": round(CHAIN_BYTES_PER_PIXEL * value / world / 1e3, 3),
                                "chain_frac": round(CHAIN_BYTES_PER_PIXEL * value / world / 1e3 / HBM_PEAK_GBS, 6)}),
        }
        if cpu_block is not None:
            out["cpu_baseline"] = cpu_block
        if checked is not None:
            out["config"]["parity_checked_frames"] = checked
        out["end_to_end"] = e2e_block
        if graph and launch_probe is not None:
            # the timed region ran as graph replays: their private pools go back before the other legs allocate (see
            # secondary_leg), and those legs take the eager pipeline
            import gc
            res = None
            pipe = legs_pipe if legs_pipe is not None else FramePipeline(ct, merged=not args.no_merge, watershed_mode=args.watershed_mode)
            gc.collect()
            torch.cuda.empty_cache()
        if world == 1 and not args.levels and not args.serial:
            if args.graph_leg_steps > 0 and not (graph and launch_probe is None):
                out["graph_replay"] = graph_leg(args, stack, ct, 1e3 * elapsed / args.steps)
            if args.batch64_frames > 0:
                out["batch64"] = batch64_leg(args, stack, ct)
            if not args.no_shape_legs:
                out["mosaic4096"] = shape_leg("mosaic4096", 1, 4096, 4096, ct, lib, dev, 16, 40_000)
                out["frames2048"] = shape_leg("frames2048", 16, 2048, 2048, ct, lib, dev, 8, 50_000)
            if args.secondary_batch > 0:
                out["secondary"] = secondary_leg(args, stack, ct, pipe)
        line = json.dumps(out)
    else:
        line = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
