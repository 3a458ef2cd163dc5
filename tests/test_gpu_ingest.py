"""SURVEY.md 8f-1, second half: z-stack TIFF -> split_zstack.process_tif -> pages -> pinned double-buffered upload ->
FramePipeline must give exactly what the pipeline gives for the in-memory stack; plus BASELINE config 4 (one
4096x4096x5 mosaic) through the FULL pipeline against the oracle."""
import os

import numpy as np
import pytest

from oracle import parity

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_zstack_split_upload_pipeline_equals_in_memory(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path has no CPU fallback")
    from particle_col_image_segmentation_amd import ingest, synth, tiffio
    from particle_col_image_segmentation_amd import split_zstack as sz
    from particle_col_image_segmentation_amd.pipeline import FramePipeline
    Z, C, H, W = 7, 5, 96, 80
    stack = synth.gen_batch(4100, Z, H, W)  # (Z, C, H, W) float32
    sample = tmp_path / "plate" / "run1"
    sample.mkdir(parents=True)
    src = str(sample / "stack_zstack.tif")
    tiffio.imwrite(src, stack)
    names = {k: "P%d" % k for k in range(C)}
    written = sz.process_tif(src, list(range(C)), channel_map=names)
    assert len(written) == Z * C
    assert os.path.basename(written[0]) == "stack_zstack_z0_P0.tif" and os.path.basename(written[-1]) == "stack_zstack_z%d_P4.tif" % (Z - 1)
    ct = dict(synth.CELL_TYPES_5)
    pipe = FramePipeline(ct)
    up = ingest.FrameUploader((C, H, W), batch=3, device="cuda", depth=2)
    results, sizes = [], []
    for dev, n in up.batches(ingest.frames_from_pages(written, C)):
        results.append(pipe.run(dev))
        sizes.append(n)
    assert sizes == [3, 3, 1] and up.bytes_uploaded == stack.nbytes
    ref = FramePipeline(ct, overlap=False).run(torch.from_numpy(stack).cuda())
    torch.cuda.synchronize()
    z0 = 0
    for res, n in zip(results, sizes):
        for key in ("denoised", "labels", "counts", "recreated", "markers", "ws_labels", "n_markers", "overlap_area"):
            assert torch.equal(res[key], ref[key][z0:z0 + n]), key
        z0 += n


def test_config4_mosaic_4096_full_pipeline_vs_oracle():
    """One 4096x4096x5 frame, untiled (every kernel takes the frame whole), full chain incl. merge, particle fill, ROI
    sums and the device-side tables."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from particle_col_image_segmentation_amd import synth
    from particle_col_image_segmentation_amd.pipeline import FramePipeline
    ct = dict(synth.CELL_TYPES_5)
    stack = synth.gen_batch_torch(31_000, 1, 4096, 4096, torch.device("cuda", 0))
    pipe = FramePipeline(ct, cap=1 << 17)
    res = pipe.run(stack)
    res.synchronize()
    assert int(res["overflow"].sum()) == 0 and int(res["ws_overflow"].sum()) == 0
    refs, _, _ = parity.run_oracle(stack.cpu().numpy(), ct, merged=True, processes=1)
    assert parity.compare(res, [0], refs, sums_rtol=1e-6) == 1
    tabs = pipe.tables(res, check=not refs[0]["nan"])
    assert tabs["rois"].shape[0] == int((refs[0]["roi_area"] > 0).sum())
    assert tabs["frames"][0, 1] == refs[0]["n_labels"] and tabs["frames"][0, 2] == refs[0]["n_markers"]
