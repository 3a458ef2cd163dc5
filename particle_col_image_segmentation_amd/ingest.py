"""Frame ingest in front of the GPU path: pages written by ``split_zstack.process_tif`` -> frames ->
device batches for ``FramePipeline.run``, through pinned host memory and a copy stream of its own.

The reference reads one image at a time from disk (``tiff_analysis.py:118-121, 639-642``) and has no upload step;
this is the build's host side of that boundary.  ``FrameUploader`` keeps ``depth`` pinned staging buffers: while the
asynchronous copy of one batch is in flight (and the GPU works on the batch before it) the host decodes the next
batch's pages into the other buffer, so decode, PCIe transfer and kernels overlap without any host thread.
"""
import numpy as np
import torch

from . import tiffio


def frames_from_pages(written_files, n_planes):
    """Group the single-page TIFFs of :func:`split_zstack.process_tif` (file order: slice-major, selected channels
    inside a slice, ``<stem>_z<i>_<CH>.tif``, split_zstack.py:57-65) into ``(n_planes, H, W)`` float32 frames."""
    if len(written_files) % n_planes:
        raise ValueError("%d pages do not make whole frames of %d planes" % (len(written_files), n_planes))
    for i in range(0, len(written_files), n_planes):
        yield np.stack([np.asarray(tiffio.imread(p), dtype=np.float32) for p in written_files[i:i + n_planes]])


class FrameUploader:
    def __init__(self, frame_shape, batch, device, depth=2):
        if not torch.cuda.is_available():
            raise RuntimeError("FrameUploader needs a ROCm GPU (no CPU fallback)")
        self.frame_shape = tuple(int(v) for v in frame_shape)
        self.batch = int(batch)
        self.device = torch.device(device)
        self.depth = max(2, int(depth))
        self.pinned = [torch.empty((self.batch,) + self.frame_shape, dtype=torch.float32).pin_memory() for _ in range(self.depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.copied = [torch.cuda.Event() for _ in range(self.depth)]  # "the copy out of pinned buffer k has finished"
        self.bytes_uploaded = 0

    def upload_staged(self, stage, out=None):
        """Asynchronous copy of a batch that already sits in pinned host memory (a reader that decodes straight into
        pinned buffers): returns the device tensor, readable by the caller's current stream; the caller must not
        overwrite ``stage`` before the returned event has completed.  ``out``: a device buffer to refill instead of a
        fresh allocation (``FramePipeline(graph=True)`` keys its graphs on the input buffer, so a dataset run refills a
        few fixed buffers in turn); the caller makes ``copy_stream`` wait for whatever still reads it.
        Returns (device stack, copy-finished event)."""
        if not stage.is_pinned():
            raise ValueError("upload_staged needs pinned host memory")
        consumer = torch.cuda.current_stream(self.device)
        done = torch.cuda.Event()
        with torch.cuda.stream(self.copy_stream):
            if out is None:
                dev = torch.empty(tuple(stage.shape), dtype=stage.dtype, device=self.device)
            else:
                if tuple(out.shape) != tuple(stage.shape) or out.dtype != stage.dtype or not out.is_cuda:
                    raise ValueError("upload_staged: `out` must be a device tensor of the staging buffer's shape and dtype")
                self.copy_stream.wait_stream(consumer)  # whatever the caller's stream last did with the buffer
                dev = out
            dev.copy_(stage, non_blocking=True)
            done.record(self.copy_stream)
        consumer.wait_event(done)
        dev.record_stream(consumer)
        self.bytes_uploaded += dev.numel() * dev.element_size()
        return dev, done

    def batches(self, frames):
        """``frames``: iterable of ``frame_shape`` float32 arrays (numpy or CPU tensors).  Yields ``(stack, n)``:
        a ``(n, C, H, W)`` float32 CUDA tensor that the caller's current stream may read at once (the stream is made to
        wait for the copy), ``n <= batch`` frames."""
        it = iter(frames)
        k = 0
        exhausted = False
        while not exhausted:
            slot = k % self.depth
            if k >= self.depth:
                self.copied[slot].synchronize()  # the copy that last read this staging buffer
            stage = self.pinned[slot]
            n = 0
            for frame in it:
                t = frame if isinstance(frame, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frame, dtype=np.float32))
                if tuple(t.shape) != self.frame_shape:
                    raise ValueError("frame of shape %s, expected %s" % (tuple(t.shape), self.frame_shape))
                stage[n].copy_(t)
                n += 1
                if n == self.batch:
                    break
            else:
                exhausted = True
            if n == 0:
                break
            consumer = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.copy_stream):
                dev = torch.empty((n,) + self.frame_shape, dtype=torch.float32, device=self.device)
                dev.copy_(stage[:n], non_blocking=True)
                self.copied[slot].record(self.copy_stream)
            consumer.wait_event(self.copied[slot])
            dev.record_stream(consumer)
            self.bytes_uploaded += dev.numel() * 4
            yield dev, n
            k += 1
