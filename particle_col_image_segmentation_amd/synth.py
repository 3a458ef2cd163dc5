"""Seeded synthetic 5-plane NanoSIMS-like frames.

One frame is a float32 stack ``(5, H, W)`` of per-pixel class probabilities
``(cell A, cell B, particle, boundary, background)`` that sum to one:

* class map   := ``argmax + 1`` (what ilastik "Simple Segmentation" exports and
  what ``tiff_analysis.py:639-643`` of the reference reads from its ``.h5``),
* boundary map := plane 3 (``refine_boundaries.py:34`` of the reference),
* isotope planes := the same five planes (per-ROI sums, ``.m:122-170``).

The numpy generator is plain numpy (no scipy) so that it runs unchanged under
the oracle interpreter that makes ``tests/golden`` and under the system python.
The torch twin builds frames on the device for the benchmark (same model, its
own random stream); it exists so that large batches never cross PCIe.
"""
import numpy as np

CELL_TYPES_5 = {1: "3D05", 2: "6B07", 3: "Particle", 4: "Boundary", 5: "Background"}
N_PLANES = 5
BOUNDARY_PLANE = 3


def _disc_params(rng, H, W):
    n = max(1, (H * W) // 1000)
    cy = rng.uniform(10.0, H - 10.0, n) if H > 20 else rng.uniform(0, H, n)
    cx = rng.uniform(10.0, W - 10.0, n) if W > 20 else rng.uniform(0, W, n)
    rad = rng.integers(3, 12, n).astype(np.float64)
    typ = rng.integers(0, 2, n)
    return cy, cx, rad, typ


def signed_distance_fields(H, W, cy, cx, rad, typ, margin=12):
    """sd[t] = min over discs of type t of (|p - c| - r); clipped at +margin."""
    sd = np.full((2, H, W), float(margin), dtype=np.float64)
    for y, x, r, t in zip(cy, cx, rad, typ):
        r0 = int(max(0, np.floor(y - r - margin)))
        r1 = int(min(H, np.ceil(y + r + margin) + 1))
        c0 = int(max(0, np.floor(x - r - margin)))
        c1 = int(min(W, np.ceil(x + r + margin) + 1))
        if r1 <= r0 or c1 <= c0:
            continue
        yy = np.arange(r0, r1, dtype=np.float64)[:, None] - y
        xx = np.arange(c0, c1, dtype=np.float64)[None, :] - x
        d = np.sqrt(yy * yy + xx * xx) - r
        np.minimum(sd[t, r0:r1, c0:c1], d, out=sd[t, r0:r1, c0:c1])
    return sd


def planes_from_fields(sd, sd_p, noise, salt_plane, salt_amp=6.0):
    """Logits -> softmax planes.  ``noise`` is (5,H,W) in [0,1); ``salt_plane``
    is (H,W) int8 with -1 = no salt, else the plane that receives a spike."""
    m_a = np.clip(0.5 - sd[0] / 2.0, 0.0, 1.0)
    m_b = np.clip(0.5 - sd[1] / 2.0, 0.0, 1.0)
    m_p = np.clip(0.5 - sd_p / 2.0, 0.0, 1.0)
    edge = np.exp(-np.abs(np.minimum(sd[0], sd[1])) / 2.0)
    logits = np.empty((N_PLANES,) + sd_p.shape, dtype=np.float64)
    logits[0] = 3.0 * m_a
    logits[1] = 3.0 * m_b
    logits[2] = 2.0 * m_p
    logits[3] = 4.0 * edge
    logits[4] = 1.0
    logits += 0.1 * noise
    for k in range(3):
        logits[k] += salt_amp * (salt_plane == k)
    logits -= logits.max(axis=0, keepdims=True)
    e = np.exp(logits)
    p = e / e.sum(axis=0, keepdims=True)
    return 0.02 + 0.96 * p


def gen_frame(seed, H, W, ties=False):
    """Return the float32 stack (5,H,W) of frame ``seed``.

    ``ties=True`` quantises to 1/100 like random-forest vote fractions, which
    makes equal-valued watershed seeds and plateaus ubiquitous."""
    rng = np.random.default_rng(seed)
    cy, cx, rad, typ = _disc_params(rng, H, W)
    sd = signed_distance_fields(H, W, cy, cx, rad, typ)
    yy = np.arange(H, dtype=np.float64)[:, None] - H / 2.0
    xx = np.arange(W, dtype=np.float64)[None, :] - W / 2.0
    sd_p = np.sqrt(yy * yy + xx * xx) - 0.3 * H
    noise = rng.random((N_PLANES, H, W))
    salt = rng.random((H, W)) < 0.01
    salt_plane = np.where(salt, rng.integers(0, 3, (H, W)), -1).astype(np.int8)
    p = planes_from_fields(sd, sd_p, noise, salt_plane)
    if ties:
        p = np.round(p * 100.0) / 100.0
    return p.astype(np.float32)


def gen_batch(base_seed, B, H, W, ties=False):
    return np.stack([gen_frame(base_seed + i, H, W, ties) for i in range(B)])


def class_map_from_stack(stack):
    """argmax+1 as uint8 (first maximum wins, like numpy.argmax)."""
    return (np.argmax(stack, axis=-3) + 1).astype(np.uint8)


def gen_batch_torch(base_seed, B, H, W, device, ties=False, chunk=32):
    """Device twin of :func:`gen_batch` (same model, torch random stream).

    The signed distance to the nearest disc edge is a chunked min over discs, entirely on the device, so that a
    64 x 1024 x 1024 x 5 batch is built in about a second and never crosses PCIe."""
    import torch

    g = torch.Generator(device="cpu")
    out = torch.empty((B, N_PLANES, H, W), dtype=torch.float32, device=device)
    ar_y = torch.arange(H, dtype=torch.float32, device=device)[None, :, None]
    ar_x = torch.arange(W, dtype=torch.float32, device=device)[None, None, :]
    sd_p = torch.sqrt((ar_y[0] - H / 2.0) ** 2 + (ar_x[0] - W / 2.0) ** 2) - 0.3 * H
    m_p = torch.clamp(0.5 - sd_p / 2.0, 0.0, 1.0)
    margin = 12.0
    n = max(1, (H * W) // 1000)
    for b in range(B):
        g.manual_seed(int(base_seed) + b)
        cy = (10.0 + torch.rand(n, generator=g) * (H - 20.0)).to(device)
        cx = (10.0 + torch.rand(n, generator=g) * (W - 20.0)).to(device)
        rad = torch.randint(3, 12, (n,), generator=g).to(device=device, dtype=torch.float32)
        typ = torch.randint(0, 2, (n,), generator=g).to(device)
        sd = torch.full((2, H, W), margin, dtype=torch.float32, device=device)
        for t in range(2):
            idx = torch.nonzero(typ == t)[:, 0]
            for i in range(0, idx.numel(), chunk):
                j = idx[i:i + chunk]
                d = torch.sqrt((ar_y - cy[j, None, None]) ** 2 + (ar_x - cx[j, None, None]) ** 2) - rad[j, None, None]
                sd[t] = torch.minimum(sd[t], d.amin(dim=0))
        dg = torch.Generator(device=device)
        dg.manual_seed(int(base_seed) + b)
        logits = 0.1 * torch.rand((N_PLANES, H, W), generator=dg, device=device)
        logits[0] += 3.0 * torch.clamp(0.5 - sd[0] / 2.0, 0.0, 1.0)
        logits[1] += 3.0 * torch.clamp(0.5 - sd[1] / 2.0, 0.0, 1.0)
        logits[2] += 2.0 * m_p
        logits[3] += 4.0 * torch.exp(-torch.abs(torch.minimum(sd[0], sd[1])) / 2.0)
        logits[4] += 1.0
        salt = torch.rand((H, W), generator=dg, device=device) < 0.01
        which = torch.randint(0, 3, (H, W), generator=dg, device=device)
        for k in range(3):
            logits[k] += 6.0 * (salt & (which == k))
        p = torch.softmax(logits, dim=0)
        p = 0.02 + 0.96 * p
        if ties:
            p = torch.round(p * 100.0) / 100.0
        out[b] = p
    return out
