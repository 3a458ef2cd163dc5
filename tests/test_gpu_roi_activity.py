"""The MATLAB ROI activity / distance script restated on the HIP kernels, against the oracle's numpy restatement.
PARITY UNPINNED by the reference (no MATLAB here, no outputs in the repository)."""
import numpy as np
import pytest

from oracle import oracle as orc

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_activity_distance_table():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    from particle_col_image_segmentation_amd import roi_activity as ra
    rng = np.random.default_rng(5)
    H = W = 96
    red = np.zeros((H, W), bool)
    green = np.zeros((H, W), bool)
    for k in range(14):
        r, c = rng.integers(5, H - 8, 2)
        (red if k % 2 else green)[r:r + rng.integers(2, 6), c:c + rng.integers(2, 6)] = True
    green &= ~red
    planes = rng.poisson(30, (7, H, W)).astype(np.float32)
    agg = np.zeros((H, W), bool)
    agg[20:70, 25:80] = True
    agg[40:50, 40:50] = False
    out = ra.activity_distance_table(red, green, planes, agg)
    # oracle: MATLAB order = label of the transposed mask
    lr = orc.label(red.T).T
    lg = orc.label(green.T).T
    ta, xa = orc.roi_activity_table(lr, planes, 1)
    tb, xb = orc.roi_activity_table(lg, planes, 2)
    exp = np.concatenate([ta, tb])
    assert out["data"].shape == exp.shape and exp.shape[1] == 17
    np.testing.assert_allclose(out["data"], exp, rtol=1e-6)
    np.testing.assert_allclose(out["data_xy"][:, 17:], np.concatenate([xa, xb]), rtol=1e-12)
    near = orc.nearest_distances(xa, xb)
    np.testing.assert_allclose(out["data_dist_nearest"][:, 17], near, rtol=1e-12)
    # boundary points: mask pixels with a 4-neighbour outside the mask, (row, col) 1-based
    pad = np.pad(agg, 1)
    inner = pad[1:-1, 1:-1] & pad[:-2, 1:-1] & pad[2:, 1:-1] & pad[1:-1, :-2] & pad[1:-1, 2:]
    bd = np.argwhere(agg & ~inner) + 1.0
    allxy = np.concatenate([xa, xb])
    d = np.sqrt(((allxy[:, None, :] - bd[None, :, :]) ** 2).sum(-1)).min(1) / (512 / 19.0)
    np.testing.assert_allclose(out["data_dist_nearest_bound"][:, 18], d, rtol=1e-12)
