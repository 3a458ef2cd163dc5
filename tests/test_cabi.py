"""CPU checks of the boundary: the C-ABI library loads and exports every symbol include/pcseg.h declares, the
ctypes table covers the header one to one, argument validation fails loudly, and no product module touches oracle/."""
import os
import re

import pytest

from conftest import ROOT


def _header_functions():
    text = open(os.path.join(ROOT, "include", "pcseg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcseg_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from particle_col_image_segmentation_amd import build
    build.build()
    from particle_col_image_segmentation_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from particle_col_image_segmentation_amd import _lib
    names = _header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names  # the ctypes binding covers the header exactly


def test_workspace_queries_and_argument_errors(lib):
    assert lib.pcseg_version() >= 100
    assert lib.pcseg_ccl_workspace_bytes(2, 64, 64) >= 2 * 64 * 64 * 4
    assert lib.pcseg_edt_workspace_bytes(1, 100, 100) > 0
    assert lib.pcseg_watershed_workspace_bytes(1, 64, 64) >= 64 * 64 * 20
    assert lib.pcseg_ccl_workspace_bytes(0, 64, 64) == 0
    rc = lib.pcseg_median5_u8(None, None, 1, 8, 8, None)
    assert rc == -1 and b"bad arguments" in lib.pcseg_last_error()
    rc = lib.pcseg_watershed4_f32(None, 0, None, None, None, None, 1, 8, 8, 0, None, 0, None)
    assert rc == -1


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from particle_col_image_segmentation_amd import tiff_analysis as ta
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ta.median_filter(np.ones((8, 8), np.uint8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "particle_col_image_segmentation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle interpreter", ""), os.path.join(dirpath, f)
