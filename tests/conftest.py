import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The CPU oracle runs tiny batched matmuls; on a many-core host (the GPU box) torch's default of one thread per
# core makes them pathologically slow, so cap the intra-op threads for the whole test session.
import torch  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is present, so a plain `pytest tests/` works here."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(params=[3, 0, 16], ids=["bf16x3", "native_fp32", "fp16x2"])
def conv_mode(request):
    """Model-level parity tests run once per 3x3-convolution arithmetic: the product default (16: sp16 pairs on the fp16 matrix cores, SplitMaps between
    the layers of a stage), the 3-way bf16 split of rounds 2-3 and the native-fp32 route (COALIGN_CONV_EMU=0); all are held to the same tolerances."""
    from coalign_amd import backbone
    saved = backbone.CONV_EMU_TERMS
    backbone.CONV_EMU_TERMS = request.param
    yield request.param
    backbone.CONV_EMU_TERMS = saved


def assert_elementwise(got, ref, what="", rtol=1e-4, floor=1e-5):
    """End-to-end tensors are compared ELEMENT-WISE, not by one max-norm scalar: |got - ref| <= rtol * |ref| + floor * max |ref| for every element
    (the bound of tests/test_hip_parity.py::feat_close: an order of magnitude inside the north star's 1e-3).  Returns max |diff| / max |ref| for the callers' printouts."""
    import torch
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, tuple(got.shape), tuple(ref.shape))
    scale = max(float(ref.abs().max()), 1e-30)
    err = (got - ref).abs()
    bad = err > rtol * ref.abs() + floor * scale
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} elements outside rtol {rtol} + {floor} of the scale; worst {float(err.max()) / scale:.3e} of the scale"
    return float(err.max()) / scale
