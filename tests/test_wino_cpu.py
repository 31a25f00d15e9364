"""CPU tests of the Winograd convolution's host side: the operand image of ``ops.pack_conv3x3_wino_weight`` read back with the kernel's own
index formulas (tests/wino_emulation.py) reproduces the float64 convolution; the C ABI exports the two entry points."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from coalign_amd import hip, ops
from wino_emulation import emulate


@pytest.mark.parametrize("case", [(2, 16, 64, 5, 6, 8, True), (1, 32, 128, 6, 35, 16, False), (3, 16, 64, 9, 18, 8, True), (2, 16, 64, 4, 20, 16, True)])
def test_winograd_data_flow_on_the_host(case):
    """Stacked images (odd and even heights), both tile-block shapes, the weight image in wavefront load order, the producer's row / column
    selection, the two-level output transform: max error vs the float64 convolution = the bf16x3 rounding of U (~3e-8 of the scale)."""
    N, Ci, Co, H, W, tbw, res = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    r = torch.randn(N, Co, H, W, generator=g) if res else None
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res:
        want = want + r.double()
    want = torch.relu(want).permute(0, 2, 3, 1).numpy()
    u = ops.pack_conv3x3_wino_weight(w)
    got = emulate(x.permute(0, 2, 3, 1).double().numpy(), u, b.double().numpy(), None if r is None else r.permute(0, 2, 3, 1).double().numpy(), True, tbw)
    assert np.abs(got - want).max() < 2e-7 * np.abs(want).max()


def test_winograd_weight_image_size_and_argument_checks():
    L = hip.lab_lib()          # (the Winograd kernel lives in the laboratory library: include/coalign_amd_lab.h)
    assert L.coalign_conv3x3_wino_weight_bytes(64, 64) == 16 * 64 * 64 * 6 + 16
    assert L.coalign_conv3x3_wino_weight_bytes(24, 64) == 0 and L.coalign_conv3x3_wino_weight_bytes(64, 96) == 0
    with pytest.raises(ValueError):
        ops.pack_conv3x3_wino_weight(torch.zeros(64, 8, 3, 3))
    with pytest.raises(hip.CoalignHipError):
        ops.conv3x3_wino(torch.zeros(1, 16, 4, 4), torch.zeros(1, dtype=torch.uint8), torch.zeros(64), 64)
