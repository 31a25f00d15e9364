"""coalign_amd -- MI355X (gfx950) implementation of the CoAlign per-frame detection hot path.

Host side (this package, Python on PyTorch-ROCm) mirrors the opencood model / post-processor API; the four hot
ops are hand-written HIP kernels behind the C ABI in ``include/coalign_amd.h`` (``coalign_amd/csrc``).
See DESIGN.md and INTEGRATION.md at the repository root.
"""
__version__ = "0.1.0"
