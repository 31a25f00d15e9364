"""Build the gfx950 shared library ``coalign_amd/lib/libcoalign_hip.so`` with hipcc (cross-compiles without a GPU).

    python -m coalign_amd.build [--force]

One translation unit per kernel family; objects are cached under ``coalign_amd/csrc/_obj``.  Flags:
``-ffp-contract=off`` so float arithmetic rounds like the reference's op-by-op evaluation (kernels use
explicit ``fmaf`` where fusing is wanted) and so the float64 NMS clipping is bit-identical to the gcc-built oracle.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcoalign_hip.so")
# The LABORATORY build (tools/ only): the same sources with -DCOALIGN_LAB, which compiles in the ablation / debug switches and the measured-and-rejected
# kernel variants (read from COALIGN_* environment variables).  The product library above contains none of them; `python -m coalign_amd.build --lab`
# builds it, COALIGN_LAB=1 makes coalign_amd.hip load it.
LAB_LIB_PATH = os.path.join(LIB_DIR, "libcoalign_hip_lab.so")
# A second laboratory build WITHOUT -fno-slp-vectorize / -fno-vectorize (packed fp32 instructions allowed again): exists only to re-examine round 3's
# "packed fp32 beside matrix wavefronts" finding (the inline-asm LDS-DMA's missing "m0" clobber, fixed in the same commit, is the likelier
# cause).  `python -m coalign_amd.build --labvec`; COALIGN_LAB=vec loads it.  tools/pk_f32_recheck.sh runs the experiment.
LABVEC_LIB_PATH = os.path.join(LIB_DIR, "libcoalign_hip_labvec.so")
INCLUDE = os.path.join(REPO, "include")
SOURCES = ["status.cpp", "pillar_scatter.hip", "pillar_sparse.hip", "warp_fuse.hip", "warp_fuse_nhwc.hip", "decode.hip", "nms.hip", "epilogue.hip", "voxelize.hip", "pose_graph.hip", "conv3x3.hip", "conv3x3_emu.hip", "conv3x3_sp.hip", "conv3x3_sp_s2.hip", "pointwise.hip"]
# Kernels that were measured and NOT adopted live in the laboratory library only: the Winograd F(2x2, 3x3) convolution (1.14 x / 0.89 x / 1.05 x
# against the direct kernel per stage) -- include/coalign_amd_lab.h, coalign_amd.hip.lab_lib(), tests/test_round4_gpu.py keep it testable.
LAB_ONLY_SOURCES = ["conv3x3_wino.hip"]
ARCH = "gfx950"
# per-source extras.  pillar_scatter.hip: its matrix-core encoder reduces the accumulators with VALU right after each instruction
# pair -- results in VGPRs (not AGPRs) save 64 v_accvgpr_read per pass; -fno-honor-nans drops the canonicalising v_max the compiler
# puts in front of every two-operand fmaxf of a raw matrix result (20 per pass; the file tests for NaN nowhere, its selects are explicit).
# -fno-slp-vectorize -fno-vectorize (all sources): no packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 formed from scalar
# code by the SLP vectoriser or by loop interleaving; `packed_fp32_count` below checks a source's ISA).  Measured
# on the MI355X (profiles/round3/README.md, "packed fp32 beside matrix wavefronts"): a wavefront executing them while it shares a SIMD with
# three densely issuing matrix wavefronts of another kernel got wrong results in lanes 48-63.  Frame rate with / without the flag: equal
# (309-314 vs 307-312 frames/s, same box, alternating).  (The fp32 VALU pillar encoder's explicit float2 arithmetic was rewritten as scalar chains.)
# Status of that finding: observed twice -- round 3, and round 4's re-run with the inline-asm m0 clobber in place (0 / 2700 vs 356 / 2700 fused
# maps differing, profiles/round4/experiments/pk_f32_recheck.txt) -- and mitigated by these flags; there is no standalone victim / aggressor reproducer, so it
# is recorded as an observation with a mitigation, not as a characterised hardware behaviour.
EXTRA_FLAGS = {"pillar_scatter.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
               "pillar_sparse.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"]}      # (no NaN test may be written in these two files)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-vectorize", "-Wno-cuda-compat", "-Wno-inline-asm", f"--offload-arch={ARCH}", f"-I{INCLUDE}", f"-I{CSRC}"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 extension cannot be built on this machine")
    return exe


def _newest(paths) -> float:
    return max(os.path.getmtime(p) for p in paths)


def build(force: bool = False, verbose: bool = False, lab: bool = False, vectorize: bool = False) -> str:
    OBJ = os.path.join(CSRC, "_obj_labvec" if vectorize else "_obj_lab" if lab else "_obj")
    LIB_PATH = LABVEC_LIB_PATH if vectorize else LAB_LIB_PATH if lab else globals()["LIB_PATH"]
    lab = lab or vectorize
    flags = [f for f in FLAGS if not (vectorize and f in ("-fno-slp-vectorize", "-fno-vectorize"))]
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(INCLUDE, "coalign_amd.h"), os.path.join(INCLUDE, "coalign_amd_lab.h"), os.path.join(CSRC, "common.h"), os.path.abspath(__file__)]
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        if force or not os.path.exists(op) or os.path.getmtime(op) < _newest([sp] + headers):
            cmd = [hipcc, "-x", "hip"] + flags + (["-DCOALIGN_LAB"] if lab else []) + EXTRA_FLAGS.get(src, []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return op

    sources = SOURCES + (LAB_ONLY_SOURCES if lab else [])
    with ThreadPoolExecutor(max_workers=len(sources)) as ex:
        objs = list(ex.map(compile_one, sources))
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < _newest(objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}"] + objs + ["-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, lab="--lab" in sys.argv, vectorize="--labvec" in sys.argv))


def packed_fp32_count(src: str) -> int:
    """Number of packed fp32 arithmetic instructions in the gfx950 ISA hipcc emits for ``csrc/<src>`` with the library's flags (0 expected:
    profiles/round3/README.md).  Compiles to assembly only; takes seconds to a minute per source."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [_hipcc(), "-x", "hip"] + FLAGS + EXTRA_FLAGS.get(src, []) + ["--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return len(re.findall(r"\bv_pk_(?:add|mul|fma)_f32\b", open(out).read()))
