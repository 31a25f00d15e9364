"""Register / scratch / occupancy table of every kernel of one csrc/ source, from hipcc's -Rpass-analysis=kernel-resource-usage remarks (runs without a GPU).

    python tools/kernel_resources.py conv3x3_emu.hip [substring]
"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coalign_amd import build as B  # noqa: E402


def main():
    src = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = [B._hipcc(), "-x", "hip"] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + (["-DCOALIGN_LAB"] if os.environ.get("LAB") else []) + \
          ["--cuda-device-only", "-c", os.path.join(B.CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, stdin=subprocess.DEVNULL)
    text = res.stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    if res.returncode != 0 or not blocks:
        print("\n".join(l for l in text.splitlines() if "error" in l or "Error" in l)[:4000] or text[-2000:])
        raise SystemExit(f"compile failed (rc {res.returncode})")
    names = [b.split("\n")[0].split()[0] for b in blocks]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=60).stdout.strip().split("\n")
    for b, d in zip(blocks, dem):
        if sub not in d:
            continue
        g = lambda k: re.search(r" " + k + r": (\d+)", b).group(1)
        short = re.sub(r"\(anonymous namespace\)::", "", d)
        short = re.sub(r"\(.*$", "", short)
        scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{short:70s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} spill {g('VGPRs Spill'):>3} scratch {scratch:>4} waves/SIMD {occ} LDS {lds}")


if __name__ == "__main__":
    main()
