"""Basic-block summary of one kernel's gfx950 ISA (instruction mix per block, and the issue sequence of the matrix-heavy blocks).

    python tools/isa_blocks.py <file.s> <mangled-name substring> [min_mfma_for_sequence]
(`hipcc ... --cuda-device-only -S` produces the .s; runs without a GPU.)"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    sub = sys.argv[2]
    min_m = int(sys.argv[3]) if len(sys.argv) > 3 else 27
    start = next(m.start() for m in re.finditer(r"^(\S*" + re.escape(sub) + r"\S*):", s, re.M))
    body = s[start:]
    body = body[:body.index("s_endpgm")]
    blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
    for i in range(1, len(blocks), 2):
        b = blocks[i + 1]
        n_m = len(re.findall(r"v_mfma", b))
        if n_m or "global_load_lds" in b or "scratch_" in b:
            print(blocks[i], "mfma", n_m, "ds_read_b128", len(re.findall("ds_read_b128", b)), "waitcnt", len(re.findall("s_waitcnt", b)), "lds_dma",
                  len(re.findall("global_load_lds", b)), "valu", len(re.findall(r"\n\s+v_(?!mfma)", b)), "salu", len(re.findall(r"\n\s+s_(?!waitcnt|nop)", b)),
                  "nop", len(re.findall(r"s_nop", b)), "scratch", len(re.findall(r"scratch_", b)), "lines", b.count("\n"))
        if n_m >= min_m:
            seq = []
            for l in b.split("\n"):
                l = l.strip()
                if l.startswith("v_mfma"):
                    seq.append("M")
                elif l.startswith("ds_read"):
                    seq.append("r")
                elif l.startswith("s_waitcnt"):
                    seq.append("[" + l.split(None, 1)[1].split(";")[0].strip() + "]")
                elif l.startswith("s_nop"):
                    seq.append("n")
                elif l.startswith("global_load_lds"):
                    seq.append("D")
                elif l.startswith("s_barrier"):
                    seq.append("|B|")
                elif l.startswith("v_"):
                    seq.append("v")
                elif l.startswith("s_"):
                    seq.append("s")
            print("   ", "".join(seq))


if __name__ == "__main__":
    main()
