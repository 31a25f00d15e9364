#!/usr/bin/env python
"""Laboratory build: per-workgroup start / end of one sparse pillar launch (COALIGN_SPARSE_DEBUG=512), by XCD / CU."""
import ctypes, os, sys, collections
os.environ["COALIGN_LAB"] = "1"; os.environ.setdefault("COALIGN_SPARSE_DEBUG", "512")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
sys.argv = [sys.argv[0], "3", "pillar_sparse"]
exec(open(os.path.join(ROOT, "tools", "kernels_only.py")).read().split("if __name__")[0]) if False else None
import runpy
runpy.run_path(os.path.join(ROOT, "tools", "kernels_only.py"), run_name="__main__")
from coalign_amd import hip
L = hip.lib()
buf = np.zeros(2048 * 4, dtype=np.int64)
L.coalign_lab_sparse_trace.argtypes = [ctypes.c_void_p]
assert L.coalign_lab_sparse_trace(buf.ctypes.data) == 0
t = buf.reshape(2048, 4); t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
hw = t[:, 2] & 0xffffffff
xcc = (t[:, 2] >> 32) & 15
cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7          # gfx9 HW_ID fields
print(f"{len(t)} workgroups: start min {st.min():.2f} median {np.median(st):.2f} max {st.max():.2f} us; end min {en.min():.2f} median {np.median(en):.2f} p90 {np.percentile(en, 90):.2f} max {en.max():.2f} us")
print("lifetime us: min %.2f median %.2f p90 %.2f max %.2f" % ((en - st).min(), np.median(en - st), np.percentile(en - st, 90), (en - st).max()))
print("round 0 cycles: min %d median %d p90 %d max %d" % (t[:, 3].min(), np.median(t[:, 3]), np.percentile(t[:, 3], 90), t[:, 3].max()))
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
cnt = collections.Counter(key.tolist())
print(f"{len(cnt)} distinct CUs used; workgroups per CU: histogram {sorted(collections.Counter(cnt.values()).items())}")
last = {k: float(en[key == k].max()) for k in cnt}
for n in sorted(set(cnt.values())):
    v = [last[k] for k in cnt if cnt[k] == n]
    print(f"   CUs with {n} workgroups: {len(v)}; their last end: median {np.median(v):.2f} max {max(v):.2f} us")
order = np.argsort(en)
print("latest 10 workgroups: (block, start, end, key)", [(int(i), round(float(st[i]), 2), round(float(en[i]), 2), int(key[i])) for i in order[-10:]])
for q in range(0, len(t), max(1, len(t) // 16)):
    print(f"  block {q}: start {st[q]:.2f} end {en[q]:.2f} round0 {t[q, 3]} cycles")
