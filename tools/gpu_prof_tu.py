#!/usr/bin/env python3
"""Cycle split inside hv_tu_job<N> (HEVCE_PROF_TU=N builds, libhevce_tuN.so): N from argv."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = int(sys.argv[1])
os.environ["HEVCE_LIB"] = os.path.join(ROOT, "hevc-image-encoder-lite_amd", "lib", f"libhevce_tu{N}.so")
os.environ["HEVCE_TRACE"] = "1"
import numpy as np
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
img = wl.gen_g2(128, 256, 16)
with hevce.Job([img.shape], 2) as job:
    job.upload(0, img); job.run()
    tr = job.trace(0)
p2 = tr[48:112].view(np.uint64)[16:22].astype(float)
calls = {32: 18 + 0, 16: 36 + 9 * 4, 8: 80 + 20 * 4}[N] * 32     # hv_tu_job<N> calls per picture (32 CTUs)
names = ("refs", "predict", "fwd cols", "fwd rows+quant", "inv cols", "inv rows+recon")
print(f"hv_tu_job<{N}>: {p2.sum()/calls/1e3:.1f} k cycles per call")
for nm, v in zip(names, p2): print(f"  {nm:16s} {v/calls/1e3:7.1f} k  {100*v/p2.sum():5.1f}%")
