#!/usr/bin/env python3
"""Cycle split of the wave-local binarisation (hv_bin_own) per CU level, job wavefront 1 (HEVCE_PROF + HEVCE_PROF_BIN build,
libhevce_prof2.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HEVCE_LIB"] = os.path.join(ROOT, "hevc-image-encoder-lite_amd", "lib", "libhevce_prof2.so")
os.environ["HEVCE_TRACE"] = "1"
import numpy as np
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
img = wl.gen_g2(128, 256, 16)
with hevce.Job([img.shape], 2) as job:
    job.upload(0, img); job.run()
    tr = job.trace(0)
p2 = tr[48:112].view(np.uint64).astype(float)
nctu = 32
for lvl, (nm, calls) in enumerate((("8x8  (16 calls, 16 candidates)", 16), ("16x16 (4 calls, 12 candidates)", 4), ("32x32 (1 call, 12 candidates)", 1))):
    v = p2[16 + 4 * lvl:20 + 4 * lvl] / nctu / calls / 1e3
    print(f"{nm:32s} k cycles per call: groups {v[0]:7.1f}  TU headers {v[1]:6.1f}  CU headers {v[2]:6.1f}  slot lists {v[3]:6.1f}")
