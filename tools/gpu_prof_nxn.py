#!/usr/bin/env python3
"""Cycle split inside the NxN job's 4x4 pipeline (HEVCE_PROF_NXN build, libhevce_prof2.so)."""
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
stages = 32 * 64
print("per 4x4 stage (k cycles): border+sync+tu4 total %.1f, tokenise %.1f, engine %.1f, select %.1f" % tuple(p2[i] / stages / 1e3 for i in range(4)))
for nm, v in zip(("predict", "residual+fwd+quant", "levels/dequant", "inverse+recon"), p2[16:20]): print(f"  tu4 {nm:20s} {v/stages/1e3:6.2f} k")
