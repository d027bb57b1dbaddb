#!/usr/bin/env python3
"""Candidate 70 inside the 8x8 engine phases (HEVCE_PROF + HEVCE_PROF_ENG build, libhevce_prof2.so): its engine loop, its byte
replay, its length in chunks; the phase as wavefront 2 (its host) sees it, and that wavefront's wait at the closing barrier."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HEVCE_LIB", os.path.join(ROOT, "hevc-image-encoder-lite_amd", "lib", "libhevce_prof2.so"))
os.environ["HEVCE_TRACE"] = "1"
import numpy as np
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
img = wl.gen_g2(128, 256, 16)
with hevce.Job([img.shape], 2) as job:
    job.upload(0, img); job.run()
    tr = job.trace(0)
p2 = tr[48:112].view(np.uint64).astype(float)
n = p2[21]
print("per 8x8 engine phase (k cycles): wavefront 2 busy %.1f, waits %.1f at the barrier; candidate 70: engine loop %.1f, replay %.1f, %.1f chunks"
      % (p2[19] / n / 1e3, p2[20] / n / 1e3, p2[16] / n / 1e3, p2[17] / n / 1e3, p2[18] / n))
