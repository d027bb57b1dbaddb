#!/usr/bin/env python3
"""Length of the jobs phase per round (k16, m) as one wavefront sees it, riders included (HEVCE_PROF + HEVCE_PROF_JDUR=<wave>
build, libhevce_prof2.so); wavefront 0 = the NxN chain."""
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
nctu = 32
for k in range(4):
    print("  k16=%d: " % k + "  ".join("m=%d %6.1f" % (m, p2[16 + 4 * k + m] / nctu / 1e3) for m in range(4)))
