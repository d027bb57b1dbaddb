#!/usr/bin/env python3
"""Average duration of each TU-job type (HEVCE_PROF + HEVCE_PROF_JOBS build, libhevce_prof2.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HEVCE_LIB", os.path.join(ROOT, "hevc-image-encoder-lite_amd", "lib", "libhevce_prof2.so"))
os.environ["HEVCE_TRACE"] = "1"
import numpy as np
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
h, w = 128, 256
img = wl.gen_g2(h, w, 16)
with hevce.Job([img.shape], 2) as job:
    job.upload(0, img); job.run()
    tr = job.trace(0)
p2 = tr[48:112].view(np.uint64)
nctu = 32
for i, (nm, cnt) in enumerate((("chain4", 48), ("single8", 80), ("chain8", 20), ("single16", 36), ("chain16", 9), ("single32", 18))):
    print(f"{nm:9s} {p2[16+i]/nctu/cnt/1e3:8.1f} k cycles per job   ({p2[16+i]/nctu/1e3:8.0f} k per CTU)")
print("NxN job per CU8: %.1f k" % (p2[0:4].sum() / nctu / 16 / 1e3))
