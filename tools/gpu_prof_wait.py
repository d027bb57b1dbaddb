#!/usr/bin/env python3
"""Where wavefront 0 waits (HEVCE_PROF + HEVCE_PROF_WAIT build, libhevce_prof2.so): per round (k16, m) the time between the
end of its NxN chain and the release of the barrier that closes the jobs phase, and the tail of the NxN job after its last
4x4 stage (candidate 70's header and context column)."""
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
print("NxN job per round (k cycles): stages %.1f, tail %.1f" % (p2[0:4].sum() / nctu / 16 / 1e3, p2[4] / nctu / 16 / 1e3))
print("wait of wavefront 0 at the jobs-phase barrier, k cycles per round:")
for k in range(4):
    print("  k16=%d: " % k + "  ".join("m=%d %6.1f" % (m, p2[16 + 4 * k + m] / nctu / 1e3) for m in range(4)))
print("  total per CTU: %.0f k" % (p2[16:32].sum() / nctu / 1e3))
