#!/usr/bin/env python3
"""Run one library variant (HEVCE_LIB) on a few images and compare with the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hevce_pkg, hevce_workload as wl, orc
hevce = hevce_pkg.load()
print("lib", os.path.basename(hevce.LIB_PATH), flush=True)
for (h, w, q, seed) in [(256, 256, 2, 0x1234ABCD), (128, 384, 2, 7), (256, 256, 0, 11)]:
    img = wl.gen_g2(h, w, 16, seed=seed)
    with hevce.Job([img.shape], q) as job:
        job.upload(0, img); job.run()
        rec, st = job.records(0); tm = job.timings()
    orec, orcon, ost = orc.analyse(img, q)
    bad = np.argwhere((st != ost).any(axis=1)).ravel()
    print(h, w, q, "kernel_ms", round(tm["kernel_ms"], 1), "states equal:", len(bad) == 0, "first bad CTU:", (int(bad[0]) if len(bad) else None), flush=True)
