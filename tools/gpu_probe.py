#!/usr/bin/env python3
"""Quick timing probe on the GPU box: kernel ms and us/CTU for a few sizes and batch widths."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
out = []
for (h, w, nimg, q) in [(64, 64, 1, 2), (256, 256, 1, 2), (256, 256, 64, 2), (256, 256, 256, 2), (256, 256, 512, 2)] + \
        ([(2160, 3840, 1, 2)] if "--full" in sys.argv else []):
    imgs = [wl.gen_g2(h, w, 16, seed=100 + i) for i in range(min(nimg, 8))]
    with hevce.Job([(h, w)] * nimg, q) as job:
        for i in range(nimg):
            job.upload(i, imgs[i % len(imgs)])
        t = time.time(); job.run(); wall = time.time() - t
        tm = job.timings()
        H, W = job.padded_size(0)
        nctu = (H // 32) * (W // 32)
        rec = dict(h=h, w=w, nimg=nimg, q=q, wall_s=round(wall, 3), **{k: round(v, 3) for k, v in tm.items()},
                   us_per_ctu_chain=round(tm["kernel_ms"] * 1e3 / nctu, 1),
                   mpix_s_kernel=round(H * W * nimg / tm["kernel_ms"] / 1e3, 3))
        print(json.dumps(rec), flush=True)
        out.append(rec)
