#!/usr/bin/env python3
"""Staged first-contact probe: every stage is logged with a timestamp and flushed, so a hang can be located."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t0 = time.time()
def log(*a):
    print(f"[{time.time()-t0:7.2f}s]", *a, flush=True)
import numpy as np
log("numpy ok")
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
log("lib loaded", hevce.lib().hevce_version(), "devices", hevce.device_count())
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import orc
for (h, w) in [(32, 32), (64, 96), (256, 256)]:
    img = wl.gen_g2(h, w, 16)
    job = hevce.Job([img.shape], 2)
    log("job created", h, w)
    job.upload(0, img)
    log("uploaded")
    job.launch()
    log("launched")
    job.finish()
    log("finished", job.timings())
    s, r = job.stream(0), job.rcon(0)
    rec, st = job.records(0)
    os_, or_ = orc.encode(img, 2)
    orec, _, ost = orc.analyse(img, 2)
    log("stream equal:", s == os_, "rcon equal:", bool((r == or_).all()), "records equal:", bool((rec == orec).all()),
        "state equal:", bool((st == ost).all()), len(s), len(os_))
    if not (st == ost).all():
        k = int(np.argwhere((st != ost).any(axis=1))[0][0])
        log("first bad CTU", k, st[k].tolist(), ost[k].tolist())
    job.close()
log("done")
