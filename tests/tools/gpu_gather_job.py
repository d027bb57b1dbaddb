#!/usr/bin/env python3
"""TEST TOOL (tests/test_gpu_parity.py).  hevce_job_device_payload: the slice data where the kernel wrote them; distributed.gather_job_streams
builds the streams from them (device-to-device into the collective's send buffer; one rank here, so no collective) -- they must equal
the streams the host assembled and the oracle's.  In a process of its own: torch brings up its HIP runtime first (as in bench.py)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
torch.cuda.set_device(0)
import hevce_pkg, hevce_workload as wl, orc
hevce = hevce_pkg.load()
D = importlib.import_module(hevce.__name__ + ".distributed")
imgs = [wl.gen_g2(64, 96, 16, seed=5), wl.gen_g2(40, 72, 16, seed=6)]
with hevce.Job([im.shape for im in imgs], 2) as job:
    for i, im in enumerate(imgs):
        job.upload(i, im)
    assert job.device_payload(0) == (None, 0)                    # nothing finished yet
    job.run()
    for i in range(2):
        ptr, n = job.device_payload(i)
        s = job.stream(i)
        hdr = len(hevce.write_header(2, *job.padded_size(i)))
        assert ptr and n == len(s) - hdr
    got = D.gather_job_streams(job, device=torch.device("cuda:0"))
    assert got == [[job.stream(0), job.stream(1)]]
    assert got[0][0] == orc.encode(imgs[0], 2)[0] and got[0][1] == orc.encode(imgs[1], 2)[0]
print("gather from the device ok")
