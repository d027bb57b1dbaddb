#!/usr/bin/env python3
"""A picture produced asynchronously on torch's current stream (handle 0 = the NULL stream) must be what the kernel reads when the
job runs on that stream handle (hevce_job_use_stream) -- no host synchronisation between producer and encoder."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
torch.cuda.set_device(0)
import hevce_pkg, hevce_workload as wl, orc
hevce = hevce_pkg.load()
dev = torch.device("cuda", 0)
img = wl.gen_g2(96, 160, 16, seed=99)
want, _ = orc.encode(img, 2)
src = torch.from_numpy(img.astype(np.int16) + 7).to(dev)
big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
with hevce.Job([img.shape], 2) as job:
    job.use_stream(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(3):
        d = torch.zeros(img.shape, dtype=torch.uint8, device=dev)
        job.set_device_image(0, d.data_ptr(), keepalive=d)
        big.normal_()                                           # a long-running producer first ...
        d.copy_((src - 7).to(torch.uint8))                      # ... then the picture, both only enqueued
        job.run()
        assert job.stream(0) == want, "the kernel read the picture before it was written"
print("stream order ok")
