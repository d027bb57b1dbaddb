#!/usr/bin/env python3
"""Batch throughput on one GPU: N independent pictures (one workgroup each) in ONE launch -- BASELINE.json configs[2]'s
shape on a single GPU (default 256 x 1920x1080, qpd6=2), checked by a sample of streams against the oracle."""
import os, sys, time, json, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
uniq = [wl.gen_g2(h, w, 16, seed=0x1234ABCD + i) for i in range(min(n, 8))]
with hevce.Job([(h, w)] * n, 2) as job:
    for i in range(n):
        job.upload(i, uniq[i % len(uniq)])
    t = time.time(); job.run(); wall = time.time() - t
    tm = job.timings()
    H, W = job.padded_size(0)
    shas = [hashlib.sha256(job.stream(i)).hexdigest() for i in range(n)]
ok = all(shas[i] == shas[i % len(uniq)] for i in range(n))
line = dict(images=n, size=f"{w}x{h}", padded=f"{W}x{H}", ctus_per_image=(H // 32) * (W // 32), kernel_s=round(tm["kernel_ms"] / 1e3, 2),
            writer_s=round(tm["writer_ms"] / 1e3, 2), wall_s=round(wall, 2), mpix_per_s_kernel=round(H * W * n / tm["kernel_ms"] / 1e3, 2),
            mpix_per_s_wall=round(H * W * n / wall / 1e6, 2), replicas_consistent=ok)
if "--check" in sys.argv:
    import orc
    s, _ = orc.encode(uniq[0], 2)
    line["stream0_matches_oracle"] = hashlib.sha256(s).hexdigest() == shas[0]
print(json.dumps(line), flush=True)
