#!/usr/bin/env python3
"""Quick GPU check: one small picture through the default path, elapsed time, cluster size / fall-backs, parity against the oracle.
usage: gpu_quick.py [h w [qpd6]]   (HEVCE_CLUSTER / HEVCE_FLAGS from the environment)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hevce_pkg, hevce_workload as wl, orc
hevce = hevce_pkg.load()
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 64)
q = int(sys.argv[3]) if len(sys.argv) > 3 else 2
img = wl.gen_g2(h, w, 16)
t = time.time()
with hevce.Job([img.shape], q) as job:
    job.upload(0, img); job.run()
    s, r = job.stream(0), job.rcon(0)
    tm = job.timings(); st = job.stats(0)
    try: cl = hevce.lib().hevce_job_cluster(job._h, None)
    except Exception: cl = -1
dt = time.time() - t
os_, or_ = orc.encode(img, q)
print(f"{w}x{h} q{q}: wall {dt:.2f} s, kernel {tm['kernel_ms']:.1f} ms, cluster {cl}, stats {list(st)}, stream {'OK' if s == os_ else 'DIFFERS'}, rcon {'OK' if np.array_equal(r, or_) else 'DIFFERS'}", flush=True)
