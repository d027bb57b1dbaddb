#!/usr/bin/env python3
"""TEST TOOL (tests/test_gpu_parity.py).  A batch that fills the device -- 768 small pictures, one workgroup each, three times the compute
units -- is launched and, while it runs, one more picture on a job of its own, which would like a cluster of workgroups.  Prints one
JSON line: both results checked against the oracle, the cluster size the single picture ran with, its fall-backs, and how long its
run() took beyond the batch's own duration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hevce_pkg, hevce_workload as wl, orc
hevce = hevce_pkg.load()

NB = 768
base = [wl.gen_g2(64, 64, 16, seed=100 + k) for k in range(8)]
want_b = [orc.encode(im, 2)[0] for im in base]
single = wl.gen_g2(96, 128, 16, seed=999)
want_s, want_r = orc.encode(single, 2)

with hevce.Job([(64, 64)] * NB, 2) as batch, hevce.Job([single.shape], 2) as one:
    for i in range(NB):
        batch.upload(i, base[i % 8])
    one.upload(0, single)
    batch.run()                                                  # warm-up + the batch's own duration
    t0 = time.time(); batch.run(); t_batch = time.time() - t0
    one.run()                                                    # alone: the cluster it would like
    alone = one.cluster()
    t0 = time.time()
    batch.launch()
    one.run()                                                    # beside the batch
    t_one = time.time() - t0
    batch.finish()
    t_all = time.time() - t0
    res = dict(batch_ok=all(batch.stream(i) == want_b[i % 8] for i in range(NB)),
               single_ok=one.stream(0) == want_s and bool(np.array_equal(one.rcon(0), want_r)),
               alone_cluster=alone[0], single_cluster=one.cluster()[0], single_fallbacks=one.cluster()[1] - alone[1],
               batch_s=round(t_batch, 3), single_s=round(t_one, 3), both_s=round(t_all, 3), single_extra_s=round(max(t_one - t_batch, 0.0), 3))
print(json.dumps(res), flush=True)
