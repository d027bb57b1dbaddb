#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ['HEVCE_DUMP'] = '1'
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
img = wl.gen_g2(128, 384, 16, seed=7)
with hevce.Job([img.shape], 2) as job:
    job.upload(0, img); job.run()
    rec, st = job.records(0); rcon = job.rcon(0); dump = job.dump(rec.shape[0])
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "dump.npz"), rec=rec, st=st, rcon=rcon, dump=dump)
print("saved")
