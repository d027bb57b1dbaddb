#!/usr/bin/env python3
"""Golden SHA-256s at BASELINE.json configs[2] picture size (run in the build container only): eight 1920x1080 G2-16 pictures
(seed 0x1234ABCD + i, the pictures bench.py / tools/dist_batch.py use) through the REAL reference (oracle/_ref), qpd6 = 2,
one process per picture.  Writes tests/golden/manifest_1080p.json; tests/test_gpu_parity.py encodes the same pictures in one
launch on the GPU and compares."""
import hashlib, json, os, sys, time
from multiprocessing import Pool
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import orc  # noqa: E402
import hevce_workload as wl  # noqa: E402


def job(i):
    img = wl.gen_g2(1080, 1920, 16, seed=0x1234ABCD + i)
    t = time.time()
    s, r = orc.encode_ref(img, 2)
    return dict(index=i, seed=0x1234ABCD + i, h=1080, w=1920, qpd6=2, len=len(s), sha256=hashlib.sha256(s).hexdigest(),
                rcon_sha256=hashlib.sha256(r.tobytes()).hexdigest(), t_ref=round(time.time() - t, 1))


if __name__ == "__main__":
    assert orc.ref() is not None, "build oracle/_ref first (make -C oracle)"
    with Pool(8) as pool:
        res = sorted(pool.map(job, range(8)), key=lambda r: r["index"])
    with open(os.path.join(ROOT, "tests", "golden", "manifest_1080p.json"), "w") as f:
        json.dump(res, f, indent=1)
    for r in res:
        print(r)
