#!/usr/bin/env python3
"""Known answers at BASELINE.json's full sizes, from the REAL reference (oracle/_ref; build container only):
  * configs[1]: the 3840x2160 G2-16 picture of bench.py (seed 0x1234ABCD), qpd6 = 2 -- stream length / SHA-256, reconstruction SHA-256;
  * configs[3]: tile-pictures of the 16384x16384 G2-16 picture as bench.py cuts it: 1024x1024 windows of the 1024-row bands
    gen_g2(1024, 16384, 16, seed 0x1234ABCD + 1000 + r): tiles (r, c) = (0, 0), (0, 5), (1, 3), (3, 15).
Writes tests/golden/manifest_fullsize.json (tests/test_gpu_parity.py and bench.py check against it)."""
import hashlib, json, os, sys, time
from multiprocessing import Pool
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import orc  # noqa: E402
import hevce_workload as wl  # noqa: E402

TILES = [(0, 0), (0, 5), (1, 3), (3, 15)]


def job(spec):
    kind = spec[0]
    if kind == "4k":
        img = wl.gen_g2(2160, 3840, 16, seed=0x1234ABCD); name = "bench_4k"
    else:
        _, r, c = spec
        band = wl.gen_g2(1024, 16384, 16, seed=0x1234ABCD + 1000 + r)
        img = band[:, c * 1024:(c + 1) * 1024].copy(); name = f"tile_r{r}_c{c}"
    t = time.time()
    s, rc = orc.encode_ref(img, 2)
    return dict(name=name, h=int(img.shape[0]), w=int(img.shape[1]), qpd6=2, len=len(s), sha256=hashlib.sha256(s).hexdigest(),
                rcon_sha256=hashlib.sha256(rc.tobytes()).hexdigest(), t_ref=round(time.time() - t, 1))


if __name__ == "__main__":
    assert orc.ref() is not None, "build oracle/_ref first (make -C oracle)"
    with Pool(5) as pool:
        res = pool.map(job, [("4k",)] + [("tile", r, c) for r, c in TILES])
    with open(os.path.join(ROOT, "tests", "golden", "manifest_fullsize.json"), "w") as f:
        json.dump(res, f, indent=1)
    for r in res:
        print(r)
