#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Differential check of the product's FAST decision rule on the CPU (oracle/hevce_oracle.c cost_bits).

The HIP kernel costs trial encodes WITHOUT their emulation-prevention bytes (its FAST engines carry the range half of the coder) and
re-analyses exactly every CTU the stream writer or the NxN verification flags.  orc_analyse_fast_rule restates that rule on the CPU;
this sweep compares it with the oracle proper on the picture classes where such bytes are frequent -- constant, flat with faint
noise, soft blobs, shallow ramps -- at random sizes and qpd6.  Any difference in records / reconstruction / coder registers is a
counter-example to DESIGN.md 4.1.

  fast_rule_sweep.py [--ctus N] [--procs P] [--seed S] [--out file.json]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np


def picture(rng):
    """one picture of a class in which runs of zero bytes (hence emulation-prevention bytes) are likely"""
    h = 32 * int(rng.randint(1, 5)) - int(rng.randint(0, 2)) * int(rng.randint(0, 9))
    w = 32 * int(rng.randint(1, 5)) - int(rng.randint(0, 2)) * int(rng.randint(0, 9))
    kind = int(rng.randint(0, 5))
    base = int(rng.randint(0, 256))
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0: img = np.full((h, w), base, np.int32)                                   # constant
    elif kind == 1: img = base + rng.randint(-1, 2, (h, w)) * (rng.rand(h, w) < rng.choice([0.02, 0.1, 0.3]))   # faint noise
    elif kind == 2:                                                                        # soft blobs
        img = np.full((h, w), float(base))
        for _ in range(int(rng.randint(1, 4))):
            cy, cx, r, a = rng.rand() * h, rng.rand() * w, 6 + rng.rand() * 40, rng.randint(-24, 25)
            img = img + a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r))
    elif kind == 3: img = base + (yy * int(rng.randint(-2, 3)) + xx * int(rng.randint(-2, 3))) // int(rng.randint(4, 33))   # shallow ramp
    else:                                                                                  # flat halves with one edge
        img = np.where(xx * int(rng.randint(1, 4)) + yy * int(rng.randint(0, 4)) < int(rng.randint(16, 200)), base, base + int(rng.randint(-40, 41)))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), kind, int(rng.randint(0, 5))


def work(args):
    seed, nctu_target = args
    import orc
    rng = np.random.RandomState(seed)
    tot = dict(pictures=0, ctus=0, redone=0, events=0, mismatches=0, by_kind=[0] * 5, first_bad=None)
    while tot["ctus"] < nctu_target:
        img, kind, q = picture(rng)
        r0, c0, s0 = orc.analyse(img, q)
        r1, c1, s1, st = orc.analyse_fast_rule(img, q)
        ok = np.array_equal(r0, r1) and np.array_equal(c0, c1) and np.array_equal(s0, s1)
        tot["pictures"] += 1; tot["ctus"] += st[0]; tot["redone"] += st[1]; tot["events"] += st[2]; tot["by_kind"][kind] += st[0]
        if not ok:
            tot["mismatches"] += 1
            if tot["first_bad"] is None: tot["first_bad"] = dict(seed=seed, picture=tot["pictures"], shape=list(img.shape), q=q, kind=kind)
    return tot


def sweep(ctus, procs, seed):
    import multiprocessing as mp
    jobs = [(seed + 7919 * i, (ctus + procs * 4 - 1) // (procs * 4)) for i in range(procs * 4)]
    t = time.time()
    if procs > 1:
        with mp.get_context("spawn").Pool(procs) as pool: parts = pool.map(work, jobs)
    else: parts = [work(j) for j in jobs]
    out = dict(pictures=0, ctus=0, redone=0, events=0, mismatches=0, by_kind=[0] * 5, first_bad=None)
    for p in parts:
        for k in ("pictures", "ctus", "redone", "events", "mismatches"): out[k] += p[k]
        out["by_kind"] = [a + b for a, b in zip(out["by_kind"], p["by_kind"])]
        out["first_bad"] = out["first_bad"] or p["first_bad"]
    out.update(seconds=round(time.time() - t, 1), seed=seed, procs=procs,
               note="events = trial encodes (any CU level / candidate) with an emulation-prevention byte inside FAST passes; redone = CTUs decided again exactly")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctus", type=int, default=2000); ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--seed", type=int, default=20261005); ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = sweep(a.ctus, a.procs, a.seed)
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f: json.dump(res, f, indent=1)
    sys.exit(1 if res["mismatches"] else 0)
