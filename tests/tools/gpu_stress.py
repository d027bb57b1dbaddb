#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: many small pictures (random sizes, content classes, qpd6) in batches through the
C ABI, every stream and reconstruction compared with the oracle (CPU restatement; test infrastructure).  The kernel's
wavefronts hand work to each other through LDS flags and idle-time riding, i.e. its timing varies from launch to launch:
this sweep is there to catch what a fixed set of goldens cannot.  usage: gpu_stress.py [n_pictures] [seed] [repeats] [batch]
batch: pictures per launch (default: all in one).  Launches of at most 32 pictures run as clusters of seven workgroups per picture (fewer workgroups up to 128 pictures)
(hevce_device.h cluster mode), larger ones with one workgroup per picture; the consistency counter of every picture
(hevce_job_stats: a state mismatch that the re-analysis healed) must stay 0 too."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hevce_pkg, hevce_workload as wl, orc
hevce = hevce_pkg.load()


def sweep(n, seed, repeats, batch=0):
    rng = np.random.default_rng(seed)
    imgs, qs = [], []
    for i in range(n):
        h, w = int(rng.integers(1, 161)), int(rng.integers(1, 225))
        kind = i % 6
        if kind == 0: img = wl.gen_g2(h, w, int(rng.choice([1, 4, 16, 48])), seed=int(rng.integers(1, 2**31)))
        elif kind == 1: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 2: img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        elif kind == 3: img = ((np.add.outer(np.arange(h) * int(rng.integers(0, 5)), np.arange(w) * int(rng.integers(0, 5))) + rng.integers(0, 6, (h, w))) & 255).astype(np.uint8)
        elif kind == 4: img = (rng.integers(0, 2, (h, w), dtype=np.uint8) * 255)
        else: img = np.clip(np.kron(rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8)), np.ones((8, 8)))[:h, :w] + rng.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)
        imgs.append(np.ascontiguousarray(img)); qs.append(int(rng.integers(0, 5)))
    t = time.perf_counter()
    want = [orc.encode(im, q) for im, q in zip(imgs, qs)]
    t_cpu = time.perf_counter() - t
    bad = []
    t = time.perf_counter()
    batch = batch or n
    healed, twice, clusters = 0, 0, set()
    for r in range(repeats):
        for b0 in range(0, n, batch):
            sel = range(b0, min(n, b0 + batch))
            with hevce.Job([imgs[i].shape for i in sel], [qs[i] for i in sel]) as job:
                clusters.add(job.cluster()[0])
                for k, i in enumerate(sel): job.upload(k, imgs[i])
                job.run()
                for k, i in enumerate(sel):
                    st = job.stats(k)
                    twice += int(st[0]); healed += int(st[1])
                    if job.stream(k) != want[i][0] or not np.array_equal(job.rcon(k), want[i][1]): bad.append((r, i, imgs[i].shape, qs[i]))
                if job.cluster()[1]: bad.append((r, b0, "cluster fallback", job.cluster()[1]))
    t_gpu = time.perf_counter() - t
    return {"pictures": n, "seed": seed, "repeats": repeats, "batch": batch, "workgroups_per_picture": sorted(clusters), "mismatches": len(bad), "first": bad[:5],
            "ctus_analysed_twice": twice, "healed_state_mismatches": healed,
            "oracle_s": round(t_cpu, 1), "gpu_s": round(t_gpu, 2),
            "ctus": int(sum(((im.shape[0] + 31) // 32) * ((im.shape[1] + 31) // 32) for im in imgs))}


if __name__ == "__main__":
    res = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 96, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                int(sys.argv[3]) if len(sys.argv) > 3 else 3, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    print(json.dumps(res))
    sys.exit(1 if res["mismatches"] or res["healed_state_mismatches"] else 0)
