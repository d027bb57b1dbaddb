#!/usr/bin/env python3
"""Latency of the drop-in symbol HEVCImageEncoder() on small pictures: the library keeps one job context per
calling thread, so only the first call pays for the device slab, the pinned buffers, the stream and the quantiser table.
Test infrastructure; prints one JSON line (kept under profiles/).  usage: gpu_call_latency.py [calls]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
out = {}
for (h, w) in ((64, 96), (256, 256), (512, 768)):
    img = wl.gen_g2(h, w, 16, seed=7)
    t = time.perf_counter(); first, _ = hevce.HEVCImageEncoder(img, 2); t_first = time.perf_counter() - t
    ts = []
    for _ in range(calls):
        t = time.perf_counter(); s, _ = hevce.HEVCImageEncoder(img, 2); ts.append(time.perf_counter() - t)
        assert s == first
    ts.sort()
    ctus = ((h + 31) // 32) * ((w + 31) // 32)
    out[f"{w}x{h}"] = {"ctus": ctus, "first_call_ms": round(t_first * 1e3, 2), "median_ms": round(ts[len(ts) // 2] * 1e3, 3), "p90_ms": round(ts[int(len(ts) * 0.9)] * 1e3, 3),
                       "median_us_per_ctu": round(ts[len(ts) // 2] * 1e6 / ctus, 1)}
print(json.dumps({"symbol": "HEVCImageEncoder (ctypes, qpd6 2, G2-16)", "calls": calls, "pictures": out,
                  "note": "first call of the first size includes loading the library's device code and building the job context; later sizes re-size the thread's context"}))
