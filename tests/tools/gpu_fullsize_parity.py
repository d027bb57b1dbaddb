#!/usr/bin/env python3
"""Full-size parity of the bench workload: the 3840x2160 G2-16 picture (qpd6 = 2) encoded by the MI355X library and by the
REAL reference (oracle/_ref, compiled from the reference sources, one CPU core, ~2 minutes): streams and reconstructions
must be byte-identical.  Test infrastructure; prints one JSON line (kept under profiles/)."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hevce_pkg, hevce_workload as wl, orc
hevce = hevce_pkg.load()
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
q = 2
img = wl.gen_g2(h, w, 16, seed=0x1234ABCD)
t = time.perf_counter(); stream, rcon = hevce.HEVCImageEncoder(img, q); t_gpu = time.perf_counter() - t
kind = "reference" if orc.ref() is not None else "oracle"
t = time.perf_counter(); rstream, rrcon = (orc.encode_ref if kind == "reference" else orc.encode)(img, q); t_cpu = time.perf_counter() - t
print(json.dumps({"image": f"{w}x{h} G2-16 seed 0x1234ABCD", "qpd6": q, "checker": kind,
                  "stream_bytes": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest(),
                  "stream_identical": stream == rstream, "reconstruction_identical": bool(np.array_equal(rcon, rrcon)),
                  "gpu_call_s": round(t_gpu, 2), "cpu_reference_s": round(t_cpu, 2), "speedup_vs_1_core": round(t_cpu / t_gpu, 2)}))
