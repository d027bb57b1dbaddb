#!/usr/bin/env python3
"""In-kernel cycle probes (probe builds of the library: `make -C hevc-image-encoder-lite_amd/csrc prof prof2 wait bin jdur-N tu-N`).

  gpu_prof.py phases [h w]   per-phase cycles per CU size; clusters of four: the 8x8 workgroup's wavefronts and the chain workgroup's
                             section between two jobs phases                        (libhevce_prof.so,  -DHEVCE_PROF)
                             HEVCE_LIB=.../libhevce_light.so (make light): only the hand-off stamps, near-product timing
  gpu_prof.py nxn            inside the NxN chain's 4x4 stages                     (libhevce_prof2.so, + -DHEVCE_PROF_NXN)
  gpu_prof.py wait           wavefront 0's wait at the barrier closing each round  (libhevce_wait.so,  + -DHEVCE_PROF_WAIT)
  gpu_prof.py bin            binarisation passes per CU level                      (libhevce_bin.so,   + -DHEVCE_PROF_BIN)
  gpu_prof.py jdur N         length of wavefront N's jobs phase per round          (libhevce_jdurN.so, + -DHEVCE_PROF_JDUR=N)
  gpu_prof.py tu N           phases of the N x N block job                         (libhevce_tuN.so,   + -DHEVCE_PROF_TU=N)
Workload: 256x128 G2-16 (32 CTUs), qpd6 = 2, unless h w are given.  Set HEVCE_FLAGS=1 to probe the EXACT engines."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kind = sys.argv[1] if len(sys.argv) > 1 else "phases"
arg = sys.argv[2] if len(sys.argv) > 2 and kind in ("jdur", "tu") else ""
libname = {"phases": "prof", "nxn": "prof2", "wait": "wait", "bin": "bin", "jdur": "jdur" + arg, "tu": "tu" + arg}[kind]
os.environ.setdefault("HEVCE_LIB", os.path.join(ROOT, "hevc-image-encoder-lite_amd", "lib", f"libhevce_{libname}.so"))
os.environ["HEVCE_TRACE"] = "1"
import numpy as np
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
h, w = (int(sys.argv[2]), int(sys.argv[3])) if kind == "phases" and len(sys.argv) > 3 else (128, 256)
img = wl.gen_g2(h, w, 16)
with hevce.Job([img.shape], 2) as job:
    job.upload(0, img); job.run()
    tr = job.trace(0); tm = job.timings(); st = job.stats(0)
nctu = (hevce.pad32(h) // 32) * (hevce.pad32(w) // 32)
prof = tr[:48].view(np.uint64).reshape(3, 8)
p2 = tr[48:112].view(np.uint64).astype(float)
whole = int(tr[112:114].view(np.uint64)[0])
print(f"{w}x{h}: kernel {tm['kernel_ms']:.1f} ms, {tm['kernel_ms']*1e3/nctu:.0f} us/CTU, WHOLE KERNEL {whole/nctu/1e3:.0f} k cycles/CTU "
      f"(clock {whole/tm['kernel_ms']/1e6:.2f} GHz), CTUs analysed twice {st[0]}, coder mismatches {st[1]}")
if kind == "phases":
    tot = prof.sum()
    print(f"  stamped {tot/nctu/1e3:.0f} k/CTU; per 8x8 round: jobs phase {prof[0,0]/nctu/16/1e3:.1f} k, engine phase {prof[0,6]/nctu/16/1e3:.1f} k, "
          f"decide {(prof[0,2]+prof[0,3]+prof[0,4])/nctu/16/1e3:.1f} k")
    names = ["jobs", "-", "split+finish", "arg-min", "commit", "ctx init", "engines", "-"]
    for lvl, nm in enumerate(("8x8", "16x16", "32x32")):
        print(f"  CU {nm:6s}", "  ".join(f"{names[p]} {100.0*prof[lvl,p]/tot:5.1f}%" for p in range(7) if names[p] != "-"))
    wk = tr[160:176].view(np.uint64).astype(float)
    if wk[5] > 0:
        print("  8x8 worker (cluster >= 4), k cycles per message: wait %.1f, accept %.1f, jobs %.1f, contexts+binarise %.1f, engines+answer %.1f" % tuple(wk[i] / wk[5] / 1e3 for i in range(5)))
        print("    the slots' engines (four candidates each, behind the slot's binarisation): %.1f k cycles per CU in all, %.0f tokens (first candidate of each slot) per CU   (HEVCE_FLAGS bit 5: the second 8x8 workgroup's numbers)" % (wk[7] / wk[5] / 1e3, wk[6] / wk[5]))
        pw = tr[160:224].view(np.uint64).astype(float)[8:32].reshape(8, 3) / wk[5] / 1e3
        for wv in range(8):
            print("    wavefront %d: single-block trial items %.1f, TU-split slots (trials + tokens) %.1f, single-block token slots %.1f   (k cycles per CU; work items are handed out by a counter)" % (wv, pw[wv, 0], pw[wv, 1], pw[wv, 2]))
    if wk[5] > 0:
        hs = p2[5:12] / nctu / 16 / 1e3
        print("  chain workgroup between two 8x8 jobs phases, k cycles per round: NxN report %.1f, wait for the answer %.1f, decision + commit %.1f, samples to HBM %.1f, "
              "enclosing CUs' decisions %.1f, entering the next CU (snapshot, level begin, CTU start / end work) %.1f, quad message or barrier %.1f" % (hs[1], hs[0], hs[3], hs[4], hs[2], hs[5], hs[6]))
    if p2[15] > 0: print("  NxN pass (four modes of a 4x4 block, one helper wavefront), k cycles: block trials %.2f, binarisation %.2f" % (p2[13] / p2[15] / 1e3, p2[14] / p2[15] / 1e3))
    wrp = tr[114:120].view(np.uint64).astype(float)
    if p2[16] > 0: print("  NxN stage, the chain wavefront (k cycles per 4x4 block): reference samples %.2f, its own pass %.2f, waiting for the helpers %.2f" % tuple(p2[16:19] / nctu / 64 / 1e3))
    if p2[24] > 0: print("    units of a 4x4 block, start to end (k cycles): chain wavefront %.2f, helpers" % (p2[24] / nctu / 64 / 1e3), " ".join("%.2f" % (v / nctu / 64 / 1e3) for v in p2[25:31]))
    if p2[12] > 0: print("  candidate 70's rider, k cycles per 4x4 block: the first three %.2f, the last one (the chain waits for it) %.2f" % (p2[31] / nctu / 48 / 1e3, p2[12] / nctu / 16 / 1e3))
    wr2 = tr[120:128].view(np.uint64).astype(float)
    if wrp[2] > 0 and wr2.sum() > 0: print("    of the coding: binarisation %.1f, tokens into LDS %.1f, resolving the contexts %.1f, the coder %.1f" % tuple(wr2 / wrp[2] / 1e3))
    if wrp[2] > 0: print("  writer workgroup (clusters of five and more), k cycles per CTU: waiting for the chain %.1f, coding %.1f" % (wrp[0] / wrp[2] / 1e3, wrp[1] / wrp[2] / 1e3))
    print("  NxN chain:", "  ".join(f"{nm} {100.0*p2[i]/tot:5.1f}%" for i, nm in enumerate(("4x4 pipelines", "tokenise", "engine", "select", "tail"))))
elif kind == "nxn":
    stages = nctu * 64
    print("per 4x4 stage (k cycles): border+sync+tu4 %.1f, tokenise %.1f, engine %.1f, select %.1f; tail per round %.1f" %
          (tuple(p2[i] / stages / 1e3 for i in range(4)) + (p2[4] / nctu / 16 / 1e3,)))
    for nm, v in zip(("predict", "residual+fwd+quant", "levels/dequant", "inverse+recon"), p2[16:20]): print(f"  tu4 {nm:20s} {v/stages/1e3:6.2f} k")
elif kind == "wait":
    print("NxN job per round (k cycles): stages %.1f, tail %.1f" % (p2[0:4].sum() / nctu / 16 / 1e3, p2[4] / nctu / 16 / 1e3))
    print("wait of wavefront 0 at the jobs-phase barrier, k cycles per round:")
    for k in range(4): print("  k16=%d: " % k + "  ".join("m=%d %6.1f" % (m, p2[16 + 4 * k + m] / nctu / 1e3) for m in range(4)))
    print("  total per CTU: %.0f k" % (p2[16:32].sum() / nctu / 1e3))
elif kind == "jdur":
    print(f"jobs phase of wavefront {arg}, k cycles per round:")
    for k in range(4): print("  k16=%d: " % k + "  ".join("m=%d %6.1f" % (m, p2[16 + 4 * k + m] / nctu / 1e3) for m in range(4)))
    print("  total per CTU: %.0f k" % (p2[16:32].sum() / nctu / 1e3))
elif kind == "bin":
    for lvl, (nm, calls) in enumerate((("8x8  (16 calls, 16 candidates)", 16), ("16x16 (4 calls, 12 candidates)", 4), ("32x32 (1 call, 12 candidates)", 1))):
        v = p2[16 + 4 * lvl:20 + 4 * lvl] / nctu / calls / 1e3
        print(f"{nm:32s} k cycles per call: groups {v[0]:7.1f}  TU headers {v[1]:6.1f}  CU headers {v[2]:6.1f}  slot lists {v[3]:6.1f}")
    wk = tr[160:224].view(np.uint64).astype(float)
    if wk[5] > 0:
        v = wk[16:20] / wk[5] / 1e3
        print(f"8x8 worker, wavefront 1 (16 TU-split candidates), k cycles per message: groups {v[0]:.1f}  TU headers {v[1]:.1f}  CU headers {v[2]:.1f}  slot lists {v[3]:.1f}")
elif kind == "tu":
    N = int(arg)
    v6 = p2[16:22]
    calls = {32: 18, 16: 36 + 9 * 4, 8: 80 + 20 * 4, 4: 16 * 3 * 4}[N] * nctu     # 4: the TU-split chains of the 8x8 CUs, three wavefronts x four blocks
    print(f"hv_tu_job<{N}>: {v6.sum()/calls/1e3:.1f} k cycles per call")
    for nm, v in zip(("refs", "predict", "fwd cols", "fwd rows+quant", "inv cols", "inv rows+recon"), v6): print(f"  {nm:16s} {v/calls/1e3:7.1f} k  {100*v/v6.sum():5.1f}%")
