#!/usr/bin/env python3
"""Pin the oracle and generate the committed golden fixtures (run in the build container only).

It needs /root/reference (inputs + the reference's own goldens) and oracle/_ref (the real reference
compiled by oracle/Makefile).  Nothing here runs on the GPU box; only its OUTPUTS (tests/golden/*) travel.

  1. reference build  vs  reference goldens  testimage_out/NN.h265 (qpd6=4)      -> must be identical
  2. oracle restatement vs reference build on 24 Kodak x qpd6 0..4 (stream + reconstruction)
  3. same on synthetic / odd-size / degenerate images
  4. writes tests/golden/manifest.json (sha256 + length of every stream and reconstruction),
     tests/golden/small_*.npz (full bytes for small cases) and tests/golden/func_vectors.npz
"""
import hashlib
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import orc  # noqa: E402

REFDIR = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def lcg_image(h, w):
    """SURVEY.md 8(d) footnote: the odd-size known-answer image"""
    s = 12345
    out = np.zeros((h, w), np.uint8)
    for y in range(h):
        for x in range(w):
            s = (s * 1103515245 + 12345) & 0x7FFFFFFF
            out[y, x] = min(255, max(0, ((2 * x + y) % 256) + ((s >> 16) & 15) - 8))
    return out


def small_cases():
    rng = np.random.RandomState(7)
    cases = {
        "g2_4_64x64": orc.gen_g2(64, 64, 4),
        "g2_16_64x96": orc.gen_g2(64, 96, 16),
        "g2_16_96x64_seed5": orc.gen_g2(96, 64, 16, seed=5),
        "lcg_100x70": lcg_image(70, 100),
        "flat_32x32": np.full((32, 32), 77, np.uint8),
        "black_40x40": np.zeros((40, 40), np.uint8),
        "white_33x65": np.full((33, 65), 255, np.uint8),
        "noise_64x64": rng.randint(0, 256, (64, 64)).astype(np.uint8),
        "noise_1x1": rng.randint(0, 256, (1, 1)).astype(np.uint8),
        "checker_64x64": ((np.indices((64, 64)).sum(0) & 1) * 255).astype(np.uint8),
        "hstripes_48x80": np.repeat(((np.arange(48) // 3 & 1) * 200 + 20).astype(np.uint8)[:, None], 80, 1),
        "gradient_64x128": np.tile(np.arange(128, dtype=np.uint8) * 2, (64, 1)),
    }
    return cases


def kodak_job(args):
    name, q = args
    img = orc.read_pgm(os.path.join(REFDIR, "testimage", name + ".pgm"))
    t = time.time(); rs, rr = orc.encode_ref(img, q); tr = time.time() - t
    t = time.time(); os_, or_ = orc.encode(img, q); to = time.time() - t
    ok = (rs == os_) and bool((rr == or_).all())
    gold_ok = None
    if q == 4:
        with open(os.path.join(REFDIR, "testimage_out", name + ".h265"), "rb") as f:
            gold_ok = f.read() == rs
    return dict(name=name, qpd6=q, h=int(img.shape[0]), w=int(img.shape[1]), len=len(rs), sha256=sha(rs),
                rcon_sha256=sha(rr.tobytes()), input_sha256=sha(img.tobytes()), oracle_matches_ref=ok,
                ref_matches_testimage_out=gold_ok, t_ref=round(tr, 2), t_oracle=round(to, 2))


def main():
    assert orc.ref() is not None, "build oracle/_ref first (make -C oracle)"
    os.makedirs(GOLD, exist_ok=True)
    manifest = {"kodak": [], "small": [], "synthetic": []}

    # --- small cases: full bytes committed -------------------------------------------------
    for name, img in small_cases().items():
        for q in range(5):
            rs, rr = orc.encode_ref(img, q)
            os_, or_ = orc.encode(img, q)
            ok = rs == os_ and bool((rr == or_).all())
            rec, rcon, st = orc.analyse(img, q)
            assert orc.write_stream(rec, rcon.shape[0], rcon.shape[1], q) == rs
            np.savez_compressed(os.path.join(GOLD, f"small_{name}_q{q}.npz"), img=img, stream=np.frombuffer(rs, np.uint8),
                                rcon=rr, records=rec, ctu_state=st)
            manifest["small"].append(dict(name=name, qpd6=q, h=int(img.shape[0]), w=int(img.shape[1]), len=len(rs),
                                          sha256=sha(rs), rcon_sha256=sha(rr.tobytes()), oracle_matches_ref=ok))
            print("small", name, q, len(rs), ok, flush=True)
            assert ok

    # --- SURVEY 8(c)/(d) known answers --------------------------------------------------------
    for noise, h, w in ((4, 256, 256), (16, 256, 256)):
        img = orc.gen_g2(h, w, noise)
        rs, rr = orc.encode_ref(img, 2)
        os_, or_ = orc.encode(img, 2)
        manifest["synthetic"].append(dict(name=f"g2_{noise}_{w}x{h}", qpd6=2, len=len(rs), sha256=sha(rs),
                                          rcon_sha256=sha(rr.tobytes()), oracle_matches_ref=(rs == os_ and bool((rr == or_).all()))))
        print("synthetic", noise, len(rs), sha(rs), flush=True)

    # --- Kodak x qpd6 -----------------------------------------------------------------------
    jobs = [(f"{i:02d}", q) for i in range(1, 25) for q in range(5)]
    with Pool(8) as pool:
        for r in pool.imap_unordered(kodak_job, jobs):
            manifest["kodak"].append(r)
            print("kodak", r["name"], r["qpd6"], r["len"], r["oracle_matches_ref"], r["ref_matches_testimage_out"], flush=True)
    manifest["kodak"].sort(key=lambda r: (r["name"], r["qpd6"]))
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    bad = [r for k in manifest for r in manifest[k] if not r["oracle_matches_ref"]]
    badg = [r for r in manifest["kodak"] if r["ref_matches_testimage_out"] is False]
    print("oracle mismatches:", len(bad), " reference-golden mismatches:", len(badg))
    return 1 if bad or badg else 0


if __name__ == "__main__":
    sys.exit(main())
