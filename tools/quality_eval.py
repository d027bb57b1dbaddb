#!/usr/bin/env python3
"""usage: quality_eval.py <input-dir> <output-dir> [<qpd6>] [--no-compare]
The reference's evaluation run (reference HEVCeval.py: same arguments, default qpd6 = 3) on the MI355X library: every
readable picture of <input-dir> is padded, encoded (all of them in ONE GPU batch), written as <output-dir>/<name>.h265,
and reported with SSIM / bits per pixel / PSNR; unless --no-compare, JPEG, JPEG 2000 and WebP are tuned in memory to the
nearest SSIM and their size is reported relative to the .h265 (hevc-image-encoder-lite_amd/quality.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    if len(args) < 2 or args[0] == args[1]:
        print(__doc__); return 255
    in_dir, out_dir = args[:2]
    qpd6 = int(args[2]) if len(args) > 2 and args[2] in "01234" and len(args[2]) == 1 else 3
    import hevce_pkg
    hevce = hevce_pkg.load()
    import importlib
    quality = importlib.import_module(hevce.__name__ + ".quality")
    print(f"input dir = {in_dir}\noutput dir = {out_dir}\nQp%6 = {qpd6} (Qp = {qpd6 * 6 + 4})\n")
    os.makedirs(out_dir, exist_ok=True)
    named = []
    for f in sorted(os.listdir(in_dir)):
        try:
            named.append((f, quality.read_grey(os.path.join(in_dir, f))))
        except Exception:
            continue
    if not named:
        print("no readable pictures"); return 255
    rows = quality.evaluate(named, qpd6, hevce.encode_batch, compare="--no-compare" not in argv)
    means = {}
    for r in rows:
        print(f"{r['name']}    width={r['width']}    height={r['height']}")
        print(f"  HEVC     : ssim={r['ssim']:.5f}    bpp={r['bpp']:.3f}    psnr={r['psnr']:.4f} dB")
        with open(os.path.join(out_dir, os.path.splitext(r["name"])[0] + ".h265"), "wb") as f:
            f.write(r["stream"])
        means.setdefault("HEVC", []).append(r["bpp"])
        for o in r["others"]:
            if "error" in o:
                print(f"  {o['codec']:<8} : unavailable ({o['error']})"); continue
            print(f"  {o['codec']:<8} : ssim={o['ssim']:.5f}    bpp={o['bpp']:.3f}    qparam={o['quality']}    size/HEVCsize={o['size_ratio']:f}")
            means.setdefault(o["codec"], []).append(o["bpp"])
    print("bpp mean : " + "   ".join(f"{k}:{np.mean(v):.5f}" for k, v in means.items()))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
