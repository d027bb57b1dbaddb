"""Quality evaluation on top of the MI355X encoder.

What the reference's evaluation script reports for a directory of pictures (reference HEVCeval.py:159-235: pad to multiples
of 32 by edge replication, encode at one qpd6, SSIM with data_range 256 and bits per pixel of the .h265, then for JPEG /
JPEG 2000 / WebP the quality setting whose SSIM is nearest, found by bisection, and its size relative to the .h265), and
the PSNR of its command line (reference src/HEVCeMain.c:116-133) -- written from the published definitions, not from the
script: all pictures go to the GPU as ONE batch (one workgroup each), the other codecs run in memory (Pillow, optional),
and nothing sleeps.  SSIM follows Wang et al. 2004 with the defaults of the library the reference calls (7x7 uniform
window, K1 = 0.01, K2 = 0.03, unbiased covariance, mean over the pixels whose window lies inside the picture); that library
is not installed here, so the SSIM figure is "parity unpinned" -- bits per pixel and PSNR are pinned by bit-exactness and by
the reference CLI's report (tests/test_cli.py)."""
import io
import os

import numpy as np


def image_pad(img, multiple=32):
    """grow to multiples of `multiple` by repeating the last column / row (reference HEVCeval.py:21-41)"""
    h, w = img.shape
    return np.pad(img, ((0, -h % multiple), (0, -w % multiple)), mode="edge")


def psnr(a, b):
    """10 log10(255^2 / MSE), MSE floored at 1e-9 (reference src/HEVCeMain.c:116-133)"""
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = max(float(np.mean(d * d)), 1e-9)
    return 10.0 * np.log10(255.0 * 255.0 / mse)


def _box_means(x, win):
    """mean of every win x win window that lies inside x (integral image)"""
    c = np.zeros((x.shape[0] + 1, x.shape[1] + 1), np.float64)
    c[1:, 1:] = np.cumsum(np.cumsum(x, axis=0), axis=1)
    return (c[win:, win:] - c[:-win, win:] - c[win:, :-win] + c[:-win, :-win]) / float(win * win)


def ssim(a, b, data_range=256.0, win=7, k1=0.01, k2=0.03):
    """mean structural similarity of two equally sized grey pictures"""
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    if a.shape != b.shape or min(a.shape) < win:
        raise ValueError("ssim: pictures must have the same shape, at least %d x %d" % (win, win))
    n = win * win
    unbias = n / (n - 1.0)
    ma, mb = _box_means(a, win), _box_means(b, win)
    va = unbias * (_box_means(a * a, win) - ma * ma)
    vb = unbias * (_box_means(b * b, win) - mb * mb)
    cab = unbias * (_box_means(a * b, win) - ma * mb)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ma * mb + c1) * (2 * cab + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2))
    return float(np.mean(s))


def read_grey(path):
    """a PGM through the CLI's own dialect, anything else through Pillow (converted to 8-bit grey)"""
    if path.lower().endswith(".pgm"):
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if root not in sys.path:
            sys.path.insert(0, root)
        import hevce_workload as wl
        return wl.read_pgm(path)
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("L"))


# name, Pillow format, save options for a quality value, lowest and highest+1 quality of the bisection (reference HEVCeval.py:102-108)
CODECS = [
    ("JPEG", "JPEG", lambda q: dict(optimize=True, quality=q), 1, 101),
    ("JPEG2000", "JPEG2000", lambda q: dict(quality_mode="dB", quality_layers=[q]), 25, 75),
    ("WEBP", "WEBP", lambda q: dict(quality=q), 1, 101),
]


def nearest_ssim_setting(img, target_ssim, fmt, options, lo, hi):
    """bisection over the codec's quality value; returns (ssim, bytes, quality) of the trial nearest to target_ssim"""
    from PIL import Image
    trials = []
    while hi - lo > 1:
        q = (hi + lo) // 2
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format=fmt, **options(q))
        blob = buf.getvalue()
        with Image.open(io.BytesIO(blob)) as im:
            s = ssim(img, np.asarray(im.convert("L")))
        trials.append((abs(s - target_ssim), s, len(blob), q))
        if s < target_ssim:
            lo = q
        else:
            hi = q
    _, s, size, q = min(trials)
    return s, size, q


def evaluate(named_images, qpd6, encode_batch, compare=True):
    """named_images: [(name, uint8 [h, w])]; encode_batch: the package's encode_batch.  Returns one dict per picture."""
    padded = [image_pad(np.ascontiguousarray(im, np.uint8)) for _, im in named_images]
    results = []
    for (name, _), img, (stream, rcon) in zip(named_images, padded, encode_batch(padded, qpd6)):
        row = dict(name=name, height=img.shape[0], width=img.shape[1], stream=stream, rcon=rcon, bytes=len(stream),
                   bpp=8.0 * len(stream) / img.size, ssim=ssim(img, rcon), psnr=psnr(img, rcon), others=[])
        if compare:
            for cname, fmt, options, lo, hi in CODECS:
                try:
                    s, size, q = nearest_ssim_setting(img, row["ssim"], fmt, options, lo, hi)
                except Exception as e:                              # a codec this Pillow build lacks
                    row["others"].append(dict(codec=cname, error=str(e)))
                    continue
                row["others"].append(dict(codec=cname, ssim=s, bytes=size, bpp=8.0 * size / img.size, quality=q,
                                          size_ratio=size / len(stream)))
        results.append(row)
    return results
