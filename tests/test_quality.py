"""the quality figures of the reference's evaluation run (reference HEVCeval.py:159-235, PSNR of
src/HEVCeMain.c:116-133) on top of the library.  The SSIM library the reference calls is not installed, so SSIM is checked
against a literal per-window evaluation of the published formula ("parity unpinned" for its absolute value); bits per
pixel and PSNR follow from bit-exactness and are pinned by SURVEY.md 8(c)'s known answers on the GPU."""
import importlib
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def quality(hevce):
    return importlib.import_module(hevce.__name__ + ".quality")


def brute_ssim(a, b, R=256.0, win=7):
    a = a.astype(np.float64); b = b.astype(np.float64)
    c1, c2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    vals = []
    for y in range(a.shape[0] - win + 1):
        for x in range(a.shape[1] - win + 1):
            p, q = a[y:y + win, x:x + win].ravel(), b[y:y + win, x:x + win].ravel()
            mp, mq = p.mean(), q.mean()
            vp, vq = p.var(ddof=1), q.var(ddof=1)
            cov = ((p - mp) * (q - mq)).sum() / (p.size - 1)
            vals.append((2 * mp * mq + c1) * (2 * cov + c2) / ((mp * mp + mq * mq + c1) * (vp + vq + c2)))
    return float(np.mean(vals))


def test_ssim_equals_the_formula_window_by_window(quality):
    rng = np.random.RandomState(3)
    a = rng.randint(0, 256, (19, 23)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.randint(-20, 21, a.shape), 0, 255).astype(np.uint8)
    assert abs(quality.ssim(a, b) - brute_ssim(a, b)) < 1e-12
    assert abs(quality.ssim(a, b) - quality.ssim(b, a)) < 1e-15
    assert quality.ssim(a, a) == pytest.approx(1.0, abs=1e-15)
    assert quality.ssim(a, b) < quality.ssim(a, np.clip(a.astype(np.int32) + 1, 0, 255).astype(np.uint8)) < 1.0
    with pytest.raises(ValueError):
        quality.ssim(a, b[:, :-1])


def test_pad_and_psnr(quality):
    img = np.arange(5 * 7, dtype=np.uint8).reshape(5, 7)
    p = quality.image_pad(img)
    assert p.shape == (32, 32) and np.array_equal(p[:5, :7], img)
    assert np.all(p[2, 7:] == img[2, 6]) and np.all(p[5:, 3] == img[4, 3]) and np.all(p[5:, 7:] == img[4, 6])
    assert quality.image_pad(p) is not None and quality.image_pad(p).shape == (32, 32)
    assert quality.psnr(img, img) == pytest.approx(10 * np.log10(255 * 255 / 1e-9))
    assert quality.psnr(np.zeros((4, 4), np.uint8), np.full((4, 4), 5, np.uint8)) == pytest.approx(10 * np.log10(65025 / 25.0))


def test_other_codecs_are_tuned_to_the_nearest_ssim(quality):
    pytest.importorskip("PIL")
    import hevce_workload as wl
    img = wl.gen_g2(64, 96, 8, seed=11)
    noisy = np.clip(img.astype(np.int32) + np.random.RandomState(1).randint(-3, 4, img.shape), 0, 255).astype(np.uint8)
    target = quality.ssim(img, noisy)
    s, size, q = quality.nearest_ssim_setting(img, target, "JPEG", lambda v: dict(optimize=True, quality=v), 1, 101)
    assert 1 <= q <= 100 and size > 0 and abs(s - target) < 0.05


@pytest.mark.gpu
def test_evaluation_run_on_the_gpu(hevce, quality, manifest, tmp_path):
    """kodak 01 at qpd6 = 2: 163557 bytes and 47.5013 dB are SURVEY.md 8(c)'s known answers of the reference"""
    import hevce_workload as wl
    from conftest import GOLD
    k01 = wl.read_pgm(os.path.join(GOLD, "kodak", "01.pgm"))
    odd = wl.gen_g2(70, 100, 16, seed=8)
    rows = quality.evaluate([("01.pgm", k01), ("odd.pgm", odd)], 2, hevce.encode_batch, compare=False)
    assert rows[0]["bytes"] == 163557 and rows[0]["bpp"] == pytest.approx(8 * 163557 / (768 * 512))
    assert round(rows[0]["psnr"], 4) == 47.5013 and 0.98 < rows[0]["ssim"] < 1.0
    assert (rows[1]["height"], rows[1]["width"]) == (96, 128)
    s, r = hevce.HEVCImageEncoder(quality.image_pad(odd), 2)
    assert rows[1]["stream"] == s and np.array_equal(rows[1]["rcon"], r)
