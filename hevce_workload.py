"""Synthetic workloads of BASELINE.json / SURVEY.md section 8(d), shared by bench.py and the tests.

G2-N: per pixel (raster order) s = xorshift32(s); v = 32 + tri(x,128) + tri(y,64) + (((s>>24) & (2N-1)) - N), clamped.
The xorshift recurrence is linear over GF(2), so the sequence is produced by doubling (x[L..2L) = M^L x[0..L)) instead
of a per-pixel Python loop; tests check it against the plain sequential form.
"""
import numpy as np


def _xs32_step(v):
    v = v ^ ((v << np.uint32(13)) & np.uint32(0xFFFFFFFF))
    v = v ^ (v >> np.uint32(17))
    v = v ^ ((v << np.uint32(5)) & np.uint32(0xFFFFFFFF))
    return v


def xorshift32_sequence(seed, n):
    """x[0] = step(seed), x[i] = step(x[i-1]) as uint32[n]."""
    seq = np.empty(max(n, 1), np.uint32)
    seq[0] = _xs32_step(np.array([seed & 0xFFFFFFFF], np.uint32))[0]
    # columns of M^L: images of the 32 unit vectors under L steps
    cols = np.uint32(1) << np.arange(32, dtype=np.uint32)
    cols = _xs32_step(cols)          # M^1
    have = 1
    while have < n:
        take = min(have, n - have)
        src = seq[:take]
        out = np.zeros(take, np.uint32)
        for b in range(32):
            out ^= np.where((src >> np.uint32(b)) & np.uint32(1), cols[b], np.uint32(0)).astype(np.uint32)
        seq[have:have + take] = out
        have += take
        # square the matrix: M^(2L) columns = M^L applied to M^L columns
        nxt = np.zeros(32, np.uint32)
        for b in range(32):
            nxt ^= np.where((cols >> np.uint32(b)) & np.uint32(1), cols[b], np.uint32(0)).astype(np.uint32)
        cols = nxt
    return seq[:n]


def gen_g2(h, w, noise=16, seed=0x1234ABCD):
    s = xorshift32_sequence(seed, h * w)
    r = ((s >> np.uint32(24)) & np.uint32(2 * noise - 1)).astype(np.int32).reshape(h, w) - noise
    x = np.arange(w, dtype=np.int32)[None, :]
    y = np.arange(h, dtype=np.int32)[:, None]
    v = 32 + np.abs((x % 256) - 128) + np.abs((y % 128) - 64) + r
    return np.clip(v, 0, 255).astype(np.uint8)


def read_pgm(path):
    """P5 with the reference CLI's dialect (reference src/HEVCeMain.c:9-64)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"P5":
        raise ValueError("not a P5 file")
    pos, vals = 2, []
    while len(vals) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        s = pos
        while data[pos:pos + 1].isdigit():
            pos += 1
        vals.append(int(data[s:pos]))
    pos += 1
    w, h, mx = vals
    if mx > 255 or len(data) - pos < w * h:
        raise ValueError("unsupported or truncated PGM")
    return np.frombuffer(data, np.uint8, w * h, pos).reshape(h, w).copy()
