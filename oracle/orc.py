"""ctypes bindings for the CPU oracle and (when built) the real reference in oracle/_ref.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)


def _p(a, t=_u8p):
    return a.ctypes.data_as(t)


def build(force=False):
    so = os.path.join(HERE, "libhevce_oracle.so")
    src = os.path.join(HERE, "hevce_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s", "all"])
    return so


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_HEVCImageEncoder.restype = C.c_int
    return _lib


def ref():
    """The real reference (oracle/_ref/libhevce_ref.so) or None when it was never built."""
    global _ref
    if _ref is None:
        so = os.path.join(HERE, "_ref", "libhevce_ref.so")
        if not os.path.exists(so):
            return None
        _ref = C.CDLL(so)
        _ref.HEVCImageEncoder.restype = C.c_int
    return _ref


def pad32(v):
    return (min(v, 8192) + 31) // 32 * 32


def _encode(fn, img, q):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    H, W = pad32(h), pad32(w)
    out = np.zeros(H * W * 2 + 4096, np.uint8)
    rcon = np.zeros((H, W), np.uint8)
    ys, xs = C.c_int(h), C.c_int(w)
    n = fn(_p(out), _p(img), _p(rcon), C.byref(ys), C.byref(xs), C.c_int(q))
    assert (ys.value, xs.value) == (H, W)
    return out[:n].tobytes(), rcon


def encode(img, q):
    """oracle restatement: (stream bytes, padded reconstruction)"""
    return _encode(lib().orc_HEVCImageEncoder, img, q)


def encode_ref(img, q):
    """real reference compiled into oracle/_ref"""
    return _encode(ref().HEVCImageEncoder, img, q)


def analyse(img, q):
    """records (nctu, 2176) uint8, rcon, per-CTU coder state (nctu, 8) int32"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    H, W = pad32(h), pad32(w)
    nctu = (H // 32) * (W // 32)
    rec = np.zeros((nctu, 2176), np.uint8)
    st = np.zeros((nctu, 8), np.int32)
    rcon = np.zeros((H, W), np.uint8)
    ys, xs = C.c_int(h), C.c_int(w)
    lib().orc_analyse(_p(img), _p(rcon), _p(rec), _p(st, _i32p), C.byref(ys), C.byref(xs), C.c_int(q))
    return rec, rcon, st


def write_stream(records, H, W, q):
    out = np.zeros(H * W * 2 + 4096, np.uint8)
    n = lib().orc_write_stream(_p(out), _p(np.ascontiguousarray(records)), C.c_int(H), C.c_int(W), C.c_int(q))
    return out[:n].tobytes()


def read_pgm(path):
    """P5 reader with the reference CLI's dialect (HEVCeMain.c:9-64): three ints, one whitespace byte, raw bytes."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:2] == b"P5"
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
    assert mx <= 255
    return np.frombuffer(data, np.uint8, w * h, pos).reshape(h, w).copy()


def gen_g2(h, w, noise, seed=0x1234ABCD):
    """SURVEY.md 8(d) generator G2-N: xorshift32 noise of amplitude N over two triangle waves (integer only)."""
    n = h * w
    s = np.uint64(seed)
    # sequential xorshift32 -- vectorised by running the recurrence in python ints on a flat array
    out = np.empty(n, np.uint32)
    sv = int(seed) & 0xFFFFFFFF
    for i in range(n):
        sv ^= (sv << 13) & 0xFFFFFFFF
        sv ^= sv >> 17
        sv ^= (sv << 5) & 0xFFFFFFFF
        out[i] = sv
    del s
    r = ((out >> 24) & (2 * noise - 1)).astype(np.int32).reshape(h, w) - noise
    x = np.arange(w, dtype=np.int32)[None, :]
    y = np.arange(h, dtype=np.int32)[:, None]
    tri = lambda t, P: np.abs((t % (2 * P)) - P)
    v = 32 + tri(x, 128) + tri(y, 64) + r
    return np.clip(v, 0, 255).astype(np.uint8)


def analyse_fast_rule(img, q):
    """orc_analyse_fast_rule: the analysis decided with the PRODUCT's rule (bit counts without the trial encodes' emulation-prevention
    bytes, flagged CTUs decided again exactly).  Returns (records, rcon, state, stats = [CTUs, decided twice, trial encodes with such a byte])."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    H, W = pad32(h), pad32(w)
    nctu = (H // 32) * (W // 32)
    rec = np.zeros((nctu, 2176), np.uint8)
    st = np.zeros((nctu, 8), np.int32)
    rcon = np.zeros((H, W), np.uint8)
    stats = np.zeros(4, np.int32)
    ys, xs = C.c_int(h), C.c_int(w)
    lib().orc_analyse_fast_rule(_p(img), _p(rcon), _p(rec), _p(st, _i32p), C.byref(ys), C.byref(xs), C.c_int(q), _p(stats, _i32p))
    return rec, rcon, st, [int(stats[0]), int(stats[1]), int(stats[2]) + (int(stats[3]) << 31)]
