#!/usr/bin/env python3
"""Function-level parity on the GPU: TU-split chains vs a chain assembled from oracle primitives."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hevce_pkg, orc
hevce = hevce_pkg.load(); L = hevce.lib(); O = orc.lib()
vp = C.c_void_p


def sub_avail(s, Lf, LB, A, AR):
    return ((1 if s & 1 else Lf), (Lf if s == 0 else (LB if s == 2 else 0)), (1 if s & 2 else A), (A if s == 0 else (AR if s == 1 else (1 if s == 2 else 0))))


def borders(buf, y, x, n, l, lb, a, ar):
    """buf[1+yy, 1+xx] = sample (yy,xx); reference getBorder substitution rules"""
    g = lambda yy, xx: int(buf[1 + yy, 1 + xx])
    c = g(y - 1, x - 1) if (l and a) else (g(y, x - 1) if l else (g(y - 1, x) if a else 128))
    left = [g(y + i, x - 1) if l else c for i in range(n)]
    left += [g(y + i, x - 1) if lb else left[n - 1] for i in range(n, 2 * n)]
    above = [g(y - 1, x + i) if a else c for i in range(n)]
    above += [g(y - 1, x + i) if ar else above[n - 1] for i in range(n, 2 * n)]
    return c, np.array(left, np.uint8), np.array(above, np.uint8)


def oracle_chain(org, nbr, n, q, m, fl):
    h = n // 2
    buf = np.zeros((66, 66), np.uint8); buf[:33, :65] = nbr
    levs, sse_t = [], 0
    for s in range(4):
        oy, ox = (s >> 1) * h, (s & 1) * h
        l, lb, a, ar = sub_avail(s, *fl)
        c, left, above = borders(buf, oy, ox, h, l, lb, a, ar)
        blk = np.ascontiguousarray(org[oy:oy + h, ox:ox + h])
        olev = np.zeros((h, h), np.int32); orec = np.zeros((h, h), np.uint8)
        sse_t += O.orc_tu_trial(q, h, m, c, left.ctypes.data_as(vp), above.ctypes.data_as(vp), blk.ctypes.data_as(vp), olev.ctypes.data_as(vp), orec.ctypes.data_as(vp))
        buf[1 + oy:1 + oy + h, 1 + ox:1 + ox + h] = orec
        kind = 0
        if h <= 8: kind = 1 if abs(m - 26) <= 4 else (2 if abs(m - 10) <= 4 else 0)
        sc = np.zeros((h * h, 2), np.uint8); O.orc_scan(h, kind, sc.ctypes.data_as(vp))
        levs.append(olev[sc[:, 0], sc[:, 1]])
    return np.concatenate(levs), buf[1:1 + n, 1:1 + n].copy(), sse_t


def check(seed=3, extreme=False):
    """returns [(n, q, flags, bad modes)] for every case with a mismatch; extreme: samples only 0 / 255"""
    rng = np.random.RandomState(seed)
    failures = []
    for n in (8, 16, 32):
        for q in (2, 0, 4, 1, 3):
            for fl in ((0, 0, 0, 0), (1, 0, 1, 1), (1, 1 if n < 32 else 0, 1, 0)):
                org = (rng.randint(-16, 17, (32, 32)) + 200).clip(0, 255).astype(np.uint8) if q == 2 else rng.randint(0, 256, (32, 32)).astype(np.uint8)
                nbr = rng.randint(0, 256, (33, 65)).astype(np.uint8)
                if extreme: org = (rng.randint(0, 2, (32, 32)) * 255).astype(np.uint8); nbr = (rng.randint(0, 2, (33, 65)) * 255).astype(np.uint8)
                lev = np.zeros((35, 1024), np.int16); rec = np.zeros((35, 1024), np.uint8); info = np.zeros((35, 16), np.int32)
                assert L.hevce_test_chain(org.ctypes.data_as(vp), nbr.ctypes.data_as(vp), n, q, *fl, lev.ctypes.data_as(vp), rec.ctypes.data_as(vp), info.ctypes.data_as(vp)) == 0
                bad = []
                for m in range(35):
                    ol, orc_, osse = oracle_chain(org, nbr, n, q, m, fl)
                    nl = int((lev[m, :n * n] != ol).sum()); nr = int((rec[m, :n * n].reshape(n, n) != orc_).sum())
                    if nl or nr or info[m, 0] != osse: bad.append((m, nl, nr, int(info[m, 0]), osse))
                if bad: failures.append((n, q, fl, bad[:5]))
    return failures


if __name__ == "__main__":
    for ext in (False, True):
        print("extreme" if ext else "random", "failures:", check(3, ext), flush=True)
