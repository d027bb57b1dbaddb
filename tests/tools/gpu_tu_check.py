#!/usr/bin/env python3
"""Function-level parity on the GPU: TU pipeline of all 35 modes vs the oracle, per size / qpd6."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import hevce_pkg, orc
hevce = hevce_pkg.load(); L = hevce.lib(); O = orc.lib()
vp = C.c_void_p


def check(seed=1, extreme=False, all_availabilities=True):
  """returns [(n, q, flags, bad modes)] for every (size, qpd6, neighbour availability) with a mismatch; extreme: samples only
  0 / 255 (the largest residuals and transform sums 8-bit input can produce).  all_availabilities: every combination of
  (left, left-below, above, above-right) the reference's getBorder substitutes for (HEVCe.c:212-243) -- a 32x32 block at the CTU
  origin has no left-below rows, so 8 combinations there, 16 for the other sizes."""
  rng = np.random.RandomState(seed)
  failures = []
  for n in (4, 8, 16, 32):
    for q in (0, 1, 2, 3, 4):
      combos = [(l, lb, a, ar) for l in (1, 0) for lb in (1, 0) for a in (1, 0) for ar in (1, 0)] if all_availabilities else [(1, 1, 1, 1)]
      if n == 32: combos = [c for c in combos if c[1] == 0] if all_availabilities else [(1, 0, 1, 1)]
      # the whole mode / size matrix at full availability; the substitution rules on a rotating subset of qpd6 to bound the run time
      if all_availabilities: combos = [c for k, c in enumerate(combos) if c == combos[0] or k % 5 == q]
      for (aL, aLB, aA, aAR) in combos:
          org = rng.randint(0, 256, (32, 32)).astype(np.uint8)
          if q == 2 and not extreme: org = (rng.randint(-16, 17, (32, 32)) + 128 + 80).clip(0, 255).astype(np.uint8)
          nbr = rng.randint(0, 256, (33, 65)).astype(np.uint8)
          if extreme: org = (rng.randint(0, 2, (32, 32)) * 255).astype(np.uint8); nbr = (rng.randint(0, 2, (33, 65)) * 255).astype(np.uint8)
          lev = np.zeros((35, 1024), np.int16); rec = np.zeros((35, 1024), np.uint8); info = np.zeros((35, 4), np.int32)
          rc = L.hevce_test_tu(org.ctypes.data_as(vp), nbr.ctypes.data_as(vp), n, q, aL, aLB, aA, aAR, lev.ctypes.data_as(vp), rec.ctypes.data_as(vp), info.ctypes.data_as(vp))
          assert rc == 0
          # the oracle's substituted reference samples for these flags (orc_fetch_refs = reference getBorder), then its block trial
          refs = np.zeros(2 * (1 + 4 * n), np.uint8)
          blk0 = nbr.ctypes.data + 65 + 1                     # &nbr[1][1]: the block's top-left sample, stride 65
          O.orc_fetch_refs(n, vp(blk0), 65, aL, aLB, aA, aAR, refs.ctypes.data_as(vp))
          corner = int(refs[0])
          left = np.ascontiguousarray(refs[1:1 + 2 * n]); above = np.ascontiguousarray(refs[1 + 2 * n:1 + 4 * n])
          bad = []
          for m in range(35):
              olev = np.zeros((n, n), np.int32); orec = np.zeros((n, n), np.uint8)
              blk = np.ascontiguousarray(org[:n, :n])
              sse = O.orc_tu_trial(q, n, m, corner, left.ctypes.data_as(vp), above.ctypes.data_as(vp), blk.ctypes.data_as(vp), olev.ctypes.data_as(vp), orec.ctypes.data_as(vp))
              kind = 0
              if n <= 8: kind = 1 if abs(m - 26) <= 4 else (2 if abs(m - 10) <= 4 else 0)
              sc = np.zeros((n * n, 2), np.uint8); O.orc_scan(n, kind, sc.ctypes.data_as(vp))
              oscan = olev[sc[:, 0], sc[:, 1]]
              ok = (lev[m, :n * n] == oscan).all() and (rec[m, :n * n].reshape(n, n) == orec).all() and info[m, 0] == sse
              if not ok: bad.append((m, int((lev[m, :n * n] != oscan).sum()), int((rec[m, :n * n].reshape(n, n) != orec).sum()), int(info[m, 0]), sse))
          if bad: failures.append((n, q, (aL, aLB, aA, aAR), bad[:6]))
  return failures


if __name__ == "__main__":
    for ext in (False, True):
        print("extreme" if ext else "random", "failures:", check(1, ext), flush=True)
