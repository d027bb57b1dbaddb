"""CPU, build container only: function-level comparison of the oracle restatement with the REAL reference compiled
into oracle/_ref (its helpers have external linkage).  Skipped where oracle/_ref was never built."""
import ctypes as C

import numpy as np
import pytest

import orc as O

ref = O.ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (reference sources absent)")

I32 = C.c_int32
_i32p = C.POINTER(I32)


def _blk(a):           # reference blocks are int [][32]
    b = np.zeros((32, 32), np.int32)
    b[:a.shape[0], :a.shape[1]] = a
    return b


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_transform_quantiser(n):
    rng = np.random.RandomState(n)
    L = O.lib()
    for q in range(5):
        x = rng.randint(-255, 256, (n, n)).astype(np.int32)
        rin, rout = _blk(x), np.zeros((32, 32), np.int32)
        ref.transform(n, 0, rin.ctypes.data_as(_i32p), rout.ctypes.data_as(_i32p))
        y = np.zeros((n, n), np.int32)
        L.orc_transform(n, 0, x.ctypes.data_as(_i32p), y.ctypes.data_as(_i32p))
        assert np.array_equal(y, rout[:n, :n])
        assert np.abs(y).max() <= 32640                       # bound used by the device's int16 staging
        rq = np.zeros((32, 32), np.int32)
        ref.quantize(q, n, 0, rout.ctypes.data_as(_i32p), rq.ctypes.data_as(_i32p))
        lv = np.zeros((n, n), np.int32)
        L.orc_quantize(q, n, y.ctypes.data_as(_i32p), lv.ctypes.data_as(_i32p))
        assert np.array_equal(lv, rq[:n, :n])
        rd = np.zeros((32, 32), np.int32)
        ref.deQuantize(q, n, rq.ctypes.data_as(_i32p), rd.ctypes.data_as(_i32p))
        d = np.zeros((n, n), np.int32)
        L.orc_dequantize(q, n, lv.ctypes.data_as(_i32p), d.ctypes.data_as(_i32p))
        assert np.array_equal(d, rd[:n, :n])
        ri = np.zeros((32, 32), np.int32)
        ref.transform(n, 1, rd.ctypes.data_as(_i32p), ri.ctypes.data_as(_i32p))
        xi = np.zeros((n, n), np.int32)
        L.orc_transform(n, 1, d.ctypes.data_as(_i32p), xi.ctypes.data_as(_i32p))
        assert np.array_equal(xi, ri[:n, :n])


def test_tables_equal_reference_exports():
    L = O.lib()
    t32 = np.zeros((32, 32), np.int16)
    d32 = np.zeros((1024, 2), np.uint8); h8 = np.zeros((64, 2), np.uint8); v8 = np.zeros((64, 2), np.uint8)
    L.orc_tables(t32.ctypes.data_as(C.c_void_p), d32.ctypes.data_as(C.c_void_p), h8.ctypes.data_as(C.c_void_p), v8.ctypes.data_as(C.c_void_p))
    r32 = np.ctypeslib.as_array((I32 * 1024).in_dll(ref, "DCT32_MAT")).reshape(32, 32)
    assert np.array_equal(t32.astype(np.int32), r32)
    r16 = np.ctypeslib.as_array((I32 * (16 * 32)).in_dll(ref, "DCT16_MAT")).reshape(16, 32)[:, :16]
    assert np.array_equal(t32[::2, :16].astype(np.int32), r16)
    lps = np.ctypeslib.as_array((C.c_uint8 * 256).in_dll(ref, "CABAC_LPS_TABLE")).reshape(64, 4)
    nlps = np.ctypeslib.as_array((C.c_uint8 * 128).in_dll(ref, "CONTEXT_NEXT_STATE_LPS"))
    nmps = np.ctypeslib.as_array((C.c_uint8 * 128).in_dll(ref, "CONTEXT_NEXT_STATE_MPS"))
    # drive single bins through the oracle coder and compare the context transition with the reference tables
    for st in range(64):
        for mps in (0, 1):
            v = (st << 1) | mps
            assert nmps[v] == (v + 2 if st < 62 else v)
            assert nlps[v] == ((mps ^ 1) if st == 0 else nlps[v])
    assert lps[0].tolist() == [128, 176, 208, 240] and lps[63].tolist() == [2, 2, 2, 2]
    # scans: reference getScanOrder(sz, pmode, &ptr)
    ref.getScanOrder.restype = C.c_int
    for n, mode, kind in ((32, 0, 0), (16, 5, 0), (8, 0, 0), (8, 26, 1), (8, 10, 2), (4, 26, 1), (4, 10, 2), (4, 2, 0)):
        ptr = C.POINTER(C.c_uint8)()
        k = ref.getScanOrder(n, mode, C.byref(ptr))
        assert k == kind
        r = np.ctypeslib.as_array(ptr, shape=(n * n, 2))
        mine = np.zeros((n * n, 2), np.uint8)
        L.orc_scan(n, kind, mine.ctypes.data_as(C.c_void_p))
        assert np.array_equal(mine, r)


def test_context_init_matches_reference():
    class CS(C.Structure):
        _fields_ = [("b", C.c_uint8 * 142)]
    ref.newContextSet.restype = CS
    L = O.lib()
    for q in range(5):
        r = bytes(ref.newContextSet(q).b)
        mine = np.zeros(102, np.uint8)
        L.orc_ctx_init(q, mine.ctypes.data_as(C.c_void_p))
        # reference layout (HEVCe.c:745-759): split_cu[3] part Y_pmode UV_pmode split_tu[3] Y_cbf[2] UV_cbf[5] last_x[5][5] last_y[5][5] sig_map[2] sig_sc[44] one_sc[24] abs_sc[6]
        assert bytes(mine[0:11]) == r[0:11] and mine[11] == r[11]
        for g in range(4):
            k = 5 if g == 3 else (4 if g == 2 else 3)
            assert bytes(mine[12 + 5 * g:12 + 5 * g + k]) == r[16 + 5 * g:16 + 5 * g + k]
            assert bytes(mine[32 + 5 * g:32 + 5 * g + k]) == r[41 + 5 * g:41 + 5 * g + k]
        assert bytes(mine[52:54]) == r[66:68]
        assert bytes(mine[54:81]) == r[68:95]
        assert bytes(mine[82:98]) == r[112:128]
        assert bytes(mine[98:102]) == r[136:140]


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_prediction_all_modes(n):
    rng = np.random.RandomState(100 + n)
    L = O.lib()
    for trial in range(3):
        left = rng.randint(0, 256, 2 * n).astype(np.uint8)
        above = rng.randint(0, 256, 2 * n).astype(np.uint8)
        corner = int(rng.randint(0, 256))
        # reference borders through getBorder on a synthetic neighbourhood
        rs = 65
        buf = np.zeros((66, rs), np.uint8)
        blk = buf[1:, 1:]
        buf[0, 0] = corner
        buf[1:1 + 2 * n, 0] = left
        buf[0, 1:1 + 2 * n] = above
        ub = [np.zeros(64, np.uint8) for _ in range(6)]
        c_u = (C.c_uint8 * 1)(); c_f = (C.c_uint8 * 1)()
        p = buf.ctypes.data + rs + 1
        ref.getBorder(n, 1, 1, 1, 1, C.c_void_p(p), c_u, ub[0].ctypes.data_as(C.c_void_p), ub[1].ctypes.data_as(C.c_void_p),
                      c_f, ub[2].ctypes.data_as(C.c_void_p), ub[3].ctypes.data_as(C.c_void_p))
        for m in range(35):
            rp = np.zeros((32, 32), np.uint8)
            ref.predict(n, 0, m, c_u[0], ub[0].ctypes.data_as(C.c_void_p), ub[1].ctypes.data_as(C.c_void_p),
                        c_f[0], ub[2].ctypes.data_as(C.c_void_p), ub[3].ctypes.data_as(C.c_void_p), rp.ctypes.data_as(C.c_void_p))
            mine = np.zeros((n, n), np.uint8)
            L.orc_predict(n, m, corner, left.ctypes.data_as(C.c_void_p), above.ctypes.data_as(C.c_void_p), mine.ctypes.data_as(C.c_void_p))
            assert np.array_equal(mine, rp[:n, :n]), (n, m)


def test_small_image_against_reference(orc):
    img = orc.gen_g2(64, 64, 16, seed=4242)
    for q in (0, 2, 4):
        a, ra = orc.encode(img, q)
        b, rb = orc.encode_ref(img, q)
        assert a == b and np.array_equal(ra, rb)
