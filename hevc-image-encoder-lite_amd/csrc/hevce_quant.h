/*
 * hevce_quant.h -- the per-coefficient decision of the reference's simplified RDOQ  [ref HEVCe.c:526-595], written once and used
 * twice: by the HOST, which tabulates it (hevce_host.cpp: for a fixed qpd6 and transform size the level chosen for a
 * coefficient is a function of its magnitude alone, |coefficient| <= 32640 for 8-bit samples), and by the CPU emulation of
 * the device code.  The device looks the level up (hv_qlut, hevce_device.h) instead of spending ~100 instructions per
 * coefficient on three rate-distortion costs.
 * Needs from its includer: HD, HV_I32MAX, hv_min, hv_max, hv_abs, hv_clip16, hv_clz, hv_umul24.
 */
#pragma once
#define HEVCE_QLUT_N 32768                                   /* magnitudes 0 .. 32767 per (transform size, qpd6) table */
static HD int hv_level_rate(int lv) {                                                                      /* [ref :526-535] */
    /* lv < 6: table {0, 70000, 90000, 92000, 157536, 190304}; else 92000 + ((4 + 2*prefix_len(lv-6)) << 15) -- selects only */
    const int i = 31 - hv_clz((unsigned)hv_max(lv - 5, 1));
    const int big = 92000 + ((4 + 2 * i) << 15);
    int r = lv == 0 ? 0 : 70000;
    r = lv >= 2 ? 90000 : r;
    r = lv >= 3 ? 92000 : r;
    r = lv >= 4 ? 157536 : r;
    r = lv >= 5 ? 190304 : r;
    return lv >= 6 ? big : r;
}
/* RD cost with the per-qpd6 weights already selected (wd, wb) and their overflow guards (I32MAX / w)  [ref :177-185] */
struct HvRd { int wd, wb, gd, gb; };
static HD HvRd hv_rd_make(int q) {
    HvRd r;
    r.wd = (q < 3) ? 11 : (q == 3 ? 5 : 1);
    r.wb = (q == 0) ? 1 : (q == 1) ? 4 : (q == 2) ? 16 : (q == 3) ? 29 : 23;
    r.gd = HV_I32MAX / r.wd; r.gb = HV_I32MAX / r.wb;
    return r;
}
static HD int hv_rd(const HvRd &w, int dist, int bits) {
    const int a = (w.gd <= dist) ? HV_I32MAX : w.wd * dist;
    const int b = (w.gb <= bits) ? HV_I32MAX : w.wb * bits;
    return (HV_I32MAX - a <= b) ? HV_I32MAX : a + b;
}
/* returns the level magnitude; d = scaled magnitude used by the group zero-out test.  Three candidates (level, level-1,
   level-2, not below zero), strict '<' so the highest level wins ties  [ref :566-580]; straight-line code */
static HD int hv_quant1(int q, int sh, int dsh, int av, int &d) {
    const HvRd w = hv_rd_make(q);
    const int half = 1 << sh >> 1, dmax = HV_I32MAX - half;
    d = (av > 0x1ffff) ? dmax : hv_min((av & 0x1ffff) << 14, dmax);
    const int lv0 = hv_clip16((d + half) >> sh);
    int best = HV_I32MAX, pick = 0;
    /* The third candidate can only win when the reference's saturating square is reached, i.e. for qpd6 = 4: with
       E = 2^(sh - dsh) = 2^(11 + q) the scaled distances of candidates 1 and 2 are e1 in [E/2, 3E/2) and e2 = e1 + E, so their
       distortions differ by at least (2 e1 E + E^2) / 128 >= E^2 / 64, weighted 11, 11, 11, 5 for q = 0..3: >= 7.2e5, 2.9e6,
       1.15e7, 2.1e7 -- an order of magnitude above the largest rate step times its weight (70000 x 1, 4, 16, 29).  For
       q = 4 both squares can saturate to the same value and the cheaper level wins. */
    const int ncand = (q >= 4) ? 3 : 2;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        if (t >= ncand) break;
        const int lv = hv_max(lv0 - t, 0);
        const int e = hv_abs(d - (lv << sh)) >> dsh;
        const int dist = ((e < 46340) ? (int)hv_umul24((unsigned)e, (unsigned)e) : HV_I32MAX) >> 7;   /* 46339^2 < 2^31 */
        /* hv_rd() without its overflow guards: dist <= 2^24 - 1 and the level rate <= 1 140 576, so with wd <= 11 and
           wb <= 29 neither product nor their sum can reach 2^31  [ref :177-185 saturates only beyond that] */
        const int cst = (int)(hv_umul24((unsigned)w.wd, (unsigned)dist) + hv_umul24((unsigned)w.wb, (unsigned)hv_level_rate(lv)));   /* all operands < 2^24 */
        const bool take = (lv0 - t >= 0) && (cst < best);
        best = take ? cst : best;
        pick = take ? lv : pick;
    }
    return pick;
}


/* table t = log2(size) - 2 (4x4 .. 32x32) of qpd6 q, entry av: the level magnitude hv_quant1 picks */
static HD int hv_qlut_entry(int q, int t, int av) {
    int d;
    return hv_quant1(q, 19 - t + q, 8 - t, av, d);
}
