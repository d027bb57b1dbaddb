/* hevce_dev_nxn.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   the FAST NxN sub-search with one lane per sample: 4x4 block trials, binarisation as resolved tokens, the resolved-token engine, the NxN chain and its helper wavefronts. */
/* --------------------------------------------------------------------------------------------- */
/* FAST NxN sub-search with ONE LANE PER SAMPLE.                                                                               */
/*                                                                                                                            */
/* A wavefront issues one instruction per ~4.5 cycles however many lanes are active, so a phase costs its per-lane INSTRUCTION  */
/* COUNT: with one lane per mode (hv_tu4_lane) a 4x4 block trial is ~1400 instructions, its binarisation ~800.  Here lane =     */
/* (mode slot, sample): a wavefront tries FOUR modes per pass, a mode's sixteen samples are one DPP row -- the transforms are   */
/* four multiply-adds with operands fetched from the row / the quad (no LDS, no barrier), sums are four rotations -- and the     */
/* binarisation has one lane per scan position: non-zero masks are ballots, "how many before me" a popcount.  ~450 instructions  */
/* per pass instead of ~2200 per mode.                                                                                         */
/* The bins are written as RESOLVED tokens: in CABACputBin  [ref :914-933] the context byte evolves from `bin != mps` alone, and  */
/* the trials of the sub-search all start from the FRESH context set  [ref :1505-1506], so the state a bin meets is a small       */
/* table look-up on the bins of its context before it (S.ftab).  Token: bits 16.. = context byte << 1 | bin = the index of the    */
/* engine's table entry (S.ctab2), bits 0..15 = the plain token for whoever codes the same bins from another state (candidate     */
/* 70).  The engine (hv_rengine) then has no memory access on its dependency chain: perm -> sub -> select -> ffbh -> shift.       */
/* --------------------------------------------------------------------------------------------- */
#define HV_RTOK(cv, bin, plain) ((unsigned)(plain) | ((unsigned)(((cv) << 1) | (bin)) << 16))
/* ... and, for whoever meets the same bins in ANOTHER state of the context (candidate 70's rider): how many bins of the same context the
   block has before this one, and those bins, the first in bit 0 -- kp = k | bins << 3 (k <= 6: 9 bits), in bits 24..31 and bit 13 */
#define HV_RTOK_KP(kp) ((((unsigned)(kp) & 0xffu) << 24) | ((((unsigned)(kp) >> 8) & 1u) << 13))
#define HV_RTOK_GETKP(t) ((((t) >> 24) & 0xffu) | ((((t) >> 13) & 1u) << 8))
#define HV_RTOK_NOP (((unsigned)(HV_PADCV << 1) << 16) | HV_TOK_NOPF)
/* FAST rows of S.nxn_tok (288 bytes): resolved tokens as 32-bit words from word 4 (<= 8 + 15 + 9 and 8 no-ops), 16-byte aligned for
   the engine's chunk loads; from byte 224 the block's non-zero magnitudes in coding order, 16-bit */
#define HV_NXR_TOK  4
#define HV_NXR_COMP 112
struct HvPx4 { int dcol[4], drow[4], tcol[4], trow[4]; };   /* this lane's transform coefficients (lane & 15 = y << 2 | x) */
/* cluster of four: the chain workgroup has no 32x32 candidates of its own, so their slot lists (S.slist2) are free: behind the copy of
   the NxN winners (hv_nxv_copy, 2.3 KB) sit the first 2048 entries of the picture's 4x4 level table (hv_fill_lut4; what a FAST
   picture's EXACT CTU overwrites is laid out again when the workgroup goes back to FAST) */
static HD const HV_LDSP u16 *hv_lut4_lds(const HvLds &S, int on) { return on ? (const HV_LDSP u16 *)((const HV_LDSP u8 *)&S.slist2[0][0] + 2560) : (const HV_LDSP u16 *)nullptr; }
static HD void hv_fill_lut4(HvLds &S, int tid) {
    HV_LDSP u16 *dst = (HV_LDSP u16 *)((HV_LDSP u8 *)&S.slist2[0][0] + 2560);
    const HvQlut lut = hv_qlut<2>(S);
    for (int i = tid; i < 2048; i += HV_NTHREADS) dst[i] = lut[i];
}
static HD void hv_px4_consts(const HvLds &S, HvPx4 &K, int lane) {
    const int y = (lane >> 2) & 3, x = lane & 3;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        K.dcol[s] = S.fx.d4[y * 4 + ((y + s) & 3)];           /* forward, columns: row y of the result sums D[y][k] * r[k][x], k = (y + s) & 3 */
        K.drow[s] = S.fx.d4[x * 4 + s];                     /* forward, rows   : coefficient (y, x) sums t[y][k] * D[x][k] */
        K.tcol[s] = S.fx.d4[((y + s) & 3) * 4 + y];           /* inverse, columns: D[k][y] * r[k][x] */
        K.trow[s] = S.fx.d4[s * 4 + x];                     /* inverse, rows   : t[y][k] * D[k][x] */
    }
}
static HD int hv_row_sum(int v) { v += hv_row_get<8>(v); v += hv_row_get<4>(v); v += hv_row_get<2>(v); v += hv_row_get<1>(v); return v; }   /* over the 16 lanes of a row, in every lane */
/* block trial of mode m for the sample of this lane: prediction from the block's reference samples (S.nxn_bord: corner, left[8],
   above[8]; 4x4 blocks never use the smoothed ones  [ref :274-275]), residual, DST, quantiser, dequantiser, inverse, reconstruction;
   levels go to S.nxn_lev[m] in scan order, samples to S.nxn_rec[m], the distortion to S.nxn_sse[m]   [ref :1508-1518] */
struct HvPx4Out { int lv, rc, sse, si; };                    /* this sample's level, its scan position, its reconstruction; the block's distortion */
static HD HvPx4Out hv_tu4_px_core(HvLds &S, const HvPx4 &K, const HV_LDSP u16 *lut4, const u8 *b, int q, int ty, int tx, int m, int lane) {
    const int p = lane & 15, y = p >> 2, x = p & 3;
    const u8 *l = b + 1, *a = b + 9;
    const int c = b[0];
    const bool horiz = m < 18, planar = m == 0, isdc = m == 1, hor = m == 10, ver = m == 26;
    const int ang = S.ang[m], inv = S.invang[m];
    int pred;
    {
        /* angular: two neighbours on the mode's reference line r(k) -- r(k>0) the main side, r(0) the corner, r(k<0) the side samples
           projected through the inverse angle  [ref :342-380]; pure horizontal / vertical are angle 0 plus their edge filter */
        const int step = horiz ? x : y, along = horiz ? y : x;
        const int off = hv_mul24(ang, step + 1), oi = off >> 5, of = off & 31, k1 = oi + along + 1, k2 = k1 + 1;
        const int mb = horiz ? 1 : 9, sb = horiz ? 9 : 1, ik = hv_mul24(inv, k1);
        const int i1 = k1 > 0 ? mb + hv_min(k1, 8) - 1 : (k1 == 0 ? 0 : sb + hv_min(hv_max(((128 - ik) >> 8) - 1, 0), 7));
        const int i2 = k2 > 0 ? mb + hv_min(k2, 8) - 1 : (k2 == 0 ? 0 : sb + hv_min(hv_max(((128 - ik - inv) >> 8) - 1, 0), 7));
        int v = (hv_mul24(32 - of, b[i1]) + hv_mul24(of, b[i2]) + 16) >> 5;
        const int ly = l[y], ax = a[x], l0 = l[0], a0 = a[0];
        int dc = 4;
#pragma unroll
        for (int k = 0; k < 4; k++) dc += l[k] + a[k];
        dc >>= 3;
        const int pl = (4 + hv_mul24(3 - x, ly) + hv_mul24(x + 1, a[4]) + hv_mul24(3 - y, ax) + hv_mul24(y + 1, l[4])) >> 3;
        const int dv = (y == 0 && x == 0) ? (2 + 2 * dc + l0 + a0) >> 2 : (y == 0) ? (2 + 3 * dc + ax) >> 2 : (x == 0) ? (2 + 3 * dc + ly) >> 2 : dc;
        v = (y == 0 && hor) ? hv_clip8(l0 + ((ax - c) >> 1)) : v;
        v = (x == 0 && ver) ? hv_clip8(a0 + ((ly - c) >> 1)) : v;
        pred = planar ? pl : (isdc ? dv : v);
    }
    const int org = S.org[ty + y][tx + x];
    const int r = org - pred;
    /* forward: columns (shift 1), rows (shift 8)  [ref :497-516] */
    const int t = (hv_mul24(K.dcol[0], r) + hv_mul24(K.dcol[1], hv_row_get<4>(r)) + hv_mul24(K.dcol[2], hv_row_get<8>(r)) + hv_mul24(K.dcol[3], hv_row_get<12>(r)) + 1) >> 1;
    const int cf = (hv_mul24(K.drow[0], hv_quad_get<0>(t)) + hv_mul24(K.drow[1], hv_quad_get<1>(t)) + hv_mul24(K.drow[2], hv_quad_get<2>(t)) + hv_mul24(K.drow[3], hv_quad_get<3>(t)) + 128) >> 8;
    /* quantiser  [ref :540-595] */
    const int sh = 19 + q, thr = 9 << sh >> 2, qsh = 5 + q;
    const int av = hv_abs(cf), d = av << 14;
    int mag;
    if (lut4) {                                                  /* the first 2048 entries of the 4x4 level table are in LDS (cluster of four) */
        mag = lut4[hv_min(av, 2047)];
        if (hv_ballot(av > 2047)) { int d2; const int g = hv_quant_lut(hv_qlut<2>(S), av, d2); mag = av > 2047 ? g : mag; }
    } else { int d2; mag = hv_quant_lut(hv_qlut<2>(S), av, d2); }
    const int zero = hv_row_sum(hv_min(d, thr)) < thr;
    const int lv = zero ? 0 : (cf < 0 ? -mag : mag);
    const int dq = hv_clip16((int)((unsigned)lv << qsh));
    /* inverse: columns with C^T (shift 7), rows (shift 12) */
    const int t2 = hv_clip16((hv_mul24(K.tcol[0], dq) + hv_mul24(K.tcol[1], hv_row_get<4>(dq)) + hv_mul24(K.tcol[2], hv_row_get<8>(dq)) + hv_mul24(K.tcol[3], hv_row_get<12>(dq)) + 64) >> 7);
    const int res = hv_clip16((hv_mul24(K.trow[0], hv_quad_get<0>(t2)) + hv_mul24(K.trow[1], hv_quad_get<1>(t2)) + hv_mul24(K.trow[2], hv_quad_get<2>(t2)) + hv_mul24(K.trow[3], hv_quad_get<3>(t2)) + 2048) >> 12);
    const int rc = hv_clip8(res + pred), dd = org - rc;
    HvPx4Out o;
    o.lv = lv; o.rc = rc; o.sse = hv_row_sum(hv_mul24(dd, dd)); o.si = S.fx.scan4[hv_scan_kind(4, m)][p];
    return o;
}
static HD void hv_tu4_px(HvLds &S, const HvPx4 &K, const HV_LDSP u16 *lut4, int q, int ty, int tx, int m, int ok, int lane) {
    m = ok ? m : 1;
    const HvPx4Out o = hv_tu4_px_core(S, K, lut4, S.nxn_bord, q, ty, tx, m, lane);
    if (ok) {
        S.nxn_lev[m][o.si] = (short)o.lv;
        S.nxn_rec[m][lane & 15] = (u8)o.rc;
        if ((lane & 15) == 0) S.nxn_sse[m] = o.sse;
    }
}
/* bins of coeff_abs_level_remaining  [ref :1154-1169] for value v at Rice parameter k (the count hv_emit_remainder makes) */
static HD int hv_remainder_bins(int v, int k) {
    const int shortf = v < (3 << k);
    const int w2 = hv_max(v - (3 << k), 0) + (1 << k);
    const int len = 31 - hv_clz((unsigned)w2);
    return shortf ? (v >> k) + 1 + k : 4 + 2 * len - k;
}
/* binarisation of the same four blocks, lane = (mode slot, scan position); needs the levels of hv_tu4_px (a wave barrier in
   between).  Writes the mode's resolved tokens, its token / bypass-bin counts (S.nxn_magn) and last position / any-level flag.
   [ref :1046-1087 last position, :1194-1268 one group that is both first and last, :1154-1169] */
static HD void hv_tok4_px(HvLds &S, int m, int ok, int lane) {
    const int ii = lane & 15, s16 = lane & 48;
    const int v = ok ? S.nxn_lev[m][ii] : 0, av = hv_abs(v);
    m = ok ? m : 0;
    const unsigned nzm = (unsigned)(hv_ballot(v != 0) >> s16) & 0xffffu;
    const int nz = __builtin_popcount(nzm), sig = nzm != 0;
    const int last = sig ? 31 - hv_clz(nzm) : 0, start = last;
    const int ks = hv_scan_kind(4, m);
    /* tokens that do not exist (and everything of a lane without a mode) go to word HV_NXR_DUMMY of the row: no divergent stores */
    HV_LDSP unsigned *row = (HV_LDSP unsigned *)S.nxn_tok[m] + HV_NXR_TOK;
    HV_LDSP u16 *comp = (HV_LDSP u16 *)S.nxn_tok[m] + HV_NXR_COMP;
    #define HV_NXR_DUMMY 64
    /* ---- last_sig_coeff_{x,y}_prefix: a coordinate c < 3 is c ones and a zero, 3 is three ones; one context per bin.
       Lanes 0..3 of the mode write the x tokens, 4..7 the y tokens ---- */
    int nlast;
    {
        const int pos = S.fx.unscan4[ks][last], iy = pos >> 2, ix = pos & 3;
        const int cx_ = ks == 2 ? iy : ix, cy_ = ks == 2 ? ix : iy;     /* the vertical scan codes the coordinates swapped */
        const int nx = cx_ < 3 ? cx_ + 1 : 3, ny = cy_ < 3 ? cy_ + 1 : 3;
        nlast = nx + ny;
        const int rr = ii & 3, isy = (ii >> 2) & 1, coord = isy ? cy_ : cx_, cnt = isy ? ny : nx, ctx = (isy ? CX_LASTY : CX_LASTX) + rr;
        const int bin = rr < coord;
        row[(ok && ii < 8 && rr < cnt) ? (isy ? nx : 0) + rr : HV_NXR_DUMMY] = HV_RTOK(S.ftab[ctx], bin, HV_CT(ctx, bin));
    }
    /* ---- sig_coeff_flag of the positions below the last one: the flag's context has seen the flags of its class's earlier
       positions, which are bits of nzm ---- */
    {
        const int info = S.fx.sig4[ks][ii];
        const int cls = info & 15, mm = (info >> 4) & 3, e1 = (info >> 6) & 15, e2 = (info >> 10) & 15;
        const int ok1 = mm >= 1 && e1 < start, ok2 = mm >= 2 && e2 < start;       /* e1 > e2: ok1 implies ok2 whenever there is a second one */
        const int b1 = (nzm >> e1) & 1, b2 = (nzm >> e2) & 1;
        const int k = ok1 + ok2, pre = ok1 ? (b1 | (ok2 ? b2 << 1 : 0)) : (ok2 ? b2 : 0);
        const int bin = (nzm >> ii) & 1;
        row[(ok && ii < start) ? nlast + start - 1 - ii : HV_NXR_DUMMY] = HV_RTOK(S.ftab[HV_FT_SIG + cls * 8 + ((1 << k) | pre)], bin, HV_CT(CX_SIG + cls, bin)) | HV_RTOK_KP(k | (pre << 3));
    }
    /* ---- the non-zero magnitudes in coding order (highest position first): item t of the mode goes to lane t ---- */
    comp[(ok && v != 0) ? __builtin_popcount(nzm >> (ii + 1)) : 16 + ii] = (u16)av;      /* (16 + ii: the row's spare words behind) */
    WAVE_SYNC();
    const int t = ii, at = (ok && t < nz) ? comp[t] : 0;
    const unsigned bigc = (unsigned)(hv_ballot(at > 1) >> s16) & 0xffffu, g2c = (unsigned)(hv_ballot(at > 2) >> s16) & 0xffffu;
    const int n1 = hv_min(nz, 8);
    const int fb = __builtin_ctz((bigc & 0xffu) | 0x100u);       /* the first level above 1 among the first eight; 8: none */
    /* greater1 flags of the first eight: contexts 1, 2, 3, 3, ... up to the first level above 1, 0 after it; so context 3 has seen
       zeros only and context 0 the flags since.  greater2 of that first one  [ref :1217-1236] */
    {
        const int c1 = t <= fb ? hv_min(t + 1, 3) : 0, k0 = hv_max(t - fb - 1, 0);
        const int p0 = (int)(bigc >> ((fb + 1) & 15)) & ((1 << k0) - 1);
        const int at_ = c1 == 0 ? HV_FT_ONE0 + ((1 << k0) | p0) : (c1 == 3 ? HV_FT_ONE3 + (t - 2) : CX_ONE + c1);
        const int bin = at > 1;
        const int kp = c1 == 0 ? (k0 | (p0 << 3)) : (c1 == 3 ? t - 2 : 0);   /* context 3 has seen t - 2 zeros, context 0 the flags since the first level above 1 */
        row[(ok && t < n1) ? nlast + start + t : HV_NXR_DUMMY] = HV_RTOK(S.ftab[at_], bin, HV_CT(CX_ONE + c1, bin)) | HV_RTOK_KP(kp);
        row[(ok && t == fb) ? nlast + start + n1 : HV_NXR_DUMMY] = HV_RTOK(S.ftab[CX_ABS], at > 2, HV_CT(CX_ABS, at > 2));
    }
    const int ntok = nlast + start + n1 + (fb < 8);
    row[(ok && ii < 8) ? ntok + ii : HV_NXR_DUMMY] = HV_RTOK_NOP;
    /* ---- bypass bins: a sign per level, and the remaining magnitudes with the adaptive Rice parameter  [ref :1238-1268].  The
       parameter only moves up, by one, behind an item above 3 << parameter: from 0 behind the first item above 3, from 1 behind the
       first item after that above 6, ... -- four ballots and four find-first-set; the bins are then counted one item per lane ---- */
    const int more = __builtin_popcount(bigc & 0xffu) > 1;
    const int esc = nz > 8 || more || (fb < 8 && ((g2c >> fb) & 1));
    const int base = t < 8 ? ((bigc & ((1u << t) - 1)) ? 2 : 3) : 1;
    const int rem = at - base, on = t < nz && esc && rem >= 0;
    int rice = 0;
    {
        const unsigned e0 = (unsigned)(hv_ballot(on && at > 3) >> s16) & 0xffffu, e1 = (unsigned)(hv_ballot(on && at > 6) >> s16) & 0xffffu;
        const unsigned e2 = (unsigned)(hv_ballot(on && at > 12) >> s16) & 0xffffu, e3 = (unsigned)(hv_ballot(on && at > 24) >> s16) & 0xffffu;
        const int p0 = __builtin_ctz(e0 | 0x10000u);                                    /* 16: never */
        const int p1 = __builtin_ctz((e1 & ~((2u << p0) - 1)) | 0x10000u);
        const int p2 = __builtin_ctz((e2 & ~((2u << p1) - 1)) | 0x10000u);
        const int p3 = __builtin_ctz((e3 & ~((2u << p2) - 1)) | 0x10000u);
        rice = (t > p0) + (t > p1) + (t > p2) + (t > p3);
    }
    const int byp = nz + hv_row_sum(on ? hv_remainder_bins(rem, rice) : 0);
    if (ok && ii == 0) { S.nxn_last[m] = last; S.nxn_sig[m] = sig; S.nxn_magn[m] = ntok | (byp << 8); }
    #undef HV_NXR_DUMMY
}
/* The engine for resolved tokens: n of them at sp (16-byte chunks of four), followed by at least 8 HV_RTOK_NOP.  Per bin the LPS
   range is a byte of the table entry picked by the range quarter (v_perm_b32), the new range `lps` or `range - lps` shifted left
   until bit 8 is set again  [ref :914-933]; the table entries of the next four bins are in flight meanwhile, the tokens of the
   four after those too. */
static HD void hv_rengine_step(HvFCod &c, unsigned lps4, unsigned nx) {
    const int lps = (int)hv_perm(lps4, 0u, (unsigned)c.range >> 6);
    const int pre = (nx >> 8) ? lps : c.range - lps;
    const int sh = hv_clz_nz((unsigned)pre) - 23;
    c.range = pre << sh;
    c.nbits -= sh;
}
template <typename ChunkPtr>
static HD void hv_rengine(const HvLds &S, HvFCod &c, ChunkPtr sp, int n) {
    if (n <= 0) return;
    /* eight bins per turn of the loop, two register sets taking turns (a rotation of one set cost eight moves per four bins); the
       streams end in at least eight no-op tokens, and their rows have room for the look-ahead behind those */
    unsigned al[4], an[4], bl[4], bn[4];
#define HV_RENG_LOAD(L_, N_, CH_) do { const HvChunk ch_ = (CH_); _Pragma("unroll") for (int k = 0; k < 4; k++) { L_[k] = S.ctab2[(ch_[k] >> 16) & 0xffu][0]; N_[k] = S.ctab2[(ch_[k] >> 16) & 0xffu][1]; } } while (0)
    HV_RENG_LOAD(al, an, sp[0]);
    HvChunk c1 = sp[1], c2 = sp[2];
    for (int base = 0; base < n; base += 8) {
        HV_RENG_LOAD(bl, bn, c1);
        c1 = sp[(base >> 2) + 3];
#pragma unroll
        for (int k = 0; k < 4; k++) hv_rengine_step(c, al[k], an[k]);
        HV_RENG_LOAD(al, an, c2);
        c2 = sp[(base >> 2) + 4];
#pragma unroll
        for (int k = 0; k < 4; k++) hv_rengine_step(c, bl[k], bn[k]);
    }
#undef HV_RENG_LOAD
}
/* one pass: the block trials and the binarisation of the four modes 4 * slot .. 4 * slot + 3 (whole wavefront) */
static HD void hv_nxn_pass(HvLds &S, const HvPx4 &K, const HV_LDSP u16 *lut4, int q, int ty, int tx, int slot, int lane) {
    const int m = 4 * slot + (lane >> 4), ok = m < HEVCE_NMODES;
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
    hv_tu4_px(S, K, lut4, q, ty, tx, m, ok, lane);
    WAVE_SYNC();
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
#endif
    hv_tok4_px(S, m, ok, lane);
    WAVE_SYNC();
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    if (slot == 2 && lane == 0) { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); S.prof2[13] += t1_ - t0_; S.prof2[14] += t2_ - t1_; S.prof2[15] += 1; }   /* (one helper's first pass) */
#endif
}
/* The nine passes of a 4x4 block on the chain wavefront and its helpers.  A pass is two UNITS -- the block trials (~1.2 k cycles) and the
   binarisation (~1.8 k) of its four modes -- and a wavefront issues one instruction per ~4.5 cycles whatever it runs, so what counts is
   the longest wavefront.  Four helpers (clusters of four: wavefronts 1..4): two passes each, 6 k.  Six helpers (clusters of five and
   more, where the stream writer has left the workgroup: + wavefronts 6 and 7): one pass each, and the two passes left over are split
   into their units -- two helpers do those block trials FIRST, two others the binarisations behind their own pass: 4.8 k.
   A unit: slot | 0x20 = a whole pass, slot = its block trials (reported in S.nxq_tmask), slot | 0x10 = its binarisation (waits for them). */
static constexpr unsigned char hv_nxq_sched[2][7][3] = {
    { { 0x28, 0xff, 0xff }, { 0x20, 0x21, 0xff }, { 0x22, 0x23, 0xff }, { 0x24, 0x25, 0xff }, { 0x26, 0x27, 0xff }, { 0xff, 0xff, 0xff }, { 0xff, 0xff, 0xff } },
    { { 0x28, 0x17, 0xff }, { 0x20, 0x16, 0xff }, { 0x07, 0x21, 0xff }, { 0x06, 0x22, 0xff }, { 0x23, 0xff, 0xff }, { 0x24, 0xff, 0xff }, { 0x25, 0xff, 0xff } } };
static HD void hv_nxq_units(HvLds &S, const HvPx4 &K, const HV_LDSP u16 *lut4, int q, int ty, int tx, int six, int who, int lane) {   /* who: 0 = the chain wavefront, 1 + h = helper h */
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    const unsigned long long u0_ = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int u = hv_nxq_sched[1][who][k], u4 = hv_nxq_sched[0][who][k];
        const int unit = hv_uni(six) ? u : u4;
        if (unit == 0xff) break;
        const int slot = unit & 15, m = 4 * slot + (lane >> 4), ok = m < HEVCE_NMODES;
        if (unit & 0x20) hv_nxn_pass(S, K, lut4, q, ty, tx, slot, lane);
        else if (!(unit & 0x10)) {
            hv_tu4_px(S, K, lut4, q, ty, tx, m, ok, lane);
            HV_FENCE_RELEASE();
            WAVE_SYNC();
            if (lane == 0) hv_atomic_or(&S.nxq_tmask, 1 << slot);
        } else {
            while (!((hv_poll(&S.nxq_tmask) >> slot) & 1)) hv_spin_short();
            HV_FENCE_ACQUIRE();
            hv_tok4_px(S, m, ok, lane);
            WAVE_SYNC();
        }
    }
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    if (lane == 0) S.prof2[24 + hv_uni(who)] += __builtin_amdgcn_s_memtime() - u0_;   /* this wavefront's units of the block, start to end */
#endif
}

/* a helper wavefront of the NxN chain (cluster of four and more): for each of the CU's four 4x4 blocks, when the chain has laid out the
   reference samples, its units of the block (hv_nxq_units) */
#define HV_NXN_HELPERS 4
static HDN void hv_nxn_helper(int q, HvCu cu, int h, int six) {
    HvLds &S = HV_S;
    q = hv_uni(q); cu.y = hv_uni(cu.y); cu.x = hv_uni(cu.x); h = hv_uni(h); six = hv_uni(six);
    const int lane = hv_tid() & 63;
    HvPx4 K;
    hv_px4_consts(S, K, lane);
    const HV_LDSP u16 *lut4 = hv_lut4_lds(S, 1);
    for (int s = 0; s < 4; s++) {
        while (hv_poll(&S.nxq_go) <= s) hv_spin_short();
        HV_FENCE_ACQUIRE();
        hv_nxq_units(S, K, lut4, q, cu.y + (s >> 1) * 4, cu.x + (s & 1) * 4, six, 1 + h, lane);
        HV_FENCE_RELEASE();
        if (lane == 0) hv_atomic_inc(&S.nxq_done);
    }
}

/* NxN job (8x8 CU only): per 4x4 sub-block choose the mode with a fresh coder and fresh contexts  [ref :1499-1528].
   PX: block trials and binarisation with one lane per sample / scan position on this wavefront and its helpers, resolved tokens (FAST,
   clusters of four and more).  Without helpers (one workgroup per picture, clusters of two and three) the nine passes would all run
   here, 27 k cycles per 4x4 block: there one lane per MODE does a block trial (~10 k), the binarisation (~8 k) and the engine
   with a context look-up per bin (~7 k), as in round 2 -- batches of pictures run that path. */
template <bool FAST, bool PX>
static HDN void hv_job_nxn(int wave, int lane, int q, HvCu cu, HV_GLOBAL u16 *c70, int helpers) {
    HvLds &S = HV_S;
    HvLds::Level &L = S.lv[0];
    q = hv_uni(q); cu.y = hv_uni(cu.y); cu.x = hv_uni(cu.x);
    int total = 0;
    HV_PROF2_T0();
    HvPx4 K;
    if constexpr (PX) hv_px4_consts(S, K, lane);
    const HV_LDSP u16 *lut4 = hv_lut4_lds(S, PX && hv_uni(helpers));
    if (lane == 0) { S.c70_len = 16; S.c70_byp = 0; }
    for (int s = 0; s < 4; s++) {
        int l, lb, a, ar;
        hv_sub_avail(s, cu.L, cu.LB, cu.A, cu.AR, l, lb, a, ar);
        const int oy = (s >> 1) * 4, ox = (s & 1) * 4;
        /* reference samples of this 4x4 (shared by all modes): corner, left + left-below, above + above-right  [ref :1502].  Lane 0 the
           corner, 1..8 the left column downwards, 9..16 the row above: one position each, read from the CU's own samples so far (inside
           the CU) or the picture's; what is not available repeats its neighbour or the corner  [ref :196-257 for a 4x4 block] */
        {
            const HV_LDSP u8 *inner = (const HV_LDSP u8 *)&S.nxn_crec[0][0];
            const int ty = cu.y + oy, tx = cu.x + ox;
            const int side = lane >= 9, i = side ? lane - 9 : lane - 1;          /* (lane 0: i = -1) */
            const int on = side ? a : l, on2 = side ? ar : lb;                   /* this side's first four, its second four */
            int yy, xx, have;
            if (lane == 0) { yy = (l && a) ? ty - 1 : (l ? ty : ty - 1); xx = (l && a) ? tx - 1 : (l ? tx - 1 : tx); have = l || a; }
            else {
                const int k = (i < 4 || on2) ? i : 3;                             /* second four missing: the fourth one again */
                yy = side ? ty - 1 : ty + k; xx = side ? tx + k : tx - 1; have = on;
            }
            const int ry = yy - cu.y, rx = xx - cu.x, in = ry >= 0 && ry < 8 && rx >= 0 && rx < 8;
            const int vi = inner[hv_min(hv_max(ry, 0), 7) * 8 + hv_min(hv_max(rx, 0), 7)], vo = S.rec[yy + 1][xx + HV_RX];
            int v = have ? (in ? vi : vo) : 128;
            const int corner = hv_readlane(v, 0);
            v = have ? v : corner;
            if (lane < 17) S.nxn_bord[lane] = (u8)v;
        }
        WAVE_SYNC();
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
#define HV_NXQ(i) do { if (lane == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[16 + (i)] += n_ - nq_; nq_ = n_; } } while (0)
        unsigned long long nq_ = p2_;
        HV_NXQ(0);
#else
#define HV_NXQ(i) do { } while (0)
#endif
        if constexpr (PX) {
            /* block trials and binarisation with one lane per sample, four modes per pass (hv_nxn_pass): nine passes, spread over this
               wavefront and its helpers (hv_nxq_units) */
            if (hv_uni(helpers)) {
                const int nh = hv_uni(helpers);                  /* 4 or 6 helper wavefronts */
                if (lane == 0) hv_flag_clear(&S.nxq_tmask);      /* (the helpers are through with the block before: their done count) */
                HV_FENCE_RELEASE();
                WAVE_SYNC();
                if (lane == 0) hv_flag_set(&S.nxq_go, s + 1);
                hv_nxq_units(S, K, lut4, q, cu.y + oy, cu.x + ox, nh == 6, 0, lane);
                HV_NXQ(1);
                while (hv_poll(&S.nxq_done) < nh * (s + 1)) hv_spin_short();
                HV_FENCE_ACQUIRE();
                HV_NXQ(2);
            } else for (int slot = 0; slot < 9; slot++) hv_nxn_pass(S, K, lut4, q, cu.y + oy, cu.x + ox, slot, lane);
        } else if (lane < HEVCE_NMODES) {
            int last;
            const int sse = hv_tu4_lane(S, q, S.nxn_bord, S.nxn_line[lane], cu.y + oy, cu.x + ox, lane, (HV_LDSP short *)S.nxn_lev[lane], (HV_LDSP u8 *)S.nxn_rec[lane], last);
            S.nxn_sse[lane] = sse; S.nxn_last[lane] = last < 0 ? 0 : last; S.nxn_sig[lane] = last >= 0;
        }
        WAVE_SYNC();
        HV_PROF2(S, 0);
        /* rate of every candidate: lane = mode  [ref :1505-1519]; residual_coding is written even for an all-zero block */
        int cost = HV_I32MAX, idx = -1;
        if (lane < HEVCE_NMODES) {
            int bits, ntok, nbyp;
            if constexpr (PX) {
                /* the mode's resolved tokens are there (hv_tok4_px): the engine has no context to look up or to write back */
                HV_PROF2(S, 1);
                if (s == 3 && lane == 0) hv_flag_clear(&S.nxn_busy);     /* (see below) */
                const int mg = S.nxn_magn[lane];
                ntok = mg & 0xff; nbyp = mg >> 8;
                HvFCod c = { 510, 23 };
                hv_rengine(S, c, (const HV_LDSP HvChunk *)((HV_LDSP unsigned *)S.nxn_tok[lane] + HV_NXR_TOK), ntok);
                bits = 23 - c.nbits + nbyp;
            } else {
                u8 *cx = &S.nxn_ctx[0][lane];                      /* the chain's own context columns (same layout as a level's) */
                for (int k = 0; k < 3; k++) { *HV_CX(CX_LASTX + k) = S.ctx_fresh[CX_LASTX + k]; *HV_CX(CX_LASTY + k) = S.ctx_fresh[CX_LASTY + k]; }
                for (int k = 0; k < 9; k++) *HV_CX(CX_SIG + k) = S.ctx_fresh[CX_SIG + k];
                for (int k = 0; k < 4; k++) *HV_CX(CX_ONE + k) = S.ctx_fresh[CX_ONE + k];
                *HV_CX(CX_ABS) = S.ctx_fresh[CX_ABS];
                *HV_CX(HV_CT_PAD) = HV_PADCV;                     /* pad context used by bypass / no-op tokens */
                HV_LDSP u16 *tk = (HV_LDSP u16 *)S.nxn_tok[lane];
                const int ks = hv_scan_kind(4, lane);
                /* the two slots are laid out so that their tokens are contiguous from tk[8]; one group: no set memory */
                HvTokW<HV_LDSP u16 *, FAST> wa = { tk, 0, 0 };
                const int na = hv_tokenize_tu<3>(S, wa, 4, 2, ks, 0, 0, 1, S.nxn_last[lane]);
                HvTokW<HV_LDSP u16 *, FAST> wb = { tk + na, 0, 0 };
                HvCg g;
                hv_cg_load(S, g, (const HV_LDSP short *)S.nxn_lev[lane], 4, 2, ks, 0, S.nxn_last[lane], (u64)(S.nxn_sig[lane] ? 1 : 0), 1);
                const int nb = hv_cg_emit(S, wb, g, 0);
#pragma unroll
                for (int i = 0; i < 8; i++) tk[HV_SLOT_TOK0 + na + nb + i] = (u16)(FAST ? HV_TOK_NOPF : HV_TOK_NOP);
                HV_PROF2(S, 1);
                /* last block, only its engine to go: tell the candidates riding in the other wavefronts to wind up, so that they are
                   at the workgroup barrier when this wavefront gets there */
                if (s == 3 && lane == 0) hv_flag_clear(&S.nxn_busy);
                if constexpr (FAST) {                            /* range-only engine with a context look-up per bin; the bypass bins were counted by the tokenisers */
                    HvFCod c = { 510, 23 };
                    hv_fengine(S, c, cx, (const HV_LDSP HvChunk *)(tk + HV_SLOT_TOK0), na + nb);
                    nbyp = wa.byp + wb.byp;
                    bits = 23 - c.nbits + nbyp;
                } else {
                    HvCod c;
                    hv_cod_fresh(c);
                    hv_engine(S, c, cx, (const HV_LDSP HvChunk *)(tk + HV_SLOT_TOK0), na + nb, (HV_LDSP u16 *)S.nxn_sink[lane]);
                    hv_cod_replay(c, (const HV_LDSP HvChunk *)S.nxn_sink[lane]);
                    bits = hv_cod_bits(c);
                    nbyp = 0;
                }
                ntok = na + nb;
            }
            S.nxn_ntok[lane] = ntok | (nbyp << 8) | (bits << 20);
            cost = hv_rdcost(q, S.nxn_sse[lane], bits);
            idx = lane;
        }
        HV_PROF2(S, 2);
        /* minimum cost, the LATER mode wins ties  [ref :1521]: inside the rows of sixteen by rotations (no LDS), then over the four rows */
#define HV_ARGMIN_STEP(OC, OI) do { const int oc_ = (OC), oi_ = (OI); const bool tk_ = oc_ < cost || (oc_ == cost && oi_ > idx); cost = tk_ ? oc_ : cost; idx = tk_ ? oi_ : idx; } while (0)
        HV_ARGMIN_STEP(hv_row_get<8>(cost), hv_row_get<8>(idx));
        HV_ARGMIN_STEP(hv_row_get<4>(cost), hv_row_get<4>(idx));
        HV_ARGMIN_STEP(hv_row_get<2>(cost), hv_row_get<2>(idx));
        HV_ARGMIN_STEP(hv_row_get<1>(cost), hv_row_get<1>(idx));
        {
            const int c0 = hv_readlane(cost, 0), i0 = hv_readlane(idx, 0), c1 = hv_readlane(cost, 16), i1 = hv_readlane(idx, 16), c2 = hv_readlane(cost, 32), i2 = hv_readlane(idx, 32);   /* (v_readlane: no trip through the LDS crossbar, and what follows is scalar) */
            cost = c0; idx = i0;
            HV_ARGMIN_STEP(c1, i1);
            HV_ARGMIN_STEP(c2, i2);                              /* (35 modes: the fourth row is empty) */
        }
#undef HV_ARGMIN_STEP
        const int w = idx;
        if (lane < 16) {
            S.nxn_crec[oy + (lane >> 2)][ox + (lane & 3)] = S.nxn_rec[w][lane];
            S.nxn_bestlev[s][lane] = S.nxn_lev[w][lane];
        }
        if (lane == 0) {
            S.nxn_modes[s] = w;
            L.tinfo[70][s].last = S.nxn_last[w]; L.tinfo[70][s].cbf = S.nxn_sig[w]; L.tinfo[70][s].cgsig = (u64)(S.nxn_sig[w] ? 1 : 0);
        }
        total += S.nxn_sse[w];
        /* candidate 70 = the four winners: its token stream in HBM is [CU header, 16 tokens reserved][per block: cbf_luma,
           then -- if the block has levels -- exactly the winner's tokens of above], contiguous, no slot padding */
        {
            const int pk = S.nxn_ntok[w];
            const int off = S.c70_len, cbf = S.nxn_sig[w], n = cbf ? (pk & 0xff) : 0;
            if (lane == 0) c70[off] = (u16)HV_CT(CX_YCBF, cbf);
            if constexpr (PX) {
                /* the plain halves of the winner's resolved tokens (<= 32): candidate 70 meets them in another context state.  With a rider
                   (cluster of four) they go to its wavefront's scratch tile, two blocks alternating: no round trip through the L2 */
                const HV_LDSP unsigned *src = (const HV_LDSP unsigned *)S.nxn_tok[w] + HV_NXR_TOK;
                if (hv_uni(helpers)) {
                    while (hv_poll(&S.c70_rd) < s - 1) hv_spin_short();      /* the rider is through with the block before the last one (same buffer) */
                    HV_LDSP unsigned *dst = (HV_LDSP unsigned *)&S.ws[5] + (s & 1) * 64;
                    if (lane < 48) dst[lane] = lane < n ? src[hv_min(lane, 39)] : (unsigned)HV_TOK_NOPF;   /* (whole words: the rider needs each bin's predecessors, HV_RTOK_KP) */
                } else if (lane < n) c70[off + 1 + lane] = (u16)src[lane];
            } else {
                const HV_LDSP u16 *src = (const HV_LDSP u16 *)S.nxn_tok[w] + HV_SLOT_TOK0;
                if (lane < n) c70[off + 1 + lane] = src[lane];
                if (lane + 64 < n) c70[off + 65 + lane] = src[lane + 64];
            }
            if constexpr (FAST) {
                /* kept for hv_verify_nxn: the winner's levels, what the tokeniser was given, the bit count it was chosen with */
                const int blk = (((cu.y >> 3) * 4 + (cu.x >> 3)) << 2) | s;
                if (lane < 16) S.nxv_lev[blk][lane] = S.nxn_lev[w][lane];
                if (lane == 0) { S.nxv_info[blk] = (u16)(S.nxn_last[w] | (hv_scan_kind(4, w) << 4) | (cbf << 6)); S.nxv_bits[blk] = (short)(pk >> 20); }
            }
            WAVE_SYNC();
            if (lane == 0) { S.c70_len = off + 1 + n; S.c70_byp += cbf ? ((pk >> 8) & 0xfff) : 0; }
            if constexpr (FAST) {
                /* candidate 70's rider (hv_c70_rider, another wavefront) may code this block now */
                HV_FENCE_RELEASE();
                if (lane == 0) { S.c70_seg[s] = off | (n << 16) | (cbf << 31); hv_flag_set(&S.c70_done, s + 1); }
            }
        }
        WAVE_SYNC();
        HV_PROF2(S, 3);
    }
    if (lane == 0) {
        L.tsse[70] = total;
        /* CU header of candidate 70 (part_mode NxN, the four modes)  [ref :1531-1538], padded to its 16 reserved tokens; the
           stream padded to whole chunks; one list entry covering it all (it runs on across the candidate's slot area) */
        const int my = 1 + (cu.y >> 2), mx = 1 + (cu.x >> 2);
        HvModes pm;
        for (int s4 = 0; s4 < 4; s4++) pm.mode[s4] = S.nxn_modes[s4];
        pm.left[0] = L.ml; pm.above[0] = L.ma;
        pm.left[1] = pm.mode[0];  pm.above[1] = S.map_pm[my - 1][mx + 1];
        pm.left[2] = S.map_pm[my + 1][mx - 1];  pm.above[2] = pm.mode[0];
        pm.left[3] = pm.mode[2];  pm.above[3] = pm.mode[1];
        HvTokW<HV_GLOBAL u16 *, FAST> wh = { c70 - HV_SLOT_TOK0, 0, 0 };
        const int nh = hv_tokenize_cu(wh, 8, 3, 2, L.split_inc, pm);
        for (int k = nh; k < 16; k++) c70[k] = (u16)(FAST ? HV_TOK_NOPF : HV_TOK_NOP);
        const int len = S.c70_len;
        for (int k = len; k < ((len + 7) & ~7); k++) c70[k] = (u16)(FAST ? HV_TOK_NOPF : HV_TOK_NOP);
        S.slist0[70][0] = (u16)(((len + 7) >> 3) << 8);          /* slot 0, that many chunks */
        L.tlen[70] = 1; L.tpos[70] = 0; L.tnlog[70] = FAST ? S.c70_byp + wh.byp : 0;   /* FAST: tnlog = the candidate's bypass bins */
        L.tregs[70] = S.ent[0];
        if constexpr (FAST) {                                    /* which of the four luma modes are most-probable-mode hits: the rider's lane  [ref :985-1018] */
            int pat = 0;
            for (int i = 0; i < 4; i++) { int p3[3]; hv_mpm(pm.left[i], pm.above[i], p3); pat |= (p3[0] == pm.mode[i] || p3[1] == pm.mode[i] || p3[2] == pm.mode[i]) << i; }
            S.c70_pat = pat;
        }
    }
    for (int r = lane; r < HEVCE_NCTXP; r += 64) L.tctx[r][70] = S.ent_ctx[0][r];
    if constexpr (FAST) { HV_FENCE_RELEASE(); WAVE_SYNC(); if (lane == 0) hv_flag_set(&S.c70_done, 5); }
    HV_PROF2(S, 4);
}
/* distortion of the CU as currently reconstructed (the "split into four CUs" alternative)  [ref :1409] */
static HD void hv_job_sse(HvLds &S, int lane, const HvCu &cu) {
    int sse = 0;
    for (int p = lane; p < cu.n * cu.n; p += 64) {
        const int yy = cu.y + p / cu.n, xx = cu.x + p % cu.n;
        const int d = (int)S.org[yy][xx] - S.rec[yy + 1][xx + HV_RX];
        sse += d * d;
    }
    for (int s = 1; s < 64; s <<= 1) sse += hv_shfl_xor(sse, s);
    if (lane == 0) S.split_sse = sse;
}

/* one context-coded bin on the committed coder (split_cu_flag = 1 of a parent)  [ref :1404] */
static HD void hv_commit_bin(HvLds &S, int ctx, int bin) {
    u8 tmp = S.com_ctx[ctx];
    if (S.exact) {
        HvCod c;
        hv_cod_load(c, S.com);
        hv_bin(S, c, &tmp, bin);
        hv_cod_store(S.com, c);
    } else {
        HvFCod c = { S.com.range, S.com.nbits };
        hv_fbin(S, c, &tmp, bin);
        S.com.range = c.range; S.com.nbits = c.nbits;
    }
    S.com_ctx[ctx] = tmp;
}
/* end_of_slice_segment_flag on the committed coder  [ref :1630] */
static HD void hv_commit_terminate(HvLds &S, int bin) {
    if (S.exact) {
        HvCod c;
        hv_cod_load(c, S.com);
        hv_terminate(c, bin);
        hv_cod_store(S.com, c);
    } else {
        HvFCod c = { S.com.range, S.com.nbits };
        hv_fterminate(c, bin);
        S.com.range = c.range; S.com.nbits = c.nbits;
    }
}

/* The winners of the CTU's 64 NxN sub-searches once more, each through an EXACT fresh coder  [ref :1505-1519]: lane = 4x4
   block.  The binarisation is the tokenisers' -- with a "writer" that sends every bin straight through the coder -- so the
   bit count can only differ from the FAST engine's by the emulation-prevention bytes; any difference poisons the CTU.
   One wavefront, at the end of a FAST CTU (the chain's context columns are free then). */
struct HvTokExec { const HvLds *S; HvCod c; u8 *cx; int n; enum { fast = 0 }; };
static HD void hv_emit(HvTokExec &w, int tok, int on) {
    if (on) {
        if (tok & HV_TOK_BYP) hv_bypass(w.c, tok & 0xff, (tok >> 8) & 15);
        else hv_bin(*w.S, w.c, w.cx + (tok & 0x1fff), (tok >> 14) & 1);
    }
}
static HD void hv_tok_checkpoint(HvTokExec &) { }
static HD int  hv_tok_finish(HvTokExec &) { return 0; }
struct HvNxv { short lev[64][16]; u16 info[64]; short bits[64]; };   /* a copy of the nxv_* arrays (deferred verification) */
static HD HV_LDSP HvNxv *hv_nxv_copy(HvLds &S) {                 /* lives in slist2: the 32x32 level's slot lists are unused in cluster mode */
    return (HV_LDSP HvNxv *)((HV_LDSP u8 *)&S.slist2[0][0] + ((16 - ((unsigned)(unsigned long long)&S.slist2[0][0] & 15)) & 15));
}
static HDN void hv_verify_nxn(int lane, int deferred) {
    HvLds &S = HV_S;
    deferred = hv_uni(deferred);
    /* deferred (cluster of four, a spare wavefront during the NEXT CTU's first round): the copy of the winners, and context
       columns of the 16x16 level, which is another workgroup's business in cluster mode */
    HV_LDSP HvNxv *V = hv_nxv_copy(S);
    u8 *cx = deferred ? &S.lv[1].tctx[0][lane] : &S.nxn_ctx[0][lane];
    for (int k = 0; k < 3; k++) { *HV_CX(CX_LASTX + k) = S.ctx_fresh[CX_LASTX + k]; *HV_CX(CX_LASTY + k) = S.ctx_fresh[CX_LASTY + k]; }
    for (int k = 0; k < 9; k++) *HV_CX(CX_SIG + k) = S.ctx_fresh[CX_SIG + k];
    for (int k = 0; k < 4; k++) *HV_CX(CX_ONE + k) = S.ctx_fresh[CX_ONE + k];
    *HV_CX(CX_ABS) = S.ctx_fresh[CX_ABS];
    const int info = deferred ? V->info[lane] : S.nxv_info[lane], last = info & 15, ks = (info >> 4) & 3, sig = (info >> 6) & 1;
    const int want = deferred ? V->bits[lane] : S.nxv_bits[lane];
    HvTokExec w;
    w.S = &S; w.cx = cx; w.n = 0;
    hv_cod_fresh(w.c);
    hv_tokenize_tu<3>(S, w, 4, 2, ks, 0, 0, 1, last);
    HvCg g;
    hv_cg_load(S, g, deferred ? (const HV_LDSP short *)V->lev[lane] : (const HV_LDSP short *)S.nxv_lev[lane], 4, 2, ks, 0, last, (u64)sig, 1);
    hv_cg_emit(S, w, g, 0);
    if (hv_cod_bits(w.c) != want) { if (deferred) S.poison_prev = 1; else S.poison = 1; }
}
