/* hevce_dev_tu.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   quantiser and TU jobs: the block trials of 64/N candidate modes of one NxN transform block per wavefront. */
/* --------------------------------------------------------------------------------------------- */
/* quantiser  [ref HEVCe.c:526-595]: the level of a coefficient is looked up                                                   */
/* --------------------------------------------------------------------------------------------- */
#include "hevce_quant.h"
/* The reference tries level, level-1, level-2 and keeps the cheapest in w1 * distortion + w2 * rate  [ref :566-580]; for a
   fixed qpd6 and transform size that choice depends on the coefficient's magnitude alone, so the host tabulates
   hv_quant1() once (im.qlut: four tables of HEVCE_QLUT_N 16-bit levels for the picture's qpd6, 256 KB, L2-resident) and the
   block jobs gather from it: one load per coefficient instead of ~100 VALU instructions.  |coefficient| <= 32640 for 8-bit
   samples (the bound the 16-bit staging of the transforms already relies on); the index is clamped all the same.
   d = the scaled magnitude for the group zero-out test  [ref :555, 585-589]. */
typedef const HV_GLOBAL u16 *HvQlut;
template <int LG> static HD HvQlut hv_qlut(const HvLds &S) { return (HvQlut)S.qlut + (LG - 2) * HEVCE_QLUT_N; }
static HD int hv_quant_lut(HvQlut lut, int av, int &d) {
    d = av << 14;
    return lut[hv_min(av, HEVCE_QLUT_N - 1)];
}

/* --------------------------------------------------------------------------------------------- */
/* TU job: 64/N candidate modes of one NxN transform block per wavefront                            */
/*   lane = slot*N + j ; j is the COLUMN it owns in column passes and the ROW it owns in row passes  */
/* --------------------------------------------------------------------------------------------- */
template <int N> struct HvTm;
template <> struct HvTm<4>  { static HD const int *f() { return hv_tm.f4;  } static HD const int *t() { return hv_tm.t4;  } enum { LG = 2 }; };
template <> struct HvTm<8>  { static HD const int *f() { return hv_tm.f8;  } static HD const int *t() { return hv_tm.t8;  } enum { LG = 3 }; };
template <> struct HvTm<16> { static HD const int *f() { return hv_tm.f16; } static HD const int *t() { return hv_tm.t16; } enum { LG = 4 }; };
template <> struct HvTm<32> { static HD const int *f() { return hv_tm.f32; } static HD const int *t() { return hv_tm.t32; } enum { LG = 5 }; };

struct HvTuOut { int sse, last, cbf; u64 cgsig; };          /* block-wide values, identical in all lanes of a slot */

/* The 8/16/32-point core transforms as partial butterflies.  Rows of even frequency are symmetric, rows of odd frequency
   antisymmetric, and the even rows of the N-point matrix are the N/2-point matrix (hv_basis): N*N/3 multiplications
   instead of N*N, with exactly the reference's integer sums (it multiplies the whole matrix out, HEVCe.c:469-516; integer
   addition is associative, nothing here can overflow: |inputs| < 2^16, |coefficients| <= 90).
   x[0..M) = the M inputs (forward) / the inputs of frequency 0, S, 2S, ... (inverse, S = N/M); all indices compile-time. */
template <int N, int M> struct HvBfly {
    enum { S = N / M, H = M / 2 };
    /* forward: out[f * S] for f = 0..M-1 from x (destroyed) */
    static HD void fwd(int *x, int *out) {
        int o[H];
#pragma unroll
        for (int k = 0; k < H; k++) { const int a = x[k], b = x[M - 1 - k]; x[k] = a + b; o[k] = a - b; }
#pragma unroll
        for (int f = 1; f < M; f += 2) {
            int acc = 0;
#pragma unroll
            for (int k = 0; k < H; k++) acc += hv_mul24(hv_basis(f * S * (32 / N), k * 1), o[k]);
            out[f * S] = acc;
        }
        HvBfly<N, H>::fwd(x, out);
    }
    /* inverse: out[0..M) (sample positions of the M-point transform) from the inputs in[f * S], f = 0..M-1 */
    static HD void inv(const int *in, int *out) {
        int e[H];
        HvBfly<N, H>::inv(in, e);
#pragma unroll
        for (int n = 0; n < H; n++) {
            int acc = 0;
#pragma unroll
            for (int f = 1; f < M; f += 2) acc += hv_mul24(hv_basis(f * S * (32 / N), n), in[f * S]);
            out[n] = e[n] + acc; out[M - 1 - n] = e[n] - acc;
        }
    }
};
template <int N> struct HvBfly<N, 1> {
    static HD void fwd(int *x, int *out) { out[0] = 64 * x[0]; }
    static HD void inv(const int *in, int *out) { out[0] = 64 * in[0]; }
};

/* reconstructed sample at CTU-relative (yy,xx).  Inside the CU under trial it comes from the candidate's own partial
   reconstruction: `inner` is either the full cu_n x cu_n block (NxN sub-search) or, when `edges` is set, the chain's edge
   store [4][H]: right columns of sub-blocks 0 and 2, bottom rows of sub-blocks 0 and 1 -- the only inner samples a later
   sub-block ever reads (they all lie on row H-1 or column H-1 of the CU). */
static HD int hv_px(const HvLds &S, const HV_LDSP u8 *inner, int edges, int cu_y, int cu_x, int cu_n, int yy, int xx) {
    if (inner && yy >= cu_y && yy < cu_y + cu_n && xx >= cu_x && xx < cu_x + cu_n) {
        const int ry = yy - cu_y, rx = xx - cu_x;
        if (!edges) return inner[ry * cu_n + rx];
        const int h = cu_n >> 1;
        if (rx == h - 1) return inner[(ry < h ? 0 : 1) * h + (ry & (h - 1))];          /* right column of sub-block 0 / 2 */
        return inner[(2 + (rx < h ? 0 : 1)) * h + (rx & (h - 1))];                       /* bottom row of sub-block 0 / 1   */
    }
    return S.rec[yy + 1][xx + HV_RX];
}
/* which (size, mode) pairs read the smoothed reference samples  [ref HEVCe.c:274-280] */
static HD int hv_smoothed(int n, int m) {
    if (n == 4 || m == 1) return 0;
    if (m == 0) return 1;
    const int d = hv_min(hv_abs(m - 10), hv_abs(m - 26));
    return n == 8 ? d > 7 : n == 16 ? d > 1 : d > 0;
}

#ifdef HEVCE_PROF_TU
#define HV_TUT(ph) do { if (N == HEVCE_PROF_TU && lane == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); HV_S.prof2[16 + (ph)] += n_ - tt_; tt_ = n_; } } while (0)
#define HV_TUT0() unsigned long long tt_ = __builtin_amdgcn_s_memtime()
#else
#define HV_TUT(ph) do { } while (0)
#define HV_TUT0() do { } while (0)
#endif
template <int N, typename LevP, typename RecP>
static HD HvTuOut hv_tu_job(int wave, int lane, int q,
                            int ty, int tx,                         /* block origin, CTU-relative            */
                            int cu_y, int cu_x, int cu_n, const HV_LDSP u8 *inner, int edges,
                            int aL, int aLB, int aA, int aAR,       /* neighbour availability                */
                            int mode,                               /* this slot's mode, or -1 (idle slot)   */
                            LevP lev_dst,                           /* N*N levels in scan order (per slot)   */
                            RecP rec_dst, int rec_stride,           /* reconstruction (per slot)             */
                            HV_LDSP u8 *ecol = nullptr, HV_LDSP u8 *erow = nullptr) {   /* chains: this block's right column / bottom row */
    HvLds &S = HV_S;
    HvWaveScr &W = S.ws[hv_uni(wave)];
    q = hv_uni(q); ty = hv_uni(ty); tx = hv_uni(tx); cu_y = hv_uni(cu_y); cu_x = hv_uni(cu_x); cu_n = hv_uni(cu_n);
    aL = hv_uni(aL); aLB = hv_uni(aLB); aA = hv_uni(aA); aAR = hv_uni(aAR);
    constexpr int LG = HvTm<N>::LG, RS = N + 2, BS = 4 * N + 4;
    const int slot = lane / N, j = lane % N;
    const int m = mode < 0 ? 1 : mode;
    short *scr = W.scr + slot * N * RS;
    u8 *pred = W.pred + slot * N * N;
    u8 *braw = W.bord + slot * 2 * BS, *bsm = braw + BS;         /* [0] corner, [1..2N] left, [1+2N..4N] above */

    HV_TUT0();
    /* ---- reference samples  [ref HEVCe.c:196-257] ---- */
    {
        int corner;
        if (aL && aA)  corner = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx - 1);
        else if (aL)   corner = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty, tx - 1);
        else if (aA)   corner = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx);
        else           corner = 128;
        const int l0 = aL ? hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty + j, tx - 1) : corner;
        const int a0 = aA ? hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx + j) : corner;
        int l1, a1;
        if (aLB) l1 = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty + N + j, tx - 1);
        else     l1 = aL ? hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty + N - 1, tx - 1) : corner;
        if (aAR) a1 = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx + N + j);
        else     a1 = aA ? hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx + N - 1) : corner;
        if (j == 0) braw[0] = (u8)corner;
        braw[1 + j] = (u8)l0; braw[1 + N + j] = (u8)l1;
        braw[1 + 2 * N + j] = (u8)a0; braw[1 + 3 * N + j] = (u8)a1;
    }
    WAVE_SYNC();
    if constexpr (N > 4) {                                       /* 4x4 blocks never read the smoothed samples (hv_smoothed) */
        const u8 *l = braw + 1, *a = braw + 1 + 2 * N;
        const int c = braw[0];
        if (j == 0) {
            bsm[0] = (u8)((2 + l[0] + a[0] + 2 * c) >> 2);
            bsm[1] = (u8)((2 + 2 * l[0] + l[1] + c) >> 2);
            bsm[1 + 2 * N] = (u8)((2 + 2 * a[0] + a[1] + c) >> 2);
        } else {
            bsm[1 + j] = (u8)((2 + 2 * l[j] + l[j - 1] + l[j + 1]) >> 2);
            bsm[1 + 2 * N + j] = (u8)((2 + 2 * a[j] + a[j - 1] + a[j + 1]) >> 2);
        }
        const int k = N + j;
        if (j == N - 1) { bsm[1 + k] = l[k]; bsm[1 + 2 * N + k] = a[k]; }
        else {
            bsm[1 + k] = (u8)((2 + 2 * l[k] + l[k - 1] + l[k + 1]) >> 2);
            bsm[1 + 2 * N + k] = (u8)((2 + 2 * a[k] + a[k - 1] + a[k + 1]) >> 2);
        }
    }
    /* DC value of every slot (wave-uniform code: shuffles must not sit in the per-mode branches) */
    int dcsum = braw[1 + j] + braw[1 + 2 * N + j];
    for (int s = 1; s < N; s <<= 1) dcsum += hv_shfl_xor(dcsum, s);
    WAVE_SYNC();
    /* angular modes: the slot's reference line r(k), k = -N..2N, laid out once by its N lanes (r(k>0) main side, r(0) corner,
       r(k<0) the side samples projected through the inverse angle [ref :353-364]); the pixel loop then only indexes it.
       Every lane takes part (the other mode forms just do not read it): no wave-level step inside a per-mode branch */
    u8 *line = W.line + slot * (3 * N + 2) + N;
    {
        const u8 *b = hv_smoothed(N, m) ? bsm : braw;
        const bool horiz = m < 18;
        const int inv = S.invang[m];
        const u8 *mainr = horiz ? b + 1 : b + 1 + 2 * N, *sider = horiz ? b + 1 + 2 * N : b + 1;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const int k = e * N + j - N + 1;                     /* -N+1 .. 2N */
            int v;
            if (k > 0) v = mainr[k - 1];
            else if (k == 0) v = b[0];
            else v = sider[hv_min(hv_max(((128 - inv * k) >> 8) - 1, 0), 2 * N - 1)];   /* entries outside the projected range are never read */
            line[k] = (u8)v;
        }
    }
    WAVE_SYNC();

    HV_TUT(0);
    /* ---- prediction of column j  [ref HEVCe.c:262-381]; residual goes to scr, prediction to pred ---- */
    {
        const u8 *b = hv_smoothed(N, m) ? bsm : braw;
        const u8 *l = b + 1, *a = b + 1 + 2 * N;
        const int c = b[0];
        const bool edge = N <= 16;
        const int ox = tx + j;
        if constexpr (N == 4) {
            /* 4x4 (the TU-split chains of an 8x8 CU: sixteen different modes in a wavefront): every form for every sample, the mode's
               own selected -- a branch per form would have the wavefront walk planar, DC, the two filtered pure directions and the
               angular form one after the other (measured: the group with modes 0..15 took 4.5 k cycles longer per CU) */
            const bool horiz = m < 18, planar = m == 0, isdc = m == 1, hor = m == 10, ver = m == 26;
            const int ang = S.ang[m];
            const int aj = a[j], lN = l[N], aN = a[N], dc = (dcsum + N) >> (LG + 1);
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int li = l[i];
                const int step = horiz ? j : i, along = horiz ? i : j;
                const int off = ang * (step + 1), oi = off >> 5, of = off & 31;
                const int k1 = oi + along + 1, k2 = hv_min(k1 + 1, 2 * N);
#ifdef HEVCE_EMU
                if (k1 < -N + 1 || k1 > 2 * N) { fprintf(stderr, "emu: tu4 line index %d (mode %d ang %d step %d along %d)\n", k1, m, ang, step, along); abort(); }
#endif
                int v = ((32 - of) * line[k1] + of * line[k2] + 16) >> 5;
                const int pl = (N + (N - 1 - j) * li + (j + 1) * aN + (N - 1 - i) * aj + (i + 1) * lN) >> (LG + 1);
                const int dv = (i == 0 && j == 0) ? (2 + 2 * dc + l[0] + a[0]) >> 2 : (i == 0) ? (2 + 3 * dc + aj) >> 2 : (j == 0) ? (2 + 3 * dc + li) >> 2 : dc;
                v = hor ? ((i == 0) ? hv_clip8(li + ((aj - c) >> 1)) : li) : v;
                v = ver ? ((j == 0) ? hv_clip8(aj + ((li - c) >> 1)) : aj) : v;
                const int p = planar ? pl : (isdc ? dv : v);
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else if (m == 0) {
            const int aj = a[j], lN = l[N], aN = a[N];
            for (int i = 0; i < N; i++) {
                const int p = (N + (N - 1 - j) * l[i] + (j + 1) * aN + (N - 1 - i) * aj + (i + 1) * lN) >> (LG + 1);
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else if (m == 1) {
            const int dc = (dcsum + N) >> (LG + 1);
            for (int i = 0; i < N; i++) {
                int p = dc;
                if (edge) {
                    if (i == 0 && j == 0) p = (2 + 2 * dc + l[0] + a[0]) >> 2;
                    else if (i == 0)      p = (2 + 3 * dc + a[j]) >> 2;
                    else if (j == 0)      p = (2 + 3 * dc + l[i]) >> 2;
                }
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else if (m == 10) {
            for (int i = 0; i < N; i++) {
                int p = l[i];
                if (edge && i == 0) p = hv_clip8(p + ((a[j] - c) >> 1));
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else if (m == 26) {
            for (int i = 0; i < N; i++) {
                int p = a[j];
                if (edge && j == 0) p = hv_clip8(p + ((l[i] - c) >> 1));
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else {
            /* angular: two neighbours on the mode's reference line, 1/32-sample weights  [ref :342-380] */
            const bool horiz = m < 18;
            const int ang = S.ang[m];
            for (int i = 0; i < N; i++) {
                const int step = horiz ? j : i, along = horiz ? i : j;
                const int off = ang * (step + 1), oi = off >> 5, of = off & 31;
                const int k1 = oi + along + 1, k2 = hv_min(k1 + 1, 2 * N);   /* weight 0 at 2N+1 (modes 2 / 34): any in-range sample */
                const int p = ((32 - of) * line[k1] + of * line[k2] + 16) >> 5;
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        }
    }

    HV_TUT(1);
    /* ---- forward transform, columns then rows  [ref HEVCe.c:497-516] ---- */
    int v[N];
    {
        constexpr int s1 = LG - 1;
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = scr[k * RS + j];
        if constexpr (N >= 8) {
            int out[N];
            HvBfly<N, N>::fwd(v, out);
#pragma unroll
            for (int i = 0; i < N; i++) scr[i * RS + j] = (short)((out[i] + (1 << s1 >> 1)) >> s1);
        } else {
            const int *tab = HvTm<N>::f();
            for (int i = 0; i < N; i++) {
                int acc = 1 << s1 >> 1;
#pragma unroll
                for (int k = 0; k < N; k++) acc += hv_mul24(tab[i * N + k], v[k]);
                scr[i * RS + j] = (short)(acc >> s1);
            }
        }
    }
    WAVE_SYNC();
    HV_TUT(2);
    /* ---- row pass + quantiser + dequantiser on row j  [ref HEVCe.c:540-615] ---- */
    const int kind = hv_scan_kind(N, m);
    int last = -1, nzmask_lo = 0, nzmask_hi = 0;
    {
        constexpr int s2 = LG - 1 + 7;
        const int sh = 21 - LG + q, thr = 9 << sh >> 2, qsh = 7 - LG + q;
        const HvQlut lut = hv_qlut<LG>(S);
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = scr[j * RS + k];
        const int *tab = HvTm<N>::f();
        if constexpr (N >= 8) {
            /* all N coefficients of the row at once; they wait in the row's own scratch line (16 bits: |coefficient| <= 32640
               for 8-bit samples, the bound the first pass already relies on) until their 4x4 group's turn; their levels are
               requested from the table right away (N gathers in flight), v[] keeps the magnitudes */
            int out[N];
            HvBfly<N, N>::fwd(v, out);
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int coef = (out[i] + (1 << s2 >> 1)) >> s2;
                scr[j * RS + i] = (short)coef;
                int d;
                v[i] = hv_quant_lut(lut, hv_abs(coef), d);
            }
        }
        for (int g = 0; g < N / 4; g++) {
            int lv[4], sum = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int cidx = g * 4 + e;
                int coef;
                if constexpr (N >= 8) coef = scr[j * RS + cidx];
                else {
                    int acc = 1 << s2 >> 1;
#pragma unroll
                    for (int k = 0; k < N; k++) acc += hv_mul24(tab[cidx * N + k], v[k]);
                    coef = acc >> s2;
                }
                int d = hv_abs(coef) << 14, mag;
                if constexpr (N >= 8) mag = v[cidx]; else mag = hv_quant_lut(lut, hv_abs(coef), d);
                lv[e] = coef < 0 ? -mag : mag;
                sum += hv_min(d, thr);
            }
            sum += hv_shfl_xor(sum, 1);                       /* the 4x4 group spans rows j&~3 .. j|3 = 4 adjacent lanes */
            sum += hv_shfl_xor(sum, 2);
            /* levels go out in scan order.  The 16 levels of this 4x4 group live in four adjacent lanes; they are put in scan
               order through a 32-byte LDS tile per group (the reference-sample buffer is free by now), so that every lane
               stores 8 contiguous bytes instead of four scattered 16-bit values */
            int any = 0, cgs = 0;
            HV_LDSP short *stage = (HV_LDSP short *)W.bord + (lane >> 2) * 16;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (sum < thr) lv[e] = 0;
                any |= lv[e];
                const int x = g * 4 + e;
                const int si = hv_scan_idx(S, LG, kind, j, x);
                if (lv[e] && si > last) last = si;
                stage[si & 15] = (short)lv[e];
                cgs = si >> 4;
                scr[j * RS + x] = (short)hv_clip16((int)((unsigned)lv[e] << qsh));
            }
            WAVE_SYNC();
            {
                const HvPair four = *(const HV_LDSP HvPair *)(stage + (lane & 3) * 4);
                if (mode >= 0) *(HV_GLOBAL HvPair *)((HV_GLOBAL short *)lev_dst + cgs * 16 + (lane & 3) * 4) = four;
            }
            WAVE_SYNC();
            any = any != 0;
            any |= hv_shfl_xor(any, 1);
            any |= hv_shfl_xor(any, 2);
            if (any) { const int bit = (j >> 2) * 8 + g; if (bit < 32) nzmask_lo |= 1 << bit; else nzmask_hi |= 1 << (bit - 32); }
        }
    }
    for (int s = 1; s < N; s <<= 1) {
        last = hv_max(last, hv_shfl_xor(last, s));
        nzmask_lo |= hv_shfl_xor(nzmask_lo, s);
        nzmask_hi |= hv_shfl_xor(nzmask_hi, s);
    }
    WAVE_SYNC();
    HV_TUT(3);
    /* ---- inverse transform, columns then rows ---- */
    {
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = scr[k * RS + j];
        if constexpr (N >= 8) {
            int out[N];
            HvBfly<N, N>::inv(v, out);
#pragma unroll
            for (int i = 0; i < N; i++) scr[i * RS + j] = (short)hv_clip16((out[i] + 64) >> 7);
        } else {
            const int *tab = HvTm<N>::t();
            for (int i = 0; i < N; i++) {
                int acc = 64;
#pragma unroll
                for (int k = 0; k < N; k++) acc += hv_mul24(tab[i * N + k], v[k]);
                scr[i * RS + j] = (short)hv_clip16(acc >> 7);
            }
        }
    }
    WAVE_SYNC();
    HV_TUT(4);
    int sse = 0;
    {
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = scr[j * RS + k];
        const int *tab = HvTm<N>::t();
        int out[N];
        if constexpr (N >= 8) HvBfly<N, N>::inv(v, out);
#pragma unroll
        for (int x4 = 0; x4 < N; x4 += 4) {                      /* four samples = one 32-bit store */
            unsigned pk = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int x = x4 + e;
                int acc = 2048;
                if constexpr (N >= 8) acc += out[x];
                else {
#pragma unroll
                    for (int k = 0; k < N; k++) acc += hv_mul24(tab[x * N + k], v[k]);
                }
                const int r = hv_clip8(hv_clip16(acc >> 12) + pred[j * N + x]);
                const int d = (int)S.org[ty + j][tx + x] - r;
                sse += d * d;
                pk |= (unsigned)r << (8 * e);
                if (ecol && x == N - 1) ecol[j] = (u8)r;
                if (erow && j == N - 1) erow[x] = (u8)r;
            }
            if (mode >= 0) *(HV_GLOBAL HvRingWord *)((HV_GLOBAL u8 *)rec_dst + j * rec_stride + x4) = pk;
        }
    }
    for (int s = 1; s < N; s <<= 1) sse += hv_shfl_xor(sse, s);
    WAVE_SYNC();
    HV_TUT(5);
    HvTuOut o;
    o.sse = sse; o.cbf = (last >= 0); o.last = last < 0 ? 0 : last;
    o.cgsig = ((u64)(unsigned)nzmask_hi << 32) | (unsigned)nzmask_lo;
    return o;
}

/* availability of the four z-ordered sub-blocks  [ref HEVCe.c:1376-1379] */
static HD void hv_sub_avail(int s, int L, int LB, int A, int AR, int &l, int &lb, int &a, int &ar) {
    l  = (s & 1) ? 1 : L;
    lb = (s == 0) ? L : (s == 2 ? LB : 0);
    a  = (s & 2) ? 1 : A;
    ar = (s == 0) ? A : (s == 1 ? AR : (s == 2 ? 1 : 0));
}

typedef HV_GLOBAL short *HvGLev;   typedef HV_GLOBAL u8 *HvGRec;      /* trial scratch in HBM */
typedef HV_LDSP short *HvLLev;     typedef HV_LDSP u8 *HvLRec;        /* wave / NxN staging in LDS */
struct HvCu { int y, x, n, lg, lvl, L, LB, A, AR; };         /* CTU-relative CU under evaluation; lvl = lg-3 */

/* step 2 jobs: one TU of the CU size  [ref HEVCe.c:1422-1432] */
template <int N>
static HDN void hv_job_single(int wave, int lane, int q, HvCu cu, int grp, HvGLev glev, HvGRec grec) {
    HvLds &S = HV_S;
    HvLds::Level &L = S.lv[hv_uni(cu.lvl)];
    grp = hv_uni(grp);
    const int slot = lane / N, j = lane % N;
    const int mode = grp * (64 / N) + slot;
    const bool ok = mode < HEVCE_NMODES;
    HvTuOut o = hv_tu_job<N, HvGLev, HvGRec>(wave, lane, q, cu.y, cu.x, cu.y, cu.x, cu.n, nullptr, 0, cu.L, cu.LB, cu.A, cu.AR,
                             ok ? mode : -1, glev + (ok ? mode : HV_TL - 1) * 1024, grec + (ok ? mode : HV_TL - 1) * 1024, N);
    if (ok && j == 0) {
        L.tsse[mode] = o.sse;
        L.tinfo[mode][0].cgsig = o.cgsig; L.tinfo[mode][0].last = o.last; L.tinfo[mode][0].cbf = o.cbf;
    }
}
/* step 3 jobs: four TUs of half the CU size, each predicted from the previous ones' reconstruction  [ref :1455-1468].
   The reconstruction goes straight to the candidate's HBM scratch; only the edge samples later sub-blocks read stay in LDS */
template <int H>
static HDN void hv_job_chain(int wave, int lane, int q, HvCu cu, int grp, HvGLev glev, HvGRec grec) {
    HvLds &S = HV_S;
    HvLds::Level &L = S.lv[hv_uni(cu.lvl)];
    HvWaveScr &W = S.ws[hv_uni(wave)];
    grp = hv_uni(grp); cu.n = hv_uni(cu.n); cu.y = hv_uni(cu.y); cu.x = hv_uni(cu.x);
    const int slot = lane / H, j = lane % H, n = cu.n;
    const int mode = grp * (64 / H) + slot;
    const bool ok = mode < HEVCE_NMODES;
    const int t = ok ? 35 + mode : HV_TL - 1;          /* idle slots aim at the unused pad trial */
    HV_LDSP u8 *edge = (HV_LDSP u8 *)(W.edge + slot * 4 * H);
    int total = 0;
    for (int s = 0; s < 4; s++) {
        int l, lb, a, ar;
        hv_sub_avail(s, cu.L, cu.LB, cu.A, cu.AR, l, lb, a, ar);
        const int oy = (s >> 1) * H, ox = (s & 1) * H;
        HV_LDSP u8 *ecol = (s == 0) ? edge : (s == 2 ? edge + H : nullptr);
        HV_LDSP u8 *erow = (s == 0) ? edge + 2 * H : (s == 1 ? edge + 3 * H : nullptr);
        HvTuOut o = hv_tu_job<H, HvGLev, HvGRec>(wave, lane, q, cu.y + oy, cu.x + ox, cu.y, cu.x, n, edge, 1, l, lb, a, ar,
                                 ok ? mode : -1, glev + t * 1024 + s * H * H, grec + t * 1024 + oy * n + ox, n, ecol, erow);
        total += o.sse;
        if (ok && j == 0) { L.tinfo[t][s].cgsig = o.cgsig; L.tinfo[t][s].last = o.last; L.tinfo[t][s].cbf = o.cbf; }
    }
    if (ok && j == 0) L.tsse[t] = total;
    WAVE_SYNC();
}
/* 4x4 TU trial with lane = mode: the whole block lives in this lane's registers (no cross-lane traffic).
   b = raw reference samples (corner, left[8], above[8]); 4x4 blocks never use the smoothed ones  [ref :274-275].
   Writes levels in scan order to lev_dst[16], the reconstruction (raster) to rec_dst[16]; returns SSE.
   [ref :1508-1518 = predict / residual / DST / RDOQ / dequant / inverse DST / reconstruct] */
#ifdef HEVCE_PROF_NXN
#define HV_NXT(ph) do { if ((hv_tid() & 63) == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); HV_S.prof2[16 + (ph)] += n_ - nt_; nt_ = n_; } } while (0)
#define HV_NXT0() unsigned long long nt_ = __builtin_amdgcn_s_memtime()
#else
#define HV_NXT(ph) do { } while (0)
#define HV_NXT0() do { } while (0)
#endif
static HD int hv_tu4_lane(const HvLds &S, int q, const u8 *b, u8 *line_buf, int ty, int tx, int m, HV_LDSP short *lev_dst, HV_LDSP u8 *rec_dst, int &last_out) {
    HV_NXT0();
    /* All five prediction forms are evaluated by every lane and the lane's own is selected at the end: lane = mode, so a
       branch per form would make the wavefront walk all of them anyway, plus the merging of sixteen registers per branch.
       Pure horizontal / vertical (10 / 26) are the angular form with angle 0 plus their edge filter. */
    const u8 *l = b + 1, *a = b + 9;
    const int c = b[0];
    int L[5], A[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { L[i] = l[i]; A[i] = a[i]; }
    int p[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    {
        /* angular: this mode's reference line r(k), k = -4..9, is laid out once (r(k>0) main side, r(0) corner, r(k<0) the
           side samples projected through the inverse angle [ref :353-364]); the pixel loop then only indexes it */
        const bool horiz = m < 18;
        const int ang = S.ang[m], inv = S.invang[m];
        const u8 *mainr = horiz ? l : a, *sider = horiz ? a : l;
        u8 *line = line_buf + 4;
#pragma unroll
        for (int k = -4; k <= 9; k++) {
            int v;
            if (k > 0) v = mainr[k > 8 ? 7 : k - 1];
            else if (k == 0) v = c;
            else v = sider[hv_min(hv_max(((128 - inv * k) >> 8) - 1, 0), 7)];   /* entries outside the projected range are never read */
            line[k] = (u8)v;
        }
#pragma unroll
        for (int st = 0; st < 4; st++) {                         /* st = the index the angle steps along (row for vertical modes) */
            const int off = ang * (st + 1), oi = off >> 5, of = off & 31;
#pragma unroll
            for (int al = 0; al < 4; al++) {
                const int k1 = oi + al + 1;
                const int v = (hv_mul24(32 - of, line[k1]) + hv_mul24(of, line[k1 + 1]) + 16) >> 5;
                p[al * 4 + st] = horiz ? v : p[al * 4 + st];
                p[st * 4 + al] = horiz ? p[st * 4 + al] : v;
            }
        }
    }
    {
        int dc = 4;
#pragma unroll
        for (int i = 0; i < 4; i++) dc += L[i] + A[i];
        dc >>= 3;
        const bool planar = m == 0, isdc = m == 1, hor = m == 10, ver = m == 26;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int pl = (4 + (3 - j) * L[i] + (j + 1) * A[4] + (3 - i) * A[j] + (i + 1) * L[4]) >> 3;
                const int dv = (i == 0 && j == 0) ? (2 + 2 * dc + L[0] + A[0]) >> 2 : (i == 0) ? (2 + 3 * dc + A[j]) >> 2 : (j == 0) ? (2 + 3 * dc + L[i]) >> 2 : dc;
                int v = p[i * 4 + j];
                if (i == 0) v = hor ? hv_clip8(L[0] + ((A[j] - c) >> 1)) : v;      /* edge filters of the pure directions */
                if (j == 0) v = ver ? hv_clip8(A[0] + ((L[i] - c) >> 1)) : v;
                p[i * 4 + j] = planar ? pl : (isdc ? dv : v);
            }
    }
    HV_NXT(0);
    int r[16], t[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) r[i * 4 + j] = (int)S.org[ty + i][tx + j] - p[i * 4 + j];
    constexpr int D[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };       /* [ref :391-396] */
    /* forward: columns (shift 1), rows (shift 8) */
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) t[i * 4 + j] = (hv_mul24(D[i * 4], r[j]) + hv_mul24(D[i * 4 + 1], r[4 + j]) + hv_mul24(D[i * 4 + 2], r[8 + j]) + hv_mul24(D[i * 4 + 3], r[12 + j]) + 1) >> 1;   /* 24-bit multiplies throughout: |operands| < 2^16 x 2^7; a plain product becomes a shift-and-add chain or a quarter-rate 32-bit multiply */
    const int sh = 19 + q, thr = 9 << sh >> 2, qsh = 5 + q;
    const HvQlut lut = hv_qlut<2>(S);
    int lv[16], cf[16], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int coef = (hv_mul24(t[i * 4], D[j * 4]) + hv_mul24(t[i * 4 + 1], D[j * 4 + 1]) + hv_mul24(t[i * 4 + 2], D[j * 4 + 2]) + hv_mul24(t[i * 4 + 3], D[j * 4 + 3]) + 128) >> 8;
            int d;
            cf[i * 4 + j] = coef;
            lv[i * 4 + j] = hv_quant_lut(lut, hv_abs(coef), d);       /* sixteen gathers in flight */
            sum += hv_min(d, thr);
        }
#pragma unroll
    for (int k = 0; k < 16; k++) lv[k] = cf[k] < 0 ? -lv[k] : lv[k];
    HV_NXT(1);
    /* levels in scan order: the three 4x4 orders are compile-time permutations of the raster positions, picked per lane by
       selects (no table reads, no divergent paths); two 16-byte LDS stores instead of sixteen scattered ones */
    const int kind = hv_scan_kind(4, m);
    const int zero = sum < thr;
#pragma unroll
    for (int k = 0; k < 16; k++) { lv[k] = zero ? 0 : lv[k]; r[k] = hv_clip16((int)((unsigned)lv[k] << qsh)); }
    constexpr int DG[16] = { 0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15 };   /* up-right diagonal  [ref :1130-1132] */
    int o[16];
    unsigned nzs = 0;
#pragma unroll
    for (int sc = 0; sc < 16; sc++) {
        o[sc] = kind == 0 ? lv[DG[sc]] : (kind == 1 ? lv[sc] : lv[(sc & 3) * 4 + (sc >> 2)]);
        nzs |= (unsigned)(o[sc] != 0) << sc;
    }
    const int last = 31 - hv_clz(nzs);                           /* -1 when all levels are zero */
    {
        HvChunk c0, c1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c0[i] = ((unsigned)o[2 * i] & 0xffffu) | ((unsigned)o[2 * i + 1] << 16);
            c1[i] = ((unsigned)o[8 + 2 * i] & 0xffffu) | ((unsigned)o[9 + 2 * i] << 16);
        }
        ((HV_LDSP HvChunk *)lev_dst)[0] = c0; ((HV_LDSP HvChunk *)lev_dst)[1] = c1;
    }
    HV_NXT(2);
    /* inverse: columns with C^T (shift 7), rows (shift 12) */
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) t[i * 4 + j] = hv_clip16((hv_mul24(D[i], r[j]) + hv_mul24(D[4 + i], r[4 + j]) + hv_mul24(D[8 + i], r[8 + j]) + hv_mul24(D[12 + i], r[12 + j]) + 64) >> 7);
    int sse = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int res = hv_clip16((hv_mul24(t[i * 4], D[j]) + hv_mul24(t[i * 4 + 1], D[4 + j]) + hv_mul24(t[i * 4 + 2], D[8 + j]) + hv_mul24(t[i * 4 + 3], D[12 + j]) + 2048) >> 12);
            const int rc = hv_clip8(res + p[i * 4 + j]);
            const int dd = (int)S.org[ty + i][tx + j] - rc;
            sse += hv_mul24(dd, dd);
            p[i * 4 + j] = rc;
        }
    {
        HvChunk rc4;
#pragma unroll
        for (int i = 0; i < 4; i++) rc4[i] = (unsigned)p[i * 4] | ((unsigned)p[i * 4 + 1] << 8) | ((unsigned)p[i * 4 + 2] << 16) | ((unsigned)p[i * 4 + 3] << 24);
        *(HV_LDSP HvChunk *)rec_dst = rc4;
    }
    HV_NXT(3);
    last_out = last;
    return sse;
}

/* The same 4x4 block trial with FOUR lanes per mode (lane = mode * 4 + row): a helper wavefront of the chain workgroup does 16
   modes at a quarter of the instructions per lane; three of them stand in for hv_tu4_lane when the workgroup's other wavefronts
   are free (cluster of four).  Rows and columns change hands through a 32-byte tile per mode in the wavefront's scratch.
   Results go where hv_tu4_lane puts them: S.nxn_lev / nxn_rec / nxn_sse / nxn_last / nxn_sig of the mode.  [ref :1508-1518] */
static HD void hv_tu4_quad(HvLds &S, int q, int ty, int tx, int m, int ok) {
    const int lane = hv_tid() & 63, qd = lane >> 2, i = lane & 3;
    HvWaveScr &W = S.ws[hv_tid() >> 6];
    HV_LDSP short *tile = (HV_LDSP short *)W.scr + qd * 16;       /* [4][4] 16-bit, exchanged between the mode's four lanes */
    HV_LDSP u8 *line = (HV_LDSP u8 *)W.pred + qd * 16 + 4;        /* the mode's reference line r(-4..9) */
    const u8 *b = S.nxn_bord, *l = b + 1, *a = b + 9;
    const int c = b[0];
    m = ok ? m : 1;
    const bool horiz = m < 18;
    const int ang = S.ang[m], inv = S.invang[m];
    {   /* reference line: lane i lays out r(4 i - 4 .. 4 i - 1)  [ref :353-364] */
        const u8 *mainr = horiz ? l : a, *sider = horiz ? a : l;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int k = 4 * i - 4 + e;
            const int vm = mainr[hv_min(hv_max(k, 1), 8) - 1], vs = sider[hv_min(hv_max(((128 - inv * k) >> 8) - 1, 0), 7)];
            line[k] = (u8)(k > 0 ? vm : (k == 0 ? c : vs));
        }
    }
    WAVE_SYNC();
    /* prediction of row i: all forms, the mode's own selected  [ref :262-381] */
    int p[4];
    {
        const int Li = l[i], L4 = l[4], A4 = a[4], L0 = l[0], A0 = a[0];
        int dc = 4;
#pragma unroll
        for (int k = 0; k < 4; k++) dc += l[k] + a[k];
        dc >>= 3;
        const bool planar = m == 0, isdc = m == 1, hor = m == 10, ver = m == 26;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int Ax = a[x];
            const int st = horiz ? x : i, al = horiz ? i : x;
            const int off = ang * (st + 1), oi = off >> 5, of = off & 31, k1 = oi + al + 1;
            int v = (hv_mul24(32 - of, line[k1]) + hv_mul24(of, line[k1 + 1]) + 16) >> 5;
            const int pl = (4 + (3 - x) * Li + (x + 1) * A4 + (3 - i) * Ax + (i + 1) * L4) >> 3;
            const int dv = (i == 0 && x == 0) ? (2 + 2 * dc + L0 + A0) >> 2 : (i == 0) ? (2 + 3 * dc + Ax) >> 2 : (x == 0) ? (2 + 3 * dc + Li) >> 2 : dc;
            v = (i == 0 && hor) ? hv_clip8(L0 + ((Ax - c) >> 1)) : v;      /* edge filters of the pure directions */
            v = (x == 0 && ver) ? hv_clip8(A0 + ((Li - c) >> 1)) : v;
            p[x] = planar ? pl : (isdc ? dv : v);
        }
    }
    constexpr int D[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };       /* [ref :391-396] */
    /* residual row i -> tile, read back as column i */
    int r[4], t[4];
#pragma unroll
    for (int x = 0; x < 4; x++) tile[x * 4 + i] = (short)((int)S.org[ty + i][tx + x] - p[x]);      /* stored transposed: tile[column][row] */
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = tile[i * 4 + k];                                            /* r[k] = residual(row k, column i) */
    WAVE_SYNC();
    /* forward, columns (shift 1): t[k] = coefficient (k, column i) */
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = (hv_mul24(D[k * 4], r[0]) + hv_mul24(D[k * 4 + 1], r[1]) + hv_mul24(D[k * 4 + 2], r[2]) + hv_mul24(D[k * 4 + 3], r[3]) + 1) >> 1;
#pragma unroll
    for (int k = 0; k < 4; k++) tile[k * 4 + i] = (short)t[k];                                      /* tile[row][column] */
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = tile[i * 4 + k];                                            /* row i of the intermediate */
    WAVE_SYNC();
    /* forward, rows (shift 8), quantiser  [ref :540-595] */
    const int sh = 19 + q, thr = 9 << sh >> 2, qsh = 5 + q;
    const HvQlut lut = hv_qlut<2>(S);
    int cf[4], lv[4], sum = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        cf[x] = (hv_mul24(t[0], D[x * 4]) + hv_mul24(t[1], D[x * 4 + 1]) + hv_mul24(t[2], D[x * 4 + 2]) + hv_mul24(t[3], D[x * 4 + 3]) + 128) >> 8;
        int d;
        lv[x] = hv_quant_lut(lut, hv_abs(cf[x]), d);
        sum += hv_min(d, thr);
    }
    sum += hv_shfl_xor(sum, 1);
    sum += hv_shfl_xor(sum, 2);
    const int zero = sum < thr;
    const int kind = hv_scan_kind(4, m);
    int last = -1;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        lv[x] = zero ? 0 : (cf[x] < 0 ? -lv[x] : lv[x]);
        const int si = kind == 0 ? S.d4idx[(i << 2) | x] : (kind == 1 ? i * 4 + x : x * 4 + i);
        if (ok) S.nxn_lev[m][si] = (short)lv[x];
        last = (lv[x] != 0 && si > last) ? si : last;
        tile[i * 4 + x] = (short)hv_clip16((int)((unsigned)lv[x] << qsh));                            /* dequantised, tile[row][column] */
    }
    last = hv_max(last, hv_shfl_xor(last, 1));
    last = hv_max(last, hv_shfl_xor(last, 2));
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = tile[k * 4 + i];                                            /* column i */
    WAVE_SYNC();
    /* inverse: columns with C^T (shift 7), rows (shift 12) */
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = hv_clip16((hv_mul24(D[k], r[0]) + hv_mul24(D[4 + k], r[1]) + hv_mul24(D[8 + k], r[2]) + hv_mul24(D[12 + k], r[3]) + 64) >> 7);
#pragma unroll
    for (int k = 0; k < 4; k++) tile[k * 4 + i] = (short)t[k];
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = tile[i * 4 + k];                                            /* row i */
    int sse = 0;
    unsigned rc4 = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int res = hv_clip16((hv_mul24(t[0], D[x]) + hv_mul24(t[1], D[4 + x]) + hv_mul24(t[2], D[8 + x]) + hv_mul24(t[3], D[12 + x]) + 2048) >> 12);
        const int rc = hv_clip8(res + p[x]);
        const int dd = (int)S.org[ty + i][tx + x] - rc;
        sse += hv_mul24(dd, dd);
        rc4 |= (unsigned)rc << (8 * x);
    }
    sse += hv_shfl_xor(sse, 1);
    sse += hv_shfl_xor(sse, 2);
    if (ok) {
        *(HV_LDSP unsigned *)&S.nxn_rec[m][i * 4] = rc4;
        if (i == 0) { S.nxn_sse[m] = sse; S.nxn_last[m] = last < 0 ? 0 : last; S.nxn_sig[m] = last >= 0; }
    }
    WAVE_SYNC();
}
