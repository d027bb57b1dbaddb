/* hevce_dev_syntax.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   syntax = binarisation (tokens) + arithmetic engines: the EXACT engines and the FAST (range-only) engines. */
/* --------------------------------------------------------------------------------------------- */
/* Syntax = binarisation + arithmetic engine, separated.                                            */
/*                                                                                                  */
/* A candidate's bins are first written as 16-bit TOKENS into fixed-size SLOTS (one slot per CU      */
/* header, per TU header and per 4x4 coefficient group, in coding order); the binarisation of a slot */
/* needs only that group's levels and a few TU-level facts, so slots are produced data-parallel.     */
/* The arithmetic engine is then a tight loop in which every lane consumes ITS candidate's token      */
/* stream: all lanes run the same instructions (no per-candidate control flow), which is what a      */
/* 64-wide wavefront needs.  The reference interleaves both (HEVCe.c:943-1340 call CABACputBin*).    */
/*                                                                                                  */
/*   token, context bin : bit15=0, bit14 = bin value, bits0-12 = byte offset of the context inside the         */
/*                        candidate's context column (index * HV_TL); the greater1 / greater2 context sets   */
/*                        already include the "previous group had a level > 1" memory                        */
/*   token, bypass chunk: bit15=1, bits8-11 length (0..8), bits0-7 bins  -- one token per <=8-bit chunk,      */
/*                        exactly the chunks reference CABACputBins makes (HEVCe.c:899-911); length 0 = no-op */
/*   A slot in HBM is padded with no-ops to whole 16-byte chunks (8 tokens); the engine walks the candidate's  */
/*   non-empty slots in order (list in LDS), so no contiguous copy of the stream is ever made.                */
/* --------------------------------------------------------------------------------------------- */
#define HV_SLOT        HEVCE_TOKEN_SLOT_WORDS /* 16-bit words per slot: header + up to 127 tokens (a group needs <= 109) */
#define HV_WSLOT       HEVCE_WRTOK_SLOT_WORDS /* the stream writer's slots: exact tokens, always the full size */
#define HV_SLOTS_TRIAL HEVCE_TOKEN_SLOTS                     /* CU header + at most 4 x (1 + 16) or 1 + 64 block/group slots            */
#define HV_TOK_BYP     0x8000
#define HV_TOK_NOP     HV_TOK_BYP
/* FAST streams (range-only engines, see hv_fengine_chunk) hold context-coded bins only; the bypass bins are COUNTED at
   binarisation (they do not move the range), and a bin on the pad context -- context byte HV_PADCV, whose table entry has an
   LPS range of 0 and is its own successor -- is the no-op that pads a slot to whole chunks */
#define HV_CT_PAD      (HEVCE_NCTXP - 1)
#define HV_PADCV       126
#define HV_TOK_NOPF    (HV_CT_PAD * HV_TL)
#define HV_CT(idx, bin) (((idx) * HV_TL) | ((bin) ? 0x4000 : 0))   /* context-coded bin */

template <typename P, bool FAST = false> struct HvTokW { P p; int n; int byp; enum { fast = FAST }; };
#define HV_SLOT_TOK0 8                      /* tokens start 16 bytes into the slot so that chunks of 8 load as one dwordx4 */
/* trial slots in HBM: EXACT streams in 256-byte slots (header + <= 109 tokens), FAST streams -- context bins only, <= 31 per slot:
   CU header <= 10, block header <= 19, 4x4 group <= 26, 4x4 block with its header <= 31 -- in 64-byte slots of their own region, so
   that the lines a FAST analysis rewrites CTU after CTU are dense and stay in the L2 (hevce_common.h HEVCE_SCR_*) */
template <bool F> struct HvSlotGeo { enum { words = F ? HEVCE_FTOKEN_SLOT_WORDS : HEVCE_TOKEN_SLOT_WORDS, tok0 = F ? 0 : HV_SLOT_TOK0, base = F ? HEVCE_SCR_FTOK : HEVCE_SCR_XTOK }; };
static HD HV_GLOBAL u16 *hv_c70_area(HV_GLOBAL u8 *s0, int fast) {   /* candidate 70 (the NxN outcome): one contiguous stream in its trial's slots */
    return fast ? (HV_GLOBAL u16 *)(s0 + HEVCE_SCR_FTOK) + (size_t)70 * HEVCE_TOKEN_SLOTS * HEVCE_FTOKEN_SLOT_WORDS   /* FAST slots have no header */
                : (HV_GLOBAL u16 *)(s0 + HEVCE_SCR_XTOK) + (size_t)70 * HEVCE_TOKEN_SLOTS * HEVCE_TOKEN_SLOT_WORDS + HV_SLOT_TOK0;
}

static HD void hv_mpm(int left, int above, int out[3]) {                                                   /* [ref :958-977] */
    if (left != above) {
        out[0] = left; out[1] = above;
        out[2] = (left != 0 && above != 0) ? 0 : (left + above < 2 ? 26 : 1);
    } else if (left > 1) {
        out[0] = left; out[1] = ((left + 29) & 31) + 2; out[2] = ((left - 1) & 31) + 2;
    } else { out[0] = 0; out[1] = 1; out[2] = 26; }
}

struct HvModes { int mode[4], left[4], above[4]; };

/* Token writers.  hv_emit() is branch-free: the token is always stored at the write position and the position advances
   only when `on`; a slot has headroom for the extra store (<= 110 of 120 token words are ever counted).
   HvTokW<P>: direct stores (slots in LDS: the NxN sub-search). */
template <typename P, bool F> static HD void hv_emit(HvTokW<P, F> &w, int tok, int on) { w.p[HV_SLOT_TOK0 + w.n] = (u16)tok; w.n += on; }
template <typename P, bool F> static HD void hv_tok_checkpoint(HvTokW<P, F> &) { }
template <typename P, bool F> static HD int  hv_tok_finish(HvTokW<P, F> &w) { return w.n; }                /* tokens */
/* one chunk of <= 8 bypass bins: a token, or -- FAST -- only its length */
template <typename WR> static HD void hv_emit_b(WR &w, int bins, int len, int on) {
    if constexpr (WR::fast) w.byp += on ? len : 0;
    else hv_emit(w, HV_TOK_BYP | (len << 8) | bins, on);
}

/* Writer for slots in HBM.  Sixty-four lanes storing single 16-bit tokens into sixty-four different slots cost one
   memory transaction per lane and token (measured: 60 % of the binarisation time).  Tokens therefore go to a small ring
   in the wavefront's LDS scratch (32 tokens per lane, lane-transposed: dword row r of lane l at r*64+l, conflict free)
   and leave it as 16-byte chunks.  hv_tok_checkpoint() writes out up to three complete chunks; between two checkpoints
   at most 24 tokens may be added. */
typedef unsigned __attribute__((may_alias)) HvRingWord;    /* written as 16-bit halves, read back as dwords */
template <bool FAST> struct HvTokRingT { HV_LDSP HvRingWord *ring; HV_GLOBAL u16 *slot; int n, fl, byp, tok0; enum { fast = FAST }; };
template <bool FAST> static HD HvTokRingT<FAST> hv_tok_ring(HvLds &S, HV_GLOBAL u16 *slot, int tok0 = HV_SLOT_TOK0) {
    const int tid = hv_tid();
    HvTokRingT<FAST> w = { (HV_LDSP HvRingWord *)&S.ws[tid >> 6] + (tid & 63), slot, 0, 0, 0, tok0 };
    return w;
}
template <bool F> static HD void hv_emit(HvTokRingT<F> &w, int tok, int on) {
    ((HV_LDSP u16 *)(w.ring + ((w.n >> 1) & 15) * 64))[w.n & 1] = (u16)tok;
    w.n += on;
}
template <bool F> static HD void hv_tok_flush_chunk(HvTokRingT<F> &w) {
    const HV_LDSP HvRingWord *r = w.ring + ((w.fl >> 1) & 15) * 64;
    HvChunk c;
    c[0] = r[0]; c[1] = r[64]; c[2] = r[128]; c[3] = r[192];
    *(HV_GLOBAL HvChunk *)(w.slot + w.tok0 + w.fl) = c;
    w.fl += 8;
}
template <bool F> static HD void hv_tok_checkpoint(HvTokRingT<F> &w) {
    if (w.n - w.fl >= 8) hv_tok_flush_chunk(w);
    if (w.n - w.fl >= 8) hv_tok_flush_chunk(w);
    if (w.n - w.fl >= 8) hv_tok_flush_chunk(w);
}
template <bool F> static HD int hv_tok_finish(HvTokRingT<F> &w) {                   /* after a checkpoint: fewer than 8 tokens are pending */
#pragma unroll
    for (int i = 0; i < 7; i++) ((HV_LDSP u16 *)(w.ring + (((w.n + i) >> 1) & 15) * 64))[(w.n + i) & 1] = (u16)(F ? HV_TOK_NOPF : HV_TOK_NOP);
    hv_tok_flush_chunk(w);                                    /* the last chunk, padded with no-ops */
#ifdef HEVCE_EMU
    if (F && w.tok0 == 0 && w.n > HEVCE_FTOKEN_SLOT_WORDS - 1) { fprintf(stderr, "emu: FAST slot overflow (%d tokens)\n", w.n); abort(); }
    if (F && w.tok0 == 0) { emu_tok_stat[0] += w.n; emu_tok_stat[1] += (w.n + 7) >> 3; emu_tok_stat[2] += 1; }
#endif
    return (w.n + 7) >> 3;                                    /* chunks */
}

/* CU header slot: split_cu_flag=0, part_mode, luma modes (all prev_intra_luma_pred_flags first, then mpm_idx /
   rem_intra_luma_pred_mode), chroma mode, split_transform_flag, cbf_cb, cbf_cr   [ref :943-1033, 1282-1287]
   Straight-line: every possible token is stored, the write position advances only for the ones that exist. */
template <typename WR> static HD int hv_tokenize_cu(WR &w, int n, int lg, int kind, int split_inc, const HvModes &pm) {
    hv_emit(w, HV_CT(CX_SPLIT_CU + split_inc, 0), n >= 16);
    hv_emit(w, HV_CT(CX_PART, kind != 2), n == 8);
    const int parts = kind == 2 ? 4 : 1;
    int hit[4], rem[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int p[3];
        hv_mpm(pm.left[i], pm.above[i], p);
        const int md = pm.mode[i];
        const int h = (p[2] == md) ? 2 : (p[1] == md) ? 1 : (p[0] == md) ? 0 : -1;
        const int hi = hv_max(p[0], hv_max(p[1], p[2])), lo = hv_min(p[0], hv_min(p[1], p[2])), mid = p[0] + p[1] + p[2] - hi - lo;
        int r = md;                                            /* skip the candidates from the largest down */
        r -= (r > hi); r -= (r > mid); r -= (r > lo);
        hit[i] = h; rem[i] = r;
        hv_emit(w, HV_CT(CX_YPM, h >= 0), i < parts);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int on = i < parts, ishit = hit[i] >= 0;
        hv_emit_b(w, hit[i] > 0, 1, on && ishit);
        hv_emit_b(w, (hit[i] - 1) & 1, 1, on && hit[i] > 0);
        hv_emit_b(w, rem[i] & 31, 5, on && !ishit);
    }
    hv_emit(w, HV_CT(CX_UVPM, 0), 1);
    hv_emit(w, HV_CT(CX_SPLIT_TU + (5 - lg), kind == 1), kind != 2);
    hv_emit(w, HV_CT(CX_UVCBF, 0), 1);
    hv_emit(w, HV_CT(CX_UVCBF, 0), 1);
    hv_tok_checkpoint(w);
    return hv_tok_finish(w);
}

/* last_sig_coeff prefix group of a coordinate and the smallest coordinate of a group, in closed form
   (tables HEVCE_TAB_LASTGRP / HEVCE_TAB_LASTMIN; a per-lane table read would be a memory access)  [ref :1038-1045] */
static HD int hv_last_group(int v) { const int l2 = 31 - hv_clz((unsigned)hv_max(v, 1)); return v < 4 ? v : 2 * l2 + ((v >> (l2 - 1)) & 1); }
static HD int hv_last_min(int g) { return g < 4 ? g : (2 + (g & 1)) << ((g >> 1) - 1); }

/* TU header slot: cbf_luma (unless the caller is the NxN sub-search, which codes bare residuals) and the position of
   the last significant coefficient   [ref :1038-1087, 1192].  Straight-line like the CU header (<= 25 tokens). */
template <int MAXG = 9, typename WR> static HD void hv_tokenize_tu_body(const HvLds &S, WR &w, int n, int lg, int kind_scan, int with_cbf, int cbf_ctx, int cbf, int last) {   /* MAXG: largest prefix group the caller's block sizes can reach (4x4: 3) */
    hv_emit(w, HV_CT(cbf_ctx, cbf), with_cbf);
    const int on0 = cbf || !with_cbf;
    int gy, gx, iy, ix;
    hv_scan_cg(S, lg, kind_scan, last >> 4, gy, gx);
    hv_scan_in(S, kind_scan, last & 15, iy, ix);
    const int y = gy * 4 + iy, x = gx * 4 + ix;
    const int cls = lg - 2, sh = (n == 4) ? 0 : 1;
    int ty = kind_scan == 2 ? x : y, tx = kind_scan == 2 ? y : x;
    const int ggy = hv_last_group(ty), ggx = hv_last_group(tx), gmax = 2 * lg - 1;       /* gmax = group of n - 1 */
#pragma unroll
    for (int i = 0; i < MAXG; i++) hv_emit(w, HV_CT(CX_LASTX + 5 * cls + (i >> sh), 1), on0 && i < ggx);
    hv_emit(w, HV_CT(CX_LASTX + 5 * cls + (ggx >> sh), 0), on0 && ggx < gmax);
#pragma unroll
    for (int i = 0; i < MAXG; i++) hv_emit(w, HV_CT(CX_LASTY + 5 * cls + (i >> sh), 1), on0 && i < ggy);
    hv_emit(w, HV_CT(CX_LASTY + 5 * cls + (ggy >> sh), 0), on0 && ggy < gmax);
    const int nx = ggx > 3 ? ((ggx - 2) >> 1) : 0, ny = ggy > 3 ? ((ggy - 2) >> 1) : 0;   /* suffix bits, one chunk per bit */
    tx -= hv_last_min(ggx); ty -= hv_last_min(ggy);
#pragma unroll
    for (int i = 2; i >= 0; i--) if (MAXG > 3) hv_emit_b(w, (tx >> i) & 1, 1, on0 && i < nx);   /* groups <= 3 have no suffix */
#pragma unroll
    for (int i = 2; i >= 0; i--) if (MAXG > 3) hv_emit_b(w, (ty >> i) & 1, 1, on0 && i < ny);
    hv_tok_checkpoint(w);
}
template <int MAXG = 9, typename WR> static HD int hv_tokenize_tu(const HvLds &S, WR &w, int n, int lg, int kind_scan, int with_cbf, int cbf_ctx, int cbf, int last) {
    hv_tokenize_tu_body<MAXG>(S, w, n, lg, kind_scan, with_cbf, cbf_ctx, cbf, last);
    return hv_tok_finish(w);
}

static HD int hv_sigctx(int n, int diag, int y, int x, int nb) {                                           /* [ref :1092-1122] */
    if ((y | x) == 0) return 0;
    if (n == 4) return hv_sig4[(y << 2) | x];
    int base = 9;
    if (n >= 16) base += 12;
    if (n == 8 && !diag) base += 6;
    if ((y >> 2) | (x >> 2)) base += 3;
    const int iy = y & 3, ix = x & 3;
    if (nb == 0) return base + hv_sigpos[iy + ix];
    if (nb == 1) return base + hv_sigpos[iy << 1];
    if (nb == 2) return base + hv_sigpos[ix << 1];
    return base + 2;
}

static HD int hv_sigtab_index(int n, int kind, int nb, int first_cg) {
    if (n == 4) return kind;
    if (n == 8) return 3 + kind * 8 + nb * 2 + (first_cg ? 0 : 1);
    return 27 + nb * 2 + (first_cg ? 0 : 1);
}
/* bypass bins of length len (<= 24) as <= 3 chunks of <= 8 from the top, exactly like hv_tok_byp  [ref :899-911] */
template <typename WR> static HD void hv_emit_byp(WR &w, int bins, int len, int on) {
    if constexpr (WR::fast) { w.byp += on ? len : 0; return; }
    bins &= (1 << len) - 1;
    const int n1 = hv_min(len, 8), r1 = len - n1, n2 = hv_min(r1, 8), n3 = r1 - n2;
    hv_emit(w, HV_TOK_BYP | (n1 << 8) | ((bins >> r1) & ((1 << n1) - 1)), on && n1 > 0);
    hv_emit(w, HV_TOK_BYP | (n2 << 8) | ((bins >> n3) & ((1 << n2) - 1)), on && n2 > 0);
    hv_emit(w, HV_TOK_BYP | (n3 << 8) | (bins & ((1 << n3) - 1)), on && n3 > 0);
}
/* coeff_abs_level_remaining  [ref :1154-1169]: a prefix and a suffix, each put as its own bin string (chunks of <= 8).
   short form (v < 3<<k): prefix of (v>>k)+1 <= 3 bins, suffix of k <= 4 bins.  long form (Exp-Golomb escape): prefix of
   t = 4+len-k bins, suffix of len bins.  Whenever both strings fit one chunk (always in the short form, and in the long
   form up to v of a few hundred) the two tokens are emitted straight-line; only longer strings, which are rare even per
   wavefront, go through the chunking code behind a branch. */
template <typename WR> static HD void hv_emit_remainder(WR &w, int v, int k, int on) {
    const int shortf = v < (3 << k);
    /* escape: the reference subtracts 2^k, 2^(k+1), ... while the rest is >= 2^len; closed form of that loop */
    const int w2 = hv_max(v - (3 << k), 0) + (1 << k);
    const int len = 31 - hv_clz((unsigned)w2), u = w2 - (1 << len);
    const int plen = shortf ? (v >> k) + 1 : 4 + len - k, pbins = (1 << plen) - 2;
    const int slen = shortf ? k : len, sbins = shortf ? (v & ((1 << k) - 1)) : u;
    if constexpr (WR::fast) { w.byp += on ? plen + slen : 0; return; }
    const int simple = (plen <= 8) & (slen <= 8);
    hv_emit(w, HV_TOK_BYP | (plen << 8) | (pbins & 0xff), on && simple);
    hv_emit(w, HV_TOK_BYP | (slen << 8) | (sbins & 0xff), on && simple && slen > 0);
    if (on && !simple) {
        hv_emit_byp(w, pbins, plen, 1);
        hv_emit_byp(w, sbins, slen, 1);
    }
}

/* one 4x4 coefficient group of a transform block whose levels lie in SCAN ORDER at lev16[0..15]   [ref :1194-1268]
   cg = the group's index in scan order; last / cgsig describe the whole block (from the TU job).
   Two steps: hv_cg_load() fetches the levels and derives what the NEXT coded group must know (g1own: one of the first
   eight non-zero levels is > 1, which bumps that group's greater1/greater2 context sets  [ref :1222-1226]);
   hv_cg_emit() writes the tokens, given that fact about the PREVIOUS coded group (g1).
   Straight-line code: fixed trip counts, unconditional stores, conditional counts -- threads binarising different groups
   stay converged and nothing is a branch. */
/* pointer to 16-byte chunks in the same address space as a pointer to levels */
template <typename T> struct HvAsChunk;
#ifdef HEVCE_EMU
template <> struct HvAsChunk<const short *> { typedef const HvChunk *type; };
#else
template <> struct HvAsChunk<const HV_GLOBAL short *> { typedef const HV_GLOBAL HvChunk *type; };
template <> struct HvAsChunk<const HV_LDSP short *> { typedef const HV_LDSP HvChunk *type; };
#endif
struct HvCg { int v[16]; unsigned nzm; int coded, cgs, nb, first_cg, at_last_cg, start, n, kind, g1own; };
template <typename LP> static HD void hv_cg_load(const HvLds &S, HvCg &g, LP lev16, int n, int lg, int kind, int cg, int last, u64 cgsig, int valid) {
    const int cg_last = last >> 4;
    int gy, gx;
    hv_scan_cg(S, lg, kind, cg, gy, gx);
    const int ncg = n >> 2;
    g.n = n; g.kind = kind;
    g.coded = valid && cg <= cg_last;
    g.first_cg = (cg == 0); g.at_last_cg = (cg == cg_last);
    g.cgs = g.coded ? (int)((cgsig >> (gy * 8 + gx)) & 1) : 0;
    const int right = (gx < ncg - 1) ? (int)((cgsig >> (gy * 8 + gx + 1)) & 1) : 0;
    const int below = (gy < ncg - 1) ? (int)((cgsig >> (gy * 8 + 8 + gx)) & 1) : 0;
    g.nb = (below << 1) | right;
    g.start = g.at_last_cg ? (last & 15) : 15;
    unsigned nzm = 0, bigm = 0;
    /* the group's 16 levels are 32 contiguous, 32-byte aligned bytes: two 16-byte loads instead of sixteen 2-byte ones
       (always readable: scratch is allocated for every group; what a group without levels holds is ignored) */
    const HvChunk lo = ((typename HvAsChunk<LP>::type)lev16)[0], hi = ((typename HvAsChunk<LP>::type)lev16)[1];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const unsigned d = (k < 8 ? lo : hi)[(k & 7) >> 1];
        const int val = (int)(short)(d >> ((k & 1) << 4));
        g.v[k] = g.cgs ? val : 0;
        nzm |= (unsigned)(g.v[k] != 0) << k;
        bigm |= (unsigned)(hv_abs(g.v[k]) > 1) << k;
    }
    g.nzm = nzm;
    unsigned m8 = nzm;                                         /* the eight non-zero levels coded first = highest scan positions */
#pragma unroll
    for (int i = 0; i < 8; i++) m8 = (__builtin_popcount(m8) > 8) ? (m8 & (m8 - 1)) : m8;
    g.g1own = (bigm & m8) != 0;
}
/* in two parts, so that two wavefronts can share a group's binarisation (the NxN chain with helpers): part 1 = the
   coded_sub_block_flag and the significance flags, part 2 = greater1 / greater2 flags, signs, remaining magnitudes */
template <typename WR> static HD void hv_cg_emit_sig(const HvLds &S, WR &w, const HvCg &g) {
    if (g.coded) {
        const int first_cg = g.first_cg, at_last_cg = g.at_last_cg, cgs = g.cgs, start = g.start;
        const unsigned nzm = g.nzm;
        hv_emit(w, HV_CT(CX_SIGCG + (g.nb != 0), cgs), !first_cg && !at_last_cg);
        if (cgs || first_cg) {
            const u8 *tab = S.sigtab[hv_sigtab_index(g.n, g.kind, g.nb, first_cg)];
            /* significance flags, highest scan position first; position 0 of a later group is inferred when nothing else is set */
            const int any_above0 = (nzm >> 1) != 0;
#pragma unroll
            for (int ii = 15; ii >= 0; ii--) {
                const int on = (ii <= start) && !(at_last_cg && ii == start) && (ii != 0 || first_cg || any_above0);
                hv_emit(w, HV_CT(CX_SIG + tab[ii], (nzm >> ii) & 1), on);
            }
            hv_tok_checkpoint(w);                                  /* <= 17 tokens so far */
        }
    }
}
template <typename WR> static HD void hv_cg_emit_mag(const HvLds &S, WR &w, const HvCg &g, int g1) {
    if (g.coded) {
        const int first_cg = g.first_cg, cgs = g.cgs;
        const unsigned nzm = g.nzm;
        if (cgs || first_cg) {
            const int nz = __builtin_popcount(nzm);
            /* greater1 flags of the first eight non-zero levels, one greater2 flag; sign bits in the same order */
            const int set4 = (first_cg ? 0 : 8) + (g1 ? 4 : 0), set1 = (first_cg ? 0 : 2) + (g1 ? 1 : 0);
            int esc = nz > 8, g2 = -1, c1 = 1, j = 0, signs = 0;
#pragma unroll
            for (int ii = 15; ii >= 0; ii--) {
                const int isnz = (nzm >> ii) & 1;
                const int a = hv_abs(g.v[ii]), on = isnz && j < 8, big = a > 1;
                hv_emit(w, HV_CT(CX_ONE + set4 + c1, big), on);
                const int first_big = on && big && g2 < 0, more_big = on && big && g2 >= 0;
                g2 = first_big ? (a > 2) : g2;
                esc |= more_big;
                c1 = (on && big) ? 0 : ((on && c1 > 0 && c1 < 3) ? c1 + 1 : c1);
                signs = isnz ? ((signs << 1) | (g.v[ii] < 0)) : signs;
                j += isnz;
            }
            const int has_g2 = nz > 0 && c1 == 0 && g2 >= 0;
            hv_emit(w, HV_CT(CX_ABS + set1, g2 > 0), has_g2);
            esc |= has_g2 && g2 > 0;
            hv_emit_byp(w, signs, nz, nz > 0);
            /* remaining magnitudes with the adaptive Rice parameter */
            int base = 3, rice = 0;
            j = 0;
#pragma unroll
            for (int ii = 15; ii >= 0; ii--) {
                const int isnz = (nzm >> ii) & 1;
                const int a = hv_abs(g.v[ii]), rem = a - (j < 8 ? base : 1);
                const int on = isnz && esc && rem >= 0;
                hv_emit_remainder(w, hv_max(rem, 0), rice, on);
                rice = (on && a > (3 << rice)) ? hv_min(rice + 1, 4) : rice;
                base = (isnz && a >= 2) ? 2 : base;
                j += isnz;
                if ((ii & 3) == 2) hv_tok_checkpoint(w);          /* <= 11 + 2 x 5, then <= 4 x 5 tokens per section */
            }
        }
    }
}
template <typename WR> static HD int hv_cg_emit(const HvLds &S, WR &w, const HvCg &g, int g1) {
    hv_cg_emit_sig(S, w, g);
    hv_cg_emit_mag(S, w, g, g1);
    hv_tok_checkpoint(w);
    return hv_tok_finish(w);
}

#define HV_CX(i) (cx + (i) * HV_TL)
/* The arithmetic engine of a candidate  [ref CABACputBin / CABACputBins :899-933]: resumable -- the coder registers and the
   position in the token stream are all the state there is.  cx = the candidate's context column (&tctx[0][candidate]).
   All lanes of a wavefront run the loop in lockstep; one iteration = one token through the same straight-line code for
   context-coded bins and bypass chunks; 16-byte chunks of tokens are requested eight tokens ahead.
   Pruning (exact): a candidate's cost w1*sse + w2*bits only grows while it is being encoded, so once its partial cost
   exceeds the cost of a candidate that has COMPLETED (or of the split alternative, known beforehand) it cannot be the
   arg-min, which is the LAST candidate of MINIMAL cost (ref :1440 adopts on >=). */
/* one token through the coder: context-coded bin or bypass chunk, same straight-line code.
   cv = the token's context byte, ct = S.ctab[cv] -- both fetched while the PREVIOUS token was processed, so that no
   memory access sits on the range -> range dependency chain (the LPS range is a byte of ct picked by the range quarter,
   its renormalisation shift a leading-zero count).  Returns the context byte to write back. */
struct HvCt { unsigned lps4, nxt; };
static HD int hv_engine_next_cv(int tok, int cv, const HvCt &ct) {
    const int isb = (tok >> 15) & 1, islps = (((tok >> 14) & 1) ^ cv) & 1;
    return isb ? cv : (int)((ct.nxt >> (islps ? 0 : 8)) & 0xff);
}
template <typename SinkP> static HD void hv_engine_step(HvCod &c, int cv, const HvCt &ct, int tok, SinkP sink) {
    const int isb = (tok >> 15) & 1;
    const int nb = (tok >> 8) & 15, bv = tok & 0xff;
    const int lps = (int)((ct.lps4 >> (((c.range >> 6) & 3) << 3)) & 0xff);
    const int shl = hv_clz_nz((unsigned)lps) - 23;               /* lps in 6..240: 1..6 */
    const int islps = (((tok >> 14) & 1) ^ cv) & 1;
    const int rm = c.range - lps;
    const int shm = (rm < 256) ? 1 : 0;
    const unsigned low_c = islps ? ((c.low + (unsigned)rm) << shl) : (c.low << shm);
    const int range_c = islps ? (lps << shl) : (rm << shm);
    const int sh_c = islps ? shl : shm;
    const unsigned low_b = (c.low << nb) + (unsigned)hv_mul24(c.range, bv);      /* range < 2^9, bins < 2^8: the full-rate 24-bit multiply */
    c.low = isb ? low_b : low_c;
    c.range = isb ? c.range : range_c;
    c.nbits -= isb ? nb : sh_c;
    hv_cod_carry_rec(c, sink);
}
static HD u8 *hv_tok_ctx_ptr(u8 *cx, int tok) { return cx + ((tok & HV_TOK_BYP) ? (HEVCE_NCTXP - 1) * HV_TL : (tok & 0x1fff)); }   /* bypass: pad context */

/* Token j of chunk pair (cur, nxt): j < 8 from cur, else from nxt. */
static HD int hv_chunk_tok(const HvChunk &cur, const HvChunk &nxt, int j) {
    const unsigned d = (j < 8) ? cur[(j & 7) >> 1] : nxt[(j & 7) >> 1];
    return (int)((d >> ((j & 1) << 4)) & 0xffffu);
}
/* One chunk (8 tokens) through the coder, software-pipelined two tokens deep so that no LDS latency is exposed:
   while token j goes through the coder registers (hv_engine_step, pure ALU), the context byte of token j+2 and the table
   entry of token j+1 are in flight.  A = token j (context byte and table entry ready), B = token j+1 (raw context byte
   read BEFORE token j's write-back: patched when both use the same context).  LDS executes in order, so a read issued
   after a write sees it.  On return A / B describe tokens 8 and 9, i.e. the first two of `nxt`. */
struct HvTokA { int tok, cv; u8 *cp; HvCt ct; };
struct HvTokB { int tok, cvl; u8 *cp; };
template <typename SinkP> static HD void hv_engine_chunk(const HvLds &S, HvCod &c, u8 *cx, const HvChunk &cur, const HvChunk &nxt, HvTokA &A, HvTokB &B, int nvalid, SinkP sink) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int ncv = hv_engine_next_cv(A.tok, A.cv, A.ct);
        *A.cp = (u8)ncv;
        const int cvb = (B.cp == A.cp) ? ncv : B.cvl;
        const HvCt ctb = { S.ctab[cvb][0], S.ctab[cvb][1] };
        int tc = hv_chunk_tok(cur, nxt, j + 2);
        tc = (j + 2 < nvalid) ? tc : HV_TOK_NOP;
        u8 *cpc = hv_tok_ctx_ptr(cx, tc);
        const int cvlc = *cpc & 0x7f;
        hv_engine_step(c, A.cv, A.ct, A.tok, sink);
        A.tok = B.tok; A.cv = cvb; A.cp = B.cp; A.ct = ctb;
        B.tok = tc; B.cvl = cvlc; B.cp = cpc;
    }
}
static HD void hv_engine_prime(const HvLds &S, u8 *cx, const HvChunk &first, HvTokA &A, HvTokB &B, int nvalid) {
    A.tok = nvalid > 0 ? (int)(first[0] & 0xffffu) : HV_TOK_NOP;
    A.cp = hv_tok_ctx_ptr(cx, A.tok);
    A.cv = *A.cp & 0x7f;
    A.ct.lps4 = S.ctab[A.cv][0]; A.ct.nxt = S.ctab[A.cv][1];
    B.tok = nvalid > 1 ? (int)(first[0] >> 16) : HV_TOK_NOP;
    B.cp = hv_tok_ctx_ptr(cx, B.tok);
    B.cvl = *B.cp & 0x7f;
}

/* contiguous stream of n tokens at sp (16-byte chunks; the NxN sub-search's streams in LDS), followed by at least 8 no-op
   tokens so that the last chunk needs no range checks */
template <typename ChunkPtr, typename SinkP>
static HD void hv_engine(const HvLds &S, HvCod &c, u8 *cx, ChunkPtr sp, int n, SinkP sink) {
    if (n <= 0) return;
    HvChunk nxt = sp[0];
    HvTokA A; HvTokB B;
    hv_engine_prime(S, cx, nxt, A, B, 8);
    for (int base = 0; base < n; base += 8) {
        const HvChunk cur = nxt;
        nxt = sp[(base >> 3) + 1];                                /* requested eight tokens ahead of its first use */
        hv_engine_chunk(S, c, cx, cur, nxt, A, B, 16, sink);
    }
}

/* The same engine over a candidate's SLOTS in HBM: ls[0..nlist) lists its non-empty slots in coding order as
   (slot index | chunks << 8); (k, ci) = list entry and chunk inside that slot where work continues -- together with the
   coder registers that is all the state, so a candidate can be advanced a few chunks at a time.  Slots are whole chunks
   (padded with no-op tokens), so no token needs a range check.  `dead` on entry / return: abandoned (see hv_engine).
   shared == nullptr: run the candidate to its end.
   shared != nullptr: a phase in which 8x8 candidates (notify = shared) run to their end and the candidates of the enclosing
   CUs (notify == nullptr) ride along for as long as that takes: *shared counts the 8x8 engines still running in the whole
   workgroup, and EVERY lane stays in the loop until it reaches zero; a lane that is done signals once and then idles.
   The signal is ordinary loop-body code followed by further iterations, so it is issued when the lane finishes and not
   when the slowest lane of its wavefront does -- the parents' lanes in that very wavefront are waiting for it. */
static HD int hv_engine_slots(const HvLds &S, HvCod &c, u8 *cx, const HV_GLOBAL u16 *slots, const HV_LDSP u16 *ls, int nlist, int &k, int &ci,
                              int dead, int *shared, int notify, int q, int sse, int bits0, int *best, HV_GLOBAL u16 *sink) {
    int e = ls[hv_min(k, nlist - 1)];
    int nch = e >> 8;
    const HV_GLOBAL HvChunk *ptr = (const HV_GLOBAL HvChunk *)(slots + (e & 0xff) * HV_SLOT + HV_SLOT_TOK0) + ci;
    HvChunk nxt = *ptr;
    int it = 0, pub = 0;
    HvTokA A; HvTokB B;
    hv_engine_prime(S, cx, nxt, A, B, 8);
    for (;;) {
        const int done = dead || k >= nlist;
        if (done && !pub) {                                       /* once, at the moment the lane finishes */
            pub = 1;
            if (!dead && best && !c.zf) hv_atomic_min(best, hv_rdcost(q, sse, hv_cod_bits(c) - bits0));   /* exact (see hv_cod_span): tightens everyone's pruning bound */
            if (notify) hv_atomic_dec(shared);
        }
        if (shared ? (hv_poll(shared) <= 0) : done) break;
        if (!done) {
            const HvChunk cur = nxt;
            ci++;                                                 /* where the chunk after this one lives; request it now */
            const int adv = ci >= nch;
            k += adv;
            e = ls[hv_min(k, nlist - 1)];
            ci = adv ? 0 : ci;
            nch = adv ? (e >> 8) : nch;
            ptr = adv ? (const HV_GLOBAL HvChunk *)(slots + (e & 0xff) * HV_SLOT + HV_SLOT_TOK0) : ptr + 1;
            nxt = *ptr;
            hv_engine_chunk(S, c, cx, cur, nxt, A, B, 16, sink);
            it++;
            if (best && !(it & 1) && hv_rdcost(q, sse, hv_cod_bits(c) - bits0) > hv_peek(best)) dead = 1;
        }
    }
    return dead;
}

/* --------------------------------------------------------------------------------------------- */
/* FAST engines: the RANGE HALF of the arithmetic coder only.                                                                */
/*                                                                                                                            */
/* What a trial encode contributes to a decision is its bit count, CABAClen(trial) - CABAClen(entry)  [ref :835-837, 1438].   */
/* Every renormalisation shift takes a bit out of `nbits` and every finished byte puts eight back while counting a byte, so    */
/* that difference is  (sum of the renormalisation shifts) + (number of bypass bins) + 8 x (emulation-prevention bytes put     */
/* during the trial)  [ref :821-832, 859-879, 899-933].  The shifts depend on `range` and the context states alone -- `low`,   */
/* the finished bytes and the 0xff / 0x00 bookkeeping only matter through the third term, which is zero except once in a few   */
/* million bytes.  So the trial engines carry (range, shift count) and the contexts, NOTHING else: no `low`, no byte records,  */
/* no bypass tokens (their bins are counted at binarisation).  Costs computed this way are lower bounds of the reference's,     */
/* equal to them for every candidate without an emulation-prevention byte.  The decision taken is the reference's whenever the  */
/* WINNER has none: it then has the reference's cost, every other candidate has at least its own bound, and among equal bounds  */
/* the later candidate was taken as the reference does.  The stream writer (hv_wr_*) encodes the chosen syntax exactly and      */
/* reports every emulation-prevention byte it inserts, and hv_verify_nxn() does the same for the winners of the NxN sub-search  */
/* (fresh coders): a CTU in which one occurred is analysed again with the EXACT engines above (hv_analyse_image).               */
/* --------------------------------------------------------------------------------------------- */
struct HvFCod { int range, nbits; };                         /* nbits only ever goes down: bits so far = start - nbits */
struct HvFA { int tok; u8 *cp; unsigned lps4, nx; };         /* token j  : table entry of (context byte, bin) fetched */
struct HvFB { int tok; u8 *cp; int cvl; };                   /* token j+1: raw context byte, read BEFORE token j's write-back */
/* One chunk (8 bins) through the range coder, software-pipelined like hv_engine_chunk: while bin j goes through the registers
   the context byte of bin j+2 and the table entry of bin j+1 are in flight.  Per bin: the LPS range is a byte of the entry
   picked by the range quarter (one v_perm_b32: range >> 6 is 4..7, which selects a byte of its first operand), the new range
   is `lps` or `range - lps` shifted left until bit 8 is set again  [ref :914-933]. */
static HD void hv_fengine_chunk(const HvLds &S, HvFCod &c, u8 *cx, const HvChunk &cur, const HvChunk &nxt, HvFA &A, HvFB &B) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int ncv = (int)(A.nx & 0xff);
        *A.cp = (u8)ncv;
        const int cvb = (B.cp == A.cp) ? ncv : B.cvl;
        const int ib = (cvb << 1) | ((B.tok >> 14) & 1);
        const unsigned bl = S.ctab2[ib][0], bn = S.ctab2[ib][1];
        const int tc = hv_chunk_tok(cur, nxt, j + 2);
        u8 *cpc = cx + (tc & 0x1fff);
        const int cvlc = *cpc;
        const int lps = (int)hv_perm(A.lps4, 0u, (unsigned)c.range >> 6);
        const int rm = c.range - lps;
        const int pre = (A.nx >> 8) ? lps : rm;
        const int sh = hv_clz_nz((unsigned)pre) - 23;
        c.range = pre << sh;
        c.nbits -= sh;
        A.tok = B.tok; A.cp = B.cp; A.lps4 = bl; A.nx = bn;
        B.tok = tc; B.cp = cpc; B.cvl = cvlc;
    }
}
static HD void hv_fengine_prime(const HvLds &S, u8 *cx, const HvChunk &first, HvFA &A, HvFB &B) {
    A.tok = (int)(first[0] & 0xffffu);
    A.cp = cx + (A.tok & 0x1fff);
    const int ia = (*A.cp << 1) | ((A.tok >> 14) & 1);
    A.lps4 = S.ctab2[ia][0]; A.nx = S.ctab2[ia][1];
    B.tok = (int)(first[0] >> 16);
    B.cp = cx + (B.tok & 0x1fff);
    B.cvl = *B.cp;
}
/* contiguous stream of n bins at sp, followed by at least 8 no-op bins (the NxN sub-search's streams in LDS) */
template <typename ChunkPtr>
static HD void hv_fengine(const HvLds &S, HvFCod &c, u8 *cx, ChunkPtr sp, int n) {
    if (n <= 0) return;
    HvChunk nxt = sp[0];
    HvFA A; HvFB B;
    hv_fengine_prime(S, cx, nxt, A, B);
    for (int base = 0; base < n; base += 8) {
        const HvChunk cur = nxt;
        nxt = sp[(base >> 3) + 1];
        hv_fengine_chunk(S, c, cx, cur, nxt, A, B);
    }
}
/* over a candidate's slots in HBM; same contract as hv_engine_slots.  bits so far = bitsbase - c.nbits (bitsbase = the entry
   snapshot's nbits + the candidate's bypass bins, which are all known beforehand and so sharpen the pruning from the start) */
static HD int hv_fengine_slots(const HvLds &S, HvFCod &c, u8 *cx, const HV_GLOBAL u16 *slots, const HV_LDSP u16 *ls, int nlist, int &k, int &ci,
                               int dead, int *shared, int notify, int q, int sse, int bitsbase, int *best) {
    int e = ls[hv_min(k, nlist - 1)];
    int nch = e >> 8;
    const HV_GLOBAL HvChunk *ptr = (const HV_GLOBAL HvChunk *)(slots + (e & 0xff) * HEVCE_FTOKEN_SLOT_WORDS) + ci;
    HvChunk nxt = *ptr;
    int it = 0, pub = 0;
    HvFA A; HvFB B;
    hv_fengine_prime(S, cx, nxt, A, B);
    for (;;) {
        const int done = dead || k >= nlist;
        if (done && !pub) {                                       /* once, at the moment the lane finishes */
            pub = 1;
            if (!dead && best) hv_atomic_min(best, hv_rdcost(q, sse, bitsbase - c.nbits));
            if (notify) hv_atomic_dec(shared);
        }
        if (shared ? (hv_poll(shared) <= 0) : done) break;
        if (!done) {
            const HvChunk cur = nxt;
            ci++;
            const int adv = ci >= nch;
            k += adv;
            e = ls[hv_min(k, nlist - 1)];
            ci = adv ? 0 : ci;
            nch = adv ? (e >> 8) : nch;
            ptr = adv ? (const HV_GLOBAL HvChunk *)(slots + (e & 0xff) * HEVCE_FTOKEN_SLOT_WORDS) : ptr + 1;
            nxt = *ptr;
            hv_fengine_chunk(S, c, cx, cur, nxt, A, B);
            it++;
            if (best && !(it & 1) && hv_rdcost(q, sse, bitsbase - c.nbits) > hv_peek(best)) dead = 1;
        }
    }
    return dead;
}
/* single bins on the committed coder  [ref :914-933, 882-896] */
static HD void hv_fbin(const HvLds &S, HvFCod &c, u8 *cx, int bin) {
    const int v = *cx, st = v >> 1, mps = v & 1;
    const int lps = (int)((S.lps4[st] >> (((c.range >> 6) & 3) << 3)) & 0xff);
    const int islps = (bin != 0) != (mps != 0);
    const int pre = islps ? lps : c.range - lps;
    const int sh = hv_clz_nz((unsigned)pre) - 23;
    c.range = pre << sh; c.nbits -= sh;
    *cx = (u8)(islps ? (st == 0 ? (mps ^ 1) : ((S.lpsnext[st] << 1) | mps)) : (st < 62 ? v + 2 : v));
}
static HD void hv_fterminate(HvFCod &c, int bin) {
    c.range -= 2;
    if (bin) { c.range = 256; c.nbits -= 7; }
    else if (c.range < 256) { c.range <<= 1; c.nbits--; }
}
