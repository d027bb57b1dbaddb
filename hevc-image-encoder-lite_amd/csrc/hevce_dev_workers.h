/* hevce_dev_workers.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   cluster mode, worker side: the 8x8 workgroup(s) with one lane per sample, the parent workgroups of the 16x16 / 32x32 CUs. */
/* --------------------------------------------------------------------------------------------- */
/* The 8x8 workgroup (cluster of four) with one lane per sample: candidates 0..69 of an 8x8 CU.                                 */
/*   TU-split candidates 35..69 (four 4x4 blocks, each predicted from the candidate's own reconstruction of the ones before):    */
/*     block trials by hv_tu4_px, four modes per pass, block after block;                                                        */
/*   single-block candidates 0..34: block trials by hv_job_single<8> as before;                                                  */
/*   binarisation of both by hv_tokgrp (lane = (candidate slot, scan position of a 4x4 group), group after group in coding        */
/*   order) as RESOLVED byte tokens -- context byte << 1 | bin, the index of the engine's table entry: the state a bin meets is    */
/*   the candidate's context byte at the group's start (its context column, L.tctx) advanced by the bins of the same context      */
/*   before it in the group, four at a time through the table t4; the column is advanced at the group's end -- and engines with    */
/*   one lane per candidate (hv_bengine) whose loop has no memory access on its dependency chain.                                 */
/* Work items (five block-trial jobs, nine TU-split slots, nine binarisation slots) are handed out by an LDS counter.            */
/* --------------------------------------------------------------------------------------------- */
/* run-time guard against the control-flow miscompilation described at hv_w8_cu's work loop: a work item must start with all 64 lanes;
   otherwise the cluster is flagged (the chain workgroup gives the picture up, the host encodes it again without clusters) */
#ifdef HEVCE_EMU
#define HV_CHECK_FULL_WAVE(im) do { } while (0)
#else
#define HV_CHECK_FULL_WAVE(im) do { if (__ballot(1) != ~0ull && (im).cluster) (im).cluster->error = 5; } while (0)
#endif
#if defined(HEVCE_W8DBG) && !defined(HEVCE_EMU)
#define HV_W8DBG(im, code) do { if ((im).trace && (hv_tid() & 63) == 0) { ((volatile int *)(im).trace)[200 + (hv_tid() >> 6)] = (code); __threadfence_system(); } } while (0)
#else
#define HV_W8DBG(im, code) do { } while (0)
#endif
struct HvW8 {                                                /* lives in S.lv[1..2]: the enclosing CUs' candidates are other workgroups' business */
    u8  t4[128][32];                                         /* context byte after k <= 4 bins: [byte][1 << k | bins], the first bin = bit 0 */
    alignas(16) u8 tok[70][176];                             /* the candidates' resolved tokens (<= 150) and at least 16 pad tokens behind */
    u16 sigmask[35][16];                                     /* per row of S.sigtab and scan position: the positions of the same context */
    u8  crec[35][64];                                        /* TU-split candidates: the reconstruction so far */
    u8  bord[36][20];                                        /* ... the reference samples of the current block: corner, left[8], above[8] (row 35: lanes without a mode) */
    short lev4[35][16];                                      /* ... its levels in scan order */
    u16 comp[35][16];                                        /* a slot's non-zero magnitudes in coding order (per TU-split mode / single candidate) */
    int ntok[70], nbyp[70];
    int work, tdone[5];
};
static_assert(sizeof(HvW8) <= 2 * sizeof(HvLds::Level), "HvW8 must fit the two unused candidate levels");
static HD HvW8 &hv_w8(HvLds &S) { return *(HvW8 *)&S.lv[1]; }
static HD void hv_w8_init(HvLds &S, int tid) {               /* once per launch (after hv_init_sigtab and a barrier) */
    HvW8 &W = hv_w8(S);
    for (int e = tid; e < 128 * 32; e += HV_NTHREADS) {
        const int idx = e & 31, k = 31 - hv_clz((unsigned)hv_max(idx, 1));
        int cv = e >> 5;
        if (cv < 126) for (int i = 0; i < k; i++) cv = hv_ctx_next(cv, (idx >> i) & 1);
        W.t4[e >> 5][idx] = (u8)cv;
    }
    for (int e = tid; e < 35 * 16; e += HV_NTHREADS) {
        const u8 *tab = S.sigtab[e >> 4];
        int m = 0;
        for (int j = 0; j < 16; j++) m |= (tab[j] == tab[e & 15]) << j;
        W.sigmask[e >> 4][e & 15] = (u16)m;
    }
}
/* context byte e after k <= 8 bins (the first one = bit 0 of bits): two table steps */
static HD int hv_t4_bits(const HvW8 &W, int e, int bits, int k) {
    const int k1 = hv_min(k, 4), k2 = k - k1;
    const int s1 = W.t4[e][(1 << k1) | (bits & ((1 << k1) - 1))];
    return W.t4[s1][(1 << k2) | ((bits >> 4) & ((1 << k2) - 1))];
}
/* ... after the bins nzm holds at the positions of mask `above`, highest position first (any number of them: four per step;
   the loop is wave-uniform) */
static HD int hv_t4_walk(const HvW8 &W, int e, unsigned above, unsigned nzm) {
    int st = e;
    while (hv_ballot(above != 0)) {
        int bits = 0, k = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int has = above != 0, pz = 31 - hv_clz(above | 1u);
            bits |= (has ? (int)((nzm >> pz) & 1) : 0) << k;
            k += has;
            above &= ~(1u << pz);
        }
        st = W.t4[st][(1 << k) | bits];
    }
    return st;
}
struct HvGrp {                                               /* what the lanes of a candidate slot carry from group to group (the same in all sixteen) */
    int cur, byp, g1;                                        /* tokens so far, bypass bins so far, the last group with levels had one above 1 */
};
/* One 4x4 coefficient group of candidate t (lane = (slot, scan position ii)): v = the lane's level, or 0 outside a coded group.
   n = 4 / 8 (transform size), cg / cg_last / last / nb as in hv_cg_load; appends the group's resolved tokens to W.tok[t], advances
   the candidate's context column.  `ok` = the slot holds a candidate.  [ref :1194-1268] */
static HD void hv_tokgrp(HvLds &S, HvW8 &W, HvLds::Level &L, HvGrp &G, int t, int ok, int v, int n, int kind, int cg, int cg_last, int lastpos, int cgs, int nb, int lane) {
    const int ii = lane & 15, s16 = lane & 48, slot4 = (hv_tid() >> 6) * 4 + (lane >> 4);
    const int coded = ok && cg <= cg_last, first_cg = cg == 0, at_last = cg == cg_last;
    v = (coded && cgs) ? v : 0;
    const int av = hv_abs(v);
    const unsigned nzm = (unsigned)(hv_ballot(v != 0) >> s16) & 0xffffu;
    const int nz = __builtin_popcount(nzm);
    const int start = at_last ? (lastpos & 15) : 15;
    u8 *col = &L.tctx[0][t];                                     /* the candidate's context column */
    u8 *tk = W.tok[t];
    #define HV_TK_DUMMY 175
    int cur = G.cur;
    /* coded_sub_block_flag */
    {
        const int on = coded && !first_cg && !at_last, ctx = CX_SIGCG + (nb != 0), e = col[ctx * HV_TL];
        if (on && ii == 0) { tk[cur] = (u8)((e << 1) | cgs); col[ctx * HV_TL] = (u8)S.ctab2[(e << 1) | cgs][1]; }
        cur += on;
    }
    const int body = coded && (cgs || first_cg);
    /* significance flags, highest scan position first; position 0 of a later group is inferred when nothing else is set */
    {
        const int any_above0 = (nzm >> 1) != 0;
        unsigned onm = (start >= 15 ? 0xffffu : ((2u << start) - 1)) & (at_last ? ~(1u << start) : 0xffffu);
        onm &= (first_cg || any_above0) ? 0xffffu : 0xfffeu;
        onm = body ? onm : 0;
        const int row = hv_sigtab_index(n, kind, nb, first_cg);
        const int ctx = CX_SIG + S.sigtab[row][ii];
        const unsigned cm = (unsigned)W.sigmask[row][ii] & onm;     /* coded positions of my context */
        const int e = col[ctx * HV_TL];
        const int on = (onm >> ii) & 1, bin = (nzm >> ii) & 1;
        const int st = hv_t4_walk(W, e, on ? (cm & ~((2u << ii) - 1)) : 0u, nzm);
        tk[on ? cur + __builtin_popcount(onm >> (ii + 1)) : HV_TK_DUMMY] = (u8)((st << 1) | bin);
        if (on && (cm & ((1u << ii) - 1)) == 0) col[ctx * HV_TL] = (u8)S.ctab2[(st << 1) | bin][1];   /* the context's last flag of the group */
        cur += __builtin_popcount(onm);
    }
    /* the non-zero magnitudes in coding order: item i of the slot goes to lane i */
    u16 *comp = W.comp[slot4];
    WAVE_SYNC();
    if (v != 0) comp[__builtin_popcount(nzm >> (ii + 1))] = (u16)av;
    WAVE_SYNC();
    const int i = ii, ai = (body && i < nz) ? comp[i] : 0;
    const unsigned bigc = (unsigned)(hv_ballot(ai > 1) >> s16) & 0xffffu, g2c = (unsigned)(hv_ballot(ai > 2) >> s16) & 0xffffu;
    const int n1 = hv_min(nz, 8), fb = __builtin_ctz((bigc & 0xffu) | 0x100u);
    /* greater1 flags of the first eight -- contexts 1, 2, 3, 3, ... of the group's set up to the first level above 1, 0 after it --
       and greater2 of that one  [ref :1217-1236] */
    {
        const int set4 = (first_cg ? 0 : 8) + (G.g1 ? 4 : 0), set1 = (first_cg ? 0 : 2) + (G.g1 ? 1 : 0);
        const int c1 = i <= fb ? hv_min(i + 1, 3) : 0, ctx = CX_ONE + set4 + c1;
        const int e = col[ctx * HV_TL], bin = ai > 1;
        const int k = c1 == 3 ? i - 2 : (c1 == 0 ? i - fb - 1 : 0);
        const int bits = c1 == 0 ? (int)(bigc >> ((fb + 1) & 15)) : 0;
        const int st = hv_t4_bits(W, e, bits, hv_max(k, 0));
        const int on = i < n1;
        tk[on ? cur + i : HV_TK_DUMMY] = (u8)((st << 1) | bin);
        /* the last flag on a context: context 1 / 2: their only one; 3: the one at min(fb, n1 - 1) if that is >= 2; 0: the last of all */
        const int lastc = c1 == 0 ? i == n1 - 1 : (c1 == 3 ? i == hv_min(fb, n1 - 1) : 1);
        if (on && lastc) col[ctx * HV_TL] = (u8)S.ctab2[(st << 1) | bin][1];
        const int e2 = col[(CX_ABS + set1) * HV_TL], b2 = ai > 2;
        if (i == fb && fb < 8) { tk[cur + n1] = (u8)((e2 << 1) | b2); col[(CX_ABS + set1) * HV_TL] = (u8)S.ctab2[(e2 << 1) | b2][1]; }
        cur += n1 + (fb < 8);
    }
    /* bypass bins: signs, remaining magnitudes with the adaptive Rice parameter (hv_tok4_px)  [ref :1238-1268] */
    {
        const int more = __builtin_popcount(bigc & 0xffu) > 1;
        const int esc = nz > 8 || more || (fb < 8 && ((g2c >> fb) & 1));
        const int base = i < 8 ? ((bigc & ((1u << i) - 1)) ? 2 : 3) : 1;
        const int rem = ai - base, on = i < nz && esc && rem >= 0;
        const unsigned e0 = (unsigned)(hv_ballot(on && ai > 3) >> s16) & 0xffffu, e1 = (unsigned)(hv_ballot(on && ai > 6) >> s16) & 0xffffu;
        const unsigned e2 = (unsigned)(hv_ballot(on && ai > 12) >> s16) & 0xffffu, e3 = (unsigned)(hv_ballot(on && ai > 24) >> s16) & 0xffffu;
        const int p0 = __builtin_ctz(e0 | 0x10000u);
        const int p1 = __builtin_ctz((e1 & ~((2u << p0) - 1)) | 0x10000u);
        const int p2 = __builtin_ctz((e2 & ~((2u << p1) - 1)) | 0x10000u);
        const int p3 = __builtin_ctz((e3 & ~((2u << p2) - 1)) | 0x10000u);
        const int rice = (i > p0) + (i > p1) + (i > p2) + (i > p3);
        G.byp += nz + hv_row_sum(on ? hv_remainder_bins(rem, rice) : 0);
    }
    G.g1 = nz ? ((bigc & 0xffu) != 0) : G.g1;
    G.cur = cur;
    #undef HV_TK_DUMMY
}
/* cbf_luma and the last position of a transform block of candidate t (lanes 0..11 of the slot write the prefix bins: x then y)
   [ref :1038-1087] */
static HD void hv_tokhdr(HvLds &S, HvW8 &W, HvLds::Level &L, HvGrp &G, int t, int ok, int n, int lg, int kind, int cbf_ctx, int cbf, int lastpos, int lane) {
    const int ii = lane & 15;
    u8 *col = &L.tctx[0][t];
    u8 *tk = W.tok[t];
    int cur = G.cur;
    {
        const int e = col[cbf_ctx * HV_TL];
        if (ok && ii == 0) { tk[cur] = (u8)((e << 1) | cbf); col[cbf_ctx * HV_TL] = (u8)S.ctab2[(e << 1) | cbf][1]; }
        cur += ok;
    }
    const int on0 = ok && cbf;
    int gy, gx, iy, ix;
    hv_scan_cg(S, lg, kind, lastpos >> 4, gy, gx);
    hv_scan_in(S, kind, lastpos & 15, iy, ix);
    const int y = gy * 4 + iy, x = gx * 4 + ix, cls = lg - 2, sh = n == 4 ? 0 : 1, gmax = 2 * lg - 1;
    const int cy_ = kind == 2 ? x : y, cx_ = kind == 2 ? y : x;
    const int ggx = hv_last_group(cx_), ggy = hv_last_group(cy_);
    const int nx = ggx < gmax ? ggx + 1 : gmax, ny = ggy < gmax ? ggy + 1 : gmax;
    {
        const int isy = ii >= 6, r = isy ? ii - 6 : ii, gg = isy ? ggy : ggx, cnt = isy ? ny : nx;
        const int ctx = (isy ? CX_LASTY : CX_LASTX) + 5 * cls + (r >> sh), bin = r < gg, on = on0 && ii < 12 && r < cnt;
        const int e = col[ctx * HV_TL], k = sh ? (r & 1) : 0;       /* 8x8: two bins per context, the first of them was a one */
        const int st = W.t4[e][k ? 3 : 1];
        if (on) {
            tk[cur + (isy ? nx : 0) + r] = (u8)((st << 1) | bin);
            if (!(sh && !(r & 1) && r + 1 < cnt)) col[ctx * HV_TL] = (u8)S.ctab2[(st << 1) | bin][1];   /* unless the context's second bin follows */
        }
    }
    cur += on0 ? nx + ny : 0;
    G.byp += on0 ? (ggx > 3 ? (ggx - 2) >> 1 : 0) + (ggy > 3 ? (ggy - 2) >> 1 : 0) : 0;
    G.cur = cur;
}
/* the CU header of candidate t: part_mode, prev_intra_luma_pred_flag (its mpm_idx / rem_intra_luma_pred_mode are bypass bins),
   intra_chroma_pred_mode, split_transform_flag, cbf_cb, cbf_cr -- one context each, cbf_cb / cbf_cr share one  [ref :943-1033,
   1282-1287].  One lane of the slot. */
static HD void hv_tokcu(HvLds &S, HvW8 &W, HvLds::Level &L, HvGrp &G, int t, int ok, int mode, int tusplit, int lane) {
    const int ii = lane & 15;
    u8 *col = &L.tctx[0][t];
    u8 *tk = W.tok[t];
    int p3[3];
    hv_mpm(L.ml, L.ma, p3);
    const int hit = (p3[2] == mode) ? 2 : (p3[1] == mode) ? 1 : (p3[0] == mode) ? 0 : -1;
    if (ok && ii == 0) {
        const int ctxs[6] = { CX_PART, CX_YPM, CX_UVPM, CX_SPLIT_TU + 2, CX_UVCBF, CX_UVCBF };
        const int bins[6] = { 1, hit >= 0, 0, tusplit, 0, 0 };
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int e = col[ctxs[j] * HV_TL];
            tk[j] = (u8)((e << 1) | bins[j]);
            col[ctxs[j] * HV_TL] = (u8)S.ctab2[(e << 1) | bins[j]][1];
        }
    }
    G.cur = ok ? 6 : 0;
    G.byp = ok ? (hit >= 0 ? 1 + (hit > 0) : 5) : 0;
    G.g1 = 0;
}
/* the engine for resolved byte tokens: n of them at sp (16-byte chunks of sixteen), at least 32 pad tokens behind */
template <typename ChunkPtr>
static HD void hv_bengine(const HvLds &S, HvFCod &c, ChunkPtr sp, int n) {
    if (n <= 0) return;
    /* sixteen tokens (one 16-byte chunk) per turn of the loop; the table entries of the NEXT four bins are fetched while four are coded,
       into the other of two register sets (no moves), across the chunk boundary too */
    unsigned al[4], an[4], bl[4], bn[4];
#define HV_BENG_LOAD(L_, N_, WORD_) do { const unsigned wd_ = (WORD_); _Pragma("unroll") for (int k = 0; k < 4; k++) { const unsigned ix = (wd_ >> (8 * k)) & 0xffu; L_[k] = S.ctab2[ix][0]; N_[k] = S.ctab2[ix][1]; } } while (0)
#define HV_BENG_RUN(L_, N_) do { _Pragma("unroll") for (int k = 0; k < 4; k++) hv_rengine_step(c, L_[k], N_[k]); } while (0)
    HvChunk cur = sp[0];
    HV_BENG_LOAD(al, an, cur[0]);
    for (int base = 0; base < n; base += 16) {
        const HvChunk nxt = sp[(base >> 4) + 1];                 /* (at least 16 pad tokens behind the last one) */
        HV_BENG_LOAD(bl, bn, cur[1]); HV_BENG_RUN(al, an);
        HV_BENG_LOAD(al, an, cur[2]); HV_BENG_RUN(bl, bn);
        HV_BENG_LOAD(bl, bn, cur[3]); HV_BENG_RUN(al, an);
        HV_BENG_LOAD(al, an, nxt[0]); HV_BENG_RUN(bl, bn);
        cur = nxt;
    }
#undef HV_BENG_LOAD
#undef HV_BENG_RUN
}
/* the engines of a slot's four candidates, right behind their binarisation and on the same wavefront (the first lane of each mode slot):
   they overlap the other wavefronts' work items -- one engine phase for all candidates behind a barrier took 16-24 k cycles of a CU's
   ~60 k with seven of eight wavefronts' lanes idle.  Results where hv_pw_stage's arg-min expects them. */
static HD void hv_w8_engine4(HvLds &S, HvW8 &W, int q, int t, int ok, int lane) {
    HvLds::Level &L = S.lv[0];
    WAVE_SYNC();                                                 /* the slot's tokens and counts are written */
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    const unsigned long long e0_ = __builtin_amdgcn_s_memtime();
#endif
    if (ok && (lane & 15) == 0) {
        HvFCod c = { S.ent[0].range, S.ent[0].nbits };
        hv_bengine(S, c, (const HV_LDSP HvChunk *)W.tok[t], W.ntok[t]);
        const int byp = W.nbyp[t];
        L.tregs[t] = S.ent[0]; L.tregs[t].range = c.range; L.tregs[t].nbits = c.nbits;
        L.tnlog[t] = byp; L.tdead[t] = 0; L.tpos[t] = -1;
        L.tcost[t] = hv_rdcost(q, L.tsse[t], S.ent[0].nbits - c.nbits + byp);
    }
    WAVE_SYNC();
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    if (lane == 0) { S.prof2[7] += __builtin_amdgcn_s_memtime() - e0_; S.prof2[6] += (unsigned long long)W.ntok[t]; }   /* (probe: all of the workgroup's slots; lane 0's candidate's tokens) */
#endif
}
/* work item: the TU-split candidates 35 + 4 slot .. 35 + 4 slot + 3 -- per 4x4 block the reference samples from the candidate's own
   reconstruction, the block trial, the block's tokens  [ref :1455-1471] */
static HD void hv_w8_split_slot(HvLds &S, const HevceImageDesc &im, HvW8 &W, const HvPx4 &K, const HvCu cu, int slot, HvGLev glev, HvGRec grec, int lane) {
    HvLds::Level &L = S.lv[0];
    const int q = im.qpd6, ii = lane & 15, m0 = 4 * slot + (lane >> 4), ok = m0 < HEVCE_NMODES, m = ok ? m0 : 1, t = 35 + (ok ? m0 : 0);
    const int kind = hv_scan_kind(4, m);
    HvGrp G;
    hv_tokcu(S, W, L, G, t, ok, m, 1, lane);
    int total = 0;
    for (int b = 0; b < 4; b++) {
        int l, lb, a, ar;
        hv_sub_avail(b, cu.L, cu.LB, cu.A, cu.AR, l, lb, a, ar);
        const int oy = (b >> 1) * 4, ox = (b & 1) * 4, ty = cu.y + oy, tx = cu.x + ox;
        /* reference samples: inside the CU the candidate's own reconstruction  [ref :196-257 for a 4x4 block].  Lanes 0..7 of the mode the
           left column downwards, 8..15 the row above: one position each (and the corner), both possible sources read and one selected */
        {
            auto px = [&](int yy, int xx) -> int {
                const int ry = yy - cu.y, rx = xx - cu.x, in = ry >= 0 && ry < 8 && rx >= 0 && rx < 8;
                const int vi = W.crec[m][hv_min(hv_max(ry, 0), 7) * 8 + hv_min(hv_max(rx, 0), 7)], vo = S.rec[yy + 1][xx + HV_RX];
                return in ? vi : vo;
            };
            const int cy_ = (l && a) ? ty - 1 : (l ? ty : ty - 1), cx_ = (l && a) ? tx - 1 : (l ? tx - 1 : tx);
            const int corner = (l || a) ? px(cy_, cx_) : 128;
            const int side = ii >= 8, i8 = ii & 7, on = side ? a : l, on2 = side ? ar : lb;
            const int k = (i8 < 4 || on2) ? i8 : 3;                               /* second four missing: the fourth one again */
            const int v = on ? px(side ? ty - 1 : ty + k, side ? tx + k : tx - 1) : corner;
            const int mb = ok ? m : 35;
            WAVE_SYNC();
            W.bord[mb][1 + ii] = (u8)v;                                           /* (left[8] at 1.., above[8] at 9..) */
            if (ii == 0) W.bord[mb][0] = (u8)corner;
            WAVE_SYNC();
        }
        const HvPx4Out o = hv_tu4_px_core(S, K, nullptr, W.bord[ok ? m : 35], q, ty, tx, m, lane);
        if (ok) {
            W.lev4[m][o.si] = (short)o.lv;
            W.crec[m][(oy + (ii >> 2)) * 8 + ox + (ii & 3)] = (u8)o.rc;
            glev[t * 1024 + b * 16 + o.si] = (short)o.lv;
            grec[t * 1024 + (oy + (ii >> 2)) * 8 + ox + (ii & 3)] = (u8)o.rc;
        }
        total += o.sse;
        WAVE_SYNC();
        const int v = ok ? W.lev4[m][ii] : 0;
        const unsigned nzm = (unsigned)(hv_ballot(v != 0) >> (lane & 48)) & 0xffffu;
        const int cbf = nzm != 0, last = cbf ? 31 - hv_clz(nzm) : 0;
        G.g1 = 0;
        hv_tokhdr(S, W, L, G, t, ok, 4, 2, kind, CX_YCBF, cbf, last, lane);
        hv_tokgrp(S, W, L, G, t, ok && cbf, v, 4, kind, 0, 0, last, cbf, 0, lane);
    }
    if (ok) {
        W.tok[t][G.cur + ii] = (u8)(HV_PADCV << 1);
        if (ii == 0) { L.tsse[t] = total; W.ntok[t] = G.cur; W.nbyp[t] = G.byp; }
    }
    hv_w8_engine4(S, W, q, t, ok, lane);
}
/* work item: the tokens of the single-block candidates 4 slot .. 4 slot + 3 (their block trials: hv_job_single<8>)  [ref :1434-1435] */
static HD void hv_w8_single_slot(HvLds &S, HvW8 &W, int q, int slot, HvGLev glev, int lane) {
    HvLds::Level &L = S.lv[0];
    const int ii = lane & 15, m0 = 4 * slot + (lane >> 4), ok = m0 < HEVCE_NMODES, t = ok ? m0 : 4 * slot;   /* (lanes without a mode: the slot's first one again -- its block trial is this workgroup's) */
    const int kind = hv_scan_kind(8, t);
    HvGrp G;
    hv_tokcu(S, W, L, G, t, ok, t, 0, lane);
    const int cbf = L.tinfo[t][0].cbf, last = L.tinfo[t][0].last, cg_last = last >> 4;
    const u64 cgsig = L.tinfo[t][0].cgsig;
    hv_tokhdr(S, W, L, G, t, ok, 8, 3, kind, CX_YCBF + 1, cbf, last, lane);
    for (int cg = 3; cg >= 0; cg--) {
        int gy, gx;
        hv_scan_cg(S, 3, kind, cg, gy, gx);
        const int cgs = (int)((cgsig >> (gy * 8 + gx)) & 1);
        const int right = gx < 1 ? (int)((cgsig >> (gy * 8 + gx + 1)) & 1) : 0, below = gy < 1 ? (int)((cgsig >> (gy * 8 + 8 + gx)) & 1) : 0;
        const int v = (ok && cbf) ? hv_ldc_i16((const short *)(glev + t * 1024 + cg * 16 + ii)) : 0;   /* written by another wavefront of this workgroup: read at the L2 */
        hv_tokgrp(S, W, L, G, t, ok && cbf, v, 8, kind, cg, cg_last, last, cgs, (below << 1) | right, lane);
    }
    if (ok) {
        W.tok[t][G.cur + ii] = (u8)(HV_PADCV << 1);
        if (ii == 0) { W.ntok[t] = G.cur; W.nbyp[t] = G.byp; }
    }
    hv_w8_engine4(S, W, q, t, ok, lane);
}
/* the two slot items as functions of their own (not inlined into the kernel's body: its register allocation is fragile enough, see
   experiments/README.md round 3) */
static HDN void hv_w8_split_slot_fn(const HevceImageDesc &im, HvCu cu, int slot, HvGLev glev, HvGRec grec) {
    HvLds &S = HV_S;
    const int lane = hv_tid() & 63;
    cu.y = hv_uni(cu.y); cu.x = hv_uni(cu.x); slot = hv_uni(slot);
    HvPx4 K;
    hv_px4_consts(S, K, lane);
    hv_w8_split_slot(S, im, hv_w8(S), K, cu, slot, glev, grec, lane);
}
static HDN void hv_w8_single_slot_fn(int q, int slot, HvGLev glev) {
    HvLds &S = HV_S;
    hv_w8_single_slot(S, hv_w8(S), hv_uni(q), hv_uni(slot), glev, hv_tid() & 63);
}
/* all 70 candidates of the 8x8 CU: results where hv_pw_stage's arg-min expects them (L.tcost / tsse / tregs / tnlog / tctx) */
/* clusters of six: which work items / candidates are 8x8 workgroup g's (items 0..4: single-block trials, 5..13: TU-split slots, 14..22: the
   single-block candidates' tokens) */
static HD int hv_w8_mine_item(int id, int g, int n8) {
    if (n8 == 2) return g == 0 ? (id < 5 || id >= 13) : (id >= 5 && id < 13);
    if (n8 == 3) return g == (id < 5 ? (id < 2 ? 0 : (id < 4 ? 1 : 2)) : (id < 14 ? (id - 5) / 3 : (id - 14 < 4 ? 0 : (id - 14 < 8 ? 1 : 2))));
    return 1;
}
static HD void hv_w8_cu(HvLds &S, const HevceImageDesc &im, const HvCu cu, HvGLev glev, HvGRec grec, int g8) {
    const int tid = hv_tid(), wave = tid >> 6, lane = tid & 63, q = im.qpd6;
    HvW8 &W = hv_w8(S);
    const int n8 = hv_n8(im);
#ifdef HEVCE_EMU
    if (tid == 0 && getenv("HEVCE_EMU_W8")) fprintf(stderr, "emu: hv_w8_cu y=%d x=%d\n", cu.y, cu.x);
#endif
    if (tid == 0) W.work = 0;
    if (tid < 5) W.tdone[tid] = 0;
    hv_level_contexts(S, im, cu);                                /* every candidate's context column = the CU's entry contexts; ends with a barrier */
    HvPx4 K;
    hv_px4_consts(S, K, lane);
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    unsigned long long w8_ = __builtin_amdgcn_s_memtime();
#define HV_W8P(i) do { if (lane == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[8 + 3 * hv_uni(wave) + (i)] += n_ - w8_; w8_ = n_; } } while (0)
#define HV_W8Q(i) do { if (tid == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[i] += n_ - w8q_; w8q_ = n_; } } while (0)
    unsigned long long w8q_ = w8_;
#else
#define HV_W8P(i) do { } while (0)
#define HV_W8Q(i) do { } while (0)
#endif
    for (;;) {
        /* (wave barriers around the one-lane regions: without them the compiler threaded the `lane == 0` region that ends an item into
           the one that fetches the next, and the next item ran with lane 0 alone -- seen on the GPU, experiments/README.md round 3) */
        WAVE_SYNC();
        int id = 0;
        if (lane == 0) id = hv_atomic_inc(&W.work);
        WAVE_SYNC();
        id = hv_bcast0(id);
        HV_CHECK_FULL_WAVE(im);
        if (id < 23 && !hv_w8_mine_item(id, g8, n8)) continue;
        HV_W8DBG(im, 1000 + id);
#if defined(HEVCE_W8DBG) && !defined(HEVCE_EMU)
        { const unsigned long long am_ = __ballot(1); if (im.trace && lane == (int)__builtin_ctzll(am_) && id < 23) { volatile int *t_ = (volatile int *)im.trace + 64 + 2 * id; t_[0] = (int)(am_ >> 32); t_[1] = (int)am_ ^ (wave << 28); __threadfence_system(); } }
#endif
        if (id >= 23) break;
        if (id < 5) {                                            /* block trials of eight single-block candidates */
            hv_job_single<8>(wave, lane, q, cu, id, glev, grec);
            HV_W8DBG(im, 2000 + id);
            HV_FENCE_RELEASE();
#ifndef HEVCE_EMU
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     /* the levels have left the compute unit (they are read at the L2) */
#endif
            WAVE_SYNC();
            hv_flag_set(&W.tdone[id], 1);                        /* (every lane stores the same value: no one-lane region here) */
            HV_W8P(0);
        } else if (id < 14) { hv_w8_split_slot_fn(im, cu, id - 5, glev, grec); HV_W8DBG(im, 2500 + id); HV_W8P(1); }
        else {
            const int slot = id - 14, j0 = (4 * slot) >> 3, j1 = hv_min(4 * slot + 3, 34) >> 3;
            HV_W8DBG(im, 3000 + id);
            while (!hv_poll(&W.tdone[j0]) || !hv_poll(&W.tdone[j1])) hv_spin_short();
            HV_FENCE_ACQUIRE();
            HV_W8DBG(im, 4000 + id);
            hv_w8_single_slot_fn(q, slot, glev);
            HV_W8DBG(im, 5000 + id);
            HV_W8P(2);
        }
    }
    HV_W8DBG(im, 6000);
    BLOCK_SYNC();
    /* (probe slots 6 / 7 now: tokens and cycles of the slots' engines, hv_w8_engine4) */
    HV_W8DBG(im, 7000);
#if defined(HEVCE_W8DBG) && !defined(HEVCE_EMU)
    if (im.trace && lane < 9 && wave * 9 + lane < 70) { ((volatile int *)im.trace)[128 + wave * 9 + lane] = W.ntok[wave * 9 + lane]; __threadfence_system(); }
#endif
#ifdef HEVCE_EMU
    if (tid == 0 && getenv("HEVCE_EMU_W8") && cu.y == 0 && cu.x == 0) { fprintf(stderr, "emu ntok/nbyp:"); for (int t_ = 0; t_ < 70; t_++) fprintf(stderr, " (%d,%d)", W.ntok[t_], W.nbyp[t_]); fprintf(stderr, "\n"); }
#endif
    /* (the candidates' engines ran with their slots: hv_w8_engine4) */
    HV_W8DBG(im, 7500);
    BLOCK_SYNC();

    HV_W8DBG(im, 7600);
}

/* parent workgroup: one stage of a CU's evaluation.  stage 0: block jobs (step 2 / step 3 of the reference's search for this CU,
   [ref :1420-1432, 1455-1468]), 1: context columns + binarisation, 2: the candidates' engines to their end, arg-min, answer. */
static HD void hv_pw_stage(HvLds &S, const HevceImageDesc &im, const HvCu cu, int stage, int lo, int hi, int slot, int g8 = 0) {   /* candidates [lo, hi), answer to res[slot]; g8: which 8x8 workgroup */
    const int tid = hv_tid(), wave = tid >> 6, lane = tid & 63, q = im.qpd6, wv = hv_uni(wave);
    HvLds::Level &L = S.lv[cu.lvl];
    HV_GLOBAL u8 *scr = cu.lvl == 0 ? hv_worker8_scratch(im, hv_uni(S.cl_ans8) + 1) : hv_level_scratch(im, cu.lvl);   /* 8x8: the CU being answered is number cl_ans8 + 1 */
    HvGLev glev = (HvGLev)scr;
    HvGRec grec = (HvGRec)(scr + HV_TL * 1024 * 2);
    if (stage == 0) {
        if (tid < HV_TL) { const int mine = tid >= lo && tid < hi && (cu.n != 8 || hv_w8_owner(tid, hv_n8(im)) == g8); L.tdead[tid] = !mine; L.tpos[tid] = mine ? 0 : -1; L.tlen[tid] = 0; }   /* candidate 70 (NxN) is never this workgroup's */
        if (tid == 0) L.best_cost = HV_I32MAX;
        if (cu.n == 8 && (!(im.flags & 8) || hv_n8(im) >= 2)) hv_w8_cu(S, im, cu, glev, grec, g8);   /* one lane per sample (flag bit 3: the round-2 path below) */
        else if (cu.n == 8) {
            /* 3 chains (16 modes each, candidates 35..69) + 5 single blocks (8 modes each, candidates 0..34): one job per wavefront,
               and everything that follows for the candidates of that job -- binarisation, engines -- in the same wavefront: nothing
               depends on another wavefront until the arg-min, so there is no barrier in between (the 8x8 answer is on the chain
               workgroup's critical path).  Gathering all engines on two wavefronts (35 lanes each) changes nothing: a wavefront's
               engine loop is bound by its own chain of LDS look-ups, not by the SIMD it shares (experiments/README.md) */
            hv_level_contexts(S, im, cu);                        /* ends with a barrier */
            const int nchain = hi > 35 ? 3 : 0;
            int tb = -1, tc = 0;
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
            unsigned long long w8_ = __builtin_amdgcn_s_memtime();
#define HV_W8T(i) do { if (lane == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[8 + 3 * wv + (i)] += n_ - w8_; w8_ = n_; } } while (0)
#else
#define HV_W8T(i) do { } while (0)
#endif
            if (wv < nchain) { hv_job_chain<4>(wave, lane, q, cu, wv, glev, grec); tb = 35 + 16 * wv; tc = hv_min(16, 70 - tb); }
            else if (lo < 35 && wv - nchain < 5) { hv_job_single<8>(wave, lane, q, cu, wv - nchain, glev, grec); tb = 8 * (wv - nchain); tc = hv_min(8, 35 - tb); }
            HV_W8T(0);
            if (tb >= 0) {
                hv_bin_own<true>(scr, cu, tb, tc);
                WAVE_SYNC();
                HV_W8T(1);
                hv_run_engines(scr, q, 1, nullptr, -tc, tb);    /* level 0 of the set at scr */
                HV_W8T(2);
            }
        } else if (cu.n == 16) {
            /* 5 chains (8 modes each) + 9 single blocks (4 modes each) over 8 wavefronts */
            if (wv < 5) hv_job_chain<8>(wave, lane, q, cu, wv, glev, grec);
            for (int j = (wv < 5 ? wv + 3 : wv - 5); j < 9; j += 8) hv_job_single<16>(wave, lane, q, cu, j, glev, grec);
        } else {
            /* 9 chains (4 modes each, the long ones) + 18 single blocks (2 modes each): wavefront 0 takes chains 0 and 8, the others a
               chain and two or three single blocks each */
            hv_job_chain<16>(wave, lane, q, cu, wv, glev, grec);
            if (wv == 0) hv_job_chain<16>(wave, lane, q, cu, 8, glev, grec);
            else for (int j = wv - 1; j < 18; j += 7) hv_job_single<32>(wave, lane, q, cu, j, glev, grec);
        }
    } else if (stage == 1) {
        if (cu.n == 8) return;                                   /* done in stage 0 */
        hv_level_contexts(S, im, cu);                            /* ends with a barrier */
        const int per = (hi - lo + 7) >> 3, tb = lo + per * wv, tc = hv_max(hv_min(per, hi - tb), 0);
        hv_bin_own<true>(scr, cu, tb, tc);
    } else {
        /* four engine wavefronts, one per SIMD (18 candidates each): the loop is one dependent instruction stream, two of them on a
           SIMD take turns (measured: 8 x 9 lanes 44.8 k cycles per 8x8 CU, 4 x 18 lanes less) */
        if (cu.n != 8) { hv_run_engines((HV_GLOBAL u8 *)im.scratch, q, 1 << cu.lvl, nullptr, 4, wave < 4 ? wave : -1); BLOCK_SYNC(); }   /* (8x8: stage 0 ended with a barrier) */
        int bc = HV_I32MAX, bi = -2;                             /* arg-min, the later candidate wins ties  [ref :1440, 1476]: every wavefront for itself (an all-reduce), no barrier */
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int t = lane + 64 * u;
            const bool ok = t < 70 && !L.tdead[hv_min(t, HV_TL - 1)];
            const int c = L.tcost[hv_min(t, HV_TL - 1)];
            const bool take = ok && (bi == -2 || c < bc || (c == bc && t > bi));
            bc = take ? c : bc; bi = take ? t : bi;
        }
        /* (inside the rows of sixteen by DPP rotations, then the four rows' results by v_readlane: six rounds of two LDS-crossbar shuffles
           each took ~0.5 k cycles more, of an answer that is on the chain's critical path at the start of a quad) */
#define HV_AM_STEP(OC, OI) do { const int oc_ = (OC), oi_ = (OI); const bool tk_ = oi_ != -2 && (bi == -2 || oc_ < bc || (oc_ == bc && oi_ > bi)); bc = tk_ ? oc_ : bc; bi = tk_ ? oi_ : bi; } while (0)
        HV_AM_STEP(hv_row_get<8>(bc), hv_row_get<8>(bi));
        HV_AM_STEP(hv_row_get<4>(bc), hv_row_get<4>(bi));
        HV_AM_STEP(hv_row_get<2>(bc), hv_row_get<2>(bi));
        HV_AM_STEP(hv_row_get<1>(bc), hv_row_get<1>(bi));
        {
            const int c0 = hv_readlane(bc, 0), i0 = hv_readlane(bi, 0), c1 = hv_readlane(bc, 16), i1 = hv_readlane(bi, 16);
            const int c2 = hv_readlane(bc, 32), i2 = hv_readlane(bi, 32), c3 = hv_readlane(bc, 48), i3 = hv_readlane(bi, 48);
            bc = c0; bi = i0;
            HV_AM_STEP(c1, i1); HV_AM_STEP(c2, i2); HV_AM_STEP(c3, i3);
        }
#undef HV_AM_STEP
        if (tid == 0) S.winner = bi;                             /* for hv_pw_advance8, behind the barrier below */
        const int w = hv_min(hv_max(hv_bcast0(bi), 0), 69);
        const int k8 = hv_uni(S.cl_ans8) + 1;                     /* 8x8: this is the answer to CU number k8 of the launch */
        HevceRes *r = &im.cluster->res[cu.lvl == 0 ? 2 + 2 * g8 + (k8 & 1) : slot];
        if (tid == 0) { r->cost = L.tcost[w]; r->winner = w; r->sse = L.tsse[w]; r->range = L.tregs[w].range; r->nbits = L.tregs[w].nbits - L.tnlog[w]; r->pad_[0] = S.cl_light ? 1 : 2; }
        if (tid < HEVCE_NCTXP) r->ctx[tid] = L.tctx[tid][w];
        hv_publish_fence(S.cl_light);                            /* the answer, and the winner's levels / samples in the level scratch */
        BLOCK_SYNC();
        if (tid == 0) {
            if (cu.lvl == 0) {
                S.cl_ans8 = k8;
                hv_st_flag64((unsigned long long *)&r->tag, ((unsigned long long)(unsigned)L.tcost[w] << 32) | ((unsigned)(k8 & 0xffffff) << 8) | (unsigned)w);
                hv_st_flag(&r->seq, k8);
            }
            else { const int sq = S.cl_seq[HV_CL_BOX(cu.lvl)] + 1; S.cl_seq[HV_CL_BOX(cu.lvl)] = sq; hv_st_flag(&r->seq, sq); }
        }
    }
    BLOCK_SYNC();
}
/* parent workgroup: a message has arrived on mailbox b: the CU, its entry state, the CTU's source samples and the
   reconstruction around it (the CU's candidates read samples outside the CU only) */
static HD HvCu hv_pw_accept(HvLds &S, const HevceImageDesc &im, int b) {
    const int tid = hv_tid();
    const HevceMsg *m = &im.cluster->msg[b];
    const int lvl = (b + 1) % 3, ncx = im.W >> 5, ctu = hv_ldc_i32(&m->ctu);
    const int y0 = (ctu / ncx) * 32, x0 = (ctu % ncx) * 32;
    HvCu cu = { hv_ldc_i32(&m->y), hv_ldc_i32(&m->x), hv_ldc_i32(&m->n), 3 + lvl, lvl, hv_ldc_i32(&m->L), hv_ldc_i32(&m->LB), hv_ldc_i32(&m->A), hv_ldc_i32(&m->AR) };
    HvLds::Level &L = S.lv[lvl];
    if (tid == 0) {
        S.ent[lvl].range = hv_ldc_i32(&m->range); S.ent[lvl].low = 0; S.ent[lvl].nbits = hv_ldc_i32(&m->nbits); S.ent[lvl].nbytes = 0; S.ent[lvl].bufbyte = 0; S.ent[lvl].zeros = 0; S.ent[lvl].nout = 0; S.ent[lvl].pad = 0;
        L.split_inc = hv_ldc_i32(&m->split_inc); L.ml = hv_ldc_i32(&m->ml); L.ma = hv_ldc_i32(&m->ma);
        S.cl_light = hv_ldc_i32(&m->pad_) == hv_xcc_id() && !(im.flags & 16);   /* the chain workgroup sits on this XCD: light hand-offs from now on (flag bit 4: never) */
        if (b == 2) {                                            /* the quad this 8x8 CU opens */
            S.q8.y = cu.y; S.q8.x = cu.x; S.q8.ctu = ctu; S.q8.ml0 = L.ml; S.q8.ma0 = L.ma; S.q8.ml2 = hv_ldc_i32(&m->ml2); S.q8.ma2 = hv_ldc_i32(&m->ma2);
            S.q8.L = hv_ldc_i32(&m->qL); S.q8.LB = hv_ldc_i32(&m->qLB); S.q8.A = hv_ldc_i32(&m->qA); S.q8.AR = hv_ldc_i32(&m->qAR);
            S.cl_seq[2] += 1;                                    /* quads accepted */
        }
    }
    if (tid < HEVCE_NCTXP) S.ent_ctx[lvl][tid] = (u8)hv_ldc_u8(&m->ctx[tid]);
    for (int p = tid; p < 1024; p += HV_NTHREADS) {                 /* source samples, edge-replicated  [ref :1620-1622] */
        const int yy = hv_min(y0 + (p >> 5), im.src_h - 1), xx = hv_min(x0 + (p & 31), im.src_w - 1);
        S.org[p >> 5][p & 31] = im.img[(size_t)yy * (size_t)im.src_stride + xx];
    }
    for (int p = tid; p < 33 * 65; p += HV_NTHREADS) {              /* rows y0-1 .. y0+31, columns x0-1 .. x0+63 of the reconstruction so far */
        const int r = p / 65, c = p % 65;
        const int yy = hv_min(hv_max(y0 - 1 + r, 0), im.H - 1), xx = hv_min(hv_max(x0 - 1 + c, 0), im.W - 1);
        S.rec[r][HV_RX - 1 + c] = (u8)hv_ldc_u8(im.rcon + (size_t)yy * im.W + xx);
    }
    BLOCK_SYNC();
    return cu;
}
/* 8x8 workgroup: CU m of the quad has been answered.  Its decision is the better of this workgroup's best candidate and the chain's NxN
   candidate, the NxN one on equal cost (it has the highest index, hv_decide_cu)  [ref :1440, 1476, 1546-1556]; the chain workgroup
   arrives at the same one with the same numbers.  The winner's samples, modes and coder state become the next CU's surroundings and
   entry state  [ref :1364-1373].  Returns 0 when the report did not come (the picture is then flagged).  All threads. */
static HD int hv_pw_advance8(HvLds &S, const HevceImageDesc &im, const HvCu cu, int m, int g8) {
    const int tid = hv_tid(), k = hv_uni(S.cl_ans8), n8 = hv_n8(im);
    HvLds::Level &L = S.lv[0];
    HevceCluster *C = im.cluster;
    if (tid == 0) {
        if (!hv_cl_wait(&C->nxn[k & 1].seq, k, &S.cl_fail, S.cl_light)) C->error = 2;
        for (int g = 0; g < n8; g++)                             /* clusters of six / seven: the other 8x8 workgroups' answers */
            if (g != g8 && !hv_cl_wait(&C->res[2 + 2 * g + (k & 1)].seq, k, &S.cl_fail, S.cl_light)) C->error = 2;
    }
    BLOCK_SYNC();
    if (hv_uni(S.cl_fail)) return 0;
    const HevceNxn *r = &C->nxn[k & 1];
    const int wm = hv_min(hv_max(S.winner, 0), 69), cm = L.tcost[wm];
    int wo = wm, co = cm, go = g8;                               /* the best answer of all, whose it is */
    for (int g = 0; g < n8; g++) if (g != g8) {
        const HevceRes *pg = &C->res[2 + 2 * g + (k & 1)];
        const int wg = hv_min(hv_max(hv_ldc_i32(&pg->winner), 0), 69), cg = hv_ldc_i32(&pg->cost);
        if (hv_w8_better(co, wo, cg, wg)) { wo = wg; co = cg; go = g; }
    }
    const int other = go != g8;                                  /* a partner's candidate is the better one */
    const HevceRes *po = &C->res[2 + 2 * go + (k & 1)];
    const int w = other ? wo : wm;
    const int win70 = hv_ldc_i32(&r->cost) <= (other ? co : cm);
    HvGRec grec = (HvGRec)(hv_worker8_scratch(im, k) + HV_TL * 1024 * 2);
    /* (what the decision reads -- the winner's cost, index -- is not what is written below: no barrier in between) */
    if (tid < 64) S.rec[cu.y + (tid >> 3) + 1][cu.x + (tid & 7) + HV_RX] = win70 ? (u8)hv_ldc_u8(&r->rec[tid]) : (other ? (u8)hv_ldc_u8((const u8 *)(grec + w * 1024 + tid)) : grec[w * 1024 + tid]);
    if (tid >= 64 && tid < 68) {
        const int s = tid - 64, md = win70 ? (hv_ldc_i32(&r->modes) >> (8 * s)) & 0xff : (w < 35 ? w : w - 35);
        S.q8.pm[2 * (m >> 1) + (s >> 1)][2 * (m & 1) + (s & 1)] = (u8)md;
    }
    if (tid == 128) {
        S.ent[0].range = win70 ? hv_ldc_i32(&r->range) : (other ? hv_ldc_i32(&po->range) : L.tregs[w].range);
        S.ent[0].nbits = win70 ? hv_ldc_i32(&r->nbits) : (other ? hv_ldc_i32(&po->nbits) : L.tregs[w].nbits - L.tnlog[w]);
    }
    if (tid >= 256 && tid < 256 + HEVCE_NCTXP) S.ent_ctx[0][tid - 256] = win70 ? (u8)hv_ldc_u8(&r->ctx[tid - 256]) : (other ? (u8)hv_ldc_u8(&po->ctx[tid - 256]) : L.tctx[tid - 256][w]);
    BLOCK_SYNC();
    if (tid == 0) {                                              /* the next CU's neighbour modes  [ref :1369-1373]; read behind the next stage's first barrier */
        const int qy = 2 * ((m + 1) >> 1), qx = 2 * ((m + 1) & 1);
        L.ml = qx > 0 ? S.q8.pm[qy][qx - 1] : (qy == 0 ? S.q8.ml0 : S.q8.ml2);
        L.ma = qy > 0 ? S.q8.pm[qy - 1][qx] : (qx == 0 ? S.q8.ma0 : S.q8.ma2);
    }
    return 1;
}
/* The parent workgroup of a picture (cluster mode, second workgroup): it answers the chain workgroup's messages until told to
   exit.  16x16 CUs are answered at once (the chain needs them four rounds later); the 32x32 CU of a CTU, whose answer is due
   at the end of the CTU, is worked on stage by stage in between. */
static HD void hv_parent_worker(HvLds &S, const HevceImageDesc &im, int serve, int g8 = 0) {   /* serve: bit 0 = the 16x16 mailbox, bit 1 = 32x32, bit 2 = 8x8 (g8: which of the 8x8 workgroups, clusters of six) */
    const int lo8 = 0, hi8 = 70, slot8 = 2;
    const int tid = hv_tid(), q = im.qpd6;
    if (tid == 0) { S.trace = nullptr; S.qlut = im.qlut; }
    BLOCK_SYNC();
    hv_init_tables(S, q, tid);
    hv_set_mode(S, 0, q, tid);
    if (tid == 0) { S.cl_seq[0] = S.cl_seq[1] = S.cl_seq[2] = 0; S.cl_ans8 = 0; S.cl_fail = 0; S.cl_pick = 0; S.cl_light = 0; S.cl_ok = 0; }
    BLOCK_SYNC();
    hv_init_sigtab(S, tid);
    BLOCK_SYNC();
    if (serve & 4) { hv_w8_init(S, tid); BLOCK_SYNC(); }
    HevceCluster *C = im.cluster;
    if (im.flags & 4) return;                                    /* test hook: a worker that never answers (as if it had not become resident) */
    if (tid == 0) hv_st_flag(&C->alive[(serve & 4) ? (g8 == 0 ? 3 : 4 + g8) : (serve == 2 ? 2 : 1)], 0x100 | hv_xcc_id());   /* this workgroup is resident, and on which XCD (hv_cl_alive); 8x8 workgroups: roles 3, 5, 6 */
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
    if (tid < 32) S.prof2[tid] = 0;
    unsigned long long wt_ = __builtin_amdgcn_s_memtime();
    BLOCK_SYNC();
#endif
    int stage32 = -1;                                            /* -1: no 32x32 CU in progress */
    HvCu c32 = { 0, 0, 32, 5, 2, 0, 0, 0, 0 };
    for (;;) {
        if (tid == 0) {
            const unsigned long long t0 = hv_realtime();
            int pick = 0, spins = 0;
            for (;;) {
                if ((serve & 4) && hv_ld_flag(&C->msg[2].seq) == S.cl_seq[2] + 1) { pick = 4; break; }   /* an 8x8 CU (or the exit message) */
                if ((serve & 1) && hv_ld_flag(&C->msg[0].seq) == S.cl_seq[0] + 1) { pick = 1; break; }   /* a 16x16 CU (or the exit message): first */
                if (stage32 >= 0) { pick = 2; break; }                                                   /* the 32x32 CU in progress */
                if ((serve & 2) && hv_ld_flag(&C->msg[1].seq) == S.cl_seq[1] + 1) { pick = 3; break; }   /* a new 32x32 CU (or the exit message) */
                if (hv_realtime() - t0 > HV_CL_TIMEOUT) { pick = -1; C->error = 2; break; }
                hv_spin_backoff(spins);
            }
            if (pick == 1 || pick == 3 || pick == 4) hv_acquire_fence(S.cl_light);
            S.cl_pick = pick;
        }
        BLOCK_SYNC();
        const int pick = hv_uni(S.cl_pick);
        BLOCK_SYNC();
        if (pick < 0) break;
        if (pick == 1 || pick == 4) {
            const int b = pick == 1 ? 0 : 2;
            if (hv_ldc_i32(&C->msg[b].kind) == 2) break;
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
            if (tid == 0 && b == 2) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[0] += n_ - wt_; wt_ = n_; S.prof2[5] += 1; }
#endif
            HvCu cu = hv_pw_accept(S, im, b);
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
            if (tid == 0 && b == 2) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[1] += n_ - wt_; wt_ = n_; }
#endif
            /* a 16x16 CU: one answer.  An 8x8 message opens a quad: its four CUs one after the other, each entered with the decision
               of the one before (hv_pw_advance8); the next message comes when the chain has decided the 16x16 CU around them */
            int alive = 1;
            for (int m8 = 0; m8 < (b == 2 ? 4 : 1) && alive; m8++) {
                if (b == 2 && m8 > 0) {
                    cu.y = S.q8.y + (m8 >> 1) * 8; cu.x = S.q8.x + (m8 & 1) * 8;
                    hv_sub_avail(m8, S.q8.L, S.q8.LB, S.q8.A, S.q8.AR, cu.L, cu.LB, cu.A, cu.AR);
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
                    if (tid == 0) S.prof2[5] += 1;
#endif
                }
                for (int st = 0; st < 3; st++) {
                    if (b == 2 && st == 1) continue;             /* an 8x8 CU's stage 0 includes its binarisation and engines */
                    hv_pw_stage(S, im, cu, st, b == 2 ? lo8 : 0, b == 2 ? hi8 : 70, b == 2 ? slot8 : b, g8);
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
                    if (tid == 0 && b == 2) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[2 + st] += n_ - wt_; wt_ = n_; }
#endif
                }
                if (b == 2 && m8 < 3) {
                    alive = hv_pw_advance8(S, im, cu, m8, g8);
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
                    if (tid == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[0] += n_ - wt_; wt_ = n_; }
#endif
                }
            }
            if (!alive) break;
        } else {
            if (pick == 3) {
                if (hv_ldc_i32(&C->msg[1].kind) == 2) break;
                c32 = hv_pw_accept(S, im, 1); stage32 = 0;               /* a 16x16 CU served in between reloads the same CTU's samples: harmless */
            }
            hv_pw_stage(S, im, c32, stage32, 0, 70, 1);
            stage32 = stage32 == 2 ? -1 : stage32 + 1;
        }
    }
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
    if (im.trace && (serve & 4) && g8 == ((im.flags >> 5) & 1) && tid < 32) ((unsigned long long *)im.trace)[80 + tid] = S.prof2[tid];   /* 8x8 worker: wait, accept, stages 0..2, messages */
#endif
}
