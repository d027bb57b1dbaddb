/* hevce_dev_writer.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   the stream writer as a wavefront of the analysis workgroup (one workgroup per picture, clusters of up to four): staging, binarisation of a whole CTU, the budgeted scalar coder. */
/* --------------------------------------------------------------------------------------------- */
/* The stream writer: wavefront 4 turns the decision record of the PREVIOUS CTU into the bytes of the slice data    */
/* while the other wavefronts analyse the next one -- the reference's syntax writers and byte-emitting coder applied to the      */
/* chosen syntax only  [ref CABACput :821-832, CABACupdate :859-879, CABACputBin(s) :899-933, putTerminate :882-896,             */
/* CABACfinish :840-856, syntax :943-1340].  hv_wr_tokenise: the whole CTU binarised in one lane-parallel pass (lane = 4x4         */
/* position); hv_wr_engine: the bins through the exact coder, a budgeted slice per analysis round, as wave-uniform scalar code.   */
/* --------------------------------------------------------------------------------------------- */
/* neighbour context of the CTU being written, with the reference's line-buffer semantics  [ref :1369-1373, 1595-1600, 1634-1637]:
   CU size 32 outside the picture; mode DC outside the picture and across a CTU-row boundary */
static HD int hv_wr_cu_left(const HvLds &S, int L, int y, int x)  { return x > 0 ? 8 << (S.wr_rec.info[y >> 2][(x >> 2) - 1] & HEVCE_INFO_LOG2_MASK) : (L ? 8 << (S.wr_linfo[y >> 2] & HEVCE_INFO_LOG2_MASK) : 32); }
static HD int hv_wr_cu_above(const HvLds &S, int A, int y, int x) { return y > 0 ? 8 << (S.wr_rec.info[(y >> 2) - 1][x >> 2] & HEVCE_INFO_LOG2_MASK) : (A ? 8 << (S.wr_ainfo[x >> 2] & HEVCE_INFO_LOG2_MASK) : 32); }
static HD int hv_wr_pm_left(const HvLds &S, int L, int y, int x)  { return x > 0 ? S.wr_rec.mode[y >> 2][(x >> 2) - 1] : (L ? S.wr_lmode[y >> 2] : 1); }
static HD int hv_wr_pm_above(const HvLds &S, int y, int x)        { return y > 0 ? S.wr_rec.mode[(y >> 2) - 1][x >> 2] : 1; }
/* the writer starts on CTU `cw`: stage its record and its neighbours' CU sizes / modes in LDS.  Whole wavefront.
   The records were stored by this workgroup (barriers since); agent-scope loads so that no stale line of a record
   that was rewritten (a CTU analysed twice) can be picked up. */
static HD void hv_wr_stage(HvLds &S, const HevceImageDesc &im, int cw) {
    const int lane = hv_tid() & 63, ncx = im.W >> 5;
    const unsigned *src = (const unsigned *)(im.records + (size_t)cw * HEVCE_REC_BYTES);
    HV_LDSP unsigned *dst = (HV_LDSP unsigned *)&S.wr_rec;
    for (int i = lane; i < HEVCE_REC_BYTES / 4; i += 64) dst[i] = hv_ldg_u32(src + i);
    if (lane < 8 && (cw % ncx) > 0) {
        const HevceRecord *l = (const HevceRecord *)(im.records + (size_t)(cw - 1) * HEVCE_REC_BYTES);
        const unsigned wi = hv_ldg_u32((const unsigned *)&l->info[lane][4]);
        const unsigned wm = hv_ldg_u32((const unsigned *)&l->mode[lane][4]);
        S.wr_linfo[lane] = (u8)(wi >> 24); S.wr_lmode[lane] = (u8)(wm >> 24);
    }
    if (lane >= 8 && lane < 10 && cw >= ncx) {
        const HevceRecord *a = (const HevceRecord *)(im.records + (size_t)(cw - ncx) * HEVCE_REC_BYTES);
        const unsigned wi = hv_ldg_u32((const unsigned *)&a->info[7][(lane - 8) * 4]);
        for (int k = 0; k < 4; k++) S.wr_ainfo[(lane - 8) * 4 + k] = (u8)(wi >> (8 * k));
    }
}
/* Binarisation of the WHOLE CTU in one lane-parallel pass (the tokenisers of the trial encodes, exact tokens): lane = 4x4
   position in z-order.  A transform block covers a contiguous, aligned range of z-positions; its groups sit in that range in
   CODING order (lane base + i codes scan group ncg^2 - 1 - i), so that "coded earlier in the same block" is "lower lane" and
   the greater1 carry is two ballots away, as in hv_bin_group.  Three token slots per z-position in the picture's writer area
   (im.wr_tok): CU header (with the split flags of the CUs that begin there), block header, coefficient group; the coder then
   walks the positions in order.  Wavefront 4's own scratch tile holds the token ring, the scan-ordered levels and the counts. */
struct HvWrTile { unsigned ring[1024]; short lev[64][16]; u8 cnt[3][64]; u8 nzgrid[64]; unsigned short last[64]; };
static HDN void hv_wr_tokenise(const HevceImageDesc &im, int cw) {
    HvLds &S = HV_S;
    HV_LDSP HvWrTile &W = *(HV_LDSP HvWrTile *)&S.ws[4];
    const int lane = hv_tid() & 63, ncx = im.W >> 5;
    const int L = (cw % ncx) > 0, A = cw >= ncx;
    /* z-order position -> 4x4 coordinates inside the CTU */
    const int zy = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4), zx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4);
    const int info = S.wr_rec.info[zy][zx], n = 8 << (info & HEVCE_INFO_LOG2_MASK), lg = 3 + (info & HEVCE_INFO_LOG2_MASK);
    const int nxn = (info & HEVCE_INFO_NXN) != 0, split = (info & HEVCE_INFO_TUSPLIT) != 0;
    const int tn = (nxn || split) ? n >> 1 : n, tlg = (nxn || split) ? lg - 1 : lg, ncg = tn >> 2, ncgt = ncg * ncg;
    const int base = lane & ~(ncgt - 1), cubase = lane & ~(((n >> 2) * (n >> 2)) - 1);
    const int by = ((base >> 1) & 1) | ((base >> 2) & 2) | ((base >> 3) & 4), bx = (base & 1) | ((base >> 1) & 2) | ((base >> 2) & 4);   /* block origin */
    const int cy = ((cubase >> 1) & 1) | ((cubase >> 2) & 2) | ((cubase >> 3) & 4), cx4 = (cubase & 1) | ((cubase >> 1) & 2) | ((cubase >> 2) & 4);
    const int mode = S.wr_rec.mode[nxn ? by : cy][nxn ? bx : cx4];
    const int kind = hv_scan_kind(tn, mode);
    const int cg = ncgt - 1 - (lane - base);
    int gy, gx;
    hv_scan_cg(S, tlg, kind, cg, gy, gx);
    /* the group's levels in scan order; which groups of the CTU hold one */
    unsigned nzm = 0;
    {
        const HV_LDSP short *lev = (const HV_LDSP short *)&S.wr_rec.lev[0][0];
#pragma unroll
        for (int ii = 0; ii < 16; ii++) {
            int iy, ix;
            hv_scan_in(S, kind, ii, iy, ix);
            const int v = lev[((by + gy) * 4 + iy) * 32 + (bx + gx) * 4 + ix];
            W.lev[lane][ii] = (short)v;
            nzm |= (unsigned)(v != 0) << ii;
        }
    }
    W.nzgrid[(by + gy) * 8 + bx + gx] = (u8)(nzm != 0);
    const u64 bnz = hv_ballot(nzm != 0);                            /* coding order */
    WAVE_SYNC();
    const u64 grid = hv_ballot(W.nzgrid[lane] != 0);                /* raster order: bit y4 * 8 + x4 */
    const u64 mine = (bnz >> base) & (ncgt == 64 ? ~(u64)0 : (((u64)1 << ncgt) - 1));
    const int cbf = mine != 0;
    const int first = mine ? 63 - hv_clzll(mine & (~mine + 1)) : 0;   /* the block's first coded group = lowest lane with a level */
    const int cg_last = ncgt - 1 - first;
    if (cbf && lane == base + first) W.last[base] = (unsigned short)((cg_last << 4) | (31 - hv_clz(nzm)));
    WAVE_SYNC();
    const int last = cbf ? W.last[base] : 0;
    const u64 cgsig = grid >> (by * 8 + bx);                        /* bit gy * 8 + gx relative to the block; hv_cg_load bounds its look-ups */
    HV_GLOBAL u16 *slots = (HV_GLOBAL u16 *)im.wr_tok;
    /* ---- coefficient groups ---- */
    {
        HvCg g;
        hv_cg_load(S, g, (const HV_LDSP short *)W.lev[lane], tn, tlg, kind, cg, last, cgsig, cbf);
        const u64 bg1 = hv_ballot(g.g1own);
        const u64 before = bnz & ((((u64)1 << (lane - base)) - 1) << base);
        const int g1 = before ? (int)((bg1 >> (63 - hv_clzll(before))) & 1) : 0;
        int ch = 0;
        if (cbf && cg <= cg_last) {
            HvTokRingT<false> w = hv_tok_ring<false>(S, slots + (size_t)(2 * 64 + lane) * HV_WSLOT);
            ch = hv_cg_emit(S, w, g, g1);
        }
        W.cnt[2][lane] = (u8)ch;
    }
    /* ---- block headers: cbf_luma + last position  [ref :1038-1087] ---- */
    {
        int ch = 0;
        if (lane == base) {
            HvTokRingT<false> w = hv_tok_ring<false>(S, slots + (size_t)(64 + lane) * HV_WSLOT);
            ch = hv_tokenize_tu(S, w, tn, tlg, kind, 1, (nxn || split) ? CX_YCBF : CX_YCBF + 1, cbf, last);
        }
        W.cnt[1][lane] = (u8)ch;
    }
    /* ---- CU headers, preceded by the split_cu_flags of the 32x32 / 16x16 CUs that begin here and are split  [ref :943-1033, 1403-1404] ---- */
    {
        int ch = 0;
        if (lane == cubase) {
            const int y = cy * 4, x = cx4 * 4;
            HvTokRingT<false> w = hv_tok_ring<false>(S, slots + (size_t)lane * HV_WSLOT);
            const int inc32 = (32 > hv_wr_cu_left(S, L, 0, 0)) + (32 > hv_wr_cu_above(S, A, 0, 0));
            hv_emit(w, HV_CT(CX_SPLIT_CU + inc32, 1), lane == 0 && n < 32);
            const int inc16 = (16 > hv_wr_cu_left(S, L, y & ~15, x & ~15)) + (16 > hv_wr_cu_above(S, A, y & ~15, x & ~15));
            hv_emit(w, HV_CT(CX_SPLIT_CU + inc16, 1), (lane & 15) == 0 && n < 16);
            HvModes pm;
            for (int s4 = 0; s4 < 4; s4++) { pm.mode[s4] = 0; pm.left[s4] = 0; pm.above[s4] = 0; }
            pm.mode[0] = S.wr_rec.mode[cy][cx4]; pm.left[0] = hv_wr_pm_left(S, L, y, x); pm.above[0] = hv_wr_pm_above(S, y, x);
            if (nxn) {
                for (int s4 = 1; s4 < 4; s4++) pm.mode[s4] = S.wr_rec.mode[cy + (s4 >> 1)][cx4 + (s4 & 1)];
                pm.left[1] = pm.mode[0];                      pm.above[1] = hv_wr_pm_above(S, y, x + 4);
                pm.left[2] = hv_wr_pm_left(S, L, y + 4, x);   pm.above[2] = pm.mode[0];
                pm.left[3] = pm.mode[2];                      pm.above[3] = pm.mode[1];
            }
            const int inc = (n > hv_wr_cu_left(S, L, y, x)) + (n > hv_wr_cu_above(S, A, y, x));
            ch = hv_tokenize_cu(w, n, lg, nxn ? 2 : (split ? 1 : 0), inc, pm);
        }
        W.cnt[0][lane] = (u8)ch;
    }
    WAVE_SYNC();
    if (lane == 0) {
        int total = 0;
        for (int i = 0; i < 64; i++) total += W.cnt[0][i] + W.cnt[1][i] + W.cnt[2][i];
        S.wst.pos = 0; S.wst.ci = 0; S.wst.done = 0; S.wst.total = total; S.wst.budget = (total + 14) / 15; S.wst.used = 0;
        if (cw <= 0) S.wst.tpc = 25600;                           /* first guess: 100 ticks (about 2.2 k cycles) per chunk */
    }
}
/* The writer's coder: `budget` chunks of CTU cw's tokens, in coding order, through the exact byte-emitting coder; after the
   last one the end_of_slice_segment_flag, the coder registers of the CTU and -- last CTU of the picture -- the coder's tail.
   A serial program, so it is written as WAVE-UNIFORM code: the whole wavefront executes it, every value is a scalar (SGPRs, scalar
   ALU: it does not compete with the NxN chain for the SIMD's vector issue), and what would be table look-ups in memory --
   the 104 context bytes, the LPS ranges, the state transitions -- are lanes of four VGPRs read and written by v_readlane /
   v_writelane: no memory latency anywhere in the bin loop except the token chunks, which are fetched one ahead. */
struct HvWrS { int range; unsigned low; int nbits, pending, held, zeros, n, epb; };
static HD void hv_wrs_put(HvWrS &w, HV_GLOBAL u8 *out, int byte, int lane0) {                            /* [ref :821-832] */
    byte &= 0xff;
    if (w.zeros >= 2 && byte <= 3) { if (lane0) out[w.n] = 3; w.n++; w.zeros = 0; w.epb++; }
    if (lane0) out[w.n] = (u8)byte;
    w.n++;
    w.zeros = byte ? 0 : w.zeros + 1;
}
static HD void hv_wrs_carry(HvWrS &w, HV_GLOBAL u8 *out, int lane0) {                                    /* [ref :859-879] */
    if (w.nbits >= 12) return;
    const int lead = (int)(w.low >> (24 - w.nbits));
    w.nbits += 8;
    w.low &= 0xffffffffu >> w.nbits;
    if (lead == 0xff) { w.pending++; return; }
    if (w.pending > 0) {
        const int c = lead >> 8, fill = (0xff + c) & 0xff;
        hv_wrs_put(w, out, w.held + c, lane0);
        w.held = lead & 0xff;
        for (; w.pending > 1; w.pending--) hv_wrs_put(w, out, fill, lane0);
    } else { w.pending = 1; w.held = lead; }                     /* first byte ever: stored unmasked, as the reference does */
}
/* the CTU's tokens are through: end_of_slice_segment_flag  [ref :882-896, 1630], the coder registers for the per-CTU state table, and --
   last CTU of the picture -- the coder's tail  [ref :840-856, 1640-1641].  Wave-uniform (im: hv_uni_ptr'd by the caller) */
static HD void hv_wr_ctu_end(HvWrS &w, const HevceImageDesc &im, HV_GLOBAL u8 *out, int cw, int nctu, int lane0) {
    w.range -= 2;
    if (cw == nctu - 1) { w.low = (w.low + (unsigned)w.range) << 7; w.range = 256; w.nbits -= 7; }
    else if (w.range < 256) { w.low <<= 1; w.range <<= 1; w.nbits--; }
    hv_wrs_carry(w, out, lane0);
    int32_t *st = hv_uni_ptr(im.ctu_state);
    if (st && lane0) {
        st += (size_t)cw * 8;
        st[0] = w.range; st[1] = (int32_t)w.low; st[2] = w.nbits; st[3] = w.pending; st[4] = w.held; st[5] = w.zeros;
        st[6] = w.n; st[7] = 8 * (w.n + w.pending) + 23 - w.nbits;
    }
    if (cw == nctu - 1) {
        int fill = 0;
        if ((w.low >> (32 - w.nbits)) > 0) { hv_wrs_put(w, out, w.held + 1, lane0); w.low -= 1u << (32 - w.nbits); }
        else { if (w.pending > 0) hv_wrs_put(w, out, w.held, lane0); fill = 0xff; }
        for (; w.pending > 1; w.pending--) hv_wrs_put(w, out, fill, lane0);
        const int tail = (int)((w.low >> 8) << w.nbits);
        hv_wrs_put(w, out, tail >> 16, lane0); hv_wrs_put(w, out, tail >> 8, lane0); hv_wrs_put(w, out, tail, lane0);
        int32_t *pl = hv_uni_ptr(im.payload_len);
        if (pl && lane0) *pl = w.n;
    }
}
static HDN void hv_wr_engine(const HevceImageDesc &im_, int cw, int budget) {
    HvLds &S = HV_S;
    HvLds::WrSt &T = S.wst;
    const int lane = hv_tid() & 63, lane0 = lane == 0;
#ifdef HEVCE_EMU
    if (lane) return;                                            /* emulation: one fiber runs the wave-uniform program */
#endif
    if (hv_uni(T.done)) return;
    cw = hv_uni(cw); budget = hv_uni(budget);
    const HevceImageDesc &im = *hv_uni_ptr(&im_);
    HV_LDSP HvWrTile &W = *(HV_LDSP HvWrTile *)&S.ws[4];
    const int ncx = hv_uni(im.W) >> 5, nctu = ncx * (hv_uni(im.H) >> 5);
    HV_GLOBAL u8 *out = (HV_GLOBAL u8 *)hv_uni_ptr(im.payload);
    const HV_GLOBAL u16 *slots = (const HV_GLOBAL u16 *)hv_uni_ptr(im.wr_tok);
    HvWrS w = { hv_uni(S.wr.range), (unsigned)hv_uni((int)S.wr.low), hv_uni(S.wr.nbits), hv_uni(S.wr.nbytes), hv_uni(S.wr.bufbyte),
                hv_uni(S.wr.zeros), hv_uni(S.wr.nout), hv_uni(S.wr_epb) };
    /* lane tables: contexts 0..63 / 64..103 (the writer's column of context bytes), LPS ranges and LPS transitions per state */
    HvLaneTab cxa, cxb, tl4, tnx;
    HV_LT_LOAD(cxa, l_, S.nxn_ctx[l_][HV_TL - 1]);
    HV_LT_LOAD(cxb, l_, S.nxn_ctx[hv_min(64 + l_, HEVCE_NCTXP - 1)][HV_TL - 1]);
    HV_LT_LOAD(tl4, l_, (int)S.lps4[l_]);
    HV_LT_LOAD(tnx, l_, (int)S.lpsnext[l_]);
    int pos = hv_uni(T.pos), ci = hv_uni(T.ci);
    /* the chunk at (pos, ci), or the first one after it; pos = 192: none left */
    #define HV_WR_SEEK() do { while (pos < 192) { const int z_ = pos / 3, k_ = pos - 3 * z_; if (ci < hv_uni((int)W.cnt[k_][z_])) break; pos++; ci = 0; } } while (0)
    #define HV_WR_ADDR() ((const HV_GLOBAL HvChunk *)(slots + (size_t)((pos - 3 * (pos / 3)) * 64 + pos / 3) * HV_WSLOT + HV_SLOT_TOK0) + ci)
    HV_WR_SEEK();
    HvChunk nxt = { 0, 0, 0, 0 };
    if (pos < 192) nxt = *HV_WR_ADDR();
#ifndef HEVCE_EMU
    const unsigned long long ts_ = hv_realtime();
#endif
    const int budget0 = budget;
    while (pos < 192 && budget > 0) {
        unsigned c4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) c4[k] = (unsigned)hv_uni((int)nxt[k]);
        ci++; budget--;
        HV_WR_SEEK();
        if (pos < 192) nxt = *HV_WR_ADDR();                      /* the next chunk is on its way while this one is coded */
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int tok = (int)((c4[j >> 1] >> ((j & 1) << 4)) & 0xffffu);
            if (tok & HV_TOK_BYP) {                                                                         /* [ref :899-911] */
                const int nb = (tok >> 8) & 15;
                if (nb) { w.low = (w.low << nb) + (unsigned)w.range * (unsigned)(tok & 0xff); w.nbits -= nb; hv_wrs_carry(w, out, lane0); }
            } else {                                                                                        /* [ref :914-933] */
                const int idx = ((tok & 0x1fff) * 911) >> 16;    /* the token holds idx * HV_TL (72); 911 / 65536 = 1 / 71.94: exact for idx < 1170 */
                const int bin = (tok >> 14) & 1;
                const int va = hv_lt_get(cxa, idx & 63), vb = hv_lt_get(cxb, idx & 63);
                const int v = idx < 64 ? va : vb, st = v >> 1, mps = v & 1;
                const unsigned l4 = (unsigned)hv_lt_get(tl4, st & 63);
                const int lps = (int)((l4 >> (((w.range >> 6) & 3) << 3)) & 0xff);
                const int rm = w.range - lps;
                int nv;
                if (bin != mps) {
                    const int sh = hv_clz_nz((unsigned)lps) - 23;
                    w.low = (w.low + (unsigned)rm) << sh;
                    w.range = lps << sh;
                    w.nbits -= sh;
                    nv = st == 0 ? (mps ^ 1) : ((hv_lt_get(tnx, st & 63) << 1) | mps);
                } else {
                    nv = st < 62 ? v + 2 : v;
                    if (rm < 256) { w.low <<= 1; w.range = rm << 1; w.nbits--; } else w.range = rm;
                }
                hv_lt_put(cxa, idx & 63, idx < 64 ? nv : va);
                hv_lt_put(cxb, idx & 63, idx < 64 ? vb : nv);
                hv_wrs_carry(w, out, lane0);
            }
        }
    }
    #undef HV_WR_SEEK
    #undef HV_WR_ADDR
    int done = 0;
    if (pos >= 192) { done = 1; hv_wr_ctu_end(w, im, out, cw, nctu, lane0); }
    HV_LT_EACH(cxa, l_, v_, S.nxn_ctx[l_][HV_TL - 1] = (u8)v_);
    HV_LT_EACH(cxb, l_, v_, if (64 + l_ < HEVCE_NCTXP) S.nxn_ctx[64 + l_][HV_TL - 1] = (u8)v_);
#ifndef HEVCE_EMU
    if (lane0 && budget0 - budget >= 4) T.tpc = (int)(((hv_realtime() - ts_) << 8) / (unsigned)(budget0 - budget));
#endif
    if (lane0) {
        T.pos = pos; T.ci = ci; T.done = done; T.used += budget0 - budget;
        S.wr.range = w.range; S.wr.low = w.low; S.wr.nbits = w.nbits; S.wr.nbytes = w.pending; S.wr.bufbyte = w.held; S.wr.zeros = w.zeros; S.wr.nout = w.n; S.wr.pad = 0;
        S.wr_epb = w.epb;
    }
}
