/* hevce_dev_wrgroup.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   the stream writer as a workgroup of its own (clusters of five and more): tokens into LDS, contexts resolved in parallel, the coder over resolved tokens. */
/* --------------------------------------------------------------------------------------------- */
/* The writer WORKGROUP's coder (clusters of five and more: hv_writer_worker has a compute unit to itself).                     */
/* hv_wr_engine above is built to share a compute unit with the analysis -- scalar code, contexts in lane tables, ~60            */
/* instructions per bin, and a wavefront issues one instruction per ~4.5 cycles: 1.27 M cycles per CTU, as long as the analysis  */
/* of the CTU itself.  With eight wavefronts and the whole LDS to hand the work is cut differently:                             */
/*   1. wavefront 4 binarises the CTU as before (hv_wr_tokenise: exact tokens, 3 x 64 slots in HBM);                             */
/*   2. all wavefronts copy the slots' chunks, in coding order, into one linear token array in LDS (hv_wr2_linearise);           */
/*   3. all wavefronts RESOLVE the context-coded tokens: a context byte evolves from its own bins alone  [ref :914-933], so the     */
/*      byte a bin meets is the context's byte at the start of the segment walked through the earlier bins of the SAME context --  */
/*      64 tokens per step, one lane per token, four earlier bins per table look-up (t4); wavefront w takes the contexts with      */
/*      index = w mod 8, so no two wavefronts touch the same context (hv_wr2_resolve);                                            */
/*   4. wavefront 4 runs the byte-exact coder over the resolved tokens (hv_wr2_engine): a token names its table entry (LPS ranges,  */
/*      is-LPS), so the serial loop has no context to read, update or write -- ~20 scalar instructions per bin.                    */
/* The bytes, the registers after each CTU and the emulation-prevention count are those of hv_wr_engine, bit for bit (the tests    */
/* compare both with the oracle's).                                                                                               */
/* --------------------------------------------------------------------------------------------- */
#define HV_WR2_SEG 512                                       /* chunks of eight tokens per segment (a CTU has ~600; at most 192 x 15) */
struct HvWrRing { HevceCoderRegs wr; u8 ctx[HEVCE_NCTXP]; };
struct HvWr2 {
    HvWrRing ring[8];                                        /* what each of the last eight CTUs started from (hv_writer_worker) */
    alignas(16) u8 t4[128][32];                              /* context byte after k <= 4 bins: [byte][1 << k | bins, the first one in bit 0] */
    alignas(16) u16 raw[HV_WR2_SEG * 8];                     /* the segment's tokens as binarised */
    /* ... resolved, two words per token: a context-coded bin = { the four LPS ranges of the state it meets, is-LPS }, bypass bins =
       { count << 8 | bins, bit 31 }: the coder looks nothing up */
    alignas(16) unsigned res[HV_WR2_SEG * 8 + 16][2];
    u16 off[196];                                            /* first chunk of slot `pos` in coding order (pos = 3 z + kind); [192] = all */
};
static_assert(sizeof(HvWr2) + 16 <= 3 * sizeof(HvLds::Level), "HvWr2 lives in the candidate levels, which the writer workgroup does not use");
static HD HvWr2 &hv_wr2(HvLds &S) { return *(HvWr2 *)(((size_t)&S.lv[0] + 15) & ~(size_t)15); }   /* (its chunks are moved 16 bytes at a time) */
static HD void hv_wr2_init(HvLds &S, int tid) {
    HvWr2 &X = hv_wr2(S);
    for (int e = tid; e < 128 * 32; e += HV_NTHREADS) {
        const int idx = e & 31, k = 31 - hv_clz((unsigned)hv_max(idx, 1));
        int cv = e >> 5;
        if (cv < 126) for (int i = 0; i < k; i++) cv = hv_ctx_next(cv, (idx >> i) & 1);
        X.t4[e >> 5][idx] = (u8)cv;
    }
}
/* chunks [seg0, seg0 + HV_WR2_SEG) of the CTU's token stream, in coding order, into X.raw and X.res; returns the CTU's chunk count.
   All threads; the slots were written by wavefront 4 (hv_wr_tokenise), whose stores have left the compute unit (caller) */
static HD int hv_wr2_linearise(HvLds &S, const HevceImageDesc &im, int seg0) {
    HvWr2 &X = hv_wr2(S);
    HV_LDSP HvWrTile &W = *(HV_LDSP HvWrTile *)&S.ws[4];
    const int tid = hv_tid(), lane = tid & 63;
    if (seg0 == 0) {
        if (tid < 64) {                                          /* exclusive prefix sum of the 192 slots' chunk counts */
            const int c0 = W.cnt[0][lane], c1 = W.cnt[1][lane], c2 = W.cnt[2][lane];
            int v = c0 + c1 + c2;
            for (int d = 1; d < 64; d <<= 1) { const int t = hv_shfl(v, hv_max(lane - d, 0)); v += lane >= d ? t : 0; }
            const int ex = v - (c0 + c1 + c2);
            X.off[3 * lane] = (u16)ex; X.off[3 * lane + 1] = (u16)(ex + c0); X.off[3 * lane + 2] = (u16)(ex + c0 + c1);
            if (lane == 63) X.off[192] = (u16)v;
        }
        BLOCK_SYNC();
    }
    const int total = X.off[192];
    const HV_GLOBAL u16 *slots = (const HV_GLOBAL u16 *)im.wr_tok;
    for (int ch = tid; ch < hv_min(total - seg0, HV_WR2_SEG); ch += HV_NTHREADS) {
        const int g = seg0 + ch;
        int lo = 0, hi = 192;                                    /* the last slot with off[slot] <= g (empty slots share their successor's offset) */
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)X.off[mid] <= g) lo = mid; else hi = mid; }
        const int pos = lo, z = pos / 3, k = pos - 3 * z, ci = g - (int)X.off[pos];
        const unsigned *src = (const unsigned *)(const void *)(slots + (size_t)(k * 64 + z) * HV_WSLOT + HV_SLOT_TOK0 + ci * 8);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned v = hv_ldg_u32(src + j);
            ((unsigned *)X.raw)[ch * 4 + j] = v;
            X.res[ch * 8 + 2 * j][0] = v & 0xfffu;             X.res[ch * 8 + 2 * j][1] = 0x80000000u;       /* as bypass tokens; the resolve pass */
            X.res[ch * 8 + 2 * j + 1][0] = (v >> 16) & 0xfffu; X.res[ch * 8 + 2 * j + 1][1] = 0x80000000u;   /* overwrites the others */
        }
    }
    return total;
}
/* the context-coded tokens of X.raw[0, n) resolved into X.res; the writer's context column (S.nxn_ctx[.][HV_TL - 1]) moves on to the
   segment's end.  All wavefronts, each for the contexts of its own residue class */
static HD void hv_wr2_resolve(HvLds &S, int n) {
    HvWr2 &X = hv_wr2(S);
    const int tid = hv_tid(), wave = tid >> 6, lane = tid & 63;
    const HV_LDSP u16 *raw = (const HV_LDSP u16 *)(const void *)X.raw;
    HV_LDSP unsigned *res = (HV_LDSP unsigned *)(void *)&X.res[0][0];
    const HV_LDSP u8 *t4 = (const HV_LDSP u8 *)(const void *)&X.t4[0][0];
    for (int b0 = 0; b0 < n; b0 += 64) {
        const int i = b0 + lane, t = i < n ? raw[i] : HV_TOK_BYP;
        const int isctx = !(t & HV_TOK_BYP), idx = ((t & 0x1fff) * 911) >> 16, bin = (t >> 14) & 1;   /* (the token holds idx * HV_TL; see hv_wr_engine) */
        const int mine = isctx && (idx & 7) == wave;
        unsigned long long todo = hv_ballot(mine), grp = 0;
        if (!todo) continue;                                     /* (wave-uniform) */
        while (todo) {                                           /* which lanes hold a token of the same context as this one */
            const int f = 63 - hv_clzll(todo & (~todo + 1)), c = hv_readlane(idx, f);
            const unsigned long long m = hv_ballot(mine && idx == c);
            grp = (mine && idx == c) ? m : grp;
            todo &= ~m;
        }
        const unsigned long long binm = hv_ballot(bin);
        unsigned long long above = grp & (((unsigned long long)1 << lane) - 1);
        const int later = (grp >> lane) > 1;                     /* a later lane of the block continues this context */
        int st = mine ? S.nxn_ctx[idx][HV_TL - 1] : 0;
        while (hv_ballot(above != 0)) {                          /* the earlier bins of the context, first to last, four per look-up */
            int bits = 0, k = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int has = above != 0;
                const unsigned long long low = above & (~above + 1);
                bits |= (has && (binm & low)) ? 1 << k : 0;
                k += has;
                above &= ~low;
            }
            st = t4[st * 32 + ((1 << k) | bits)];
        }
        if (mine) {
            res[2 * i] = S.lps4[(st >> 1) & 63]; res[2 * i + 1] = (unsigned)(bin != (st & 1));
            if (!later) S.nxn_ctx[idx][HV_TL - 1] = t4[st * 32 + (2 | bin)];   /* the context after its last bin of the block */
        }
        WAVE_SYNC();
    }
}
/* byte-exact coder over the resolved tokens X.res[0, n)  [ref CABACputBin :914-933, CABACputBins :899-911, CABACupdate :859-879]; with
   `last`: then the CTU's end (hv_wr_ctu_end).  One wavefront, wave-uniform scalar code as hv_wr_engine. */
static HD void hv_wr_ctu_end(HvWrS &w, const HevceImageDesc &im, HV_GLOBAL u8 *out, int cw, int nctu, int lane0);
static HDN void hv_wr2_engine(const HevceImageDesc &im_, int cw, int n, int last) {
    HvLds &S = HV_S;
    HvWr2 &X = hv_wr2(S);
    const int lane = hv_tid() & 63, lane0 = lane == 0;
#ifdef HEVCE_EMU
    if (lane) return;                                            /* emulation: one fiber runs the wave-uniform program */
#endif
    cw = hv_uni(cw); n = hv_uni(n); last = hv_uni(last);
    const HevceImageDesc &im = *hv_uni_ptr(&im_);
    const int ncx = hv_uni(im.W) >> 5, nctu = ncx * (hv_uni(im.H) >> 5);
    HV_GLOBAL u8 *out = (HV_GLOBAL u8 *)hv_uni_ptr(im.payload);
    HvWrS w = { hv_uni(S.wr.range), (unsigned)hv_uni((int)S.wr.low), hv_uni(S.wr.nbits), hv_uni(S.wr.nbytes), hv_uni(S.wr.bufbyte),
                hv_uni(S.wr.zeros), hv_uni(S.wr.nout), hv_uni(S.wr_epb) };
    const HV_LDSP HvChunk *sp = (const HV_LDSP HvChunk *)(const void *)&X.res[0][0];
    HvChunk nx0 = sp[0], nx1 = sp[1];
    /* The reference calls CABACupdate after every bin  [ref :859-879]: when fewer than 12 free bits are left it takes the top byte (and a
       carry) off `low`.  Here `low` is 64 bits wide and the bytes are taken off after every FOURTH token (at most 4 x 8 more bits on top
       of the 21 a normalised `low` has): the same bytes in the same order -- a carry that the reference adds to the byte it holds back
       arrives here inside the byte itself -- and the same registers once the loop has caught up. */
    unsigned long long low = w.low;
    for (int base = 0; base < n; base += 4) {
        unsigned a[4], b[4];                                     /* four tokens, all of them in SGPRs before the first is coded */
        a[0] = (unsigned)hv_uni((int)nx0[0]); b[0] = (unsigned)hv_uni((int)nx0[1]); a[1] = (unsigned)hv_uni((int)nx0[2]); b[1] = (unsigned)hv_uni((int)nx0[3]);
        a[2] = (unsigned)hv_uni((int)nx1[0]); b[2] = (unsigned)hv_uni((int)nx1[1]); a[3] = (unsigned)hv_uni((int)nx1[2]); b[3] = (unsigned)hv_uni((int)nx1[3]);
        nx0 = sp[(base >> 1) + 2]; nx1 = sp[(base >> 1) + 3];    /* the next four are on their way meanwhile (the array has spare chunks behind) */
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((int)b[j] < 0) {                                                                            /* [ref :899-911] */
                const int nb = (int)(a[j] >> 8);
                low = (low << nb) + (unsigned long long)((unsigned)w.range * (a[j] & 0xffu));
                w.nbits -= nb;
            } else {                                                                                        /* [ref :914-933] */
                const int islps = (int)b[j];
                const int lps = (int)((a[j] >> (((w.range >> 6) & 3) << 3)) & 0xff);
                const int rm = w.range - lps;
                const int pre = islps ? lps : rm;
                const int sh = hv_clz_nz((unsigned)pre) - 23;    /* an MPS leaves the range in 128 .. 510: shift 0 or 1 */
                low = (low + (unsigned long long)(islps ? (unsigned)rm : 0u)) << sh;
                w.range = pre << sh;
                w.nbits -= sh;
            }
        }
        while (w.nbits < 12) {                                                                              /* [ref :859-879] */
            const int lead = (int)(low >> (24 - w.nbits));
            w.nbits += 8;
            low &= (((unsigned long long)1 << (32 - w.nbits)) - 1);
            if (lead == 0xff) { w.pending++; continue; }
            if (w.pending > 0) {
                const int c = lead >> 8, fill = (0xff + c) & 0xff;
                hv_wrs_put(w, out, w.held + c, lane0);
                w.held = lead & 0xff;
                for (; w.pending > 1; w.pending--) hv_wrs_put(w, out, fill, lane0);
            } else { w.pending = 1; w.held = lead; }             /* first byte ever: stored unmasked, as the reference does */
        }
    }
    w.low = (unsigned)low;
    if (last) hv_wr_ctu_end(w, im, out, cw, nctu, lane0);
    if (lane0) {
        S.wr.range = w.range; S.wr.low = w.low; S.wr.nbits = w.nbits; S.wr.nbytes = w.pending; S.wr.bufbyte = w.held; S.wr.zeros = w.zeros; S.wr.nout = w.n; S.wr.pad = 0;
        S.wr_epb = w.epb;
    }
}
/* the writer workgroup does a whole CTU: all threads */
static HD void hv_wr2_whole(HvLds &S, const HevceImageDesc &im, int cw) {
    const int wave = hv_tid() >> 6;
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
    unsigned long long wq_ = __builtin_amdgcn_s_memtime();
#define HV_WR2P(i) do { if (hv_tid() == 256) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[20 + (i)] += n_ - wq_; wq_ = n_; } } while (0)
#else
#define HV_WR2P(i) do { } while (0)
#endif
    if (wave == 4) {
        hv_wr_stage(S, im, cw);
        WAVE_SYNC();
        hv_wr_tokenise(im, cw);
        HV_FENCE_RELEASE();
#ifndef HEVCE_EMU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         /* the token slots have left the compute unit (they are read back at the L2) */
#endif
    }
    BLOCK_SYNC();
#ifdef HEVCE_EMU
    if (hv_tid() == 0 && getenv("HEVCE_EMU_WR")) fprintf(stderr, "wr2 ctu %d tokenised\n", cw);
#endif
    HV_WR2P(0);
    for (int seg0 = 0;; seg0 += HV_WR2_SEG) {
        const int total = hv_wr2_linearise(S, im, seg0);
        BLOCK_SYNC();
        HV_WR2P(1);
#ifdef HEVCE_EMU
        if (hv_tid() == 0 && getenv("HEVCE_EMU_WR")) fprintf(stderr, "wr2 ctu %d seg %d total %d\n", cw, seg0, total);
#endif
        const int nch = hv_min(total - seg0, HV_WR2_SEG), fin = seg0 + HV_WR2_SEG >= total;
        hv_wr2_resolve(S, nch * 8);
        BLOCK_SYNC();
        HV_WR2P(2);
        if (wave == 4) hv_wr2_engine(im, cw, nch * 8, fin);
        BLOCK_SYNC();
        HV_WR2P(3);
        if (fin) break;
    }
}

/* the writer does a whole CTU in one go (wavefront 4; the others wait at the caller's barrier) */
static HD void hv_wr_whole(HvLds &S, const HevceImageDesc &im, int cw) {
    if ((hv_tid() >> 6) == 4) {
        hv_wr_stage(S, im, cw);
        WAVE_SYNC();
        hv_wr_tokenise(im, cw);
        WAVE_SYNC();
        hv_wr_engine(im, cw, 100000);
    }
}
/* The writer workgroup of a picture (clusters of five): the CTUs' decision records -> slice data, as soon as the chain workgroup
   calls them final, reporting what it finds (HevceWrx).  It keeps the state each of its last eight CTUs started from, and goes
   back when the chain starts a new epoch.  Wavefront 4 does the coding (hv_wr_whole). */
static HD void hv_writer_worker(HvLds &S, const HevceImageDesc &im) {
    const int tid = hv_tid(), q = im.qpd6;
    if (tid == 0) { S.trace = nullptr; S.qlut = im.qlut; }
    BLOCK_SYNC();
    hv_init_tables(S, q, tid);
    BLOCK_SYNC();
    hv_init_sigtab(S, tid);
    if (tid == 0) { S.cl_fail = 0; S.cl_pick = 0; }
    BLOCK_SYNC();
    HevceWrx *X = &im.cluster->wrx;
    HvWrRing *ring = hv_wr2(S).ring;                             /* (no candidates here: the levels' LDS is the writer's, HvWr2) */
    hv_wr2_init(S, tid);
    BLOCK_SYNC();
    const int nctu = (im.W >> 5) * (im.H >> 5);
    int c = 0, epoch = 0;
    if (im.flags & 4) return;                                    /* test hook: a worker that never answers */
    if (tid == 0) hv_st_flag(&im.cluster->alive[4], 0x100 | hv_xcc_id());   /* this workgroup is resident, and on which XCD (hv_cl_alive) */
    if (tid == 0) { X->wregs[0] = S.wr; hv_publish_fence(0); hv_st_flag64((unsigned long long *)&X->wdone, 0ull); S.cl_light = 0; }
    /* hand-offs with the chain workgroup: LIGHT (hv_publish_fence) once this workgroup knows that the chain sits on its XCD -- the chain
       leaves its XCC id in alive[0] before anything else; until then, and with flag bit 4, agent-scope fences: each of those writes
       the XCD's dirty L2 lines back (measured: 1.2 MB per CTU, the analysis' trial scratch, with two such fences per CTU) */
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
    if (tid < 32) S.prof2[tid] = 0;
    BLOCK_SYNC();
    unsigned long long wp_wait = 0, wp_busy = 0, wp_n = 0, wp_t = __builtin_amdgcn_s_memtime();
#define HV_WRP(acc) do { if (tid == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - wp_t; wp_t = n_; } } while (0)
#else
#define HV_WRP(acc) do { } while (0)
#endif
    for (;;) {
        HV_WRP(wp_busy);
        if (tid == 0) {
            const unsigned long long t0 = hv_realtime();
            int pick, spins = 0;
            for (;;) {
                const unsigned long long g = hv_ld_flag64((const unsigned long long *)&X->go);
                const int e = (int)(g >> 32), cnt = (int)(unsigned)g;
                if (cnt == 0x7fffffff) { pick = 0; break; }
                if (e != epoch) { pick = 2; break; }
                if (cnt > c) { pick = 1; break; }
                if (hv_realtime() - t0 > HV_CL_TIMEOUT) { pick = 0; im.cluster->error = 4; break; }
                hv_spin_backoff(spins);
            }
            if (!S.cl_light && !(im.flags & 16)) { const int a0 = hv_ld_flag(&im.cluster->alive[0]); if (a0 && (a0 & 0xff) == hv_xcc_id()) S.cl_light = 1; }
            hv_acquire_fence(S.cl_light);
            S.cl_pick = pick;
        }
        HV_WRP(wp_wait);
        BLOCK_SYNC();
        const int pick = hv_uni(S.cl_pick);
        BLOCK_SYNC();
        if (pick == 0) break;
        if (pick == 2) {                                         /* a new epoch: back to the CTU it restarts at (if this workgroup is past it) */
            epoch++;
            const int r = hv_ldc_i32(&X->rst);
            if (c > r) {
                if (tid == 0) S.wr = ring[r & 7].wr;
                if (tid < HEVCE_NCTXP) S.nxn_ctx[tid][HV_TL - 1] = ring[r & 7].ctx[tid];
                c = r;
            }
            BLOCK_SYNC();
            if (tid == 0) {
                hv_st_flag64((unsigned long long *)&X->wbad, (unsigned long long)(unsigned)epoch << 32);
                hv_st_flag64((unsigned long long *)&X->wdone, ((unsigned long long)(unsigned)epoch << 32) | (unsigned)c);
            }
            continue;
        }
        /* CTU c: remember what it starts from, check the range against the analysis', write it */
        if (tid == 0) { ring[c & 7].wr = S.wr; S.wr_epb = 0; S.winner = S.wr.range != hv_ldc_i32(&X->crange[c & 7]); }
        if (tid < HEVCE_NCTXP) ring[c & 7].ctx[tid] = S.nxn_ctx[tid][HV_TL - 1];
        BLOCK_SYNC();
        if (im.flags & 64) hv_wr_whole(S, im, c);                /* flag bit 6: the shared-compute-unit coder on wavefront 4 (comparison) */
        else hv_wr2_whole(S, im, c);
        BLOCK_SYNC();
        if (tid == 0) {
            const int bad = (S.wr_epb > 0 && !hv_ldc_i32(&X->xflag[c & 7])) || S.winner;
            X->wregs[(c + 1) & 7] = S.wr;
            hv_publish_fence(S.cl_light);
            if (bad) {
                const unsigned long long wb = hv_ld_flag64((const unsigned long long *)&X->wbad);
                if ((int)(wb >> 32) != epoch || (unsigned)wb == 0)
                    hv_st_flag64((unsigned long long *)&X->wbad, ((unsigned long long)(unsigned)epoch << 32) | (unsigned)(c + 1) | (S.winner ? 0x40000000u : 0u));
            }
            hv_st_flag64((unsigned long long *)&X->wdone, ((unsigned long long)(unsigned)epoch << 32) | (unsigned)(c + 1));
        }
        c++;
        BLOCK_SYNC();
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
        wp_n++;
#endif
    }
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
    if (im.trace && tid == 0) { ((unsigned long long *)im.trace)[57] = wp_wait; ((unsigned long long *)im.trace)[58] = wp_busy; ((unsigned long long *)im.trace)[59] = wp_n; }
    BLOCK_SYNC();
    if (im.trace && tid < 4) ((unsigned long long *)im.trace)[60 + tid] = S.prof2[20 + tid];   /* hv_wr2_whole: tokenise, linearise, resolve, engine */
#endif
    (void)nctu;
}
