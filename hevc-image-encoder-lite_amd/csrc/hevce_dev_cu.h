/* hevce_dev_cu.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   evaluation of the CUs: binarisation of the candidates, engines, the rounds of an 8x8 CU. */
/* --------------------------------------------------------------------------------------------- */
/* evaluation of the CUs  [ref HEVCe.c:1350-1560]: one round of three phases per 8x8 CU                */
/*   hv_jobs_phase  : wavefront 0 walks the NxN chain; the job wavefronts run the block jobs of the 8x8 */
/*                    CU and of the enclosing CUs, binarise what they produced, and let prepared        */
/*                    candidates of the enclosing CUs ride in their idle time                           */
/*   hv_engine_phase: arithmetic engines; the 8x8 candidates run to completion, 16x16 / 32x32           */
/*                    candidates ride along                                                             */
/*   hv_decide_cu   : split alternative, arg-min (last of minimum), commit                              */
/* --------------------------------------------------------------------------------------------- */
static HD HV_GLOBAL u8 *hv_level_scratch(const HevceImageDesc &im, int lvl) { return (HV_GLOBAL u8 *)(im.scratch + (size_t)lvl * HEVCE_SCRATCH_LEVEL_BYTES); }
/* cluster of four: the 8x8 workgroup's candidates 0..69 of 8x8 CU number k (counted from the launch) live in set 0 or 3 by k's parity */
static HD HV_GLOBAL u8 *hv_worker8_scratch(const HevceImageDesc &im, int k) { return (HV_GLOBAL u8 *)(im.scratch + ((k & 1) ? (size_t)3 * HEVCE_SCRATCH_LEVEL_BYTES : 0)); }
static HD int hv_trial_slots(int n, int t) { const int tn = t < 35 ? n : (n >> 1); return 1 + (t < 35 ? 1 : 4) * (1 + (tn >> 2) * (tn >> 2)); }

/* a CU is entered: neighbour context  [ref :1369-1373] and empty candidate state.  Needs a barrier before use. */
static HD void hv_level_begin(HvLds &S, const HvCu cu) {
    const int tid = hv_tid(), n = cu.n;
    HvLds::Level &L = S.lv[cu.lvl];
    const int my = 1 + (cu.y >> 2), mx = 1 + (cu.x >> 2);
    if (tid == 0) {
        L.split_inc = (n > S.map_cu[my][mx - 1]) + (n > S.map_cu[my - 1][mx]);
        L.ml = S.map_pm[my][mx - 1]; L.ma = S.map_pm[my - 1][mx];
        L.best_cost = HV_I32MAX;
    }
    if (tid < HV_TL) { L.tdead[tid] = 0; L.tpos[tid] = 0; L.tlen[tid] = 0; }
    if (n == 8) for (int k = tid; k < 8 * 8; k += HV_NTHREADS) (&S.nxn_crec[0][0])[k] = 0;
    if (n == 8 && tid == 0) { S.nxn_busy = 1; S.l0_active = 71; S.c70_done = 0; S.nxq_go = 0; S.nxq_done = 0; S.nxq_tmask = 0; S.c70_rd = 0; }
}

/* chunks per slot of the candidates a wavefront is binarising: rows of HEVCE_TOKEN_SLOTS + 2 bytes, row 0 = candidate t0, in
   a piece of that wavefront's scratch tile -- the counts only live from the work items to hv_build_slist, and wavefronts
   work on different ranges of candidates (and CU levels) at the same time. */
struct HvCnt { HV_LDSP u8 *p; int t0; };
static HD HV_LDSP u8 &hv_cnt(const HvCnt &cn, int t, int f) { return cn.p[(t - cn.t0) * (HEVCE_TOKEN_SLOTS + 2) + f]; }

/* binarisation work items (slot f of candidate t: f = 0 CU header; then per transform block -- 1 for candidates 0..34, 4
   otherwise -- a header slot followed by its 4x4 groups from the highest scan index down: the order the engine consumes them)
   [ref :1434-1435, 1470-1471, 1540-1541 call the syntax writers per candidate; here only their bins are formed] */
static HD HV_LDSP u16 *hv_slist(HvLds &S, int lvl, int t) { return lvl == 0 ? (HV_LDSP u16 *)S.slist0[t] : lvl == 1 ? (HV_LDSP u16 *)S.slist1[t] : (HV_LDSP u16 *)S.slist2[t]; }
static HD const HV_LDSP u16 *hv_slist(const HvLds &S, int lvl, int t) { return lvl == 0 ? (const HV_LDSP u16 *)S.slist0[t] : lvl == 1 ? (const HV_LDSP u16 *)S.slist1[t] : (const HV_LDSP u16 *)S.slist2[t]; }

/* Group item number `item` (candidate-major, coding order inside a transform block).  EVERY lane of the wavefront calls
   this (valid = 0 beyond the last item): the groups of one transform block sit in adjacent lanes, aligned to their
   count, so the "previous coded group had a level > 1" memory  [ref :1222-1226] is two wave-wide ballots away. */
template <bool FAST>
static HD void hv_bin_group(HvLds &S, HvLds::Level &L, const HvCu &cu, HV_GLOBAL u16 *gtok, HvGLev glev, const HvCnt &cn, int item, int valid) {
    const int ncg_cu = (cu.n >> 2) * (cu.n >> 2);
    item = valid ? item : 0;
    const int t = item / ncg_cu, gi = item % ncg_cu;
    const int kind = t < 35 ? 0 : 1;                             /* one block of the CU size / four of half the size */
    const int tn = kind == 0 ? cu.n : (cu.n >> 1), tlg = kind == 0 ? cu.lg : cu.lg - 1, ncgt = (tn >> 2) * (tn >> 2);
    const int tu = gi / ncgt, r = gi % ncgt, cg = ncgt - 1 - r;
    const int cbf = valid && L.tinfo[t][tu].cbf;
    HvCg g;
    hv_cg_load(S, g, (const HV_GLOBAL short *)(glev + t * 1024 + tu * tn * tn + cg * 16), tn, tlg,
               hv_scan_kind(tn, kind == 0 ? t : t - 35), cg, L.tinfo[t][tu].last, L.tinfo[t][tu].cgsig, cbf);
    const u64 bnz = hv_ballot(g.nzm != 0), bg1 = hv_ballot(g.g1own);
    const int lane = hv_tid() & 63;
    const u64 before = bnz & ((((u64)1 << (lane & (ncgt - 1))) - 1) << (lane & ~(ncgt - 1)));   /* coded earlier, same block */
    const int g1 = before ? (int)((bg1 >> (63 - hv_clzll(before))) & 1) : 0;
    if (!valid) return;
    HV_LDSP u8 &sh = hv_cnt(cn, t, 1 + tu * (1 + ncgt) + 1 + r);
    if (tn == 4) {
        /* a 4x4 block is one group: its header (cbf_luma, last position) goes into the group's slot, in front of the group's
           tokens, and the header slot stays empty -- one padded slot less per block for the engines to walk (TU-split 8x8
           candidates: five slots instead of nine; they are the longest of an 8x8 CU's candidates) */
        HvTokRingT<FAST> w4 = hv_tok_ring<FAST>(S, gtok + ((size_t)t * HV_SLOTS_TRIAL + 1 + tu * 2 + 1) * HvSlotGeo<FAST>::words, HvSlotGeo<FAST>::tok0);
        hv_tokenize_tu_body<3>(S, w4, 4, 2, hv_scan_kind(4, t - 35), 1, CX_YCBF, cbf, L.tinfo[t][tu].last);
        sh = (u8)hv_cg_emit(S, w4, g, g1);
        hv_cnt(cn, t, 1 + tu * 2) = 0;
        if (FAST) hv_atomic_add(&L.tnlog[t], w4.byp);
        return;
    }
    if (!cbf) { sh = 0; return; }
    HvTokRingT<FAST> w = hv_tok_ring<FAST>(S, gtok + ((size_t)t * HV_SLOTS_TRIAL + 1 + tu * (1 + ncgt) + 1 + r) * HvSlotGeo<FAST>::words, HvSlotGeo<FAST>::tok0);
    sh = (u8)hv_cg_emit(S, w, g, g1);
    if (FAST) hv_atomic_add(&L.tnlog[t], w.byp);
}
template <bool FAST>
static HD void hv_bin_tu_header(HvLds &S, HvLds::Level &L, const HvCu &cu, HV_GLOBAL u16 *gtok, const HvCnt &cn, int t, int tu) {
    const int kind = t < 35 ? 0 : (t < 70 ? 1 : 2);
    const int tn = kind == 0 ? cu.n : (cu.n >> 1), tlg = kind == 0 ? cu.lg : cu.lg - 1, ncgt = (tn >> 2) * (tn >> 2);
    const int md = kind == 0 ? t : kind == 1 ? t - 35 : S.nxn_modes[tu];
    HvTokRingT<FAST> w = hv_tok_ring<FAST>(S, gtok + ((size_t)t * HV_SLOTS_TRIAL + 1 + tu * (1 + ncgt)) * HvSlotGeo<FAST>::words, HvSlotGeo<FAST>::tok0);
    hv_cnt(cn, t, 1 + tu * (1 + ncgt)) = (u8)hv_tokenize_tu(S, w, tn, tlg, hv_scan_kind(tn, md), 1, kind == 0 ? CX_YCBF + 1 : CX_YCBF,
                                                         L.tinfo[t][tu].cbf, L.tinfo[t][tu].last);
    if (FAST) hv_atomic_add(&L.tnlog[t], w.byp);
}
template <bool FAST>
static HD void hv_bin_cu_header(HvLds &S, HvLds::Level &L, const HvCu &cu, HV_GLOBAL u16 *gtok, const HvCnt &cn, int t) {
    const int kind = t < 35 ? 0 : (t < 70 ? 1 : 2);
    const int my = 1 + (cu.y >> 2), mx = 1 + (cu.x >> 2);
    HvModes pm;
    for (int s4 = 0; s4 < 4; s4++) { pm.mode[s4] = 0; pm.left[s4] = 0; pm.above[s4] = 0; }
    pm.left[0] = L.ml; pm.above[0] = L.ma;
    if (kind == 0) pm.mode[0] = t;
    else if (kind == 1) pm.mode[0] = t - 35;
    else {
        for (int s4 = 0; s4 < 4; s4++) pm.mode[s4] = S.nxn_modes[s4];
        pm.left[1] = pm.mode[0];  pm.above[1] = S.map_pm[my - 1][mx + 1];                 /* [ref :1531-1538] */
        pm.left[2] = S.map_pm[my + 1][mx - 1];  pm.above[2] = pm.mode[0];
        pm.left[3] = pm.mode[2];  pm.above[3] = pm.mode[1];
    }
    HvTokRingT<FAST> w = hv_tok_ring<FAST>(S, gtok + (size_t)t * HV_SLOTS_TRIAL * HvSlotGeo<FAST>::words, HvSlotGeo<FAST>::tok0);
    hv_cnt(cn, t, 0) = (u8)hv_tokenize_cu(w, cu.n, cu.lg, kind, L.split_inc, pm);
    if (FAST) hv_atomic_add(&L.tnlog[t], w.byp);
}

/* link: the list of candidate t's non-empty slots; the engine walks it */
static HD void hv_build_slist(HvLds &S, HvLds::Level &L, const HvCnt &cn, int n, int lvl, int t, int fast) {
    const int nslots = hv_trial_slots(n, t);
    HV_LDSP u16 *ls = hv_slist(S, lvl, t);
    int k = 0;
    for (int f = 0; f < nslots; f++) {
        const int ch = hv_cnt(cn, t, f);
        ls[k] = (u16)(f | (ch << 8));
        k += (ch != 0);
    }
    L.tlen[t] = k; L.tpos[t] = 0;
    if (!fast) L.tnlog[t] = 0;                                   /* FAST: tnlog holds the candidate's bypass bins */
}

/* candidates of a 16x16 / 32x32 CU start from its entry snapshot  [ref :1423-1424]: context columns and coder registers
   (their binarisation happens in the job wavefronts, hv_bin_own) */
static HD void hv_level_contexts(HvLds &S, const HevceImageDesc &im, const HvCu cu) {
    const int tid = hv_tid();
    HvLds::Level &L = S.lv[cu.lvl];
    HV_PROF_T0();
    /* a row of the context columns is HV_TL = 72 bytes = 18 dwords of the same byte */
    for (int j = tid; j < HEVCE_NCTXP * HV_TL / 4; j += HV_NTHREADS) ((HV_LDSP unsigned *)&L.tctx[0][0])[j] = (unsigned)S.ent_ctx[cu.lvl][j / (HV_TL / 4)] * 0x01010101u;
    if (tid < 70) L.tregs[tid] = S.ent[cu.lvl];
    BLOCK_SYNC();
    HV_PROF(S, cu.lvl, 5);
}

/* Arithmetic engines of the candidates, lane = candidate.  `nw` wavefronts take part, this one is number `ew` (or -1: none);
   each carries HV_TL / nw candidates per CU level: lanes [0, per) 8x8 candidates, [per, 2 per) 16x16, [2 per, 3 per) 32x32,
   candidate = lane' * nw + ew.  mask: bit l set = level l takes part.
   shared == nullptr: everything selected runs to its end.  Otherwise *shared is a count that other lanes / wavefronts bring
   to zero, and the candidates ride along until it gets there (8x8 candidates, if selected, run to their end and are the
   ones that count it down). */
static HDN void hv_run_engines(HV_GLOBAL u8 *scratch, int q, int mask, int *shared, int nw, int ew) {
    HvLds &S = HV_S;
    const int lane = hv_tid() & 63;
    q = hv_uni(q); mask = hv_uni(mask); nw = hv_uni(nw); ew = hv_uni(ew);
    const int fast = !hv_uni(S.exact);
    /* nw > 0: `nw` wavefronts share all candidates of the levels in `mask`, this one is number ew.  nw < 0: this wavefront runs
       the -nw candidates ew, ew + 1, ... of the (one) level in mask, one per lane -- its own, straight after binarising them */
    const int per = nw > 0 ? HV_TL / nw : HV_TL;
    const int lvl = nw > 0 ? lane / per : ((mask & 1) ? 0 : (mask & 2) ? 1 : 2);
    const int t = nw > 0 ? (lane % per) * nw + ew : (lane < -nw ? ew + lane : HV_TL);
    if (ew >= 0 && lvl < 3 && ((mask >> lvl) & 1)) {
        HvLds::Level &L = S.lv[lvl];
        const int ntrials = (lvl == 0) ? 71 : 70;
        if (t < ntrials && L.tpos[t] >= 0) {
            const int pos = L.tpos[t], nlist = L.tlen[t];
            int k = pos & 0xff, ci = pos >> 8;
            HV_GLOBAL u8 *lscr = scratch + (size_t)lvl * HEVCE_SCRATCH_LEVEL_BYTES;
            const HV_GLOBAL u16 *slots = fast ? (const HV_GLOBAL u16 *)(lscr + HEVCE_SCR_FTOK) + (size_t)t * HV_SLOTS_TRIAL * HEVCE_FTOKEN_SLOT_WORDS
                                              : (const HV_GLOBAL u16 *)(lscr + HEVCE_SCR_XTOK) + (size_t)t * HV_SLOTS_TRIAL * HV_SLOT;
            int dead, cost;
            if (fast) {
                HvFCod c = { L.tregs[t].range, L.tregs[t].nbits };
                const int bitsbase = S.ent[lvl].nbits + L.tnlog[t];
                dead = hv_rdcost(q, L.tsse[t], bitsbase - c.nbits) > hv_peek(&L.best_cost);
                dead = hv_fengine_slots(S, c, &L.tctx[0][t], slots, hv_slist(S, lvl, t), nlist, k, ci, dead, shared,
                                        shared && lvl == 0, q, L.tsse[t], bitsbase, &L.best_cost);
                L.tregs[t].range = c.range; L.tregs[t].nbits = c.nbits;
                cost = hv_rdcost(q, L.tsse[t], bitsbase - c.nbits);
            } else {
                HvCod c;
                hv_cod_load(c, L.tregs[t]);
                const int bits0 = hv_regs_bits(S.ent[lvl]);
                HV_GLOBAL u16 *sink = (HV_GLOBAL u16 *)(lscr + HEVCE_SCR_SINK) + (size_t)t * HEVCE_SINK_WORDS;
                dead = hv_rdcost(q, L.tsse[t], hv_cod_bits(c) - bits0) > hv_peek(&L.best_cost);
                hv_cod_span(c);
                const int nlog = L.tnlog[t];                      /* EXACT: tnlog = byte records in the sink so far */
                dead = hv_engine_slots(S, c, &L.tctx[0][t], slots, hv_slist(S, lvl, t), nlist, k, ci, dead, shared,
                                       shared && lvl == 0, q, L.tsse[t], bits0, &L.best_cost, sink + nlog);
                L.tnlog[t] = nlog + c.nev;
                hv_cod_replay(c, (const HV_GLOBAL HvChunk *)sink, nlog);
                hv_cod_store(L.tregs[t], c);
                cost = hv_rdcost(q, L.tsse[t], hv_cod_bits(c) - bits0);
            }
            if (dead) { L.tdead[t] = 1; L.tcost[t] = HV_I32MAX; L.tpos[t] = -1; }
            else if (k >= nlist) {
                L.tcost[t] = cost;
                L.tpos[t] = -1;
                hv_atomic_min(&L.best_cost, cost);
            } else L.tpos[t] = k | (ci << 8);
        }
    }
}
/* engine phase of its own: four wavefronts, one per SIMD (the loop is a single dependent instruction stream; two engine
   wavefronts on one SIMD would take turns).  finish = 0: the 8x8 candidates run to their end, the candidates of the
   enclosing CUs ride along for as long as that takes (S.l0_active counts the 8x8 engines still running);
   finish = 1 (a 16x16 / 32x32 CU is decided): what the ride-along left runs to its end. */
static HD void hv_engine_phase(HvLds &S, const HevceImageDesc &im, int mask, int finish) {
    const int wave = hv_tid() >> 6;
    HV_PROF_T0();
    hv_run_engines((HV_GLOBAL u8 *)im.scratch, im.qpd6, mask, finish ? nullptr : &S.l0_active, 4, wave < 4 ? wave : -1);
    BLOCK_SYNC();
    HV_PROF(S, 0, 6);
}

/* A job wavefront binarises the candidates [tb, tb + tc) that its own block jobs have just produced, and builds their slot
   lists: nothing here depends on another wavefront, so it happens in the shadow of the NxN chain without any
   synchronisation.  Group items: 64 / (groups per CU) candidates per pass, lane = candidate * groups + group (the groups of
   one transform block in adjacent, aligned lanes: see hv_bin_group). */
template <bool FAST>
static HDN void hv_bin_own(HV_GLOBAL u8 *scr, HvCu cu, int tb, int tc) {
    HvLds &S = HV_S;
    const int lane = hv_tid() & 63;
    cu.lvl = hv_uni(cu.lvl); cu.n = hv_uni(cu.n); cu.lg = hv_uni(cu.lg); cu.y = hv_uni(cu.y); cu.x = hv_uni(cu.x);
    HvLds::Level &L = S.lv[cu.lvl];
    HV_GLOBAL u16 *gtok = (HV_GLOBAL u16 *)(scr + HvSlotGeo<FAST>::base);
    const int ncg = (cu.n >> 2) * (cu.n >> 2), per = 64 / ncg;
    tb = hv_uni(tb); tc = hv_uni(tc);
    /* slot counts of these (<= 16) candidates: behind the token ring in this wavefront's scratch tile */
    const HvCnt cn = { (HV_LDSP u8 *)&S.ws[hv_tid() >> 6] + 4096, tb };
    if (FAST && lane < tc) L.tnlog[tb + lane] = 0;               /* FAST: the work items add up the candidate's bypass bins here */
    HV_FENCE_RELEASE();                                          /* the jobs' levels in HBM, their results in LDS */
    WAVE_SYNC();
#ifdef HEVCE_PROF_BIN
#define HV_BINT(i) do { if (hv_tid() == 64) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[16 + 4 * cu.lvl + (i)] += n_ - bt_; bt_ = n_; } } while (0)
    unsigned long long bt_ = __builtin_amdgcn_s_memtime();
#else
#define HV_BINT(i) do { } while (0)
#endif
    for (int c0 = 0; c0 < tc; c0 += per)
        hv_bin_group<FAST>(S, L, cu, gtok, (HvGLev)scr, cn, (tb + c0 + lane / ncg) * ncg + lane % ncg, c0 + lane / ncg < tc);
    HV_BINT(0);
    for (int j = lane; j < tc * 4; j += 64) {
        const int t = tb + (j >> 2), tu = j & 3;
        if (t >= 35 ? cu.n > 8 : tu == 0) hv_bin_tu_header<FAST>(S, L, cu, gtok, cn, t, tu);   /* 4x4 blocks: header inside the group's slot (hv_bin_group) */
    }
    HV_BINT(1);
    for (int j = lane; j < tc; j += 64) hv_bin_cu_header<FAST>(S, L, cu, gtok, cn, tb + j);
    WAVE_SYNC();
    HV_BINT(2);
    for (int j = lane; j < tc; j += 64) hv_build_slist(S, L, cn, cu.n, cu.lvl, tb + j, FAST);
    HV_BINT(3);
}
