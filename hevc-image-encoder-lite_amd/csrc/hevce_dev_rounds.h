/* hevce_dev_rounds.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   the jobs phase of a round, the decisions of the CUs, table set-up per launch / per mode, candidate 70's rider. */
/* TU-job phase of the 8x8 CU number m (0..3) of the 16x16 CU number k16 (0..3).  Wavefront 0 walks the NxN chain, the
   longest job by far; wavefront 4, on the same SIMD, stays idle.  The other six ("job wavefronts", work slots 0..5):
     1. the 8x8 CU's eight block jobs, then the binarisation of the candidates they produced (hv_bin_own);
     2. the jobs of the enclosing CUs, which only depend on samples outside those CUs: all 14 jobs of the 16x16 CU during
        its first child (m == 0), the 27 jobs of the 32x32 CU one per slot in rounds (0,1) (0,2) (0,3) (1,1) (1,2);
     3. in the time the NxN chain still takes, the prepared candidates of the enclosing CUs ride on.
   The schedule is static and wave-uniform (SGPRs): no lane-dependent control flow may surround the jobs (wave shuffles). */
#define HV_JOBT(slot, call) call
/* Cluster of four: the 8x8 workgroup's answer arrives a few thousand cycles before the NxN chain ends.  Wavefront 4, idle in that
   mode, watches for it during the jobs phase, so that the decision does not begin with a round trip to the L2 for a flag that has
   long been set, and a second one for the cost and the winner: both come with the flag (HevceRes.tag).  Gives up when the chain
   is through. */
/* clusters of six: two 8x8 workgroups.  Workgroup 0 takes the single-block candidates 0..34 and the TU-split candidates 67..69 (the ninth
   slot of four modes), workgroup 1 the TU-split candidates 35..66; the better of their two answers -- on equal cost the higher
   candidate, [ref :1440, 1476] -- is "the 8x8 workgroup's answer" */
static HD int hv_n8(const HevceImageDesc &im) { return !im.cluster ? 1 : (im.cluster->size >= 7 ? 3 : (im.cluster->size >= 6 ? 2 : 1)); }
/* clusters of seven: three 8x8 workgroups -- a third each of the single-block candidates (0..15, 16..31, 32..34: their block trials come in
   items of eight modes, their tokens in items of four) and of the TU-split candidates (modes 0..11, 12..23, 24..34): per workgroup at
   most five trial items side by side and then at most four token items, ~28 k cycles instead of ~45 k */
static HD int hv_w8_owner(int t, int n8) {
    if (n8 == 2) return (t >= 35 && t < 67) ? 1 : 0;
    if (n8 == 3) return t < 35 ? (t < 16 ? 0 : (t < 32 ? 1 : 2)) : (t < 47 ? 0 : (t < 59 ? 1 : 2));
    return 0;
}
static HD int hv_w8_better(int ca, int wa, int cb, int wb) { return cb < ca || (cb == ca && wb > wa); }   /* answer b beats answer a */
/* (the prefetched winner lives in wavefront 1's scratch tile: in clusters of four and more that wavefront is an NxN helper, whose passes use
   no tile; the LDS image has no room of its own left in the probe builds) */
static HD HvLds::PeekD &hv_pfd(HvLds &S) { return *(HvLds::PeekD *)&S.ws[1]; }
static HD void hv_cl_peek8(HvLds &S, const HevceImageDesc &im, int lane) {
    const int k = hv_uni(S.cl_ans8) + 1, n8 = hv_n8(im);         /* the CU of this round (its decision counts it) */
    int got = 0;
    if (lane == 0) {
        while (hv_poll(&S.c70_done) < 3 && !hv_peek(&S.cl_fail)) hv_spin_long();   /* (not before the chain's last 4x4 block: the answer is of no use earlier, and the polls share an L2 channel with every hand-off) */
        for (;;) {
            int all = 1, bw = 0, bc = 0;
            for (int g = 0; g < n8; g++) {                       /* every 8x8 workgroup's answer; the best of them, on equal cost the higher candidate */
                const unsigned long long t = hv_ld_flag64((const unsigned long long *)&im.cluster->res[2 + 2 * g + (k & 1)].tag);
                all &= (((unsigned)t >> 8) & 0xffffff) == (unsigned)(k & 0xffffff);
                const int w = (int)((unsigned)t & 0xff), c = (int)(t >> 32);
                if (g == 0 || hv_w8_better(bc, bw, c, w)) { bw = w; bc = c; }
            }
            if (all) {
                hv_acquire_fence(S.cl_light);
                S.pf.w = bw; S.pf.cost = bc; S.pf.k = k;
                got = 1;
                break;
            }
            if (hv_poll(&S.c70_done) >= 5 || hv_peek(&S.cl_fail)) break;
            hv_spin_pause();
        }
    }
    WAVE_SYNC();
    if (!hv_bcast0(got)) return;
    /* ... and what the decision will want of that candidate if it beats the NxN one: its coder, contexts, samples, levels -- one round
       trip to the L2 here, while the chain winds up, instead of at the start of the decision (hv_decide8_cluster) */
    const int w = hv_min(hv_max(hv_uni(S.pf.w), 0), 69);
    const HevceRes *r = &im.cluster->res[2 + 2 * hv_w8_owner(w, n8) + (k & 1)];
    HV_GLOBAL u8 *scr = hv_worker8_scratch(im, k);
    HvLds::PeekD &D = hv_pfd(S);
    if (lane == 0) { D.range = hv_ldc_i32(&r->range); D.nbits = hv_ldc_i32(&r->nbits); }
    for (int i = lane; i < HEVCE_NCTXP; i += 64) D.ctx[i] = (u8)hv_ldc_u8(&r->ctx[i]);
    D.rec[lane] = (u8)hv_ldc_u8((const u8 *)(scr + HV_TL * 1024 * 2 + w * 1024 + lane));
    D.lev[lane] = (short)hv_ldc_i16((const short *)scr + w * 1024 + lane);
    WAVE_SYNC();
    if (lane == 0) S.pf.dk = k;
}
static HD void hv_cl_report_nxn(HvLds &S, const HevceImageDesc &im, int lane);
static HDN void hv_c70_rider(const HevceImageDesc &im, HV_GLOBAL u16 *c70, int q);
static HDN void hv_nxn_helper(int q, HvCu cu, int h, int six);
/* six helper wavefronts for the NxN chain of this round?  The workgroup's wavefronts 6 and 7 are free when the stream writer is a workgroup
   of its own (clusters of five and more) -- except wavefront 6 in a CTU's first round, when it checks the CTU before (hv_verify_nxn) */
static HD int hv_nxq_six(const HvLds &S, const HevceImageDesc &im, int k16, int m) {
    return im.cluster != nullptr && im.cluster->size >= 5 && !(hv_uni(k16) == 0 && hv_uni(m) == 0 && hv_uni(S.nxv_pend));
}
static HD void hv_jobs_phase(HvLds &S, const HevceImageDesc &im, const HvCu c8, const HvCu c16, const HvCu c32, int k16, int m, int ctu, int wr_on, int cl) {
    const int tid = hv_tid(), wave = tid >> 6, lane = tid & 63, q = im.qpd6;
    const int wv = hv_uni(wave);
    const int fast = !hv_uni(S.exact);
    k16 = hv_uni(k16); m = hv_uni(m);
    HV_GLOBAL u8 *s0 = hv_level_scratch(im, 0), *s1 = hv_level_scratch(im, 1), *s2 = hv_level_scratch(im, 2);
    if (wv == 0) {
        /* the NxN chain; it also writes the token stream, slot list and engine state of its outcome, candidate 70, and
           clears S.nxn_busy shortly before its end */
        HV_GLOBAL u16 *c70 = hv_c70_area(s0, fast);
        if (fast && hv_uni(cl) == 2) hv_job_nxn<true, true>(wave, lane, q, c8, c70, hv_nxq_six(S, im, k16, m) ? 6 : 4);
        else if (fast) hv_job_nxn<true, false>(wave, lane, q, c8, c70, 0);   /* no helper wavefronts: one lane per mode */
        else hv_job_nxn<false, false>(wave, lane, q, c8, c70, 0);
        return;
    }
    if (wv == (hv_uni(cl) == 2 ? 7 : 4) && !(hv_uni(cl) == 2 && im.cluster->size >= 5)) {   /* (clusters of five and more: the writer is a workgroup of its own) */
        /* wavefront 4 shares its SIMD with wavefront 0, so no jobs: it writes the previous CTU's slice data instead (slice = this
           round's 8x8 position); the older wavefront 0 wins the SIMD's issue arbitration.  Cluster of four: wavefronts are to spare,
           the writer is wavefront 7 (measured: on wavefront 4 its slice, 91 k cycles, outlasted the NxN chain's 79 k and took issue
           slots from it), wavefront 4 idles and the NxN chain has its SIMD to itself; the tile in ws[4] stays the writer's */
        if (wr_on) {
            const unsigned long long t0 = hv_realtime();
            if (k16 == 0 && m == 0) { hv_wr_stage(S, im, ctu - 1); WAVE_SYNC(); hv_wr_tokenise(im, ctu - 1); WAVE_SYNC(); }
#ifndef HEVCE_EMU
            /* cluster of four: this wavefront's slice is what ends the jobs phase when it outlasts the NxN chain, and equal shares of the
               CTU's chunks over fifteen rounds did (87 k cycles against 81 k; the first round, with the binarisation, 133 k; the last
               one idle).  So: equal shares of the TIME over all sixteen rounds -- the binarisation counted at what it took, the chunks
               at the pace of the last slice (100 MHz ticks, x 256) */
            if (hv_uni(cl) == 2) {
                const int r = k16 * 4 + m, tpc = hv_max(hv_uni(S.wst.tpc), 64), total = hv_uni(S.wst.total), left = total - hv_uni(S.wst.used);
                int budget = (left + 15 - r) / (16 - r);
                if (r == 0) { const long long tok = (long long)(hv_realtime() - t0) << 8; budget = (int)(((tok + (long long)total * tpc) / 16 - tok) / tpc); }
                hv_wr_engine(im, ctu - 1, r == 15 ? 100000 : hv_max(budget, 1));
            } else
#endif
            hv_wr_engine(im, ctu - 1, (k16 * 4 + m) == 15 ? 100000 : S.wst.budget);
        }
        return;
    }
    if (hv_uni(cl) == 2) {                                       /* cluster of 4: this CU's block jobs and candidates 0..69 are another workgroup's */
        if (wv == 5) hv_c70_rider(im, hv_c70_area(s0, 1), q);      /* FAST analysis only (cl == 2); it also reports the candidate to the 8x8 workgroup */
        const int six = hv_nxq_six(S, im, k16, m);
        if (wv >= 1 && wv <= 4) hv_nxn_helper(q, c8, wv - 1, six);   /* the NxN chain's block trials and binarisation (hv_nxq_units) */
        if (wv == 4) hv_cl_peek8(S, im, lane);                    /* then: the 8x8 workgroup's answer, noticed while the chain winds up */
        if (wv == 6 && k16 == 0 && m == 0 && hv_uni(S.nxv_pend)) hv_verify_nxn(lane, 1);   /* the previous CTU's NxN winners, off the critical path */
        if (wv >= 6 && six) hv_nxn_helper(q, c8, wv - 2, six);    /* the stream writer is a workgroup of its own: two more helpers */
        return;
    }
    const int w1 = wv < 4 ? wv - 1 : wv - 2;                     /* work slot 0..5 */
    HvGLev l0 = (HvGLev)s0, l1 = (HvGLev)s1, l2 = (HvGLev)s2;
    HvGRec r0 = (HvGRec)(s0 + HV_TL * 1024 * 2), r1 = (HvGRec)(s1 + HV_TL * 1024 * 2), r2 = (HvGRec)(s2 + HV_TL * 1024 * 2);
    /* 8x8: 3 TU-split chains (16 modes each) on slots 0..2, 5 single TUs (8 modes each) on slots 3..5 */
    if (w1 < 3) HV_JOBT(8, hv_job_chain<4>(wave, lane, q, c8, w1, l0, r0));
    else {
        HV_JOBT(9, hv_job_single<8>(wave, lane, q, c8, 2 * (w1 - 3), l0, r0));
        if (w1 < 5) HV_JOBT(9, hv_job_single<8>(wave, lane, q, c8, 2 * (w1 - 3) + 1, l0, r0));
    }
    {
        const int tb = w1 < 3 ? 35 + 16 * w1 : 16 * (w1 - 3), tc = hv_min(16, (w1 < 3 ? 70 : 35) - tb);
        if (fast) hv_bin_own<true>(s0, c8, tb, tc); else hv_bin_own<false>(s0, c8, tb, tc);
    }
    {   /* the 8x8 candidates' engines start from the CU's entry snapshot  [ref :1423-1424] */
        HvLds::Level &L0 = S.lv[0];
        const int t6 = w1 * 64 + lane;
        for (int k = t6; k < HEVCE_NCTXP * HV_TL; k += 384) L0.tctx[k / HV_TL][k % HV_TL] = S.ent_ctx[0][k / HV_TL];
        if (t6 < 71) L0.tregs[t6] = S.ent[0];
    }
    if (hv_uni(cl)) return;                                      /* cluster: the enclosing CUs are the parent workgroup's */
    if (m == 0) {
        /* 16x16: 5 chains (8 modes each) on slots 0..4, 9 single TUs (4 modes each): one each on slots 0..4, four on slot 5 */
        if (w1 < 5) { HV_JOBT(10, hv_job_chain<8>(wave, lane, q, c16, w1, l1, r1)); HV_JOBT(11, hv_job_single<16>(wave, lane, q, c16, w1, l1, r1)); }
        else for (int j = 5; j < 9; j++) HV_JOBT(11, hv_job_single<16>(wave, lane, q, c16, j, l1, r1));
    } else if (k16 < 2 && !(k16 == 1 && m == 3)) {
        /* 32x32: 9 chains (4 modes each, ~150 k cycles) and 18 single TUs (2 modes each, ~75 k): one per work slot in the
           five phases (k16, m) = (0,1) (0,2) (0,3) (1,1) (1,2).  The two phases with m == 1 also carry the 16x16 candidates'
           binarisation (~53 k): they get the short jobs, the chains go to (0,2) and (0,3) */
        const int r5 = k16 * 3 + m - 1;
        int chain = -1, single = -1;
        if (r5 == 0) single = w1;
        else if (r5 == 1) chain = w1;
        else if (r5 == 2) { if (w1 < 3) chain = 6 + w1; else single = 3 + w1; }
        else if (r5 == 3) single = 9 + w1;
        else if (w1 < 3) single = 15 + w1;
        if (chain >= 0) HV_JOBT(12, hv_job_chain<16>(wave, lane, q, c32, chain, l2, r2));
        if (single >= 0) HV_JOBT(13, hv_job_single<32>(wave, lane, q, c32, single, l2, r2));
    }
    /* whatever time the NxN chain still takes: the candidates of the enclosing CUs that are already prepared ride on, twelve
       lanes per CU level and job wavefront, until wavefront 0 clears S.nxn_busy (16x16 candidates exist from their CU's
       second round on, 32x32 candidates after round (1,2)) */
    /* the 16x16 CU's candidates (their jobs ran in the previous round): twelve per job wavefront, any wavefront will do */
    if (m == 1) { if (fast) hv_bin_own<true>(s1, c16, 12 * w1, hv_min(12, 70 - 12 * w1)); else hv_bin_own<false>(s1, c16, 12 * w1, hv_min(12, 70 - 12 * w1)); }
    if (k16 == 1 && m == 3) {                                    /* the 32x32 CU's: its last jobs ran in round (1,2) */
        if (fast) hv_bin_own<true>(s2, c32, 12 * w1, hv_min(12, 70 - 12 * w1)); else hv_bin_own<false>(s2, c32, 12 * w1, hv_min(12, 70 - 12 * w1));
    }
    hv_run_engines((HV_GLOBAL u8 *)im.scratch, q, (m >= 2 ? 2 : 0) | (k16 >= 2 ? 4 : 0), &S.nxn_busy, 6, w1);
}

/* cluster mode (described further down): mailbox numbering and the bounded wait */
#define HV_CL_BOX(lvl) (((lvl) + 2) % 3)                    /* mailbox of a CU level: 16x16 -> 0, 32x32 -> 1, 8x8 -> 2 */
/* Waits between workgroups that are all RESIDENT (the chain workgroup has seen every alive word, hv_cl_alive) end when the partner has
   done its work -- tens of milliseconds at the very most (a CTU analysed again with the EXACT engines) -- so the hand-off timeout is a
   backstop against a wedged partner, not the detector of a cluster that was never placed: that one is the alive wait, 20 ms. */
#define HV_CL_TIMEOUT 300000000ull                           /* 3 s of the 100 MHz clock */
#define HV_CL_ALIVE_TIMEOUT 2000000ull                       /* 20 ms */
static HD int hv_cl_wait(const int *seq, int want, int *fail, int light, unsigned long long timeout = HV_CL_TIMEOUT) {   /* one thread */
    const unsigned long long t0 = hv_realtime();
    for (;;) {
        if (hv_ld_flag(seq) == want) { hv_acquire_fence(light); return 1; }
        if (*fail || hv_realtime() - t0 > timeout) { *fail = 1; return 0; }   /* after one failure the picture is lost anyway: no more waiting */
        hv_spin_pause();
    }
}
static HD void hv_cl_receive(HvLds &S, const HevceImageDesc &im, const HvCu cu);
static HD void hv_decide8_cluster(HvLds &S, const HevceImageDesc &im, const HvCu cu, HevceRecord *rec);
static HD void hv_decide_cu(HvLds &S, const HevceImageDesc &im, const HvCu cu, int ctu, HevceRecord *rec, int32_t *dump, int cl) {
    const int tid = hv_tid(), wave = tid >> 6, lane = tid & 63, q = im.qpd6, n = cu.n;
    HvLds::Level &L = S.lv[cu.lvl];
    const int k8 = hv_uni(S.cl_ans8) + 1;                        /* cluster of four: the number of this 8x8 CU since the launch; counted at the end of its decision */
    HV_GLOBAL u8 *scr = (n == 8 && cl == 2) ? hv_worker8_scratch(im, k8) : hv_level_scratch(im, cu.lvl);
    HvGLev glev = (HvGLev)scr;
    HvGRec grec = (HvGRec)(scr + HV_TL * 1024 * 2);
    const int ntrials = (n == 8) ? 71 : 70;
    const int my = 1 + (cu.y >> 2), mx = 1 + (cu.x >> 2);
    HV_PROF_T0();
    if (n == 8 && cl == 2 && !dump) { hv_decide8_cluster(S, im, cu, rec); HV_PROF(S, 0, 4); return; }   /* the same decision and commit, in two barriers */
    if (n == 8 && cl == 2) { HV_PROFC(S, -1); hv_cl_receive(S, im, cu); HV_PROFC(S, 1); }   /* cluster of 4: candidates 0..69 were evaluated elsewhere, the best of them joins candidate 70 */
    const int lean = n > 8 && cl && !dump;
    if (lean) {
        /* cluster: two alternatives, the parent workgroup's best unsplit candidate and the split the children left behind, which
           any candidate of equal cost beats (index -1 in the arg-min below)  [ref :1409-1410, 1440]: hv_cl_receive + the arg-min,
           every thread for itself, in two barriers instead of five */
        const int b = HV_CL_BOX(cu.lvl);
        if (hv_uni(wave) == 0) hv_job_sse(S, lane, cu);
        if (tid == 0) {
            const unsigned long long to = (im.flags & 4) ? 100000ull : HV_CL_TIMEOUT;   /* test hook, see hv_cl_receive */
            if (!hv_cl_wait(&im.cluster->res[b].seq, S.cl_seq[b], &S.cl_fail, S.cl_light, to)) im.cluster->error = 1;
        }
        BLOCK_SYNC();
        const HevceRes *r = &im.cluster->res[b];
        const int wa = hv_min(hv_max(hv_ldc_i32(&r->winner), 0), 69);
        const int csplit = hv_rdcost(q, S.split_sse, hv_regs_bits(S.com) - hv_regs_bits(S.ent[cu.lvl]));
        const int wl = hv_ldc_i32(&r->cost) <= csplit ? wa : -1;
        if (tid == 0) {
            S.winner = wl;
            L.tregs[wa] = S.ent[cu.lvl]; L.tregs[wa].range = hv_ldc_i32(&r->range); L.tregs[wa].nbits = hv_ldc_i32(&r->nbits); L.tnlog[wa] = 0;
            const int ok = S.cl_ok | (hv_ldc_i32(&r->pad_[0]) == 1 ? 1 << b : 0);
            S.cl_ok = ok;
            if ((ok & (im.cluster->size <= 3 ? 3 : 7)) == (im.cluster->size <= 3 ? 3 : 7) && !(im.flags & 16)) S.cl_light = 1;
        }
        if (wl >= 0 && tid >= 64 && tid < 64 + HEVCE_NCTXP) L.tctx[tid - 64][wa] = (u8)hv_ldc_u8(&r->ctx[tid - 64]);
        BLOCK_SYNC();
    }
    if (n > 8 && !lean) {
        /* the split alternative is the state the children left behind  [ref :1409-1410]; it also bounds the pruning */
        if (hv_uni(wave) == 0) hv_job_sse(S, lane, cu);
        BLOCK_SYNC();
        if (cl) hv_cl_receive(S, im, cu);                        /* cluster: the parent workgroup's best unsplit candidate is the level's only one */
        else {
            if (tid == 0) hv_atomic_min(&L.best_cost, hv_rdcost(q, S.split_sse, hv_regs_bits(S.com) - hv_regs_bits(S.ent[cu.lvl])));
            BLOCK_SYNC();
            hv_engine_phase(S, im, 1 << cu.lvl, 1);             /* whatever the slices left over */
        }
    }
    HV_PROF(S, cu.lvl, 2);
    if (dump && tid < HV_TL) {
        int32_t *d = dump + 16 * tid;
        const bool v = tid < ntrials;
        d[0] = v ? L.tsse[tid] : 0; d[1] = v ? L.tcost[tid] : 0; d[2] = v ? L.tdead[tid] : 0; d[3] = v ? L.tlen[tid] : 0;
    }
    /* ---------------- arg-min in the reference's order, later candidate wins ties ---------------- */
    if (!lean && hv_uni(wave) == 0) {
        /* key = (cost, index): lowest cost, the highest index among equals [ref :1440 adopts on >=]; the split alternative
           takes part as index -1 (any candidate of equal cost beats it); -2 = nothing yet.  Two candidates per lane, then a
           butterfly over the wavefront. */
        int bc = HV_I32MAX, bi = -2;
        if (lane == 0 && n > 8) { bc = hv_rdcost(q, S.split_sse, hv_regs_bits(S.com) - hv_regs_bits(S.ent[cu.lvl])); bi = -1; }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int t = lane + 64 * u;
            const bool ok = t < ntrials && !L.tdead[hv_min(t, HV_TL - 1)];
            const int c = L.tcost[hv_min(t, HV_TL - 1)];
            const bool take = ok && (bi == -2 || c < bc || (c == bc && t > bi));
            bc = take ? c : bc; bi = take ? t : bi;
        }
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int oc = hv_shfl_xor(bc, sft), oi = hv_shfl_xor(bi, sft);
            const bool take = oi != -2 && (bi == -2 || oc < bc || (oc == bc && oi > bi));
            bc = take ? oc : bc; bi = take ? oi : bi;
        }
        if (lane == 0) S.winner = bi < 0 ? -1 : bi;
    }
    if (!lean) BLOCK_SYNC();
    const int w = S.winner;
    HV_PROF(S, cu.lvl, 3);
    if (n == 8 && cl == 2) HV_PROFC(S, 2);
    if (w < 0) return;                                          /* the four sub-CUs stay */

    /* ---------------- commit the winner: coder, contexts, reconstruction, maps, record ---------------- */
    if (tid == 0) { S.com = L.tregs[w]; if (!S.exact) S.com.nbits -= L.tnlog[w]; }   /* FAST: the winner's bypass bins are bits of the stream too */
    if (tid < HEVCE_NCTXP) S.com_ctx[tid] = L.tctx[tid][w];
    const int kindw = w < 35 ? 0 : (w < 70 ? 1 : 2);
    for (int p = tid; p < n * n; p += HV_NTHREADS) {
        const int yy = p / n, xx = p % n;
        /* cluster mode: the winner's samples / levels were written by another workgroup: read at the L2 (hv_ldc_*) */
        S.rec[cu.y + yy + 1][cu.x + xx + HV_RX] = (kindw == 2) ? S.nxn_crec[yy][xx] : (u8)(cl ? hv_ldc_u8((const u8 *)(grec + w * 1024 + p)) : grec[w * 1024 + p]);
        int lv;
        if (kindw == 0) {
            const int at = w * 1024 + hv_scan_idx(S, cu.lg, hv_scan_kind(n, w), yy, xx);
            lv = cl ? hv_ldc_i16((const short *)(glev + at)) : glev[at];
        } else {
            const int h = n >> 1, s = (yy >= h) * 2 + (xx >= h);
            const int md = kindw == 1 ? w - 35 : S.nxn_modes[s];
            const int si = hv_scan_idx(S, cu.lg - 1, hv_scan_kind(h, md), yy & (h - 1), xx & (h - 1));
            const int at = w * 1024 + s * h * h + si;
            lv = kindw == 1 ? (cl ? hv_ldc_i16((const short *)(glev + at)) : glev[at]) : S.nxn_bestlev[s][si];
        }
        rec->lev[cu.y + yy][cu.x + xx] = (short)lv;
    }
    const int n4 = n >> 2;
    if (tid < n4 * n4) {
        const int i = tid / n4, jx = tid % n4;
        const int md = kindw == 0 ? w : kindw == 1 ? w - 35 : S.nxn_modes[(i * 4 >= n / 2) * 2 + (jx * 4 >= n / 2)];
        S.map_cu[my + i][mx + jx] = (u8)n;
        S.map_pm[my + i][mx + jx] = (u8)md;
        rec->mode[(cu.y >> 2) + i][(cu.x >> 2) + jx] = (u8)md;
        rec->info[(cu.y >> 2) + i][(cu.x >> 2) + jx] = (u8)(cu.lvl | (kindw == 2 ? HEVCE_INFO_NXN : 0) | (kindw == 1 ? HEVCE_INFO_TUSPLIT : 0));
    }
    BLOCK_SYNC();
    HV_PROF(S, cu.lvl, 4);
    if (n == 8 && cl == 2) { HV_PROFC(S, 3); if (tid == 0) S.cl_ans8 = k8; }   /* read again only behind the next barrier */
}

/* --------------------------------------------------------------------------------------------- */
/* per-image driver: raster CTU loop  [ref HEVCe.c:1578-1647]                                       */
/* --------------------------------------------------------------------------------------------- */
static HD void hv_init_tables(HvLds &S, int q, int tid) {
    if (tid < 64) { S.lps4[tid] = hv_lps4[tid]; S.lpsnext[tid] = hv_lpsnext[tid]; }
    if (tid >= 384 && tid < 384 + 35) { S.ang[tid - 384] = hv_ang[tid - 384]; S.invang[tid - 384] = hv_invang[tid - 384]; }
    if (tid == 64) {
        /* up-right diagonal order inside a 4x4 group and over 2x2 / 4x4 / 8x8 groups  [ref :1130-1132] */
        int k = 0;
        for (int d = 0; d <= 6; d++)
            for (int y = (d < 4 ? d : 3); y >= 0 && d - y < 4; y--) { S.d4pos[k] = (u8)((y << 2) | (d - y)); S.d4idx[(y << 2) | (d - y)] = (u8)k; k++; }
        for (int t = 0; t < 3; t++) {
            const int ng = 2 << t;
            k = 0;
            for (int d = 0; d <= 2 * (ng - 1); d++)
                for (int y = (d < ng ? d : ng - 1); y >= 0 && d - y < ng; y--) { S.cgpos[t][k] = (u8)((y << 3) | (d - y)); S.cgidx[t][y * ng + (d - y)] = (u8)k; k++; }
        }
    }
    if (tid >= 128 && tid < 128 + HEVCE_NCTXP) {                                                          /* [ref :727-735] */
        const int i = tid - 128;
        const u8 v = (u8)hv_ctx_fresh_value(i, q);
        S.ctx_fresh[i] = v; S.com_ctx[i] = (i == HV_CT_PAD) ? (u8)HV_PADCV : v; S.nxn_ctx[i][HV_TL - 1] = v;
    }
    if (tid == 300) { S.com.range = 510; S.com.low = 0; S.com.nbits = 23; S.com.nbytes = 0; S.com.bufbyte = 0xff; S.com.zeros = 0; S.com.nout = 0; S.com.pad = 0; S.wr = S.com; S.wr_epb = 0; }
}

/* The engines' context table for the mode the workgroup is in (EXACT: ctab, FAST: ctab2 -- they share their LDS).  All threads;
   needs a barrier before the next engine runs  [ref :701-713, 914-933]. */
struct HvSig4Info { u16 v[3][16]; u8 scan[3][16]; };
constexpr HvSig4Info hv_make_sig4info() {
    /* sig_coeff_flag of a 4x4 block, per scan kind and scan position: context class (H.265 9.3.4.2.5, sigCtx of a 4x4 block) | number
       m of earlier-coded positions of the same class << 4 | the first of them << 6 | the second << 10 (a class has at most three
       positions that can carry a flag: position 15 never does)  [ref :1092-1122] */
    HvSig4Info t{};
    constexpr int sig4[16] = HEVCE_TAB_SIG4;
    constexpr int dg[16] = { 0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15 };   /* up-right diagonal: scan position -> raster  [ref :1130-1132] */
    for (int kind = 0; kind < 3; kind++) {
        int cls[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int ii = 0; ii < 16; ii++) {
            const int pos = kind == 0 ? dg[ii] : (kind == 1 ? ii : ((ii & 3) * 4 + (ii >> 2)));
            cls[ii] = pos == 0 ? 0 : sig4[pos];
            t.scan[kind][pos] = (u8)ii;
        }
        for (int ii = 0; ii < 16; ii++) {
            int m = 0, e[2] = { 0, 0 };
            for (int jj = 14; jj > ii; jj--) if (cls[jj] == cls[ii] && m < 2) e[m++] = jj;
            t.v[kind][ii] = (u16)(cls[ii] | (m << 4) | (e[0] << 6) | (e[1] << 10));
        }
    }
    return t;
}
DEVCONST HvSig4Info hv_sig4info = hv_make_sig4info();
static HD void hv_set_mode(HvLds &S, int exact, int q, int tid) {
    if (!exact) {
        /* FAST: the fresh-context resolution tables (they share their LDS with the EXACT engines' byte sinks) and the tables of
           the one-lane-per-sample 4x4 trials (they share theirs with the EXACT trials' reference lines) */
        for (int e = tid; e < HV_FT_SIZE; e += HV_NTHREADS) {
            int cv, idx = 1;
            if (e < HV_FT_SIG) cv = hv_ctx_fresh_value(e, q);
            else if (e < HV_FT_ONE3) { cv = hv_ctx_fresh_value(CX_SIG + ((e - HV_FT_SIG) >> 3), q); idx = (e - HV_FT_SIG) & 7; }
            else if (e < HV_FT_ONE0) { cv = hv_ctx_fresh_value(CX_ONE + 3, q); idx = 1 << (e - HV_FT_ONE3); }
            else { cv = hv_ctx_fresh_value(CX_ONE, q); idx = e - HV_FT_ONE0; }
            const int k = 31 - hv_clz((unsigned)hv_max(idx, 1));   /* the bins below idx's leading 1, lowest first */
            for (int i = 0; i < k; i++) cv = hv_ctx_next(cv, (idx >> i) & 1);
            S.ftab[e] = (e == HV_CT_PAD) ? (u8)HV_PADCV : (u8)cv;
        }
        for (int e = tid; e < 128 * 32; e += HV_NTHREADS) {
            const int idx = e & 31, k = 31 - hv_clz((unsigned)hv_max(idx, 1));
            int cv = e >> 5;
            if (cv < 126) for (int i = 0; i < k; i++) cv = hv_ctx_next(cv, (idx >> i) & 1);
            S.t4c[e >> 5][idx] = (u8)cv;
        }
        if (tid < 48) {
            S.fx.sig4[tid >> 4][tid & 15] = hv_sig4info.v[tid >> 4][tid & 15];
            S.fx.scan4[tid >> 4][tid & 15] = hv_sig4info.scan[tid >> 4][tid & 15];
            S.fx.unscan4[tid >> 4][hv_sig4info.scan[tid >> 4][tid & 15]] = (u8)(tid & 15);
        }
        if (tid >= 64 && tid < 80) S.fx.d4[tid - 64] = hv_tm.f4[tid - 64];
    }
    if (exact) {
        if (tid < 128) {
            const int cv = tid, st = cv >> 1, mps = cv & 1;
            const int nm = st < 62 ? cv + 2 : cv, nl = st == 0 ? (mps ^ 1) : ((hv_lpsnext[st] << 1) | mps);
            S.ctab[cv][0] = hv_lps4[st];
            S.ctab[cv][1] = (unsigned)(nl | (nm << 8));
        }
    } else if (tid < 256) {
        const int cv = tid >> 1, bin = tid & 1, st = cv >> 1, mps = cv & 1, islps = bin != mps;
        const int nm = st < 62 ? cv + 2 : cv, nl = st == 0 ? (mps ^ 1) : ((hv_lpsnext[st] << 1) | mps);
        const int real = st < 63;                                /* context bytes 126 / 127 (state 63) never occur: 126 is the pad context */
        S.ctab2[tid][0] = real ? hv_lps4[st] : 0u;
        S.ctab2[tid][1] = real ? (unsigned)((islps ? nl : nm) | (islps << 8)) : (unsigned)cv;
    }
    if (tid == 0) S.exact = exact;
}

/* second init step (after the scan tables exist): significance-context tables  [ref :1092-1122 evaluated per position] */
static HD void hv_init_sigtab(HvLds &S, int tid) {
    for (int e = tid; e < 35 * 16; e += HV_NTHREADS) {
        const int tb = e >> 4, ii = e & 15;
        int n, kind, nb, first;
        if (tb < 3) { n = 4; kind = tb; nb = 0; first = 1; }
        else if (tb < 27) { n = 8; kind = (tb - 3) >> 3; nb = ((tb - 3) >> 1) & 3; first = !((tb - 3) & 1); }
        else { n = 16; kind = 0; nb = ((tb - 27) >> 1) & 3; first = !((tb - 27) & 1); }
        int iy, ix;
        hv_scan_in(S, kind, ii, iy, ix);
        const int gofs = first ? 0 : 4;                       /* any group other than the first: only "(y>>2)|(x>>2) != 0" matters */
        S.sigtab[tb][ii] = (u8)hv_sigctx(n, kind == 0, gofs + iy, ix, nb);
    }
}
static HD void hv_snapshot(HvLds &S, int lvl, int tid) {
    if (tid == 0) S.ent[lvl] = S.com;
    if (tid < HEVCE_NCTXP) S.ent_ctx[lvl][tid] = S.com_ctx[tid];
}

/* Candidate 70's rider (cluster of four: the chain workgroup's other wavefronts are free).  Candidate 70 -- the CU coded as four
   4x4 blocks with the NxN chain's winning modes -- can only be costed from the CU's entry state when the chain is done: its
   CU header comes first and holds the four prev_intra_luma_pred_flags, which depend on all four modes  [ref :985-1018,
   1531-1541].  But the header's other bins are the same whatever the modes, the flags are four bins on one context, and bypass
   bins do not move the range: so SIXTEEN lanes code the candidate at once, one per flag pattern, block by block as the chain
   delivers the winners' tokens (the tokens are the same in every lane), and when the chain has fixed the modes the lane of the
   pattern that came true IS candidate 70.  Only the last block's tokens are left to code when the chain ends. */
static HDN void hv_c70_rider(const HevceImageDesc &im, HV_GLOBAL u16 *c70, int q) {
    HvLds &S = HV_S;
    HvLds::Level &L = S.lv[0];
    const int lane = hv_tid() & 63, pl = lane & 15, act = lane < 16;
    q = hv_uni(q);
    u8 *cx = &L.tctx[0][pl];                                     /* context columns 0..15 of the level: its candidates 0..69 live elsewhere */
    for (int r = lane >> 4; r < HEVCE_NCTXP; r += 4) L.tctx[r][pl] = S.ent_ctx[0][r];
    for (int r = lane; r < HEVCE_NCTXP; r += 64) ((HV_LDSP int *)((HV_LDSP u8 *)&S.ws[hv_tid() >> 6] + 768))[r] = 0;
    WAVE_SYNC();
    HvFCod c = { S.ent[0].range, S.ent[0].nbits };
    if (act) {                                                                                              /* [ref :952-955, 985-1033] */
        hv_fbin(S, c, HV_CX(CX_PART), 0);
        for (int i = 0; i < 4; i++) hv_fbin(S, c, HV_CX(CX_YPM), (pl >> i) & 1);
        hv_fbin(S, c, HV_CX(CX_UVPM), 0);
        hv_fbin(S, c, HV_CX(CX_UVCBF), 0);
        hv_fbin(S, c, HV_CX(CX_UVCBF), 0);
    }
    for (int s = 0; s < 4; s++) {
        while (hv_poll(&S.c70_done) <= s) hv_spin_short();
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
        const unsigned long long rb_ = __builtin_amdgcn_s_memtime();
#endif
        const int seg = hv_uni(S.c70_seg[s]), n = (seg >> 16) & 0x7fff, cbf = (seg >> 31) & 1;
        const HV_LDSP unsigned *stage = (const HV_LDSP unsigned *)&S.ws[hv_tid() >> 6] + (s & 1) * 64;   /* the block's tokens, padded: put there by the chain (hv_job_nxn) */
        if (act) hv_fbin(S, c, HV_CX(CX_YCBF), cbf);                                                        /* [ref :1038-1043] */
        /* The block's tokens are the same in all sixteen lanes and none is on the flags' context, so the context byte a bin meets is the
           same in every lane: column 0's byte walked through the bin's predecessors of the same context in this block -- which the
           chain's tokeniser has put into the token (HV_RTOK_KP: at most six, one or two look-ups in S.t4c).  One lane per token resolves,
           the sixteen lanes then run an engine without context look-ups (hv_rengine, ~46 cycles per bin; with a look-up per bin
           ~250: the last block's bins are on the round's critical path).  A context's byte after the block = after its last bin here:
           the bin with the most predecessors (an LDS maximum per context). */
        {
            HV_LDSP u8 *tile = (HV_LDSP u8 *)&S.ws[hv_tid() >> 6];
            HV_LDSP unsigned *rt = (HV_LDSP unsigned *)(tile + 512);                     /* 64 resolved tokens */
            HV_LDSP int *cnt = (HV_LDSP int *)(tile + 768);                              /* per context: bins in this block (zero between blocks) */
            HV_LDSP u8 *after = tile + 768 + 4 * HEVCE_NCTXP;                            /* [context][k]: the byte after its bin number k */
            u8 *cx0 = &L.tctx[0][0];
            const unsigned t = lane < n ? stage[lane] : (unsigned)HV_TOK_NOPF;
            const int idx = (int)(((t & 0x1fffu) * 911u) >> 16), bin = (int)((t >> 14) & 1u), valid = lane < n && idx != HV_CT_PAD;
            const int kp = (int)HV_RTOK_GETKP(t), k = kp & 7, pre = kp >> 3, k1 = hv_min(k, 4), k2 = k - k1;
            const int c0 = valid ? cx0[idx * HV_TL] : 0;
            const int s1 = S.t4c[c0][(1 << k1) | (pre & ((1 << k1) - 1))];
            const int st = S.t4c[s1][(1 << k2) | (pre >> 4)];                            /* (k2 = 0: entry 1 is the byte itself) */
            rt[lane] = valid ? HV_RTOK(st, bin, 0) : HV_RTOK_NOP;
            if (valid) { after[idx * 8 + k] = S.t4c[st][2 | bin]; hv_lds_max(&cnt[idx], k + 1); }
            WAVE_SYNC();
            for (int r = lane; r < HEVCE_NCTXP; r += 64) {
                const int m = cnt[r];
                if (m) { cx0[r * HV_TL] = after[r * 8 + m - 1]; cnt[r] = 0; }
            }
            WAVE_SYNC();
            if (act) hv_rengine(S, c, (const HV_LDSP HvChunk *)rt, n);
        }
        WAVE_SYNC();
        if (lane == 0) hv_flag_set(&S.c70_rd, s + 1);
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
        if (lane == 0) S.prof2[s == 3 ? 12 : 31] += __builtin_amdgcn_s_memtime() - rb_;   /* the rider's blocks: the last one / the others */
#endif
    }
    while (hv_poll(&S.c70_done) < 5) hv_spin_short();
    const int pat = hv_uni(S.c70_pat) & 15;
    if (lane == pat) {
        const int byp = L.tnlog[70];
        L.tregs[70].range = c.range; L.tregs[70].nbits = c.nbits;
        L.tcost[70] = hv_rdcost(q, L.tsse[70], S.ent[0].nbits - c.nbits + byp);
        L.tdead[70] = 0; L.tpos[70] = -1;
    }
    for (int r = lane; r < HEVCE_NCTXP; r += 64) L.tctx[r][70] = L.tctx[r][r == CX_YPM ? pat : 0];   /* the flags' context from the pattern's lane, everything else from column 0 */
    WAVE_SYNC();
    hv_cl_report_nxn(S, im, lane);
}
