/* hevce_dev_picture.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   the picture: CTU snapshots, the alive check of a cluster, the raster loop over the CTUs with FAST analysis and EXACT re-analysis (hv_analyse_image). */
/* CTU `ctu` starts from here (the committed coder and contexts, what it reads of its left neighbour and of the CTU row above) */
static HD void hv_snap_take(HvLds &S, int ctu, int x0, int tid) {
    HvLds::Snap &P = S.snap[ctu & 1];
    if (tid == 0) P.com = S.com;
    if (tid < HEVCE_NCTXP) P.com_ctx[tid] = S.com_ctx[tid];
    if (tid >= 128 && tid < 136) { P.map_cu8[tid - 128] = S.map_cu[1 + tid - 128][8]; P.map_pm8[tid - 128] = S.map_pm[1 + tid - 128][8]; P.line_cu[tid - 128] = S.line_cu[(x0 >> 2) + tid - 128]; }
    if (tid >= 192 && tid < 224) P.rec_col[tid - 192] = S.rec[1 + tid - 192][HV_RX + 31];
}
static HD void hv_snap_restore(HvLds &S, int ctu, int x0, int tid) {
    const HvLds::Snap &P = S.snap[ctu & 1];
    if (tid == 0) S.com = P.com;
    if (tid < HEVCE_NCTXP) S.com_ctx[tid] = P.com_ctx[tid];
    if (tid >= 128 && tid < 136) { S.map_cu[1 + tid - 128][8] = P.map_cu8[tid - 128]; S.map_pm[1 + tid - 128][8] = P.map_pm8[tid - 128]; S.line_cu[(x0 >> 2) + tid - 128] = P.line_cu[tid - 128]; }
    if (tid >= 192 && tid < 224) S.rec[1 + tid - 192][HV_RX + 31] = P.rec_col[tid - 192];
}

/* chain workgroup, before the first CTU: every other workgroup of the cluster has started (HevceCluster.alive).  One thread; returns 0
   and flags the cluster when one has not shown up in time */
static HD int hv_cl_alive(const HevceImageDesc &im, int *same_xcd) {
    HevceCluster *C = im.cluster;
    const unsigned long long t0 = hv_realtime(), to = (im.flags & 4) ? 100000ull : HV_CL_ALIVE_TIMEOUT;   /* test hook: 1 ms */
    const int me = hv_xcc_id();
    hv_st_flag(&C->alive[0], 0x100 | me);                        /* (the writer workgroup reads it: light hand-offs with this one) */
    for (;;) {
        int all = 1, same = 1;
        for (int r = 1; r < C->size; r++) { const int a = hv_ld_flag(&C->alive[r]); all &= a != 0; same &= (a & 0xff) == me; }
        if (all) { *same_xcd = same; return 1; }
        if (hv_realtime() - t0 > to) { C->error = 6; return 0; }
        hv_spin_pause();
    }
}
/* The picture: CTUs in raster order, analysed with the FAST engines while the writer (one CTU behind) produces the bytes.
   A CTU whose decisions the FAST bit counts cannot vouch for -- the writer put an emulation-prevention byte while encoding it,
   or one of its NxN winners would have had one -- is POISONED: the workgroup goes back to that CTU's start, analyses it with
   the EXACT engines, lets the writer encode it in one go, and carries on FAST.  A few CTUs per 4K picture.
   im.flags (tests): bit 0 = EXACT throughout; bits 8-15 = poison every k-th CTU as an NxN winner would, bits 16-23 = as the
   writer would (each CTU once). */
static HD void hv_analyse_image(HvLds &S, const HevceImageDesc &im) {
    const int tid = hv_tid(), q = im.qpd6, H = im.H, W = im.W;
    if (tid == 0) { S.trace = im.trace; S.qlut = im.qlut; }
#ifdef HEVCE_ANYPROF
    if (tid < 24) (&S.prof[0][0])[tid] = 0;
    if (tid < 32) S.prof2[tid] = 0;
#endif
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
    const unsigned long long prof_start_ = __builtin_amdgcn_s_memtime();
    if (tid == 0) S.prof_pt = prof_start_;
#endif
    BLOCK_SYNC();
    HV_TRACE(S, 0, 1);
    hv_init_tables(S, q, tid);
    const int force_exact = (im.flags & 1) || !im.payload;      /* without the writer nothing vouches for the FAST decisions */
    const int pk_nxn = (im.flags >> 8) & 0xff, pk_wr = (im.flags >> 16) & 0xff;
    hv_set_mode(S, force_exact, q, tid);
    if (tid == 0) { S.poison = 0; S.nxv_pend = 0; S.poison_prev = 0; S.wr_done = 0; S.cl_seq[0] = S.cl_seq[1] = S.cl_seq[2] = 0; S.cl_ans8 = 0; S.pf.k = 0; S.pf.dk = 0; S.cl_fail = 0; S.cl_light = 0; S.cl_ok = 0; }
    BLOCK_SYNC();
    hv_init_sigtab(S, tid);
    const int lut4_lds = im.cluster != nullptr && im.cluster->size >= 4;   /* cluster of four: the NxN chain's helpers look the 4x4 levels up in LDS */
    if (lut4_lds && !force_exact) hv_fill_lut4(S, tid);
    BLOCK_SYNC();
    const int ncx = W >> 5, nctu = ncx * (H >> 5);
    int ctu = 0, redone = -1, n_redo = 0, n_bad = 0;            /* redone: the CTU most recently analysed twice (test hooks fire once) */
    /* cluster of five: the stream writer is a workgroup of its own (hv_writer_worker), several CTUs behind */
    const int wrx = im.cluster != nullptr && im.cluster->size >= 5 && !force_exact;
    HevceWrx *X = wrx ? &im.cluster->wrx : nullptr;
    int epoch = 0;
    static_assert(sizeof(HvLds::Snap) == 192, "HevceWrx::snap holds an HvLds::Snap");
    if (im.cluster) {                                            /* all of the cluster is resident, or the picture is given back to the host at once */
        if (tid == 0) {
            int same = 0;
            if (!hv_cl_alive(im, &same)) S.cl_fail = 1;
            else if (same && !(im.flags & 16)) S.cl_light = 1;   /* the whole cluster on this XCD (the launch's layout, checked): light hand-offs from the start */
        }
        BLOCK_SYNC();
    }
    while (!hv_uni(S.cl_fail)) {
        const int exact = hv_uni(S.exact);
        const int tail = ctu == nctu;                            /* past the last CTU: only the writer has work left */
        const int cl = im.cluster != nullptr && !exact;          /* the parent workgroup evaluates this CTU's 16x16 / 32x32 CUs */
        const int cl8 = cl && im.cluster->size >= 4;             /* ... and a fourth workgroup the 8x8 CUs' candidates 0..69 */
        const int wr_on = !wrx && im.payload && ctu > 0 && hv_uni(S.wr_done) < ctu;   /* the writer does CTU ctu - 1 during this round */
        const int y0 = (ctu / ncx) * 32, x0 = (ctu % ncx) * 32;
        if (wr_on && tid == 4 * 64) { S.wsnap = S.wr; S.wr_epb = 0; }
        if (wr_on && (tid >> 6) == 4) for (int i = tid & 63; i < HEVCE_NCTXP; i += 64) S.wsnap_ctx[i] = S.nxn_ctx[i][HV_TL - 1];
        if (tail) { if (wr_on) hv_wr_whole(S, im, ctu - 1); }
        else {
            HevceRecord *rec = (HevceRecord *)(im.records + (size_t)ctu * HEVCE_REC_BYTES);
            HV_TRACE(S, 0, 1000 + ctu);
            const int L = x0 > 0, A = y0 > 0, AR = A && (x0 + 32 < W);
            hv_snap_take(S, ctu, x0, tid);
            /* ---- load the CTU: source samples (edge-replicated), neighbours, maps  [ref :1611-1622] ---- */
            for (int p = tid; p < 1024; p += HV_NTHREADS) {
                const int yy = hv_min(y0 + (p >> 5), im.src_h - 1), xx = hv_min(x0 + (p & 31), im.src_w - 1);
                S.org[p >> 5][p & 31] = im.img[(size_t)yy * (size_t)im.src_stride + xx];
            }
            if (tid < 32) S.rec[1 + tid][HV_RX - 1] = L ? S.rec[1 + tid][HV_RX + 31] : 128;   /* left column = previous CTU's last column */
            if (wrx && tid == 0) {                               /* never more than six CTUs ahead of the writer: the rings hold eight */
                const unsigned long long t0 = hv_realtime();
                for (;;) {
                    const unsigned long long wd = hv_ld_flag64((const unsigned long long *)&X->wdone);
                    if ((int)(wd >> 32) == epoch && (int)(unsigned)wd >= ctu - 6) break;
                    if (hv_realtime() - t0 > HV_CL_TIMEOUT) { S.cl_fail = 1; im.cluster->error = 3; break; }
                    hv_spin_pause();
                }
            }
            BLOCK_SYNC();
            if (wrx) {                                           /* what this CTU starts from, for a later return to it */
                if (tid < 48) ((unsigned *)X->snap[ctu & 7])[tid] = ((const HV_LDSP unsigned *)&S.snap[ctu & 1])[tid];
                if (tid == 64) { X->crange[ctu & 7] = S.com.range; X->xflag[ctu & 7] = exact; }
            }
            if (tid < 65 && A) {                                 /* row above (x0-1 .. x0+63) from the reconstruction in HBM */
                const int xx = hv_min(hv_max(x0 + tid - 1, 0), W - 1);
                S.rec[0][HV_RX - 1 + tid] = im.rcon[(size_t)(y0 - 1) * W + xx];
            }
            if (tid >= 128 && tid < 136) {
                const int i = tid - 128;
                S.map_cu[1 + i][0] = L ? S.map_cu[1 + i][8] : 32;
                S.map_pm[1 + i][0] = L ? S.map_pm[1 + i][8] : 1;
            }
            BLOCK_SYNC();
            if (tid >= 192 && tid < 201) {
                const int jx = tid - 192;                        /* row above: CU sizes of the CTU row above, modes always DC */
                S.map_cu[0][jx] = (A && (jx > 0 || L)) ? S.line_cu[(x0 >> 2) + jx - 1] : 32;
                S.map_pm[0][jx] = 1;
            }
            for (int p = tid; p < (int)(sizeof(HevceRecord) / 4); p += HV_NTHREADS) ((int *)rec)[p] = 0;
            BLOCK_SYNC();

            /* ---- quad-tree, depth first in z-order  [ref :1403-1407]: 21 CUs = 4 x (4 x 8x8, 16x16), 32x32.
               A CU's candidates are prepared when the CU is ENTERED (before its split_cu_flag / children), decided after
               its children; parents' engines advance inside every 8x8 engine phase. ---- */
            const HvCu c32 = { 0, 0, 32, 5, 2, L, 0, A, AR };
            for (int step = 0; step < 21; step++) {
                const int k = step / 5, m = step % 5;
                HvCu c16 = { (k >> 1) * 16, (k & 1) * 16, 16, 4, 1, 0, 0, 0, 0 };
                hv_sub_avail(k & 3, c32.L, c32.LB, c32.A, c32.AR, c16.L, c16.LB, c16.A, c16.AR);
                HvCu cu = c16;
                if (step == 20) cu = c32;
                else if (m < 4) {
                    cu.y = c16.y + (m >> 1) * 8; cu.x = c16.x + (m & 1) * 8; cu.n = 8; cu.lg = 3; cu.lvl = 0;
                    hv_sub_avail(m, c16.L, c16.LB, c16.A, c16.AR, cu.L, cu.LB, cu.A, cu.AR);
                }
                if (step == 0) {                                 /* entering the 32x32 CU */
                    hv_snapshot(S, 2, tid);
                    hv_level_begin(S, c32);
                    BLOCK_SYNC();
                    if (cl) hv_cl_send(S, im, c32, ctu, 1);
                    if (tid == 0) hv_commit_bin(S, CX_SPLIT_CU + S.lv[2].split_inc, 1);
                    BLOCK_SYNC();
                }
                const int quad_first = cl8 && step < 20 && m == 0;   /* clusters of four and more: the quad's message goes out BEFORE the 16x16 CU's (below) */
                if (step < 20 && m == 0) {                       /* entering a 16x16 CU */
                    hv_snapshot(S, 1, tid);
                    hv_level_begin(S, c16);
                    BLOCK_SYNC();
                    if (cl && !quad_first) hv_cl_send(S, im, c16, ctu, 1);
                    if (tid == 0) hv_commit_bin(S, CX_SPLIT_CU + S.lv[1].split_inc, 1);
                    BLOCK_SYNC();
                }
                if (cu.n == 8) {
                    hv_snapshot(S, 0, tid);
                    hv_level_begin(S, cu);
                    if (!cl8) BLOCK_SYNC();
                    else {                                       /* (no barrier needed first: every thread below reads what it wrote itself above) */
                        /* the candidates of this CU other than the NxN one are another workgroup's: only candidate 70 runs here */
                        if (tid < 70) { S.lv[0].tdead[tid] = 1; S.lv[0].tpos[tid] = -1; }
                        if (tid == 0) S.l0_active = 1;
                        HV_PROFC(S, 5);                          /* since the last 8x8 decision: enclosing CUs' decisions and entries, snapshot, level begin */
                        if (m == 0) {
                            hv_cl_send(S, im, cu, ctu, 1, &c16);   /* opens the quad; the 8x8 workgroups find their way through the other three CUs */
                            /* ... and only now the 16x16 CU's own message (its entry snapshot, S.ent[1], is from before the split flag): the 8x8
                               workgroups' first CU of a quad is what the chain waits for, the 16x16 answer is due four rounds later */
                            if (quad_first) hv_cl_send(S, im, c16, ctu, 1);
                        }
                        else BLOCK_SYNC();
                        HV_PROFC(S, 6);
                    }
                    HV_PROF_T0();
#ifdef HEVCE_PROF_JDUR
                    const unsigned long long j0_ = __builtin_amdgcn_s_memtime();
#endif
                    hv_jobs_phase(S, im, cu, c16, c32, k, m, ctu, wr_on, cl + cl8);
#ifdef HEVCE_PROF_JDUR
                    if (tid == HEVCE_PROF_JDUR * 64) S.prof2[16 + k * 4 + m] += __builtin_amdgcn_s_memtime() - j0_;   /* that wavefront's jobs phase, riders included */
#endif
#ifdef HEVCE_PROF_WAIT
                    const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
                    BLOCK_SYNC();                                /* the NxN job borrows the 8x8 level's context columns */
                    if (cl8 && (tid >> 6) == 5) hv_cl_report_nxn_flag(S, im, tid & 63);   /* the rider's wavefront: its report's sequence word */
#ifdef HEVCE_PROF_WAIT
                    if (tid == 0) S.prof2[16 + k * 4 + m] += __builtin_amdgcn_s_memtime() - w0_;
#endif
                    HV_PROF(S, 0, 0);
                    if (!cl) {
                        if (m == 0) hv_level_contexts(S, im, c16);
                        if (k == 1 && m == 2) hv_level_contexts(S, im, c32);     /* its jobs ran in five phases up to this one */
                    }
                    if (!cl8) hv_engine_phase(S, im, cl ? 1 : (1 | (m >= 1 ? 2 : 0) | ((k >= 2 || (k == 1 && m == 3)) ? 4 : 0)), 0);   /* cluster of 4: candidate 70 has been coded by its rider */
                    else { HV_PROFC(S, -1); HV_PROFC(S, 1); }     /* ... who has also reported it to the 8x8 workgroup */
                }
                hv_decide_cu(S, im, cu, ctu, rec, im.dump ? im.dump + ((size_t)ctu * 21 + step) * HV_TL * 16 : nullptr, cl + cl8);
                if (cl && (cu.n == 16 || (cl8 && cu.n == 8))) hv_cl_store_rcon(S, im, cu, y0, x0);   /* the next CU's candidates elsewhere predict from these samples */
                if (cl8 && cu.n == 8) HV_PROFC(S, 4);
                if (cl8 && cu.n > 8) HV_PROFC(S, 2);             /* decisions of the enclosing CUs */
            }

        }
        BLOCK_SYNC();
        /* ---- is everything decided so far the reference's decision?  (FAST only; see hv_fengine_chunk) ----
           The NxN winners of this CTU: verified here (wavefront 4, the others wait), or -- cluster of four, where wavefronts are to
           spare -- copied and verified during the next CTU's first round (hv_jobs_phase); then a failure rolls back one CTU more,
           like the writer's. */
        const int pend = hv_uni(S.nxv_pend);                     /* the previous CTU's winners were deferred to this round */
        const int defer = cl8 && !tail;
        if (!tail && !exact) {
            if (defer) { HV_LDSP HvNxv *V = hv_nxv_copy(S); for (int i = tid; i < (int)(sizeof(HvNxv) / 4); i += HV_NTHREADS) ((HV_LDSP unsigned *)V)[i] = ((const HV_LDSP unsigned *)&S.nxv_lev[0][0])[i]; }
            else if ((tid >> 6) == 4) hv_verify_nxn(tid & 63, 0);
        }
        if (tail && pend && (tid >> 6) == 4) hv_verify_nxn(tid & 63, 1);    /* the last CTU's: nothing left to overlap them with */
        BLOCK_SYNC();
        const int chk_range = tail ? S.com.range : S.snap[ctu & 1].com.range;     /* the analysis' range where the writer now stands */
        const int bad_wr = wr_on && S.wr.range != chk_range;                       /* must never happen: counted, and healed like a poisoned CTU */
        const int p_prev = pend && (S.poison_prev || (pk_nxn && (ctu - 1) % pk_nxn == pk_nxn - 1 && redone != ctu - 1));
        const int p_wr = p_prev || (wr_on && (S.wr_epb > 0 || bad_wr || (pk_wr && (ctu - 1) % pk_wr == pk_wr - 1 && redone != ctu - 1)));
        const int p_nxn = !tail && !exact && !defer && (S.poison || (pk_nxn && ctu % pk_nxn == pk_nxn - 1 && redone != ctu));
        BLOCK_SYNC();
        if (wrx) {
            /* ---- cluster of five: what does the writer workgroup say?  At the end of the picture: wait for it ---- */
            if (tid == 0) {
                int roll = p_prev ? ctu - 1 : -1, mism = 0;
                if (pk_wr && !tail && ctu > 0 && (ctu - 1) % pk_wr == pk_wr - 1 && redone < ctu - 1 - (int)((im.flags >> 24) & 7))
                    roll = hv_max(ctu - 1 - (int)((im.flags >> 24) & 7), 0);   /* test hook: as if the writer had objected to an earlier CTU (flag bits 24-26: how many further back) */
                const unsigned long long t0 = hv_realtime();
                for (;;) {
                    const unsigned long long wb = hv_ld_flag64((const unsigned long long *)&X->wbad), wd = hv_ld_flag64((const unsigned long long *)&X->wdone);
                    if ((int)(wb >> 32) == epoch && (unsigned)wb != 0) {
                        const int r = (int)((unsigned)wb & 0x3fffffffu) - 1;
                        mism = (int)(((unsigned)wb >> 30) & 1);
                        roll = (roll < 0 || r < roll) ? r : roll;
                    }
                    if (!tail || roll >= 0 || ((int)(wd >> 32) == epoch && (int)(unsigned)wd >= nctu)) break;
                    if (hv_realtime() - t0 > HV_CL_TIMEOUT) { S.cl_fail = 1; im.cluster->error = 3; break; }
                    hv_spin_pause();
                }
                if (roll >= 0) {                                 /* the writer's registers at that CTU's start: it has to have got there */
                    for (;;) {
                        const unsigned long long wd = hv_ld_flag64((const unsigned long long *)&X->wdone);
                        if (((int)(wd >> 32) == epoch && (int)(unsigned)wd >= roll) || hv_peek(&S.cl_fail)) break;
                        if (hv_realtime() - t0 > HV_CL_TIMEOUT) { S.cl_fail = 1; im.cluster->error = 3; break; }
                        hv_spin_pause();
                    }
                    hv_acquire_fence(0);
                }
                S.cl_pick = roll; S.winner = mism;
                S.nxv_pend = (defer && !exact && roll < 0) ? 1 : 0; S.poison_prev = 0;
            }
            BLOCK_SYNC();
            const int roll = hv_uni(S.cl_pick);
            if (hv_uni(S.cl_fail)) break;
            if (roll >= 0) {
                /* back to CTU `roll` (up to seven back): the line buffer entries the CTUs since have overwritten, that CTU's start state,
                   the writer's coder registers there; a new epoch sends the writer back too.  EXACT engines for that one CTU. */
                n_redo++; n_bad += hv_uni(S.winner);
                for (int c = hv_min(ctu, nctu - 1); c >= roll; c--)
                    if (tid < 8) S.line_cu[(((c % ncx) * 32) >> 2) + tid] = (u8)hv_ldc_u8(&((const HvLds::Snap *)X->snap[c & 7])->line_cu[tid]);
                if (tid < 48) ((HV_LDSP unsigned *)&S.snap[roll & 1])[tid] = hv_ldg_u32((const unsigned *)X->snap[roll & 7] + tid);
                BLOCK_SYNC();
                hv_snap_restore(S, roll, (roll % ncx) * 32, tid);
                BLOCK_SYNC();
                epoch++;
                if (tid == 0) {
                    const HevceCoderRegs *wr_ = &X->wregs[roll & 7];
                    S.com.range = hv_ldc_i32(&wr_->range); S.com.low = (unsigned)hv_ldc_i32((const int32_t *)&wr_->low); S.com.nbits = hv_ldc_i32(&wr_->nbits);
                    S.com.nbytes = hv_ldc_i32(&wr_->nbytes); S.com.bufbyte = hv_ldc_i32(&wr_->bufbyte); S.com.zeros = hv_ldc_i32(&wr_->zeros); S.com.nout = hv_ldc_i32(&wr_->nout); S.com.pad = 0;
                    S.poison = 0;
                    X->rst = roll;
                    hv_publish_fence(0);
                    hv_st_flag64((unsigned long long *)&X->go, ((unsigned long long)(unsigned)epoch << 32) | (unsigned)roll);
                }
                hv_set_mode(S, 1, q, tid);
                redone = roll; ctu = roll;
                BLOCK_SYNC();
                continue;
            }
        } else {
        if (tid == 0) { S.nxv_pend = (defer && !exact && !p_wr && !p_nxn) ? 1 : 0; S.poison_prev = 0; }
        if (p_wr || p_nxn) {
            /* back to the start of the poisoned CTU (the previous one if the writer or the deferred verification objected), EXACT this
               time; the EXACT coder continues from the writer's registers at that point */
            n_redo++; n_bad += bad_wr;
            if (p_wr) {
                ctu--;
                if (tid == 0) S.wr = S.wsnap;
                if (tid < HEVCE_NCTXP) S.nxn_ctx[tid][HV_TL - 1] = S.wsnap_ctx[tid];
            } else if (wr_on && tid == 0) S.wr_done = ctu;
            BLOCK_SYNC();
            hv_snap_restore(S, ctu, (ctu % ncx) * 32, tid);
            BLOCK_SYNC();
            if (tid == 0) { S.com = S.wr; S.poison = 0; }
            hv_set_mode(S, 1, q, tid);
            redone = ctu;
            BLOCK_SYNC();
            continue;
        }
        }
        if (wr_on && tid == 0) S.wr_done = ctu;
        if (im.cluster && tid == 0 && hv_ldc_i32(&im.cluster->error)) S.cl_fail = 1;   /* a partner workgroup has flagged the cluster (hand-off, HV_CHECK_FULL_WAVE) */
        if (tail) break;
        if (im.cluster) { BLOCK_SYNC(); if (hv_uni(S.cl_fail)) break; }   /* the picture is lost to this launch: no point in walking the rest of it */
        /* ---- end of CTU: terminate bin, publish reconstruction and line buffers  [ref :1626-1637] ---- */
        if (tid == 0) {
            hv_commit_terminate(S, ctu == nctu - 1);
            if (im.ctu_state && !im.payload) {                   /* without the stream writer: the analysis' own coder registers */
                int32_t *st = im.ctu_state + (size_t)ctu * 8;
                const HevceCoderRegs &c = S.com;
                st[0] = c.range; st[1] = (int32_t)c.low; st[2] = c.nbits; st[3] = c.nbytes; st[4] = c.bufbyte; st[5] = c.zeros;
                st[6] = c.nout; st[7] = hv_regs_bits(c);
            }
        }
        for (int p = tid; p < 1024; p += HV_NTHREADS)
            im.rcon[(size_t)(y0 + (p >> 5)) * W + x0 + (p & 31)] = S.rec[1 + (p >> 5)][HV_RX + (p & 31)];
        BLOCK_SYNC();
        if (tid >= 64 && tid < 72) S.line_cu[(x0 >> 2) + tid - 64] = S.map_cu[8][1 + tid - 64];
        BLOCK_SYNC();
        if (exact && im.payload) {
            /* an EXACT CTU is written at once: the writer's registers are what the next FAST or EXACT CTU is checked against
               (cluster of five: the writer workgroup does it, and checks the range at the next CTU's start) */
            if (!wrx) {
                hv_wr_whole(S, im, ctu);
                BLOCK_SYNC();
                if (tid == 0) {
                    S.wr_done = ctu + 1;
                    const HevceCoderRegs &a = S.com, &b = S.wr;
                    if (ctu != nctu - 1) n_bad += a.range != b.range || a.low != b.low || a.nbits != b.nbits || a.nbytes != b.nbytes || a.bufbyte != b.bufbyte || a.zeros != b.zeros || a.nout != b.nout;
                }
            }
            if (!force_exact) { hv_set_mode(S, 0, q, tid); if (lut4_lds) hv_fill_lut4(S, tid); }
            BLOCK_SYNC();
        }
        ctu++;
        if (wrx) {                                               /* the CTU's record and samples are final: the writer may have it */
            hv_publish_fence(S.cl_light);                       /* (every thread's stores: record, samples, snapshot) */
            BLOCK_SYNC();
            if (tid == 0) hv_st_flag64((unsigned long long *)&X->go, ((unsigned long long)(unsigned)epoch << 32) | (unsigned)ctu);
        }
    }
    if (im.cluster) {                                            /* the parent workgroup may go home */
        const HvCu none16 = { 0, 0, 16, 4, 1, 0, 0, 0, 0 }, none32 = { 0, 0, 32, 5, 2, 0, 0, 0, 0 };
        BLOCK_SYNC();
        hv_cl_send(S, im, none16, nctu, 2);                      /* on both mailboxes: one workgroup may serve both, or one each */
        hv_cl_send(S, im, none32, nctu, 2);
        if (im.cluster->size >= 4) { const HvCu none8 = { 0, 0, 8, 3, 0, 0, 0, 0, 0 }; hv_cl_send(S, im, none8, nctu, 2); }
        if (im.cluster->size >= 5 && tid == 0) hv_st_flag64((unsigned long long *)&im.cluster->wrx.go, ((unsigned long long)(unsigned)epoch << 32) | 0x7fffffffu);
    }
    if (im.stats && tid == 0) { im.stats[0] = n_redo; im.stats[1] = n_bad + S.cl_fail; }
    if (S.cl_fail && im.payload_len && tid == 0) *im.payload_len = -1;   /* a hand-off timed out: the host encodes the picture again without a cluster */
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
    if (im.trace && tid < 24) { ((unsigned long long *)im.trace)[tid] = (&S.prof[0][0])[tid]; }
    if (im.trace && tid < 32) { ((unsigned long long *)im.trace)[24 + tid] = S.prof2[tid]; }
    if (im.trace && tid == 0) ((unsigned long long *)im.trace)[56] = __builtin_amdgcn_s_memtime() - prof_start_;   /* whole picture */
#endif
}
