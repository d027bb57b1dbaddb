/* hevce_dev_cluster.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   cluster mode, chain side: messages to the worker workgroups, their answers, the 8x8 decision. */
/* --------------------------------------------------------------------------------------------- */
/* Cluster mode: TWO to FOUR workgroups per picture.  The chain workgroup (hv_analyse_image) keeps what is serial -- the sixteen   */
/* 8x8 CUs of a CTU, one after the other, their NxN chains, the commits, the stream writer -- and hands every 16x16 / 32x32 CU to   */
/* a parent workgroup (hv_parent_worker, on another compute unit) at the moment the CU is ENTERED: its 70 candidates start from    */
/* the entry snapshot and read only samples outside the CU  [ref :1364-1365, 1420-1424], so they are independent of everything the */
/* chain does until the CU's decision, four (sixteen) rounds later.  Back comes the best unsplit candidate; the chain compares it  */
/* with the split alternative and commits.  Clusters of four: a fourth workgroup evaluates candidates 0..69 of the 8x8 CUs too     */
/* (the chain keeps the NxN candidate, number 70).  It is on the critical path, so it gets ONE message per 16x16 quad and walks    */
/* the quad's four CUs on its own: after its answer for a CU it takes the chain's report of the NxN candidate (hv_cl_report_nxn),   */
/* makes the CU's decision itself -- the chain makes the same one from the same numbers -- and enters the next CU with the winner's */
/* samples, modes, coder and contexts (hv_pw_advance8) while the chain is still committing.  Answers, reports and the 8x8          */
/* candidates' trial scratch are double-buffered by the parity of the CU's number, so neither side can overwrite what the other     */
/* still reads.  Hand-off through HBM (HevceCluster): sequence words, light or agent-scope fences (hv_publish_fence), every wait    */
/* bounded (a partner that is not resident must not hang the GPU: the picture is then flagged and the host encodes it again        */
/* without a cluster).  EXACT CTUs (re-analysis) are done by the chain workgroup alone, as without a cluster.                      */
/* --------------------------------------------------------------------------------------------- */
/* chain workgroup: the CU of level lvl is entered (S.ent / S.ent_ctx / S.lv[lvl] hold its entry state): tell the parent workgroup.
   All threads; what the parents read of the picture's reconstruction has been stored by now (fence below). */
static HD void hv_cl_send(HvLds &S, const HevceImageDesc &im, const HvCu cu, int ctu, int kind, const HvCu *quad = nullptr) {   /* quad: the 16x16 CU an 8x8 message opens */
    const int tid = hv_tid(), b = HV_CL_BOX(cu.lvl);
    HevceMsg *m = &im.cluster->msg[b];
    const HvLds::Level &L = S.lv[cu.lvl];
    if (tid == 0) {
        m->kind = kind; m->ctu = ctu; m->y = cu.y; m->x = cu.x; m->n = cu.n;
        m->range = S.ent[cu.lvl].range; m->nbits = S.ent[cu.lvl].nbits;
        m->split_inc = L.split_inc; m->ml = L.ml; m->ma = L.ma;
        m->L = cu.L; m->LB = cu.LB; m->A = cu.A; m->AR = cu.AR;
        if (quad) {
            const int my = 1 + (cu.y >> 2), mx = 1 + (cu.x >> 2);
            m->ml2 = S.map_pm[my + 2][mx - 1]; m->ma2 = S.map_pm[my - 1][mx + 2];
            m->qL = quad->L; m->qLB = quad->LB; m->qA = quad->A; m->qAR = quad->AR;
        }
    }
    if (tid < HEVCE_NCTXP) m->ctx[tid] = S.ent_ctx[cu.lvl][tid];
    if (tid == 0) m->pad_ = hv_xcc_id();                           /* where this workgroup sits: the receiver may take the light path if it sits there too */
    hv_publish_fence(S.cl_light);
    BLOCK_SYNC();
    if (tid == 0) { const int s = S.cl_seq[b] + 1; S.cl_seq[b] = s; hv_st_flag(&m->seq, s); }
}
/* chain workgroup, cluster of four: the NxN candidate is complete -- its rider (hv_c70_rider) has just coded it.  The rider's own
   wavefront reports it to the 8x8 workgroup, which then knows the CU's decision as soon as it has its own best candidate and goes on
   to the quad's next CU without waiting for this workgroup's commit.  One wavefront, inside the jobs phase: no barrier, and nothing
   of it left between two jobs phases. */
static HD void hv_cl_report_nxn(HvLds &S, const HevceImageDesc &im, int lane) {
    const int k = hv_uni(S.cl_ans8) + 1;                         /* the CU of this round (its decision counts it) */
    HevceNxn *r = &im.cluster->nxn[k & 1];
    const HvLds::Level &L = S.lv[0];
    if (lane == 0) {
        r->cost = L.tdead[70] ? HV_I32MAX : L.tcost[70]; r->range = L.tregs[70].range; r->nbits = L.tregs[70].nbits - L.tnlog[70];
        r->modes = S.nxn_modes[0] | (S.nxn_modes[1] << 8) | (S.nxn_modes[2] << 16) | (S.nxn_modes[3] << 24);
    }
    for (int i = lane; i < HEVCE_NCTXP; i += 64) r->ctx[i] = L.tctx[i][70];
    r->rec[lane] = S.nxn_crec[lane >> 3][lane & 7];
}
/* ... and its sequence word, by the same wavefront BEHIND the barrier that ends the jobs phase: the wait for the stores above to have
   left the compute unit (a round trip to the L2) is then beside the workgroup's decision instead of in front of the barrier; the 8x8
   workgroups, who read the report, have the time */
static HD void hv_cl_report_nxn_flag(HvLds &S, const HevceImageDesc &im, int lane) {
    const int k = hv_uni(S.cl_ans8) + 1;
    hv_publish_fence(S.cl_light);
    WAVE_SYNC();
    if (lane == 0) hv_st_flag(&im.cluster->nxn[k & 1].seq, k);
}
/* chain workgroup, at the CU's decision: wait for the answer and make its candidate the only one alive of the level.  All threads. */
static HD void hv_cl_receive(HvLds &S, const HevceImageDesc &im, const HvCu cu) {
    const int tid = hv_tid(), b = HV_CL_BOX(cu.lvl);
    HvLds::Level &L = S.lv[cu.lvl];
    const int size = im.cluster->size;
    /* the answer to 8x8 CU number k = cl_ans8 + 1 is in res[2 + (k & 1)] */
    const int want = cu.lvl == 0 ? hv_uni(S.cl_ans8) + 1 : hv_uni(S.cl_seq[b]), slot = cu.lvl == 0 ? 2 + (want & 1) : b;
    if (tid == 0) {
        const unsigned long long to = (im.flags & 4) ? 100000ull : HV_CL_TIMEOUT;   /* test hook (flag bit 2): the workers have left, give up after 1 ms */
        if (!hv_cl_wait(&im.cluster->res[slot].seq, want, &S.cl_fail, S.cl_light, to)) im.cluster->error = 1;
        if (cu.lvl == 0) HV_PROFC(S, 0);                         /* waited for the 8x8 answer */
    }
    BLOCK_SYNC();
    const HevceRes *r = &im.cluster->res[slot];
    const int w = hv_min(hv_max(hv_ldc_i32(&r->winner), 0), 69);
    if (tid < HV_TL) L.tdead[tid] = tid != w && !(cu.lvl == 0 && tid == 70);   /* 8x8: the NxN candidate is the chain workgroup's own */
    if (tid == 0) {
        L.tcost[w] = hv_ldc_i32(&r->cost); L.tsse[w] = hv_ldc_i32(&r->sse); L.tregs[w] = S.ent[cu.lvl];
        L.tregs[w].range = hv_ldc_i32(&r->range); L.tregs[w].nbits = hv_ldc_i32(&r->nbits); L.tnlog[w] = 0; L.tpos[w] = -1;
        /* the senders say whether they share this workgroup's XCD; when all of the cluster's do, hand-offs go the light way */
        int ok = S.cl_ok | (hv_ldc_i32(&r->pad_[0]) == 1 ? 1 << b : 0);
        S.cl_ok = ok;
        const int need = size <= 3 ? 3 : 7;
        if ((ok & need) == need && !(im.flags & 16)) S.cl_light = 1;
    }
    if (tid < HEVCE_NCTXP) L.tctx[tid][w] = (u8)hv_ldc_u8(&r->ctx[tid]);
    BLOCK_SYNC();
}
/* chain workgroup, cluster of four, decision of an 8x8 CU: hv_cl_receive + the arg-min + the commit of hv_decide_cu for the two
   candidates there are -- the 8x8 workgroup's best (its answer) and the NxN candidate, number 70, which wins on equal cost (highest
   index, [ref :1440, 1476]) -- with every thread working the decision out for itself, so that the wait for the answer is followed by
   one barrier instead of five.  All threads. */
static HD void hv_decide8_cluster(HvLds &S, const HevceImageDesc &im, const HvCu cu, HevceRecord *rec) {
    const int tid = hv_tid();
    HvLds::Level &L = S.lv[0];
    const int k = hv_uni(S.cl_ans8) + 1, n8 = hv_n8(im);         /* this CU's number; counted below */
    const int pf = hv_uni(S.pf.k) == k;                          /* seen during the jobs phase (hv_cl_peek8; barriers since) */
    const int pfd = pf && hv_uni(S.pf.dk) == k;                   /* ... and its data fetched */
    const HvLds::PeekD &D = hv_pfd(S);
    HV_PROFC(S, -1);
    if (!pf) {
        if (tid == 0) {
            const unsigned long long to = (im.flags & 4) ? 100000ull : HV_CL_TIMEOUT;   /* test hook, see hv_cl_receive */
            for (int g = 0; g < n8; g++) if (!hv_cl_wait(&im.cluster->res[2 + 2 * g + (k & 1)].seq, k, &S.cl_fail, S.cl_light, to)) im.cluster->error = 1;
        }
        BLOCK_SYNC();
    }
    HV_PROFC(S, 0);
    int wa, ca;
    if (pf) { wa = hv_min(hv_uni(S.pf.w), 69); ca = hv_uni(S.pf.cost); }
    else {
        const HevceRes *r0 = &im.cluster->res[2 + (k & 1)];
        wa = hv_min(hv_max(hv_ldc_i32(&r0->winner), 0), 69); ca = hv_ldc_i32(&r0->cost);
        for (int g = 1; g < n8; g++) {
            const HevceRes *r1 = &im.cluster->res[2 + 2 * g + (k & 1)];
            const int wb = hv_min(hv_max(hv_ldc_i32(&r1->winner), 0), 69), cb = hv_ldc_i32(&r1->cost);
            if (hv_w8_better(ca, wa, cb, wb)) { wa = wb; ca = cb; }
        }
    }
    const int slot = 2 + 2 * hv_w8_owner(wa, n8) + (k & 1);
    const HevceRes *r = &im.cluster->res[slot];                  /* the answer of the workgroup that candidate came from */
    const int win70 = L.tcost[70] <= ca;
    const int w = win70 ? 70 : wa;
    HV_GLOBAL u8 *scr = hv_worker8_scratch(im, k);
    HvGLev glev = (HvGLev)scr;
    HvGRec grec = (HvGRec)(scr + HV_TL * 1024 * 2);
    if (tid == 0) {
        S.winner = w;
        if (win70) { S.com = L.tregs[70]; S.com.nbits -= L.tnlog[70]; }
        else { S.com = S.ent[0]; S.com.range = pfd ? D.range : hv_ldc_i32(&r->range); S.com.nbits = pfd ? D.nbits : hv_ldc_i32(&r->nbits); }
        if (!S.cl_light && !(im.flags & 16)) {                   /* all of the cluster on this XCD: light hand-offs (hv_cl_receive; normally known since the alive check) */
            int ok = S.cl_ok;
            for (int g = 0; g < n8; g++) ok |= hv_ldc_i32(&im.cluster->res[2 + 2 * g + (k & 1)].pad_[0]) == 1 ? 4 << g : 0;
            S.cl_ok = ok;
            const int need8 = 3 | (((1 << n8) - 1) << 2);
            if ((ok & need8) == need8) S.cl_light = 1;
        }
    }
    if (tid >= 64 && tid < 64 + HEVCE_NCTXP) S.com_ctx[tid - 64] = win70 ? L.tctx[tid - 64][70] : (pfd ? D.ctx[tid - 64] : (u8)hv_ldc_u8(&r->ctx[tid - 64]));
    const int kindw = w < 35 ? 0 : (w < 70 ? 1 : 2);
    if (tid >= 256 && tid < 320) {
        const int p = tid - 256, yy = p >> 3, xx = p & 7;
        S.rec[cu.y + yy + 1][cu.x + xx + HV_RX] = (kindw == 2) ? S.nxn_crec[yy][xx] : (pfd ? D.rec[p] : (u8)hv_ldc_u8((const u8 *)(grec + w * 1024 + p)));
        int lv;
        if (kindw == 0) { const int li = hv_scan_idx(S, 3, hv_scan_kind(8, w), yy, xx); lv = pfd ? D.lev[li] : hv_ldc_i16((const short *)(glev + w * 1024 + li)); }
        else {
            const int s4 = (yy >= 4) * 2 + (xx >= 4);
            const int md = kindw == 1 ? w - 35 : S.nxn_modes[s4];
            const int si = hv_scan_idx(S, 2, hv_scan_kind(4, md), yy & 3, xx & 3);
            lv = kindw == 1 ? (pfd ? D.lev[s4 * 16 + si] : hv_ldc_i16((const short *)(glev + w * 1024 + s4 * 16 + si))) : S.nxn_bestlev[s4][si];
        }
        rec->lev[cu.y + yy][cu.x + xx] = (short)lv;
    }
    if (tid >= 384 && tid < 388) {
        const int i = (tid - 384) >> 1, jx = (tid - 384) & 1, my = 1 + (cu.y >> 2), mx = 1 + (cu.x >> 2);
        const int md = kindw == 0 ? w : kindw == 1 ? w - 35 : S.nxn_modes[i * 2 + jx];
        S.map_cu[my + i][mx + jx] = 8;
        S.map_pm[my + i][mx + jx] = (u8)md;
        rec->mode[(cu.y >> 2) + i][(cu.x >> 2) + jx] = (u8)md;
        rec->info[(cu.y >> 2) + i][(cu.x >> 2) + jx] = (u8)(0 | (kindw == 2 ? HEVCE_INFO_NXN : 0) | (kindw == 1 ? HEVCE_INFO_TUSPLIT : 0));
    }
    BLOCK_SYNC();
    HV_PROFC(S, 3);
    if (tid == 0) S.cl_ans8 = k;                                 /* read again only behind the next barrier (the next round's jobs phase) */
}
/* the samples of a CU that has just been decided go to the picture's reconstruction at once (the parents of the NEXT CUs read them there) */
static HD void hv_cl_store_rcon(HvLds &S, const HevceImageDesc &im, const HvCu cu, int y0, int x0) {
    for (int p = hv_tid(); p < cu.n * cu.n; p += HV_NTHREADS) {
        const int yy = cu.y + p / cu.n, xx = cu.x + p % cu.n;
        im.rcon[(size_t)(y0 + yy) * im.W + x0 + xx] = S.rec[yy + 1][xx + HV_RX];
    }
}
