/*
 * hevce_kernel.hip -- gfx950 entry points of the analysis path.  One 512-thread workgroup per image of the batch.
 */
#include "hevce_device.h"

/* The roles of a cluster are functions of their own (not inlined into the kernel): each gets its own register allocation, so that a change
   in the workers' code does not move spill code into the chain's loops (experiments/README.md, round 3). */
static __device__ __noinline__ void hv_role_chain(const HevceImageDesc &im) { hv_analyse_image(HV_S, im); }
static __device__ __noinline__ void hv_role_writer(const HevceImageDesc &im) { hv_writer_worker(HV_S, im); }
static __device__ __noinline__ void hv_role_worker(const HevceImageDesc &im, int serve, int g8) { hv_parent_worker(HV_S, im, serve, g8); }

extern "C" __global__ void __launch_bounds__(HV_NTHREADS, 2)
hevce_analyse_kernel(const HevceImageDesc *__restrict__ images, int nimages, int cluster) {
    /* cluster mode: `cluster` (2 .. 4) workgroups per picture -- the first walks the picture's chain of decisions, the others
       evaluate candidates for it: the 16x16 / 32x32 CUs (one workgroup for both levels, or one each), with four also the 8x8 CUs' */
    /* the workgroups of a picture get block ids that are equal modulo 8: consecutive ids go round the 8 XCDs, so a cluster sits
       on one XCD and shares its L2 (hand-offs without L2 maintenance; checked at run time, hevce_dev_base.h hv_publish_fence) */
    const int b = blockIdx.x, npad = (nimages + 7) & ~7;
    const int pic = cluster > 1 ? b % npad : b, role = cluster > 1 ? b / npad : 0;
    if (pic >= nimages) return;
    const HevceImageDesc im = images[pic];
    if (role == 0) hv_role_chain(im);
    else if (role == 4) hv_role_writer(im);                      /* clusters of five: the stream writer */
    else if (role == 3) hv_role_worker(im, 4, 0);
    else if (role == 5) hv_role_worker(im, 4, 1);               /* clusters of six: the second 8x8 workgroup */
    else if (role == 6) hv_role_worker(im, 4, 2);               /* clusters of seven: the third */
    else hv_role_worker(im, cluster == 2 ? 3 : role, 0);
}

/* host-side launcher used by hevce_host.cpp: grid = images of the batch (times the cluster size) */
extern "C" int hevce_launch_analyse(const HevceImageDesc *d_images, int n, int cluster, hipStream_t stream) {
    hipLaunchKernelGGL(hevce_analyse_kernel, dim3((unsigned)(cluster > 1 ? cluster * ((n + 7) & ~7) : n)), dim3(HV_NTHREADS), 0, stream, d_images, n, cluster);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* ---- function-level test hook: the TU pipeline (prediction -> T -> Q -> Q^-1 -> T^-1 -> reconstruction) for all 35
   modes of ONE N x N block at the CTU origin, from an explicit 33x65 neighbourhood.  One wavefront.  One kernel per block size:
   all four instantiations of hv_tu_job inlined into one kernel needed 342 registers, and the allocator's VGPR -> AGPR spill copies
   ended up in front of an EXEC restore (tools/isa_lint.py, experiments/README.md round 3). ------------------- */
static __device__ void hv_test_init(HvLds &S, const unsigned char *org, const unsigned char *nbr, int q, const unsigned short *qlut, int tid) {
    hv_init_tables(S, q, tid);
    for (int p = tid; p < 1024; p += 64) S.org[p >> 5][p & 31] = org[p];
    for (int p = tid; p < 33 * 65; p += 64) S.rec[p / 65][HV_RX - 1 + p % 65] = nbr[p];
    if (tid == 0) { S.trace = nullptr; S.qlut = qlut; }
    __syncthreads();
    /* the single-wave block has no thread 64/128/300/384: build the angle tables, scan tables and contexts here */
    if (tid < 35) { S.ang[tid] = hv_ang[tid]; S.invang[tid] = hv_invang[tid]; }
    if (tid == 0) {
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
    __syncthreads();
}
template <int N> __global__ void __launch_bounds__(64, 1)
hevce_test_tu_kernel(const unsigned char *org /*32x32*/, const unsigned char *nbr /*33x65: row 0 = y-1, col 0 = x-1*/,
                     int q, int aL, int aLB, int aA, int aAR,
                     short *lev /*35 x 1024, scan order*/, unsigned char *rec /*35 x 1024*/, int *info /*35 x 4: sse,last,cbf,cgsig_lo*/,
                     const unsigned short *qlut) {
    HvLds &S = HV_S;
    const int tid = (int)threadIdx.x;
    hv_test_init(S, org, nbr, q, qlut, tid);
    constexpr int per = 64 / N;
    for (int grp = 0; grp * per < HEVCE_NMODES; grp++) {
        const int mode = grp * per + tid / N;
        const bool ok = mode < HEVCE_NMODES;
        HvGLev ld = (HvGLev)(lev + (ok ? mode : 0) * 1024);
        HvGRec rd = (HvGRec)(rec + (ok ? mode : 0) * 1024);
        const HvTuOut o = hv_tu_job<N, HvGLev, HvGRec>(0, tid, q, 0, 0, 0, 0, N, nullptr, 0, aL, aLB, aA, aAR, ok ? mode : -1, ld, rd, N);
        if (ok && tid % N == 0) { info[mode * 4] = o.sse; info[mode * 4 + 1] = o.last; info[mode * 4 + 2] = o.cbf; info[mode * 4 + 3] = (int)(o.cgsig & 0xffffffffu); }
    }
}

extern "C" int hevce_launch_test_tu(const unsigned char *org, const unsigned char *nbr, int n, int q, int aL, int aLB, int aA, int aAR,
                                    short *lev, unsigned char *rec, int *info, const unsigned short *qlut, hipStream_t stream) {
    if (n == 4)       hipLaunchKernelGGL(hevce_test_tu_kernel<4>, dim3(1), dim3(64), 0, stream, org, nbr, q, aL, aLB, aA, aAR, lev, rec, info, qlut);
    else if (n == 8)  hipLaunchKernelGGL(hevce_test_tu_kernel<8>, dim3(1), dim3(64), 0, stream, org, nbr, q, aL, aLB, aA, aAR, lev, rec, info, qlut);
    else if (n == 16) hipLaunchKernelGGL(hevce_test_tu_kernel<16>, dim3(1), dim3(64), 0, stream, org, nbr, q, aL, aLB, aA, aAR, lev, rec, info, qlut);
    else if (n == 32) hipLaunchKernelGGL(hevce_test_tu_kernel<32>, dim3(1), dim3(64), 0, stream, org, nbr, q, aL, aLB, aA, aAR, lev, rec, info, qlut);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* ---- function-level test hook: the TU-split chains (four TUs of n/2, each predicted from the previous ones'
   reconstruction) of all 35 modes for ONE n x n CU at the CTU origin.  One wavefront runs the chain jobs in turn. --- */
extern "C" __global__ void __launch_bounds__(64, 1)
hevce_test_chain_kernel(const unsigned char *org, const unsigned char *nbr, int n, int q, int aL, int aLB, int aA, int aAR,
                        unsigned char *scratch /*HEVCE_SCRATCH_BYTES*/, int *info /*35 x 16*/, const unsigned short *qlut) {
    HvLds &S = HV_S;
    const int tid = (int)threadIdx.x;
    hv_test_init(S, org, nbr, q, qlut, tid);
    HvGLev glev = (HvGLev)scratch;
    HvGRec grec = (HvGRec)(scratch + HV_TL * 1024 * 2);
    HvCu cu = { 0, 0, n, n == 8 ? 3 : n == 16 ? 4 : 5, n == 8 ? 0 : n == 16 ? 1 : 2, aL, aLB, aA, aAR };
    const int per = 64 / (n / 2);
    for (int grp = 0; grp * per < HEVCE_NMODES; grp++) {
        if (n == 8) hv_job_chain<4>(0, tid, q, cu, grp, glev, grec);
        else if (n == 16) hv_job_chain<8>(0, tid, q, cu, grp, glev, grec);
        else hv_job_chain<16>(0, tid, q, cu, grp, glev, grec);
    }
    __syncthreads();
    for (int m = tid; m < HEVCE_NMODES; m += 64) {
        int *d = info + m * 16;
        const HvLds::Level &L = S.lv[cu.lvl];
        d[0] = L.tsse[35 + m];
        for (int s = 0; s < 4; s++) { d[1 + s] = L.tinfo[35 + m][s].last; d[5 + s] = L.tinfo[35 + m][s].cbf; d[9 + s] = (int)(L.tinfo[35 + m][s].cgsig & 0xffffffffu); }
    }
}
extern "C" int hevce_launch_test_chain(const unsigned char *org, const unsigned char *nbr, int n, int q, int aL, int aLB, int aA, int aAR,
                                       unsigned char *scratch, int *info, const unsigned short *qlut, hipStream_t stream) {
    hipLaunchKernelGGL(hevce_test_chain_kernel, dim3(1), dim3(64), 0, stream, org, nbr, n, q, aL, aLB, aA, aAR, scratch, info, qlut);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
