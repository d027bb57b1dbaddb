/* hevce_dev_base.h -- part of the device code of hevce_device.h (included from there, in this order; not a stand-alone header):
   platform layer (GPU / CPU fiber emulation), compile-time tables, the LDS image (HvLds), small helpers, the arithmetic coder in counting mode. */

#ifdef HEVCE_EMU
#include "emu_rt.h"
#define HV_GLOBAL
#define HV_LDSP
#define HV_KEEP(x) do { } while (0)
#define HV_NOUNROLL
static inline unsigned hv_ldg_u32(const unsigned *p) { return *p; }
void emu_yield(void);
static inline void hv_publish_fence(int) { }
static inline void hv_acquire_fence(int) { }
static inline void hv_st_flag(int *p, int v) { *p = v; }
static inline int  hv_ld_flag(const int *p) { return *p; }
static inline void hv_st_flag64(unsigned long long *p, unsigned long long v) { *p = v; }
static inline unsigned long long hv_ld_flag64(const unsigned long long *p) { return *p; }
static inline int  hv_xcc_id() { return 0; }
static inline void hv_spin_pause() { emu_yield(); }           /* lets the other workgroup's fibers run */
static inline void hv_spin_backoff(int &) { emu_yield(); }
static inline void hv_spin_long() { emu_yield(); }
static inline void hv_spin_short() { emu_yield(); }
static inline unsigned long long hv_realtime() { static unsigned long long t; return t += 1; }
static inline unsigned hv_perm(unsigned a, unsigned b, unsigned sel) {
    const unsigned long long src = ((unsigned long long)a << 32) | b;
    unsigned r = 0;
    for (int i = 0; i < 4; i++) { const unsigned k = (sel >> (8 * i)) & 0xff; r |= (k < 8 ? (unsigned)((src >> (8 * k)) & 0xff) : (k >= 0x0d ? 0xffu : 0u)) << (8 * i); }
    return r;
}
static inline void hv_atomic_add(int *p, int v) { *p += v; }
static inline void hv_atomic_or(int *p, int v) { *p |= v; }
static inline void hv_lds_max(int *p, int v) { if (v > *p) *p = v; }
static inline int hv_shfl(int v, int src) {
    const int w = emu_cur >> 6, l = emu_cur & 63;
    emu_stage[w][l] = v;
    emu_wave_sync();
    const int r = emu_stage[w][src & 63];
    emu_wave_sync();
    return r;
}
static inline int hv_readlane(int v, int src) { return hv_shfl(v, src); }
template <int S_> static inline int hv_row_get(int v) { const int l = emu_cur & 63; return hv_shfl(v, (l & ~15) | ((l + S_) & 15)); }
template <int K_> static inline int hv_quad_get(int v) { const int l = emu_cur & 63; return hv_shfl(v, (l & ~3) | K_); }
struct HvLaneTab { int a[64]; };                            /* emulation: the one fiber that runs wave-uniform code owns all 64 entries */
#define HV_LT_LOAD(tab, L, expr) do { for (int L = 0; L < 64; L++) (tab).a[L] = (expr); } while (0)
#define HV_LT_EACH(tab, L, VAL, stmt) do { for (int L = 0; L < 64; L++) { const int VAL = (tab).a[L]; stmt; } } while (0)
static inline int  hv_lt_get(const HvLaneTab &t, int i) { return t.a[i]; }
static inline void hv_lt_put(HvLaneTab &t, int i, int val) { t.a[i] = val; }
template <typename T> static inline T *hv_uni_ptr(T *p) { return p; }
#else
#include <hip/hip_runtime.h>
#define HD  __device__ __forceinline__
#define HDN __device__ __noinline__
#define DEVCONST __constant__ const
/* WAVE_SYNC: lanes of ONE wavefront exchange data through LDS.  Every such access is a ds_* instruction (all hot
   pointers are typed by address space, never generic/flat), and a wave's ds_* operations execute in order, so the
   fence only has to stop compiler reordering.  HBM stores of a TU job are consumed after the next workgroup barrier. */
#ifndef HEVCE_WSYNC_STRONG
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#endif
#ifdef HEVCE_BSYNC_AGENT
#define BLOCK_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_s_barrier(); \
                          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); } while (0)
#else
#define BLOCK_SYNC() __syncthreads()
#endif
#define HV_NOUNROLL _Pragma("clang loop unroll(disable)")   /* one-lane code: compact beats unrolled (instruction cache) */
#define HV_KEEP(x) asm volatile("" : "+v"(x))          /* pins the first use of a prefetched value to this point */
#define HV_GLOBAL __attribute__((address_space(1)))    /* pointer known to address HBM: global_* instead of flat_* */
#define HV_LDSP   __attribute__((address_space(3)))    /* pointer known to address LDS: ds_* instead of flat_* */
static HD int  hv_tid() { return (int)threadIdx.x; }
static HD int  hv_shfl_xor(int v, int m) { return __shfl_xor(v, m, 64); }
static HD int  hv_bcast0(int v) { return __builtin_amdgcn_readfirstlane(v); }   /* lane 0's value, wave-uniform (SGPR) */
static HD int  hv_atomic_inc(int *p) { return atomicAdd(p, 1); }
static HD void hv_atomic_min(int *p, int v) { atomicMin(p, v); }
static HD int  hv_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
static HD int  hv_poll(const int *p) { return hv_peek(p); }      /* value other wavefronts are changing meanwhile */
static HD void hv_atomic_dec(int *p) { atomicSub(p, 1); }
static HD void hv_flag_clear(int *p) { __hip_atomic_store(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
static HD void hv_flag_set(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define HV_FENCE_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#define HV_FENCE_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup")
static HD int  hv_mul24(int a, int b) { return __mul24(a, b); }
static HD unsigned hv_umul24(unsigned a, unsigned b) { return __umul24(a, b); }   /* both < 2^24: low 32 bits of the product, full rate */
static HD int  hv_clz(unsigned v) { return __clz((int)v); }          /* 32 for v == 0 */
static HD int  hv_clz_nz(unsigned v) { return __builtin_clz(v); }   /* v != 0: the bare instruction */
static HD unsigned long long hv_ballot(int p) { return __ballot(p); }   /* one bit per active lane */
static HD int  hv_clzll(unsigned long long v) { return __clzll((long long)v); }
static HD unsigned hv_perm(unsigned a, unsigned b, unsigned sel) { return __builtin_amdgcn_perm(a, b, sel); }   /* v_perm_b32: result byte i = byte sel.byte[i] of {a:7..4, b:3..0} */
static HD void hv_atomic_add(int *p, int v) { atomicAdd(p, v); }
static HD void hv_atomic_or(int *p, int v) { atomicOr(p, v); }
static HD void hv_lds_max(__attribute__((address_space(3))) int *p, int v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
/* data-parallel primitives (DPP: the operand comes from another lane of the same row of 16 / quad of 4, no LDS involved):
   hv_row_get<s>(v): v of lane (l & ~15) | ((l + s) & 15)   [row_ror:16-s]      hv_quad_get<k>(v): v of lane (l & ~3) | k   [quad_perm] */
template <int S_> static HD int hv_row_get(int v) { return (S_ & 15) ? __builtin_amdgcn_update_dpp(0, v, 0x120 + ((16 - S_) & 15), 0xf, 0xf, false) : v; }
template <int K_> static HD int hv_quad_get(int v) { return __builtin_amdgcn_update_dpp(0, v, K_ * 0x55, 0xf, 0xf, false); }
static HD int hv_shfl(int v, int src) { return __shfl(v, src, 64); }                 /* any lane (ds_bpermute) */
static HD int hv_readlane(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }   /* a wave-uniform lane: no LDS trip */
/* a 64-entry table with entry i in lane i of one VGPR: read / written with a wave-uniform index by v_readlane / v_writelane,
   i.e. without any memory access, and the result is wave-uniform (scalar arithmetic from there on) */
struct HvLaneTab { int v; };
#define HV_LT_LOAD(tab, L, expr) do { const int L = hv_tid() & 63; (tab).v = (expr); } while (0)
#define HV_LT_EACH(tab, L, VAL, stmt) do { const int L = hv_tid() & 63; const int VAL = (tab).v; stmt; } while (0)
static HD int  hv_lt_get(const HvLaneTab &t, int i) { return __builtin_amdgcn_readlane(t.v, i); }
static HD void hv_lt_put(HvLaneTab &t, int i, int val) { t.v = ((int)(threadIdx.x & 63) == i) ? val : t.v; }   /* a compare and a select (v_writelane takes one SGPR only) */
template <typename T> static HD T *hv_uni_ptr(T *p) {
    const unsigned long long a = (unsigned long long)p;
    return (T *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a));
}
/* cluster mode: hand-off between workgroups through HBM.  Payload first (plain stores by any thread), hv_publish_fence() by every
   thread that stored, a workgroup barrier, then ONE thread stores the sequence word; the consumer polls it with loads that
   bypass its L1 (relaxed, agent scope: no cache maintenance per poll), and when it has the value it wants, hv_acquire_fence().
   Two flavours of the fences.  HEAVY: agent-scope release / acquire -- on this multi-XCD part an L2 write-back and an L2
   invalidate, what the memory model asks for between arbitrary workgroups.  LIGHT, when both workgroups are known to sit on the
   SAME XCD (they then share the L2, and the L1s are write-through): the producer only waits until its stores have left the
   compute unit, the consumer reads the payload with L1-bypassing loads (hv_ldg_*); no L2 maintenance at all.  Which one applies
   is decided at run time from the XCC_ID hardware register of the workgroups involved (HvLds::cl_light).
   What the LIGHT flavour ASSUMES of gfx950, beyond the HSA memory model: (1) workgroups that report the same XCC_ID are served by the
   same L2, which is the point of coherence for them; (2) a vector store has reached that L2 when `s_waitcnt vmcnt(0)` lets the wave go on
   (the L1 is write-through, stores are acknowledged by the L2); (3) loads marked system-coherent / L1-bypassing (hv_ldg_*, hv_ldc_*: sc1
   sc0 loads) are served by the L2 and never by a stale L1 line.  The HEAVY flavour assumes none of this; im.flags bit 4 (HEVCE_FLAGS=16)
   forces it for every hand-off, so that the two can be told apart on the device (tests/test_gpu_parity.py runs both). */
static HD void hv_publish_fence(int light) {
    if (light) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
static HD void hv_acquire_fence(int light) { if (!light) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
static HD void hv_st_flag(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static HD int  hv_ld_flag(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static HD void hv_st_flag64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static HD unsigned long long hv_ld_flag64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static HD int  hv_xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }
static HD void hv_spin_pause() { __builtin_amdgcn_s_sleep(2); }   /* between two polls of a word in HBM: 128 cycles (a hand-off is on the critical path) */
/* ... of a wait that may be long (a worker workgroup waiting for its next message): after the first polls the pauses grow to 512 cycles.
   Seven workgroups of a cluster polling one L2 channel 128 cycles apart slow the chain workgroup's own traffic there (measured: its
   sections between two jobs phases 11.5 k -> 16.5 k cycles with the seventh workgroup) */
static HD void hv_spin_long() { __builtin_amdgcn_s_sleep(8); }
static HD void hv_spin_backoff(int &n) { if (n < 24) { n++; __builtin_amdgcn_s_sleep(2); } else __builtin_amdgcn_s_sleep(8); }
static HD void hv_spin_short() { __builtin_amdgcn_s_sleep(1); }   /* between two polls of an LDS flag */
static HD unsigned long long hv_realtime() { return __builtin_amdgcn_s_memrealtime(); }   /* 100 MHz */
static HD unsigned hv_ldg_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   /* read at the L2: never a stale L1 line */
#endif

/* diagnostics: -DHEVCE_TRACE makes lane 0 of every wavefront publish its last checkpoint to host-pinned memory */
#if defined(HEVCE_TRACE) && !defined(HEVCE_EMU)
#define HV_TRACE(S, slotno, code) do { if ((S).trace && (hv_tid() & 63) == 0) { \
        volatile int *t_ = (volatile int *)(S).trace + (hv_tid() >> 6) * 16; t_[slotno] = (code); t_[15] += 1; __threadfence_system(); } } while (0)
#define HV_TRACE_LANE(S, idx, code) do { if ((S).trace) { volatile int *t_ = (volatile int *)(S).trace; t_[128 + (idx)] = (code); __threadfence_system(); } } while (0)
#else
#define HV_TRACE(S, slotno, code) do { } while (0)
#define HV_TRACE_LANE(S, idx, code) do { } while (0)
#endif

/* diagnostics: -DHEVCE_PROF accumulates shader-clock cycles per phase and CU size (thread 0, after the barriers) */
#if defined(HEVCE_PROF) || defined(HEVCE_PROF_TU) || defined(HEVCE_PROF_NXN) || defined(HEVCE_PROF_ENG) || defined(HEVCE_PROF_BIN) || defined(HEVCE_PROF_JDUR) || defined(HEVCE_PROF_WAIT)
#define HEVCE_ANYPROF 1
#endif
#if defined(HEVCE_PROF) && !defined(HEVCE_EMU)
#define HV_PROF_T0()  unsigned long long pt_ = __builtin_amdgcn_s_memtime()
#define HV_PROF2_T0() unsigned long long p2_ = __builtin_amdgcn_s_memtime()
/* chain workgroup, cluster of four: the section between two 8x8 jobs phases, piece i into prof2[5 + i] (i < 0: only restart the clock) */
#define HV_PROFC(S, i) do { if (hv_tid() == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if ((i) >= 0) (S).prof2[5 + (i)] += n_ - (S).prof_pt; (S).prof_pt = n_; } } while (0)
#ifdef HEVCE_PROF_LIGHT                                      /* only the stamps between two jobs phases and the 8x8 workgroup's waits: close to the product's timing */
#define HV_PROF2(S, ph) do { (void)p2_; } while (0)
#define HV_PROF(S, lvl, ph) do { (void)pt_; } while (0)
#else
#define HV_PROF2(S, ph) do { if ((hv_tid() & 63) == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); (S).prof2[ph] += n_ - p2_; p2_ = n_; } } while (0)
#define HV_PROF(S, lvl, ph) do { if (hv_tid() == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); (S).prof[lvl][ph] += n_ - pt_; pt_ = n_; } } while (0)
#endif
#else
#define HV_PROF_T0() do { } while (0)
#define HV_PROF2_T0() do { } while (0)
#define HV_PROF2(S, ph) do { } while (0)
#define HV_PROFC(S, i) do { } while (0)
#define HV_PROF(S, lvl, ph) do { } while (0)
#endif

#define HV_NTHREADS 512
#define HV_NWAVES   8
#define HV_TL       HEVCE_TRIALS
#define HV_I32MAX   0x7fffffff
#define HV_RX       4            /* column offset of x=0 inside Lds::rec rows (x=-1 lives at column 3) */

typedef unsigned char  u8;
typedef unsigned short u16;
typedef unsigned long long u64;
/* loads of what ANOTHER workgroup has written (cluster mode): through hv_ldg_u32, i.e. at the L2, never a stale L1 line */
static HD int hv_ldc_i32(const int32_t *p) { return (int)hv_ldg_u32((const unsigned *)p); }
static HD int hv_ldc_u8(const u8 *p) { const unsigned long long a = (unsigned long long)p; return (int)((hv_ldg_u32((const unsigned *)(a & ~3ull)) >> ((a & 3) << 3)) & 0xff); }
static HD int hv_ldc_i16(const short *p) { const unsigned long long a = (unsigned long long)p; return (int)(short)(hv_ldg_u32((const unsigned *)(a & ~3ull)) >> ((a & 2) << 3)); }

/* --------------------------------------------------------------------------------------------- */
/* compile-time tables                                                                             */
/* --------------------------------------------------------------------------------------------- */
struct HvTMats {
    int f4[16], f8[64], f16[256], f32[1024];   /* C[k][n]  (k = frequency)             */
    int t4[16], t8[64], t16[256], t32[1024];   /* C^T : t[n][k] = C[k][n]              */
};
constexpr int hv_basis(int k, int n) {          /* HEVC core transform entry, 32-point   [ref HEVCe.c:431-464] */
    constexpr int mag[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                              61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
    if (k == 0) return 64;
    int m = (k * (2 * n + 1)) & 127;
    if (m > 64) m = 128 - m;
    return m > 32 ? -mag[64 - m] : mag[m];
}
constexpr HvTMats hv_make_tmats() {
    HvTMats t{};
    constexpr int dst[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };   /* [ref HEVCe.c:391-396] */
    for (int k = 0; k < 4; k++) for (int n = 0; n < 4; n++) { t.f4[k * 4 + n] = dst[k * 4 + n]; t.t4[n * 4 + k] = dst[k * 4 + n]; }
    for (int k = 0; k < 8; k++) for (int n = 0; n < 8; n++) { t.f8[k * 8 + n] = hv_basis(4 * k, n); t.t8[n * 8 + k] = hv_basis(4 * k, n); }
    for (int k = 0; k < 16; k++) for (int n = 0; n < 16; n++) { t.f16[k * 16 + n] = hv_basis(2 * k, n); t.t16[n * 16 + k] = hv_basis(2 * k, n); }
    for (int k = 0; k < 32; k++) for (int n = 0; n < 32; n++) { t.f32[k * 32 + n] = hv_basis(k, n); t.t32[n * 32 + k] = hv_basis(k, n); }
    return t;
}
DEVCONST HvTMats hv_tm = hv_make_tmats();

/* H.265 table 9-46 (rangeTabLps), four bytes per state packed little-endian; table 9-47 (transIdxLps) */
DEVCONST unsigned hv_lps4[64] = HEVCE_TAB_LPS4;
DEVCONST u8 hv_lpsnext[64] = HEVCE_TAB_LPSNEXT;
/* H.265 9.3.2.2 initValue (I slices) in the compact layout of hevce_common.h  [ref HEVCe.c:763-777] */
DEVCONST u8 hv_ctxinit[HEVCE_NCTXP] = HEVCE_TAB_CTXINIT;
DEVCONST signed char hv_ang[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                    -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
DEVCONST short hv_invang[35] = { 0, 0, 256, 315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482,
                                 390, 315, 256, 315, 390, 482, 630, 910, 1638, 4096, 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
DEVCONST u8 hv_lastgrp[32] = HEVCE_TAB_LASTGRP;
DEVCONST u8 hv_lastmin[10] = HEVCE_TAB_LASTMIN;
DEVCONST u8 hv_sig4[16]    = HEVCE_TAB_SIG4;
DEVCONST u8 hv_sigpos[8]   = HEVCE_TAB_SIGPOS;

/* --------------------------------------------------------------------------------------------- */
/* LDS image                                                                                       */
/* --------------------------------------------------------------------------------------------- */
struct HvTuInfo { u64 cgsig; int last; int cbf; };          /* per (trial, TU): 8x8 bitmap of non-zero 4x4 groups */
/* HvLds::ftab -- what a context byte of the FRESH set becomes after a short string of bins (the NxN sub-search codes every 4x4
   block trial from fresh contexts, and no context of a 4x4 block sees more than a few bins): [0, NCTXP) the fresh bytes themselves;
   SIG: sig_coeff_flag class c after the bins idx = 1 << k | bits (k <= 2 earlier flags of the class, the first one bit 0);
   ONE3: greater1 context 3 after k zeros; ONE0: greater1 context 0 after idx = 1 << k | bits (k <= 6, the first one bit 0) */
#define HV_FT_SIG   HEVCE_NCTXP
#define HV_FT_ONE3  (HV_FT_SIG + 9 * 8)
#define HV_FT_ONE0  (HV_FT_ONE3 + 8)
#define HV_FT_SIZE  (HV_FT_ONE0 + 128)
/* FAST tables of the 4x4 block trials with one lane per sample (hv_tu4_px / hv_tok4_px), built by hv_set_mode */
struct HvFx4 {
    u16 sig4[3][16];                                        /* per scan kind and scan position: sig_coeff_flag context class | earlier positions of the class (HvSig4Info) */
    u8  scan4[3][16];                                       /* per scan kind: raster position (y << 2 | x) -> scan position */
    u8  unscan4[3][16];                                     /* ... and scan position -> raster position */
    int d4[16];                                             /* the 4-point DST  [ref :391-396] */
};

struct alignas(16) HvWaveScr {                              /* private to one wavefront */
    short scr[64 * 34];                                     /* [slot][row][N+2]: transposition buffer           */
    u8    pred[64 * 32];                                    /* [slot][row][N] prediction                        */
    u8    bord[672];                                        /* [slot][raw|smoothed][4N+4] reference samples     */
    u8    edge[256];                                        /* [slot][4][H] TU-split chains: right columns of sub-blocks 0,2 and bottom rows of 0,1 */
    u8    line[256];                                        /* [slot][3N+2] angular modes: the mode's reference line r(-N..2N) */
};

struct alignas(16) HvLds {
    /* tables (copied from constant memory so that per-lane divergent lookups are LDS reads) */
    unsigned lps4[64];
    u8  lpsnext[64];
    union {                                                 /* rebuilt when the workgroup changes mode (hv_set_mode) */
        alignas(8) unsigned ctab[128][2];                   /* EXACT: [context byte]: { four LPS ranges (one byte per range quarter), next byte after LPS | after MPS << 8 } */
        alignas(8) unsigned ctab2[256][2];                  /* FAST: [context byte << 1 | bin]: { four LPS ranges, next context byte | is-LPS << 8 } */
    };
    u8  sigtab[35][16];                                     /* sig_coeff_flag context per scan position of a 4x4 group, by (size class, scan, neighbours, first group) */                         /* context byte after an MPS / LPS  [ref HEVCe.c:701-702] */
    u8  d4pos[16], d4idx[16];                               /* diagonal scan inside a 4x4 group: idx->(y<<2|x), (y<<2|x)->idx */
    short ang[36], invang[36];                              /* intra angle / inverse angle per mode: a per-lane read of a __constant__ table is a waterfall of scalar loads */
    u8  cgpos[3][64], cgidx[3][64];                         /* group order for 2x2 / 4x4 / 8x8 groups: idx->(gy<<3|gx), gy*ncg+gx->idx */
    u8  ctx_fresh[HEVCE_NCTXP];
    /* committed coder state, entry snapshots per CU level (0: 8x8, 1: 16x16, 2: 32x32) */
    HevceCoderRegs com, ent[3];
    u8  com_ctx[HEVCE_NCTXP], ent_ctx[3][HEVCE_NCTXP];
    /* current CTU */
    u8  org[32][32];
    u8  rec[33][72];                                        /* row 0 = y-1, column HV_RX-1 = x-1; 64+1 columns used */
    u8  map_cu[9][12], map_pm[9][12];                       /* [1+i][1+j]; row 0 / column 0 = neighbours */
    u8  line_cu[HEVCE_MAXDIM / 4];
    /* candidates ("trials") of the CU under evaluation at each level: 0 = 8x8, 1 = 16x16, 2 = 32x32.  A parent's
       candidates are prepared at ITS entry and their arithmetic engines advance in slices during the children's phases. */
    struct Level {
        alignas(4) u8 tctx[HEVCE_NCTXP][HV_TL];                     /* context bytes, one column per candidate */
        HevceCoderRegs tregs[HV_TL];
        int tcost[HV_TL], tsse[HV_TL];
        int tlen[HV_TL], tpos[HV_TL];                       /* list entries; position (entry | chunk << 8), -1 = finished / abandoned */
        int tnlog[HV_TL];                                   /* finished-byte records in the candidate's sink so far */
        HvTuInfo tinfo[HV_TL][4];
        u8  tdead[HV_TL];                                   /* abandoned: partial cost already above best_cost */
        int best_cost;                                      /* lowest cost of a COMPLETED candidate (pruning bound) */
        int split_inc, ml, ma;                              /* neighbour context read at the CU's entry [ref :1369-1373] */
    } lv[3];
    u16 slist0[HV_TL][10], slist1[HV_TL][22], slist2[HV_TL][70];   /* per level and candidate: non-empty slots, slot | chunks << 8 */
    /* NxN sub-search */
    alignas(16) short nxn_lev[35][16];                      /* 16-byte rows: written / read as vectors */
    alignas(16) u8    nxn_rec[35][16];
    int   nxn_sse[35], nxn_last[35], nxn_sig[35];
    alignas(16) short nxn_bestlev[4][16];
    alignas(16) u16 nxn_tok[35][144];                                 /* per candidate: its token stream (header + one group), 16-byte rows */
    u8    nxn_crec[8][8];
    union {
        u8    nxn_line[35][16];                             /* EXACT (one lane per mode): the mode's angular reference line r(-4..9) */
        HvFx4 fx;                                           /* FAST (one lane per sample) */
    };
    u8    nxn_bord[20];                                     /* raw reference samples of the current 4x4: corner, left[8], above[8] */
    int   nxn_modes[4];
    union {
        alignas(16) u16 nxn_sink[35][96];                   /* EXACT: finished-byte records of the chain's engines (a 4x4 block: < 80 bytes) */
        struct {                                            /* FAST: the winners of the CTU's 64 sub-searches, for hv_verify_nxn */
            alignas(16) short nxv_lev[64][16];              /* levels in scan order */
            u16   nxv_info[64];                             /* last position | scan kind << 4 | has levels << 6 */
            short nxv_bits[64];                             /* the bit count the FAST engine gave it */
            u8    ftab[HV_FT_SIZE];                         /* context bytes a FRESH context set passes through (hv_set_mode): the resolved tokens of the NxN sub-search */
            u8    t4c[128][32];                             /* context byte after k <= 4 bins ([byte][1 << k | bins, the first one in bit 0]): candidate 70's rider (it fills the union exactly) */
        };
    };
    int   nxn_ntok[35];                                     /* per mode of the current 4x4 block: tokens | bypass bins << 8 | bits << 20 */
    int   c70_len, c70_byp;                                 /* tokens / bypass bins of candidate 70's stream so far */
    int   c70_rd, nxn_magn[35];                             /* cluster of four: blocks candidate 70's rider has consumed; FAST, per mode of the current 4x4 block: tokens | bypass bins << 8 */
    int   nxq_go, nxq_done;                                 /* cluster of four: 4x4 blocks whose reference samples are laid out; helper wavefronts done, in all */
    int   nxq_tmask;                                        /* ... passes of the current 4x4 block whose block trials are done (hv_nxq_unit) */
    int   c70_done, c70_seg[4], c70_pat;                    /* FAST: stages whose winner's tokens are in the stream (5: header known too); per stage offset | tokens << 16 | cbf << 31; the CU header's four prev_intra_luma_pred_flags */
    u8    nxn_ctx[HEVCE_NCTXP][HV_TL];                      /* context columns of the NxN chain's 35 modes (fresh per 4x4 block) */
    /* control */
    int cl_seq[3];                                          /* cluster mode: messages sent (chain workgroup) / answered (parent workgroup) per mailbox; 8x8 worker: quads accepted */
    int cl_ans8;                                            /* 8x8 CUs reported (chain workgroup) / answered (8x8 workgroup) since the launch */
    struct Quad8 { int ctu; u8 y, x, L, LB, A, AR, ml0, ml2, ma0, ma2, pad_[2]; u8 pm[4][4]; };
    struct Peek8 { int k, w, cost, dk; };                   /* dk: ... and its data have been fetched for CU dk (PeekD, hv_pfd) */
    struct PeekD { int range, nbits; u8 ctx[HEVCE_NCTXP]; u8 rec[64]; short lev[64]; };   /* the winner's coder, contexts, samples, levels */
    union {
        Quad8 q8;                                           /* 8x8 workgroup: the quad it is walking */
        Peek8 pf;                                           /* chain workgroup: the answer to 8x8 CU pf.k has been seen (hv_cl_peek8): its winner and cost */
    };
    int cl_pick, cl_fail;                                   /* parent workgroup: the mailbox to serve next (-1 exit); a hand-off timed out */
    int cl_light, cl_ok;                                    /* all partners share this workgroup's XCD (light hand-offs, see hv_publish_fence); chain workgroup: answer slots whose sender said so */
    int exact;                                              /* 1: EXACT engines (the reference's bit counts by construction); 0: FAST (see hv_fengine_chunk) */
    int nxv_pend, poison_prev;                              /* cluster of four: the previous CTU's NxN winners wait (copied into slist2) to be verified during this CTU; they failed */
    int poison;                                             /* FAST: this CTU's decisions may differ from the reference's: 1 = NxN winner, 2 = the writer says so */
    int wr_done;                                            /* CTUs the writer has finished */
    struct Snap { HevceCoderRegs com; u8 com_ctx[HEVCE_NCTXP]; u8 map_cu8[8], map_pm8[8], rec_col[32], line_cu[8]; } snap[2];   /* what a CTU starts from, [ctu & 1] */
    HevceCoderRegs wsnap; u8 wsnap_ctx[HEVCE_NCTXP];        /* what the writer started its current CTU from */
    int winner, split_sse;
    int l0_active;                                          /* 8x8 candidates whose engines still run in this engine phase */
    int nxn_busy;                                           /* 1 while wavefront 0 walks the NxN chain of the current 8x8 CU */
    /* the stream writer (wavefront 4, one lane): the exact byte-emitting coder of the FINAL syntax, one CTU behind the
       analysis; it re-derives the bins of the previous CTU's decision record  [ref :797-933, 943-1340 for the chosen syntax] */
    HevceCoderRegs wr;                                      /* range, low, nbits, pending bytes, held byte, zero run, bytes written */
    int wr_epb;                                             /* emulation-prevention bytes inserted while writing the current CTU */
    u8  wr_linfo[8], wr_lmode[8], wr_ainfo[8];              /* CU size / mode of the 4x4 blocks left of and above the CTU being written */
    alignas(16) HevceRecord wr_rec;                         /* its decision record, staged from HBM */
    struct WrSt {                                           /* where the writer's coder stands in that CTU's token slots (hv_wr_engine works in slices) */
        int pos, ci, done;                                  /* next slot in coding order (3 per 4x4 z-position: CU header, block header, group), chunk inside it; CTU complete */
        int total, budget;                                  /* chunks of the CTU, chunks per slice */
        int tpc, used;                                      /* 256 x the 100 MHz ticks the coder took per chunk in its last slice; chunks of the CTU coded so far */
    } wst;
#ifdef HEVCE_ANYPROF
    unsigned long long prof[3][8];
    unsigned long long prof_pt;                              /* HEVCE_PROF: running stamp of the chain workgroup's hand-off section (HV_PROFC) */
    unsigned long long prof2[32];                           /* HEVCE_PROF: inside the NxN job (tu jobs, tokenise, engine, select) */
#endif
    int *trace;
    const u16 *qlut;                                        /* the picture's four level tables (hv_qlut) */
    alignas(16) HvWaveScr ws[HV_NWAVES];                    /* 16-byte rows are read / written as vectors */
};

typedef unsigned HvChunk __attribute__((vector_size(16), may_alias));   /* 16 bytes moved as one dwordx4: eight 16-bit tokens / records, packed levels, samples */
typedef unsigned HvPair __attribute__((vector_size(8), may_alias));     /* 8 bytes moved as one dwordx2 */
struct HvCod { int range; unsigned low; int nbits, nbytes, bufbyte, zeros, nout;
               int nev, zf; };   /* lockstep engines: finished bytes recorded but not yet accounted (hv_cod_replay), "a zero byte or a carry may be among them" */

/* The workgroup's LDS image.  The large routines below are separate (__noinline__) functions: they bind to this
   symbol directly instead of taking the image by reference, so that every access stays an LDS (ds_*) access. */
#ifdef HEVCE_EMU
extern HvLds *emu_lds_ptr;
#define HV_S (*emu_lds_ptr)
static inline int hv_uni(int v) { return v; }
#else
__shared__ HvLds hv_lds_image;
#define HV_S hv_lds_image
static HD int hv_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }     /* value known to be wave-uniform -> SGPR */
#endif

/* --------------------------------------------------------------------------------------------- */
/* small helpers                                                                                   */
/* --------------------------------------------------------------------------------------------- */
static HD int hv_min(int a, int b) { return a < b ? a : b; }
static HD int hv_max(int a, int b) { return a > b ? a : b; }
static HD int hv_abs(int a) { return a < 0 ? -a : a; }
static HD int hv_clip16(int v) { return hv_min(hv_max(v, -32768), 32767); }
static HD int hv_clip8(int v) { return hv_min(hv_max(v, 0), 255); }

/* context byte (state << 1 | mps) of context i at the start of a slice  [ref :727-735]; after one more bin  [ref :914-933] */
static HD int hv_ctx_fresh_value(int i, int q) {
    const int iv = hv_ctxinit[i], qp = 6 * q + 4;
    int st = ((((iv >> 4) * 5 - 45) * qp) >> 4) + ((iv & 15) << 3) - 16;
    st = hv_min(hv_max(st, 1), 126);
    return st >= 64 ? (((st - 64) << 1) | 1) : ((63 - st) << 1);
}
static HD int hv_ctx_next(int cv, int bin) {
    const int st = cv >> 1, mps = cv & 1;
    if (bin != mps) return st == 0 ? (mps ^ 1) : ((hv_lpsnext[st] << 1) | mps);
    return st < 62 ? cv + 2 : cv;
}

/* [ref HEVCe.c:177-185] */
static HD int hv_rdcost(int q, int dist, int bits) {
    const int wd = (q < 3) ? 11 : (q == 3 ? 5 : 1);
    const int wb = (q == 0) ? 1 : (q == 1) ? 4 : (q == 2) ? 16 : (q == 3) ? 29 : 23;
    const int gd = (q < 3) ? HV_I32MAX / 11 : (q == 3 ? HV_I32MAX / 5 : HV_I32MAX);                  /* I32MAX / weight: no division at run time */
    const int gb = (q == 0) ? HV_I32MAX : (q == 1) ? HV_I32MAX / 4 : (q == 2) ? HV_I32MAX / 16 : (q == 3) ? HV_I32MAX / 29 : HV_I32MAX / 23;
    const int a = (gd <= dist) ? HV_I32MAX : wd * dist;
    const int b = (gb <= bits) ? HV_I32MAX : wb * bits;
    return (HV_I32MAX - a <= b) ? HV_I32MAX : a + b;
}

/* scan order  [ref HEVCe.c:1127-1151]: kind 0 diagonal, 1 horizontal, 2 vertical (the latter two only for 4x4 / 8x8) */
static HD int hv_scan_kind(int n, int mode) {
    if (n <= 8) { if (hv_abs(mode - 26) <= 4) return 1; if (hv_abs(mode - 10) <= 4) return 2; }
    return 0;
}
/* lg = log2(n); tables index lg-3 for 8,16,32; a 4x4 block is a single group */
static HD int hv_scan_idx(const HvLds &S, int lg, int kind, int y, int x) {
    const int gy = y >> 2, gx = x >> 2, iy = y & 3, ix = x & 3;
    if (kind == 1) return ((gy * 2 + gx) << 4) + iy * 4 + ix;
    if (kind == 2) return ((gx * 2 + gy) << 4) + ix * 4 + iy;
    const int cg = (lg == 2) ? 0 : S.cgidx[lg - 3][(gy << (lg - 2)) + gx];
    return (cg << 4) + S.d4idx[(iy << 2) | ix];
}
static HD void hv_scan_cg(const HvLds &S, int lg, int kind, int cg, int &gy, int &gx) {
    if (lg == 2) { gy = 0; gx = 0; }
    else if (kind == 1) { gy = cg >> 1; gx = cg & 1; }
    else if (kind == 2) { gx = cg >> 1; gy = cg & 1; }
    else { const int p = S.cgpos[lg - 3][cg]; gy = p >> 3; gx = p & 7; }
}
static HD void hv_scan_in(const HvLds &S, int kind, int ii, int &iy, int &ix) {
    if (kind == 1) { iy = ii >> 2; ix = ii & 3; }
    else if (kind == 2) { ix = ii >> 2; iy = ii & 3; }
    else { const int p = S.d4pos[ii]; iy = p >> 2; ix = p & 3; }
}

/* --------------------------------------------------------------------------------------------- */
/* arithmetic coder in counting mode (registers only)  [ref HEVCe.c:797-933]                       */
/* --------------------------------------------------------------------------------------------- */
static HD void hv_cod_load(HvCod &c, const HevceCoderRegs &r) {
    c.range = r.range; c.low = r.low; c.nbits = r.nbits; c.nbytes = r.nbytes; c.bufbyte = r.bufbyte; c.zeros = r.zeros; c.nout = r.nout;
    c.nev = 0; c.zf = 0;
}
static HD void hv_cod_store(HevceCoderRegs &r, const HvCod &c) {
    r.range = c.range; r.low = c.low; r.nbits = c.nbits; r.nbytes = c.nbytes; r.bufbyte = c.bufbyte; r.zeros = c.zeros; r.nout = c.nout; r.pad = 0;
}
static HD void hv_cod_fresh(HvCod &c) { c.range = 510; c.low = 0; c.nbits = 23; c.nbytes = 0; c.bufbyte = 0xff; c.zeros = 0; c.nout = 0; c.nev = 0; c.zf = 0; }
/* bits so far  [ref :835-837].  With recorded-but-unaccounted bytes (nev > 0) this is a LOWER bound: the emulation-
   prevention bytes among them, which the reference counts, are only known after hv_cod_replay; it is exact when !zf. */
static HD int  hv_cod_bits(const HvCod &c) { return 8 * (c.nout + c.nbytes + c.nev) + 23 - c.nbits; }
static HD int  hv_regs_bits(const HevceCoderRegs &c) { return 8 * (c.nout + c.nbytes) + 23 - c.nbits; }

static HD void hv_cod_out(HvCod &c, int byte) {                                                            /* [ref :821-832] */
    byte &= 0xff;
    if (c.zeros >= 2 && byte <= 3) { c.nout++; c.zeros = 0; }      /* emulation-prevention byte */
    c.nout++;
    c.zeros = byte ? 0 : c.zeros + 1;
}
static HD void hv_cod_carry(HvCod &c) {                                                                    /* [ref :859-879] */
    if (c.nbits < 12) {
        const int lead = (int)(c.low >> (24 - c.nbits));
        c.nbits += 8;
        c.low &= 0xffffffffu >> c.nbits;
        if (lead == 0xff) {
            c.nbytes++;
        } else if (c.nbytes > 0) {
            const int carry = lead >> 8, fill = (0xff + carry) & 0xff;
            hv_cod_out(c, c.bufbyte + carry);
            c.bufbyte = lead & 0xff;
            for (; c.nbytes > 1; c.nbytes--) hv_cod_out(c, fill);
        } else {
            c.nbytes = 1; c.bufbyte = lead;
        }
    }
}
/* The lockstep engines split that state machine in two.  Per token (hv_cod_carry_rec) a finished byte is only RECORDED:
   its 9-bit value (carry + byte) goes to a per-candidate sink and the count goes up -- a handful of instructions, where
   the full byte bookkeeping (pending byte, runs of 0xff, emulation prevention) costs as much as the arithmetic itself
   and, with dozens of candidates per wavefront, would be executed at every token.  hv_cod_replay() then runs the exact
   bookkeeping once per recorded byte, i.e. for about one token in ten.
   Start of a span: hv_cod_span(); zf stays 0 as long as no zero byte can have been emitted and no carry occurred -- then no
   emulation-prevention byte can have been inserted and hv_cod_bits() is exact even before the replay. */
static HD void hv_cod_span(HvCod &c) {
    c.nev = 0;
    c.zf = (c.zeros > 0) | ((c.bufbyte & 0xff) == 0) | ((c.bufbyte & 0xff) == 0xff) | (c.nbytes > 1);
}
template <typename SinkP> static HD void hv_cod_carry_rec(HvCod &c, SinkP sink) {
    if (c.nbits < 12) {                                          /* one region under the lanes' mask: a store and four updates */
        const unsigned lead = c.low >> (24 - c.nbits);
        sink[c.nev] = (u16)lead;
        c.nbits += 8;
        c.low &= 0xffffffffu >> c.nbits;
        c.zf |= (int)(((lead - 1u) | lead) >> 8);                 /* non-zero iff the byte is 0 or a carry came with it (lead is 9 bits) */
        c.nev++;
    }
}
/* exact bookkeeping of one finished byte (`on` = there is one): hv_cod_carry without the register part; the common case
   straight-line, the flush of a run of 0xff bytes a (rare) branch */
static HD void hv_cod_byte(HvCod &c, int lead, int on) {
    const int rare = on & (c.nbytes > 1) & (lead != 0xff);
    const int go = on & (rare ^ 1);
    {
        const int isff = lead == 0xff, emit = go & (isff ^ 1) & (c.nbytes > 0);
        const int byte = (c.bufbyte + (lead >> 8)) & 0xff;
        const int epb = emit & (c.zeros >= 2) & (byte <= 3);
        c.nout += emit + epb;
        c.zeros = emit ? (byte ? 0 : (epb ? 1 : c.zeros + 1)) : c.zeros;
        c.nbytes = go ? (isff ? c.nbytes + 1 : 1) : c.nbytes;
        c.bufbyte = (go & (isff ^ 1)) ? (emit ? (lead & 0xff) : lead) : c.bufbyte;   /* first byte ever: stored unmasked, as the reference does */
    }
    if (__builtin_expect(rare, 0)) {
        const int carry = lead >> 8, fill = (0xff + carry) & 0xff;
        hv_cod_out(c, c.bufbyte + carry);
        c.bufbyte = lead & 0xff;
        for (; c.nbytes > 1; c.nbytes--) hv_cod_out(c, fill);
    }
}
/* sink8 = the sink as 16-byte chunks of eight records; the c.nev records to account start at record `first` of it */
template <typename ChunkP> static HD void hv_cod_replay(HvCod &c, ChunkP sink8, int first = 0) {
    const int lo = first, hi = first + c.nev;
    c.nev = 0; c.zf = 0;
    HvChunk nxt = sink8[lo >> 3];
    for (int base = lo & ~7; base < hi; base += 8) {
        const HvChunk ch = nxt;
        nxt = sink8[(base >> 3) + 1];                             /* the sink may be in HBM: fetch ahead */
#pragma unroll
        for (int j = 0; j < 8; j++) hv_cod_byte(c, (int)((ch[j >> 1] >> ((j & 1) << 4)) & 0xffffu), base + j >= lo && base + j < hi);
    }
}
/* context-coded bin; cx points at this trial's context byte  [ref :914-933] */
static HD void hv_bin(const HvLds &S, HvCod &c, u8 *cx, int bin) {
    const int v = *cx, st = v >> 1, mps = v & 1;
    const int lps = (int)((S.lps4[st] >> (((c.range >> 6) & 3) << 3)) & 0xff);
    c.range -= lps;
    if ((bin != 0) != (mps != 0)) {
        const int sh = (lps < 8) ? 6 : hv_clz((unsigned)(lps >> 3)) - 26;       /* renormalisation shift 6..1 */
        *cx = (u8)(st == 0 ? (mps ^ 1) : ((S.lpsnext[st] << 1) | mps));
        c.low = (c.low + (unsigned)c.range) << sh;
        c.range = lps << sh;
        c.nbits -= sh;
    } else {
        if (st < 62) *cx = (u8)(v + 2);
        if (c.range < 256) { c.low <<= 1; c.range <<= 1; c.nbits--; }
    }
    hv_cod_carry(c);
}
static HD void hv_bypass(HvCod &c, int bins, int len) {                                                    /* [ref :899-911] */
    bins &= (1 << len) - 1;
    while (len > 0) {
        const int n = hv_min(len, 8);
        len -= n;
        c.low = (c.low << n) + (unsigned)c.range * (unsigned)((bins >> len) & ((1 << n) - 1));
        c.nbits -= n;
        hv_cod_carry(c);
    }
}
static HD void hv_terminate(HvCod &c, int bin) {              /* [ref :882-896] */
    c.range -= 2;
    if (bin) { c.low = (c.low + (unsigned)c.range) << 7; c.range = 256; c.nbits -= 7; }
    else if (c.range < 256) { c.low <<= 1; c.range <<= 1; c.nbits--; }
    hv_cod_carry(c);
}
