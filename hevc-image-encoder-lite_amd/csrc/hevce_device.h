/*
 * hevce_device.h -- the per-CTU analysis loop of an HEVC intra still-picture encoder, written for
 * CDNA4 (gfx950): ONE WORKGROUP (8 wavefronts of 64) WALKS ONE IMAGE'S CTUs in raster order with all
 * coder / neighbour state resident in LDS.  Bit-exact with lidongxuan/HEVC-image-encoder-lite
 * (reference src/HEVCe.c); each routine cites the reference lines whose results it reproduces.
 *
 * Why one workgroup per image: every RD decision is costed with the live CABAC state, so an image is a
 * single dependency chain.  Parallelism is taken (a) across images = workgroups,
 * (b) inside a CU: the 35 prediction modes x {one TU, four TUs, NxN} run as wave-level "TU jobs"
 * (lane = one column / row of a transform block) and as lane-level trial encodes (lane = one candidate).
 *
 * The same source is compiled by tests/emu (HEVCE_EMU) for a CPU fiber emulation used by the CPU test
 * suite to debug the control flow; that build is test infrastructure and is never part of the library.
 */
#pragma once
#include "hevce_common.h"
#include "hevce_tables.h"

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
    struct Peek8 { int k, w, cost; };
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

/* --------------------------------------------------------------------------------------------- */
/* quantiser  [ref HEVCe.c:526-595]: the level of a coefficient is looked up                                                   */
/* --------------------------------------------------------------------------------------------- */
#include "hevce_quant.h"
/* The reference tries level, level-1, level-2 and keeps the cheapest in w1 * distortion + w2 * rate  [ref :566-580]; for a
   fixed qpd6 and transform size that choice depends on the coefficient's magnitude alone, so the host tabulates
   hv_quant1() once (im.qlut: four tables of HEVCE_QLUT_N 16-bit levels for the picture's qpd6, 256 KB, L2-resident) and the
   block jobs gather from it: one load per coefficient instead of ~100 VALU instructions.  |coefficient| <= 32640 for 8-bit
   samples (the bound the 16-bit staging of the transforms already relies on); the index is clamped all the same.
   d = the scaled magnitude for the group zero-out test  [ref :555, 585-589]. */
typedef const HV_GLOBAL u16 *HvQlut;
template <int LG> static HD HvQlut hv_qlut(const HvLds &S) { return (HvQlut)S.qlut + (LG - 2) * HEVCE_QLUT_N; }
static HD int hv_quant_lut(HvQlut lut, int av, int &d) {
    d = av << 14;
    return lut[hv_min(av, HEVCE_QLUT_N - 1)];
}

/* --------------------------------------------------------------------------------------------- */
/* TU job: 64/N candidate modes of one NxN transform block per wavefront                            */
/*   lane = slot*N + j ; j is the COLUMN it owns in column passes and the ROW it owns in row passes  */
/* --------------------------------------------------------------------------------------------- */
template <int N> struct HvTm;
template <> struct HvTm<4>  { static HD const int *f() { return hv_tm.f4;  } static HD const int *t() { return hv_tm.t4;  } enum { LG = 2 }; };
template <> struct HvTm<8>  { static HD const int *f() { return hv_tm.f8;  } static HD const int *t() { return hv_tm.t8;  } enum { LG = 3 }; };
template <> struct HvTm<16> { static HD const int *f() { return hv_tm.f16; } static HD const int *t() { return hv_tm.t16; } enum { LG = 4 }; };
template <> struct HvTm<32> { static HD const int *f() { return hv_tm.f32; } static HD const int *t() { return hv_tm.t32; } enum { LG = 5 }; };

struct HvTuOut { int sse, last, cbf; u64 cgsig; };          /* block-wide values, identical in all lanes of a slot */

/* The 8/16/32-point core transforms as partial butterflies.  Rows of even frequency are symmetric, rows of odd frequency
   antisymmetric, and the even rows of the N-point matrix are the N/2-point matrix (hv_basis): N*N/3 multiplications
   instead of N*N, with exactly the reference's integer sums (it multiplies the whole matrix out, HEVCe.c:469-516; integer
   addition is associative, nothing here can overflow: |inputs| < 2^16, |coefficients| <= 90).
   x[0..M) = the M inputs (forward) / the inputs of frequency 0, S, 2S, ... (inverse, S = N/M); all indices compile-time. */
template <int N, int M> struct HvBfly {
    enum { S = N / M, H = M / 2 };
    /* forward: out[f * S] for f = 0..M-1 from x (destroyed) */
    static HD void fwd(int *x, int *out) {
        int o[H];
#pragma unroll
        for (int k = 0; k < H; k++) { const int a = x[k], b = x[M - 1 - k]; x[k] = a + b; o[k] = a - b; }
#pragma unroll
        for (int f = 1; f < M; f += 2) {
            int acc = 0;
#pragma unroll
            for (int k = 0; k < H; k++) acc += hv_mul24(hv_basis(f * S * (32 / N), k * 1), o[k]);
            out[f * S] = acc;
        }
        HvBfly<N, H>::fwd(x, out);
    }
    /* inverse: out[0..M) (sample positions of the M-point transform) from the inputs in[f * S], f = 0..M-1 */
    static HD void inv(const int *in, int *out) {
        int e[H];
        HvBfly<N, H>::inv(in, e);
#pragma unroll
        for (int n = 0; n < H; n++) {
            int acc = 0;
#pragma unroll
            for (int f = 1; f < M; f += 2) acc += hv_mul24(hv_basis(f * S * (32 / N), n), in[f * S]);
            out[n] = e[n] + acc; out[M - 1 - n] = e[n] - acc;
        }
    }
};
template <int N> struct HvBfly<N, 1> {
    static HD void fwd(int *x, int *out) { out[0] = 64 * x[0]; }
    static HD void inv(const int *in, int *out) { out[0] = 64 * in[0]; }
};

/* reconstructed sample at CTU-relative (yy,xx).  Inside the CU under trial it comes from the candidate's own partial
   reconstruction: `inner` is either the full cu_n x cu_n block (NxN sub-search) or, when `edges` is set, the chain's edge
   store [4][H]: right columns of sub-blocks 0 and 2, bottom rows of sub-blocks 0 and 1 -- the only inner samples a later
   sub-block ever reads (they all lie on row H-1 or column H-1 of the CU). */
static HD int hv_px(const HvLds &S, const HV_LDSP u8 *inner, int edges, int cu_y, int cu_x, int cu_n, int yy, int xx) {
    if (inner && yy >= cu_y && yy < cu_y + cu_n && xx >= cu_x && xx < cu_x + cu_n) {
        const int ry = yy - cu_y, rx = xx - cu_x;
        if (!edges) return inner[ry * cu_n + rx];
        const int h = cu_n >> 1;
        if (rx == h - 1) return inner[(ry < h ? 0 : 1) * h + (ry & (h - 1))];          /* right column of sub-block 0 / 2 */
        return inner[(2 + (rx < h ? 0 : 1)) * h + (rx & (h - 1))];                       /* bottom row of sub-block 0 / 1   */
    }
    return S.rec[yy + 1][xx + HV_RX];
}
/* which (size, mode) pairs read the smoothed reference samples  [ref HEVCe.c:274-280] */
static HD int hv_smoothed(int n, int m) {
    if (n == 4 || m == 1) return 0;
    if (m == 0) return 1;
    const int d = hv_min(hv_abs(m - 10), hv_abs(m - 26));
    return n == 8 ? d > 7 : n == 16 ? d > 1 : d > 0;
}

#ifdef HEVCE_PROF_TU
#define HV_TUT(ph) do { if (N == HEVCE_PROF_TU && lane == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); HV_S.prof2[16 + (ph)] += n_ - tt_; tt_ = n_; } } while (0)
#define HV_TUT0() unsigned long long tt_ = __builtin_amdgcn_s_memtime()
#else
#define HV_TUT(ph) do { } while (0)
#define HV_TUT0() do { } while (0)
#endif
template <int N, typename LevP, typename RecP>
static HD HvTuOut hv_tu_job(int wave, int lane, int q,
                            int ty, int tx,                         /* block origin, CTU-relative            */
                            int cu_y, int cu_x, int cu_n, const HV_LDSP u8 *inner, int edges,
                            int aL, int aLB, int aA, int aAR,       /* neighbour availability                */
                            int mode,                               /* this slot's mode, or -1 (idle slot)   */
                            LevP lev_dst,                           /* N*N levels in scan order (per slot)   */
                            RecP rec_dst, int rec_stride,           /* reconstruction (per slot)             */
                            HV_LDSP u8 *ecol = nullptr, HV_LDSP u8 *erow = nullptr) {   /* chains: this block's right column / bottom row */
    HvLds &S = HV_S;
    HvWaveScr &W = S.ws[hv_uni(wave)];
    q = hv_uni(q); ty = hv_uni(ty); tx = hv_uni(tx); cu_y = hv_uni(cu_y); cu_x = hv_uni(cu_x); cu_n = hv_uni(cu_n);
    aL = hv_uni(aL); aLB = hv_uni(aLB); aA = hv_uni(aA); aAR = hv_uni(aAR);
    constexpr int LG = HvTm<N>::LG, RS = N + 2, BS = 4 * N + 4;
    const int slot = lane / N, j = lane % N;
    const int m = mode < 0 ? 1 : mode;
    short *scr = W.scr + slot * N * RS;
    u8 *pred = W.pred + slot * N * N;
    u8 *braw = W.bord + slot * 2 * BS, *bsm = braw + BS;         /* [0] corner, [1..2N] left, [1+2N..4N] above */

    HV_TUT0();
    /* ---- reference samples  [ref HEVCe.c:196-257] ---- */
    {
        int corner;
        if (aL && aA)  corner = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx - 1);
        else if (aL)   corner = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty, tx - 1);
        else if (aA)   corner = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx);
        else           corner = 128;
        const int l0 = aL ? hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty + j, tx - 1) : corner;
        const int a0 = aA ? hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx + j) : corner;
        int l1, a1;
        if (aLB) l1 = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty + N + j, tx - 1);
        else     l1 = aL ? hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty + N - 1, tx - 1) : corner;
        if (aAR) a1 = hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx + N + j);
        else     a1 = aA ? hv_px(S, inner, edges, cu_y, cu_x, cu_n, ty - 1, tx + N - 1) : corner;
        if (j == 0) braw[0] = (u8)corner;
        braw[1 + j] = (u8)l0; braw[1 + N + j] = (u8)l1;
        braw[1 + 2 * N + j] = (u8)a0; braw[1 + 3 * N + j] = (u8)a1;
    }
    WAVE_SYNC();
    if constexpr (N > 4) {                                       /* 4x4 blocks never read the smoothed samples (hv_smoothed) */
        const u8 *l = braw + 1, *a = braw + 1 + 2 * N;
        const int c = braw[0];
        if (j == 0) {
            bsm[0] = (u8)((2 + l[0] + a[0] + 2 * c) >> 2);
            bsm[1] = (u8)((2 + 2 * l[0] + l[1] + c) >> 2);
            bsm[1 + 2 * N] = (u8)((2 + 2 * a[0] + a[1] + c) >> 2);
        } else {
            bsm[1 + j] = (u8)((2 + 2 * l[j] + l[j - 1] + l[j + 1]) >> 2);
            bsm[1 + 2 * N + j] = (u8)((2 + 2 * a[j] + a[j - 1] + a[j + 1]) >> 2);
        }
        const int k = N + j;
        if (j == N - 1) { bsm[1 + k] = l[k]; bsm[1 + 2 * N + k] = a[k]; }
        else {
            bsm[1 + k] = (u8)((2 + 2 * l[k] + l[k - 1] + l[k + 1]) >> 2);
            bsm[1 + 2 * N + k] = (u8)((2 + 2 * a[k] + a[k - 1] + a[k + 1]) >> 2);
        }
    }
    /* DC value of every slot (wave-uniform code: shuffles must not sit in the per-mode branches) */
    int dcsum = braw[1 + j] + braw[1 + 2 * N + j];
    for (int s = 1; s < N; s <<= 1) dcsum += hv_shfl_xor(dcsum, s);
    WAVE_SYNC();
    /* angular modes: the slot's reference line r(k), k = -N..2N, laid out once by its N lanes (r(k>0) main side, r(0) corner,
       r(k<0) the side samples projected through the inverse angle [ref :353-364]); the pixel loop then only indexes it.
       Every lane takes part (the other mode forms just do not read it): no wave-level step inside a per-mode branch */
    u8 *line = W.line + slot * (3 * N + 2) + N;
    {
        const u8 *b = hv_smoothed(N, m) ? bsm : braw;
        const bool horiz = m < 18;
        const int inv = S.invang[m];
        const u8 *mainr = horiz ? b + 1 : b + 1 + 2 * N, *sider = horiz ? b + 1 + 2 * N : b + 1;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const int k = e * N + j - N + 1;                     /* -N+1 .. 2N */
            int v;
            if (k > 0) v = mainr[k - 1];
            else if (k == 0) v = b[0];
            else v = sider[hv_min(hv_max(((128 - inv * k) >> 8) - 1, 0), 2 * N - 1)];   /* entries outside the projected range are never read */
            line[k] = (u8)v;
        }
    }
    WAVE_SYNC();

    HV_TUT(0);
    /* ---- prediction of column j  [ref HEVCe.c:262-381]; residual goes to scr, prediction to pred ---- */
    {
        const u8 *b = hv_smoothed(N, m) ? bsm : braw;
        const u8 *l = b + 1, *a = b + 1 + 2 * N;
        const int c = b[0];
        const bool edge = N <= 16;
        const int ox = tx + j;
        if constexpr (N == 4) {
            /* 4x4 (the TU-split chains of an 8x8 CU: sixteen different modes in a wavefront): every form for every sample, the mode's
               own selected -- a branch per form would have the wavefront walk planar, DC, the two filtered pure directions and the
               angular form one after the other (measured: the group with modes 0..15 took 4.5 k cycles longer per CU) */
            const bool horiz = m < 18, planar = m == 0, isdc = m == 1, hor = m == 10, ver = m == 26;
            const int ang = S.ang[m];
            const int aj = a[j], lN = l[N], aN = a[N], dc = (dcsum + N) >> (LG + 1);
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int li = l[i];
                const int step = horiz ? j : i, along = horiz ? i : j;
                const int off = ang * (step + 1), oi = off >> 5, of = off & 31;
                const int k1 = oi + along + 1, k2 = hv_min(k1 + 1, 2 * N);
#ifdef HEVCE_EMU
                if (k1 < -N + 1 || k1 > 2 * N) { fprintf(stderr, "emu: tu4 line index %d (mode %d ang %d step %d along %d)\n", k1, m, ang, step, along); abort(); }
#endif
                int v = ((32 - of) * line[k1] + of * line[k2] + 16) >> 5;
                const int pl = (N + (N - 1 - j) * li + (j + 1) * aN + (N - 1 - i) * aj + (i + 1) * lN) >> (LG + 1);
                const int dv = (i == 0 && j == 0) ? (2 + 2 * dc + l[0] + a[0]) >> 2 : (i == 0) ? (2 + 3 * dc + aj) >> 2 : (j == 0) ? (2 + 3 * dc + li) >> 2 : dc;
                v = hor ? ((i == 0) ? hv_clip8(li + ((aj - c) >> 1)) : li) : v;
                v = ver ? ((j == 0) ? hv_clip8(aj + ((li - c) >> 1)) : aj) : v;
                const int p = planar ? pl : (isdc ? dv : v);
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else if (m == 0) {
            const int aj = a[j], lN = l[N], aN = a[N];
            for (int i = 0; i < N; i++) {
                const int p = (N + (N - 1 - j) * l[i] + (j + 1) * aN + (N - 1 - i) * aj + (i + 1) * lN) >> (LG + 1);
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else if (m == 1) {
            const int dc = (dcsum + N) >> (LG + 1);
            for (int i = 0; i < N; i++) {
                int p = dc;
                if (edge) {
                    if (i == 0 && j == 0) p = (2 + 2 * dc + l[0] + a[0]) >> 2;
                    else if (i == 0)      p = (2 + 3 * dc + a[j]) >> 2;
                    else if (j == 0)      p = (2 + 3 * dc + l[i]) >> 2;
                }
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else if (m == 10) {
            for (int i = 0; i < N; i++) {
                int p = l[i];
                if (edge && i == 0) p = hv_clip8(p + ((a[j] - c) >> 1));
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else if (m == 26) {
            for (int i = 0; i < N; i++) {
                int p = a[j];
                if (edge && j == 0) p = hv_clip8(p + ((l[i] - c) >> 1));
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        } else {
            /* angular: two neighbours on the mode's reference line, 1/32-sample weights  [ref :342-380] */
            const bool horiz = m < 18;
            const int ang = S.ang[m];
            for (int i = 0; i < N; i++) {
                const int step = horiz ? j : i, along = horiz ? i : j;
                const int off = ang * (step + 1), oi = off >> 5, of = off & 31;
                const int k1 = oi + along + 1, k2 = hv_min(k1 + 1, 2 * N);   /* weight 0 at 2N+1 (modes 2 / 34): any in-range sample */
                const int p = ((32 - of) * line[k1] + of * line[k2] + 16) >> 5;
                pred[i * N + j] = (u8)p; scr[i * RS + j] = (short)((int)S.org[ty + i][ox] - p);
            }
        }
    }

    HV_TUT(1);
    /* ---- forward transform, columns then rows  [ref HEVCe.c:497-516] ---- */
    int v[N];
    {
        constexpr int s1 = LG - 1;
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = scr[k * RS + j];
        if constexpr (N >= 8) {
            int out[N];
            HvBfly<N, N>::fwd(v, out);
#pragma unroll
            for (int i = 0; i < N; i++) scr[i * RS + j] = (short)((out[i] + (1 << s1 >> 1)) >> s1);
        } else {
            const int *tab = HvTm<N>::f();
            for (int i = 0; i < N; i++) {
                int acc = 1 << s1 >> 1;
#pragma unroll
                for (int k = 0; k < N; k++) acc += hv_mul24(tab[i * N + k], v[k]);
                scr[i * RS + j] = (short)(acc >> s1);
            }
        }
    }
    WAVE_SYNC();
    HV_TUT(2);
    /* ---- row pass + quantiser + dequantiser on row j  [ref HEVCe.c:540-615] ---- */
    const int kind = hv_scan_kind(N, m);
    int last = -1, nzmask_lo = 0, nzmask_hi = 0;
    {
        constexpr int s2 = LG - 1 + 7;
        const int sh = 21 - LG + q, thr = 9 << sh >> 2, qsh = 7 - LG + q;
        const HvQlut lut = hv_qlut<LG>(S);
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = scr[j * RS + k];
        const int *tab = HvTm<N>::f();
        if constexpr (N >= 8) {
            /* all N coefficients of the row at once; they wait in the row's own scratch line (16 bits: |coefficient| <= 32640
               for 8-bit samples, the bound the first pass already relies on) until their 4x4 group's turn; their levels are
               requested from the table right away (N gathers in flight), v[] keeps the magnitudes */
            int out[N];
            HvBfly<N, N>::fwd(v, out);
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int coef = (out[i] + (1 << s2 >> 1)) >> s2;
                scr[j * RS + i] = (short)coef;
                int d;
                v[i] = hv_quant_lut(lut, hv_abs(coef), d);
            }
        }
        for (int g = 0; g < N / 4; g++) {
            int lv[4], sum = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int cidx = g * 4 + e;
                int coef;
                if constexpr (N >= 8) coef = scr[j * RS + cidx];
                else {
                    int acc = 1 << s2 >> 1;
#pragma unroll
                    for (int k = 0; k < N; k++) acc += hv_mul24(tab[cidx * N + k], v[k]);
                    coef = acc >> s2;
                }
                int d = hv_abs(coef) << 14, mag;
                if constexpr (N >= 8) mag = v[cidx]; else mag = hv_quant_lut(lut, hv_abs(coef), d);
                lv[e] = coef < 0 ? -mag : mag;
                sum += hv_min(d, thr);
            }
            sum += hv_shfl_xor(sum, 1);                       /* the 4x4 group spans rows j&~3 .. j|3 = 4 adjacent lanes */
            sum += hv_shfl_xor(sum, 2);
            /* levels go out in scan order.  The 16 levels of this 4x4 group live in four adjacent lanes; they are put in scan
               order through a 32-byte LDS tile per group (the reference-sample buffer is free by now), so that every lane
               stores 8 contiguous bytes instead of four scattered 16-bit values */
            int any = 0, cgs = 0;
            HV_LDSP short *stage = (HV_LDSP short *)W.bord + (lane >> 2) * 16;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (sum < thr) lv[e] = 0;
                any |= lv[e];
                const int x = g * 4 + e;
                const int si = hv_scan_idx(S, LG, kind, j, x);
                if (lv[e] && si > last) last = si;
                stage[si & 15] = (short)lv[e];
                cgs = si >> 4;
                scr[j * RS + x] = (short)hv_clip16((int)((unsigned)lv[e] << qsh));
            }
            WAVE_SYNC();
            {
                const HvPair four = *(const HV_LDSP HvPair *)(stage + (lane & 3) * 4);
                if (mode >= 0) *(HV_GLOBAL HvPair *)((HV_GLOBAL short *)lev_dst + cgs * 16 + (lane & 3) * 4) = four;
            }
            WAVE_SYNC();
            any = any != 0;
            any |= hv_shfl_xor(any, 1);
            any |= hv_shfl_xor(any, 2);
            if (any) { const int bit = (j >> 2) * 8 + g; if (bit < 32) nzmask_lo |= 1 << bit; else nzmask_hi |= 1 << (bit - 32); }
        }
    }
    for (int s = 1; s < N; s <<= 1) {
        last = hv_max(last, hv_shfl_xor(last, s));
        nzmask_lo |= hv_shfl_xor(nzmask_lo, s);
        nzmask_hi |= hv_shfl_xor(nzmask_hi, s);
    }
    WAVE_SYNC();
    HV_TUT(3);
    /* ---- inverse transform, columns then rows ---- */
    {
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = scr[k * RS + j];
        if constexpr (N >= 8) {
            int out[N];
            HvBfly<N, N>::inv(v, out);
#pragma unroll
            for (int i = 0; i < N; i++) scr[i * RS + j] = (short)hv_clip16((out[i] + 64) >> 7);
        } else {
            const int *tab = HvTm<N>::t();
            for (int i = 0; i < N; i++) {
                int acc = 64;
#pragma unroll
                for (int k = 0; k < N; k++) acc += hv_mul24(tab[i * N + k], v[k]);
                scr[i * RS + j] = (short)hv_clip16(acc >> 7);
            }
        }
    }
    WAVE_SYNC();
    HV_TUT(4);
    int sse = 0;
    {
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = scr[j * RS + k];
        const int *tab = HvTm<N>::t();
        int out[N];
        if constexpr (N >= 8) HvBfly<N, N>::inv(v, out);
#pragma unroll
        for (int x4 = 0; x4 < N; x4 += 4) {                      /* four samples = one 32-bit store */
            unsigned pk = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int x = x4 + e;
                int acc = 2048;
                if constexpr (N >= 8) acc += out[x];
                else {
#pragma unroll
                    for (int k = 0; k < N; k++) acc += hv_mul24(tab[x * N + k], v[k]);
                }
                const int r = hv_clip8(hv_clip16(acc >> 12) + pred[j * N + x]);
                const int d = (int)S.org[ty + j][tx + x] - r;
                sse += d * d;
                pk |= (unsigned)r << (8 * e);
                if (ecol && x == N - 1) ecol[j] = (u8)r;
                if (erow && j == N - 1) erow[x] = (u8)r;
            }
            if (mode >= 0) *(HV_GLOBAL HvRingWord *)((HV_GLOBAL u8 *)rec_dst + j * rec_stride + x4) = pk;
        }
    }
    for (int s = 1; s < N; s <<= 1) sse += hv_shfl_xor(sse, s);
    WAVE_SYNC();
    HV_TUT(5);
    HvTuOut o;
    o.sse = sse; o.cbf = (last >= 0); o.last = last < 0 ? 0 : last;
    o.cgsig = ((u64)(unsigned)nzmask_hi << 32) | (unsigned)nzmask_lo;
    return o;
}

/* availability of the four z-ordered sub-blocks  [ref HEVCe.c:1376-1379] */
static HD void hv_sub_avail(int s, int L, int LB, int A, int AR, int &l, int &lb, int &a, int &ar) {
    l  = (s & 1) ? 1 : L;
    lb = (s == 0) ? L : (s == 2 ? LB : 0);
    a  = (s & 2) ? 1 : A;
    ar = (s == 0) ? A : (s == 1 ? AR : (s == 2 ? 1 : 0));
}

typedef HV_GLOBAL short *HvGLev;   typedef HV_GLOBAL u8 *HvGRec;      /* trial scratch in HBM */
typedef HV_LDSP short *HvLLev;     typedef HV_LDSP u8 *HvLRec;        /* wave / NxN staging in LDS */
struct HvCu { int y, x, n, lg, lvl, L, LB, A, AR; };         /* CTU-relative CU under evaluation; lvl = lg-3 */

/* step 2 jobs: one TU of the CU size  [ref HEVCe.c:1422-1432] */
template <int N>
static HDN void hv_job_single(int wave, int lane, int q, HvCu cu, int grp, HvGLev glev, HvGRec grec) {
    HvLds &S = HV_S;
    HvLds::Level &L = S.lv[hv_uni(cu.lvl)];
    grp = hv_uni(grp);
    const int slot = lane / N, j = lane % N;
    const int mode = grp * (64 / N) + slot;
    const bool ok = mode < HEVCE_NMODES;
    HvTuOut o = hv_tu_job<N, HvGLev, HvGRec>(wave, lane, q, cu.y, cu.x, cu.y, cu.x, cu.n, nullptr, 0, cu.L, cu.LB, cu.A, cu.AR,
                             ok ? mode : -1, glev + (ok ? mode : HV_TL - 1) * 1024, grec + (ok ? mode : HV_TL - 1) * 1024, N);
    if (ok && j == 0) {
        L.tsse[mode] = o.sse;
        L.tinfo[mode][0].cgsig = o.cgsig; L.tinfo[mode][0].last = o.last; L.tinfo[mode][0].cbf = o.cbf;
    }
}
/* step 3 jobs: four TUs of half the CU size, each predicted from the previous ones' reconstruction  [ref :1455-1468].
   The reconstruction goes straight to the candidate's HBM scratch; only the edge samples later sub-blocks read stay in LDS */
template <int H>
static HDN void hv_job_chain(int wave, int lane, int q, HvCu cu, int grp, HvGLev glev, HvGRec grec) {
    HvLds &S = HV_S;
    HvLds::Level &L = S.lv[hv_uni(cu.lvl)];
    HvWaveScr &W = S.ws[hv_uni(wave)];
    grp = hv_uni(grp); cu.n = hv_uni(cu.n); cu.y = hv_uni(cu.y); cu.x = hv_uni(cu.x);
    const int slot = lane / H, j = lane % H, n = cu.n;
    const int mode = grp * (64 / H) + slot;
    const bool ok = mode < HEVCE_NMODES;
    const int t = ok ? 35 + mode : HV_TL - 1;          /* idle slots aim at the unused pad trial */
    HV_LDSP u8 *edge = (HV_LDSP u8 *)(W.edge + slot * 4 * H);
    int total = 0;
    for (int s = 0; s < 4; s++) {
        int l, lb, a, ar;
        hv_sub_avail(s, cu.L, cu.LB, cu.A, cu.AR, l, lb, a, ar);
        const int oy = (s >> 1) * H, ox = (s & 1) * H;
        HV_LDSP u8 *ecol = (s == 0) ? edge : (s == 2 ? edge + H : nullptr);
        HV_LDSP u8 *erow = (s == 0) ? edge + 2 * H : (s == 1 ? edge + 3 * H : nullptr);
        HvTuOut o = hv_tu_job<H, HvGLev, HvGRec>(wave, lane, q, cu.y + oy, cu.x + ox, cu.y, cu.x, n, edge, 1, l, lb, a, ar,
                                 ok ? mode : -1, glev + t * 1024 + s * H * H, grec + t * 1024 + oy * n + ox, n, ecol, erow);
        total += o.sse;
        if (ok && j == 0) { L.tinfo[t][s].cgsig = o.cgsig; L.tinfo[t][s].last = o.last; L.tinfo[t][s].cbf = o.cbf; }
    }
    if (ok && j == 0) L.tsse[t] = total;
    WAVE_SYNC();
}
/* 4x4 TU trial with lane = mode: the whole block lives in this lane's registers (no cross-lane traffic).
   b = raw reference samples (corner, left[8], above[8]); 4x4 blocks never use the smoothed ones  [ref :274-275].
   Writes levels in scan order to lev_dst[16], the reconstruction (raster) to rec_dst[16]; returns SSE.
   [ref :1508-1518 = predict / residual / DST / RDOQ / dequant / inverse DST / reconstruct] */
#ifdef HEVCE_PROF_NXN
#define HV_NXT(ph) do { if ((hv_tid() & 63) == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); HV_S.prof2[16 + (ph)] += n_ - nt_; nt_ = n_; } } while (0)
#define HV_NXT0() unsigned long long nt_ = __builtin_amdgcn_s_memtime()
#else
#define HV_NXT(ph) do { } while (0)
#define HV_NXT0() do { } while (0)
#endif
static HD int hv_tu4_lane(const HvLds &S, int q, const u8 *b, u8 *line_buf, int ty, int tx, int m, HV_LDSP short *lev_dst, HV_LDSP u8 *rec_dst, int &last_out) {
    HV_NXT0();
    /* All five prediction forms are evaluated by every lane and the lane's own is selected at the end: lane = mode, so a
       branch per form would make the wavefront walk all of them anyway, plus the merging of sixteen registers per branch.
       Pure horizontal / vertical (10 / 26) are the angular form with angle 0 plus their edge filter. */
    const u8 *l = b + 1, *a = b + 9;
    const int c = b[0];
    int L[5], A[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { L[i] = l[i]; A[i] = a[i]; }
    int p[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    {
        /* angular: this mode's reference line r(k), k = -4..9, is laid out once (r(k>0) main side, r(0) corner, r(k<0) the
           side samples projected through the inverse angle [ref :353-364]); the pixel loop then only indexes it */
        const bool horiz = m < 18;
        const int ang = S.ang[m], inv = S.invang[m];
        const u8 *mainr = horiz ? l : a, *sider = horiz ? a : l;
        u8 *line = line_buf + 4;
#pragma unroll
        for (int k = -4; k <= 9; k++) {
            int v;
            if (k > 0) v = mainr[k > 8 ? 7 : k - 1];
            else if (k == 0) v = c;
            else v = sider[hv_min(hv_max(((128 - inv * k) >> 8) - 1, 0), 7)];   /* entries outside the projected range are never read */
            line[k] = (u8)v;
        }
#pragma unroll
        for (int st = 0; st < 4; st++) {                         /* st = the index the angle steps along (row for vertical modes) */
            const int off = ang * (st + 1), oi = off >> 5, of = off & 31;
#pragma unroll
            for (int al = 0; al < 4; al++) {
                const int k1 = oi + al + 1;
                const int v = (hv_mul24(32 - of, line[k1]) + hv_mul24(of, line[k1 + 1]) + 16) >> 5;
                p[al * 4 + st] = horiz ? v : p[al * 4 + st];
                p[st * 4 + al] = horiz ? p[st * 4 + al] : v;
            }
        }
    }
    {
        int dc = 4;
#pragma unroll
        for (int i = 0; i < 4; i++) dc += L[i] + A[i];
        dc >>= 3;
        const bool planar = m == 0, isdc = m == 1, hor = m == 10, ver = m == 26;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int pl = (4 + (3 - j) * L[i] + (j + 1) * A[4] + (3 - i) * A[j] + (i + 1) * L[4]) >> 3;
                const int dv = (i == 0 && j == 0) ? (2 + 2 * dc + L[0] + A[0]) >> 2 : (i == 0) ? (2 + 3 * dc + A[j]) >> 2 : (j == 0) ? (2 + 3 * dc + L[i]) >> 2 : dc;
                int v = p[i * 4 + j];
                if (i == 0) v = hor ? hv_clip8(L[0] + ((A[j] - c) >> 1)) : v;      /* edge filters of the pure directions */
                if (j == 0) v = ver ? hv_clip8(A[0] + ((L[i] - c) >> 1)) : v;
                p[i * 4 + j] = planar ? pl : (isdc ? dv : v);
            }
    }
    HV_NXT(0);
    int r[16], t[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) r[i * 4 + j] = (int)S.org[ty + i][tx + j] - p[i * 4 + j];
    constexpr int D[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };       /* [ref :391-396] */
    /* forward: columns (shift 1), rows (shift 8) */
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) t[i * 4 + j] = (hv_mul24(D[i * 4], r[j]) + hv_mul24(D[i * 4 + 1], r[4 + j]) + hv_mul24(D[i * 4 + 2], r[8 + j]) + hv_mul24(D[i * 4 + 3], r[12 + j]) + 1) >> 1;   /* 24-bit multiplies throughout: |operands| < 2^16 x 2^7; a plain product becomes a shift-and-add chain or a quarter-rate 32-bit multiply */
    const int sh = 19 + q, thr = 9 << sh >> 2, qsh = 5 + q;
    const HvQlut lut = hv_qlut<2>(S);
    int lv[16], cf[16], sum = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int coef = (hv_mul24(t[i * 4], D[j * 4]) + hv_mul24(t[i * 4 + 1], D[j * 4 + 1]) + hv_mul24(t[i * 4 + 2], D[j * 4 + 2]) + hv_mul24(t[i * 4 + 3], D[j * 4 + 3]) + 128) >> 8;
            int d;
            cf[i * 4 + j] = coef;
            lv[i * 4 + j] = hv_quant_lut(lut, hv_abs(coef), d);       /* sixteen gathers in flight */
            sum += hv_min(d, thr);
        }
#pragma unroll
    for (int k = 0; k < 16; k++) lv[k] = cf[k] < 0 ? -lv[k] : lv[k];
    HV_NXT(1);
    /* levels in scan order: the three 4x4 orders are compile-time permutations of the raster positions, picked per lane by
       selects (no table reads, no divergent paths); two 16-byte LDS stores instead of sixteen scattered ones */
    const int kind = hv_scan_kind(4, m);
    const int zero = sum < thr;
#pragma unroll
    for (int k = 0; k < 16; k++) { lv[k] = zero ? 0 : lv[k]; r[k] = hv_clip16((int)((unsigned)lv[k] << qsh)); }
    constexpr int DG[16] = { 0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15 };   /* up-right diagonal  [ref :1130-1132] */
    int o[16];
    unsigned nzs = 0;
#pragma unroll
    for (int sc = 0; sc < 16; sc++) {
        o[sc] = kind == 0 ? lv[DG[sc]] : (kind == 1 ? lv[sc] : lv[(sc & 3) * 4 + (sc >> 2)]);
        nzs |= (unsigned)(o[sc] != 0) << sc;
    }
    const int last = 31 - hv_clz(nzs);                           /* -1 when all levels are zero */
    {
        HvChunk c0, c1;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c0[i] = ((unsigned)o[2 * i] & 0xffffu) | ((unsigned)o[2 * i + 1] << 16);
            c1[i] = ((unsigned)o[8 + 2 * i] & 0xffffu) | ((unsigned)o[9 + 2 * i] << 16);
        }
        ((HV_LDSP HvChunk *)lev_dst)[0] = c0; ((HV_LDSP HvChunk *)lev_dst)[1] = c1;
    }
    HV_NXT(2);
    /* inverse: columns with C^T (shift 7), rows (shift 12) */
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) t[i * 4 + j] = hv_clip16((hv_mul24(D[i], r[j]) + hv_mul24(D[4 + i], r[4 + j]) + hv_mul24(D[8 + i], r[8 + j]) + hv_mul24(D[12 + i], r[12 + j]) + 64) >> 7);
    int sse = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int res = hv_clip16((hv_mul24(t[i * 4], D[j]) + hv_mul24(t[i * 4 + 1], D[4 + j]) + hv_mul24(t[i * 4 + 2], D[8 + j]) + hv_mul24(t[i * 4 + 3], D[12 + j]) + 2048) >> 12);
            const int rc = hv_clip8(res + p[i * 4 + j]);
            const int dd = (int)S.org[ty + i][tx + j] - rc;
            sse += hv_mul24(dd, dd);
            p[i * 4 + j] = rc;
        }
    {
        HvChunk rc4;
#pragma unroll
        for (int i = 0; i < 4; i++) rc4[i] = (unsigned)p[i * 4] | ((unsigned)p[i * 4 + 1] << 8) | ((unsigned)p[i * 4 + 2] << 16) | ((unsigned)p[i * 4 + 3] << 24);
        *(HV_LDSP HvChunk *)rec_dst = rc4;
    }
    HV_NXT(3);
    last_out = last;
    return sse;
}

/* The same 4x4 block trial with FOUR lanes per mode (lane = mode * 4 + row): a helper wavefront of the chain workgroup does 16
   modes at a quarter of the instructions per lane; three of them stand in for hv_tu4_lane when the workgroup's other wavefronts
   are free (cluster of four).  Rows and columns change hands through a 32-byte tile per mode in the wavefront's scratch.
   Results go where hv_tu4_lane puts them: S.nxn_lev / nxn_rec / nxn_sse / nxn_last / nxn_sig of the mode.  [ref :1508-1518] */
static HD void hv_tu4_quad(HvLds &S, int q, int ty, int tx, int m, int ok) {
    const int lane = hv_tid() & 63, qd = lane >> 2, i = lane & 3;
    HvWaveScr &W = S.ws[hv_tid() >> 6];
    HV_LDSP short *tile = (HV_LDSP short *)W.scr + qd * 16;       /* [4][4] 16-bit, exchanged between the mode's four lanes */
    HV_LDSP u8 *line = (HV_LDSP u8 *)W.pred + qd * 16 + 4;        /* the mode's reference line r(-4..9) */
    const u8 *b = S.nxn_bord, *l = b + 1, *a = b + 9;
    const int c = b[0];
    m = ok ? m : 1;
    const bool horiz = m < 18;
    const int ang = S.ang[m], inv = S.invang[m];
    {   /* reference line: lane i lays out r(4 i - 4 .. 4 i - 1)  [ref :353-364] */
        const u8 *mainr = horiz ? l : a, *sider = horiz ? a : l;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int k = 4 * i - 4 + e;
            const int vm = mainr[hv_min(hv_max(k, 1), 8) - 1], vs = sider[hv_min(hv_max(((128 - inv * k) >> 8) - 1, 0), 7)];
            line[k] = (u8)(k > 0 ? vm : (k == 0 ? c : vs));
        }
    }
    WAVE_SYNC();
    /* prediction of row i: all forms, the mode's own selected  [ref :262-381] */
    int p[4];
    {
        const int Li = l[i], L4 = l[4], A4 = a[4], L0 = l[0], A0 = a[0];
        int dc = 4;
#pragma unroll
        for (int k = 0; k < 4; k++) dc += l[k] + a[k];
        dc >>= 3;
        const bool planar = m == 0, isdc = m == 1, hor = m == 10, ver = m == 26;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int Ax = a[x];
            const int st = horiz ? x : i, al = horiz ? i : x;
            const int off = ang * (st + 1), oi = off >> 5, of = off & 31, k1 = oi + al + 1;
            int v = (hv_mul24(32 - of, line[k1]) + hv_mul24(of, line[k1 + 1]) + 16) >> 5;
            const int pl = (4 + (3 - x) * Li + (x + 1) * A4 + (3 - i) * Ax + (i + 1) * L4) >> 3;
            const int dv = (i == 0 && x == 0) ? (2 + 2 * dc + L0 + A0) >> 2 : (i == 0) ? (2 + 3 * dc + Ax) >> 2 : (x == 0) ? (2 + 3 * dc + Li) >> 2 : dc;
            v = (i == 0 && hor) ? hv_clip8(L0 + ((Ax - c) >> 1)) : v;      /* edge filters of the pure directions */
            v = (x == 0 && ver) ? hv_clip8(A0 + ((Li - c) >> 1)) : v;
            p[x] = planar ? pl : (isdc ? dv : v);
        }
    }
    constexpr int D[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };       /* [ref :391-396] */
    /* residual row i -> tile, read back as column i */
    int r[4], t[4];
#pragma unroll
    for (int x = 0; x < 4; x++) tile[x * 4 + i] = (short)((int)S.org[ty + i][tx + x] - p[x]);      /* stored transposed: tile[column][row] */
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = tile[i * 4 + k];                                            /* r[k] = residual(row k, column i) */
    WAVE_SYNC();
    /* forward, columns (shift 1): t[k] = coefficient (k, column i) */
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = (hv_mul24(D[k * 4], r[0]) + hv_mul24(D[k * 4 + 1], r[1]) + hv_mul24(D[k * 4 + 2], r[2]) + hv_mul24(D[k * 4 + 3], r[3]) + 1) >> 1;
#pragma unroll
    for (int k = 0; k < 4; k++) tile[k * 4 + i] = (short)t[k];                                      /* tile[row][column] */
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = tile[i * 4 + k];                                            /* row i of the intermediate */
    WAVE_SYNC();
    /* forward, rows (shift 8), quantiser  [ref :540-595] */
    const int sh = 19 + q, thr = 9 << sh >> 2, qsh = 5 + q;
    const HvQlut lut = hv_qlut<2>(S);
    int cf[4], lv[4], sum = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        cf[x] = (hv_mul24(t[0], D[x * 4]) + hv_mul24(t[1], D[x * 4 + 1]) + hv_mul24(t[2], D[x * 4 + 2]) + hv_mul24(t[3], D[x * 4 + 3]) + 128) >> 8;
        int d;
        lv[x] = hv_quant_lut(lut, hv_abs(cf[x]), d);
        sum += hv_min(d, thr);
    }
    sum += hv_shfl_xor(sum, 1);
    sum += hv_shfl_xor(sum, 2);
    const int zero = sum < thr;
    const int kind = hv_scan_kind(4, m);
    int last = -1;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        lv[x] = zero ? 0 : (cf[x] < 0 ? -lv[x] : lv[x]);
        const int si = kind == 0 ? S.d4idx[(i << 2) | x] : (kind == 1 ? i * 4 + x : x * 4 + i);
        if (ok) S.nxn_lev[m][si] = (short)lv[x];
        last = (lv[x] != 0 && si > last) ? si : last;
        tile[i * 4 + x] = (short)hv_clip16((int)((unsigned)lv[x] << qsh));                            /* dequantised, tile[row][column] */
    }
    last = hv_max(last, hv_shfl_xor(last, 1));
    last = hv_max(last, hv_shfl_xor(last, 2));
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = tile[k * 4 + i];                                            /* column i */
    WAVE_SYNC();
    /* inverse: columns with C^T (shift 7), rows (shift 12) */
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = hv_clip16((hv_mul24(D[k], r[0]) + hv_mul24(D[4 + k], r[1]) + hv_mul24(D[8 + k], r[2]) + hv_mul24(D[12 + k], r[3]) + 64) >> 7);
#pragma unroll
    for (int k = 0; k < 4; k++) tile[k * 4 + i] = (short)t[k];
    WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = tile[i * 4 + k];                                            /* row i */
    int sse = 0;
    unsigned rc4 = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int res = hv_clip16((hv_mul24(t[0], D[x]) + hv_mul24(t[1], D[4 + x]) + hv_mul24(t[2], D[8 + x]) + hv_mul24(t[3], D[12 + x]) + 2048) >> 12);
        const int rc = hv_clip8(res + p[x]);
        const int dd = (int)S.org[ty + i][tx + x] - rc;
        sse += hv_mul24(dd, dd);
        rc4 |= (unsigned)rc << (8 * x);
    }
    sse += hv_shfl_xor(sse, 1);
    sse += hv_shfl_xor(sse, 2);
    if (ok) {
        *(HV_LDSP unsigned *)&S.nxn_rec[m][i * 4] = rc4;
        if (i == 0) { S.nxn_sse[m] = sse; S.nxn_last[m] = last < 0 ? 0 : last; S.nxn_sig[m] = last >= 0; }
    }
    WAVE_SYNC();
}
/* --------------------------------------------------------------------------------------------- */
/* FAST NxN sub-search with ONE LANE PER SAMPLE.                                                                               */
/*                                                                                                                            */
/* A wavefront issues one instruction per ~4.5 cycles however many lanes are active, so a phase costs its per-lane INSTRUCTION  */
/* COUNT: with one lane per mode (hv_tu4_lane) a 4x4 block trial is ~1400 instructions, its binarisation ~800.  Here lane =     */
/* (mode slot, sample): a wavefront tries FOUR modes per pass, a mode's sixteen samples are one DPP row -- the transforms are   */
/* four multiply-adds with operands fetched from the row / the quad (no LDS, no barrier), sums are four rotations -- and the     */
/* binarisation has one lane per scan position: non-zero masks are ballots, "how many before me" a popcount.  ~450 instructions  */
/* per pass instead of ~2200 per mode.                                                                                         */
/* The bins are written as RESOLVED tokens: in CABACputBin  [ref :914-933] the context byte evolves from `bin != mps` alone, and  */
/* the trials of the sub-search all start from the FRESH context set  [ref :1505-1506], so the state a bin meets is a small       */
/* table look-up on the bins of its context before it (S.ftab).  Token: bits 16.. = context byte << 1 | bin = the index of the    */
/* engine's table entry (S.ctab2), bits 0..15 = the plain token for whoever codes the same bins from another state (candidate     */
/* 70).  The engine (hv_rengine) then has no memory access on its dependency chain: perm -> sub -> select -> ffbh -> shift.       */
/* --------------------------------------------------------------------------------------------- */
#define HV_RTOK(cv, bin, plain) ((unsigned)(plain) | ((unsigned)(((cv) << 1) | (bin)) << 16))
#define HV_RTOK_NOP (((unsigned)(HV_PADCV << 1) << 16) | HV_TOK_NOPF)
/* FAST rows of S.nxn_tok (288 bytes): resolved tokens as 32-bit words from word 4 (<= 8 + 15 + 9 and 8 no-ops), 16-byte aligned for
   the engine's chunk loads; from byte 224 the block's non-zero magnitudes in coding order, 16-bit */
#define HV_NXR_TOK  4
#define HV_NXR_COMP 112
struct HvPx4 { int dcol[4], drow[4], tcol[4], trow[4]; };   /* this lane's transform coefficients (lane & 15 = y << 2 | x) */
/* cluster of four: the chain workgroup has no 32x32 candidates of its own, so their slot lists (S.slist2) are free: behind the copy of
   the NxN winners (hv_nxv_copy, 2.3 KB) sit the first 2048 entries of the picture's 4x4 level table (hv_fill_lut4; what a FAST
   picture's EXACT CTU overwrites is laid out again when the workgroup goes back to FAST) */
static HD const HV_LDSP u16 *hv_lut4_lds(const HvLds &S, int on) { return on ? (const HV_LDSP u16 *)((const HV_LDSP u8 *)&S.slist2[0][0] + 2560) : (const HV_LDSP u16 *)nullptr; }
static HD void hv_fill_lut4(HvLds &S, int tid) {
    HV_LDSP u16 *dst = (HV_LDSP u16 *)((HV_LDSP u8 *)&S.slist2[0][0] + 2560);
    const HvQlut lut = hv_qlut<2>(S);
    for (int i = tid; i < 2048; i += HV_NTHREADS) dst[i] = lut[i];
}
static HD void hv_px4_consts(const HvLds &S, HvPx4 &K, int lane) {
    const int y = (lane >> 2) & 3, x = lane & 3;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        K.dcol[s] = S.fx.d4[y * 4 + ((y + s) & 3)];           /* forward, columns: row y of the result sums D[y][k] * r[k][x], k = (y + s) & 3 */
        K.drow[s] = S.fx.d4[x * 4 + s];                     /* forward, rows   : coefficient (y, x) sums t[y][k] * D[x][k] */
        K.tcol[s] = S.fx.d4[((y + s) & 3) * 4 + y];           /* inverse, columns: D[k][y] * r[k][x] */
        K.trow[s] = S.fx.d4[s * 4 + x];                     /* inverse, rows   : t[y][k] * D[k][x] */
    }
}
static HD int hv_row_sum(int v) { v += hv_row_get<8>(v); v += hv_row_get<4>(v); v += hv_row_get<2>(v); v += hv_row_get<1>(v); return v; }   /* over the 16 lanes of a row, in every lane */
/* block trial of mode m for the sample of this lane: prediction from the block's reference samples (S.nxn_bord: corner, left[8],
   above[8]; 4x4 blocks never use the smoothed ones  [ref :274-275]), residual, DST, quantiser, dequantiser, inverse, reconstruction;
   levels go to S.nxn_lev[m] in scan order, samples to S.nxn_rec[m], the distortion to S.nxn_sse[m]   [ref :1508-1518] */
struct HvPx4Out { int lv, rc, sse, si; };                    /* this sample's level, its scan position, its reconstruction; the block's distortion */
static HD HvPx4Out hv_tu4_px_core(HvLds &S, const HvPx4 &K, const HV_LDSP u16 *lut4, const u8 *b, int q, int ty, int tx, int m, int lane) {
    const int p = lane & 15, y = p >> 2, x = p & 3;
    const u8 *l = b + 1, *a = b + 9;
    const int c = b[0];
    const bool horiz = m < 18, planar = m == 0, isdc = m == 1, hor = m == 10, ver = m == 26;
    const int ang = S.ang[m], inv = S.invang[m];
    int pred;
    {
        /* angular: two neighbours on the mode's reference line r(k) -- r(k>0) the main side, r(0) the corner, r(k<0) the side samples
           projected through the inverse angle  [ref :342-380]; pure horizontal / vertical are angle 0 plus their edge filter */
        const int step = horiz ? x : y, along = horiz ? y : x;
        const int off = hv_mul24(ang, step + 1), oi = off >> 5, of = off & 31, k1 = oi + along + 1, k2 = k1 + 1;
        const int mb = horiz ? 1 : 9, sb = horiz ? 9 : 1, ik = hv_mul24(inv, k1);
        const int i1 = k1 > 0 ? mb + hv_min(k1, 8) - 1 : (k1 == 0 ? 0 : sb + hv_min(hv_max(((128 - ik) >> 8) - 1, 0), 7));
        const int i2 = k2 > 0 ? mb + hv_min(k2, 8) - 1 : (k2 == 0 ? 0 : sb + hv_min(hv_max(((128 - ik - inv) >> 8) - 1, 0), 7));
        int v = (hv_mul24(32 - of, b[i1]) + hv_mul24(of, b[i2]) + 16) >> 5;
        const int ly = l[y], ax = a[x], l0 = l[0], a0 = a[0];
        int dc = 4;
#pragma unroll
        for (int k = 0; k < 4; k++) dc += l[k] + a[k];
        dc >>= 3;
        const int pl = (4 + hv_mul24(3 - x, ly) + hv_mul24(x + 1, a[4]) + hv_mul24(3 - y, ax) + hv_mul24(y + 1, l[4])) >> 3;
        const int dv = (y == 0 && x == 0) ? (2 + 2 * dc + l0 + a0) >> 2 : (y == 0) ? (2 + 3 * dc + ax) >> 2 : (x == 0) ? (2 + 3 * dc + ly) >> 2 : dc;
        v = (y == 0 && hor) ? hv_clip8(l0 + ((ax - c) >> 1)) : v;
        v = (x == 0 && ver) ? hv_clip8(a0 + ((ly - c) >> 1)) : v;
        pred = planar ? pl : (isdc ? dv : v);
    }
    const int org = S.org[ty + y][tx + x];
    const int r = org - pred;
    /* forward: columns (shift 1), rows (shift 8)  [ref :497-516] */
    const int t = (hv_mul24(K.dcol[0], r) + hv_mul24(K.dcol[1], hv_row_get<4>(r)) + hv_mul24(K.dcol[2], hv_row_get<8>(r)) + hv_mul24(K.dcol[3], hv_row_get<12>(r)) + 1) >> 1;
    const int cf = (hv_mul24(K.drow[0], hv_quad_get<0>(t)) + hv_mul24(K.drow[1], hv_quad_get<1>(t)) + hv_mul24(K.drow[2], hv_quad_get<2>(t)) + hv_mul24(K.drow[3], hv_quad_get<3>(t)) + 128) >> 8;
    /* quantiser  [ref :540-595] */
    const int sh = 19 + q, thr = 9 << sh >> 2, qsh = 5 + q;
    const int av = hv_abs(cf), d = av << 14;
    int mag;
    if (lut4) {                                                  /* the first 2048 entries of the 4x4 level table are in LDS (cluster of four) */
        mag = lut4[hv_min(av, 2047)];
        if (hv_ballot(av > 2047)) { int d2; const int g = hv_quant_lut(hv_qlut<2>(S), av, d2); mag = av > 2047 ? g : mag; }
    } else { int d2; mag = hv_quant_lut(hv_qlut<2>(S), av, d2); }
    const int zero = hv_row_sum(hv_min(d, thr)) < thr;
    const int lv = zero ? 0 : (cf < 0 ? -mag : mag);
    const int dq = hv_clip16((int)((unsigned)lv << qsh));
    /* inverse: columns with C^T (shift 7), rows (shift 12) */
    const int t2 = hv_clip16((hv_mul24(K.tcol[0], dq) + hv_mul24(K.tcol[1], hv_row_get<4>(dq)) + hv_mul24(K.tcol[2], hv_row_get<8>(dq)) + hv_mul24(K.tcol[3], hv_row_get<12>(dq)) + 64) >> 7);
    const int res = hv_clip16((hv_mul24(K.trow[0], hv_quad_get<0>(t2)) + hv_mul24(K.trow[1], hv_quad_get<1>(t2)) + hv_mul24(K.trow[2], hv_quad_get<2>(t2)) + hv_mul24(K.trow[3], hv_quad_get<3>(t2)) + 2048) >> 12);
    const int rc = hv_clip8(res + pred), dd = org - rc;
    HvPx4Out o;
    o.lv = lv; o.rc = rc; o.sse = hv_row_sum(hv_mul24(dd, dd)); o.si = S.fx.scan4[hv_scan_kind(4, m)][p];
    return o;
}
static HD void hv_tu4_px(HvLds &S, const HvPx4 &K, const HV_LDSP u16 *lut4, int q, int ty, int tx, int m, int ok, int lane) {
    m = ok ? m : 1;
    const HvPx4Out o = hv_tu4_px_core(S, K, lut4, S.nxn_bord, q, ty, tx, m, lane);
    if (ok) {
        S.nxn_lev[m][o.si] = (short)o.lv;
        S.nxn_rec[m][lane & 15] = (u8)o.rc;
        if ((lane & 15) == 0) S.nxn_sse[m] = o.sse;
    }
}
/* bins of coeff_abs_level_remaining  [ref :1154-1169] for value v at Rice parameter k (the count hv_emit_remainder makes) */
static HD int hv_remainder_bins(int v, int k) {
    const int shortf = v < (3 << k);
    const int w2 = hv_max(v - (3 << k), 0) + (1 << k);
    const int len = 31 - hv_clz((unsigned)w2);
    return shortf ? (v >> k) + 1 + k : 4 + 2 * len - k;
}
/* binarisation of the same four blocks, lane = (mode slot, scan position); needs the levels of hv_tu4_px (a wave barrier in
   between).  Writes the mode's resolved tokens, its token / bypass-bin counts (S.nxn_magn) and last position / any-level flag.
   [ref :1046-1087 last position, :1194-1268 one group that is both first and last, :1154-1169] */
static HD void hv_tok4_px(HvLds &S, int m, int ok, int lane) {
    const int ii = lane & 15, s16 = lane & 48;
    const int v = ok ? S.nxn_lev[m][ii] : 0, av = hv_abs(v);
    m = ok ? m : 0;
    const unsigned nzm = (unsigned)(hv_ballot(v != 0) >> s16) & 0xffffu;
    const int nz = __builtin_popcount(nzm), sig = nzm != 0;
    const int last = sig ? 31 - hv_clz(nzm) : 0, start = last;
    const int ks = hv_scan_kind(4, m);
    /* tokens that do not exist (and everything of a lane without a mode) go to word HV_NXR_DUMMY of the row: no divergent stores */
    HV_LDSP unsigned *row = (HV_LDSP unsigned *)S.nxn_tok[m] + HV_NXR_TOK;
    HV_LDSP u16 *comp = (HV_LDSP u16 *)S.nxn_tok[m] + HV_NXR_COMP;
    #define HV_NXR_DUMMY 64
    /* ---- last_sig_coeff_{x,y}_prefix: a coordinate c < 3 is c ones and a zero, 3 is three ones; one context per bin.
       Lanes 0..3 of the mode write the x tokens, 4..7 the y tokens ---- */
    int nlast;
    {
        const int pos = S.fx.unscan4[ks][last], iy = pos >> 2, ix = pos & 3;
        const int cx_ = ks == 2 ? iy : ix, cy_ = ks == 2 ? ix : iy;     /* the vertical scan codes the coordinates swapped */
        const int nx = cx_ < 3 ? cx_ + 1 : 3, ny = cy_ < 3 ? cy_ + 1 : 3;
        nlast = nx + ny;
        const int rr = ii & 3, isy = (ii >> 2) & 1, coord = isy ? cy_ : cx_, cnt = isy ? ny : nx, ctx = (isy ? CX_LASTY : CX_LASTX) + rr;
        const int bin = rr < coord;
        row[(ok && ii < 8 && rr < cnt) ? (isy ? nx : 0) + rr : HV_NXR_DUMMY] = HV_RTOK(S.ftab[ctx], bin, HV_CT(ctx, bin));
    }
    /* ---- sig_coeff_flag of the positions below the last one: the flag's context has seen the flags of its class's earlier
       positions, which are bits of nzm ---- */
    {
        const int info = S.fx.sig4[ks][ii];
        const int cls = info & 15, mm = (info >> 4) & 3, e1 = (info >> 6) & 15, e2 = (info >> 10) & 15;
        const int ok1 = mm >= 1 && e1 < start, ok2 = mm >= 2 && e2 < start;       /* e1 > e2: ok1 implies ok2 whenever there is a second one */
        const int b1 = (nzm >> e1) & 1, b2 = (nzm >> e2) & 1;
        const int k = ok1 + ok2, pre = ok1 ? (b1 | (ok2 ? b2 << 1 : 0)) : (ok2 ? b2 : 0);
        const int bin = (nzm >> ii) & 1;
        row[(ok && ii < start) ? nlast + start - 1 - ii : HV_NXR_DUMMY] = HV_RTOK(S.ftab[HV_FT_SIG + cls * 8 + ((1 << k) | pre)], bin, HV_CT(CX_SIG + cls, bin));
    }
    /* ---- the non-zero magnitudes in coding order (highest position first): item t of the mode goes to lane t ---- */
    comp[(ok && v != 0) ? __builtin_popcount(nzm >> (ii + 1)) : 16 + ii] = (u16)av;      /* (16 + ii: the row's spare words behind) */
    WAVE_SYNC();
    const int t = ii, at = (ok && t < nz) ? comp[t] : 0;
    const unsigned bigc = (unsigned)(hv_ballot(at > 1) >> s16) & 0xffffu, g2c = (unsigned)(hv_ballot(at > 2) >> s16) & 0xffffu;
    const int n1 = hv_min(nz, 8);
    const int fb = __builtin_ctz((bigc & 0xffu) | 0x100u);       /* the first level above 1 among the first eight; 8: none */
    /* greater1 flags of the first eight: contexts 1, 2, 3, 3, ... up to the first level above 1, 0 after it; so context 3 has seen
       zeros only and context 0 the flags since.  greater2 of that first one  [ref :1217-1236] */
    {
        const int c1 = t <= fb ? hv_min(t + 1, 3) : 0, k0 = hv_max(t - fb - 1, 0);
        const int p0 = (int)(bigc >> ((fb + 1) & 15)) & ((1 << k0) - 1);
        const int at_ = c1 == 0 ? HV_FT_ONE0 + ((1 << k0) | p0) : (c1 == 3 ? HV_FT_ONE3 + (t - 2) : CX_ONE + c1);
        const int bin = at > 1;
        row[(ok && t < n1) ? nlast + start + t : HV_NXR_DUMMY] = HV_RTOK(S.ftab[at_], bin, HV_CT(CX_ONE + c1, bin));
        row[(ok && t == fb) ? nlast + start + n1 : HV_NXR_DUMMY] = HV_RTOK(S.ftab[CX_ABS], at > 2, HV_CT(CX_ABS, at > 2));
    }
    const int ntok = nlast + start + n1 + (fb < 8);
    row[(ok && ii < 8) ? ntok + ii : HV_NXR_DUMMY] = HV_RTOK_NOP;
    /* ---- bypass bins: a sign per level, and the remaining magnitudes with the adaptive Rice parameter  [ref :1238-1268].  The
       parameter only moves up, by one, behind an item above 3 << parameter: from 0 behind the first item above 3, from 1 behind the
       first item after that above 6, ... -- four ballots and four find-first-set; the bins are then counted one item per lane ---- */
    const int more = __builtin_popcount(bigc & 0xffu) > 1;
    const int esc = nz > 8 || more || (fb < 8 && ((g2c >> fb) & 1));
    const int base = t < 8 ? ((bigc & ((1u << t) - 1)) ? 2 : 3) : 1;
    const int rem = at - base, on = t < nz && esc && rem >= 0;
    int rice = 0;
    {
        const unsigned e0 = (unsigned)(hv_ballot(on && at > 3) >> s16) & 0xffffu, e1 = (unsigned)(hv_ballot(on && at > 6) >> s16) & 0xffffu;
        const unsigned e2 = (unsigned)(hv_ballot(on && at > 12) >> s16) & 0xffffu, e3 = (unsigned)(hv_ballot(on && at > 24) >> s16) & 0xffffu;
        const int p0 = __builtin_ctz(e0 | 0x10000u);                                    /* 16: never */
        const int p1 = __builtin_ctz((e1 & ~((2u << p0) - 1)) | 0x10000u);
        const int p2 = __builtin_ctz((e2 & ~((2u << p1) - 1)) | 0x10000u);
        const int p3 = __builtin_ctz((e3 & ~((2u << p2) - 1)) | 0x10000u);
        rice = (t > p0) + (t > p1) + (t > p2) + (t > p3);
    }
    const int byp = nz + hv_row_sum(on ? hv_remainder_bins(rem, rice) : 0);
    if (ok && ii == 0) { S.nxn_last[m] = last; S.nxn_sig[m] = sig; S.nxn_magn[m] = ntok | (byp << 8); }
    #undef HV_NXR_DUMMY
}
/* The engine for resolved tokens: n of them at sp (16-byte chunks of four), followed by at least 8 HV_RTOK_NOP.  Per bin the LPS
   range is a byte of the table entry picked by the range quarter (v_perm_b32), the new range `lps` or `range - lps` shifted left
   until bit 8 is set again  [ref :914-933]; the table entries of the next four bins are in flight meanwhile, the tokens of the
   four after those too. */
static HD void hv_rengine_step(HvFCod &c, unsigned lps4, unsigned nx) {
    const int lps = (int)hv_perm(lps4, 0u, (unsigned)c.range >> 6);
    const int pre = (nx >> 8) ? lps : c.range - lps;
    const int sh = hv_clz_nz((unsigned)pre) - 23;
    c.range = pre << sh;
    c.nbits -= sh;
}
template <typename ChunkPtr>
static HD void hv_rengine(const HvLds &S, HvFCod &c, ChunkPtr sp, int n) {
    if (n <= 0) return;
    unsigned el[4], en[4];
    {
        const HvChunk cur = sp[0];
#pragma unroll
        for (int k = 0; k < 4; k++) { el[k] = S.ctab2[cur[k] >> 16][0]; en[k] = S.ctab2[cur[k] >> 16][1]; }
    }
    HvChunk nxt = sp[1];
    for (int base = 0; base < n; base += 4) {
        const HvChunk tk = nxt;
        nxt = sp[(base >> 2) + 2];
        unsigned nl[4], nn[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { nl[k] = S.ctab2[tk[k] >> 16][0]; nn[k] = S.ctab2[tk[k] >> 16][1]; }
#pragma unroll
        for (int k = 0; k < 4; k++) hv_rengine_step(c, el[k], en[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) { el[k] = nl[k]; en[k] = nn[k]; }
    }
}
/* one pass: the block trials and the binarisation of the four modes 4 * slot .. 4 * slot + 3 (whole wavefront) */
static HD void hv_nxn_pass(HvLds &S, const HvPx4 &K, const HV_LDSP u16 *lut4, int q, int ty, int tx, int slot, int lane) {
    const int m = 4 * slot + (lane >> 4), ok = m < HEVCE_NMODES;
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
    hv_tu4_px(S, K, lut4, q, ty, tx, m, ok, lane);
    WAVE_SYNC();
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
#endif
    hv_tok4_px(S, m, ok, lane);
    WAVE_SYNC();
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    if (slot == 2 && lane == 0) { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); S.prof2[13] += t1_ - t0_; S.prof2[14] += t2_ - t1_; S.prof2[15] += 1; }   /* (one helper's first pass) */
#endif
}
/* The nine passes of a 4x4 block on the chain wavefront and its helpers.  A pass is two UNITS -- the block trials (~1.2 k cycles) and the
   binarisation (~1.8 k) of its four modes -- and a wavefront issues one instruction per ~4.5 cycles whatever it runs, so what counts is
   the longest wavefront.  Four helpers (clusters of four: wavefronts 1..4): two passes each, 6 k.  Six helpers (clusters of five and
   more, where the stream writer has left the workgroup: + wavefronts 6 and 7): one pass each, and the two passes left over are split
   into their units -- two helpers do those block trials FIRST, two others the binarisations behind their own pass: 4.8 k.
   A unit: slot | 0x20 = a whole pass, slot = its block trials (reported in S.nxq_tmask), slot | 0x10 = its binarisation (waits for them). */
static constexpr unsigned char hv_nxq_sched[2][7][3] = {
    { { 0x28, 0xff, 0xff }, { 0x20, 0x21, 0xff }, { 0x22, 0x23, 0xff }, { 0x24, 0x25, 0xff }, { 0x26, 0x27, 0xff }, { 0xff, 0xff, 0xff }, { 0xff, 0xff, 0xff } },
    { { 0x28, 0xff, 0xff }, { 0x06, 0x20, 0xff }, { 0x07, 0x21, 0xff }, { 0x22, 0x16, 0xff }, { 0x23, 0x17, 0xff }, { 0x24, 0xff, 0xff }, { 0x25, 0xff, 0xff } } };
static HD void hv_nxq_units(HvLds &S, const HvPx4 &K, const HV_LDSP u16 *lut4, int q, int ty, int tx, int six, int who, int lane) {   /* who: 0 = the chain wavefront, 1 + h = helper h */
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    const unsigned long long u0_ = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int u = hv_nxq_sched[1][who][k], u4 = hv_nxq_sched[0][who][k];
        const int unit = hv_uni(six) ? u : u4;
        if (unit == 0xff) break;
        const int slot = unit & 15, m = 4 * slot + (lane >> 4), ok = m < HEVCE_NMODES;
        if (unit & 0x20) hv_nxn_pass(S, K, lut4, q, ty, tx, slot, lane);
        else if (!(unit & 0x10)) {
            hv_tu4_px(S, K, lut4, q, ty, tx, m, ok, lane);
            HV_FENCE_RELEASE();
            WAVE_SYNC();
            if (lane == 0) hv_atomic_or(&S.nxq_tmask, 1 << slot);
        } else {
            while (!((hv_poll(&S.nxq_tmask) >> slot) & 1)) hv_spin_short();
            HV_FENCE_ACQUIRE();
            hv_tok4_px(S, m, ok, lane);
            WAVE_SYNC();
        }
    }
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
    if (lane == 0) S.prof2[24 + hv_uni(who)] += __builtin_amdgcn_s_memtime() - u0_;   /* this wavefront's units of the block, start to end */
#endif
}

/* a helper wavefront of the NxN chain (cluster of four and more): for each of the CU's four 4x4 blocks, when the chain has laid out the
   reference samples, its units of the block (hv_nxq_units) */
#define HV_NXN_HELPERS 4
static HDN void hv_nxn_helper(int q, HvCu cu, int h, int six) {
    HvLds &S = HV_S;
    q = hv_uni(q); cu.y = hv_uni(cu.y); cu.x = hv_uni(cu.x); h = hv_uni(h); six = hv_uni(six);
    const int lane = hv_tid() & 63;
    HvPx4 K;
    hv_px4_consts(S, K, lane);
    const HV_LDSP u16 *lut4 = hv_lut4_lds(S, 1);
    for (int s = 0; s < 4; s++) {
        while (hv_poll(&S.nxq_go) <= s) hv_spin_short();
        HV_FENCE_ACQUIRE();
        hv_nxq_units(S, K, lut4, q, cu.y + (s >> 1) * 4, cu.x + (s & 1) * 4, six, 1 + h, lane);
        HV_FENCE_RELEASE();
        if (lane == 0) hv_atomic_inc(&S.nxq_done);
    }
}

/* NxN job (8x8 CU only): per 4x4 sub-block choose the mode with a fresh coder and fresh contexts  [ref :1499-1528] */
template <bool FAST>
static HDN void hv_job_nxn(int wave, int lane, int q, HvCu cu, HV_GLOBAL u16 *c70, int helpers) {
    HvLds &S = HV_S;
    HvLds::Level &L = S.lv[0];
    q = hv_uni(q); cu.y = hv_uni(cu.y); cu.x = hv_uni(cu.x);
    int total = 0;
    HV_PROF2_T0();
    HvPx4 K;
    if constexpr (FAST) hv_px4_consts(S, K, lane);
    const HV_LDSP u16 *lut4 = hv_lut4_lds(S, FAST && hv_uni(helpers));
    if (lane == 0) { S.c70_len = 16; S.c70_byp = 0; }
    for (int s = 0; s < 4; s++) {
        int l, lb, a, ar;
        hv_sub_avail(s, cu.L, cu.LB, cu.A, cu.AR, l, lb, a, ar);
        const int oy = (s >> 1) * 4, ox = (s & 1) * 4;
        /* reference samples of this 4x4 (shared by all modes): corner, left + left-below, above + above-right  [ref :1502].  Lane 0 the
           corner, 1..8 the left column downwards, 9..16 the row above: one position each, read from the CU's own samples so far (inside
           the CU) or the picture's; what is not available repeats its neighbour or the corner  [ref :196-257 for a 4x4 block] */
        {
            const HV_LDSP u8 *inner = (const HV_LDSP u8 *)&S.nxn_crec[0][0];
            const int ty = cu.y + oy, tx = cu.x + ox;
            const int side = lane >= 9, i = side ? lane - 9 : lane - 1;          /* (lane 0: i = -1) */
            const int on = side ? a : l, on2 = side ? ar : lb;                   /* this side's first four, its second four */
            int yy, xx, have;
            if (lane == 0) { yy = (l && a) ? ty - 1 : (l ? ty : ty - 1); xx = (l && a) ? tx - 1 : (l ? tx - 1 : tx); have = l || a; }
            else {
                const int k = (i < 4 || on2) ? i : 3;                             /* second four missing: the fourth one again */
                yy = side ? ty - 1 : ty + k; xx = side ? tx + k : tx - 1; have = on;
            }
            const int ry = yy - cu.y, rx = xx - cu.x, in = ry >= 0 && ry < 8 && rx >= 0 && rx < 8;
            const int vi = inner[hv_min(hv_max(ry, 0), 7) * 8 + hv_min(hv_max(rx, 0), 7)], vo = S.rec[yy + 1][xx + HV_RX];
            int v = have ? (in ? vi : vo) : 128;
            const int corner = hv_readlane(v, 0);
            v = have ? v : corner;
            if (lane < 17) S.nxn_bord[lane] = (u8)v;
        }
        WAVE_SYNC();
#if defined(HEVCE_PROF) && !defined(HEVCE_PROF_LIGHT) && !defined(HEVCE_EMU)
#define HV_NXQ(i) do { if (lane == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); S.prof2[16 + (i)] += n_ - nq_; nq_ = n_; } } while (0)
        unsigned long long nq_ = p2_;
        HV_NXQ(0);
#else
#define HV_NXQ(i) do { } while (0)
#endif
        if constexpr (FAST) {
            /* block trials and binarisation with one lane per sample, four modes per pass (hv_nxn_pass): nine passes -- all here, or
               eight on four helper wavefronts (hv_nxn_helper) and the last one here */
            if (hv_uni(helpers)) {
                const int nh = hv_uni(helpers);                  /* 4 or 6 helper wavefronts */
                if (lane == 0) hv_flag_clear(&S.nxq_tmask);      /* (the helpers are through with the block before: their done count) */
                HV_FENCE_RELEASE();
                WAVE_SYNC();
                if (lane == 0) hv_flag_set(&S.nxq_go, s + 1);
                hv_nxq_units(S, K, lut4, q, cu.y + oy, cu.x + ox, nh == 6, 0, lane);
                HV_NXQ(1);
                while (hv_poll(&S.nxq_done) < nh * (s + 1)) hv_spin_short();
                HV_FENCE_ACQUIRE();
                HV_NXQ(2);
            } else for (int slot = 0; slot < 9; slot++) hv_nxn_pass(S, K, lut4, q, cu.y + oy, cu.x + ox, slot, lane);
        } else if (lane < HEVCE_NMODES) {
            int last;
            const int sse = hv_tu4_lane(S, q, S.nxn_bord, S.nxn_line[lane], cu.y + oy, cu.x + ox, lane, (HV_LDSP short *)S.nxn_lev[lane], (HV_LDSP u8 *)S.nxn_rec[lane], last);
            S.nxn_sse[lane] = sse; S.nxn_last[lane] = last < 0 ? 0 : last; S.nxn_sig[lane] = last >= 0;
        }
        WAVE_SYNC();
        HV_PROF2(S, 0);
        /* rate of every candidate: lane = mode  [ref :1505-1519]; residual_coding is written even for an all-zero block */
        int cost = HV_I32MAX, idx = -1;
        if (lane < HEVCE_NMODES) {
            int bits, ntok, nbyp;
            if constexpr (FAST) {
                /* the mode's resolved tokens are there (hv_tok4_px): the engine has no context to look up or to write back */
                HV_PROF2(S, 1);
                if (s == 3 && lane == 0) hv_flag_clear(&S.nxn_busy);     /* (see below) */
                const int mg = S.nxn_magn[lane];
                ntok = mg & 0xff; nbyp = mg >> 8;
                HvFCod c = { 510, 23 };
                hv_rengine(S, c, (const HV_LDSP HvChunk *)((HV_LDSP unsigned *)S.nxn_tok[lane] + HV_NXR_TOK), ntok);
                bits = 23 - c.nbits + nbyp;
            } else {
                u8 *cx = &S.nxn_ctx[0][lane];                      /* the chain's own context columns (same layout as a level's) */
                for (int k = 0; k < 3; k++) { *HV_CX(CX_LASTX + k) = S.ctx_fresh[CX_LASTX + k]; *HV_CX(CX_LASTY + k) = S.ctx_fresh[CX_LASTY + k]; }
                for (int k = 0; k < 9; k++) *HV_CX(CX_SIG + k) = S.ctx_fresh[CX_SIG + k];
                for (int k = 0; k < 4; k++) *HV_CX(CX_ONE + k) = S.ctx_fresh[CX_ONE + k];
                *HV_CX(CX_ABS) = S.ctx_fresh[CX_ABS];
                *HV_CX(HV_CT_PAD) = HV_PADCV;                     /* pad context used by bypass / no-op tokens */
                HV_LDSP u16 *tk = (HV_LDSP u16 *)S.nxn_tok[lane];
                const int ks = hv_scan_kind(4, lane);
                /* the two slots are laid out so that their tokens are contiguous from tk[8]; one group: no set memory */
                HvTokW<HV_LDSP u16 *, FAST> wa = { tk, 0, 0 };
                const int na = hv_tokenize_tu<3>(S, wa, 4, 2, ks, 0, 0, 1, S.nxn_last[lane]);
                HvTokW<HV_LDSP u16 *, FAST> wb = { tk + na, 0, 0 };
                HvCg g;
                hv_cg_load(S, g, (const HV_LDSP short *)S.nxn_lev[lane], 4, 2, ks, 0, S.nxn_last[lane], (u64)(S.nxn_sig[lane] ? 1 : 0), 1);
                const int nb = hv_cg_emit(S, wb, g, 0);
#pragma unroll
                for (int i = 0; i < 8; i++) tk[HV_SLOT_TOK0 + na + nb + i] = (u16)HV_TOK_NOP;
                HV_PROF2(S, 1);
                /* last block, only its engine to go: tell the candidates riding in the other wavefronts to wind up, so that they are
                   at the workgroup barrier when this wavefront gets there */
                if (s == 3 && lane == 0) hv_flag_clear(&S.nxn_busy);
                HvCod c;
                hv_cod_fresh(c);
                hv_engine(S, c, cx, (const HV_LDSP HvChunk *)(tk + HV_SLOT_TOK0), na + nb, (HV_LDSP u16 *)S.nxn_sink[lane]);
                hv_cod_replay(c, (const HV_LDSP HvChunk *)S.nxn_sink[lane]);
                bits = hv_cod_bits(c);
                ntok = na + nb; nbyp = 0;
            }
            S.nxn_ntok[lane] = ntok | (nbyp << 8) | (bits << 20);
            cost = hv_rdcost(q, S.nxn_sse[lane], bits);
            idx = lane;
        }
        HV_PROF2(S, 2);
        /* minimum cost, the LATER mode wins ties  [ref :1521]: inside the rows of sixteen by rotations (no LDS), then over the four rows */
#define HV_ARGMIN_STEP(OC, OI) do { const int oc_ = (OC), oi_ = (OI); const bool tk_ = oc_ < cost || (oc_ == cost && oi_ > idx); cost = tk_ ? oc_ : cost; idx = tk_ ? oi_ : idx; } while (0)
        HV_ARGMIN_STEP(hv_row_get<8>(cost), hv_row_get<8>(idx));
        HV_ARGMIN_STEP(hv_row_get<4>(cost), hv_row_get<4>(idx));
        HV_ARGMIN_STEP(hv_row_get<2>(cost), hv_row_get<2>(idx));
        HV_ARGMIN_STEP(hv_row_get<1>(cost), hv_row_get<1>(idx));
        {
            const int c0 = hv_shfl(cost, 0), i0 = hv_shfl(idx, 0), c1 = hv_shfl(cost, 16), i1 = hv_shfl(idx, 16), c2 = hv_shfl(cost, 32), i2 = hv_shfl(idx, 32);
            cost = c0; idx = i0;
            HV_ARGMIN_STEP(c1, i1);
            HV_ARGMIN_STEP(c2, i2);                              /* (35 modes: the fourth row is empty) */
        }
#undef HV_ARGMIN_STEP
        const int w = idx;
        if (lane < 16) {
            S.nxn_crec[oy + (lane >> 2)][ox + (lane & 3)] = S.nxn_rec[w][lane];
            S.nxn_bestlev[s][lane] = S.nxn_lev[w][lane];
        }
        if (lane == 0) {
            S.nxn_modes[s] = w;
            L.tinfo[70][s].last = S.nxn_last[w]; L.tinfo[70][s].cbf = S.nxn_sig[w]; L.tinfo[70][s].cgsig = (u64)(S.nxn_sig[w] ? 1 : 0);
        }
        total += S.nxn_sse[w];
        /* candidate 70 = the four winners: its token stream in HBM is [CU header, 16 tokens reserved][per block: cbf_luma,
           then -- if the block has levels -- exactly the winner's tokens of above], contiguous, no slot padding */
        {
            const int pk = S.nxn_ntok[w];
            const int off = S.c70_len, cbf = S.nxn_sig[w], n = cbf ? (pk & 0xff) : 0;
            if (lane == 0) c70[off] = (u16)HV_CT(CX_YCBF, cbf);
            if constexpr (FAST) {
                /* the plain halves of the winner's resolved tokens (<= 32): candidate 70 meets them in another context state.  With a rider
                   (cluster of four) they go to its wavefront's scratch tile, two blocks alternating: no round trip through the L2 */
                const HV_LDSP unsigned *src = (const HV_LDSP unsigned *)S.nxn_tok[w] + HV_NXR_TOK;
                if (hv_uni(helpers)) {
                    while (hv_poll(&S.c70_rd) < s - 1) hv_spin_short();      /* the rider is through with the block before the last one (same buffer) */
                    HV_LDSP u16 *dst = (HV_LDSP u16 *)&S.ws[5] + (s & 1) * 64;
                    if (lane < 48) dst[lane] = lane < n ? (u16)src[hv_min(lane, 39)] : (u16)HV_TOK_NOPF;   /* whole chunks of no-ops behind, and one more for the engine's prefetch */
                } else if (lane < n) c70[off + 1 + lane] = (u16)src[lane];
            } else {
                const HV_LDSP u16 *src = (const HV_LDSP u16 *)S.nxn_tok[w] + HV_SLOT_TOK0;
                if (lane < n) c70[off + 1 + lane] = src[lane];
                if (lane + 64 < n) c70[off + 65 + lane] = src[lane + 64];
            }
            if constexpr (FAST) {
                /* kept for hv_verify_nxn: the winner's levels, what the tokeniser was given, the bit count it was chosen with */
                const int blk = (((cu.y >> 3) * 4 + (cu.x >> 3)) << 2) | s;
                if (lane < 16) S.nxv_lev[blk][lane] = S.nxn_lev[w][lane];
                if (lane == 0) { S.nxv_info[blk] = (u16)(S.nxn_last[w] | (hv_scan_kind(4, w) << 4) | (cbf << 6)); S.nxv_bits[blk] = (short)(pk >> 20); }
            }
            WAVE_SYNC();
            if (lane == 0) { S.c70_len = off + 1 + n; S.c70_byp += cbf ? ((pk >> 8) & 0xfff) : 0; }
            if constexpr (FAST) {
                /* candidate 70's rider (hv_c70_rider, another wavefront) may code this block now */
                HV_FENCE_RELEASE();
                if (lane == 0) { S.c70_seg[s] = off | (n << 16) | (cbf << 31); hv_flag_set(&S.c70_done, s + 1); }
            }
        }
        WAVE_SYNC();
        HV_PROF2(S, 3);
    }
    if (lane == 0) {
        L.tsse[70] = total;
        /* CU header of candidate 70 (part_mode NxN, the four modes)  [ref :1531-1538], padded to its 16 reserved tokens; the
           stream padded to whole chunks; one list entry covering it all (it runs on across the candidate's slot area) */
        const int my = 1 + (cu.y >> 2), mx = 1 + (cu.x >> 2);
        HvModes pm;
        for (int s4 = 0; s4 < 4; s4++) pm.mode[s4] = S.nxn_modes[s4];
        pm.left[0] = L.ml; pm.above[0] = L.ma;
        pm.left[1] = pm.mode[0];  pm.above[1] = S.map_pm[my - 1][mx + 1];
        pm.left[2] = S.map_pm[my + 1][mx - 1];  pm.above[2] = pm.mode[0];
        pm.left[3] = pm.mode[2];  pm.above[3] = pm.mode[1];
        HvTokW<HV_GLOBAL u16 *, FAST> wh = { c70 - HV_SLOT_TOK0, 0, 0 };
        const int nh = hv_tokenize_cu(wh, 8, 3, 2, L.split_inc, pm);
        for (int k = nh; k < 16; k++) c70[k] = (u16)(FAST ? HV_TOK_NOPF : HV_TOK_NOP);
        const int len = S.c70_len;
        for (int k = len; k < ((len + 7) & ~7); k++) c70[k] = (u16)(FAST ? HV_TOK_NOPF : HV_TOK_NOP);
        S.slist0[70][0] = (u16)(((len + 7) >> 3) << 8);          /* slot 0, that many chunks */
        L.tlen[70] = 1; L.tpos[70] = 0; L.tnlog[70] = FAST ? S.c70_byp + wh.byp : 0;   /* FAST: tnlog = the candidate's bypass bins */
        L.tregs[70] = S.ent[0];
        if constexpr (FAST) {                                    /* which of the four luma modes are most-probable-mode hits: the rider's lane  [ref :985-1018] */
            int pat = 0;
            for (int i = 0; i < 4; i++) { int p3[3]; hv_mpm(pm.left[i], pm.above[i], p3); pat |= (p3[0] == pm.mode[i] || p3[1] == pm.mode[i] || p3[2] == pm.mode[i]) << i; }
            S.c70_pat = pat;
        }
    }
    for (int r = lane; r < HEVCE_NCTXP; r += 64) L.tctx[r][70] = S.ent_ctx[0][r];
    if constexpr (FAST) { HV_FENCE_RELEASE(); WAVE_SYNC(); if (lane == 0) hv_flag_set(&S.c70_done, 5); }
    HV_PROF2(S, 4);
}
/* distortion of the CU as currently reconstructed (the "split into four CUs" alternative)  [ref :1409] */
static HD void hv_job_sse(HvLds &S, int lane, const HvCu &cu) {
    int sse = 0;
    for (int p = lane; p < cu.n * cu.n; p += 64) {
        const int yy = cu.y + p / cu.n, xx = cu.x + p % cu.n;
        const int d = (int)S.org[yy][xx] - S.rec[yy + 1][xx + HV_RX];
        sse += d * d;
    }
    for (int s = 1; s < 64; s <<= 1) sse += hv_shfl_xor(sse, s);
    if (lane == 0) S.split_sse = sse;
}

/* one context-coded bin on the committed coder (split_cu_flag = 1 of a parent)  [ref :1404] */
static HD void hv_commit_bin(HvLds &S, int ctx, int bin) {
    u8 tmp = S.com_ctx[ctx];
    if (S.exact) {
        HvCod c;
        hv_cod_load(c, S.com);
        hv_bin(S, c, &tmp, bin);
        hv_cod_store(S.com, c);
    } else {
        HvFCod c = { S.com.range, S.com.nbits };
        hv_fbin(S, c, &tmp, bin);
        S.com.range = c.range; S.com.nbits = c.nbits;
    }
    S.com_ctx[ctx] = tmp;
}
/* end_of_slice_segment_flag on the committed coder  [ref :1630] */
static HD void hv_commit_terminate(HvLds &S, int bin) {
    if (S.exact) {
        HvCod c;
        hv_cod_load(c, S.com);
        hv_terminate(c, bin);
        hv_cod_store(S.com, c);
    } else {
        HvFCod c = { S.com.range, S.com.nbits };
        hv_fterminate(c, bin);
        S.com.range = c.range; S.com.nbits = c.nbits;
    }
}

/* The winners of the CTU's 64 NxN sub-searches once more, each through an EXACT fresh coder  [ref :1505-1519]: lane = 4x4
   block.  The binarisation is the tokenisers' -- with a "writer" that sends every bin straight through the coder -- so the
   bit count can only differ from the FAST engine's by the emulation-prevention bytes; any difference poisons the CTU.
   One wavefront, at the end of a FAST CTU (the chain's context columns are free then). */
struct HvTokExec { const HvLds *S; HvCod c; u8 *cx; int n; enum { fast = 0 }; };
static HD void hv_emit(HvTokExec &w, int tok, int on) {
    if (on) {
        if (tok & HV_TOK_BYP) hv_bypass(w.c, tok & 0xff, (tok >> 8) & 15);
        else hv_bin(*w.S, w.c, w.cx + (tok & 0x1fff), (tok >> 14) & 1);
    }
}
static HD void hv_tok_checkpoint(HvTokExec &) { }
static HD int  hv_tok_finish(HvTokExec &) { return 0; }
struct HvNxv { short lev[64][16]; u16 info[64]; short bits[64]; };   /* a copy of the nxv_* arrays (deferred verification) */
static HD HV_LDSP HvNxv *hv_nxv_copy(HvLds &S) {                 /* lives in slist2: the 32x32 level's slot lists are unused in cluster mode */
    return (HV_LDSP HvNxv *)((HV_LDSP u8 *)&S.slist2[0][0] + ((16 - ((unsigned)(unsigned long long)&S.slist2[0][0] & 15)) & 15));
}
static HDN void hv_verify_nxn(int lane, int deferred) {
    HvLds &S = HV_S;
    deferred = hv_uni(deferred);
    /* deferred (cluster of four, a spare wavefront during the NEXT CTU's first round): the copy of the winners, and context
       columns of the 16x16 level, which is another workgroup's business in cluster mode */
    HV_LDSP HvNxv *V = hv_nxv_copy(S);
    u8 *cx = deferred ? &S.lv[1].tctx[0][lane] : &S.nxn_ctx[0][lane];
    for (int k = 0; k < 3; k++) { *HV_CX(CX_LASTX + k) = S.ctx_fresh[CX_LASTX + k]; *HV_CX(CX_LASTY + k) = S.ctx_fresh[CX_LASTY + k]; }
    for (int k = 0; k < 9; k++) *HV_CX(CX_SIG + k) = S.ctx_fresh[CX_SIG + k];
    for (int k = 0; k < 4; k++) *HV_CX(CX_ONE + k) = S.ctx_fresh[CX_ONE + k];
    *HV_CX(CX_ABS) = S.ctx_fresh[CX_ABS];
    const int info = deferred ? V->info[lane] : S.nxv_info[lane], last = info & 15, ks = (info >> 4) & 3, sig = (info >> 6) & 1;
    const int want = deferred ? V->bits[lane] : S.nxv_bits[lane];
    HvTokExec w;
    w.S = &S; w.cx = cx; w.n = 0;
    hv_cod_fresh(w.c);
    hv_tokenize_tu<3>(S, w, 4, 2, ks, 0, 0, 1, last);
    HvCg g;
    hv_cg_load(S, g, deferred ? (const HV_LDSP short *)V->lev[lane] : (const HV_LDSP short *)S.nxv_lev[lane], 4, 2, ks, 0, last, (u64)sig, 1);
    hv_cg_emit(S, w, g, 0);
    if (hv_cod_bits(w.c) != want) { if (deferred) S.poison_prev = 1; else S.poison = 1; }
}

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
static HD void hv_cl_peek8(HvLds &S, const HevceImageDesc &im, int lane) {
    if (lane) return;
    const int k = S.cl_ans8 + 1, n8 = hv_n8(im);                 /* the CU of this round (its decision counts it) */
    while (hv_poll(&S.c70_done) < 3 && !hv_peek(&S.cl_fail)) hv_spin_long();   /* (not before the chain's last 4x4 block: the answer is of no use earlier, and the polls share an L2 channel with every hand-off) */
    for (;;) {
        int all = 1, bw = 0, bc = 0;
        for (int g = 0; g < n8; g++) {                           /* every 8x8 workgroup's answer; the best of them, on equal cost the higher candidate */
            const unsigned long long t = hv_ld_flag64((const unsigned long long *)&im.cluster->res[2 + 2 * g + (k & 1)].tag);
            all &= (((unsigned)t >> 8) & 0xffffff) == (unsigned)(k & 0xffffff);
            const int w = (int)((unsigned)t & 0xff), c = (int)(t >> 32);
            if (g == 0 || hv_w8_better(bc, bw, c, w)) { bw = w; bc = c; }
        }
        if (all) {
            hv_acquire_fence(S.cl_light);
            S.pf.w = bw; S.pf.cost = bc; S.pf.k = k;
            return;
        }
        if (hv_poll(&S.c70_done) >= 5 || hv_peek(&S.cl_fail)) return;
        hv_spin_pause();
    }
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
        if (fast) hv_job_nxn<true>(wave, lane, q, c8, c70, hv_uni(cl) == 2 ? (hv_nxq_six(S, im, k16, m) ? 6 : 4) : 0);
        else hv_job_nxn<false>(wave, lane, q, c8, c70, 0);
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
        const int seg = hv_uni(S.c70_seg[s]), n = (seg >> 16) & 0x7fff, cbf = (seg >> 31) & 1;
        const HV_LDSP u16 *stage = (const HV_LDSP u16 *)&S.ws[hv_tid() >> 6] + (s & 1) * 64;   /* the block's tokens, padded: put there by the chain (hv_job_nxn) */
        if (act) {
            *HV_CX(HV_CT_PAD) = HV_PADCV;
            hv_fbin(S, c, HV_CX(CX_YCBF), cbf);                                                             /* [ref :1038-1043] */
            hv_fengine(S, c, cx, (const HV_LDSP HvChunk *)stage, n);   /* the same bins in all sixteen lanes, each with its own range and contexts */
        }
        WAVE_SYNC();
        if (lane == 0) hv_flag_set(&S.c70_rd, s + 1);
    }
    while (hv_poll(&S.c70_done) < 5) hv_spin_short();
    const int pat = hv_uni(S.c70_pat) & 15;
    if (lane == pat) {
        const int byp = L.tnlog[70];
        L.tregs[70].range = c.range; L.tregs[70].nbits = c.nbits;
        L.tcost[70] = hv_rdcost(q, L.tsse[70], S.ent[0].nbits - c.nbits + byp);
        L.tdead[70] = 0; L.tpos[70] = -1;
    }
    for (int r = lane; r < HEVCE_NCTXP; r += 64) L.tctx[r][70] = L.tctx[r][pat];
    WAVE_SYNC();
    hv_cl_report_nxn(S, im, lane);
}

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
    hv_publish_fence(S.cl_light);
    WAVE_SYNC();
    if (lane == 0) hv_st_flag(&r->seq, k);
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
        else { S.com = S.ent[0]; S.com.range = hv_ldc_i32(&r->range); S.com.nbits = hv_ldc_i32(&r->nbits); }
        int ok = S.cl_ok;                                        /* all of the cluster on this XCD: light hand-offs (hv_cl_receive) */
        for (int g = 0; g < n8; g++) ok |= hv_ldc_i32(&im.cluster->res[2 + 2 * g + (k & 1)].pad_[0]) == 1 ? 4 << g : 0;
        S.cl_ok = ok;
        const int need8 = 3 | (((1 << n8) - 1) << 2);
        if ((ok & need8) == need8 && !(im.flags & 16)) S.cl_light = 1;
    }
    if (tid >= 64 && tid < 64 + HEVCE_NCTXP) S.com_ctx[tid - 64] = win70 ? L.tctx[tid - 64][70] : (u8)hv_ldc_u8(&r->ctx[tid - 64]);
    const int kindw = w < 35 ? 0 : (w < 70 ? 1 : 2);
    if (tid >= 256 && tid < 320) {
        const int p = tid - 256, yy = p >> 3, xx = p & 7;
        S.rec[cu.y + yy + 1][cu.x + xx + HV_RX] = (kindw == 2) ? S.nxn_crec[yy][xx] : (u8)hv_ldc_u8((const u8 *)(grec + w * 1024 + p));
        int lv;
        if (kindw == 0) lv = hv_ldc_i16((const short *)(glev + w * 1024 + hv_scan_idx(S, 3, hv_scan_kind(8, w), yy, xx)));
        else {
            const int s4 = (yy >= 4) * 2 + (xx >= 4);
            const int md = kindw == 1 ? w - 35 : S.nxn_modes[s4];
            const int si = hv_scan_idx(S, 2, hv_scan_kind(4, md), yy & 3, xx & 3);
            lv = kindw == 1 ? hv_ldc_i16((const short *)(glev + w * 1024 + s4 * 16 + si)) : S.nxn_bestlev[s4][si];
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
    HvChunk cur = sp[0];
    for (int base = 0; base < n; base += 16) {
        const HvChunk nxt = sp[(base >> 4) + 1];
#pragma unroll
        for (int w4 = 0; w4 < 4; w4++) {
            unsigned el[4], en[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const unsigned ix = (cur[w4] >> (8 * k)) & 0xffu; el[k] = S.ctab2[ix][0]; en[k] = S.ctab2[ix][1]; }
#pragma unroll
            for (int k = 0; k < 4; k++) hv_rengine_step(c, el[k], en[k]);
        }
        cur = nxt;
    }
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
        /* reference samples: inside the CU the candidate's own reconstruction  [ref :196-257 for a 4x4 block] */
        {
            auto px = [&](int yy, int xx) -> int {
                const int ry = yy - cu.y, rx = xx - cu.x;
                return (ry >= 0 && ry < 8 && rx >= 0 && rx < 8) ? W.crec[m][ry * 8 + rx] : S.rec[yy + 1][xx + HV_RX];
            };
            const int corner = (l && a) ? px(ty - 1, tx - 1) : (l ? px(ty, tx - 1) : (a ? px(ty - 1, tx) : 128));
            const int i8 = ii & 7;
            int vl = l ? px(ty + hv_min(i8, 3), tx - 1) : corner, va = a ? px(ty - 1, tx + hv_min(i8, 3)) : corner;
            if (i8 >= 4) { vl = lb ? px(ty + i8, tx - 1) : vl; va = ar ? px(ty - 1, tx + i8) : va; }
            const int mb = ok ? m : 35;
            WAVE_SYNC();
            if (ii < 8) W.bord[mb][1 + i8] = (u8)vl; else W.bord[mb][9 + i8] = (u8)va;
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
}
/* work item: the tokens of the single-block candidates 4 slot .. 4 slot + 3 (their block trials: hv_job_single<8>)  [ref :1434-1435] */
static HD void hv_w8_single_slot(HvLds &S, HvW8 &W, int slot, HvGLev glev, int lane) {
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
    HvLds::Level &L = S.lv[0];
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
        } else if (id < 14) { hv_w8_split_slot(S, im, W, K, cu, id - 5, glev, grec, lane); HV_W8DBG(im, 2500 + id); HV_W8P(1); }
        else {
            const int slot = id - 14, j0 = (4 * slot) >> 3, j1 = hv_min(4 * slot + 3, 34) >> 3;
            HV_W8DBG(im, 3000 + id);
            while (!hv_poll(&W.tdone[j0]) || !hv_poll(&W.tdone[j1])) hv_spin_short();
            HV_FENCE_ACQUIRE();
            HV_W8DBG(im, 4000 + id);
            hv_w8_single_slot(S, W, slot, glev, lane);
            HV_W8DBG(im, 5000 + id);
            HV_W8P(2);
        }
    }
    HV_W8DBG(im, 6000);
    BLOCK_SYNC();
    HV_W8Q(6);                                                   /* the work items, to the last wavefront's end */
    HV_W8DBG(im, 7000);
#if defined(HEVCE_W8DBG) && !defined(HEVCE_EMU)
    if (im.trace && lane < 9 && wave * 9 + lane < 70) { ((volatile int *)im.trace)[128 + wave * 9 + lane] = W.ntok[wave * 9 + lane]; __threadfence_system(); }
#endif
#ifdef HEVCE_EMU
    if (tid == 0 && getenv("HEVCE_EMU_W8") && cu.y == 0 && cu.x == 0) { fprintf(stderr, "emu ntok/nbyp:"); for (int t_ = 0; t_ < 70; t_++) fprintf(stderr, " (%d,%d)", W.ntok[t_], W.nbyp[t_]); fprintf(stderr, "\n"); }
#endif
    /* engines: one lane per candidate */
    {
        const int t = wave * 9 + lane;
        if (lane < 9 && t < 70 && hv_w8_owner(t, n8) == g8) {
            HvFCod c = { S.ent[0].range, S.ent[0].nbits };
            hv_bengine(S, c, (const HV_LDSP HvChunk *)W.tok[t], W.ntok[t]);
            const int byp = W.nbyp[t];
            L.tregs[t] = S.ent[0]; L.tregs[t].range = c.range; L.tregs[t].nbits = c.nbits;
            L.tnlog[t] = byp; L.tdead[t] = 0; L.tpos[t] = -1;
            L.tcost[t] = hv_rdcost(q, L.tsse[t], S.ent[0].nbits - c.nbits + byp);
        }
    }
    HV_W8DBG(im, 7500);
    BLOCK_SYNC();
    HV_W8Q(7);                                                   /* the candidates' engines */
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
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int oc = hv_shfl_xor(bc, sft), oi = hv_shfl_xor(bi, sft);
            const bool take = oi != -2 && (bi == -2 || oc < bc || (oc == bc && oi > bi));
            bc = take ? oc : bc; bi = take ? oi : bi;
        }
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
    if (tid == 0) hv_st_flag(&C->alive[(serve & 4) ? (g8 == 0 ? 3 : 4 + g8) : (serve == 2 ? 2 : 1)], 1);   /* this workgroup is resident (hv_cl_alive); 8x8 workgroups: roles 3, 5, 6 */
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
    if (tid == 0) hv_st_flag(&im.cluster->alive[4], 1);          /* this workgroup is resident (hv_cl_alive) */
    if (tid == 0) { X->wregs[0] = S.wr; hv_publish_fence(0); hv_st_flag64((unsigned long long *)&X->wdone, 0ull); }
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
            hv_acquire_fence(0);
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
            hv_publish_fence(0);
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
static HD int hv_cl_alive(const HevceImageDesc &im) {
    HevceCluster *C = im.cluster;
    const unsigned long long t0 = hv_realtime(), to = (im.flags & 4) ? 100000ull : HV_CL_ALIVE_TIMEOUT;   /* test hook: 1 ms */
    for (;;) {
        int all = 1;
        for (int r = 1; r < C->size; r++) all &= hv_ld_flag(&C->alive[r]) != 0;
        if (all) return 1;
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
    if (tid == 0) { S.poison = 0; S.nxv_pend = 0; S.poison_prev = 0; S.wr_done = 0; S.cl_seq[0] = S.cl_seq[1] = S.cl_seq[2] = 0; S.cl_ans8 = 0; S.pf.k = 0; S.cl_fail = 0; S.cl_light = 0; S.cl_ok = 0; }
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
        if (tid == 0 && !hv_cl_alive(im)) S.cl_fail = 1;
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
                if (step < 20 && m == 0) {                       /* entering a 16x16 CU */
                    hv_snapshot(S, 1, tid);
                    hv_level_begin(S, c16);
                    BLOCK_SYNC();
                    if (cl) hv_cl_send(S, im, c16, ctu, 1);
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
                        if (m == 0) hv_cl_send(S, im, cu, ctu, 1, &c16);   /* opens the quad; the 8x8 workgroup finds its way through the other three CUs */
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
            hv_publish_fence(0);
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
