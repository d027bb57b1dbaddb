/* TEST INFRASTRUCTURE: runs hv_analyse_image() of the product's device header on the CPU fiber emulator. */
#define HEVCE_EMU 1
#include "emu_rt.h"
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include "../../hevc-image-encoder-lite_amd/csrc/hevce_device.h"
#include <vector>

int emu_cur = 0;
HvLds *emu_lds_ptr = nullptr;
int emu_stage[56][64];

namespace {
enum { NT = HV_NTHREADS, NWG = 7, STACK = 256 * 1024 };
enum State { RUN, WAIT_WAVE, WAIT_BLOCK, DONE };
struct Fiber { void *sp; State st; char *stack; };
Fiber fib[NWG * NT];
void *sched_sp;
int wave_wait[NWG * 8], block_wait[NWG], ndone;
HvLds *g_lds[NWG];
int g_nwg = 1;
const HevceImageDesc *g_im;

extern "C" void emu_switch(void **save_sp, void *new_sp);
asm(R"(
.text
.globl emu_switch
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

void yield_to_sched() { emu_switch(&fib[emu_cur].sp, sched_sp); }

void fiber_entry() {
    const int role = emu_cur / NT;                               /* cluster mode: workgroup 0 = chain, the others = parents */
    if (role == 4) hv_writer_worker(*g_lds[role], *g_im);
    else if (role == 3) hv_parent_worker(*g_lds[role], *g_im, 4, 0);
    else if (role == 5) hv_parent_worker(*g_lds[role], *g_im, 4, 1);
    else if (role == 6) hv_parent_worker(*g_lds[role], *g_im, 4, 2);
    else if (role) hv_parent_worker(*g_lds[role], *g_im, g_nwg == 2 ? 3 : role);
    else hv_analyse_image(*g_lds[0], *g_im);
    fib[emu_cur].st = DONE;
    ndone++;
    yield_to_sched();
    abort();
}
}  // namespace

void emu_wave_sync(void) {
    fib[emu_cur].st = WAIT_WAVE;
    wave_wait[emu_cur >> 6]++;
    yield_to_sched();
}
static void emu_segv(int) {                                  /* HEVCE_EMU_BT=1: where a fiber crashed */
    void *bt[48];
    const int n = backtrace(bt, 48);
    backtrace_symbols_fd(bt, n, 2);
    fprintf(stderr, "emu: fault in fiber %d (workgroup %d, thread %d)\n", emu_cur, emu_cur / NT, emu_cur % NT);
    _exit(139);
}
void emu_yield(void) { yield_to_sched(); }                  /* stays runnable */
void emu_block_sync(void) {
    fib[emu_cur].st = WAIT_BLOCK;
    block_wait[emu_cur / NT]++;
    yield_to_sched();
}

long long emu_tok_stat[4];
extern "C" __attribute__((visibility("default")))
int emu_analyse(const uint8_t *img, int src_h, int src_w, int q, uint8_t *rcon, uint8_t *records, int32_t *ctu_state, int32_t *dump,
                uint8_t *payload, int32_t *payload_len, int flags, int32_t *stats, int cluster) {
    HevceImageDesc im;
    memset(&im, 0, sizeof im);
    im.img = img; im.rcon = rcon; im.records = records; im.ctu_state = ctu_state;
    im.dump = dump;
    if (payload) { im.payload = payload; im.payload_len = payload_len; }     /* the kernel's stream writer on */
    im.flags = flags; im.stats = stats;
    static std::vector<uint16_t> qlut[5];                    /* the quantiser's level tables, as the host builds them */
    if (qlut[q].empty()) {
        qlut[q].resize(4 * HEVCE_QLUT_N);
        for (int t = 0; t < 4; t++) for (int av = 0; av < HEVCE_QLUT_N; av++) qlut[q][t * HEVCE_QLUT_N + av] = (uint16_t)hv_qlut_entry(q, t, av);
    }
    im.qlut = qlut[q].data();
    std::vector<uint8_t> wrtok(HEVCE_WRTOK_BYTES, 0xEF);
    im.wr_tok = wrtok.data();
    im.src_h = src_h; im.src_w = src_w; im.src_stride = src_w; im.qpd6 = q;
    im.H = ((src_h < HEVCE_MAXDIM ? src_h : HEVCE_MAXDIM) + 31) / 32 * 32;
    im.W = ((src_w < HEVCE_MAXDIM ? src_w : HEVCE_MAXDIM) + 31) / 32 * 32;
    std::vector<uint8_t> scratch(HEVCE_SCRATCH_BYTES, 0xCD);
    im.scratch = scratch.data();
    const int nwg = cluster > 1 ? cluster : 1, nfib = nwg * NT;
    g_nwg = nwg;
    static HevceCluster cl;
    memset(&cl, 0, sizeof cl);
    cl.size = cluster;
    im.cluster = cluster > 1 ? &cl : nullptr;
    for (int g = 0; g < nwg; g++) {
        g_lds[g] = (HvLds *)aligned_alloc(64, (sizeof(HvLds) + 63) / 64 * 64);
        memset(g_lds[g], 0xAB, sizeof(HvLds));                /* LDS is not zero-initialised on the GPU either */
    }
    g_im = &im;
    if (getenv("HEVCE_EMU_BT")) signal(SIGSEGV, emu_segv);
    memset(wave_wait, 0, sizeof wave_wait); memset(block_wait, 0, sizeof block_wait); ndone = 0;
    for (int i = 0; i < nfib; i++) {
        fib[i].stack = (char *)aligned_alloc(64, STACK);
        void **sp = (void **)(fib[i].stack + STACK - 64 - 8);   /* at fiber_entry's first instruction rsp must be 8 mod 16 */
        *--sp = (void *)fiber_entry;                      /* return address for emu_switch's ret */
        for (int k = 0; k < 6; k++) *--sp = 0;            /* rbp rbx r12..r15 */
        fib[i].sp = sp; fib[i].st = RUN;
    }
    int rc = 0;
    while (ndone < nfib) {
        bool progress = false;
        for (int i = 0; i < nfib; i++) {
            if (fib[i].st != RUN) continue;
            emu_cur = i;
            emu_lds_ptr = g_lds[i / NT];
            emu_switch(&sched_sp, fib[i].sp);
            progress = true;
        }
        for (int w = 0; w < nwg * 8; w++)
            if (wave_wait[w] == 64) {
                wave_wait[w] = 0;
                for (int l = 0; l < 64; l++) fib[w * 64 + l].st = RUN;
                progress = true;
            }
        for (int g = 0; g < nwg; g++)
            if (block_wait[g] == NT) {
                block_wait[g] = 0;
                for (int i = 0; i < NT; i++) fib[g * NT + i].st = RUN;
                progress = true;
            }
        if (!progress) {
            fprintf(stderr, "emu: deadlock (divergent synchronisation): block_wait=%d %d %d %d done=%d waves:", block_wait[0], block_wait[1], block_wait[2], block_wait[3] + 1000 * block_wait[4] + 1000000 * block_wait[5], ndone);
            for (int w = 0; w < nwg * 8; w++) fprintf(stderr, " %d", wave_wait[w]);
            fprintf(stderr, "\n");
            rc = -1;
            break;
        }
    }
    for (int i = 0; i < nfib; i++) free(fib[i].stack);
    for (int g = 0; g < nwg; g++) free(g_lds[g]);
    if (getenv("HEVCE_EMU_FOOTPRINT")) fprintf(stderr, "FAST trial slots: %lld tokens in %lld chunks of 8 in %lld slots (%.1f tokens, %.2f chunks per slot; fill %.0f %%)\n", emu_tok_stat[0], emu_tok_stat[1], emu_tok_stat[2],
        (double)emu_tok_stat[0] / emu_tok_stat[2], (double)emu_tok_stat[1] / emu_tok_stat[2], 12.5 * emu_tok_stat[0] / emu_tok_stat[1]);
    if (getenv("HEVCE_EMU_FOOTPRINT")) {                   /* 128-byte lines of the trial scratch the picture touched, per CU level and region */
        const size_t reg[5] = { 0, HEVCE_SCR_FTOK, HEVCE_SCR_XTOK, HEVCE_SCR_SINK, HEVCE_SCRATCH_LEVEL_BYTES };
        size_t all = 0;
        for (int l = 0; l < 3; l++) for (int r = 0; r < 4; r++) {
            size_t n = 0;
            for (size_t a = l * (size_t)HEVCE_SCRATCH_LEVEL_BYTES + reg[r]; a < l * (size_t)HEVCE_SCRATCH_LEVEL_BYTES + reg[r + 1]; a += 128) {
                bool hit = false;
                for (int b = 0; b < 128 && !hit; b++) hit = scratch[a + b] != 0xCD;
                n += hit;
            }
            fprintf(stderr, "footprint level %d %s: %zu lines = %zu KB\n", l, r == 0 ? "levels+samples" : r == 1 ? "FAST tokens" : r == 2 ? "EXACT tokens" : "sinks", n, n / 8);
            all += n;
        }
        size_t w = 0;
        for (size_t a = 0; a < HEVCE_WRTOK_BYTES; a += 128) { bool hit = false; for (int b = 0; b < 128 && !hit; b++) hit = wrtok[a + b] != 0xEF; w += hit; }
        fprintf(stderr, "footprint writer tokens: %zu lines; trial scratch in all: %zu KB\n", w, all / 8);
    }
    if (cluster > 1 && cl.error) rc = -2;
    return rc;
}
extern "C" __attribute__((visibility("default"))) int emu_lds_bytes(void) { return (int)sizeof(HvLds); }
