/*
 * hevce_common.h -- layouts shared by the HIP analysis kernel and the host stream writer.
 *
 * Decision record: what the device hands to the host per 32x32 CTU (DESIGN.md "Data layout in HBM").
 * Context layout : compact index space for the CABAC contexts a luma-only HEVC intra stream touches
 *                  (the reference's ContextSet, HEVCe.c:745-759, minus its chroma / unused slots).
 */
#pragma once
#include <stdint.h>

#define HEVCE_CTU        32
#define HEVCE_MAXDIM     8192
#define HEVCE_NMODES     35
#define HEVCE_REC_BYTES  2176

enum {
    CX_SPLIT_CU = 0,   /* 3 */
    CX_PART     = 3,
    CX_YPM      = 4,
    CX_UVPM     = 5,
    CX_SPLIT_TU = 6,   /* 3 : 32,16,8 */
    CX_YCBF     = 9,   /* 2 : index = (transform depth == 0) */
    CX_UVCBF    = 11,
    CX_LASTX    = 12,  /* 4 size classes x 5 */
    CX_LASTY    = 32,
    CX_SIGCG    = 52,  /* 2 */
    CX_SIG      = 54,  /* 27 (+1 pad) */
    CX_ONE      = 82,  /* 16 */
    CX_ABS      = 98,  /* 4 */
    HEVCE_NCTX  = 102,
    HEVCE_NCTXP = 104  /* padded */
};

/* bits of HevceRecord.info */
#define HEVCE_INFO_LOG2_MASK 3   /* log2(cu size) - 3 */
#define HEVCE_INFO_NXN       4
#define HEVCE_INFO_TUSPLIT   8

typedef struct HevceRecord {
    int16_t lev[32][32];   /* quantised levels, natural position inside the CTU */
    uint8_t mode[8][8];    /* luma intra mode per 4x4 block */
    uint8_t info[8][8];    /* CU size / partition / transform split per 4x4 block */
} HevceRecord;

/* arithmetic-coder registers (reference CABACcoder, HEVCe.c:797-806, without its byte buffer) */
typedef struct HevceCoderRegs {
    int32_t  range;
    uint32_t low;
    int32_t  nbits, nbytes, bufbyte, zeros;
    int32_t  nout;   /* bytes produced so far, emulation-prevention bytes included */
    int32_t  pad;
} HevceCoderRegs;

/* Cluster mode (two workgroups per picture, hevce_dev_workers.h hv_parent_worker): mailboxes in HBM between the workgroup that walks
   the picture's chain of 8x8 decisions and the ones that evaluate candidates for it.  index = (CU level + 2) % 3: 16x16, 32x32, 8x8. */
typedef struct HevceMsg {      /* analysis workgroup -> parent workgroup: "evaluate this CU from this entry state" */
    int32_t seq;               /* message number on this mailbox, written last (release); 0 = none yet */
    int32_t kind;              /* 1 = evaluate, 2 = exit */
    int32_t ctu, y, x, n;      /* the CU: CTU index, position inside the CTU, size */
    int32_t range, nbits;      /* coder at the CU's entry (range half) */
    int32_t split_inc, ml, ma; /* neighbour context  [ref HEVCe.c:1369-1373] */
    int32_t L, LB, A, AR;      /* neighbour availability */
    int32_t pad_;
    uint8_t ctx[104];          /* contexts at the CU's entry */
    /* 8x8 mailbox: the message opens a 16x16 quad.  The 8x8 workgroup walks its four CUs on its own (it learns each decision from
       its own answer and the chain's HevceNxn report), so it also needs the quad's availability and the modes beside its lower /
       right halves (ml, ma above are those beside CU 0)  [ref HEVCe.c:1369-1373] */
    int32_t ml2, ma2;          /* mode left of the quad's row 8, above its column 8 */
    int32_t qL, qLB, qA, qAR;  /* the 16x16 CU's neighbour availability */
    int32_t pad2_[2];
} HevceMsg;
typedef struct HevceRes {      /* parent workgroup -> analysis workgroup: the best unsplit candidate */
    int32_t seq;               /* = the message it answers, written last (release) */
    int32_t cost, winner, sse; /* its cost, index (0..69) and distortion */
    int32_t range, nbits;      /* coder after it (nbits with its bypass bins taken off) */
    int32_t pad_[2];
    uint8_t ctx[104];          /* contexts after it */
    uint64_t tag;              /* 8x8 answers: cost << 32 | (seq & 0xffffff) << 8 | winner, one 64-bit store after everything else: whoever polls it has the decision's inputs in the same load */
} HevceRes;
typedef struct HevceNxn {      /* analysis workgroup -> 8x8 workgroup, once per 8x8 CU: the CU's NxN candidate (number 70), which only the chain knows */
    int32_t seq;               /* number of the 8x8 CU since the launch, written last (release) */
    int32_t cost;
    int32_t range, nbits;      /* coder after it (nbits with its bypass bins taken off) */
    int32_t modes;             /* its four luma modes, 8 bits each, block 0 lowest */
    int32_t pad_[3];
    uint8_t ctx[104];          /* contexts after it */
    uint8_t rec[64];           /* its reconstruction, 8 x 8 */
} HevceNxn;
/* clusters of six / seven: TWO / THREE 8x8 workgroups share the 8x8 CUs' candidates 0..69 (hevce_dev_rounds.h hv_w8_owner); all read msg[2],
   workgroup g answers in res[2 + 2 g] / res[3 + 2 g], and each takes the others' answers into its own decision of the CU.
   mailboxes: msg / res 0: 16x16, 1: 32x32, msg 2: 8x8 (clusters of four: one message per 16x16 quad); the 8x8 answers alternate between
   res[2] and res[3] and the NxN reports between nxn[0] and nxn[1] (by the parity of the CU's number), so that neither side can
   overwrite what the other is still reading */
/* clusters of five: the stream writer is a workgroup of its own, several CTUs behind the analysis.  The chain workgroup publishes how many
   CTUs have final decision records (`go`); the writer encodes them and reports, per CTU, whether it put an emulation-prevention byte
   (`wbad`: the FAST decisions of that CTU are then not vouched for).  The chain goes back to that CTU -- its start state, up to seven CTUs
   back, is in `snap`; the exact coder registers there are the writer's, `wregs` -- analyses it with the EXACT engines, and starts a new
   EPOCH: the writer goes back to the same CTU (its own ring of states) and encodes it again.  Every status word carries its epoch. */
typedef struct HevceWrx {
    uint64_t go;               /* chain : epoch << 32 | CTUs whose records are final (0x7fffffff: exit) */
    uint64_t wdone;            /* writer: epoch << 32 | CTUs written */
    uint64_t wbad;             /* writer: epoch << 32 | 1 + the first CTU of the epoch it objects to (bit 30: its range differs from the analysis'); 0: none */
    int32_t  rst;              /* chain : the CTU the current epoch restarted at */
    int32_t  pad_[9];
    HevceCoderRegs wregs[8];   /* writer: its registers at the start of CTU c, [c & 7] */
    int32_t  crange[8];        /* chain : the analysis' range at the start of CTU c */
    int32_t  xflag[8];         /* chain : CTU c was analysed EXACT (its emulation-prevention bytes are part of its bit counts) */
    uint8_t  snap[8][192];     /* chain : what CTU c starts from (HvLds::Snap) */
} HevceWrx;
/* alive[r]: workgroup r of the cluster has started (it is RESIDENT: its hand-offs will be answered); the word is 0x100 | its XCC id, so the
   chain workgroup also learns whether the whole cluster shares its XCD (light hand-offs, hevce_dev_base.h hv_publish_fence).  The chain workgroup waits for all of
   them before the first CTU, a few milliseconds at most (HV_CL_ALIVE_TIMEOUT): a cluster whose workgroups the device has not placed
   side by side -- other kernels hold the compute units -- is given up at once and the host encodes the picture without clusters.
   error: 1-4 a hand-off timed out, 5 a wavefront ran with lanes missing, 6 a workgroup of the cluster was not resident */
typedef struct HevceCluster { HevceMsg msg[3]; HevceRes res[8]; HevceNxn nxn[2]; int32_t error; int32_t size; int32_t pad_[2]; int32_t alive[8]; HevceWrx wrx; } HevceCluster;

/* one image of a batch, as the kernel sees it (all pointers are device pointers) */
typedef struct HevceImageDesc {
    const uint8_t *img;       /* src_h x src_w luma, row stride src_stride (>= src_w: the picture may be a window of a larger one) */
    uint8_t       *rcon;      /* H x W reconstruction, stride W */
    uint8_t       *records;   /* (H/32)*(W/32) HevceRecord, raster CTU order */
    int32_t       *ctu_state; /* optional: 8 ints per CTU (coder registers after the CTU) */
    uint8_t       *scratch;   /* HEVCE_SCRATCH_BYTES of per-image trial scratch */
    const uint16_t *qlut;     /* 4 x 32768 levels: the quantiser's decision per coefficient magnitude, tables for 4x4 .. 32x32 at this qpd6 (hevce_quant.h) */
    int32_t src_h, src_w, H, W, qpd6, status;
    int64_t src_stride;
    int32_t       *dump;      /* diagnostics: per CU evaluation, 72 x (sse, cost) ints (NULL = off) */
    int32_t       *trace;     /* diagnostics: host-pinned, 16 ints per wavefront (only written by HEVCE_TRACE builds) */
    /* the kernel's stream writer: the slice data bytes.  payload == NULL: no writer; the analysis then runs
       with the EXACT engines throughout and the host replays the records (hevce_write_stream) */
    uint8_t       *payload;   /* out: the slice data bytes (what follows the VPS/SPS/PPS/slice header) */
    uint8_t       *wr_tok;    /* HEVCE_WRTOK_BYTES: the token slots of the CTU the writer is encoding */
    int32_t       *payload_len;
    int32_t       *stats;     /* optional, 2 ints: CTUs analysed twice (EXACT after FAST), consistency failures (must be 0) */
    HevceCluster  *cluster;   /* non-NULL: cluster mode, the launch has two workgroups per picture (2 i: chain, 2 i + 1: parents) */
    int32_t        flags;     /* tests: bit 0 EXACT engines throughout; bits 8-15 / 16-23: forced re-analysis periods (hevce_device.h) */
    int32_t        pad_;
} HevceImageDesc;

#define HEVCE_WRTOK_SLOT_WORDS 128
#define HEVCE_WRTOK_BYTES   (3 * 64 * HEVCE_WRTOK_SLOT_WORDS * 2)            /* CU header, block header, group slot per 4x4 position */
#define HEVCE_TRIALS        72                               /* 35 + 35 + 1 (+1 pad) trial slots */
#define HEVCE_TOKEN_SLOT_WORDS 128                           /* 16-bit words per bin-token slot, EXACT streams (context bins + bypass chunks) */
#define HEVCE_FTOKEN_SLOT_WORDS 32                           /* FAST streams: context bins only, at most 31 per slot (hevce_dev_nxn.h hv_tok_finish) */
#define HEVCE_TOKEN_SLOTS      70                            /* slots per trial: CU header + block headers + 4x4 groups */
#define HEVCE_SINK_WORDS       8192                          /* finished-byte records per trial between two replays (a 32x32 CU: < 6200 bytes in all) */
/* one CU level's trial scratch: levels + samples | FAST token slots | EXACT token slots | EXACT byte sinks.  The FAST analysis touches
   only the first two (dense: 64-byte slots), which keeps the picture's rewritten lines inside the L2 (write traffic) */
#define HEVCE_SCR_FTOK   (HEVCE_TRIALS * 1024 * 3)
#define HEVCE_SCR_XTOK   (HEVCE_SCR_FTOK + HEVCE_TRIALS * HEVCE_TOKEN_SLOTS * HEVCE_FTOKEN_SLOT_WORDS * 2)
#define HEVCE_SCR_SINK   (HEVCE_SCR_XTOK + HEVCE_TRIALS * HEVCE_TOKEN_SLOTS * HEVCE_TOKEN_SLOT_WORDS * 2)
#define HEVCE_SCRATCH_LEVEL_BYTES (HEVCE_SCR_SINK + HEVCE_TRIALS * HEVCE_SINK_WORDS * 2)
#define HEVCE_SCRATCH_BYTES (4 * HEVCE_SCRATCH_LEVEL_BYTES)   /* one set per CU level: 8x8, 16x16, 32x32 candidates coexist; the fourth: the 8x8 workgroup of a
                                                                cluster alternates between sets 0 and 3 (the chain workgroup still reads the winner of CU k
                                                                while the candidates of CU k + 1 are written) */
                                                             /* int16 levels + u8 reconstruction + bin tokens per trial */
