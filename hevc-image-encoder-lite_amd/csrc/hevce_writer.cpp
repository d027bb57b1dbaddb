/*
 * hevce_writer.cpp -- host half of the encoder: stream headers and the CABAC byte writer that replays the
 * decision records produced by the HIP analysis kernel into the final .h265 bytes.
 *
 * The device already made every decision with the exact coder state; this file only turns the chosen
 * syntax back into bins and the bins into bytes.  It replaces, on the host side,
 *   reference src/HEVCe.c:625-691   (VPS / SPS / PPS / slice header),
 *   reference src/HEVCe.c:797-933   (arithmetic coder, byte-emitting form, start-code emulation prevention),
 *   reference src/HEVCe.c:943-1340  (syntax binarisation) for the WINNING candidates only,
 *   reference src/HEVCe.c:1630-1641 (terminate bins, finish).
 * Pure host code: usable (and unit-tested) without a GPU.
 */
#include "hevce_common.h"
#include "hevce_tables.h"
#include "../../include/hevce_mi355x.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

const unsigned kLps4[64] = HEVCE_TAB_LPS4;
const uint8_t kLpsNext[64] = HEVCE_TAB_LPSNEXT;
const uint8_t kCtxInit[HEVCE_NCTXP] = HEVCE_TAB_CTXINIT;
const uint8_t kLastGrp[32] = HEVCE_TAB_LASTGRP;
const uint8_t kLastMin[10] = HEVCE_TAB_LASTMIN;
const uint8_t kSig4[16] = HEVCE_TAB_SIG4;
const uint8_t kSigPos[8] = HEVCE_TAB_SIGPOS;

/* scan orders as (y,x) pairs: diagonal for 4/8/16/32, horizontal and vertical for 8 (4x4 uses their first 16) */
struct Scans {
    uint8_t d4[16][2], d8[64][2], d16[256][2], d32[1024][2], h8[64][2], v8[64][2];
    Scans() {
        diag(4, d4); diag(8, d8); diag(16, d16); diag(32, d32);
        for (int g = 0; g < 4; g++)
            for (int i = 0; i < 16; i++) {
                h8[g * 16 + i][0] = uint8_t((g >> 1) * 4 + (i >> 2)); h8[g * 16 + i][1] = uint8_t((g & 1) * 4 + (i & 3));
                v8[g * 16 + i][0] = uint8_t((g & 1) * 4 + (i & 3));  v8[g * 16 + i][1] = uint8_t((g >> 1) * 4 + (i >> 2));
            }
    }
    static void diag(int n, uint8_t (*out)[2]) {
        const int ng = n / 4;
        int k = 0;
        for (int gd = 0; gd <= 2 * (ng - 1); gd++)
            for (int gy = (gd < ng ? gd : ng - 1); gy >= 0 && gd - gy < ng; gy--)
                for (int d = 0; d <= 6; d++)
                    for (int y = (d < 4 ? d : 3); y >= 0 && d - y < 4; y--, k++) {
                        out[k][0] = uint8_t(gy * 4 + y);
                        out[k][1] = uint8_t((gd - gy) * 4 + d - y);
                    }
    }
};
const Scans &scans() { static const Scans s; return s; }

inline int log2n(int n) { return n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5; }

/* ---- byte-emitting arithmetic coder ---- */
class Cabac {
public:
    void start(uint8_t *dst, int qpd6) {
        out_ = dst; n_ = 0; range_ = 510; low_ = 0; bits_ = 23; pending_ = 0; held_ = 0xff; zeros_ = 0;
        const int qp = 6 * qpd6 + 4;
        for (int i = 0; i < HEVCE_NCTX; i++) {
            const int iv = kCtxInit[i];
            int st = ((((iv >> 4) * 5 - 45) * qp) >> 4) + ((iv & 15) << 3) - 16;
            st = st < 1 ? 1 : (st > 126 ? 126 : st);
            ctx_[i] = uint8_t(st >= 64 ? (((st - 64) << 1) | 1) : ((63 - st) << 1));
        }
    }
    void bin(int ci, int b) {
        uint8_t &cx = ctx_[ci];
        const int st = cx >> 1, mps = cx & 1;
        const int lps = int((kLps4[st] >> (((range_ >> 6) & 3) << 3)) & 0xff);
        range_ -= lps;
        if ((b != 0) != (mps != 0)) {
            const int sh = lps < 8 ? 6 : __builtin_clz(unsigned(lps >> 3)) - 26;
            cx = uint8_t(st == 0 ? (mps ^ 1) : ((kLpsNext[st] << 1) | mps));
            low_ = (low_ + uint32_t(range_)) << sh;
            range_ = lps << sh;
            bits_ -= sh;
        } else {
            if (st < 62) cx = uint8_t(cx + 2);
            if (range_ < 256) { low_ <<= 1; range_ <<= 1; bits_--; }
        }
        carry();
    }
    void bypass(int v, int len) {                         /* chunks of at most 8 bins, like the reference */
        v &= (1 << len) - 1;
        while (len > 0) {
            const int n = len < 8 ? len : 8;
            len -= n;
            low_ = (low_ << n) + uint32_t(range_) * uint32_t((v >> len) & ((1 << n) - 1));
            bits_ -= n;
            carry();
        }
    }
    void terminate(int b) {
        range_ -= 2;
        if (b) { low_ = (low_ + uint32_t(range_)) << 7; range_ = 256; bits_ -= 7; }
        else if (range_ < 256) { low_ <<= 1; range_ <<= 1; bits_--; }
        carry();
    }
    int finish() {
        int fill = 0;
        if ((low_ >> (32 - bits_)) > 0) { put(held_ + 1); low_ -= 1u << (32 - bits_); }
        else { if (pending_ > 0) put(held_); fill = 0xff; }
        for (; pending_ > 1; pending_--) put(fill);
        const int tail = int((low_ >> 8) << bits_);
        put(tail >> 16); put(tail >> 8); put(tail);
        return n_;
    }
    /* registers, for cross-checks against the device's per-CTU state */
    void regs(int32_t r[8]) const {
        r[0] = range_; r[1] = int32_t(low_); r[2] = bits_; r[3] = pending_; r[4] = held_; r[5] = zeros_; r[6] = n_;
        r[7] = 8 * (n_ + pending_) + 23 - bits_;
    }
private:
    void put(int byte) {
        byte &= 0xff;
        if (zeros_ >= 2 && byte <= 3) { out_[n_++] = 3; zeros_ = 0; }
        out_[n_++] = uint8_t(byte);
        zeros_ = byte ? 0 : zeros_ + 1;
    }
    void carry() {
        if (bits_ >= 12) return;
        const int lead = int(low_ >> (24 - bits_));
        bits_ += 8;
        low_ &= 0xffffffffu >> bits_;
        if (lead == 0xff) { pending_++; return; }
        if (pending_ > 0) {
            const int c = lead >> 8, fill = (0xff + c) & 0xff;
            put(held_ + c);
            held_ = lead & 0xff;
            for (; pending_ > 1; pending_--) put(fill);
        } else { pending_ = 1; held_ = lead; }
    }
    uint8_t *out_; int n_;
    int range_, bits_, pending_, held_, zeros_;
    uint32_t low_;
    uint8_t ctx_[HEVCE_NCTXP];
};

/* ---- syntax of the chosen quad-tree ---- */
class StreamWriter {
public:
    StreamWriter(int H, int W, int q) : H_(H), W_(W), q_(q), mw_(W / 4), cu_(size_t(H / 4) * (W / 4), 32), pm_(size_t(H / 4) * (W / 4), 1) {}

    int run(uint8_t *out, const uint8_t *records) {
        int len = hevce_write_header(out, q_, H_, W_);
        c_.start(out + len, q_);
        size_t k = 0;
        for (int y = 0; y < H_; y += 32)
            for (int x = 0; x < W_; x += 32, k++) {
                rec_ = reinterpret_cast<const HevceRecord *>(records + k * HEVCE_REC_BYTES);
                cy_ = y; cx_ = x;
                cu(y, x, 32);
                c_.terminate(y + 32 >= H_ && x + 32 >= W_);
            }
        return len + c_.finish();
    }
private:
    /* neighbour context with the reference's line-buffer semantics: CU size 32 outside the picture;
       mode DC outside the picture and ACROSS A CTU-ROW BOUNDARY (reference HEVCe.c:1634-1637 never scrolls the mode line) */
    int cuLeft(int y, int x) const  { return x == 0 ? 32 : cu_[size_t(y / 4) * mw_ + x / 4 - 1]; }
    int cuAbove(int y, int x) const { return y == 0 ? 32 : cu_[size_t(y / 4 - 1) * mw_ + x / 4]; }
    int pmLeft(int y, int x) const  { return x == 0 ? 1 : pm_[size_t(y / 4) * mw_ + x / 4 - 1]; }
    int pmAbove(int y, int x) const { return (y & 31) == 0 ? 1 : pm_[size_t(y / 4 - 1) * mw_ + x / 4]; }

    void cu(int y, int x, int n) {
        const int i4 = (y - cy_) / 4, j4 = (x - cx_) / 4;
        const int info = rec_->info[i4][j4], size = 8 << (info & HEVCE_INFO_LOG2_MASK);
        const int inc = (n > cuLeft(y, x)) + (n > cuAbove(y, x));
        if (size < n) {
            if (n >= 16) c_.bin(CX_SPLIT_CU + inc, 1);
            for (int s = 0; s < 4; s++) cu(y + (s >> 1) * (n / 2), x + (s & 1) * (n / 2), n / 2);
            return;
        }
        if (n >= 16) c_.bin(CX_SPLIT_CU + inc, 0);
        const bool nxn = (info & HEVCE_INFO_NXN) != 0, split = (info & HEVCE_INFO_TUSPLIT) != 0;
        int modes[4], left[4], above[4];
        if (nxn) {
            for (int s = 0; s < 4; s++) modes[s] = rec_->mode[i4 + (s >> 1)][j4 + (s & 1)];
            left[0] = pmLeft(y, x);      above[0] = pmAbove(y, x);
            left[1] = modes[0];          above[1] = pmAbove(y, x + 4);
            left[2] = pmLeft(y + 4, x);  above[2] = modes[0];
            left[3] = modes[2];          above[3] = modes[1];
        } else {
            modes[0] = rec_->mode[i4][j4]; left[0] = pmLeft(y, x); above[0] = pmAbove(y, x);
        }
        if (n == 8) c_.bin(CX_PART, !nxn);
        lumaModes(nxn ? 4 : 1, modes, left, above);
        c_.bin(CX_UVPM, 0);
        if (!nxn) c_.bin(CX_SPLIT_TU + (n == 32 ? 0 : n == 16 ? 1 : 2), split);
        c_.bin(CX_UVCBF, 0);
        c_.bin(CX_UVCBF, 0);
        if (!nxn && !split) {
            const bool cbf = nonzero(y, x, n);
            c_.bin(CX_YCBF + 1, cbf);
            if (cbf) residual(y, x, n, modes[0]);
        } else {
            const int h = n / 2;
            for (int s = 0; s < 4; s++) {
                const int sy = y + (s >> 1) * h, sx = x + (s & 1) * h;
                const bool cbf = nonzero(sy, sx, h);
                c_.bin(CX_YCBF, cbf);
                if (cbf) residual(sy, sx, h, modes[nxn ? s : 0]);
            }
        }
        for (int i = 0; i < n / 4; i++)
            for (int j = 0; j < n / 4; j++) {
                cu_[size_t(y / 4 + i) * mw_ + x / 4 + j] = uint8_t(n);
                pm_[size_t(y / 4 + i) * mw_ + x / 4 + j] = rec_->mode[i4 + i][j4 + j];
            }
    }

    void lumaModes(int parts, const int *mode, const int *left, const int *above) {
        int hit[4], rem[4];
        for (int i = 0; i < parts; i++) {
            int p[3];
            if (left[i] != above[i]) {
                p[0] = left[i]; p[1] = above[i];
                p[2] = (left[i] != 0 && above[i] != 0) ? 0 : (left[i] + above[i] < 2 ? 26 : 1);
            } else if (left[i] > 1) {
                p[0] = left[i]; p[1] = ((left[i] + 29) & 31) + 2; p[2] = ((left[i] - 1) & 31) + 2;
            } else { p[0] = 0; p[1] = 1; p[2] = 26; }
            hit[i] = -1;
            for (int j = 0; j < 3; j++) if (p[j] == mode[i]) hit[i] = j;
            c_.bin(CX_YPM, hit[i] >= 0);
            int r = mode[i];
            for (int pass = 0; pass < 3; pass++) {          /* remove the candidates in descending order */
                int big = -1, at = -1;
                for (int j = 0; j < 3; j++) if (p[j] > big) { big = p[j]; at = j; }
                if (r > big) r--;
                p[at] = -1;
            }
            rem[i] = r;
        }
        for (int i = 0; i < parts; i++) {
            if (hit[i] >= 0) { c_.bypass(hit[i] > 0, 1); if (hit[i] > 0) c_.bypass(hit[i] - 1, 1); }
            else c_.bypass(rem[i], 5);
        }
    }

    int lev(int y, int x) const { return rec_->lev[y - cy_][x - cx_]; }
    bool nonzero(int y, int x, int n) const {
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) if (lev(y + i, x + j)) return true;
        return false;
    }

    void residual(int y0, int x0, int n, int mode) {
        const Scans &S = scans();
        int kind = 0;
        if (n <= 8) { if (std::abs(mode - 26) <= 4) kind = 1; else if (std::abs(mode - 10) <= 4) kind = 2; }
        const uint8_t (*sc)[2] = kind == 1 ? S.h8 : kind == 2 ? S.v8 : n == 4 ? S.d4 : n == 8 ? S.d8 : n == 16 ? S.d16 : S.d32;
        const int ncg = n / 4, lg = log2n(n);
        bool cgsig[8][8] = {};
        int last = 0;
        for (int i = 0; i < n * n; i++)
            if (lev(y0 + sc[i][0], x0 + sc[i][1])) { cgsig[sc[i][0] >> 2][sc[i][1] >> 2] = true; last = i; }
        {   /* last significant position */
            const int cls = lg - 2, sh = n == 4 ? 0 : 1;
            int ty = kind == 2 ? sc[last][1] : sc[last][0], tx = kind == 2 ? sc[last][0] : sc[last][1];
            const int gy = kLastGrp[ty], gx = kLastGrp[tx], gmax = kLastGrp[n - 1];
            for (int i = 0; i < gx; i++) c_.bin(CX_LASTX + 5 * cls + (i >> sh), 1);
            if (gx < gmax) c_.bin(CX_LASTX + 5 * cls + (gx >> sh), 0);
            for (int i = 0; i < gy; i++) c_.bin(CX_LASTY + 5 * cls + (i >> sh), 1);
            if (gy < gmax) c_.bin(CX_LASTY + 5 * cls + (gy >> sh), 0);
            if (gx > 3) { tx -= kLastMin[gx]; for (int i = ((gx - 2) >> 1) - 1; i >= 0; i--) c_.bypass((tx >> i) & 1, 1); }
            if (gy > 3) { ty -= kLastMin[gy]; for (int i = ((gy - 2) >> 1) - 1; i >= 0; i--) c_.bypass((ty >> i) & 1, 1); }
        }
        int nz = 0, signs = 0, nb = 0, c1 = 1, mag[16];
        for (int i = last; i >= 0; i--) {
            const int y = sc[i][0], x = sc[i][1], gy = y >> 2, gx = x >> 2, v = lev(y0 + y, x0 + x);
            const bool firstCg = (gy | gx) == 0, atLast = i == last, cgFirst = (i & 15) == 0;
            if ((i & 15) == 15 || atLast) {
                const int right = gx < ncg - 1 && cgsig[gy][gx + 1], below = gy < ncg - 1 && cgsig[gy + 1][gx];
                nb = (below << 1) | right; nz = 0; signs = 0;
                if (!firstCg && !atLast) c_.bin(CX_SIGCG + (nb != 0), cgsig[gy][gx]);
            }
            if (!atLast && (firstCg || (cgsig[gy][gx] && (!cgFirst || nz > 0)))) {
                int ci;
                if ((y | x) == 0) ci = 0;
                else if (n == 4) ci = kSig4[(y << 2) | x];
                else {
                    ci = 9 + (n >= 16 ? 12 : 0) + ((n == 8 && kind != 0) ? 6 : 0) + (firstCg ? 0 : 3);
                    ci += nb == 0 ? kSigPos[(y & 3) + (x & 3)] : nb == 1 ? kSigPos[(y & 3) << 1] : nb == 2 ? kSigPos[(x & 3) << 1] : 2;
                }
                c_.bin(CX_SIG + ci, v != 0);
            }
            if (v) { mag[nz++] = std::abs(v); signs = (signs << 1) | (v < 0); }
            if (cgFirst && nz > 0) {
                const int set = (firstCg ? 0 : 2) + (c1 == 0);
                int esc = nz > 8, g2 = -1;
                c1 = 1;
                for (int j = 0; j < 8 && j < nz; j++) {
                    c_.bin(CX_ONE + 4 * set + c1, mag[j] > 1);
                    if (mag[j] > 1) { c1 = 0; if (g2 < 0) g2 = mag[j] > 2; else esc = 1; }
                    else if (c1 > 0 && c1 < 3) c1++;
                }
                if (c1 == 0 && g2 >= 0) { c_.bin(CX_ABS + set, g2); esc |= g2; }
                c_.bypass(signs, nz);
                if (esc) {
                    int base = 3, rice = 0;
                    for (int j = 0; j < nz; j++) {
                        const int rem = mag[j] - (j < 8 ? base : 1);
                        if (rem >= 0) {
                            remainder(rem, rice);
                            if (mag[j] > (3 << rice)) rice = rice < 4 ? rice + 1 : 4;
                        }
                        if (mag[j] >= 2) base = 2;
                    }
                }
            }
        }
    }
    void remainder(int v, int k) {
        if (v < (3 << k)) {
            const int pre = v >> k;
            c_.bypass((1 << (pre + 1)) - 2, pre + 1);
            c_.bypass(v & ((1 << k) - 1), k);
        } else {
            int len = k;
            v -= 3 << k;
            for (; v >= (1 << len); len++) v -= 1 << len;
            c_.bypass((1 << (4 + len - k)) - 2, 4 + len - k);
            c_.bypass(v, len);
        }
    }

    int H_, W_, q_, mw_;
    std::vector<uint8_t> cu_, pm_;
    Cabac c_;
    const HevceRecord *rec_ = nullptr;
    int cy_ = 0, cx_ = 0;
};

/* MSB-first bit packer for the parameter sets */
struct Bits {
    uint8_t *p; unsigned acc = 0; int n = 0;
    void put(unsigned v, int len) { for (int i = len - 1; i >= 0; i--) { acc = (acc << 1) | ((v >> i) & 1); if (++n == 8) { *p++ = uint8_t(acc); acc = 0; n = 0; } } }
    void ue(unsigned v) { int len = 0; v++; for (unsigned t = v; t > 1; t >>= 1) len++; put(0, len); put(v, len + 1); }
    void align() { if (n) put(0, 8 - n); }
};

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int hevce_write_header(unsigned char *out, int qpd6, int H, int W) {
    /* fixed-profile parameter sets: Main Still Picture, 4:2:0 with empty chroma, CTB 32, min CB 8, TB 4..32,
       transform hierarchy depth 1, one slice, no tiles / wavefronts  [ref HEVCe.c:666-690] */
    static const uint8_t vps[] = { 0, 0, 1, 0x40, 1, 0x0c, 1, 0xff, 0xff, 3, 0x10, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0xb4, 0xf0, 0x24 };
    static const uint8_t sps[] = { 0, 0, 1, 0x42, 1, 1, 3, 0x10, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0, 3, 0, 0xb4 };
    static const uint8_t pps[] = { 0, 0, 1, 0x44, 1, 0xc0, 0x90, 0x91, 0x81, 0xd9, 0x20 };
    static const uint8_t slice[] = { 0, 0, 1, 0x26, 1, 0xac };
    static const uint8_t qp_bits[5][2] = { { 0x16, 0xde }, { 0x10, 0xde }, { 0x2b, 0x78 }, { 0x4d, 0xe0 }, { 0x97, 0x80 } };
    uint8_t *p = out;
    memcpy(p, vps, sizeof vps); p += sizeof vps;
    memcpy(p, sps, sizeof sps); p += sizeof sps;
    Bits b{ p };
    b.put(0xa, 4);
    b.ue(unsigned(W)); b.ue(unsigned(H));
    b.put(0x197ee4, 22);
    b.put(0x681ed1, 24);
    b.align();
    p = b.p;
    memcpy(p, pps, sizeof pps); p += sizeof pps;
    memcpy(p, slice, sizeof slice); p += sizeof slice;
    *p++ = qp_bits[qpd6][0]; *p++ = qp_bits[qpd6][1];
    return int(p - out);
}

int hevce_write_stream(unsigned char *out, const unsigned char *records, int H, int W, int qpd6) {
    StreamWriter w(H, W, qpd6);
    return w.run(out, records);
}

int hevce_record_bytes(void) { return HEVCE_REC_BYTES; }

long hevce_stream_bound(int ysz, int xsz) {
    /* header + one coder buffer per CTU (the reference cannot exceed 3200 bytes per CTU, HEVCe.c:795) + finish */
    const long H = ((ysz < HEVCE_MAXDIM ? ysz : HEVCE_MAXDIM) + 31) / 32, W = ((xsz < HEVCE_MAXDIM ? xsz : HEVCE_MAXDIM) + 31) / 32;
    return 256 + H * W * 3200L + 16;
}

#pragma GCC visibility pop
}  // extern "C"
