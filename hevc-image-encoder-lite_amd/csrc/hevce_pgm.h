/*
 * hevce_pgm.h -- the reference CLI's PGM dialect (reader: reference src/HEVCeMain.c:9-64, writer: :69-90), shared by
 * bin/HEVCe and bin/HEVCeDir.  See hevce_cli.cpp for the rules (one-sample-short files, separators, deviations).
 */
#pragma once
#include <cstdio>
#include <string>
#include <vector>

namespace hevce_pgm {

/* ---- PGM reader over the whole file image ------------------------------------------------------------------ */
struct Bytes {
    std::vector<unsigned char> d;
    size_t pos = 0;
    int get() { return pos < d.size() ? d[pos++] : (pos++, EOF); }       /* like fgetc: EOF past the end */
    bool at_eof_flag() const { return pos > d.size(); }                   /* like feof: set only after a read hit the end */
};

inline bool is_space(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

/* one fscanf("%d"): skip white space, optional sign, at least one digit; the first non-digit stays unread */
inline bool scan_int(Bytes &b, int *out) {
    while (b.pos < b.d.size() && is_space(b.d[b.pos])) b.pos++;
    size_t p = b.pos;
    bool neg = false;
    if (p < b.d.size() && (b.d[p] == '+' || b.d[p] == '-')) { neg = b.d[p] == '-'; p++; }
    if (p >= b.d.size() || b.d[p] < '0' || b.d[p] > '9') return false;
    long long v = 0;
    while (p < b.d.size() && b.d[p] >= '0' && b.d[p] <= '9') { v = v * 10 + (b.d[p] - '0'); if (v > 0x7fffffffLL) v = 0x7fffffffLL; p++; }
    b.pos = p;
    *out = (int)(neg ? -v : v);
    return true;
}

inline bool load_pgm(const char *name, std::vector<unsigned char> &img, int *ysz, int *xsz) {
    FILE *fp = fopen(name, "rb");
    if (!fp) return false;
    Bytes b;
    unsigned char chunk[1 << 16];
    size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, fp)) > 0) b.d.insert(b.d.end(), chunk, chunk + n);
    fclose(fp);
    int maxval = -1;
    if (b.get() != 'P' || b.get() != '5') return false;
    if (!scan_int(b, xsz) || !scan_int(b, ysz) || !scan_int(b, &maxval)) return false;
    if (maxval > 255) return false;
    const int sep = b.get();
    if (sep != ' ' && sep != '\n' && sep != '\r' && sep != '\t') return false;
    if (*xsz <= 0 || *ysz <= 0 || (long long)*xsz * *ysz > 8192LL * 8192LL) return false;   /* deviation: see header */
    const size_t count = (size_t)*xsz * (size_t)*ysz;
    img.resize(count);
    for (size_t i = 0; i < count; i++) {
        if (b.at_eof_flag()) return false;                     /* pixels not enough */
        img[i] = (unsigned char)b.get();                        /* the first read past the end yields EOF = 0xff */
    }
    return true;
}

inline bool write_all(const char *name, const std::string &head, const unsigned char *data, size_t len) {
    FILE *fp = fopen(name, "wb");
    if (!fp) return false;
    bool ok = head.empty() || fwrite(head.data(), 1, head.size(), fp) == head.size();
    ok = ok && (len == 0 || fwrite(data, 1, len, fp) == len);
    ok = (fclose(fp) == 0) && ok;
    return ok;
}

}  // namespace hevce_pgm
