/*
 * hevce_dir.cpp -- batch front-end: every *.pgm of a directory -> <out-dir>/<name>.h265
 * (and <name>.pgm, the reconstruction, with --rcon).  No reference equivalent (its evaluation script spawns the CLI once
 * per image, reference HEVCeval.py:159-186); every output is byte-identical to what bin/HEVCe writes for that file.
 *
 *   HEVCeDir <in-dir> <out-dir> [<qpd6>] [--rcon] [--batch N]
 *
 * One picture is one serial chain on one workgroup, so pictures are what fills the GPU: the files are encoded in batches
 * of up to N pictures (default 256, one per CU).  Two jobs alternate: while the kernel of one batch runs, the next batch's
 * files are read, uploaded and launched behind it, then the running batch's results are collected and written
 * (hevce_job_launch / hevce_job_finish).
 * Same PGM dialect and the same failure rule as the CLI: a file that does not load is reported ("open <file> failed") and
 * makes the exit status -1, the others are still encoded.
 */
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dirent.h>
#include <sys/stat.h>

#include "../../include/hevce_mi355x.h"
#include "hevce_pgm.h"

using namespace hevce_pgm;

namespace {

struct Batch {
    std::vector<std::string> names;
    std::vector<std::vector<unsigned char>> imgs;
    std::vector<int> ys, xs, qs;
    hevce_job *job = nullptr;
};

bool ends_with(const std::string &s, const char *suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

/* read + upload + launch */
bool start(Batch &b, const std::string &in_dir, int qpd6) {
    if (b.names.empty()) return true;
    b.qs.assign(b.names.size(), qpd6);
    b.job = hevce_job_create(0, (int)b.names.size(), b.ys.data(), b.xs.data(), b.qs.data());
    if (!b.job) return false;
    for (size_t i = 0; i < b.names.size(); i++) if (hevce_job_upload(b.job, (int)i, b.imgs[i].data()) != 0) return false;
    return hevce_job_launch(b.job) == 0;
}

/* wait + write results */
bool finish(Batch &b, const std::string &out_dir, bool rcon, long *bytes, long long *pixels) {
    if (!b.job) return true;
    bool ok = hevce_job_finish(b.job) == 0;
    for (size_t i = 0; ok && i < b.names.size(); i++) {
        const std::string stem = out_dir + "/" + b.names[i].substr(0, b.names[i].size() - 4);
        int len = 0, H = 0, W = 0;
        const unsigned char *s = hevce_job_stream(b.job, (int)i, &len);
        hevce_job_padded_size(b.job, (int)i, &H, &W);
        ok = s && write_all((stem + ".h265").c_str(), std::string(), s, (size_t)len);
        if (ok && rcon) {
            std::vector<unsigned char> r((size_t)H * W);
            char head[64];
            snprintf(head, sizeof head, "P5\n%d %d\n255\n", W, H);
            ok = hevce_job_download_rcon(b.job, (int)i, r.data()) == 0 && write_all((stem + ".pgm").c_str(), head, r.data(), r.size());
        }
        if (!ok) printf("write file %s failed\n", stem.c_str());
        *bytes += len; *pixels += (long long)H * W;
    }
    hevce_job_destroy(b.job);
    b = Batch();
    return ok;
}

}  // namespace

int main(int argc, char **argv) {
    const char *in_dir = nullptr, *out_dir = nullptr;
    int qpd6 = -1, batch = 256;
    bool rcon = false;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "--rcon")) rcon = true;
        else if (!strcmp(a, "--batch") && i + 1 < argc) batch = atoi(argv[++i]);
        else if (a[0] >= '0' && a[0] <= '4' && a[1] == '\0') qpd6 = a[0] - '0';
        else if (!in_dir) in_dir = a;
        else if (!out_dir) out_dir = a;
    }
    if (!in_dir || !out_dir || batch <= 0) {
        printf("Usage:\n    %s  <input-directory>  <output-directory>  [<qpd6>]  [--rcon]  [--batch N]\n\n", argv[0]);
        return -1;
    }
    if (qpd6 < 0 || qpd6 > 4) qpd6 = 3;                          /* the CLI's default */
    std::vector<std::string> files;
    if (DIR *d = opendir(in_dir)) {
        while (dirent *e = readdir(d)) { const std::string n = e->d_name; if (ends_with(n, ".pgm")) files.push_back(n); }
        closedir(d);
    } else { printf("open %s failed\n", in_dir); return -1; }
    std::sort(files.begin(), files.end());
    mkdir(out_dir, 0777);
    printf("%zu files, Qp%%6 = %d, batches of %d\n", files.size(), qpd6, batch);

    int status = 0;
    long bytes = 0; long long pixels = 0;
    Batch cur;                                                   /* launched, not yet collected */
    size_t next = 0;
    while (next < files.size() || cur.job) {
        /* read the next batch's files while the current one is on the GPU, launch it behind, then collect the current one */
        Batch nb;
        while (next < files.size() && (int)nb.names.size() < batch) {
            const std::string &n = files[next++];
            std::vector<unsigned char> img;
            int ys = 0, xs = 0;
            if (!load_pgm((std::string(in_dir) + "/" + n).c_str(), img, &ys, &xs)) { printf("open %s failed\n", n.c_str()); status = -1; continue; }
            nb.names.push_back(n); nb.imgs.push_back(std::move(img)); nb.ys.push_back(ys); nb.xs.push_back(xs);
        }
        if (!start(nb, in_dir, qpd6)) { fprintf(stderr, "HEVCeDir: the MI355X path failed: %s\n", hevce_last_error()); return -1; }
        if (!finish(cur, out_dir, rcon, &bytes, &pixels)) status = -1;
        cur = std::move(nb);
    }
    printf("%lld padded pixels -> %ld stream bytes (%.5f bits per pixel)\n", pixels, bytes, pixels ? 8.0 * bytes / pixels : 0.0);
    return status;
}
