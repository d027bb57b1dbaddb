/*
 * hevce_cli.cpp -- the reference's command-line contract on top of libhevce_mi355x.so.
 *
 *   HEVCe <input.pgm> <output.h265> [<qpd6>] [<reconstructed.pgm>]
 *
 * Mirrors reference src/HEVCeMain.c:138-230 as an interface, not as code:
 *   - arguments: a single character '0'..'4' anywhere is qpd6 (the last one wins), the other arguments are, in order,
 *     input PGM, output stream, output reconstruction; further ones are ignored; default qpd6 = 3      [ref :148-170]
 *   - exit status 0, or -1 (255) after the usage text / "open ... failed" / "write file ... failed"    [ref :162-167,185,217,225]
 *   - stdout report with the reference's exact line formats                                            [ref :174-211]
 *   - PGM dialect of the reader [ref :9-64]: "P5", three decimal integers as fscanf("%d") reads them, maxval <= 255,
 *     exactly one separator byte out of ' ' '\n' '\r' '\t', then width*height samples.  A file that is exactly ONE
 *     sample short is accepted and its last sample reads as 255 (the reference tests feof() before each fgetc());
 *     two or more missing samples are an error.  Writer [ref :69-90]: "P5\n<w> <h>\n255\n" + samples.
 * Deviations (the reference has undefined behaviour there): non-positive sizes and images above 8192*8192 samples are
 * load errors instead of buffer overruns; a failing GPU path (no device) prints the library's message on stderr and
 * exits with -1 -- the reference encoder cannot fail.
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hevce_mi355x.h"
#include "hevce_pgm.h"

using namespace hevce_pgm;

namespace {

/* distortion over the common top-left area  [ref :116-133] */
double psnr_of(const unsigned char *a, int ya, int xa, const unsigned char *b, int yb, int xb, double *mse) {
    const int ymin = ya < yb ? ya : yb, xmin = xa < xb ? xa : xb;
    long long sse = 0;
    for (int y = 0; y < ymin; y++)
        for (int x = 0; x < xmin; x++) {
            const long long d = (long long)a[(size_t)y * xa + x] - b[(size_t)y * xb + x];
            sse += d * d;
        }
    *mse = ((double)sse) / ymin / xmin;
    if (*mse < 1e-9) *mse = 1e-9;
    return 10.0 * log10(255 * 255 / (*mse));
}

}  // namespace

int main(int argc, char **argv) {
    const char *in_name = nullptr, *out_name = nullptr, *rcon_name = nullptr;
    int qpd6 = -1;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] >= '0' && a[0] <= '4' && a[1] == '\0') qpd6 = a[0] - '0';
        else if (!in_name) in_name = a;
        else if (!out_name) out_name = a;
        else if (!rcon_name) rcon_name = a;
    }
    if (!in_name || !out_name) {
        printf("Usage:\n");
        printf("    %s  <input-image-file(.pgm)>  <output-file(.hevc/.h265)>  [<qpd6>]  [<output-reconstructed-image-file(.pgm)>]\n", argv[0]);
        printf("\n");
        return -1;
    }
    if (qpd6 < 0 || qpd6 > 4) qpd6 = 3;

    printf("arguments:\n");
    printf("  input  image file               = %s\n", in_name);
    printf("  output stream file              = %s\n", out_name);
    printf("  Qp%%6                            = %d     (Qp=%d)\n", qpd6, qpd6 * 6 + 4);
    if (rcon_name) printf("  output reconstructed image file = %s\n", rcon_name);

    std::vector<unsigned char> img;
    int ysz = -1, xsz = -1;
    if (!load_pgm(in_name, img, &ysz, &xsz)) {
        printf("open %s failed\n", in_name);
        return -1;
    }
    printf("  image size                      = %d x %d\n", xsz, ysz);
    printf("compressing...\n");
    fflush(stdout);

    int yszn = ysz, xszn = xsz;
    const int yc = ysz > 8192 ? 8192 : ysz, xc = xsz > 8192 ? 8192 : xsz;
    std::vector<unsigned char> rcon((size_t)((yc + 31) / 32 * 32) * (size_t)((xc + 31) / 32 * 32));
    std::vector<unsigned char> stream((size_t)hevce_stream_bound(ysz, xsz));
    const int len = HEVCImageEncoder(stream.data(), img.data(), rcon.data(), &yszn, &xszn, qpd6);
    if (len <= 0) {
        fprintf(stderr, "HEVCe: the MI355X path failed: %s\n", hevce_last_error());
        return -1;
    }

    double mse;
    const double psnr = psnr_of(img.data(), ysz, xsz, rcon.data(), yszn, xszn, &mse);
    printf("  padded image size               = %d x %d\n", xszn, yszn);
    printf("  original   length               = %d Bytes\n", xszn * yszn);
    printf("  compressed length               = %d Bytes\n", len);
    printf("  compression ratio               = %.5f\n", 1.0 * xszn * yszn / len);
    printf("  bits per pixel                  = %.5f\n", 8.0 * len / (xszn * yszn));
    printf("  mean square error (MSE)         = %.7lf\n", mse);
    printf("  peak signal/noise ratio (PSNR)  = %.4lf dB\n", psnr);

    if (!write_all(out_name, std::string(), stream.data(), (size_t)len)) {
        printf("write file %s failed\n", out_name);
        return -1;
    }
    if (rcon_name) {
        char head[64];
        snprintf(head, sizeof head, "P5\n%d %d\n255\n", xszn, yszn);
        if (!write_all(rcon_name, head, rcon.data(), (size_t)xszn * (size_t)yszn)) {
            printf("write file %s failed\n", rcon_name);
            return -1;
        }
    }
    return 0;
}
