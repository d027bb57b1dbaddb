/*
 * hevce_host.cpp -- C-ABI shim and HIP plumbing around the analysis kernel (include/hevce_mi355x.h).
 *
 * One job = a batch of independent images.  All of a job's device memory is one slab in HBM:
 *   per image: source luma | padded reconstruction | decision records | per-CTU coder state | trial scratch | slice data and its
 *   length | re-analysis statistics | the stream writer's token slots | the cluster's mailboxes
 * The kernel (hevce_kernel.hip) analyses a picture AND writes its slice data: one workgroup per picture, or -- when the device has the
 * compute units to spare -- a cluster of up to six workgroups per picture (hevce_device.h: chain, 16x16, 32x32, two 8x8 workgroups,
 * stream writer).  The host puts the VPS/SPS/PPS/slice header in front and fetches exactly the bytes of each stream.  Only with
 * HEVCE_HOST_WRITER=1 (diagnostics) do the decision records travel and are replayed into bytes on the host (hevce_writer.cpp).
 *
 * HEVCImageEncoder() keeps the reference's signature (reference src/HEVCe.h:5-12).  The reference cannot fail;
 * this implementation can (no device): it then returns a negative length and reports on stderr -- it never
 * falls back to a CPU encoder.
 */
#include "hevce_common.h"
#include "../../include/hevce_mi355x.h"

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" int hevce_launch_test_tu(const unsigned char *org, const unsigned char *nbr, int n, int q, int aL, int aLB, int aA, int aAR,
                                    short *lev, unsigned char *rec, int *info, const unsigned short *qlut, hipStream_t stream);
extern "C" int hevce_launch_test_chain(const unsigned char *org, const unsigned char *nbr, int n, int q, int aL, int aLB, int aA, int aAR,
                                       unsigned char *scratch, int *info, const unsigned short *qlut, hipStream_t stream);
extern "C" int hevce_launch_analyse(const HevceImageDesc *d_images, int n, int cluster, hipStream_t stream);   /* hevce_kernel.hip */

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    fprintf(stderr, "hevce_mi355x: %s\n", buf);
    return -1;
}
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

/* ---- the quantiser's level tables (hevce_quant.h): 5 qpd6 x 4 transform sizes x 32768 magnitudes, built once per process
   and uploaded once per device; the kernel gathers from the four tables of a picture's qpd6 ---- */
namespace quant {
#define HD inline
#define HV_I32MAX 0x7fffffff
inline int hv_min(int a, int b) { return a < b ? a : b; }
inline int hv_max(int a, int b) { return a > b ? a : b; }
inline int hv_abs(int a) { return a < 0 ? -a : a; }
inline int hv_clip16(int v) { return hv_min(hv_max(v, -32768), 32767); }
inline int hv_clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline unsigned hv_umul24(unsigned a, unsigned b) { return a * b; }
#include "hevce_quant.h"
#undef HD
#undef HV_I32MAX
}  // namespace quant
constexpr size_t kQlutPerQ = size_t(4) * HEVCE_QLUT_N;       /* entries per qpd6 */
std::mutex g_qlut_mutex;
uint16_t *g_qlut_dev[64] = {};                               /* per HIP device: 5 x kQlutPerQ entries */
const uint16_t *qlut_host() {
    static std::vector<uint16_t> tab;
    if (tab.empty()) {
        tab.resize(5 * kQlutPerQ);
        for (int q = 0; q < 5; q++)
            for (int t = 0; t < 4; t++)
                for (int av = 0; av < HEVCE_QLUT_N; av++) tab[q * kQlutPerQ + size_t(t) * HEVCE_QLUT_N + av] = (uint16_t)quant::hv_qlut_entry(q, t, av);
    }
    return tab.data();
}
/* device pointer to the tables of qpd6 q on `device` (current device must be `device`); nullptr on failure */
const uint16_t *qlut_device(int device, int q) {
    std::lock_guard<std::mutex> lock(g_qlut_mutex);
    if (device < 0 || device >= 64) return nullptr;
    if (!g_qlut_dev[device]) {
        uint16_t *d = nullptr;
        if (hipMalloc((void **)&d, 5 * kQlutPerQ * 2) != hipSuccess) return nullptr;
        if (hipMemcpy(d, qlut_host(), 5 * kQlutPerQ * 2, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
        g_qlut_dev[device] = d;
    }
    return g_qlut_dev[device] + size_t(q) * kQlutPerQ;
}

/* ---- compute units claimed by launches in flight, per device.  A cluster's workgroups wait for each other, so all of them have to be
   resident at once: a launch takes the largest cluster size that fits beside what this process already has on the device, and
   gives its units back when it has finished.  (Other processes' kernels are not seen here: the kernel's own alive check, hevce_device.h
   hv_cl_alive, catches a cluster that was not placed and the launch is repeated without clusters.)  HEVCE_CU_BUDGET=0 switches the
   accounting off (tests of that check) ---- */
std::mutex g_cu_mutex;
int g_cu_claimed[64] = {};
int cu_claim(int device, int want_per_role, int max_cluster, int cus, int *claimed) {
    static const bool on = !(getenv("HEVCE_CU_BUDGET") && atoi(getenv("HEVCE_CU_BUDGET")) == 0);
    std::lock_guard<std::mutex> lock(g_cu_mutex);
    const int held = (on && device >= 0 && device < 64) ? g_cu_claimed[device] : 0;
    int c = max_cluster;
    while (c > 1 && held + c * want_per_role > cus) c--;
    *claimed = c > 1 ? c * want_per_role : (want_per_role < cus ? want_per_role : cus);
    if (on && device >= 0 && device < 64) g_cu_claimed[device] += *claimed;
    return c;
}
void cu_release(int device, int claimed) {
    std::lock_guard<std::mutex> lock(g_cu_mutex);
    if (device >= 0 && device < 64) { g_cu_claimed[device] -= claimed; if (g_cu_claimed[device] < 0) g_cu_claimed[device] = 0; }
}

inline int pad32(int v) { return ((v < HEVCE_MAXDIM ? v : HEVCE_MAXDIM) + 31) / 32 * 32; }
inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

struct Image {
    int h0, w0, H, W, q, nctu;
    size_t off_img, off_rcon, off_rec, off_state, off_scratch;   /* into the device slab */
    size_t off_payload, off_plen, off_stats, off_wrtok, off_cluster;                     /* the kernel's stream writer: slice data, its length; re-analysis statistics */
    size_t hoff_rec, hoff_state, hoff_plen;                      /* into the pinned host slab */
    const void *ext_img = nullptr;
    long ext_stride = 0;                                         /* row stride of ext_img (0: w0) */
    std::vector<uint8_t> stream;                                 /* host-writer mode only */
    const uint8_t *stream_ptr = nullptr;                         /* the finished .h265: in the job's pinned stream slab (or `stream`) */
    int length = 0;
};

}  // namespace

struct hevce_job;
extern "C" void hevce_job_destroy(hevce_job *job);
namespace {
/* HEVCImageEncoder()'s per-thread job context (see there) */
std::atomic<bool> g_alive{true};                             /* false once the process is exiting: the HIP runtime may be gone */
struct CachedJob {
    hevce_job *j = nullptr; int device = -1, h = 0, w = 0, q = -1;
    ~CachedJob() { if (j && g_alive.load()) hevce_job_destroy(j); }
};
thread_local CachedJob g_cached;
}  // namespace

struct hevce_job {
    int device = 0, n = 0;
    std::vector<Image> im;
    uint8_t *d_slab = nullptr, *h_slab = nullptr;
    size_t d_bytes = 0, h_bytes = 0;
    HevceImageDesc *d_desc = nullptr;
    std::vector<HevceImageDesc> h_desc;
    bool desc_dirty = true, launched = false, records_fetched = false;
    uint8_t *h_streams = nullptr; size_t h_streams_bytes = 0;   /* pinned: header + slice data of every picture, back to back; grows on demand */
    bool device_emit = true;         /* the kernel emits the slice data itself; HEVCE_HOST_WRITER=1: replay the records on the host instead */
    int flags = 0;                   /* HevceImageDesc.flags (tests: HEVCE_FLAGS) */
    int cluster_fallbacks = 0;       /* launches repeated without clusters after a hand-off timed out */
    int cluster = 1;                 /* workgroups per picture (hevce_device.h cluster mode) at most: what the whole device has room for */
    int cluster_now = 1;             /* ... of the launch in flight / the last one: what was free beside this process's other launches (cu_claim) */
    bool single_once = false;        /* the next launch runs without clusters (the repeat of one whose cluster was not placed) */
    int cus = 0, cu_claimed = 0;     /* the device's compute units; those this job's launch in flight holds */
    HevceCluster *h_cluster_init = nullptr;   /* pinned: the zeroed mailboxes every launch starts from */
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    float kernel_ms = 0, d2h_ms = 0, writer_ms = 0;
    int32_t *d_dump = nullptr; size_t dump_ints = 0;   /* diagnostics (HEVCE_DUMP=1) */
    int32_t *h_trace = nullptr;      /* host-pinned diagnostics, 128 ints per image */
};

extern "C" {
#pragma GCC visibility push(default)

const char *hevce_last_error(void) { return g_err.c_str(); }
const char *hevce_version(void) { return "hevce_mi355x 0.1 (gfx950)"; }

int hevce_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

hevce_job *hevce_job_create(int device, int n, const int *yszs, const int *xszs, const int *qpd6s) {
    if (n <= 0) { fail("hevce_job_create: n must be positive"); return nullptr; }
    if (hevce_device_count() <= 0) { fail("no HIP device available: the encoder has no CPU fallback"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail("hipSetDevice(%d) failed", device); return nullptr; }
    hevce_job *j = new hevce_job;
    j->device = device; j->n = n; j->im.resize(n); j->h_desc.resize(n);
    size_t d = 0, h = 0;
    for (int i = 0; i < n; i++) {
        Image &m = j->im[i];
        if (yszs[i] <= 0 || xszs[i] <= 0 || qpd6s[i] < 0 || qpd6s[i] > 4) { fail("image %d: bad size or qpd6", i); delete j; return nullptr; }
        m.h0 = yszs[i]; m.w0 = xszs[i]; m.q = qpd6s[i];
        m.H = pad32(m.h0); m.W = pad32(m.w0); m.nctu = (m.H / 32) * (m.W / 32);
        m.off_img = d;     d = align256(d + size_t(m.h0) * m.w0);
        m.off_rcon = d;    d = align256(d + size_t(m.H) * m.W);
        m.off_rec = d;     d = align256(d + size_t(m.nctu) * HEVCE_REC_BYTES);
        m.off_state = d;   d = align256(d + size_t(m.nctu) * 32);
        m.off_scratch = d; d = align256(d + HEVCE_SCRATCH_BYTES);
        m.off_payload = d; d = align256(d + size_t(m.nctu) * 3200 + 64);
        m.off_plen = d;    d = align256(d + 4);
        m.off_stats = d;   d = align256(d + 8);
        m.off_wrtok = d;   d = align256(d + HEVCE_WRTOK_BYTES);
        m.off_cluster = d; d = align256(d + sizeof(HevceCluster));
        m.hoff_rec = h;    h = align256(h + size_t(m.nctu) * HEVCE_REC_BYTES);
        m.hoff_state = h;  h = align256(h + size_t(m.nctu) * 32);
        m.hoff_plen = h;   h = align256(h + 4);
    }
    j->d_bytes = d; j->h_bytes = h;
    bool ok = hipMalloc((void **)&j->d_slab, d) == hipSuccess
           && hipMalloc((void **)&j->d_desc, sizeof(HevceImageDesc) * n) == hipSuccess
           && hipHostMalloc((void **)&j->h_slab, h, hipHostMallocDefault) == hipSuccess
           && hipStreamCreateWithFlags(&j->own_stream, hipStreamNonBlocking) == hipSuccess
           && hipEventCreate(&j->ev0) == hipSuccess && hipEventCreate(&j->ev1) == hipSuccess && hipEventCreate(&j->ev2) == hipSuccess;
    if (!ok) { fail("hevce_job_create: device allocation failed (%zu bytes HBM, %zu bytes pinned)", d, h); hevce_job_destroy(j); return nullptr; }
    j->stream = j->own_stream;
    j->device_emit = getenv("HEVCE_HOST_WRITER") == nullptr;
    if (const char *e = getenv("HEVCE_FLAGS")) j->flags = (int)strtol(e, nullptr, 0);
    {
        /* Cluster mode needs both workgroups of every picture resident at the same time (they wait for each other): one workgroup
           per compute unit (its LDS), so at most CUs / 2 pictures; HEVCE_CLUSTER=0 / 1 overrides the choice (1 is still bounded) */
        hipDeviceProp_t prop;
        int cus = 0;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
        j->cus = cus;
        const int n8 = (n + 7) & ~7;                             /* the grid is padded to eight pictures per role (one cluster = one XCD, hevce_kernel.hip) */
        /* sizes: 2 = one more workgroup for the 16x16 and 32x32 CUs, 3 = one each, 4 = + one for the 8x8 CUs' candidates 0..69 (and
           the chain workgroup's freed wavefronts help the NxN chain), 5 = + the stream writer on a workgroup of its own, 6 = + a second
           8x8 workgroup (the candidates 0..69 shared between the two).  The per-candidate dump (HEVCE_DUMP) knows clusters up to 5 */
        const int top = getenv("HEVCE_DUMP") ? 5 : 7;          /* 7 = + a third 8x8 workgroup */
        j->cluster = 1;
        if (j->device_emit) for (int c = top; c >= 2; c--) if (c * n8 <= cus) { j->cluster = c; break; }
        if (const char *e = getenv("HEVCE_CLUSTER")) { const int mx = atoi(e); if (mx >= 0 && mx < j->cluster) j->cluster = mx < 1 ? 1 : mx; }
    }
    j->cluster_now = j->cluster;
    if (j->cluster > 1) {
        if (hipHostMalloc((void **)&j->h_cluster_init, sizeof(HevceCluster), hipHostMallocDefault) != hipSuccess) { fail("hevce_job_create: pinned allocation failed"); hevce_job_destroy(j); return nullptr; }
        memset(j->h_cluster_init, 0, sizeof(HevceCluster));      /* sequence words start at 0 */
    }
    if (getenv("HEVCE_DUMP")) {
        for (int i = 0; i < n; i++) j->dump_ints += size_t(j->im[i].nctu) * 21 * HEVCE_TRIALS * 16;
        if (hipMalloc((void **)&j->d_dump, j->dump_ints * 4) != hipSuccess) j->d_dump = nullptr;
        else (void)hipMemset(j->d_dump, 0, j->dump_ints * 4);
    }
    if (getenv("HEVCE_TRACE")) {
        if (hipHostMalloc((void **)&j->h_trace, sizeof(int32_t) * 1024 * n, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) memset(j->h_trace, 0, sizeof(int32_t) * 1024 * n);
        else j->h_trace = nullptr;
    }
    return j;
}

void hevce_job_destroy(hevce_job *j) {
    if (!j) return;
    (void)hipSetDevice(j->device);
    if (j->launched) (void)hipStreamSynchronize(j->stream);
    if (j->cu_claimed) { cu_release(j->device, j->cu_claimed); j->cu_claimed = 0; }
    if (j->h_cluster_init) (void)hipHostFree(j->h_cluster_init);
    if (j->d_slab) (void)hipFree(j->d_slab);
    if (j->d_desc) (void)hipFree(j->d_desc);
    if (j->h_slab) (void)hipHostFree(j->h_slab);
    if (j->h_streams) (void)hipHostFree(j->h_streams);
    if (j->h_trace) (void)hipHostFree(j->h_trace);
    if (j->d_dump) (void)hipFree(j->d_dump);
    if (j->ev0) (void)hipEventDestroy(j->ev0);
    if (j->ev1) (void)hipEventDestroy(j->ev1);
    if (j->ev2) (void)hipEventDestroy(j->ev2);
    if (j->own_stream) (void)hipStreamDestroy(j->own_stream);
    delete j;
}

int hevce_job_upload(hevce_job *j, int i, const unsigned char *img) {
    if (!j || i < 0 || i >= j->n) return fail("hevce_job_upload: bad arguments");
    HIPCHK(hipSetDevice(j->device));
    Image &m = j->im[i];
    m.ext_img = nullptr; m.ext_stride = 0; j->desc_dirty = true;
    /* stream-ordered before the kernel; a pageable source has been staged when the call returns, so the caller may reuse it */
    HIPCHK(hipMemcpyAsync(j->d_slab + m.off_img, img, size_t(m.h0) * m.w0, hipMemcpyHostToDevice, j->stream));
    return 0;
}

/* picture i = the h0 x w0 window at `origin` of a larger host picture with row stride `stride` (tile-pictures) */
int hevce_job_upload_window(hevce_job *j, int i, const unsigned char *origin, long stride) {
    if (!j || i < 0 || i >= j->n || !origin || stride < j->im[i].w0) return fail("hevce_job_upload_window: bad arguments");
    HIPCHK(hipSetDevice(j->device));
    Image &m = j->im[i];
    m.ext_img = nullptr; m.ext_stride = 0; j->desc_dirty = true;
    HIPCHK(hipMemcpy2DAsync(j->d_slab + m.off_img, size_t(m.w0), origin, size_t(stride), size_t(m.w0), size_t(m.h0), hipMemcpyHostToDevice, j->stream));
    return 0;
}

int hevce_job_set_device_image(hevce_job *j, int i, const void *d_img) {
    if (!j || i < 0 || i >= j->n) return fail("hevce_job_set_device_image: bad arguments");
    j->im[i].ext_img = d_img; j->im[i].ext_stride = 0; j->desc_dirty = true;
    return 0;
}

/* picture i = the h0 x w0 window at `d_origin` of a larger picture resident in HBM, read in place */
int hevce_job_set_device_window(hevce_job *j, int i, const void *d_origin, long stride) {
    if (!j || i < 0 || i >= j->n || !d_origin || stride < j->im[i].w0) return fail("hevce_job_set_device_window: bad arguments");
    j->im[i].ext_img = d_origin; j->im[i].ext_stride = stride; j->desc_dirty = true;
    return 0;
}

int hevce_job_set_stream(hevce_job *j, void *s) {
    if (!j) return fail("hevce_job_set_stream: null job");
    j->stream = s ? (hipStream_t)s : j->own_stream;
    return 0;
}
/* run on exactly this stream handle; 0 is the NULL (legacy default) stream -- torch's default current stream has handle 0 --
   and not "the job's own stream" as in hevce_job_set_stream */
int hevce_job_use_stream(hevce_job *j, void *s) {
    if (!j) return fail("hevce_job_use_stream: null job");
    j->stream = (hipStream_t)s;
    return 0;
}

int hevce_job_set_flags(hevce_job *j, int flags) {
    if (!j) return fail("hevce_job_set_flags: null job");
    j->flags = flags; j->desc_dirty = true;
    return 0;
}
int hevce_job_cluster(hevce_job *j, int *fallbacks) {
    if (!j) return fail("hevce_job_cluster: null job");
    if (fallbacks) *fallbacks = j->cluster_fallbacks;
    return j->cluster_now;                                       /* of the last launch (before any: what the device has room for) */
}
int hevce_job_stats(hevce_job *j, int i, int *out2) {
    if (!j || i < 0 || i >= j->n || !out2) return fail("hevce_job_stats: bad arguments");
    HIPCHK(hipSetDevice(j->device));
    HIPCHK(hipMemcpy(out2, j->d_slab + j->im[i].off_stats, 8, hipMemcpyDeviceToHost));
    return 0;
}

int hevce_job_launch(hevce_job *j) {
    if (!j) return fail("hevce_job_launch: null job");
    HIPCHK(hipSetDevice(j->device));
    if (j->launched) return fail("hevce_job_launch: the previous launch has not been finished");
    /* the cluster size of this launch: what fits beside this process's other launches in flight (their units come back at their finish) */
    {
        if (j->cu_claimed) { cu_release(j->device, j->cu_claimed); j->cu_claimed = 0; }
        const int want = (j->cluster > 1 && !(j->flags & 2) && !j->single_once) ? j->cluster : 1;
        j->single_once = false;
        const int now = cu_claim(j->device, want > 1 ? (j->n + 7) & ~7 : j->n, want, j->cus, &j->cu_claimed);
        if (now != j->cluster_now) { j->cluster_now = now; j->desc_dirty = true; }
    }
    if (j->desc_dirty) {
        size_t dump_off = 0;
        for (int i = 0; i < j->n; i++) {
            const Image &m = j->im[i];
            HevceImageDesc &d = j->h_desc[i];
            d.dump = j->d_dump ? j->d_dump + dump_off : nullptr;
            dump_off += size_t(m.nctu) * 21 * HEVCE_TRIALS * 16;
            d.img = m.ext_img ? (const uint8_t *)m.ext_img : j->d_slab + m.off_img;
            d.rcon = j->d_slab + m.off_rcon;
            d.records = j->d_slab + m.off_rec;
            d.ctu_state = (int32_t *)(j->d_slab + m.off_state);
            d.scratch = j->d_slab + m.off_scratch;
            d.qlut = qlut_device(j->device, m.q);
            if (!d.qlut) return fail("the quantiser tables could not be placed on device %d", j->device);
            d.src_h = m.h0; d.src_w = m.w0; d.src_stride = (m.ext_img && m.ext_stride) ? m.ext_stride : m.w0; d.H = m.H; d.W = m.W; d.qpd6 = m.q; d.status = 0;
            d.trace = j->h_trace ? j->h_trace + 1024 * i : nullptr;
            d.payload = j->device_emit ? j->d_slab + m.off_payload : nullptr;
            d.stats = (int32_t *)(j->d_slab + m.off_stats);
            d.wr_tok = j->d_slab + m.off_wrtok;
            d.cluster = j->cluster_now > 1 ? (HevceCluster *)(j->d_slab + m.off_cluster) : nullptr;
            d.flags = j->flags; d.pad_ = 0;
            d.payload_len = j->device_emit ? (int32_t *)(j->d_slab + m.off_plen) : nullptr;
        }
        HIPCHK(hipMemcpyAsync(j->d_desc, j->h_desc.data(), sizeof(HevceImageDesc) * j->n, hipMemcpyHostToDevice, j->stream));
        j->desc_dirty = false;
    }
    const int cluster = j->cluster_now;
    if (cluster > 1) {
        /* (the pinned block belongs to the job and is only written here, behind the previous launch's finish: the copies below may still
           be reading it when this call returns) */
        j->h_cluster_init->size = cluster;
        for (int i = 0; i < j->n; i++) HIPCHK(hipMemcpyAsync(j->d_slab + j->im[i].off_cluster, j->h_cluster_init, sizeof(HevceCluster), hipMemcpyHostToDevice, j->stream));
    }
    HIPCHK(hipEventRecord(j->ev0, j->stream));
    if (hevce_launch_analyse(j->d_desc, j->n, cluster, j->stream) != 0) return fail("kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIPCHK(hipEventRecord(j->ev1, j->stream));
    /* what the host needs in every case: the length of each picture's slice data.  The decision records and per-CTU coder
       registers (2176 + 32 bytes per CTU) only travel when somebody wants them: the host-writer mode, hevce_job_download_records */
    for (int i = 0; i < j->n; i++) {
        const Image &m = j->im[i];
        if (j->device_emit) HIPCHK(hipMemcpyAsync(j->h_slab + m.hoff_plen, j->d_slab + m.off_plen, 4, hipMemcpyDeviceToHost, j->stream));
        else {
            HIPCHK(hipMemcpyAsync(j->h_slab + m.hoff_rec, j->d_slab + m.off_rec, size_t(m.nctu) * HEVCE_REC_BYTES, hipMemcpyDeviceToHost, j->stream));
            HIPCHK(hipMemcpyAsync(j->h_slab + m.hoff_state, j->d_slab + m.off_state, size_t(m.nctu) * 32, hipMemcpyDeviceToHost, j->stream));
        }
    }
    j->records_fetched = !j->device_emit;
    HIPCHK(hipEventRecord(j->ev2, j->stream));
    j->launched = true;
    return 0;
}

int hevce_job_finish(hevce_job *j) {
    if (!j || !j->launched) return fail("hevce_job_finish: nothing launched");
    HIPCHK(hipSetDevice(j->device));
    HIPCHK(hipEventSynchronize(j->ev2));
    j->launched = false;
    if (j->cu_claimed) { cu_release(j->device, j->cu_claimed); j->cu_claimed = 0; }
    HIPCHK(hipEventElapsedTime(&j->kernel_ms, j->ev0, j->ev1));
    HIPCHK(hipEventElapsedTime(&j->d2h_ms, j->ev1, j->ev2));
    const auto t0 = std::chrono::steady_clock::now();
    if (j->device_emit) {
        /* the kernel has written the slice data; the host puts the VPS/SPS/PPS/slice header in front  [ref :625-691].  The streams
           live back to back in one pinned slab (grown on demand, kept for the next run): header written in place, slice data
           fetched by one asynchronous copy per picture of exactly its length, one wait for all of them */
        size_t need = 0;
        for (int i = 0; i < j->n; i++) {
            Image &m = j->im[i];
            const int plen = *(const int32_t *)(j->h_slab + m.hoff_plen);
            if (plen == -1 && j->cluster_now > 1) {
                /* the picture's workgroups were not resident side by side (the kernel's alive check, milliseconds) or a hand-off
                   between them timed out: once more, one workgroup per picture */
                fprintf(stderr, "hevce_mi355x: a cluster was not placed or a hand-off timed out (image %d); encoding again without clusters\n", i);
                j->single_once = true; j->cluster_fallbacks++;   /* this launch only: the device may have room again for the next */
                if (hevce_job_launch(j) != 0) return -1;
                return hevce_job_finish(j);
            }
            if (plen <= 0 || size_t(plen) > size_t(m.nctu) * 3200 + 64) return fail("image %d: the device emitted %d bytes", i, plen);
            need += 128 + size_t(plen);
        }
        if (need > j->h_streams_bytes) {
            if (j->h_streams) (void)hipHostFree(j->h_streams);
            j->h_streams = nullptr; j->h_streams_bytes = 0;
            const size_t cap = need + need / 4 + 4096;
            HIPCHK(hipHostMalloc((void **)&j->h_streams, cap, hipHostMallocDefault));
            j->h_streams_bytes = cap;
        }
        size_t off = 0;
        for (int i = 0; i < j->n; i++) {
            Image &m = j->im[i];
            const int plen = *(const int32_t *)(j->h_slab + m.hoff_plen);
            uint8_t *dst = j->h_streams + off;
            const int hdr = hevce_write_header(dst, m.q, m.H, m.W);
            HIPCHK(hipMemcpyAsync(dst + hdr, j->d_slab + m.off_payload, size_t(plen), hipMemcpyDeviceToHost, j->stream));
            m.stream_ptr = dst; m.length = hdr + plen;
            off += 128 + size_t(plen);
        }
        HIPCHK(hipStreamSynchronize(j->stream));
        j->writer_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    /* host stream writer: images are independent, one thread each (bounded by the host's cores) */
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *e = getenv("HEVCE_WRITER_THREADS")) hw = (unsigned)atoi(e);
    int nthreads = (int)(hw ? hw : 1);
    if (nthreads > j->n) nthreads = j->n;
    std::atomic<int> next(0);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= j->n) return;
            Image &m = j->im[i];
            m.stream.resize((size_t)hevce_stream_bound(m.h0, m.w0));
            m.length = hevce_write_stream(m.stream.data(), j->h_slab + m.hoff_rec, m.H, m.W, m.q);
            m.stream.resize((size_t)m.length);
            m.stream_ptr = m.stream.data();
        }
    };
    if (nthreads <= 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work);
        for (auto &t : th) t.join();
    }
    j->writer_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

/* test hook: TU pipeline of all 35 modes for one n x n block (org 32x32 bytes, nbr 33x65 bytes); outputs are
   35 x 1024 levels (scan order), 35 x 1024 reconstruction bytes (stride n), 35 x 4 ints (sse, last, cbf, cgsig low) */
int hevce_test_tu(const unsigned char *org, const unsigned char *nbr, int n, int qpd6, int aL, int aLB, int aA, int aAR,
                  short *lev, unsigned char *rec, int *info) {
    if (hevce_device_count() <= 0) return fail("no HIP device available");
    unsigned char *d = nullptr;
    const size_t o_org = 0, o_nbr = 1024, o_lev = 4096, o_rec = o_lev + 35 * 2048, o_info = o_rec + 35 * 1024, total = o_info + 35 * 16;
    HIPCHK(hipMalloc((void **)&d, total));
    HIPCHK(hipMemset(d, 0, total));
    HIPCHK(hipMemcpy(d + o_org, org, 1024, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d + o_nbr, nbr, 33 * 65, hipMemcpyHostToDevice));
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (hevce_launch_test_tu(d + o_org, d + o_nbr, n, qpd6, aL, aLB, aA, aAR, (short *)(d + o_lev), d + o_rec, (int *)(d + o_info), qlut_device(dev, qpd6), nullptr) != 0)
        return fail("test kernel launch failed");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(lev, d + o_lev, 35 * 2048, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rec, d + o_rec, 35 * 1024, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(info, d + o_info, 35 * 16, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    return 0;
}
/* test hook: TU-split chains of all 35 modes for one n x n CU; outputs: levels 35 x 1024 (four scan-ordered TUs of
   (n/2)^2 each), reconstruction 35 x 1024 (stride n), info 35 x 16 ints (sse, last[4], cbf[4], cgsig_lo[4]) */
int hevce_test_chain(const unsigned char *org, const unsigned char *nbr, int n, int qpd6, int aL, int aLB, int aA, int aAR,
                     short *lev, unsigned char *rec, int *info) {
    if (hevce_device_count() <= 0) return fail("no HIP device available");
    unsigned char *d = nullptr;
    const size_t o_nbr = 1024, o_scr = 4096, o_info = o_scr + HEVCE_SCRATCH_BYTES, total = o_info + 35 * 64;
    HIPCHK(hipMalloc((void **)&d, total));
    HIPCHK(hipMemset(d, 0, total));
    HIPCHK(hipMemcpy(d, org, 1024, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d + o_nbr, nbr, 33 * 65, hipMemcpyHostToDevice));
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (hevce_launch_test_chain(d, d + o_nbr, n, qpd6, aL, aLB, aA, aAR, d + o_scr, (int *)(d + o_info), qlut_device(dev, qpd6), nullptr) != 0)
        return fail("test kernel launch failed");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(lev, d + o_scr + 35 * 2048, 35 * 2048, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(rec, d + o_scr + HEVCE_TRIALS * 2048 + 35 * 1024, 35 * 1024, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(info, d + o_info, 35 * 64, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    return 0;
}
/* diagnostics: per-CU (sse, cost) of every trial of image 0..n (HEVCE_DUMP=1); returns ints copied */
long hevce_job_dump(hevce_job *j, int *out, long max_ints) {
    if (!j || !j->d_dump) return -1;
    long n = (long)j->dump_ints < max_ints ? (long)j->dump_ints : max_ints;
    if (hipMemcpy(out, j->d_dump, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
/* diagnostics: 1 when the launched work has completed, 0 while it is running */
int hevce_job_poll(hevce_job *j) {
    if (!j || !j->launched) return 1;
    return hipEventQuery(j->ev2) == hipSuccess ? 1 : 0;
}
/* diagnostics: copy image i's first 256 trace ints (HEVCE_TRACE builds + HEVCE_TRACE=1 in the environment) */
int hevce_job_trace(hevce_job *j, int i, int *out) {
    if (!j || !j->h_trace || i < 0 || i >= j->n) return -1;
    memcpy(out, j->h_trace + 1024 * i, sizeof(int32_t) * 256);
    return 0;
}

int hevce_job_run(hevce_job *j) {
    if (hevce_job_launch(j) != 0) return -1;
    return hevce_job_finish(j);
}

const unsigned char *hevce_job_stream(hevce_job *j, int i, int *length) {
    if (!j || i < 0 || i >= j->n) { fail("hevce_job_stream: bad arguments"); return nullptr; }
    if (length) *length = j->im[i].length;
    return j->im[i].stream_ptr;
}

int hevce_job_padded_size(hevce_job *j, int i, int *ysz, int *xsz) {
    if (!j || i < 0 || i >= j->n) return fail("hevce_job_padded_size: bad arguments");
    if (ysz) *ysz = j->im[i].H;
    if (xsz) *xsz = j->im[i].W;
    return 0;
}

int hevce_job_download_rcon(hevce_job *j, int i, unsigned char *dst) {
    if (!j || i < 0 || i >= j->n) return fail("hevce_job_download_rcon: bad arguments");
    HIPCHK(hipSetDevice(j->device));
    const Image &m = j->im[i];
    HIPCHK(hipMemcpy(dst, j->d_slab + m.off_rcon, size_t(m.H) * m.W, hipMemcpyDeviceToHost));
    return 0;
}

void *hevce_job_device_rcon(hevce_job *j, int i) { return (j && i >= 0 && i < j->n) ? j->d_slab + j->im[i].off_rcon : nullptr; }
const void *hevce_job_device_payload(hevce_job *j, int i, int *length) {
    if (!j || i < 0 || i >= j->n || !j->device_emit || j->launched || !j->im[i].stream_ptr) return nullptr;
    const int plen = *(const int32_t *)(j->h_slab + j->im[i].hoff_plen);
    if (plen <= 0) return nullptr;
    if (length) *length = plen;
    return j->d_slab + j->im[i].off_payload;
}
void *hevce_job_device_records(hevce_job *j, int i) { return (j && i >= 0 && i < j->n) ? j->d_slab + j->im[i].off_rec : nullptr; }

int hevce_job_download_records(hevce_job *j, int i, unsigned char *records, int *ctu_state) {
    if (!j || i < 0 || i >= j->n) return fail("hevce_job_download_records: bad arguments");
    const Image &m = j->im[i];
    if (!j->records_fetched) {                                   /* fetched on demand: the default path never needs them on the host */
        HIPCHK(hipSetDevice(j->device));
        HIPCHK(hipMemcpy(records, j->d_slab + m.off_rec, size_t(m.nctu) * HEVCE_REC_BYTES, hipMemcpyDeviceToHost));
        if (ctu_state) HIPCHK(hipMemcpy(ctu_state, j->d_slab + m.off_state, size_t(m.nctu) * 32, hipMemcpyDeviceToHost));
        return 0;
    }
    memcpy(records, j->h_slab + m.hoff_rec, size_t(m.nctu) * HEVCE_REC_BYTES);
    if (ctu_state) memcpy(ctu_state, j->h_slab + m.hoff_state, size_t(m.nctu) * 32);
    return 0;
}

int hevce_job_timings(hevce_job *j, float *kernel_ms, float *d2h_ms, float *writer_ms) {
    if (!j) return fail("hevce_job_timings: null job");
    if (kernel_ms) *kernel_ms = j->kernel_ms;
    if (d2h_ms) *d2h_ms = j->d2h_ms;
    if (writer_ms) *writer_ms = j->writer_ms;
    return 0;
}

/* ---- the reference's entry point and its batch form --------------------------------------------------- */
int HEVCImageEncoderBatch(int n, unsigned char *const *pbuffers, const unsigned char *const *imgs,
                          unsigned char *const *img_rcons, int *yszs, int *xszs, const int *qpd6s, int *lengths) {
    int device = 0;
    if (const char *e = getenv("HEVCE_DEVICE")) device = atoi(e);
    hevce_job *j = hevce_job_create(device, n, yszs, xszs, qpd6s);
    if (!j) return -1;
    int rc = 0;
    for (int i = 0; i < n && rc == 0; i++) rc = hevce_job_upload(j, i, imgs[i]);
    if (rc == 0) rc = hevce_job_run(j);
    for (int i = 0; i < n && rc == 0; i++) {
        const Image &m = j->im[i];
        memcpy(pbuffers[i], m.stream_ptr, (size_t)m.length);
        if (lengths) lengths[i] = m.length;
        rc = hevce_job_download_rcon(j, i, img_rcons[i]);
        yszs[i] = m.H; xszs[i] = m.W;
    }
    hevce_job_destroy(j);
    return rc;
}

/* one picture as independent tile-pictures, each with its own header and stream, all in one launch.
   Tiles in raster order; tile (r, c) is the window rows [r*tile_ysz, min(ysz, (r+1)*tile_ysz)), columns likewise. */
int HEVCImageEncoderTiles(const unsigned char *img, int ysz, int xsz, int tile_ysz, int tile_xsz, int qpd6,
                          unsigned char *const *pbuffers, int *lengths, unsigned char *const *img_rcons) {
    if (!img || ysz <= 0 || xsz <= 0 || tile_ysz <= 0 || tile_xsz <= 0 || tile_ysz > HEVCE_MAXDIM || tile_xsz > HEVCE_MAXDIM || !pbuffers)
        return fail("HEVCImageEncoderTiles: bad arguments (tile sizes must be 1..%d)", HEVCE_MAXDIM);
    const int rows = (ysz + tile_ysz - 1) / tile_ysz, cols = (xsz + tile_xsz - 1) / tile_xsz;
    if ((long)rows * cols > (1 << 20)) return fail("HEVCImageEncoderTiles: more than 2^20 tiles");
    const int n = rows * cols;
    std::vector<int> hs(n), ws(n), qs(n, qpd6);
    for (int t = 0; t < n; t++) {
        const int r = t / cols, c = t % cols;
        hs[t] = ysz - r * tile_ysz < tile_ysz ? ysz - r * tile_ysz : tile_ysz;
        ws[t] = xsz - c * tile_xsz < tile_xsz ? xsz - c * tile_xsz : tile_xsz;
    }
    int device = 0;
    if (const char *e = getenv("HEVCE_DEVICE")) device = atoi(e);
    hevce_job *j = hevce_job_create(device, n, hs.data(), ws.data(), qs.data());
    if (!j) return -1;
    int rc = 0;
    for (int t = 0; t < n && rc == 0; t++)
        rc = hevce_job_upload_window(j, t, img + (size_t)(t / cols) * tile_ysz * (size_t)xsz + (size_t)(t % cols) * tile_xsz, xsz);
    if (rc == 0) rc = hevce_job_run(j);
    for (int t = 0; t < n && rc == 0; t++) {
        const Image &m = j->im[t];
        memcpy(pbuffers[t], m.stream_ptr, (size_t)m.length);
        if (lengths) lengths[t] = m.length;
        if (img_rcons && img_rcons[t]) rc = hevce_job_download_rcon(j, t, img_rcons[t]);
    }
    hevce_job_destroy(j);
    return rc == 0 ? n : -1;
}

/* The reference's entry point  [ref src/HEVCe.h:5-12, HEVCe.c:1570-1647].  The reference keeps no state between calls and is
   re-entrant (everything on the caller's stack, HEVCe.c:1578-1593); here every calling THREAD keeps one job context alive between
   calls -- device slab, pinned buffers, stream and events of the last (size, qpd6) it encoded -- so that a caller looping over
   pictures of one size pays for the allocations once.  Threads never share a context: concurrent callers are independent. */
__attribute__((visibility("default"))) int HEVCImageEncoder(unsigned char *pbuffer, const unsigned char *img, unsigned char *img_rcon,
                     int *ysz, int *xsz, const int qpd6) {
    if (!pbuffer || !img || !img_rcon || !ysz || !xsz) return fail("HEVCImageEncoder: null argument");
    int device = 0;
    if (const char *e = getenv("HEVCE_DEVICE")) device = atoi(e);
    CachedJob &c = g_cached;
    if (!c.j || c.device != device || c.h != *ysz || c.w != *xsz || c.q != qpd6) {
        if (c.j) hevce_job_destroy(c.j);
        c.j = hevce_job_create(device, 1, ysz, xsz, &qpd6);
        if (!c.j) return -1;
        c.device = device; c.h = *ysz; c.w = *xsz; c.q = qpd6;
        static std::once_flag once;
        std::call_once(once, [] { atexit([] { g_alive.store(false); }); });   /* after HIP's own registration: runs before its teardown */
    }
    hevce_job *j = c.j;
    if (hevce_job_upload(j, 0, img) != 0 || hevce_job_run(j) != 0) { hevce_job_destroy(j); c.j = nullptr; return -1; }
    const Image &m = j->im[0];
    memcpy(pbuffer, m.stream_ptr, (size_t)m.length);
    if (hevce_job_download_rcon(j, 0, img_rcon) != 0) return -1;
    *ysz = m.H; *xsz = m.W;
    return m.length;
}

#pragma GCC visibility pop
}  // extern "C"
