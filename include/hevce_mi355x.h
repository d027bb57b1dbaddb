/*
 * hevce_mi355x.h -- C ABI of the MI355X-native HEVC intra still-picture encoder (libhevce_mi355x.so).
 *
 * Drop-in boundary: HEVCImageEncoder() has the signature and semantics of the reference's only exported
 * function (reference src/HEVCe.h:5-12, defined src/HEVCe.c:1570-1647, called from src/HEVCeMain.c:197).
 * Everything else is additive: a batch entry point (one image = one serial chain = one workgroup, so
 * only batches fill a GPU) and a job object that keeps images, reconstructions and decision records
 * resident in HBM.  Plain pointers and sizes only; no C++ or torch types.
 *
 * There is NO CPU fallback: if no HIP device is usable the calls fail (return < 0, message on stderr and
 * in hevce_last_error()).
 */
#ifndef HEVCE_MI355X_H
#define HEVCE_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the reference interface (replaces reference src/HEVCe.h:5-12) ---------------------------------------
 * pbuffer  : out, HEVC stream (VPS+SPS+PPS+slice).  Caller-owned, no capacity argument (as in the reference);
 *            hevce_stream_bound() gives a safe size.
 * img      : in, 8-bit luma, *ysz rows of *xsz samples (stride *xsz).
 * img_rcon : out, reconstruction of the PADDED picture: ceil32(*ysz) rows of ceil32(*xsz) samples.
 * ysz,xsz  : in/out, replaced by the padded size; inputs above 8192 are cropped like the reference does.
 * qpd6     : 0..4 (QP = 6*qpd6+4).
 * returns  : stream length in bytes (> 0), or a negative value if the GPU path is unavailable / failed. */
int HEVCImageEncoder(unsigned char *pbuffer, const unsigned char *img, unsigned char *img_rcon,
                     int *ysz, int *xsz, const int qpd6);

/* ---- additive: n independent images in one launch (arrays of the six arguments + per-image lengths) ---- */
int HEVCImageEncoderBatch(int n, unsigned char *const *pbuffers, const unsigned char *const *imgs,
                          unsigned char *const *img_rcons, int *yszs, int *xszs, const int *qpd6s, int *lengths);

/* ---- additive: ONE picture as independent tile-pictures ---------------------
 * The reference crops pictures to 8192 x 8192 (HEVCe.c:1581-1582) and one picture is one serial chain; a larger
 * picture, or any picture that should occupy more than one compute unit, is cut into tiles that are encoded as
 * pictures of their own (own VPS/SPS/PPS/slice header, own coder) in one launch.  Tile t = r*cols + c (raster order,
 * rows = ceil(ysz/tile_ysz), cols = ceil(xsz/tile_xsz)) covers rows [r*tile_ysz, min(ysz, (r+1)*tile_ysz)) and the
 * columns likewise; pbuffers[t] / lengths[t] / img_rcons[t] are what HEVCImageEncoder() would produce for that
 * window stored as a picture (edge tiles are padded to multiples of 32 on their own).  img_rcons or single entries
 * of it may be NULL.  tile sizes: 1..8192.  Returns the number of tiles, or a negative value. */
int HEVCImageEncoderTiles(const unsigned char *img, int ysz, int xsz, int tile_ysz, int tile_xsz, int qpd6,
                          unsigned char *const *pbuffers, int *lengths, unsigned char *const *img_rcons);

/* upper bound of the stream size for a picture of the given (unpadded) size */
long hevce_stream_bound(int ysz, int xsz);

/* ---- device-resident jobs (bench / batch front-ends) -------------------------------------------------- */
typedef struct hevce_job hevce_job;

/* allocate device + pinned host buffers for n images of the given sizes on HIP device `device` */
hevce_job *hevce_job_create(int device, int n, const int *yszs, const int *xszs, const int *qpd6s);
void       hevce_job_destroy(hevce_job *job);
/* copy image i from host memory (stride = xszs[i]) into the job's HBM buffer.
   LIFETIME OF `img`: the copy is enqueued on the job's stream (hipMemcpyAsync), ordered before the kernel of the next launch.
     - pageable memory (malloc, numpy, ...): the HIP runtime has staged the bytes when the call returns -- the caller may reuse or
       free `img` at once;
     - PINNED memory (hipHostMalloc / hipHostRegister): the copy is a true asynchronous DMA -- `img` must stay valid and unchanged
       until hevce_job_finish() / hevce_job_run() of the launch that reads it has returned (or the caller has synchronised the
       job's stream itself).
   A caller that cannot tell which kind it holds keeps the buffer until the launch has finished. */
int  hevce_job_upload(hevce_job *job, int i, const unsigned char *img);
/* the same from a window of a larger host picture: yszs[i] rows of xszs[i] samples at origin, row stride `stride`; the lifetime rule of
   hevce_job_upload() applies to the whole window */
int  hevce_job_upload_window(hevce_job *job, int i, const unsigned char *origin, long stride);
/* use an image that already lives in HBM (e.g. a torch tensor's data_ptr); the job does not own it */
int  hevce_job_set_device_image(hevce_job *job, int i, const void *d_img);
/* image i is a window of a larger picture in HBM, read in place (no copy): origin pointer + row stride in bytes */
int  hevce_job_set_device_window(hevce_job *job, int i, const void *d_origin, long stride);
/* run on this hipStream_t (NULL = the job's own stream) */
int  hevce_job_set_stream(hevce_job *job, void *hip_stream);
/* run on exactly this hipStream_t handle, where 0 means the NULL (legacy default) stream -- e.g. torch's default current stream;
   uploads, the kernel and the result copies are then ordered behind whatever the caller enqueued there before */
int  hevce_job_use_stream(hevce_job *job, void *hip_stream);
/* analysis kernel over all images (it also emits each image's slice data) + D2H of the decision records and of the
   slice data behind a host-written VPS/SPS/PPS/slice header.  0 on success.
   HEVCE_HOST_WRITER=1 in the environment: the host replays the decision records into the stream instead (cross-check) */
int  hevce_job_run(hevce_job *job);
/* the two halves of hevce_job_run, for callers that overlap them */
int  hevce_job_launch(hevce_job *job);          /* enqueue kernel + async D2H of records */
int  hevce_job_finish(hevce_job *job);          /* wait, then assemble the streams (header + device-emitted slice data) */
/* results */
const unsigned char *hevce_job_stream(hevce_job *job, int i, int *length);
int  hevce_job_padded_size(hevce_job *job, int i, int *ysz, int *xsz);
int  hevce_job_download_rcon(hevce_job *job, int i, unsigned char *img_rcon);
void *hevce_job_device_rcon(hevce_job *job, int i);
/* image i's slice data where the kernel wrote it, in HBM (everything of the stream behind the VPS/SPS/PPS/slice header, which is a
   function of qpd6 and the padded size alone: hevce_write_header), and its length in bytes; valid from hevce_job_finish() to the job's
   next launch.  For consumers on the device -- e.g. a collective that gathers the streams of several GPUs without a host round trip
   (hevc-image-encoder-lite_amd/distributed.py gather_job_streams).  NULL before the first finished launch or in the host-writer mode. */
const void *hevce_job_device_payload(hevce_job *job, int i, int *length);
void *hevce_job_device_records(hevce_job *job, int i);
int  hevce_job_download_records(hevce_job *job, int i, unsigned char *records, int *ctu_state /* 8 ints per CTU, may be NULL */);
/* timings of the last run, milliseconds: kernel (HIP events on the launch stream), D2H wait, stream assembly on the host */
int  hevce_job_timings(hevce_job *job, float *kernel_ms, float *d2h_ms, float *writer_ms);

/* The kernel analyses with range-only ("FAST") trial coders and re-analyses with the exact ones the few CTUs whose bit counts
   an emulation-prevention byte could have changed; stats: out2[0] = CTUs of image i analysed twice in the last
   run, out2[1] = coder consistency failures (always 0).  flags (tests and comparisons; 0 for production):
     bit 0        exact coders throughout (the reference's bit counts by construction; ~2.4 x slower)
     bit 1        one workgroup per picture even when a cluster would fit
     bit 2        the cluster's worker workgroups leave at once without reporting alive: exercises the residency check and the fall-back
     bit 3        clusters of four and five: the round-2 path of the 8x8 workgroup (one block job per wavefront)
     bit 4        agent-scope (HEAVY) fences for every hand-off between a cluster's workgroups, also on one XCD
     bit 5        probe builds: report the second 8x8 workgroup's cycle counters instead of the first one's
     bit 6        clusters of five and more: the writer workgroup runs the shared-compute-unit coder (hv_wr_engine) instead of its own
     bits 8-15 / 16-23   force a re-analysis every k-th CTU through the NxN-verification / the stream-writer detection path
     bits 24-26   with bits 16-23 in clusters of five and more: how many CTUs the forced report lies back (roll-back depth) */
int  hevce_job_set_flags(hevce_job *job, int flags);
/* workgroups per picture of the last launch (before the first: what the device has room for; 1 = no clusters); *fallbacks (may be NULL) =
   launches that had to be repeated without clusters because a cluster was not placed (the kernel's alive check, milliseconds) or a
   hand-off timed out.  The size of a launch is what fits beside this process's other launches in flight. */
int  hevce_job_cluster(hevce_job *job, int *fallbacks);
int  hevce_job_stats(hevce_job *job, int i, int *out2);

/* diagnostics and function-level test hooks */
int  hevce_test_tu(const unsigned char *org32x32, const unsigned char *nbr33x65, int n, int qpd6, int aL, int aLB, int aA, int aAR,
                   short *lev35x1024, unsigned char *rec35x1024, int *info35x4);
int  hevce_job_poll(hevce_job *job);                      /* 1 = launched work finished, 0 = still running */
long hevce_job_dump(hevce_job *job, int *out, long max_ints); /* per-trial (sse, cost) of every CU evaluation (HEVCE_DUMP=1) */
int  hevce_job_trace(hevce_job *job, int i, int *out256);  /* checkpoints of a -DHEVCE_TRACE build (HEVCE_TRACE=1) */

int  hevce_test_chain(const unsigned char *org32x32, const unsigned char *nbr33x65, int n, int qpd6, int aL, int aLB, int aA, int aAR,
                      short *lev35x1024, unsigned char *rec35x1024, int *info35x16);

/* ---- host-side pieces, usable without a GPU ---------------------------------------------------------------- */
/* decision records (hevce_record_bytes() per CTU, raster order) -> complete .h265; returns its length
   (replaces reference src/HEVCe.c:625-691 header writer, :840-856 finish, and the byte-emitting half of :797-933) */
int  hevce_write_stream(unsigned char *out, const unsigned char *records, int ysz_padded, int xsz_padded, int qpd6);
int  hevce_write_header(unsigned char *out, int qpd6, int ysz_padded, int xsz_padded);
int  hevce_record_bytes(void);
int  hevce_device_count(void);                  /* HIP devices visible; <= 0 means the GPU path cannot run */
const char *hevce_last_error(void);
const char *hevce_version(void);

#ifdef __cplusplus
}
#endif
#endif
