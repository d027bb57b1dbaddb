"""hevc-image-encoder-lite_amd -- host-side mirror of the reference's encoder interface on top of the
MI355X-native C-ABI library (include/hevce_mi355x.h, built from csrc/ into lib/libhevce_mi355x.so).

The reference's operator API is one C function, ``HEVCImageEncoder(pbuffer, img, img_rcon, *ysz, *xsz, qpd6)``
(reference src/HEVCe.h:5-12).  ``HEVCImageEncoder`` below calls the drop-in symbol of the same name with the same
argument meaning; ``Job`` exposes the additive batch / device-resident path.  There is no CPU fallback: without
the HIP library or without a GPU these calls raise.

The directory name contains dashes, so import it through ``hevce_pkg.load()`` at the repository root.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("HEVCE_LIB") or os.path.join(_HERE, "lib", "libhevce_mi355x.so")   # HEVCE_LIB: diagnostics builds
HEADER = os.path.join(os.path.dirname(_HERE), "include", "hevce_mi355x.h")
RECORD_BYTES = 2176

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int)
_lib = None


class HevceError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP kernel + host shim for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def lib():
    """The C-ABI library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HevceError(f"{LIB_PATH} is missing: run hevc-image-encoder-lite_amd.build() "
                             "(make -C hevc-image-encoder-lite_amd/csrc); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.HEVCImageEncoder.restype = C.c_int
        L.HEVCImageEncoder.argtypes = [_u8p, _u8p, _u8p, _i32p, _i32p, C.c_int]
        L.HEVCImageEncoderBatch.restype = C.c_int
        L.hevce_stream_bound.restype = C.c_long
        L.hevce_stream_bound.argtypes = [C.c_int, C.c_int]
        L.hevce_job_create.restype = C.c_void_p
        L.hevce_job_create.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _i32p]
        L.hevce_job_destroy.argtypes = [C.c_void_p]
        L.hevce_job_upload.argtypes = [C.c_void_p, C.c_int, _u8p]
        L.hevce_job_set_device_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.hevce_job_upload_window.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
        L.hevce_job_set_device_window.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
        L.HEVCImageEncoderTiles.restype = C.c_int
        L.HEVCImageEncoderTiles.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_u8p), _i32p, C.POINTER(_u8p)]
        L.hevce_job_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.hevce_job_use_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.hevce_job_set_flags.argtypes = [C.c_void_p, C.c_int]
        L.hevce_job_stats.argtypes = [C.c_void_p, C.c_int, _i32p]
        L.hevce_job_cluster.argtypes = [C.c_void_p, _i32p]
        for f in ("hevce_job_run", "hevce_job_launch", "hevce_job_finish"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.hevce_job_stream.restype = _u8p
        L.hevce_job_stream.argtypes = [C.c_void_p, C.c_int, _i32p]
        L.hevce_job_padded_size.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p]
        L.hevce_job_download_rcon.argtypes = [C.c_void_p, C.c_int, _u8p]
        L.hevce_job_device_rcon.restype = C.c_void_p
        L.hevce_job_device_rcon.argtypes = [C.c_void_p, C.c_int]
        L.hevce_job_device_payload.restype = C.c_void_p
        L.hevce_job_device_payload.argtypes = [C.c_void_p, C.c_int, _i32p]
        L.hevce_job_device_records.restype = C.c_void_p
        L.hevce_job_device_records.argtypes = [C.c_void_p, C.c_int]
        L.hevce_job_download_records.argtypes = [C.c_void_p, C.c_int, _u8p, _i32p]
        L.hevce_job_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.hevce_job_dump.restype = C.c_long
        L.hevce_job_dump.argtypes = [C.c_void_p, _i32p, C.c_long]
        L.hevce_job_poll.argtypes = [C.c_void_p]
        L.hevce_job_trace.argtypes = [C.c_void_p, C.c_int, _i32p]
        L.hevce_write_stream.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        L.hevce_write_header.argtypes = [_u8p, C.c_int, C.c_int, C.c_int]
        L.hevce_last_error.restype = C.c_char_p
        L.hevce_version.restype = C.c_char_p
        _lib = L
    return _lib


def _p(a, t=_u8p):
    return a.ctypes.data_as(t)


def pad32(v):
    return (min(int(v), 8192) + 31) // 32 * 32


def device_count():
    return int(lib().hevce_device_count())


def _check(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise HevceError(f"{what} failed: {lib().hevce_last_error().decode()}")


def HEVCImageEncoder(img, qpd6):
    """Drop-in call: returns (stream bytes, padded reconstruction [yszn, xszn] uint8).

    Same contract as reference src/HEVCe.c:1570-1647: img is 8-bit luma [ysz, xsz]; the reconstruction has the
    padded size (multiples of 32); sizes above 8192 are cropped; qpd6 in 0..4."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    H, W = pad32(h), pad32(w)
    out = np.empty(int(lib().hevce_stream_bound(h, w)), np.uint8)
    rcon = np.empty((H, W), np.uint8)
    ys, xs = C.c_int(h), C.c_int(w)
    n = lib().HEVCImageEncoder(_p(out), _p(img), _p(rcon), C.byref(ys), C.byref(xs), int(qpd6))
    _check(n, "HEVCImageEncoder")
    assert (ys.value, xs.value) == (H, W)
    return out[:n].tobytes(), rcon


def write_stream(records, H, W, qpd6):
    """Host stream writer alone (no GPU needed): decision records -> .h265 bytes."""
    records = np.ascontiguousarray(records, np.uint8)
    out = np.empty(256 + (H // 32) * (W // 32) * 3200 + 16, np.uint8)
    n = lib().hevce_write_stream(_p(out), _p(records), H, W, int(qpd6))
    return out[:n].tobytes()


def write_header(qpd6, H, W):
    out = np.zeros(256, np.uint8)
    n = lib().hevce_write_header(_p(out), int(qpd6), H, W)
    return out[:n].tobytes()


class Job:
    """A batch of independent images resident in HBM: one workgroup per image (additive API)."""

    def __init__(self, sizes, qpd6, device=0):
        self.n = len(sizes)
        self.sizes = [(int(h), int(w)) for h, w in sizes]
        q = [int(qpd6)] * self.n if np.isscalar(qpd6) else [int(v) for v in qpd6]
        self.qpd6 = q
        ys = (C.c_int * self.n)(*[s[0] for s in self.sizes])
        xs = (C.c_int * self.n)(*[s[1] for s in self.sizes])
        qs = (C.c_int * self.n)(*q)
        self._h = lib().hevce_job_create(int(device), self.n, ys, xs, qs)
        if not self._h:
            raise HevceError(f"hevce_job_create failed: {lib().hevce_last_error().decode()}")
        self._keep = {}
        self._up = {}

    def close(self):
        if getattr(self, "_h", None):
            lib().hevce_job_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload(self, i, img):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == self.sizes[i]
        # the copy is asynchronous on the job's stream; a pinned source is read by DMA after this call returns (include/hevce_mi355x.h,
        # hevce_job_upload): keep the array until the picture is replaced or the job closed
        self._up[i] = img
        _check(lib().hevce_job_upload(self._h, i, _p(img)), "hevce_job_upload")

    def set_device_image(self, i, ptr, keepalive=None):
        """Use an image already in HBM (e.g. torch tensor .data_ptr())."""
        self._keep[i] = keepalive
        _check(lib().hevce_job_set_device_image(self._h, i, C.c_void_p(ptr)), "hevce_job_set_device_image")

    def upload_window(self, i, picture, y0, x0):
        """image i = the window of this job's size at (y0, x0) of a larger host picture (copied with its stride)."""
        picture = np.ascontiguousarray(picture, np.uint8)
        h, w = self.sizes[i]
        assert 0 <= y0 and 0 <= x0 and y0 + h <= picture.shape[0] and x0 + w <= picture.shape[1]
        origin = picture.ctypes.data + y0 * picture.shape[1] + x0
        self._up[i] = picture                                    # (see upload)
        _check(lib().hevce_job_upload_window(self._h, i, C.c_void_p(origin), picture.shape[1]), "hevce_job_upload_window")

    def set_device_window(self, i, origin_ptr, stride, keepalive=None):
        """image i = a window of a larger picture resident in HBM, read in place (origin pointer, row stride in bytes)."""
        self._keep[i] = keepalive
        _check(lib().hevce_job_set_device_window(self._h, i, C.c_void_p(origin_ptr), int(stride)), "hevce_job_set_device_window")

    def set_stream(self, hip_stream):
        _check(lib().hevce_job_set_stream(self._h, C.c_void_p(hip_stream)), "hevce_job_set_stream")

    def use_stream(self, hip_stream_handle):
        """run on exactly this stream handle; 0 = the NULL (legacy default) stream, e.g. torch.cuda.current_stream().cuda_stream"""
        _check(lib().hevce_job_use_stream(self._h, C.c_void_p(int(hip_stream_handle))), "hevce_job_use_stream")

    def run(self):
        _check(lib().hevce_job_run(self._h), "hevce_job_run")

    def set_flags(self, flags):
        """tests: bit 0 exact coders throughout; bits 8-15 / 16-23 forced re-analysis periods (include/hevce_mi355x.h)"""
        _check(lib().hevce_job_set_flags(self._h, int(flags)), "hevce_job_set_flags")

    def cluster(self):
        """(workgroups per picture of the next launch, launches repeated without clusters after a timed-out hand-off)"""
        fb = C.c_int(0)
        k = lib().hevce_job_cluster(self._h, C.byref(fb))
        return int(k), int(fb.value)

    def stats(self, i=0):
        """(CTUs analysed twice, coder consistency failures) of image i in the last run"""
        out = (C.c_int * 2)()
        _check(lib().hevce_job_stats(self._h, int(i), out), "hevce_job_stats")
        return int(out[0]), int(out[1])

    def launch(self):
        _check(lib().hevce_job_launch(self._h), "hevce_job_launch")

    def finish(self):
        _check(lib().hevce_job_finish(self._h), "hevce_job_finish")

    def dump(self, nctu):
        out = np.zeros(nctu * 21 * 72 * 16, np.int32)
        n = lib().hevce_job_dump(self._h, _p(out, _i32p), out.size)
        return None if n < 0 else out.reshape(nctu, 21, 72, 16)

    def poll(self):
        return bool(lib().hevce_job_poll(self._h))

    def trace(self, i=0):
        out = np.zeros(256, np.int32)
        if lib().hevce_job_trace(self._h, i, _p(out, _i32p)) != 0:
            return None
        return out

    def stream(self, i):
        n = C.c_int(0)
        p = lib().hevce_job_stream(self._h, i, C.byref(n))
        return C.string_at(p, n.value)

    def padded_size(self, i):
        return pad32(self.sizes[i][0]), pad32(self.sizes[i][1])

    def rcon(self, i):
        H, W = self.padded_size(i)
        out = np.empty((H, W), np.uint8)
        _check(lib().hevce_job_download_rcon(self._h, i, _p(out)), "hevce_job_download_rcon")
        return out

    def records(self, i):
        H, W = self.padded_size(i)
        nctu = (H // 32) * (W // 32)
        rec = np.empty((nctu, RECORD_BYTES), np.uint8)
        st = np.empty((nctu, 8), np.int32)
        _check(lib().hevce_job_download_records(self._h, i, _p(rec), _p(st, _i32p)), "hevce_job_download_records")
        return rec, st

    def device_rcon_ptr(self, i):
        return lib().hevce_job_device_rcon(self._h, i)

    def device_payload(self, i):
        """(device pointer, length) of image i's slice data in HBM -- the stream without its parameter-set header -- after a finished
        launch; (None, 0) when there is none"""
        n = C.c_int(0)
        p = lib().hevce_job_device_payload(self._h, int(i), C.byref(n))
        return (int(p), int(n.value)) if p else (None, 0)

    def timings(self):
        k, d, w = C.c_float(), C.c_float(), C.c_float()
        lib().hevce_job_timings(self._h, C.byref(k), C.byref(d), C.byref(w))
        return dict(kernel_ms=k.value, d2h_ms=d.value, writer_ms=w.value)


def encode_batch(imgs, qpd6, device=0):
    """Encode independent images in one launch; returns [(stream, rcon), ...]."""
    with Job([im.shape for im in imgs], qpd6, device) as job:
        for i, im in enumerate(imgs):
            job.upload(i, im)
        job.run()
        return [(job.stream(i), job.rcon(i)) for i in range(job.n)]


def tile_grid(h, w, tile_h, tile_w):
    """[(y0, x0, th, tw), ...] in the raster order HEVCImageEncoderTiles uses."""
    return [(y, x, min(tile_h, h - y), min(tile_w, w - x)) for y in range(0, h, tile_h) for x in range(0, w, tile_w)]


def encode_tiles(img, qpd6, tile_h, tile_w, want_rcon=True):
    """One picture as independent tile-pictures: returns [(y0, x0, stream, rcon-or-None), ...];
    every stream is what HEVCImageEncoder gives for that window stored as a picture of its own."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    if not (1 <= int(tile_h) <= 8192 and 1 <= int(tile_w) <= 8192):
        raise HevceError("encode_tiles: tile sizes must be 1..8192 (the reference's picture limit, HEVCe.c:1581-1582)")
    grid = tile_grid(h, w, tile_h, tile_w)
    outs = [np.empty(int(lib().hevce_stream_bound(th, tw)), np.uint8) for (_, _, th, tw) in grid]
    rcons = [np.empty((pad32(th), pad32(tw)), np.uint8) if want_rcon else None for (_, _, th, tw) in grid]
    lens = np.zeros(len(grid), np.int32)
    pb = (_u8p * len(grid))(*[_p(o) for o in outs])
    pr = (_u8p * len(grid))(*[_p(r) if r is not None else None for r in rcons])
    n = lib().HEVCImageEncoderTiles(_p(img), h, w, int(tile_h), int(tile_w), int(qpd6), pb, _p(lens, _i32p), pr)
    _check(n, "HEVCImageEncoderTiles")
    assert n == len(grid)
    return [(y, x, outs[t][:lens[t]].tobytes(), rcons[t]) for t, (y, x, _, _) in enumerate(grid)]
