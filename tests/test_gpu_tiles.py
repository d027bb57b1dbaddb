"""GPU (-m gpu): SURVEY.md 8(d) config 4 / 8(e) -- one picture as independent tile-pictures.

The reference crops at 8192 x 8192 (reference src/HEVCe.c:1581-1582) and cannot split a picture, so parity for a tiled
picture is defined tile by tile: every tile's stream and reconstruction equal what the oracle produces for that window
stored as a picture of its own (own header, own coder, own padding)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tile_pictures_match_oracle_tile_by_tile(hevce, orc):
    """3 x 3 grid with ragged right / bottom tiles (200 = 96+96+8 rows, 330 = 160+160+10 columns)"""
    import hevce_workload as wl
    img = wl.gen_g2(200, 330, 16, seed=4242)
    tiles = hevce.encode_tiles(img, 2, 96, 160)
    assert [(y, x) for y, x, _, _ in tiles] == [(y, x) for y in (0, 96, 192) for x in (0, 160, 320)]
    for y, x, stream, rcon in tiles:
        crop = np.ascontiguousarray(img[y:y + 96, x:x + 160])
        s, r = orc.encode(crop, 2)
        assert stream == s, (y, x)
        assert np.array_equal(rcon, r), (y, x)


def test_window_of_a_resident_picture_is_read_in_place(hevce):
    """hevce_job_set_device_window: tiles of a picture that already lives in HBM, without a copy.  The resident picture
    here is the reconstruction another job left on the device (hevce_job_device_rcon, stride = its padded width); the
    result must equal the host-side tile entry point fed with the downloaded copy, for several qpd6."""
    import hevce_workload as wl
    src = wl.gen_g2(120, 190, 16, seed=99)
    with hevce.Job([src.shape], 1) as first:
        first.upload(0, src)
        first.run()
        pic = first.rcon(0)                                   # 128 x 192, what the windows below read in place
        H, W = pic.shape
        grid = hevce.tile_grid(H, W, 64, 96)
        for q in (0, 2, 4):
            want = hevce.encode_tiles(pic, q, 64, 96)
            with hevce.Job([(th, tw) for _, _, th, tw in grid], q) as job:
                for t, (y, x, th, tw) in enumerate(grid):
                    job.set_device_window(t, first.device_rcon_ptr(0) + y * W + x, W, keepalive=first)
                job.run()
                for t in range(len(grid)):
                    assert job.stream(t) == want[t][2], (q, t)
                    assert np.array_equal(job.rcon(t), want[t][3]), (q, t)


def test_picture_wider_than_the_reference_limit(hevce, orc):
    """a 40 x 8300 picture: the plain entry point crops to 8192 columns like the reference; two tiles cover all of it"""
    import hevce_workload as wl
    img = wl.gen_g2(40, 8300, 8, seed=5)
    plain, rcon = hevce.HEVCImageEncoder(img, 3)
    assert rcon.shape == (64, 8192)
    tiles = hevce.encode_tiles(img, 3, 40, 4150)
    assert len(tiles) == 2 and tiles[1][1] == 4150 and tiles[1][3].shape == (64, 4160)
    for y, x, stream, r in tiles:
        s, ro = orc.encode(np.ascontiguousarray(img[:, x:x + 4150]), 3)
        assert stream == s and np.array_equal(r, ro)
    s, ro = orc.encode(img, 3)                      # the oracle crops like the reference
    assert plain == s and np.array_equal(rcon, ro)


def test_tile_entry_point_rejects_bad_arguments(hevce):
    img = np.zeros((64, 64), np.uint8)
    for th, tw in ((0, 32), (32, 0), (8193, 32), (32, 9000)):
        with pytest.raises(hevce.HevceError):
            hevce.encode_tiles(img, 2, th, tw)
