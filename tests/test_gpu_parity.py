"""GPU (-m gpu): parity of the HIP analysis path, called through the C ABI, against the oracle and the golden vectors.
Bit-exact bar: .h265 bytes, reconstruction, decision records and per-CTU coder registers are all integers."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLD, load_case, small_cases

pytestmark = pytest.mark.gpu


def _first_diff(hevce, orc, img, q, job_rec, job_st):
    """locate the first CTU whose decisions or coder state differ (debugging aid for failures)"""
    rec, rcon, st = orc.analyse(img, q)
    for k in range(rec.shape[0]):
        if not np.array_equal(job_rec[k], rec[k]) or not np.array_equal(job_st[k], st[k]):
            return f"first differing CTU {k}: state gpu={job_st[k].tolist()} oracle={st[k].tolist()} " \
                   f"record bytes differing={int((job_rec[k] != rec[k]).sum())}"
    return "records and states equal"


def test_library_loaded_and_device_present(hevce):
    assert os.path.exists(hevce.LIB_PATH)
    assert hevce.device_count() >= 1


@pytest.mark.parametrize("path", small_cases(), ids=lambda p: os.path.basename(p)[6:-4])
def test_small_golden_through_c_abi(hevce, path):
    c = load_case(path)
    stream, rcon = hevce.HEVCImageEncoder(c["img"], c["q"])
    assert np.array_equal(rcon, c["rcon"])
    assert stream == c["stream"]


def test_records_and_states_match_oracle(hevce, orc):
    import hevce_workload as wl
    img = wl.gen_g2(96, 160, 16, seed=31337)
    for q in range(5):
        with hevce.Job([img.shape], q) as job:
            job.upload(0, img)
            job.run()
            rec, st = job.records(0)
            orec, orcon, ost = orc.analyse(img, q)
            assert np.array_equal(st, ost), _first_diff(hevce, orc, img, q, rec, st)
            assert np.array_equal(rec, orec), _first_diff(hevce, orc, img, q, rec, st)
            assert np.array_equal(job.rcon(0), orcon)
            assert job.stream(0) == orc.write_stream(orec, orcon.shape[0], orcon.shape[1], q)


@pytest.mark.parametrize("single", [0, 2], ids=["cluster", "one-workgroup"])
@pytest.mark.parametrize("flags", [0, 1, 2 << 8, 3 << 16, (1 << 8) | (1 << 16), (5 << 8) | (7 << 16)],
                         ids=["fast", "exact", "nxn2", "wr3", "every-ctu-both", "nxn5-wr7"])
def test_fast_analysis_and_exact_reanalysis_paths(hevce, orc, flags, single):
    """The kernel decides with range-only trial coders and re-analyses poisoned CTUs with the exact ones.  The test
    hooks poison CTUs through both detection paths; streams, reconstruction, records and coder registers stay the oracle's and
    the writer's coder never disagrees with the analysis' (stats[1] == 0)."""
    import hevce_workload as wl
    imgs = [wl.gen_g2(96, 160, 16, seed=4242), wl.gen_g2(70, 100, 4, seed=7), (np.random.RandomState(3).randint(0, 2, (64, 96)) * 255).astype(np.uint8)]
    qs = [2, 0, 4]
    with hevce.Job([im.shape for im in imgs], qs) as job:
        job.set_flags(flags | single)                            # bit 1: one workgroup per picture; otherwise two (cluster mode) as they fit the device
        for i, im in enumerate(imgs):
            job.upload(i, im)
        job.run()
        for i, im in enumerate(imgs):
            s, r = orc.encode(im, qs[i])
            orec, orcon, ost = orc.analyse(im, qs[i])
            rec, st = job.records(i)
            assert job.stream(i) == s and np.array_equal(job.rcon(i), r), i
            assert np.array_equal(rec, orec) and np.array_equal(st, ost), i
            redo, bad = job.stats(i)
            assert bad == 0 and (redo > 0) == (flags > 1), (i, redo, bad)


def test_batch_of_ragged_images(hevce, orc):
    """independent images of different sizes and qpd6 in ONE launch (one workgroup each)"""
    import hevce_workload as wl
    rng = np.random.RandomState(5)
    imgs = [wl.gen_g2(64, 64, 16, seed=1), wl.gen_g2(33, 97, 4, seed=2), rng.randint(0, 256, (70, 40)).astype(np.uint8),
            np.full((32, 32), 200, np.uint8), wl.gen_g2(128, 64, 16, seed=3), rng.randint(0, 256, (1, 1)).astype(np.uint8),
            wl.gen_g2(64, 96, 8, seed=4), wl.gen_g2(50, 50, 16, seed=6)]
    qs = [2, 0, 4, 1, 3, 2, 2, 2]
    with hevce.Job([im.shape for im in imgs], qs) as job:
        for i, im in enumerate(imgs):
            job.upload(i, im)
        job.run()
        for i, im in enumerate(imgs):
            s, r = orc.encode(im, qs[i])
            assert job.stream(i) == s, i
            assert np.array_equal(job.rcon(i), r), i


def test_kodak01_qpd6_2_known_answer(hevce, manifest):
    """BASELINE.json configs[0]: testimage/01.pgm at qpd6=2 -> 163 557 bytes, sha256 e19966a4..."""
    import hevce_workload as wl
    img = wl.read_pgm(os.path.join(GOLD, "kodak", "01.pgm"))
    stream, rcon = hevce.HEVCImageEncoder(img, 2)
    r = next(r for r in manifest["kodak"] if r["name"] == "01" and r["qpd6"] == 2)
    assert len(stream) == 163557 == r["len"]
    assert hashlib.sha256(stream).hexdigest() == r["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == r["rcon_sha256"]


def test_kodak_sweep_against_manifest(hevce, manifest):
    """BASELINE.json configs[4]: 24 Kodak x qpd6 0..4 in one batch launch; qpd6=4 equals the reference's own
    testimage_out/*.h265 (manifest sha256), qpd6 0..3 equal the reference build's output"""
    import hevce_workload as wl
    cases = manifest["kodak"]
    imgs = {n: wl.read_pgm(os.path.join(GOLD, "kodak", n + ".pgm")) for n in sorted({r["name"] for r in cases})}
    with hevce.Job([imgs[r["name"]].shape for r in cases], [r["qpd6"] for r in cases]) as job:
        for i, r in enumerate(cases):
            job.upload(i, imgs[r["name"]])
        job.run()
        bad = []
        for i, r in enumerate(cases):
            s = job.stream(i)
            if len(s) != r["len"] or hashlib.sha256(s).hexdigest() != r["sha256"] \
               or hashlib.sha256(job.rcon(i).tobytes()).hexdigest() != r["rcon_sha256"]:
                bad.append((r["name"], r["qpd6"]))
        assert not bad, bad


def test_full_size_known_answers(hevce):
    """BASELINE.json configs[1] and configs[3] at full size against hashes from the REAL reference (oracle/gen_golden_fullsize.py,
    tests/golden/manifest_fullsize.json): the bench picture (3840x2160 G2-16, 8160 CTUs) and four 1024x1024 tile-pictures of the
    16384x16384 picture, stream and reconstruction."""
    import json
    import hevce_workload as wl
    with open(os.path.join(os.path.dirname(__file__), "golden", "manifest_fullsize.json")) as f:
        man = {r["name"]: r for r in json.load(f)}
    r = man["bench_4k"]
    img = wl.gen_g2(2160, 3840, 16)
    with hevce.Job([img.shape], 2) as job:
        job.upload(0, img)
        job.run()
        s, rc = job.stream(0), job.rcon(0)
    assert len(s) == r["len"] and hashlib.sha256(s).hexdigest() == r["sha256"]
    assert hashlib.sha256(rc.tobytes()).hexdigest() == r["rcon_sha256"]
    tiles = [(int(k.split("_")[1][1:]), int(k.split("_")[2][1:]), v) for k, v in man.items() if k.startswith("tile_")]
    assert len(tiles) >= 2
    bands = {rr: wl.gen_g2(1024, 16384, 16, seed=0x1234ABCD + 1000 + rr) for rr in sorted({t[0] for t in tiles})}
    with hevce.Job([(1024, 1024)] * len(tiles), 2) as job:
        for j, (rr, cc, _) in enumerate(tiles):
            job.upload_window(j, bands[rr], 0, cc * 1024)
        job.run()
        for j, (rr, cc, v) in enumerate(tiles):
            s = job.stream(j)
            assert len(s) == v["len"] and hashlib.sha256(s).hexdigest() == v["sha256"], (rr, cc)
            assert hashlib.sha256(job.rcon(j).tobytes()).hexdigest() == v["rcon_sha256"], (rr, cc)


def test_full_size_properties(hevce, orc):
    """BASELINE.json configs[1] shape (3840x2160, pads to 3840x2176 = 8160 CTUs), checked through size-independent
    properties (the CPU oracle would need minutes here): stream == host replay of the device records, every CTU has a
    complete quad-tree, reconstruction error bounded, bit counter strictly increasing, and the first CTU row -- which
    depends on nothing below it -- equals the oracle's analysis of the top 32 rows alone."""
    import hevce_workload as wl
    img = wl.gen_g2(2160, 3840, 16)
    with hevce.Job([img.shape], 2) as job:
        job.upload(0, img)
        job.run()
        s1, r1 = job.stream(0), job.rcon(0)
        rec, st = job.records(0)
    assert r1.shape == (2176, 3840) and rec.shape[0] == 8160
    assert hevce.write_stream(rec, 2176, 3840, 2) == s1
    info = rec[:, 2048 + 64:].reshape(-1, 8, 8)
    assert (info & 3).max() <= 2
    err = r1[:2160].astype(np.int32) - img.astype(np.int32)
    assert np.abs(err).mean() < 4.0                                 # qpd6=2 is near-lossless (about 47 dB on Kodak)
    assert np.all(st[:, 7] > 0) and np.all(np.diff(st[:, 7]) > 0)   # bit counter strictly increases CTU by CTU
    orec, orcon, ost = orc.analyse(img[:32, :1024], 2)              # 32 CTUs of the first CTU row (narrower picture:
    assert np.array_equal(rec[:31], orec[:31])                      #  identical up to its last CTU, whose right
    assert np.array_equal(st[:31], ost[:31])                        #  neighbourhood differs)
    assert np.array_equal(r1[:32, :31 * 32], orcon[:, :31 * 32])


def test_determinism_and_reuse_of_a_job(hevce):
    """the same job object run twice gives identical bytes (no state leaks between launches)"""
    import hevce_workload as wl
    img = wl.gen_g2(160, 224, 16, seed=77)
    with hevce.Job([img.shape], 2) as job:
        job.upload(0, img)
        job.run()
        s1, r1 = job.stream(0), job.rcon(0)
        job.run()
        assert job.stream(0) == s1 and np.array_equal(job.rcon(0), r1)


def test_upper_left_crop_is_prefix_consistent(hevce, orc):
    """the first CTU row of a wide image depends only on that row: encoding the top 32 rows alone gives the same
    decisions for row 0 (no hidden cross-row state leaks)"""
    import hevce_workload as wl
    img = wl.gen_g2(96, 256, 16, seed=9)
    with hevce.Job([img.shape, (32, 256)], 2) as job:
        job.upload(0, img)
        job.upload(1, img[:32])
        job.run()
        ra, sa = job.records(0)
        rb, sb = job.records(1)
    assert np.array_equal(ra[:7], rb[:7]) and np.array_equal(sa[:7], sb[:7])   # the last CTU differs only by its terminate bin


@pytest.mark.gpu
def test_device_emitted_stream_equals_host_writer_replay():
    """the kernel emits the slice data itself (default); HEVCE_HOST_WRITER=1 makes the host replay the decision
    records instead.  Both must give the same bytes, and hevce_write_stream() on the downloaded records as well."""
    import subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r); import hevce_pkg, hevce_workload as wl; h = hevce_pkg.load();"
            "imgs = [wl.gen_g2(200, 328, 16, seed=5), wl.gen_g2(97, 65, 4, seed=6), wl.gen_g2(32, 32, 48, seed=7)];"
            "out = h.encode_batch(imgs, [2, 0, 4]);"
            "print(' '.join(hashlib.sha256(s).hexdigest() for s, _ in out))" % ROOT)
    env = dict(os.environ); env.pop("HEVCE_HOST_WRITER", None)
    b = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    env["HEVCE_HOST_WRITER"] = "1"
    c = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert b.returncode == 0 and c.returncode == 0, (b.stderr, c.stderr)
    assert b.stdout.strip() == c.stdout.strip() and len(b.stdout.split()) == 3


def test_1080p_pictures_in_one_launch_match_reference_hashes(hevce):
    """BASELINE.json configs[2] picture size: eight 1920x1080 G2-16 pictures (the seeds bench.py / tools/dist_batch.py use) in ONE
    launch against SHA-256s produced by the REAL reference in the build container (oracle/gen_golden_1080p.py)."""
    import json
    import hevce_workload as wl
    path = os.path.join(GOLD, "manifest_1080p.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/manifest_1080p.json not generated")
    gold = json.load(open(path))
    imgs = [wl.gen_g2(g["h"], g["w"], 16, seed=g["seed"]) for g in gold]
    with hevce.Job([im.shape for im in imgs], 2) as job:
        for i, im in enumerate(imgs):
            job.upload(i, im)
        job.run()
        for i, g in enumerate(gold):
            s = job.stream(i)
            assert len(s) == g["len"] and hashlib.sha256(s).hexdigest() == g["sha256"], i
            assert hashlib.sha256(job.rcon(i).tobytes()).hexdigest() == g["rcon_sha256"], i
            assert job.stats(i)[1] == 0


def test_concurrent_callers_of_the_drop_in_symbol(hevce):
    """The reference's entry point is re-entrant (all state on the caller's stack, HEVCe.c:1578-1593): four threads call
    HEVCImageEncoder at the same time on different pictures (each thread several times: its cached job context is reused, and
    rebuilt when the size changes) -- every result is the golden one."""
    import threading
    cases = [load_case(p) for p in small_cases() if any(k in p for k in ("g2_16_64x96_q2", "lcg_100x70_q2", "noise_64x64_q1", "checker_64x64_q2"))]
    assert len(cases) == 4
    errors = []

    def worker(k):
        try:
            for rep in range(3):
                c = cases[(k + rep) % 4] if rep == 2 else cases[k]      # the last call changes the picture size
                stream, rcon = hevce.HEVCImageEncoder(c["img"], c["q"])
                if stream != c["stream"] or not np.array_equal(rcon, c["rcon"]):
                    errors.append((k, rep))
        except Exception as e:                                          # noqa: BLE001
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []


def test_job_is_ordered_behind_the_callers_stream():
    """hevce_job_use_stream(0) = the NULL stream torch's default stream maps to: a picture produced asynchronously on torch's
    current stream (no host synchronisation in between) is what the kernel reads.  In a process of its own, where torch brings up
    its HIP runtime first (as in bench.py): this process has already initialised the library's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "gpu_stream_order.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "stream order ok" in p.stdout, p.stdout + p.stderr


def test_cluster_falls_back_when_its_workers_do_not_answer(hevce, orc):
    """cluster mode needs a picture's workgroups resident side by side; when a hand-off times out the picture is flagged and the host
    encodes the launch again with one workgroup per picture.  Flag bit 2 makes the workers leave at once (and shortens the wait)."""
    import hevce_workload as wl
    img = wl.gen_g2(64, 96, 16, seed=77)
    want, wrc = orc.encode(img, 2)
    with hevce.Job([img.shape], 2) as job:
        if job.cluster()[0] < 2:
            pytest.skip("the device is too small for clusters")
        job.set_flags(4)
        job.upload(0, img)
        import time
        t0 = time.time()
        job.run()
        dt = time.time() - t0
        assert job.stream(0) == want and np.array_equal(job.rcon(0), wrc)
        assert job.cluster() == (1, 1)
        # the chain workgroup's alive wait (hv_cl_alive; 1 ms under this flag, 20 ms otherwise) gives the cluster up before the first
        # CTU: the whole detour -- failed launch, second launch without clusters -- is a matter of milliseconds, not of a hand-off timeout
        assert dt < 1.0, dt
        job.set_flags(0)                                         # the fall-back was for that launch only
        job.run()
        assert job.stream(0) == want and job.cluster()[0] >= 2 and job.cluster()[1] == 1


def test_slice_data_on_the_device_and_gather_job_streams():
    """hevce_job_device_payload + distributed.gather_job_streams on the device (tests/tools/gpu_gather_job.py, a process of its own
    because torch has to bring up its HIP runtime before the library does)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "gpu_gather_job.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "gather from the device ok" in p.stdout, p.stdout + p.stderr


def test_cluster_beside_a_batch_that_fills_the_device():
    """a cluster's workgroups must be resident together.  tests/tools/gpu_saturate.py launches a batch that fills every compute unit
    (and keeps refilling them) and, while it runs, a single picture: with the process-wide compute-unit accounting the picture takes
    no cluster; with the accounting switched off (HEVCE_CU_BUDGET=0) it asks for one, and the kernel's alive check either sees the
    whole cluster placed or gives it up within milliseconds.  Either way: right bytes, no multi-second stall."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for budget in ("1", "0"):
        env = dict(os.environ, HEVCE_CU_BUDGET=budget)
        p = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "gpu_saturate.py")], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        r = json.loads(p.stdout.strip().splitlines()[-1])
        assert r["single_ok"] and r["batch_ok"], r
        if budget == "1":
            assert r["single_cluster"] == 1 and r["single_fallbacks"] == 0, r
        assert r["single_extra_s"] < 2.0, r                      # beyond the batch's own duration


@pytest.mark.parametrize("flags", [0, 16], ids=["light", "heavy"])
def test_cluster_hand_offs_light_and_heavy(hevce, orc, flags):
    """hand-offs between a cluster's workgroups: LIGHT when they share an XCD (no L2 maintenance), HEAVY -- agent-scope release / acquire
    -- otherwise (hevce_dev_base.h hv_publish_fence); flag bit 4 forces HEAVY everywhere.  Same bytes either way, no healed mismatches."""
    import hevce_workload as wl
    img = wl.gen_g2(160, 224, 16, seed=4242)
    want, wrc = orc.encode(img, 2)
    with hevce.Job([img.shape], 2) as job:
        if job.cluster()[0] < 2:
            pytest.skip("the device is too small for clusters")
        job.set_flags(flags)
        job.upload(0, img)
        for _ in range(2):
            job.run()
            assert job.stream(0) == want and np.array_equal(job.rcon(0), wrc)
            assert job.stats(0)[1] == 0 and job.cluster()[1] == 0
