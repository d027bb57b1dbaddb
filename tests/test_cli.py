"""the reference's command line (argv contract, PGM dialect, stdout report, exit status) on top of the
MI355X library.  The checker is the REAL reference CLI compiled from /root/reference/src into oracle/_ref/HEVCe_ref by
oracle/Makefile (test infrastructure; it travels to the GPU box as a built binary).  Without a GPU only the parts that
precede the encode can be compared: usage, argument echo, every PGM load failure, the accepted one-sample-short file."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "hevc-image-encoder-lite_amd", "bin", "HEVCe")
REF = os.path.join(ROOT, "oracle", "_ref", "HEVCe_ref")

need_bins = pytest.mark.skipif(not (os.path.exists(CLI) and os.path.exists(REF)),
                               reason="bin/HEVCe or oracle/_ref/HEVCe_ref not built (make -C csrc all; make -C oracle)")


def run(exe, args, cwd):
    p = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    return p.returncode, p.stdout.decode("latin-1")


def same_argv0(out, exe):
    return out.replace(exe, "HEVCe")


def pgm(w, h, maxval=255, sep=b"\n", data=None, magic=b"P5", header=None):
    if data is None:
        data = bytes((7 * x + 13 * y) & 255 for y in range(h) for x in range(w))
    head = header if header is not None else magic + b"\n%d %d\n%d" % (w, h, maxval) + sep
    return head + data


BAD_CASES = {
    "missing_file": None,
    "bad_magic_P2": pgm(8, 8, magic=b"P2"),
    "bad_magic_first": pgm(8, 8, magic=b"Q5"),
    "maxval_256": pgm(8, 8, maxval=256),
    "maxval_65535": pgm(8, 8, maxval=65535),
    "separator_letter": pgm(8, 8, sep=b"x"),
    "separator_missing_eof": b"P5\n8 8\n255",
    "no_height": b"P5\n8 \n",
    "garbage_width": b"P5\nab 8\n255\n" + bytes(64),
    "short_by_two": pgm(8, 8)[:-2],
    "short_by_many": pgm(8, 8)[:-40],
    "empty": b"",
    "only_P": b"P",
}


@need_bins
def test_usage_text_and_status(tmp_path):
    for args in ([], ["only_one_name"], ["3"], ["2", "4"]):
        rc_a, out_a = run(CLI, args, str(tmp_path))
        rc_b, out_b = run(REF, args, str(tmp_path))
        assert rc_a == rc_b == 255
        assert same_argv0(out_a, CLI) == same_argv0(out_b, REF)


@need_bins
@pytest.mark.parametrize("name", sorted(BAD_CASES))
def test_pgm_load_failures_match_reference(tmp_path, name):
    path = tmp_path / (name + ".pgm")
    if BAD_CASES[name] is not None:
        path.write_bytes(BAD_CASES[name])
    for extra in ([], ["1"], ["rec.pgm", "4"]):
        args = [path.name, "out.h265"] + extra
        rc_a, out_a = run(CLI, args, str(tmp_path))
        rc_b, out_b = run(REF, args, str(tmp_path))
        assert (rc_a, out_a) == (rc_b, out_b), name
        assert rc_a == 255 and out_a.endswith("open %s failed\n" % path.name)
        assert not (tmp_path / "out.h265").exists()


@need_bins
def test_argument_echo_and_accepted_dialects_up_to_the_encode(tmp_path):
    """files the reference accepts: every separator byte, leading white space / signs in the numbers, maxval < 255,
    one sample short.  Without a GPU the product stops at the encode (no CPU fallback); the report up to
    'compressing...' must already be identical."""
    cases = {
        "sep_space": pgm(8, 8, sep=b" "), "sep_tab": pgm(8, 8, sep=b"\t"), "sep_cr": pgm(8, 8, sep=b"\r"),
        "maxval_15": pgm(8, 8, maxval=15), "plus_signs": pgm(8, 8, header=b"P5 +8\t\t+8\r\n +255\n"),
        "short_by_one": pgm(8, 8)[:-1], "trailing_garbage": pgm(8, 8) + b"tail",
    }
    for name, blob in cases.items():
        (tmp_path / (name + ".pgm")).write_bytes(blob)
        for extra in ([], ["0"], ["r.pgm"], ["4", "r.pgm", "ignored", "ignored2"]):
            args = [name + ".pgm", "o.h265"] + extra
            rc_b, out_b = run(REF, args, str(tmp_path))
            assert rc_b == 0
            rc_a, out_a = run(CLI, args, str(tmp_path))
            head_b = out_b[:out_b.index("compressing...\n") + len("compressing...\n")]
            assert out_a.startswith(head_b), name
            if rc_a == 0:                      # a GPU is present: the whole report must match
                assert out_a == out_b


@pytest.mark.gpu
@need_bins
@pytest.mark.parametrize("case", ["kodak01_q2_rcon", "odd_100x70_default_q", "g2_256_q0_digit_first", "short_by_one_q4"])
def test_cli_end_to_end_matches_reference(tmp_path, case):
    import sys
    sys.path.insert(0, ROOT)
    import hevce_workload as wl
    if case == "kodak01_q2_rcon":
        src = open(os.path.join(ROOT, "tests", "golden", "kodak", "01.pgm"), "rb").read()
        args_of = lambda o, r: ["in.pgm", o, "2", r]
    elif case == "odd_100x70_default_q":
        s, px = 12345, bytearray()
        for y in range(70):
            for x in range(100):
                s = (s * 1103515245 + 12345) & 0x7fffffff
                px.append(min(255, max(0, ((2 * x + y) % 256) + ((s >> 16) & 15) - 8)))
        src = pgm(100, 70, data=bytes(px))
        args_of = lambda o, r: ["in.pgm", o, r]                 # default qpd6 = 3
    elif case == "g2_256_q0_digit_first":
        src = pgm(256, 256, data=wl.gen_g2(256, 256, 16).tobytes())
        args_of = lambda o, r: ["0", "in.pgm", o]               # the digit may come first; no reconstruction file
    else:
        src = pgm(64, 40, sep=b"\t", data=wl.gen_g2(40, 64, 4).tobytes())[:-1]
        args_of = lambda o, r: ["in.pgm", "4", o, r]
    (tmp_path / "in.pgm").write_bytes(src)
    rc_a, out_a = run(CLI, args_of("a.h265", "a.pgm"), str(tmp_path))
    rc_b, out_b = run(REF, args_of("b.h265", "b.pgm"), str(tmp_path))
    assert rc_a == rc_b == 0
    assert out_a.replace("a.h265", "b.h265").replace("a.pgm", "b.pgm") == out_b
    assert (tmp_path / "a.h265").read_bytes() == (tmp_path / "b.h265").read_bytes()
    assert (tmp_path / "a.pgm").exists() == (tmp_path / "b.pgm").exists()
    if (tmp_path / "a.pgm").exists():
        assert (tmp_path / "a.pgm").read_bytes() == (tmp_path / "b.pgm").read_bytes()


DIRCLI = os.path.join(ROOT, "hevc-image-encoder-lite_amd", "bin", "HEVCeDir")


@pytest.mark.skipif(not os.path.exists(DIRCLI), reason="bin/HEVCeDir not built")
def test_dir_frontend_usage_and_missing_directory(tmp_path):
    rc, out = run(DIRCLI, [], str(tmp_path))
    assert rc == 255 and out.startswith("Usage:")
    rc, out = run(DIRCLI, ["nowhere", "out"], str(tmp_path))
    assert rc == 255 and out == "open nowhere failed\n"


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(DIRCLI) and os.path.exists(CLI)), reason="front-ends not built")
def test_dir_frontend_outputs_equal_single_file_cli(tmp_path):
    """a directory of PGMs through the batch front-end (two small batches in flight) gives, file by file, what
    the single-file CLI gives; a file that does not load is reported and the others are still encoded."""
    import sys
    sys.path.insert(0, ROOT)
    import hevce_workload as wl
    src = tmp_path / "in"; src.mkdir()
    sizes = [(40, 64), (33, 65), (96, 100), (8, 8), (70, 100), (64, 64), (17, 130)]
    for i, (h, w) in enumerate(sizes):
        (src / ("p%02d.pgm" % i)).write_bytes(pgm(w, h, data=wl.gen_g2(h, w, 16, seed=100 + i).tobytes()))
    (src / "broken.pgm").write_bytes(b"P5\n8 8\n255\n" + bytes(10))
    (src / "ignored.txt").write_bytes(b"x")
    rc, out = run(DIRCLI, ["in", "out", "2", "--rcon", "--batch", "3"], str(tmp_path))
    assert rc == 255 and "open broken.pgm failed" in out and out.startswith("8 files")
    checker = REF if os.path.exists(REF) else CLI          # the real reference's CLI when it travelled with the snapshot (oracle/_ref)
    for i in range(len(sizes)):
        rc1, _ = run(checker, ["in/p%02d.pgm" % i, "one.h265", "2", "one.pgm"], str(tmp_path))
        assert rc1 == 0
        assert (tmp_path / "out" / ("p%02d.h265" % i)).read_bytes() == (tmp_path / "one.h265").read_bytes()
        assert (tmp_path / "out" / ("p%02d.pgm" % i)).read_bytes() == (tmp_path / "one.pgm").read_bytes()
    assert not (tmp_path / "out" / "broken.h265").exists()
