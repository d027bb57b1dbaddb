"""CPU: the oracle restatement against the committed golden vectors (generated from the real reference by
oracle/gen_golden.py) and against the reference's own goldens (testimage_out/*.h265, qpd6=4, via the manifest)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLD, load_case, small_cases


@pytest.mark.parametrize("path", small_cases(), ids=lambda p: os.path.basename(p)[6:-4])
def test_small_golden(orc, path):
    c = load_case(path)
    stream, rcon = orc.encode(c["img"], c["q"])
    assert stream == c["stream"]
    assert np.array_equal(rcon, c["rcon"])


def test_records_replay_equals_direct(orc):
    c = load_case(os.path.join(GOLD, "small_lcg_100x70_q2.npz"))
    rec, rcon, st = orc.analyse(c["img"], 2)
    assert np.array_equal(rec, c["records"]) and np.array_equal(st, c["ctu_state"])
    assert orc.write_stream(rec, rcon.shape[0], rcon.shape[1], 2) == c["stream"]
    assert rcon.shape == (96, 128)                       # 100x70 pads to 128x96
    assert len(c["stream"]) == 2468 and hashlib.sha256(c["stream"]).hexdigest().startswith("8bb5b1fa93dcd1b1")


def test_manifest_is_clean(manifest):
    """every Kodak x qpd6 case matched the real reference when the fixtures were generated; the reference build
    reproduced all 24 testimage_out goldens (qpd6=4)"""
    assert len(manifest["kodak"]) == 120
    assert all(r["oracle_matches_ref"] for r in manifest["kodak"])
    q4 = [r for r in manifest["kodak"] if r["qpd6"] == 4]
    assert len(q4) == 24 and all(r["ref_matches_testimage_out"] for r in q4)
    r = next(r for r in manifest["kodak"] if r["name"] == "01" and r["qpd6"] == 2)
    assert r["len"] == 163557 and r["sha256"] == "e19966a4df0aee17368883e07c206aaeeeb3bf5bc454e981ce77429b82183dc0"


@pytest.mark.slow
def test_kodak23_q4_against_manifest(orc, manifest):
    """one full Kodak picture end to end on the CPU (about 5 s): equals testimage_out/23.h265 by sha256"""
    import hevce_workload as wl
    img = wl.read_pgm(os.path.join(GOLD, "kodak", "23.pgm"))
    stream, rcon = orc.encode(img, 4)
    r = next(r for r in manifest["kodak"] if r["name"] == "23" and r["qpd6"] == 4)
    assert len(stream) == r["len"] and hashlib.sha256(stream).hexdigest() == r["sha256"]
    assert hashlib.sha256(rcon.tobytes()).hexdigest() == r["rcon_sha256"]


def test_generator_matches_sequential(orc):
    import hevce_workload as wl
    assert np.array_equal(wl.gen_g2(37, 53, 16, seed=0x1234ABCD), orc.gen_g2(37, 53, 16, seed=0x1234ABCD))
    assert np.array_equal(wl.gen_g2(8, 300, 4, seed=77), orc.gen_g2(8, 300, 4, seed=77))
