"""CPU: the device analysis code (hevc-image-encoder-lite_amd/csrc/hevce_device.h) compiled for the fiber emulator in
tests/emu -- checks its control flow, decisions, reconstruction and per-CTU coder state against the golden vectors
without a GPU.  This is a debugging aid for the CPU suite; the parity gate proper is tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from conftest import load_case, small_cases

CASES = [p for p in small_cases() if any(k in p for k in ("lcg_100x70_q2", "g2_16_64x96_q0", "g2_16_64x96_q4", "noise_64x64_q1",
                                                           "white_33x65_q3", "black_40x40_q2", "noise_1x1_q2", "checker_64x64_q2",
                                                           "hstripes_48x80_q3", "g2_4_64x64_q2"))]


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[6:-4])
def test_emulated_kernel_matches_golden(path):
    import emu
    c = load_case(path)
    rec, rcon, st = emu.analyse(c["img"], c["q"])
    assert np.array_equal(rcon, c["rcon"])
    assert np.array_equal(st, c["ctu_state"])
    assert np.array_equal(rec, c["records"])
    # device-side byte emission: the golden stream minus its VPS/SPS/PPS/slice header
    stream = bytes(c["stream"])
    assert len(emu.analyse.payload) > 0 and stream.endswith(emu.analyse.payload)
    assert len(stream) - len(emu.analyse.payload) in range(70, 100)


@pytest.mark.parametrize("q", [0, 2, 4])
def test_emulated_kernel_on_black_and_white_noise(orc, q):
    """a picture of 0 / 255 samples only: the largest residuals, transform sums (the 16-bit staging of the butterflies relies on
    |coefficient| <= 32640) and level magnitudes 8-bit input can produce -- against the oracle, one 32x32 and 16x16 CU deep"""
    import emu
    img = (np.random.RandomState(40 + q).randint(0, 2, (40, 72)) * 255).astype(np.uint8)
    rec, rcon, st = emu.analyse(img, q)
    orec, orcon, ost = orc.analyse(img, q)
    assert np.array_equal(rcon, orcon) and np.array_equal(st, ost) and np.array_equal(rec, orec)
    stream, _ = orc.encode(img, q)
    assert stream.endswith(emu.analyse.payload) and len(emu.analyse.payload) > 0


MODE_CASES = [p for p in small_cases() if any(k in p for k in ("lcg_100x70_q2", "g2_16_64x96_q0", "g2_16_64x96_q4", "checker_64x64_q2"))]


@pytest.mark.parametrize("flags", [1, 2 << 8, 3 << 16, (1 << 8) | (1 << 16), (3 << 8) | (2 << 16)],
                         ids=["exact", "nxn2", "wr3", "every-ctu-both", "nxn3-wr2"])
@pytest.mark.parametrize("path", MODE_CASES, ids=lambda p: os.path.basename(p)[6:-4])
def test_emulated_kernel_reanalysis_paths(path, flags):
    """FAST analysis + EXACT re-analysis of poisoned CTUs (hevce_dev_picture.h, hv_analyse_image): the test hooks poison CTUs
    the way an NxN winner / the writer would; the output must stay the golden one and the two coders must never disagree."""
    import emu
    c = load_case(path)
    rec, rcon, st = emu.analyse(c["img"], c["q"], flags=flags)
    assert np.array_equal(rcon, c["rcon"]) and np.array_equal(st, c["ctu_state"]) and np.array_equal(rec, c["records"])
    assert bytes(c["stream"]).endswith(emu.analyse.payload) and len(emu.analyse.payload) > 0
    redo, bad = emu.analyse.stats
    assert bad == 0
    assert (redo > 0) == (flags != 1)


_CL_FLAGS = [(0, "fast"), (2 << 8, "nxn2"), (3 << 16, "wr3"), ((3 << 16) | (4 << 24), "wr3-back5")]
# every picture x every re-analysis path on the two sizes that ship as defaults (4: round 2's, 7: this round's); the sizes in between
# (each adds one role) on one picture: the CPU suite has to stay within minutes, the GPU suite runs the clusters on real pictures
_CL_CASES = [pytest.param(p, f, n, id=f"{os.path.basename(p)[6:-4]}-{fid}-{n}")
             for n in (2, 3, 4, 5, 6, 7) for (f, fid) in _CL_FLAGS for p in (MODE_CASES if n in (4, 7) else MODE_CASES[:1])]


@pytest.mark.parametrize("path,flags,nwg", _CL_CASES)
def test_emulated_cluster_of_workgroups(path, flags, nwg):
    """cluster mode (hevce_dev_workers.h hv_parent_worker): the chain workgroup and the parent workgroup(s) of one picture as 2 / 3 / 4 x 512
    fibers exchanging messages through the HevceCluster mailboxes; also with re-analysed CTUs, which the chain workgroup does alone."""
    import emu
    c = load_case(path)
    rec, rcon, st = emu.analyse(c["img"], c["q"], flags=flags, cluster=nwg)
    assert np.array_equal(rcon, c["rcon"]) and np.array_equal(st, c["ctu_state"]) and np.array_equal(rec, c["records"])
    assert bytes(c["stream"]).endswith(emu.analyse.payload) and len(emu.analyse.payload) > 0
    assert emu.analyse.stats[1] == 0


@pytest.mark.parametrize("nwg", [2, 4, 7])
def test_emulated_cluster_is_given_up_when_a_workgroup_is_not_resident(nwg):
    """flag bit 2: the workers leave before they report alive, as workgroups the device has not placed would; the chain workgroup's
    alive wait (hv_cl_alive) flags the cluster before the first CTU and reports the picture as not encoded (payload length -1) --
    the host then encodes it without clusters (tests/test_gpu_parity.py covers that side)."""
    import emu
    c = load_case(MODE_CASES[0])
    emu.analyse(c["img"], c["q"], flags=4, cluster=nwg)
    assert emu.analyse.cluster_error and emu.analyse.payload_len == -1 and emu.analyse.stats[1] > 0
