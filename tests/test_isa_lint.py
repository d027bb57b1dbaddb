"""CPU: the build-time check of the device assembly (tools/isa_lint.py) -- VGPR spill code in front of an EXEC restore in a
join block's prologue, the register-allocator placement that miscompiled the function-level test kernel in round 2
(experiments/README.md, round 3)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

BAD = """
kern:
	s_and_saveexec_b64 s[42:43], s[44:45]
	s_cbranch_execz .LBB0_3
.LBB0_2:
	v_add_u32_e32 v1, v2, v3
.LBB0_3:
	v_accvgpr_write_b32 a81, v127
	s_movk_i32 s6, 0x49
	s_or_b64 exec, exec, s[42:43]
	v_accvgpr_read_b32 v127, a81
	s_endpgm
"""
GOOD = BAD.replace("\tv_accvgpr_write_b32 a81, v127\n\ts_movk_i32 s6, 0x49\n\ts_or_b64 exec, exec, s[42:43]\n",
                   "\ts_or_b64 exec, exec, s[42:43]\n\ts_movk_i32 s6, 0x49\n\tv_accvgpr_write_b32 a81, v127\n")
SCRATCH = BAD.replace("v_accvgpr_write_b32 a81, v127", "scratch_store_dword off, v40, s32 offset:4 ; 4-byte Folded Spill")
LONG_BLOCK = BAD.replace("\ts_movk_i32 s6, 0x49\n", "\tv_mul_i32_i24_e32 v5, v6, v7\n")   # not a prologue any more: a narrowed block that restores later


def _lint(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return isa_lint.lint(str(p))


def test_lint_flags_spills_in_front_of_an_exec_restore(tmp_path):
    assert len(_lint(tmp_path, BAD)) == 1
    assert len(_lint(tmp_path, SCRATCH)) == 1
    assert _lint(tmp_path, GOOD) == []
    assert _lint(tmp_path, LONG_BLOCK) == []


def test_the_library_s_device_assembly_is_clean():
    """`make lint` (part of `make all`, so of __graft_entry__.build()) leaves the checked assembly next to the library; it must be
    newer than the kernel sources and free of the pattern."""
    asm = os.path.join(ROOT, "hevc-image-encoder-lite_amd", "lib", "hevce_kernel.gfx950.s")
    csrc = os.path.join(ROOT, "hevc-image-encoder-lite_amd", "csrc")
    if not os.path.exists(asm) or any(os.path.getmtime(os.path.join(csrc, f)) > os.path.getmtime(asm) for f in [f for f in os.listdir(csrc) if f.startswith("hevce_dev") or f == "hevce_kernel.hip"]):
        subprocess.check_call(["make", "-C", csrc, "-s", "lint"])
    assert isa_lint.lint(asm) == []
