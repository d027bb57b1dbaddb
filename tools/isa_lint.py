#!/usr/bin/env python3
"""Build-time check of the gfx950 assembly for one miscompilation pattern that bit this code base (experiments/README.md, round 3):

  a VGPR spill or reload placed at the top of a JOIN block *before* the instruction that restores the full EXEC mask
  (`s_or_b64 exec, exec, s[..]`).  The copy then only runs for the lanes that went through the divergent region; the lanes
  that skipped it reload a stale value later (under full EXEC).  LLVM places spill code "after the block prologue", and the
  prologue scan stops at the first ordinary instruction -- a rematerialised `s_movk_i32` ahead of the EXEC restore is enough.
  Seen in hevce_test_tu_kernel: the block's mode and store pointer were spilled to AGPRs under the narrowed mask; idle lanes then
  passed the `mode >= 0` guard with a stale mode and stored through a stale pointer (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION).

usage: isa_lint.py file.s [...]      exit status 1 if the pattern occurs (every occurrence is printed)"""
import re
import sys

SPILL = re.compile(r"\bv_accvgpr_(write|read)_b32\b|Folded Spill|Folded Reload")
EXEC_RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
PROLOGUE_OK = re.compile(r"^\s*(s_mov_b32|s_mov_b64|s_movk_i32|s_waitcnt|s_nop|v_readlane_b32|v_writelane_b32)\b(?!.*\bexec\b)")
LABEL = re.compile(r"^(\.LBB\w+|[A-Za-z_][\w$.]*):")


def lint(path):
    bad = []
    func = "?"
    block, insns = None, []

    def flush():
        # a join block: the EXEC restore sits in the block's prologue, i.e. everything in front of it is spill code or scalar
        # bookkeeping (constants, waits, SGPR spills to lanes).  VGPR spill code in front of it is the bug.
        for i, (no, text) in enumerate(insns):
            if EXEC_RESTORE.search(text):
                for no2, t2 in insns[:i]:
                    if SPILL.search(t2):
                        bad.append((path, func, block, no2, t2.strip(), no))
                return
            if not (SPILL.search(text) or PROLOGUE_OK.search(text)):
                return

    with open(path) as f:
        for no, line in enumerate(f, 1):
            code = line.split(";")[0].rstrip() if "Folded" not in line else line.rstrip()
            m = LABEL.match(line)
            if m:
                flush()
                block, insns = m.group(1), []
                if not block.startswith(".L"):
                    func = block
                continue
            if code.strip() and not code.lstrip().startswith((".", ";")):
                insns.append((no, code))
    flush()
    return bad


if __name__ == "__main__":
    allbad = []
    for p in sys.argv[1:]:
        allbad += lint(p)
    for path, func, block, no, text, no_restore in allbad:
        print(f"{path}:{no}: {func} {block}: `{text}` executes before the EXEC restore at line {no_restore}")
    print(f"isa_lint: {len(allbad)} spill(s) under a narrowed EXEC mask in {len(sys.argv) - 1} file(s)")
    sys.exit(1 if allbad else 0)
