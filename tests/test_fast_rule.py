"""CPU: the product's FAST decision rule against the oracle proper.

The HIP kernel decides with bit counts that leave out the emulation-prevention bytes of the trial encodes and re-analyses, with
exact counts, the CTUs its stream writer / NxN verification flag (hevce_dev_syntax.h, hv_fengine_chunk).
oracle/hevce_oracle.c restates that rule (orc_analyse_fast_rule); here it must reproduce the oracle's records, reconstruction
and per-CTU coder registers on pictures where such bytes are frequent: constant, faint noise, soft blobs, shallow ramps, one edge.
tests/tools/fast_rule_sweep.py is the same check at 2 x 10^5 CTUs (profiles/r03_fast_rule_sweep.json)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def test_fast_rule_matches_the_oracle_on_flat_content():
    import fast_rule_sweep as frs
    res = frs.sweep(300, 1, 12345)
    assert res["mismatches"] == 0, res
    assert res["ctus"] >= 300
    assert res["events"] > 0          # the precondition does occur on this content: the rule is being exercised
    assert res["redone"] > 0          # ... and so do flagged CTUs


@pytest.mark.slow
def test_fast_rule_sweep_10k_ctus():
    import fast_rule_sweep as frs
    res = frs.sweep(10000, max(1, min(8, os.cpu_count() or 1)), 777)
    assert res["mismatches"] == 0, res
