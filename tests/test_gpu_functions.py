"""GPU (-m gpu): function-level parity of the block jobs -- prediction, transforms (partial butterflies), RDOQ, dequantiser,
inverse transforms, reconstruction, SSE for all 35 modes of one block -- against the oracle's primitives, through the library's
test hooks (hevce_test_tu / hevce_test_chain run the same hv_tu_job / hv_job_chain code as the kernel).  Uniformly random and
0/255-only samples: the largest residuals and transform sums 8-bit input can produce, which pictures never reach."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extreme", [False, True], ids=["random", "black-white"])
def test_block_jobs_all_sizes_modes_qpd6(hevce, orc, extreme):
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import gpu_tu_check
    assert gpu_tu_check.check(1 + extreme, extreme) == []


@pytest.mark.parametrize("extreme", [False, True], ids=["random", "black-white"])
def test_tu_split_chains_all_sizes_modes_qpd6(hevce, orc, extreme):
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import gpu_chain_check
    assert gpu_chain_check.check(3 + extreme, extreme) == []
