"""Randomised parity sweep (tests/tools/gpu_stress.py): pictures of random size / content class / qpd6 in one batch, repeated
launches, every stream and reconstruction against the oracle.  The kernel's wavefronts hand work to each other through
LDS flags and idle-time riding, so its internal timing varies from launch to launch; this covers what fixed goldens cannot."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,batch", [(11, 0), (12, 0), (13, 24)], ids=["one-workgroup-11", "one-workgroup-12", "clusters-13"])
def test_random_pictures_match_oracle_over_repeated_launches(seed, batch):
    """72 pictures in one launch run with one workgroup per picture; in launches of 24 they run as clusters of seven workgroups per
    picture, whose hand-offs (quad messages, NxN reports, answers, double-buffered scratch) are timing dependent as well.  The
    consistency counter must stay 0: a healed state mismatch would leave the streams right and still be a bug."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import gpu_stress
    res = gpu_stress.sweep(72, seed, 3, batch)
    assert res["mismatches"] == 0 and res["healed_state_mismatches"] == 0, res
    if batch:
        assert res["workgroups_per_picture"] == [7], res
