"""The conditioning bound the fp16 parity tests lean on (givepose_amd/rot_cond.py), checked against the oracle's own 6-D -> R map."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle.posenet_ref import rot6d_to_mat_ref  # noqa: E402
from givepose_amd.rot_cond import rot6d_amplification, rot_error_bound  # noqa: E402


def test_rot_error_bound_holds_for_random_and_badly_conditioned_logits():
    g = torch.Generator().manual_seed(5)
    n = 20000
    d6 = torch.randn(n, 6, generator=g, dtype=torch.float64)
    # a quarter of the crops badly conditioned: a2 nearly parallel to a1, or a1 short
    d6[: n // 8, 3:6] = d6[: n // 8, 0:3] * torch.randn(n // 8, 1, generator=g, dtype=torch.float64) + 0.03 * torch.randn(n // 8, 3, generator=g, dtype=torch.float64)
    d6[n // 8: n // 4, 0:3] *= 0.05
    for rel in (1e-4, 1e-3, 5e-3, 2e-2):
        delta = (torch.rand(n, 6, generator=g, dtype=torch.float64) * 2 - 1) * rel
        R0, R1 = rot6d_to_mat_ref(d6), rot6d_to_mat_ref(d6 + delta)
        dR = (R1 - R0).abs().reshape(n, -1).max(1).values
        b = rot_error_bound(d6, d6 + delta, floor=0.0)
        assert bool((dR <= b).all()), (rel, float((dR / b).max()))
        # and it is a bound worth having: within ~20 x of the actual error for the median crop
        fin = torch.isfinite(b)
        assert float((b[fin] / dR[fin].clamp_min(1e-12)).median()) < 25


def test_amplification_grows_with_bad_conditioning():
    good = torch.tensor([[1.0, 0, 0, 0, 1.0, 0]])
    bad = torch.tensor([[1.0, 0, 0, 1.0, 0.01, 0]])
    assert float(rot6d_amplification(bad)) > 50 * float(rot6d_amplification(good))
