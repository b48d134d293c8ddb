"""How far a perturbation of the rot6d logits can move R: the first-order amplification of the reference's 6-D -> matrix map
(network/pose_utils/rot_reps.py:34-55: x = a1 / |a1|, z = normalize(x cross a2), y = z cross x).

   |dx| <= |da1| / |a1|,   |dz| <= (|dx| |a2| + |da2|) / |a2 - (a2 . x) x|,   |dy| <= |dz| + |dx|

so with |da| <= sqrt(3) max|d6|:  ||dR||_F <= sqrt(3) * amp * max|d6|,  amp = 2 / |a1| + 2 (1 + |a2| / |a1|) / |a2_perp|.
The egocentric correction that follows (pose decode) multiplies by a rotation: entries of dR stay below ||dR||_F.

The fp16 mode's WORST crop error of R is this amplification times an ordinary logit error -- tests bound every crop by it instead
of giving the maximum a ceiling that the next batch seed's worst-conditioned crop breaks."""
import math

import torch


def rot6d_amplification(d6):
    """d6 (B, 6) -> amp (B,): ||dR||_F <= sqrt(3) * amp * max|delta d6| to first order."""
    d6 = d6.double()
    a1, a2 = d6[:, 0:3], d6[:, 3:6]
    n1 = a1.norm(dim=1)
    x = a1 / n1[:, None]
    perp = (a2 - (a2 * x).sum(1, keepdim=True) * x).norm(dim=1)
    return 2.0 / n1 + 2.0 * (1.0 + a2.norm(dim=1) / n1) / perp


def rot_error_bound(d6_ref, d6_got, slack=1.5, floor=1e-3, max_logit_err=None):
    """Per crop: the largest |dR| entry the logit error of that crop explains (first order x `slack`, plus `floor` for the fp32
    arithmetic of the map itself).  Where the perturbation is not small against the crop's conditioning (first-order term > 0.25)
    nothing can be said about |dR| -- but WHY it is not small matters:
      max_logit_err=None  inf for every such crop (the round-4 behaviour: accepts the crop unconditionally);
      max_logit_err=e     inf only if the crop is ill-conditioned BY THE REFERENCE'S LOGITS ALONE, i.e. an ordinary logit error of
                          size e (the bound the caller puts on the logits) already takes the first-order term past 0.25; a crop
                          that is well conditioned and still moved that far has a large logit error of its own: bound 0, the
                          caller's `per_crop <= bound` fails on it."""
    d = (d6_got.double() - d6_ref.double()).abs().max(1).values
    amp = rot6d_amplification(d6_ref)
    lin = math.sqrt(3.0) * amp * d
    b = slack * lin + floor
    big = lin > 0.25
    if max_logit_err is None:
        b[big] = float("inf")
    else:
        ill = math.sqrt(3.0) * amp * float(max_logit_err) > 0.25
        b[big & ill] = float("inf")
        b[big & ~ill] = 0.0
    return b


def ill_conditioned(d6_ref, max_logit_err):
    """(B,) bool: crops whose 6-D -> R map turns a logit error of `max_logit_err` into more than 0.25 (first order) -- no bound on
    their |dR| exists; judged on the reference's logits only."""
    return math.sqrt(3.0) * rot6d_amplification(d6_ref) * float(max_logit_err) > 0.25
