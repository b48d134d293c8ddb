"""GPU parity of the DCNv3 operator at the reference's full breadth (csrc/dcnv3_any.hip through the C ABI):
double / float / half, per-axis geometry, any group_channels, and the BACKWARD -- against
  (a) tests/golden/dcnv3_any_*.npz = the reference's own dcnv3_core_pytorch and its autograd
      (scripts/gen_golden_dcnv3_any.py; the recipe of network/ops_dcnv3/test.py:35-170), and
  (b) the C oracle (oracle/dcnv3_ref.c, pinned by the same vectors) on the parameter sets of the reference test that the
      fixtures do not hold: channels 1, 16, 30, 32, 64, 71, 1025 (test.py:262-265) and the PoseNet stride-2 quarter-buffer case.
Tolerances: fp64 1e-6 relative-to-scale (dcnv3_core_pytorch builds its sampling grid from fp32 linspace), fp32 the
reference test's own rtol 1e-2 / atol 1e-3 (test.py:88, 136) tightened to 1e-4 relative-to-scale, fp16 3e-3.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TOL = {torch.float64: 1e-6, torch.float32: 1e-4, torch.float16: 3e-3}


def _geom(z):
    kh, kw, sh, sw, ph, pw, dh, dw, G, D, rc = (int(v) for v in z["params"])
    return (kh, kw, sh, sw, ph, pw, dh, dw, G, D, float(z["offset_scale"])), rc


def _close(got, exp, dt, what):
    got, exp = got.double().cpu().numpy(), np.asarray(exp, dtype=np.float64)
    err, scale = float(np.abs(got - exp).max()), max(1.0, float(np.abs(exp).max()))
    assert err < TOL[dt] * scale, (what, dt, err, scale)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32, torch.float16])
@pytest.mark.parametrize("name", ["dcnv3_any_fwd_ref", "dcnv3_any_fwd_hw", "dcnv3_any_fwd_dil_rc", "dcnv3_any_bwd_D1", "dcnv3_any_bwd_D30"])
def test_forward_golden(golden, name, dt):
    from givepose_amd import dcnv3_forward
    z = golden(name)
    g, rc = _geom(z)
    a = [T(z[k]).to("cuda", dt) for k in ("input", "offset", "mask")]
    out = dcnv3_forward(*a, *g, 64, rc)
    assert out.dtype == dt and tuple(out.shape) == z["expected"].shape
    exp = z["expected"]
    if dt == torch.float16:      # expected = oracle on the rounded operands the kernel sees
        from oracle.dcnv3_c import dcnv3_forward_any_c
        exp = dcnv3_forward_any_c(*[t.double().cpu().numpy() for t in a], *g, rc)
    _close(out, exp, dt, name)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32, torch.float16])
@pytest.mark.parametrize("name", ["dcnv3_any_bwd_D1", "dcnv3_any_bwd_D16", "dcnv3_any_bwd_D30", "dcnv3_any_bwd_hw"])
def test_backward_golden(golden, name, dt):
    from givepose_amd import dcnv3_backward
    z = golden(name)
    g, rc = _geom(z)
    a = [T(z[k]).to("cuda", dt) for k in ("input", "offset", "mask", "grad_output")]
    gi, go, gm = dcnv3_backward(a[0], a[1], a[2], *g, a[3], 64, rc)
    assert gi.dtype == go.dtype == gm.dtype == dt            # dcnv3_cuda.cu:167-173
    exp = {k: z[k] for k in ("grad_input", "grad_offset", "grad_mask")}
    if dt == torch.float16:      # the kernel sees ROUNDED offsets (sampling positions move): expected = oracle on the same rounded operands
        from oracle.dcnv3_c import dcnv3_backward_any_c
        r = dcnv3_backward_any_c(*[t.double().cpu().numpy() for t in a[:3]], a[3].double().cpu().numpy(), *g, rc)
        exp = dict(zip(("grad_input", "grad_offset", "grad_mask"), r))
    for got, key in ((gi, "grad_input"), (go, "grad_offset"), (gm, "grad_mask")):
        assert tuple(got.shape) == z[key].shape
        _close(got, exp[key], dt, name + "." + key)


@pytest.mark.parametrize("channels", [1, 16, 30, 32, 64, 71, 1025])
def test_backward_reference_test_channel_counts(channels):
    """check_backward_equal_with_pytorch_double / _float (test.py:92-218): N=2, M=2, 8x8, K=3, offset_scale 2.0."""
    from givepose_amd import dcnv3_backward, dcnv3_forward
    from oracle.dcnv3_c import dcnv3_backward_any_c, dcnv3_forward_any_c
    g = torch.Generator().manual_seed(channels)
    N, M, H, W_, P = 2, 2, 8, 8, 9
    inp = torch.rand(N, H, W_, M * channels, generator=g, dtype=torch.float64) * 0.01
    off = torch.rand(N, H, W_, M * P * 2, generator=g, dtype=torch.float64) * 10
    msk = torch.rand(N, H, W_, M, P, generator=g, dtype=torch.float64) + 1e-5
    msk = (msk / msk.sum(-1, keepdim=True)).reshape(N, H, W_, M * P)
    go = torch.ones(N, H, W_, M * channels, dtype=torch.float64)          # output.sum().backward()
    geom = (3, 3, 1, 1, 1, 1, 1, 1, M, channels, 2.0)
    ref_out = dcnv3_forward_any_c(inp.numpy(), off.numpy(), msk.numpy(), *geom)
    ref = dcnv3_backward_any_c(inp.numpy(), off.numpy(), msk.numpy(), go.numpy(), *geom)
    for dt in (torch.float64, torch.float32):
        a = [t.to("cuda", dt) for t in (inp, off, msk)]
        _close(dcnv3_forward(*a, *geom, 2), ref_out, dt, "forward")
        got = dcnv3_backward(*a, *geom, go.to("cuda", dt), 2)
        for x, r, key in zip(got, ref, ("grad_input", "grad_offset", "grad_mask")):
            assert torch.allclose(x.double().cpu(), T(r), rtol=1e-2, atol=1e-3), (channels, dt, key)     # the reference test's bar
            _close(x, r, dt, f"D{channels}.{key}")


def test_backward_stride2_quarter_buffer_and_autograd_function():
    """The PoseNet geometry (stride 2, offset / mask handed over at FULL resolution, flat prefix consumed: SURVEY.md 0.3)
    through DCNv3Function.apply: forward on the wave kernel, backward on gp_dcnv3_backward; the gradient of the
    unconsumed three quarters of offset / mask is exactly zero (dcnv3_cuda.cu:128-130)."""
    from givepose_amd import DCNv3Function
    from oracle.dcnv3_c import dcnv3_backward_any_c
    g = torch.Generator().manual_seed(9)
    N, H, G, D, P = 4, 8, 4, 16, 9
    inp = (torch.rand(N, H, H, G * D, generator=g) - 0.5).cuda().requires_grad_()
    off = ((torch.rand(N, H, H, G * P * 2, generator=g) - 0.5) * 4).cuda().requires_grad_()
    msk = torch.softmax(torch.rand(N, H, H, G, P, generator=g), -1).reshape(N, H, H, G * P).cuda().requires_grad_()
    out = DCNv3Function.apply(inp, off, msk, 3, 3, 2, 2, 1, 1, 1, 1, G, D, 1.0, 256, 0)
    assert tuple(out.shape) == (N, 4, 4, G * D)
    go = torch.rand(out.shape, generator=g).cuda()
    out.backward(go)
    ref = dcnv3_backward_any_c(inp.detach().cpu().numpy(), off.detach().cpu().numpy(), msk.detach().cpu().numpy(), go.cpu().numpy(),
                               3, 3, 2, 2, 1, 1, 1, 1, G, D, 1.0)
    for got, r, key in ((inp.grad, ref[0], "grad_input"), (off.grad, ref[1], "grad_offset"), (msk.grad, ref[2], "grad_mask")):
        _close(got, r, torch.float32, key)
    consumed = N * 4 * 4 * G * P
    assert float(off.grad.reshape(-1)[consumed * 2:].abs().max()) == 0.0 and float(msk.grad.reshape(-1)[consumed:].abs().max()) == 0.0
    assert float(off.grad.reshape(-1)[:consumed * 2].abs().max()) > 0.0


def test_error_conditions_of_the_reference():
    from givepose_amd import _lib, dcnv3_backward, dcnv3_forward
    x = torch.zeros(3, 8, 8, 12, device="cuda", dtype=torch.float64)
    o, m = torch.zeros(3, 8, 8, 3 * 9 * 2, device="cuda", dtype=torch.float64), torch.zeros(3, 8, 8, 27, device="cuda", dtype=torch.float64)
    with pytest.raises(RuntimeError, match="wont match"):                                   # dcnv3_cuda.cu:50-53
        dcnv3_forward(x, o, m, 3, 3, 1, 1, 1, 1, 1, 1, 3, 5, 1.0, 256)
    with pytest.raises(_lib.GivePoseHipError, match="must divide im2col_step"):             # dcnv3_cuda.cu:48-49
        dcnv3_forward(x, o, m, 3, 3, 1, 1, 1, 1, 1, 1, 3, 4, 1.0, 2)
    with pytest.raises(_lib.GivePoseHipError, match="remove_center"):                      # dcnv3_func.py:181-182
        dcnv3_forward(x, torch.zeros(3 * 8 * 12 * 3 * 14 * 2, device="cuda", dtype=torch.float64),
                      torch.zeros(3 * 8 * 12 * 3 * 14, device="cuda", dtype=torch.float64), 3, 5, 1, 1, 1, 2, 1, 1, 3, 4, 1.0, 256, 1)
    with pytest.raises(RuntimeError, match="contiguous"):                                   # dcnv3_cuda.cu:29-31
        dcnv3_forward(x.transpose(1, 2), o, m, 3, 3, 1, 1, 1, 1, 1, 1, 3, 4, 1.0, 256)
    with pytest.raises(RuntimeError, match="CUDA tensor"):                                  # dcnv3_cuda.cu:32-34
        dcnv3_backward(x.cpu(), o, m, 3, 3, 1, 1, 1, 1, 1, 1, 3, 4, 1.0, torch.zeros_like(x), 256)
