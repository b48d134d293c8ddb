"""GPU parity, module by module: the HIP sub-sequences of givepose_amd.PoseNet (the very launches forward_device runs
for that module) against the per-module golden vectors scripts/gen_golden.py captured from the reference's own classes:
TopDownXyzHead (network/xyz_head.py:349-366), DCNv3_C (network/dcnv3.py:32-38), MAPEncoder
(network/conv_pnp_net.py:303-332), ConvPnPNet (network/conv_pnp_net.py:137-201), MAPTransformerEncoer
(network/attention_pnp_net.py:143-157).

Tolerances: fp32 storage (fp32 MFMA, and the split-operand fp16 MFMA mode) -- 2e-4 abs on the maps / features, 1e-4 on
rot6d and t (north_star's bar applied at module level); fp16 storage -- relative to the output scale, reported by the print, bounded loosely (it is the throughput mode).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


SPLIT = "split"      # fp32 storage, dense contractions as split-operand fp16 MFMA: held to the fp32 tolerances


def _model(dtype, **kw):
    from givepose_amd import PoseNet, PoseNetConfig
    if dtype == SPLIT:
        return PoseNet(PoseNetConfig(**kw), dtype=torch.float32, seed=0, split_gemm=True).cuda()
    return PoseNet(PoseNetConfig(**kw), dtype=dtype, seed=0).cuda()


@pytest.fixture(scope="module")
def nets():
    return {torch.float32: _model(torch.float32), torch.float16: _model(torch.float16), SPLIT: _model(SPLIT)}


def _check(got, exp, dt, tol32, rel16, what):
    err = float(np.abs(got - exp).max())
    scale = float(np.abs(exp).max())
    print(f"{what} {dt}: max abs err {err:.3e} (output scale {scale:.3e})")
    if dt != torch.float16:
        assert err < tol32, (what, err)
    else:
        assert err < rel16 * max(scale, 1.0), (what, err)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, SPLIT])
@pytest.mark.parametrize("head", ["xyz_nocs_head", "xyz_deform_head"])
def test_xyz_head_golden(golden, nets, head, dt):
    z = golden(head)
    got = nets[dt].run_xyz_head(head, T(z["x"])).cpu().numpy()
    _check(got, z["expected"], dt, 2e-4, 2e-2, head)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, SPLIT])
def test_dcnv3_c_module_golden(golden, nets, dt):
    """features.3's DCNv3_C on a 16x16 map (the fixture's geometry = layer 2's, weights = layer 1's), B = 4:
    conv1x1 folded into input_proj, dw3x3 -> LN -> GELU on the consumed quarter only, fused softmax, gather, output_proj."""
    z = golden("dcnv3_module")
    got = nets[dt].run_dcnv3_c(2, T(z["x"]), weights_of=1).cpu().numpy()
    _check(got, z["expected"], dt, 2e-4, 2e-2, "dcnv3_c")


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, SPLIT])
@pytest.mark.parametrize("B", [1, 4, 5])
def test_map_encoder_golden(golden, nets, B, dt):
    z = golden(f"map_encoder_B{B}")
    got = nets[dt].run_map_encoder(T(z["x"])).cpu().numpy()
    _check(got, z["expected"], dt, 3e-4, 3e-2, f"map_encoder B{B}")


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, SPLIT])
def test_pnp_net_golden(golden, nets, dt):
    from givepose_amd import synth
    z = golden("pnp_net")
    data = {k: T(v).cuda() for k, v in synth.synth_batch(2, seed=5).items()}
    rot6d, t = nets[dt].run_pnp(T(z["x"]).cuda(), data)
    _check(rot6d.cpu().numpy(), z["rot"], dt, 1e-4, 2e-2, "pnp rot6d")
    _check(t.cpu().numpy(), z["t"], dt, 1e-4, 2e-2, "pnp t")


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, SPLIT])
def test_map_transformer_golden(golden, dt):
    z = golden("map_transformer")
    net = _model(dt, nocsmap_encoder="att")
    got = net.run_map_encoder(T(z["x"])).cpu().numpy()
    _check(got, z["expected"], dt, 3e-4, 3e-2, "map_transformer")
