"""-m gpu: the fused stage-0 ConvNeXt MLP in the three-product form (csrc/gemm_mlp_fused.hip, gdrnpp_convnext_mlp_f32_fused):
y = resid + gamma * fc2(gelu(fc1(x))) in one launch, computed transposed, hidden tensor never in HBM.  Pinned here: the result
against fp64 at the error level of the two three-product launches it replaces (and of the six-product form), ragged row counts, the
two range words (x rows below range -> fc1's word, hidden beyond fp16 -> fc2's), the dispatch in hip_layers.convnext_mlp including
the fall-back to two launches once a layer was demoted, and the whole network with it against the reference fixture."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _problem(m, seed=0, c=128):
    torch.manual_seed(seed)
    x = torch.randn(m, c, device=DEV)
    x[:, :3] *= 30.0                                              # outlier channels, as LayerNorm outputs have
    w1 = torch.randn(4 * c, c, device=DEV) * c ** -0.5
    w2 = torch.randn(c, 4 * c, device=DEV) * (4 * c) ** -0.5
    b1, b2 = torch.randn(4 * c, device=DEV) * 0.3, torch.randn(c, device=DEV) * 0.3
    gamma, res = torch.randn(c, device=DEV), torch.randn(m, c, device=DEV)
    return x, w1, w2, b1, b2, gamma, res


def _want(x, w1, w2, b1, b2, gamma, res):
    h = F.gelu(x.double() @ w1.double().t() + b1.double())
    return res.double() + gamma.double() * (h @ w2.double().t() + b2.double())


@pytest.mark.parametrize("m,c", [(256 * 256, 128), (70000, 128), (300, 128), (1, 128), (256 * 256, 256), (33000, 256), (129, 256)])
def test_fused_mlp_vs_fp64_and_the_two_launches(hip, m, c):
    """C = 128: ConvNeXt-B stage 0 (two workgroups per CU); C = 256: stage 1 (one workgroup per CU, 512 registers per lane)."""
    x, w1, w2, b1, b2, gamma, res = _problem(m, seed=m, c=c)
    want = _want(x, w1, w2, b1, b2, gamma, res)
    pk = hip.pack_mlp_fused_f16x2(w1, w2)
    assert hip.mlp_fused_rows_in_range(pk) == (True, True)
    y = hip.convnext_mlp_f32_fused(x, pk, b1, b2, gamma, res)
    assert hip.split2_range_words() == {}
    errs = {}
    for name, pack in (("three", hip.pack_weight_f16x2), ("six", hip.pack_weight_bf16x3)):
        hid = hip.linear_f32_split(x, pack(w1), b1, "gelu")
        errs[name] = hip.linear_f32_split(hid, pack(w2), b2, "scale_res", gamma, res)
    scale = want.abs().max().item()
    e_f, e3, e6 = ((o.double() - want).abs().max().item() / scale for o in (y, errs["three"], errs["six"]))
    print(f"\nM={m} C={c}: fused {e_f:.2e}  two three-product launches {e3:.2e}  six products {e6:.2e}")
    assert torch.isfinite(y).all()
    assert e_f <= 1.3 * e6 + 4e-7 and e_f <= 1.5 * e3 + 2e-7, (e_f, e3, e6)
    assert ((y - errs["three"]).abs().max() / scale).item() < 3e-6
    try:                                                          # the A/B form: the plain tile loop (first GEMM in two accumulators: fp32 rounding)
        hip.set_option("mlp_fused_pipe", 0)
        y0 = hip.convnext_mlp_f32_fused(x, pk, b1, b2, gamma, res)
        assert ((y0 - y).abs().max() / scale).item() < 2e-6
        assert torch.equal(hip.convnext_mlp_f32_fused(x, pk, b1, b2, gamma, res), y0)      # run to run: bit-identical
    finally:
        hip.set_option("mlp_fused_pipe", 1)


@pytest.mark.parametrize("c", [128, 256])
def test_fused_mlp_range_words(hip, c):
    m = 4096
    x, w1, w2, b1, b2, gamma, res = _problem(m, seed=5, c=c)
    pk = hip.pack_mlp_fused_f16x2(w1, w2)
    hip.convnext_mlp_f32_fused(x, pk, b1, b2, gamma, res, 11, 12)
    assert hip.split2_range_words() == {}
    xs = x.clone()
    xs[1234] *= 1e-3                                              # one pixel's row below the range: fc1's word
    hip.convnext_mlp_f32_fused(xs, pk, b1, b2, gamma, res, 11, 12)
    assert hip.split2_range_words() == {11: hip.X3_SMALL_ROWS}
    xs[1234] = 0.0                                                # an all-zero row is exact
    hip.convnext_mlp_f32_fused(xs, pk, b1, b2, gamma, res, 11, 12)
    assert hip.split2_range_words() == {}
    b1_big = b1.clone()
    b1_big[77] = 9.0e4                                            # GELU(9e4) = 9e4 > 65504: the hidden operand of fc2 overflows
    y = hip.convnext_mlp_f32_fused(x, pk, b1_big, b2, gamma, res, 11, 12)
    assert hip.split2_range_words() == {12: hip.X3_NONFINITE} and not torch.isfinite(y).all()
    xs = x.clone()
    xs[7, 3] = float("inf")                                       # a non-finite input: fc1's word (and everything behind it)
    hip.convnext_mlp_f32_fused(xs, pk, b1, b2, gamma, res, 11, 12)
    words = hip.split2_range_words()
    assert words[11] == hip.X3_NONFINITE and words[12] & hip.X3_NONFINITE
    # GELU(-25 + ..): most hidden values are ~ -1e-8 (h = l = 0 in fp16: such a row counts as a zero row), the pixels with a large
    # outlier channel reach -5 .. -3 in a few hidden units -> GELU ~ -1e-4: non-zero rows far below the range
    tiny_h = hip.convnext_mlp_f32_fused(x, pk, b1 - 25.0, b2, gamma, res, 11, 12)
    assert hip.split2_range_words() == {12: hip.X3_SMALL_ROWS} and torch.isfinite(tiny_h).all()
    # weight rows: a non-zero row 2^-17 below the tensor maximum makes the layer ineligible
    w1b = w1.clone()
    w1b[5] *= 2.0 ** -18
    assert hip.mlp_fused_rows_in_range(hip.pack_mlp_fused_f16x2(w1b, w2)) == (False, True)


def test_convnext_mlp_dispatches_the_fused_kernel_and_falls_back_when_demoted(hip):
    from gdrnpp_bop2022_amd.gdrn_modeling import engine, hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling.backbones import Mlp

    assert hip_layers.gemm_products() == 3
    hip_layers.reset_x3_demotions()
    torch.manual_seed(2)
    c = 128
    mlp = Mlp(c, 4 * c).cuda().eval()
    with torch.no_grad():
        mlp.fc1.bias.uniform_(0.5, 1.5)                           # keeps the hidden rows in range when x is tiny (only fc1 is flagged below)
    gamma = torch.randn(c, device=DEV)
    x = torch.randn(16, 64, 64, c, device=DEV)                    # 65 536 pixels: above the fused form's threshold
    sc = torch.randn(16, 64, 64, c, device=DEV)
    cache = {}

    def run(inp):
        timer = hip.LaunchTimer()
        hip.set_launch_timer(timer)
        try:
            with torch.no_grad():
                y = engine.run_with_range_check(lambda: hip_layers.convnext_mlp(mlp, gamma, inp, sc[:inp.shape[0]], cache))
        finally:
            hip.set_launch_timer(None)
        return y, [r[0] for r in timer.records]

    try:
        y, kinds = run(x)
        assert kinds == ["mlp_fused" + hip.X3]
        with torch.no_grad():
            want = sc.double() + gamma.double() * mlp.double()(x.double())
            mlp.float()
        assert ((y.double() - want).abs().max() / want.abs().max()).item() < 2e-6
        _, kinds = run(x[:4])                                      # below the threshold (32 768 pixels): two launches (fc2 has too few tiles for x3)
        assert len(kinds) == 2 and kinds[0] == "linear" + hip.X3 and "mlp_fused" + hip.X3 not in kinds
        hip_layers.set_fused_mlp_x3(False)
        _, kinds = run(x)
        assert kinds == ["linear" + hip.X3] * 2
        hip_layers.set_fused_mlp_x3(True)
        y_t, kinds = run(x * 1e-3)                                # fc1's rows below range: repeated with six products, fc1 demoted
        assert kinds[0] == "mlp_fused" + hip.X3 and len(kinds) == 3 and all(k in ("linear", "linear_splitk") for k in kinds[1:])
        assert list(hip_layers.x3_demoted().values()) == [hip.X3_SMALL_ROWS]
        _, kinds = run(x)                                          # fc1 on six products from now on: two launches, fc2 on three
        assert len(kinds) == 2 and kinds[0] in ("linear", "linear_splitk") and kinds[1] == "linear" + hip.X3
        assert hip.split2_range_words() == {}
    finally:
        hip_layers.set_fused_mlp_x3(True)
        hip_layers.reset_x3_demotions()
