"""The "f16x2 rows" hand-over between the kernels of a ConvNeXt block (include/gdrnpp_hip.h: gdrnpp_linear_f32_split2_rows,
gdrnpp_dwconv7x7_ln_nhwc_rows): producers write an activation tensor already split into its fp16 h / l halves, the three-product
GEMM reads its MFMA operands straight from it.  The bar is bit-exactness against the fp32 hand-over: same conversions, same
products, same order — only the place where the split is computed moves.  Reference semantics of the block:
timm ConvNeXtBlock.forward (the reference's backbone, /root/reference/core/utils/timm_utils.py:34)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rows_of(x: torch.Tensor) -> torch.Tensor:
    """Host restatement of the layout: per 8 consecutive elements of a row, 8 h halves then 8 l halves (h = rn_f16(x), l = rn_f16(x - h))."""
    h = x.half()
    l = (x - h.float()).half()
    k = x.shape[-1]
    packed = torch.stack([h.view(*x.shape[:-1], k // 8, 8), l.view(*x.shape[:-1], k // 8, 8)], dim=-2).contiguous()
    return packed.view(torch.float32).view(x.shape)


@pytest.mark.parametrize("m,c", [(70000, 256), (32768, 512), (300, 128)])
def test_mlp_chain_through_rows_is_bitwise_the_fp32_chain(hip, m, c):
    torch.manual_seed(m + c)
    x = torch.randn(m, c, device=DEV)
    x[:, 5] *= 40.0                                              # an outlier channel, as ConvNeXt activations have
    x[7] = 0.0                                                   # an all-zero row (exempt from the small-rows test)
    w1 = torch.randn(4 * c, c, device=DEV) * c ** -0.5
    w2 = torch.randn(c, 4 * c, device=DEV) * (4 * c) ** -0.5
    b1, b2, gamma = torch.randn(4 * c, device=DEV), torch.randn(c, device=DEV), torch.randn(c, device=DEV)
    res = torch.randn(m, c, device=DEV)
    p1, p2 = hip.pack_weight_f16x2(w1), hip.pack_weight_f16x2(w2)
    hid = hip.linear_f32_split(x, p1, b1, "gelu")
    want = hip.linear_f32_split(hid, p2, b2, "scale_res", gamma, res)

    x_rows = _rows_of(x)
    hid_rows = hip.linear_f32_split(x_rows, p1, b1, "gelu", a_rows=True, c_rows=True)
    hh, hl = hip.f16x2_rows_decode(hid_rows)
    assert torch.equal(hh, hid.half()) and torch.equal(hl, (hid - hid.half().float()).half())
    assert torch.equal(hid_rows.view(torch.int32), _rows_of(hid).view(torch.int32))
    got = hip.linear_f32_split(hid_rows, p2, b2, "scale_res", gamma, res, a_rows=True)
    assert torch.equal(got, want)
    # mixed hand-overs: rows in / fp32 out, fp32 in / rows out
    assert torch.equal(hip.linear_f32_split(x_rows, p1, b1, "gelu", a_rows=True), hid)
    assert torch.equal(hip.linear_f32_split(x, p1, b1, "gelu", c_rows=True).view(torch.int32), hid_rows.view(torch.int32))
    assert torch.equal(hip.linear_f32_split(x, p1, b1, "none", c_rows=True).view(torch.int32),
                       _rows_of(hip.linear_f32_split(x, p1, b1, "none")).view(torch.int32))
    assert hip.split2_range_words() == {}


def test_rows_keep_the_range_words(hip):
    """Both sides of the range are still judged per launch: the consumer sums the squares of the h halves it reads."""
    torch.manual_seed(3)
    m, c = 4096, 256
    w = torch.randn(4 * c, c, device=DEV) * c ** -0.5
    b = torch.zeros(4 * c, device=DEV)
    p = hip.pack_weight_f16x2(w)
    x = torch.randn(m, c, device=DEV)
    hip.linear_f32_split(_rows_of(x), p, b, "gelu", a_rows=True, x3_slot=5)
    assert hip.split2_range_words() == {}
    x_small = x.clone(); x_small[100] *= 1e-3                     # one row below 2^-4 rms
    hip.linear_f32_split(_rows_of(x_small), p, b, "gelu", a_rows=True, x3_slot=5)
    assert hip.split2_range_words() == {5: hip.X3_SMALL_ROWS}
    x_big = x.clone(); x_big[9, 3] = 1e5                          # beyond fp16: h = inf in the rows tensor
    hip.linear_f32_split(_rows_of(x_big), p, b, "gelu", a_rows=True, x3_slot=6)
    assert hip.split2_range_words().get(6, 0) & hip.X3_NONFINITE
    with pytest.raises(Exception):
        hip.linear_f32_split(_rows_of(x), p, b, "none", a_rows=True)          # the rows-A kernels exist for the two MLP epilogues
    with pytest.raises(Exception):
        hip.linear_f32_split(x, hip.pack_weight_bf16x3(w), b, "gelu", c_rows=True)   # six-product kernels hand over fp32


@pytest.mark.parametrize("n,c,hw", [(8, 256, 32), (4, 512, 16), (3, 1024, 8), (2, 128, 64)])
def test_dwconv_ln_rows_output_is_the_split_of_its_fp32_output(hip, n, c, hw):
    torch.manual_seed(c)
    conv = nn.Conv2d(c, c, 7, padding=3, groups=c).to(DEV)
    ln = nn.LayerNorm(c, eps=1e-6).to(DEV)
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.2); ln.bias.normal_(0.0, 0.2)
        x = torch.randn(n, c, hw, hw, device=DEV).contiguous(memory_format=torch.channels_last)
        w49c = conv.weight.reshape(c, 49).t().contiguous()
        y = hip.dwconv7x7_ln(x, w49c, conv.bias, ln.weight, ln.bias, 1e-6)
        y_rows = hip.dwconv7x7_ln(x, w49c, conv.bias, ln.weight, ln.bias, 1e-6, y_rows=True)
    nhwc = y.permute(0, 2, 3, 1).contiguous()
    assert torch.equal(y_rows.permute(0, 2, 3, 1).contiguous().view(torch.int32), _rows_of(nhwc).view(torch.int32))


def test_convnext_stage_with_and_without_rows_is_bitwise_equal(hip):
    """Whole blocks (dwconv + LN -> fc1 -> GELU -> fc2 -> scale, residual) of the deep stages at the benchmark's row counts."""
    from gdrnpp_bop2022_amd.gdrn_modeling import engine, hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling.backbones import ConvNeXtBlock

    assert hip_layers.gemm_products() == 3
    hip_layers.set_fused_mlp_x3(True, 128)            # C = 256 blocks on the two-launch path here (from 32 768 pixels on they take the fused kernel)
    try:
        for c, hw, n in ((512, 16, 128), (256, 32, 32), (1024, 8, 128)):
            torch.manual_seed(c)
            blk = ConvNeXtBlock(c).to(DEV).eval()
            with torch.no_grad():
                blk.gamma.normal_(0.0, 0.5)
            x = torch.randn(n, c, hw, hw, device=DEV).contiguous(memory_format=torch.channels_last)

            def run():
                timer = hip.LaunchTimer()
                hip.set_launch_timer(timer)
                try:
                    with torch.no_grad():
                        y = engine.run_with_range_check(lambda: blk(x))
                finally:
                    hip.set_launch_timer(None)
                return y, [r[0] for r in timer.records]

            def takes_rows():
                with torch.no_grad():
                    return hip_layers.mlp_takes_rows(blk.mlp, blk.conv_dw, x, blk._cache)

            try:
                assert takes_rows()
                y_rows, kinds = run()
                assert kinds == ["hbm:dwconv7_ln", "linear" + hip.X3, "linear" + hip.X3]
                hip_layers.set_f16x2_rows(False)
                assert not takes_rows()
                y_f32, kinds = run()
                assert kinds == ["hbm:dwconv7_ln", "linear" + hip.X3, "linear" + hip.X3]
                assert torch.equal(y_rows, y_f32)
                hip_layers.set_f16x2_rows(True)
                # a demoted fc2 takes fc1's result as fp32, a demoted fc1 takes the LayerNorm output as fp32
                hip_layers.demote_x3({hip_layers.x3_slot(blk._cache, "fc2"): hip.X3_SMALL_ROWS})
                y_d2, kinds = run()
                assert kinds[1] == "linear" + hip.X3 and kinds[2] in ("linear", "linear_splitk")
                hip_layers.reset_x3_demotions()
                hip_layers.demote_x3({hip_layers.x3_slot(blk._cache, "fc1"): hip.X3_SMALL_ROWS})
                assert not takes_rows()
                y_d1, kinds = run()
                assert kinds[1] in ("linear", "linear_splitk") and kinds[2] == "linear" + hip.X3
                scale = y_f32.abs().max()
                assert ((y_d2 - y_f32).abs().max() / scale).item() < 5e-6 and ((y_d1 - y_f32).abs().max() / scale).item() < 5e-6
            finally:
                hip_layers.set_f16x2_rows(True)
                hip_layers.reset_x3_demotions()
    finally:
        hip_layers.set_fused_mlp_x3(True, 256)
