"""-m gpu: the three-product ("fp16x2") form of the split GEMM — the default for large launches (csrc/gemm_split2_pipe.hip,
hip_layers.set_gemm_products(3)).  What is pinned here:
  * the packed weight image is h + l = w * 2^e to 2^-22, with the scale in the trailer;
  * linear / conv3x3 (+ GroupNorm statistics) against fp64 at MLP / head shapes, every epilogue, ragged M, the panel walk, operands
    with outlier channels: the error is that of the fp32 accumulation chain, which all kernels share — three products 4e-7 ..
    3e-6 of the output scale at K = 128 .. 4096, six products 5e-7 .. 3e-6, hipBLASLt fp32 6e-7 .. 4e-6 on the same operands;
    the operand representation alone (fp64 evaluation of the same three products) contributes 1.3e-7 .. 2.2e-7
    (tools/split2_error_probe.py).  Bars: <= 1.3 x six products + 4e-7 and <= the library's fp32 GEMM + 1e-7;
  * the range of the form, checked by every launch on both sides: a tensor at scale 1e-3 loses l to the fp16 subnormal spacing
    (2^-25 absolute) and raises the SMALL_ROWS bit of the launch's range word, an activation beyond 65504 the NONFINITE bit; the
    same test on the weight rows at pack time; engine.inference_step then repeats the step with six products (records bit-equal to
    the six-product mode) and keeps the flagged layer there — also when the first batches were healthy, also under a hipGraph;
  * the whole network with three products against the REFERENCE's recorded outputs (tolerances of BASELINE.json's north_star:
    maps 1e-4 of scale, R / t 1e-4) and against the six-product path at 128 ROIs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import netgolden as NG

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _amax(t):
    assert torch.isfinite(t).all(), "non-finite elements"
    return t.abs().max().item()


def test_packed_image_is_the_scaled_weight_to_22_bits(hip):
    torch.manual_seed(0)
    w = torch.randn(256, 96 * 2, device=DEV) * 0.02
    w[3, 5] = 0.7          # one large entry sets the scale; the small ones must keep their bits
    w[7, :8] = 0.0
    pk = hip.pack_weight_f16x2(w)
    assert pk.dtype == torch.float16 and pk.shape == (2, 12, 2, 2, 128, 8)
    planes, inv = hip.unpack_weight_f16x2(pk)
    e = round(np.log2(1.0 / inv))
    assert 2.0 ** e == 1.0 / inv and 2 ** 13 <= 0.7 * 2.0 ** e < 2 ** 14
    back = (planes[0].double() + planes[1].double()) * inv
    err = (back - w.double()).abs()
    assert (err <= w.double().abs() * 2.0 ** -21 + 2.0 ** -25 * inv).all()
    assert (back[7, :8] == 0).all()
    # h is the fp16 rounding of the scaled weight, l the rounding of the exact residual
    ws = w.double() * 2.0 ** e
    assert torch.equal(planes[0], ws.float().half())
    assert torch.equal(planes[1], (ws - planes[0].double()).float().half())
    # rows verdict of the pack kernel: all-zero rows are fine, a non-zero row 2^-17 below the tensor maximum is not
    assert hip.packed_rows_in_range(pk)
    w[9] = 0.0
    assert hip.packed_rows_in_range(hip.pack_weight_f16x2(w))
    w[9] *= 0.0
    w[9, 3] = 0.7 * 2.0 ** -15
    assert not hip.packed_rows_in_range(hip.pack_weight_f16x2(w))


@pytest.mark.parametrize("m,k,n,epi", [(4096, 128, 512, "gelu"), (4096, 512, 128, "scale_res"), (1000, 2048, 512, "scale_res"),
                                       (777, 1024, 2304, "none"), (8192, 512, 2048, "gelu"), (2048, 4096, 1024, "none")])
def test_linear_three_products_vs_fp64(hip, m, k, n, epi):
    torch.manual_seed(m + k + n)
    x = torch.randn(m, k, device=DEV)
    x[:, :3] *= 40.0                                  # outlier channels, as ConvNeXt activations have
    w = torch.randn(n, k, device=DEV) * (k ** -0.5)
    b = torch.randn(n, device=DEV)
    gamma = torch.randn(n, device=DEV) if epi == "scale_res" else None
    res = torch.randn(m, n, device=DEV) if epi == "scale_res" else None
    out3 = hip.linear_f32_split(x, hip.pack_weight_f16x2(w), b, epi, gamma, res)
    out6 = hip.linear_f32_split(x, hip.pack_weight_bf16x3(w), b, epi, gamma, res)
    want = x.double() @ w.double().t() + b.double()
    got32 = F.linear(x, w, b)
    if epi == "gelu":
        want, got32 = F.gelu(want), F.gelu(got32)
    elif epi == "scale_res":
        want, got32 = res.double() + gamma.double() * want, torch.addcmul(res, got32, gamma)
    scale = _amax(want)
    e3, e6, e32 = (_amax(o.double() - want) / scale for o in (out3, out6, got32))
    print(f"\nM={m} K={k} N={n} {epi}: three products {e3:.2e}  six products {e6:.2e}  torch fp32 {e32:.2e}")
    assert e3 <= 1.3 * e6 + 4e-7 and e3 <= e32 + 1e-7, (e3, e6, e32)
    assert not hip.split2_nonfinite()


def test_linear_three_products_panel_walk_is_bitwise_row_major(hip):
    """N / 128 >= 8 and a packed image above 2 MB: tiles are walked in panels — same tiles, same arithmetic."""
    torch.manual_seed(1)
    m, k, n = 5000, 512, 2048
    x, w, b = torch.randn(m, k, device=DEV), torch.randn(n, k, device=DEV) * k ** -0.5, torch.randn(n, device=DEV)
    pk = hip.pack_weight_f16x2(w)
    outs = []
    for panel in (0, 3, 4, 8):
        hip.set_option("split_gemm_panel", panel)
        outs.append(hip.linear_f32_split(x, pk, b, "gelu"))
    hip.set_option("split_gemm_panel", 4)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("m,k,n,epi", [(3000, 256, 512, "gelu"), (1024, 2048, 256, "scale_res"), (70000, 128, 1024, "none")])
def test_wide_block_tiles_are_bitwise_equal(hip, m, k, n, epi):
    """256 x 256 block tiles (one wave per SIMD, two packed weight tiles side by side) against 256 x 128: same products, same order."""
    torch.manual_seed(m)
    x, w, b = torch.randn(m, k, device=DEV), torch.randn(n, k, device=DEV) * k ** -0.5, torch.randn(n, device=DEV)
    gamma = torch.randn(n, device=DEV) if epi == "scale_res" else None
    res = torch.randn(m, n, device=DEV) if epi == "scale_res" else None
    pk = hip.pack_weight_f16x2(w)
    outs = []
    for wide in (0, 1):
        hip.set_option("split2_wide", wide)
        outs.append(hip.linear_f32_split(x, pk, b, epi, gamma, res))
    hip.set_option("split2_wide", -1)
    assert torch.equal(outs[0], outs[1])
    # the 3x3 convolution with GroupNorm statistics, both forms
    xc = torch.randn(8, 64, 32, 32, device=DEV).contiguous(memory_format=torch.channels_last)
    wc = torch.randn(256, 64, 3, 3, device=DEV) * (9 * 64) ** -0.5
    pkc = hip.pack_conv_weight_f16x2(wc)
    gw, gb = torch.randn(256, device=DEV), torch.randn(256, device=DEV)
    ys = []
    for wide in (0, 1):
        hip.set_option("split2_wide", wide)
        ys.append(hip.conv3x3_f32_split(xc, pkc, b[:256].contiguous(), True))
    hip.set_option("split2_wide", -1)
    assert torch.equal(ys[0], ys[1])


def test_small_scale_activations_raise_the_small_rows_bit(hip):
    """A tensor at scale 1e-3: l falls into the fp16 subnormal range, absolute operand error 2^-25 -> ~1e-5 relative — outside what
    the mode promises, and the launch says so (SMALL_ROWS); rows that are entirely zero are exact and raise nothing; one small row
    among healthy ones raises the bit; the conv forms judge their im2col rows."""
    torch.manual_seed(2)
    m, k, n = 2048, 512, 256
    x, w = torch.randn(m, k, device=DEV), torch.randn(n, k, device=DEV) * k ** -0.5
    pk = hip.pack_weight_f16x2(w)
    hip.linear_f32_split(x, pk, None)
    assert hip.split2_range_words() == {}
    out3 = hip.linear_f32_split(x * 1e-3, pk, None)
    want = (x * 1e-3).double() @ w.double().t()
    e3 = _amax(out3.double() - want) / _amax(want)
    assert 5e-7 < e3 < 4e-5, e3
    assert hip.split2_range_words() == {0: hip.X3_SMALL_ROWS} and hip.split2_range_words() == {}
    x[100:300] = 0.0                                     # zero rows (padding): exact, no word
    hip.linear_f32_split(x, pk, None)
    assert hip.split2_range_words() == {}
    x[1999] *= 0.03                                      # rms 0.03 < 2^-4 in ONE row of the last tile
    hip.linear_f32_split(x, pk, None, x3_slot=17)
    assert hip.split2_range_words() == {17: hip.X3_SMALL_ROWS}
    x[1999] *= 3.0                                       # rms 0.09: inside
    hip.linear_f32_split(x, pk, None, x3_slot=17)
    assert hip.split2_range_words() == {}
    # ragged M: the rows behind M are copies of the last row, never a verdict of their own
    hip.linear_f32_split(x[:1000], pk, None)
    assert hip.split2_range_words() == {}
    xc = torch.randn(4, 64, 32, 32, device=DEV).contiguous(memory_format=torch.channels_last)
    pkc = hip.pack_conv_weight_f16x2(torch.randn(128, 64, 3, 3, device=DEV) * (9 * 64) ** -0.5)
    hip.conv3x3_f32_split(xc, pkc, None, x3_slot=5)      # corner pixels see 4 of 9 taps: rms 2/3, still inside
    assert hip.split2_range_words() == {}
    xc[2, :, 10:20, 10:20] *= 0.01                       # a patch of tiny pixels in one image
    hip.conv3x3_f32_split(xc, pkc, None, x3_slot=5)
    assert hip.split2_range_words() == {5: hip.X3_SMALL_ROWS}


@pytest.mark.parametrize("n,cin,cout,hw,gelu", [(4, 256, 256, 64, False), (6, 256, 256, 32, True), (3, 96, 128, 16, False)])
def test_conv3x3_three_products_vs_fp64(hip, n, cin, cout, hw, gelu):
    torch.manual_seed(n + cin)
    x = torch.randn(n, cin, hw, hw, device=DEV).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * (9 * cin) ** -0.5
    b = torch.randn(cout, device=DEV)
    out3 = hip.conv3x3_f32_split(x, hip.pack_conv_weight_f16x2(w), b, gelu)
    out6 = hip.conv3x3_f32_split(x, hip.pack_conv_weight_bf16x3(w), b, gelu)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if gelu:
        want = F.gelu(want)
    e3, e6 = (_amax(o.double() - want) / _amax(want) for o in (out3, out6))
    assert out3.is_contiguous(memory_format=torch.channels_last)
    assert e3 <= 1.3 * e6 + 4e-7, (e3, e6)


@pytest.mark.parametrize("n,cin,cout,hw,k,stride,pad,gelu", [(8, 128, 256, 64, 2, 2, 0, False), (4, 96, 128, 64, 3, 2, 1, False),
                                                         (3, 64, 128, 20, 3, 1, 0, True), (2, 32, 128, 17, 5, 2, 2, False)])
def test_general_conv_three_products_vs_fp64(hip, n, cin, cout, hw, k, stride, pad, gelu):
    """KH x KW / stride / zero-pad convolutions (ConvNeXt's 2x2/2 downsamples, Patch-PnP's 3x3/2, odd sizes with image borders)."""
    torch.manual_seed(n + cin + k)
    x = torch.randn(n, cin, hw, hw, device=DEV).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device=DEV) * (k * k * cin) ** -0.5
    b = torch.randn(cout, device=DEV)
    hip.set_conv_splitk(False)
    try:
        out3 = hip.conv2d_f32_split(x, hip.pack_conv_weight_f16x2(w), b, k, k, stride, pad, gelu)
        out6 = hip.conv2d_f32_split(x, hip.pack_conv_weight_bf16x3(w), b, k, k, stride, pad, gelu)
    finally:
        hip.set_conv_splitk(True)
    want = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    if gelu:
        want = F.gelu(want)
    assert out3.shape == want.shape
    e3, e6 = (_amax(o.double() - want) / _amax(want) for o in (out3, out6))
    assert e3 <= 1.3 * e6 + 4e-7, (e3, e6)


def test_conv3x3_groupnorm_three_products_matches_six(hip):
    torch.manual_seed(5)
    n, c, hw, groups = 16, 256, 64, 32
    x = torch.randn(n, c, hw, hw, device=DEV).contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, c, 3, 3, device=DEV) * (9 * c) ** -0.5
    b, gw, gb = torch.randn(c, device=DEV), torch.randn(c, device=DEV), torch.randn(c, device=DEV)
    y3 = hip.conv3x3_groupnorm_act(x, hip.pack_conv_weight_f16x2(w), b, gw, gb, groups, 1e-5, gelu=True)
    y6 = hip.conv3x3_groupnorm_act(x, hip.pack_conv_weight_bf16x3(w), b, gw, gb, groups, 1e-5, gelu=True)
    want = F.gelu(F.group_norm(F.conv2d(x.double(), w.double(), b.double(), padding=1), groups, gw.double(), gb.double(), 1e-5))
    assert y3 is not None and y6 is not None
    s = _amax(want)
    e3, e6 = _amax(y3.double() - want) / s, _amax(y6.double() - want) / s
    assert e3 <= 1.3 * e6 + 4e-7 and e6 <= 3e-6, (e3, e6)


def test_overflow_raises_the_nonfinite_bit(hip):
    torch.manual_seed(3)
    x, w = torch.randn(512, 64, device=DEV), torch.randn(128, 64, device=DEV)
    pk = hip.pack_weight_f16x2(w)
    hip.linear_f32_split(x, pk, None)
    assert not hip.split2_nonfinite()
    x[17, 5] = 7.0e4                                   # beyond fp16: h = inf
    out = hip.linear_f32_split(x, pk, None, x3_slot=3)
    assert not torch.isfinite(out[17]).all()
    assert hip.split2_range_words(reset=False) == {3: hip.X3_NONFINITE} and hip.split2_nonfinite() and not hip.split2_nonfinite()
    x[17, 5] = 6.5e4                                   # the largest binade still works
    out = hip.linear_f32_split(x, pk, None)
    assert torch.isfinite(out).all() and not hip.split2_nonfinite()
    x[17, 5] = float("nan")                            # non-finite input: same bit (the A side is judged by its own row sums)
    hip.linear_f32_split(x, pk, None)
    assert hip.split2_range_words() == {0: hip.X3_NONFINITE}
    # words are per stream: an overflow on a side stream is not consumed by (and does not leak into) the current one
    x[17, 5] = 7.0e4
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hip.linear_f32_split(x, pk, None)
        side.synchronize()
        assert not torch.cuda.current_stream() == torch.cuda.default_stream()
    assert not hip.split2_nonfinite()
    with torch.cuda.stream(side):
        assert hip.split2_nonfinite() and not hip.split2_nonfinite()
    # C ABI with range_flag = NULL: the library's own word
    import ctypes
    out = torch.empty(512, 128, device=DEV)
    lib = hip.load()
    args = (x.data_ptr(), pk.data_ptr(), None, None, None, out.data_ptr(), 512, 128, 64, 0, None, None)
    assert lib.gdrnpp_linear_f32_split2(*args) == 0
    f = ctypes.c_int(0)
    assert lib.gdrnpp_split2_range_word(ctypes.byref(f), 1, None) == 0 and f.value == hip.X3_NONFINITE
    assert lib.gdrnpp_split2_range_word(ctypes.byref(f), 1, None) == 0 and f.value == 0
    x[17] = 1e-3
    x[17, 5] = 1e-3
    assert lib.gdrnpp_linear_f32_split2(*args) == 0
    assert lib.gdrnpp_split2_range_word(ctypes.byref(f), 1, None) == 0 and f.value == hip.X3_SMALL_ROWS


@pytest.fixture()
def three_products(hip):
    from gdrnpp_bop2022_amd.gdrn_modeling import engine, hip_layers
    old_products, old_tiles = hip_layers.gemm_products(), hip.SPLIT2_MIN_TILES
    hip_layers.set_gemm_products(3)
    hip_layers.reset_x3_demotions()
    engine._X3_OVERFLOW_STEPS = 0
    yield hip_layers
    hip_layers.set_gemm_products(old_products)          # what the session ran before (the library default: 3)
    hip.SPLIT2_MIN_TILES = old_tiles
    hip_layers.reset_x3_demotions()
    engine._X3_OVERFLOW_STEPS = 0


@pytest.mark.parametrize("ds", ["ycbv", "ycbvso"])
def test_network_with_three_products_matches_reference_forward(hip, three_products, ds):
    """The 4-ROI fixtures of the reference's own forward, every eligible layer forced onto the three-product kernels."""
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

    hip.SPLIT2_MIN_TILES = 1
    fx = NG.load_fixture(ds)
    model, _ = build_model_optimizer(get_cfg(NG.cfg_name(ds), opts=["TEST.USE_DEPTH_REFINE=True"]))
    model.load_state_dict(NG.seeded_reference_state_dict(model, fx), strict=True)
    x = torch.from_numpy(NG.net_image()).cuda()
    timer = hip.LaunchTimer()
    hip.set_launch_timer(timer)
    try:
        with torch.no_grad():
            out = {k: v.cpu().numpy() for k, v in model(x, **NG.forward_kwargs(fx, "cuda")).items()}
    finally:
        hip.set_launch_timer(None)
    kinds = [r[0] for r in timer.records]
    assert sum(k == "linear" + hip.X3 for k in kinds) == 72 and sum(k == "conv3x3" + hip.X3 for k in kinds) >= 5
    assert sum(k == "conv" + hip.X3 for k in kinds) >= 4           # the three downsamples + Patch-PnP's strided convolutions
    words = hip.split2_range_words()
    assert sum(k == "deconv" + hip.X3 for k in kinds) == 1 and words == {}, f"range words {words}"

    def err(a, ref, scale=None):
        a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
        return np.abs(a - ref).max() / (np.abs(ref).max() if scale is None else scale)

    for k in ("mask", "full_mask", "coor_x", "coor_y", "coor_z"):
        assert err(out[k], fx[k]) <= 1e-4, k
    assert err(out["region"][:, :, 1::4, 2::4], fx["region_sub"], float(fx["region_absmax"])) <= 1e-4
    assert np.abs(out["rot"] - fx["rot"]).max() <= 1e-4
    assert np.abs(out["trans"] - fx["trans"]).max() <= 1e-4 * max(1.0, np.abs(fx["trans"]).max())


def _headline_model_and_batch(hip, seed=7):
    from gdrnpp_bop2022_amd import synthetic as S
    from gdrnpp_bop2022_amd.gdrn_modeling import engine
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

    cfg = get_cfg("ycbv_convnext_a6", opts=["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True"])
    model, _ = build_model_optimizer(cfg)
    sd = model.state_dict()
    model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], seed), strict=True)
    b = 128
    x = torch.from_numpy(NG.net_image(b)).cuda()
    det = NG.net_detections(21, b)
    fx = dict(roi_cls=det["roi_cls"], roi_cam=det["roi_cam"], roi_wh=det["roi_wh"], roi_center=det["roi_center"],
              resize_ratio=det["resize_ratio"], scale=det["scale"], roi_extent=det["roi_extent"])
    kw = NG.forward_kwargs(fx, "cuda")
    # records straight from the network pose (no refine, no PnP)
    post = engine.GdrnHipPost(get_cfg("ycbv_convnext_a6"))
    batch = dict(roi_img=x, roi_cls=kw["roi_classes"], roi_cam=kw["roi_cams"], roi_wh=kw["roi_whs"], roi_center=kw["roi_centers"],
                 resize_ratio=kw["resize_ratios"], roi_coord_2d=kw["roi_coord_2d"], roi_extent=kw["roi_extents"])
    return model, post, batch, x, kw


def _six(hip_layers, fn):
    with hip_layers.forced_gemm_products(6):
        return fn()


def test_headline_batch_three_vs_six_products_and_overflow_retry(hip, three_products):
    """128 ROIs (the shapes bench.py times): maps / poses of the two modes agree far inside the 1e-4 tolerance and no layer leaves
    the range; with an fc1 bias that pushes the hidden tensor beyond the fp16 range the consuming fc2 raises NONFINITE,
    engine.inference_step returns the six-product result and from then on runs THAT layer on six products (no second repeat)."""
    from gdrnpp_bop2022_amd.gdrn_modeling import engine

    hip_layers = three_products
    model, post, batch, x, kw = _headline_model_and_batch(hip)
    with torch.no_grad():
        o3 = model(x, **kw)
        assert hip.split2_range_words() == {}, "a layer of the seeded headline model left the three-product range"
        o6 = _six(hip_layers, lambda: model(x, **kw))
    for k in ("mask", "full_mask", "coor_x", "coor_y", "coor_z", "region"):
        a, r = o3[k].float(), o6[k].float()
        assert ((a - r).abs().max() / r.abs().max()).item() <= 2e-5, k
    assert (o3["rot"] - o6["rot"]).abs().max().item() <= 1e-4          # north_star tolerance
    assert (o3["trans"] - o6["trans"]).abs().max().item() <= 1e-4 * max(1.0, o6["trans"].abs().max().item())

    with torch.no_grad():
        model.backbone.stages_2.blocks[5].mlp.fc1.bias[7] = 9.0e4      # GELU(9e4) = 9e4 > 65504 in the fc2 input
        want = _six(hip_layers, lambda: engine.inference_step(model, post, batch))
        model(x, **kw)
        words = hip.split2_range_words()
        assert words and all(w_ & hip.X3_NONFINITE for w_ in words.values())     # the plain forward trips the word ...
        reruns = engine.range_reruns()
        got = engine.inference_step(model, post, batch)  # ... and the step repeats itself with six products
    assert hip_layers.gemm_products() == 3 and hip.split2_range_words() == {} and engine.range_reruns() == reruns + 1
    assert torch.isfinite(got).all() and torch.equal(got, want)
    assert list(hip_layers.x3_demoted()) == [min(words)] and engine._X3_OVERFLOW_STEPS == 1     # the first layer that saw the inf
    with torch.no_grad():
        again = engine.inference_step(model, post, batch)               # that fc2 now runs on six products: nothing to repeat
    assert engine.range_reruns() == reruns + 1 and torch.isfinite(again).all()
    assert (again[:, :12] - want[:, :12]).abs().max().item() <= 1e-3     # (a 9e4 outlier in front of a LayerNorm: sane, not parity)
    # the same through a captured hipGraph built from scratch: the eager warm-up step trips, demotes, and the capture holds the mix
    hip_layers.reset_x3_demotions()
    engine._X3_OVERFLOW_STEPS = 0
    with torch.no_grad():
        g = engine.GraphedInference(model, post, batch, warmup=1)
        assert g.uses_x3 and g.captures == 1 and len(hip_layers.x3_demoted()) == 1
        got = g.replay()
    assert g.captures == 1 and engine.range_reruns() == reruns + 2 and torch.isfinite(got).all()
    assert (got[:, :12] - want[:, :12]).abs().max().item() <= 1e-3


@pytest.mark.parametrize("stage,index", [("stages_2", 9), ("stages_0", 1)])
def test_third_batch_leaves_the_range_on_the_small_side(hip, three_products, stage, index):
    """(stages_0: a block on the FUSED MLP kernel — once one of its layers is demoted the block runs as two launches whose
    three-product image has never been packed; GraphedInference packs it in an eager pass, never under capture.)
    Batches 1 and 2 are healthy; before batch 3 ONE layer's input shrinks to the 1e-3 scale (its LayerNorm affine is scaled in
    place — no load_state_dict, nothing that resets anything): the launch reports SMALL_ROWS, the step's records are bit-equal to
    the six-product mode, the layer stays on six products afterwards; the same under GraphedInference, which captures again."""
    from gdrnpp_bop2022_amd.gdrn_modeling import engine

    hip_layers = three_products
    model, post, batch, x, kw = _headline_model_and_batch(hip, seed=11)
    blk = getattr(model.backbone, stage).blocks[index]
    with torch.no_grad():
        reruns = engine.range_reruns()
        for _ in range(2):
            rec = engine.inference_step(model, post, batch)
        assert engine.range_reruns() == reruns and hip_layers.x3_demoted() == {} and torch.isfinite(rec).all()
        g = engine.GraphedInference(model, post, batch, warmup=1)
        assert g.uses_x3 and g.captures == 1 and torch.equal(g.replay(), rec)
        w_ok, b_ok = blk.norm.weight.clone(), blk.norm.bias.clone()
        w_tiny, b_tiny = w_ok * 1e-3, b_ok * 1e-3
        blk.norm.weight.copy_(w_tiny)
        blk.norm.bias.copy_(b_tiny)
        want = _six(hip_layers, lambda: engine.inference_step(model, post, batch))
        got = engine.inference_step(model, post, batch)
        assert engine.range_reruns() == reruns + 1 and torch.equal(got, want)
        demoted = hip_layers.x3_demoted()      # the block's fc1 (and its fc2 when GELU(bias) is as small: seeded biases are 0.1 u)
        assert 1 <= len(demoted) <= 2 and set(demoted.values()) == {hip.X3_SMALL_ROWS}
        again = engine.inference_step(model, post, batch)
        assert engine.range_reruns() == reruns + 1 and (again[:, :12] - want[:, :12]).abs().max().item() <= 1e-4
        # the graph was captured before the change and still holds the layer's three-product kernel: it sees the demotion and
        # captures again before replaying
        got_g = g.replay()
        assert g.captures == 2 and torch.equal(got_g, again) and engine.range_reruns() == reruns + 1
        # ... and a graph whose own replay trips: forget the demotion, capture afresh with healthy weights, shrink, replay
        blk.norm.weight.copy_(w_ok)
        blk.norm.bias.copy_(b_ok)
        hip_layers.reset_x3_demotions()
        g2 = engine.GraphedInference(model, post, batch, warmup=1)
        assert g2.captures == 1 and hip_layers.x3_demoted() == {}
        blk.norm.weight.copy_(w_tiny)
        blk.norm.bias.copy_(b_tiny)
        got_g2 = g2.replay()
        assert g2.captures == 2 and engine.range_reruns() == reruns + 2 and len(hip_layers.x3_demoted()) == len(demoted)
        assert torch.equal(got_g2, want)
        assert torch.equal(g2.replay(), again) and g2.captures == 2


def test_inference_step_needs_no_outer_no_grad(hip):
    """engine.inference_step switches autograd off itself: called bare it must still run this library's kernels (hip_layers are
    inference-only and stand aside when grad is enabled — a lost decorator once sent the whole step to the vendor libraries)."""
    from gdrnpp_bop2022_amd.gdrn_modeling import engine
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

    fx = NG.load_fixture("ycbv")
    model, _ = build_model_optimizer(get_cfg("ycbv_convnext_a6"))
    kw = NG.forward_kwargs(fx, "cuda")
    batch = dict(roi_img=torch.from_numpy(NG.net_image()).cuda(), roi_cls=kw["roi_classes"], roi_cam=kw["roi_cams"],
                 roi_wh=kw["roi_whs"], roi_center=kw["roi_centers"], resize_ratio=kw["resize_ratios"],
                 roi_coord_2d=kw["roi_coord_2d"], roi_extent=kw["roi_extents"])
    timer = hip.LaunchTimer()
    hip.set_launch_timer(timer)
    try:
        assert torch.is_grad_enabled()
        rec = engine.inference_step(model, engine.GdrnHipPost(get_cfg("ycbv_convnext_a6")), batch)
    finally:
        hip.set_launch_timer(None)
    assert rec.shape == (NG.B, 16) and not rec.requires_grad
    assert sum(r[0] in ("linear", "linear_splitk", "linear_x3") for r in timer.records) == 74


def test_small_scale_layer_input_moves_that_layer_to_six_products(hip, three_products):
    """The small side of the fp16x2 range is handled per layer and per launch: a layer fed with a tiny-scale tensor reports it, the
    work is repeated with six products (as exact as ever) and the layer stays there; its healthy sibling keeps three products."""
    from gdrnpp_bop2022_amd.gdrn_modeling import engine
    from gdrnpp_bop2022_amd.gdrn_modeling.backbones import Mlp

    hip_layers = three_products
    hip.SPLIT2_MIN_TILES = 1
    torch.manual_seed(21)
    c = 128
    mlp = Mlp(c, 4 * c).cuda().eval()
    with torch.no_grad():
        mlp.fc1.bias.add_(1.0)            # the hidden tensor (fc2's input) stays at scale ~1 whatever fc1's input is
    gamma = torch.randn(c, device=DEV)
    x = torch.randn(4, 32, 32, c, device=DEV)
    sc = torch.randn(4, 32, 32, c, device=DEV)
    cache = {}

    def run(inp):
        timer = hip.LaunchTimer()
        hip.set_launch_timer(timer)
        try:
            with torch.no_grad():
                y = engine.run_with_range_check(lambda: hip_layers.convnext_mlp(mlp, gamma, inp, sc, cache))
        finally:
            hip.set_launch_timer(None)
        return y, [r[0] for r in timer.records]

    y, kinds = run(x)
    assert kinds == ["linear" + hip.X3] * 2 and hip_layers.x3_demoted() == {}
    tiny = x * 1e-3
    y_t, kinds = run(tiny)
    assert kinds[:2] == ["linear" + hip.X3] * 2 and len(kinds) == 4 and all(k in ("linear", "linear_splitk") for k in kinds[2:])
    with torch.no_grad():
        want = sc.double() + gamma.double() * mlp.double()(tiny.double())
        mlp.float()
    assert ((y_t.double() - want).abs().max() / want.abs().max()).item() < 5e-7
    assert list(hip_layers.x3_demoted().values()) == [hip.X3_SMALL_ROWS]
    y_t2, kinds = run(tiny)               # fc1 on six products, fc2 still on three; nothing to repeat
    assert len(kinds) == 2 and kinds[0] in ("linear", "linear_splitk") and kinds[1] == "linear" + hip.X3
    assert ((y_t2.double() - want).abs().max() / want.abs().max()).item() < 2e-6
    assert hip.split2_range_words() == {}
