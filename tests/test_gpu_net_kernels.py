"""Network-side NHWC HIP kernels against the plain PyTorch fp32 operators they replace (-m gpu).
Floating-point kernels with a different summation order: tolerance 1e-5 (rel) / 1e-5 (abs) stated per test."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cl(x):
    return x.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("n,c,h,w", [(3, 128, 64, 64), (2, 256, 32, 32), (2, 512, 16, 16), (2, 1024, 8, 8),
                                      (1, 128, 5, 7)])
@pytest.mark.parametrize("fuse_ln", [True, False])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2])
def test_dwconv7x7_ln(hip, n, c, h, w, fuse_ln, tile):
    """tile: -1 = the launch-size rule, 0 / 1 / 2 = the 2x8 / 2x4 / 1x4 pixel tile forced (option dwconv_tile)."""
    torch.manual_seed(c + h)
    conv = nn.Conv2d(c, c, 7, padding=3, groups=c).to(DEV)
    ln = nn.LayerNorm(c, eps=1e-6).to(DEV)
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.2)
        ln.bias.normal_(0.0, 0.2)
        x = _cl(torch.randn(n, c, h, w, device=DEV))
        w49c = conv.weight.reshape(c, 49).t().contiguous()
        hip.set_option("dwconv_tile", tile)
        try:
            y = hip.dwconv7x7_ln(x, w49c, conv.bias, ln.weight if fuse_ln else None, ln.bias if fuse_ln else None, 1e-6)
        finally:
            hip.set_option("dwconv_tile", -1)
        ref = conv(x)
        if fuse_ln:
            ref = F.layer_norm(ref.permute(0, 2, 3, 1), (c,), ln.weight, ln.bias, 1e-6).permute(0, 3, 1, 2)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("n,c,h,w", [(4, 256, 16, 16), (2, 256, 32, 32), (1, 8, 3, 5), (1, 4, 1, 1), (2, 12, 1, 7), (3, 16, 2, 2),
                                     (1, 128, 64, 48)])
def test_upsample_bilinear2x(hip, n, c, h, w):
    torch.manual_seed(0)
    x = _cl(torch.randn(n, c, h, w, device=DEV))
    y = hip.upsample_bilinear2x(x)
    ref = nn.UpsamplingBilinear2d(scale_factor=2)(x)
    assert y.shape == ref.shape
    torch.testing.assert_close(y, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n,c,g,h,w", [(4, 256, 32, 64, 64), (3, 256, 32, 16, 16), (5, 128, 32, 32, 32),
                                        (2, 128, 32, 8, 8)])
@pytest.mark.parametrize("gelu", [True, False])
def test_groupnorm_act(hip, n, c, g, h, w, gelu):
    torch.manual_seed(h)
    gn = nn.GroupNorm(g, c).to(DEV)
    with torch.no_grad():
        gn.weight.normal_(1.0, 0.3)
        gn.bias.normal_(0.0, 0.3)
        x = _cl(torch.randn(n, c, h, w, device=DEV) * 2.0 + 0.7)
        y = hip.groupnorm_act(x, gn.weight, gn.bias, g, gn.eps, gelu=gelu)
        ref = gn(x)
        if gelu:
            ref = F.gelu(ref)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mlp_gemm", ["split", "torch"])
def test_model_forward_hip_layers_vs_torch_ops(hip, mlp_gemm):
    """Whole GDRN_Net forward with the HIP layers on vs off (same weights): maps within 1e-4, pose within 1e-4.
    Both MLP GEMM engines (bf16x6 split on the bf16 matrix cores, fp32 MFMA) against hipBLASLt + PyTorch ops."""
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

    cfg = get_cfg("ycbv_convnext_a6", ["TEST.USE_DEPTH_REFINE=True"])
    torch.manual_seed(0)
    model, _ = build_model_optimizer(cfg)
    with torch.no_grad():
        model.pnp_net.fc_t.bias.copy_(torch.tensor([0.0, 0.0, 4.0]))
        for m in model.modules():  # make layer-scale / out layer non-trivial so differences would show
            if hasattr(m, "gamma") and isinstance(m.gamma, nn.Parameter):
                m.gamma.fill_(0.3)
        nn.init.normal_(model.geo_head_net.out_layer.weight, 0, 0.05)
    b = 6
    x = torch.rand(b, 3, 256, 256, device=DEV)
    cls = torch.randint(0, 21, (b,), device=DEV)
    K = torch.tensor([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]], device=DEV).repeat(b, 1, 1)
    args = dict(roi_classes=cls, roi_cams=K, roi_whs=torch.full((b, 2), 120.0, device=DEV),
                roi_centers=torch.full((b, 2), 250.0, device=DEV), resize_ratios=torch.full((b,), 64 / 180.0, device=DEV),
                roi_coord_2d=torch.rand(b, 2, 64, 64, device=DEV), roi_extents=torch.full((b, 3), 0.1, device=DEV))
    with torch.no_grad():
        hip_layers.set_enabled(True)
        hip_layers.set_mlp_gemm(mlp_gemm)
        try:
            o1 = model(x, **args)
        finally:
            hip_layers.set_mlp_gemm("split")
        hip_layers.set_enabled(False)
        o2 = model(x, **args)
        hip_layers.set_enabled(True)
    for k in ("mask", "coor_x", "coor_y", "coor_z", "region"):
        scale = o2[k].abs().max().item()
        assert (o1[k] - o2[k]).abs().max().item() <= 1e-4 * max(scale, 1.0), k
    torch.testing.assert_close(o1["rot"], o2["rot"], rtol=0, atol=1e-4)
    torch.testing.assert_close(o1["trans"], o2["trans"], rtol=0, atol=1e-4)


def test_pack_weight_bf16x3_is_exact(hip):
    """w == h + m + l exactly for every element (incl. tiny / huge magnitudes), and the tile layout round-trips."""
    torch.manual_seed(3)
    w = torch.randn(256, 96, device=DEV) * torch.logspace(-20, 20, 96, device=DEV)
    w[0, :4] = torch.tensor([0.0, -0.0, 1.0, -1.5], device=DEV)
    planes = hip.unpack_weight_bf16x3(hip.pack_weight_bf16x3(w)).float()
    assert torch.equal((planes[0] + planes[1]) + planes[2], w)
    assert (planes[1].abs() <= planes[0].abs() * 2.0 ** -8 + 1e-45).all()
    assert (planes[2].abs() <= planes[0].abs() * 2.0 ** -16 + 1e-45).all()


@pytest.mark.parametrize("m,k,n", [(128 * 64, 128, 512), (128 * 16, 512, 128), (256, 2048, 512), (128, 32, 128), (128 * 9, 64, 384)])
def test_linear_f32_split_is_fp32_accurate(hip, m, k, n):
    """bf16x6 split GEMM vs an fp64 product: its error must not exceed that of the fp32 GEMM of PyTorch-ROCm (hipBLASLt)
    by more than rounding noise, for all three epilogues; and it agrees with fp32 to 4e-6 of the output scale."""
    torch.manual_seed(m + k + n)
    x = torch.randn(m, k, device=DEV)
    w = torch.randn(n, k, device=DEV) * (k ** -0.5)
    b = torch.randn(n, device=DEV)
    gamma = torch.randn(n, device=DEV)
    res = torch.randn(m, n, device=DEV)
    pk = hip.pack_weight_bf16x3(w)
    ref64 = x.double() @ w.double().t() + b.double()
    ref32 = F.linear(x, w, b)
    cases = (("none", ref64, ref32), ("gelu", F.gelu(ref64), F.gelu(ref32)),
             ("scale_res", res.double() + gamma.double() * ref64, torch.addcmul(res, ref32, gamma)))
    for epi, want64, got32 in cases:
        out = hip.linear_f32_split(x, pk, b, epi, gamma if epi == "scale_res" else None, res if epi == "scale_res" else None)
        scale = want64.abs().max().item()
        e_split = (out.double() - want64).abs().max().item() / scale
        e_f32 = (got32.double() - want64).abs().max().item() / scale
        assert e_split <= max(1.25 * e_f32 + 1.2e-7, 4e-8 * k ** 0.5), (epi, e_split, e_f32)
        assert ((out - got32).abs().max() / scale).item() < 4e-6, epi
    # K must be a multiple of 32 (M is free): the C entry point refuses with a status and a message, nothing is launched
    import ctypes
    rc = hip.load().gdrnpp_linear_f32_split(x.data_ptr(), pk.data_ptr(), None, None, None, x.data_ptr(), m, n, k + 8, 0,
                                            ctypes.c_void_p(0))
    assert rc != 0 and b"multiples" in hip.load().gdrnpp_last_error()


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 256, 256, 32, 32), (8, 64, 128, 16, 16), (1, 32, 128, 8, 16), (4, 96, 256, 64, 64)])
def test_conv3x3_f32_split_vs_fp64(hip, n, cin, cout, h, w):
    """Implicit-GEMM 3x3 convolution (zero pad 1) on the split kernel vs an fp64 convolution: error at the level of
    MIOpen's fp32 convolution; borders (zero padding) included; bias and GELU epilogue."""
    torch.manual_seed(n * cin + h)
    x = torch.randn(n, cin, h, w, device=DEV).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, device=DEV) * (9 * cin) ** -0.5
    b = torch.randn(cout, device=DEV)
    pk = hip.pack_conv3x3_weight_bf16x3(wt)
    ref64 = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    ref32 = F.conv2d(x, wt, b, padding=1)
    out = hip.conv3x3_f32_split(x, pk, b)
    assert out.is_contiguous(memory_format=torch.channels_last) and out.shape == ref32.shape
    scale = ref64.abs().max().item()
    e_split = (out.double() - ref64).abs().max().item() / scale
    e_f32 = (ref32.double() - ref64).abs().max().item() / scale
    bound = lambda e: max(1.5 * e + 1.5e-7, 4e-8 * (9 * cin) ** 0.5)  # k-ordered fp32 accumulation: ~sqrt(K) ulp
    assert e_split <= bound(e_f32), (e_split, e_f32)
    out_nb = hip.conv3x3_f32_split(x, pk, None, gelu=True)
    want = F.gelu(F.conv2d(x.double(), wt.double(), None, padding=1))
    e_f32 = ((F.gelu(F.conv2d(x, wt, None, padding=1)).double() - want).abs().max() / want.abs().max()).item()
    assert ((out_nb.double() - want).abs().max() / want.abs().max()).item() <= bound(e_f32)


@pytest.mark.parametrize("n,c,h,w", [(4, 128, 64, 64), (3, 256, 17, 9), (2, 512, 16, 16), (5, 1024, 3, 3), (2, 32, 8, 8), (1, 8, 5, 7)])
def test_layernorm_nhwc(hip, n, c, h, w):
    torch.manual_seed(c + h)
    x = (torch.randn(n, c, h, w, device=DEV) * 3 + 1.5).contiguous(memory_format=torch.channels_last)
    g = torch.randn(c, device=DEV)
    b = torch.randn(c, device=DEV)
    ref = F.layer_norm(x.permute(0, 2, 3, 1), (c,), g, b, 1e-6).permute(0, 3, 1, 2)
    y = hip.layernorm_nhwc(x, g, b, 1e-6)
    assert y.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)


def test_split_gemm_kernels_are_bitwise_equal(hip):
    """The 256-row kernels — register-staged (A split before the LDS store), LDS-DMA (fp32 A, split at fragment-read time)
    and the software-pipelined LDS-DMA kernel with 2 and 3 A stages (split of k-tile t+1 inside the MFMA stream of k-tile t,
    counted vmcnt) — add the same six partial products per k-step in the same order: identical bits, for the linear, 3x3
    and general convolution forms, ragged M included (the general convolution stays on the first two kernels)."""
    modes = [(0, 0), (1, 0), (1, 2), (1, 3)]   # (split_gemm_glds, split_gemm_pipe)
    try:
        hip.set_option("split_gemm_mi4", 1)            # force 256-row tiles at these small sizes
        outs = []
        for glds, pipe in modes:
            torch.manual_seed(5)
            hip.set_option("split_gemm_glds", glds)
            hip.set_option("split_gemm_pipe", pipe)
            hip.set_option("split_gemm_pipe_conv", 1 if pipe else 0)
            m, k, n = 1024 + 37, 512, 256
            x = torch.randn(m, k, device=DEV)
            w = torch.randn(n, k, device=DEV) * (k ** -0.5)
            b = torch.randn(n, device=DEV)
            g = torch.randn(n, device=DEV)
            r = torch.randn(m, n, device=DEV)
            pk = hip.pack_weight_bf16x3(w)
            xc = torch.randn(3, 64, 16, 24, device=DEV).contiguous(memory_format=torch.channels_last)
            wc = torch.randn(128, 64, 3, 3, device=DEV) * 0.05
            wd = torch.randn(128, 64, 2, 2, device=DEV) * 0.05
            xk = torch.randn(300, 32, device=DEV)          # shortest K (two k-tiles): prologue / clamp paths
            wk = torch.randn(128, 32, device=DEV)
            outs.append((hip.linear_f32_split(x, pk, b, "gelu"), hip.linear_f32_split(x, pk, b, "scale_res", g, r),
                         hip.linear_f32_split(x, pk, None, "none"),
                         hip.conv3x3_f32_split(xc, hip.pack_conv_weight_bf16x3(wc), None),
                         hip.conv3x3_f32_split(xc, hip.pack_conv_weight_bf16x3(wc), b[:128].contiguous(), gelu=True),
                         hip.conv2d_f32_split(xc, hip.pack_conv_weight_bf16x3(wd), None, 2, 2, 2, 0),
                         hip.linear_f32_split(xk, hip.pack_weight_bf16x3(wk), None, "none")))
    finally:
        hip.set_option("split_gemm_mi4", -1)
        hip.set_option("split_gemm_glds", 1)
        hip.set_option("split_gemm_pipe", 3)
        hip.set_option("split_gemm_pipe_conv", 0)
    for other in outs[1:]:
        for a_, b_ in zip(outs[0], other):
            assert torch.equal(a_, b_)
    ref = F.conv2d(xc, wc, None, padding=1)
    assert ((outs[2][3] - ref).abs().max() / ref.abs().max()).item() < 4e-6
    ref = x.double() @ w.double().T
    assert ((outs[3][2].double() - ref).abs().max() / ref.abs().max()).item() < 1e-6


def test_split_gemm_gelu_epilogue_matches_fp64_gelu(hip):
    """The GELU epilogue of the split GEMM against an fp64 GELU of the SAME pre-activations (taken from the
    epilogue-free launch, bit-identical accumulators) over [-7, 7]: as close as PyTorch's fp32 GELU."""
    torch.manual_seed(11)
    m, k, n = 2048, 32, 128
    x = torch.randn(m, k, device=DEV)
    w = torch.randn(n, k, device=DEV) * 0.35   # pre-activations ~ N(0, 2^2): covers both erf branches and the tails
    b = torch.linspace(-3, 3, n, device=DEV)
    pk = hip.pack_weight_bf16x3(w)
    pre = hip.linear_f32_split(x, pk, b, "none")
    act = hip.linear_f32_split(x, pk, b, "gelu")
    assert pre.abs().max().item() > 6.0 and (pre.abs() < 0.5).float().mean().item() > 0.05
    want = F.gelu(pre.double())
    err = (act.double() - want).abs()
    ref32 = (F.gelu(pre).double() - want).abs()
    assert err.max().item() <= max(1.5 * ref32.max().item(), 6e-7), (err.max().item(), ref32.max().item())
    assert (err / pre.double().abs().clamp_min(1.0)).max().item() < 2.5e-7


@pytest.mark.parametrize("m", [1, 7, 100, 129, 448, 257 * 3])
def test_linear_f32_split_any_row_count(hip, m):
    """ROI counts are arbitrary in production: rows that do not fill the last 128/256-row tile are clamped on load and
    masked on store.  Result == the same rows computed inside a padded, tile-aligned problem (bitwise), and nothing is
    written past row M (canary rows stay intact)."""
    torch.manual_seed(m)
    k, n = 256, 384
    w = torch.randn(n, k, device=DEV) * (k ** -0.5)
    b = torch.randn(n, device=DEV)
    gamma = torch.randn(n, device=DEV)
    pk = hip.pack_weight_bf16x3(w)
    mp = (m + 255) // 256 * 256
    xp = torch.randn(mp, k, device=DEV)
    rp = torch.randn(mp, n, device=DEV)
    for epi in ("none", "gelu", "scale_res"):
        full = hip.linear_f32_split(xp, pk, b, epi, *( (gamma, rp) if epi == "scale_res" else ()))
        part = hip.linear_f32_split(xp[:m].contiguous(), pk, b, epi, *((gamma, rp[:m].contiguous()) if epi == "scale_res" else ()))
        assert part.shape == (m, n) and torch.equal(part, full[:m]), epi
    # canary: the output buffer is exactly m rows inside a larger allocation
    import ctypes
    big = torch.full((m + 4, n), 777.0, device=DEV)
    xs = xp[:m].contiguous()
    rc = hip.load().gdrnpp_linear_f32_split(xs.data_ptr(), pk.data_ptr(), b.data_ptr(), None, None, big.data_ptr(), m, n, k, 0,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(big[m:], torch.full((4, n), 777.0, device=DEV))
    assert torch.equal(big[:m], hip.linear_f32_split(xs, pk, b, "none"))


def test_conv3x3_f32_split_any_pixel_count(hip):
    """3 images of 5x7 pixels (105 rows, not a multiple of 128): same as PyTorch's convolution to fp32 noise."""
    torch.manual_seed(2)
    x = torch.randn(3, 64, 5, 7, device=DEV).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(128, 64, 3, 3, device=DEV) * 0.04
    out = hip.conv3x3_f32_split(x, hip.pack_conv3x3_weight_bf16x3(wt), None)
    ref = F.conv2d(x.double(), wt.double(), None, padding=1)
    assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 1e-6


def test_model_forward_odd_roi_count(hip):
    """B = 5 ROIs: every ConvNeXt stage and the head run through the split GEMM (stage 3 has 320 rows); maps and pose
    agree with the PyTorch operators to 1e-4."""
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

    cfg = get_cfg("ycbv_convnext_a6", ["TEST.USE_DEPTH_REFINE=True"])
    torch.manual_seed(1)
    model, _ = build_model_optimizer(cfg)
    with torch.no_grad():
        model.pnp_net.fc_t.bias.copy_(torch.tensor([0.0, 0.0, 4.0]))
    b = 5
    x = torch.rand(b, 3, 256, 256, device=DEV)
    K = torch.tensor([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]], device=DEV).repeat(b, 1, 1)
    args = dict(roi_classes=torch.randint(0, 21, (b,), device=DEV), roi_cams=K, roi_whs=torch.full((b, 2), 120.0, device=DEV),
                roi_centers=torch.full((b, 2), 250.0, device=DEV), resize_ratios=torch.full((b,), 64 / 180.0, device=DEV),
                roi_coord_2d=torch.rand(b, 2, 64, 64, device=DEV), roi_extents=torch.full((b, 3), 0.1, device=DEV))
    timer, timer_lib = hip.LaunchTimer(), hip.LaunchTimer()
    with torch.no_grad():
        hip.set_launch_timer(timer)                # default dispatch: every block on this library's GEMM, whatever the ROI count
        try:
            o1 = model(x, **args)
        finally:
            hip.set_launch_timer(None)
        hip.set_launch_timer(timer_lib)            # A/B switch of round 2: deep stages (few tiles at 5 ROIs) on hipBLASLt
        hip_layers.set_library_below_tiles(128)
        try:
            o3 = model(x, **args)
        finally:
            hip.set_launch_timer(None)
            hip_layers.set_library_below_tiles(0)
        hip_layers.set_enabled(False)
        o2 = model(x, **args)
        hip_layers.set_enabled(True)
    # 36 blocks x (fc1, fc2) + the two Patch-PnP fc layers, plain or split-K depending on the tile count
    assert sum(1 for r in timer.records if r[0] in ("linear", "linear_splitk", "linear_x3")) == 74
    assert sum(1 for r in timer.records if r[0] in ("conv3x3", "conv_splitk")) >= 4
    assert sum(1 for r in timer_lib.records if r[0] in ("linear", "linear_splitk", "linear_x3")) == 3 * 2 + 2   # stage 0 + Patch-PnP fc
    torch.testing.assert_close(o3["trans"], o1["trans"], rtol=0, atol=1e-4)
    for key in ("mask", "coor_x", "coor_y", "coor_z", "region"):
        assert (o1[key] - o2[key]).abs().max().item() <= 1e-4 * max(o2[key].abs().max().item(), 1.0), key
    torch.testing.assert_close(o1["rot"], o2["rot"], rtol=0, atol=1e-4)
    torch.testing.assert_close(o1["trans"], o2["trans"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("m,k,n", [(128, 8192, 1024), (5, 8192, 1024), (300, 1024, 256), (64, 96, 128), (2048, 2048, 512), (2048, 512, 2048),
                                   (512, 1024, 4096), (512, 4096, 1024), (4096, 2048, 512), (8192, 2048, 512), (200, 2048, 512), (256, 512, 2048)])
def test_linear_f32_splitk(hip, m, k, n):
    """Split-K form (Patch-PnP fc layers, deep ConvNeXt stages at small ROI counts): vs fp64 as accurate as the fp32 GEMM,
    all three epilogues; deterministic across runs."""
    torch.manual_seed(k + m)
    x = torch.randn(m, k, device=DEV)
    w = torch.randn(n, k, device=DEV) * (k ** -0.5)
    b = torch.randn(n, device=DEV)
    gamma = torch.randn(n, device=DEV)
    res = torch.randn(m, n, device=DEV)
    pk = hip.pack_weight_bf16x3(w)
    ref64 = x.double() @ w.double().t() + b.double()
    ref32 = F.linear(x, w, b)
    for epi, want64, got32 in (("none", ref64, ref32), ("gelu", F.gelu(ref64), F.gelu(ref32)),
                               ("scale_res", res.double() + gamma.double() * ref64, torch.addcmul(res, ref32, gamma))):
        extra = (gamma, res) if epi == "scale_res" else ()
        out = hip.linear_f32_splitk(x, pk, b, epi, *extra)
        scale = want64.abs().max().item()
        e_f32 = (got32.double() - want64).abs().max().item() / scale
        # floor: the library may pick a more accurate algorithm on another box; ~sqrt(K) ulp is what fp32 accumulation allows
        assert (out.double() - want64).abs().max().item() / scale <= max(1.25 * e_f32 + 1.2e-7, 4e-8 * k ** 0.5), epi
        assert torch.equal(out, hip.linear_f32_splitk(x, pk, b, epi, *extra))


@pytest.mark.parametrize("n,cin,cout,h,w,k,stride,pad", [(4, 128, 256, 32, 32, 2, 2, 0), (3, 128, 128, 17, 20, 3, 2, 1), (2, 64, 128, 9, 9, 3, 1, 0),
                                                          (2, 32, 128, 12, 16, 4, 4, 0), (1, 64, 128, 10, 11, 5, 2, 2)])
def test_conv2d_f32_split_general(hip, n, cin, cout, h, w, k, stride, pad):
    """KxK / stride / zero-pad implicit GEMM (ConvNeXt 2x2/2 downsamples, Patch-PnP 3x3/2, ...) vs an fp64 convolution."""
    torch.manual_seed(cin + k + stride)
    x = torch.randn(n, cin, h, w, device=DEV).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, k, k, device=DEV) * (k * k * cin) ** -0.5
    b = torch.randn(cout, device=DEV)
    out = hip.conv2d_f32_split(x, hip.pack_conv_weight_bf16x3(wt), b, k, k, stride, pad)
    ref64 = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad)
    ref32 = F.conv2d(x, wt, b, stride=stride, padding=pad)
    assert out.shape == ref32.shape and out.is_contiguous(memory_format=torch.channels_last)
    scale = ref64.abs().max().item()
    e_split = (out.double() - ref64).abs().max().item() / scale
    e_f32 = (ref32.double() - ref64).abs().max().item() / scale
    assert e_split <= max(1.5 * e_f32 + 1.5e-7, 4e-8 * (k * k * cin) ** 0.5), (e_split, e_f32)


@pytest.mark.parametrize("n,h,w", [(3, 256, 256), (2, 64, 96)])
def test_stem_conv_ln_fused_vs_torch(hip, n, h, w):
    """Fused ConvNeXt stem (4x4/4 convolution from 3 channels + bias + LayerNorm2d, NCHW in / NHWC out) against
    F.conv2d + F.layer_norm in fp64."""
    torch.manual_seed(7)
    x = torch.rand(n, 3, h, w, device=DEV)
    wt = torch.randn(128, 3, 4, 4, device=DEV) * 0.2
    b = torch.randn(128, device=DEV) * 0.1
    g = torch.rand(128, device=DEV) + 0.5
    be = torch.randn(128, device=DEV) * 0.1
    y = hip.stem_conv4x4_ln(x, wt, b, g, be, 1e-6)
    assert y.shape == (n, 128, h // 4, w // 4) and y.is_contiguous(memory_format=torch.channels_last)
    c = F.conv2d(x.double(), wt.double(), b.double(), stride=4)
    ref = F.layer_norm(c.permute(0, 2, 3, 1), (128,), g.double(), be.double(), 1e-6).permute(0, 3, 1, 2)
    assert (y.double() - ref).abs().max().item() < 2e-5
    y0 = hip.stem_conv4x4_ln(x, wt, None, g, be, 1e-6)
    c0 = F.conv2d(x.double(), wt.double(), None, stride=4)
    ref0 = F.layer_norm(c0.permute(0, 2, 3, 1), (128,), g.double(), be.double(), 1e-6).permute(0, 3, 1, 2)
    assert (y0.double() - ref0).abs().max().item() < 2e-5


@pytest.mark.parametrize("n,h,w,gelu,bias", [(16, 64, 64, True, False), (64, 32, 32, False, True), (65, 32, 32, True, True)])
def test_conv3x3_groupnorm_fused_vs_torch(hip, n, h, w, gelu, bias):
    """conv3x3 -> GroupNorm(32) (-> GELU) with the statistics from the convolution's epilogue: against F.conv2d +
    F.group_norm (+ gelu) in fp64, and bit-for-bit conv output / 1e-6 norm output against the two separate HIP layers."""
    torch.manual_seed(n + h)
    x = _cl(torch.randn(n, 256, h, w, device=DEV))
    wt = torch.randn(256, 256, 3, 3, device=DEV) * 0.03
    b = torch.randn(256, device=DEV) * 0.1 if bias else None
    g = torch.rand(256, device=DEV) + 0.5
    be = torch.randn(256, device=DEV) * 0.1
    pk = hip.pack_conv_weight_bf16x3(wt)
    y = hip.conv3x3_groupnorm_act(x, pk, b, g, be, 32, 1e-5, gelu=gelu)
    assert y is not None and y.shape == (n, 256, h, w) and y.is_contiguous(memory_format=torch.channels_last)
    c = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), padding=1)
    ref = F.group_norm(c, 32, g.double(), be.double(), 1e-5)
    if gelu:
        ref = F.gelu(ref)
    assert (y.double() - ref).abs().max().item() < 3e-5
    two = hip.groupnorm_act(hip.conv3x3_f32_split(x, pk, b), g, be, 32, 1e-5, gelu=gelu)
    assert (y - two).abs().max().item() < 1e-6
    # shapes outside the fused form are declined, not mis-computed
    assert hip.conv3x3_groupnorm_act(_cl(torch.randn(2, 256, 8, 8, device=DEV)), pk, b, g, be, 32) is None
    assert hip.conv3x3_groupnorm_act(x, pk, b, g[:64].repeat(4), be, 16) is None


@pytest.mark.parametrize("ks,pad,out_pad", [(3, 1, 1), (4, 1, 0), (2, 0, 0)])
@pytest.mark.parametrize("n,h,w,cin,cout,bias", [(5, 8, 8, 1024, 256, False), (2, 5, 7, 64, 128, True)])
def test_conv_transpose2d_split_vs_torch(hip, ks, pad, out_pad, n, h, w, cin, cout, bias):
    """ConvTranspose2d (stride 2; the three kernel sizes of the head's deconv table) as split GEMM + col2im gather against
    F.conv_transpose2d in fp64."""
    torch.manual_seed(ks + n)
    x = _cl(torch.randn(n, cin, h, w, device=DEV))
    wt = torch.randn(cin, cout, ks, ks, device=DEV) * 0.03
    b = torch.randn(cout, device=DEV) * 0.1 if bias else None
    y = hip.conv_transpose2d_f32_split(x, hip.pack_deconv_weight_bf16x3(wt), b, ks, 2, pad, out_pad)
    ref = F.conv_transpose2d(x.double(), wt.double(), None if b is None else b.double(), stride=2, padding=pad, output_padding=out_pad)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert (y.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_resnet_bn_folded_path_vs_module_path(hip):
    """ResNet-34 features (config 1's backbone) in inference mode: BatchNorms folded into the convolutions + one
    bias/shortcut/ReLU kernel per convolution (gdrnpp_bias_act_nhwc) against the plain module path (BatchNorm2d, add, ReLU),
    with non-trivial running statistics and affine parameters; and the kernel itself against torch."""
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling.backbones import create_backbone

    torch.manual_seed(11)
    x = _cl(torch.randn(3, 64, 9, 7, device=DEV))
    r = _cl(torch.randn(3, 64, 9, 7, device=DEV))
    b = torch.randn(64, device=DEV)
    for resid, relu in [(None, True), (r, True), (r, False)]:
        want = x + b.view(1, -1, 1, 1) + (0 if resid is None else resid)
        want = torch.relu(want) if relu else want
        got = hip.bias_act_nhwc_(x.clone(memory_format=torch.channels_last), b, resid, relu)
        assert torch.equal(got, want)
    net = create_backbone("timm/resnet34", out_indices=(4,)).to(DEV).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.2)
        img = torch.randn(4, 3, 256, 256, device=DEV)
        a = net(img)[0]
        hip_layers.set_enabled(False)
        try:
            ref = net(img)[0]
        finally:
            hip_layers.set_enabled(True)
    assert a.shape == ref.shape == (4, 512, 8, 8)
    assert (a - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("b", [1, 17])
def test_model_forward_more_roi_counts(hip, b):
    """Whole forward (fused stem, pipelined / small-tile GEMMs, grouped output layer, head tail, Patch-PnP) on the HIP path
    against the PyTorch-operator path at ROI counts that leave ragged row tiles everywhere (1 ROI: 64 rows at stage 3)."""
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

    cfg = get_cfg("tless_convnext_a6", ["TEST.USE_DEPTH_REFINE=True"])
    torch.manual_seed(2)
    model, _ = build_model_optimizer(cfg)
    with torch.no_grad():
        model.pnp_net.fc_t.bias.copy_(torch.tensor([0.0, 0.0, 4.0]))
    x = torch.rand(b, 3, 256, 256, device=DEV)
    K = torch.tensor([[1075.65, 0, 360.0], [0, 1073.9, 270.0], [0, 0, 1]], device=DEV).repeat(b, 1, 1)
    args = dict(roi_classes=torch.randint(0, 30, (b,), device=DEV), roi_cams=K, roi_whs=torch.full((b, 2), 110.0, device=DEV),
                roi_centers=torch.full((b, 2), 300.0, device=DEV), resize_ratios=torch.full((b,), 64 / 165.0, device=DEV),
                roi_coord_2d=torch.rand(b, 2, 64, 64, device=DEV), roi_extents=torch.rand(b, 3, device=DEV) * 0.2 + 0.05)
    with torch.no_grad():
        o1 = model(x, **args)
        hip_layers.set_enabled(False)
        try:
            o2 = model(x, **args)
        finally:
            hip_layers.set_enabled(True)
    for key in ("mask", "full_mask", "coor_x", "coor_y", "coor_z", "region"):
        assert o1[key].shape == o2[key].shape
        assert (o1[key] - o2[key]).abs().max().item() <= 1e-4 * max(o2[key].abs().max().item(), 1.0), key
    torch.testing.assert_close(o1["rot"], o2["rot"], rtol=0, atol=1e-4)
    torch.testing.assert_close(o1["trans"], o2["trans"], rtol=0, atol=1e-4)



@pytest.mark.parametrize("n,cin,cout,h,w,k,stride,pad,gelu", [(8, 256, 256, 16, 16, 3, 1, 1, False), (8, 256, 256, 32, 32, 3, 1, 1, True),
                                                              (3, 96, 128, 64, 64, 3, 2, 1, False), (5, 128, 128, 17, 19, 3, 1, 1, False),
                                                              (2, 256, 256, 8, 8, 2, 2, 0, True)])
def test_conv_splitk_small_batches(hip, n, cin, cout, h, w, k, stride, pad, gelu):
    """Few output tiles (the head's small maps / Patch-PnP at the reference's own batch sizes): the split-K form of the implicit
    GEMM vs an fp64 convolution at the usual bar, deterministic, and really taken (a workspace is requested) for these shapes."""
    torch.manual_seed(n + h + k)
    x = _cl(torch.randn(n, cin, h, w, device=DEV))
    wt = torch.randn(cout, cin, k, k, device=DEV) * (k * k * cin) ** -0.5
    b = torch.randn(cout, device=DEV)
    pk = hip.pack_conv_weight_bf16x3(wt)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    assert hip.load().gdrnpp_conv2d_f32_splitk_workspace_bytes(n, oh, ow, cin, cout, k, k) > 0
    out = hip.conv2d_f32_split(x, pk, b, k, k, stride, pad, gelu)
    ref64 = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad)
    ref32 = F.conv2d(x, wt, b, stride=stride, padding=pad)
    if gelu:
        ref64, ref32 = F.gelu(ref64), F.gelu(ref32)
    scale = ref64.abs().max().item()
    e_split = (out.double() - ref64).abs().max().item() / scale
    e_f32 = (ref32.double() - ref64).abs().max().item() / scale
    assert e_split <= max(1.5 * e_f32 + 1.5e-7, 4e-8 * (k * k * cin) ** 0.5), (e_split, e_f32)
    assert torch.equal(out, hip.conv2d_f32_split(x, pk, b, k, k, stride, pad, gelu))
    hip.set_conv_splitk(False)
    try:
        one = hip.conv2d_f32_split(x, pk, b, k, k, stride, pad, gelu)
    finally:
        hip.set_conv_splitk(True)
    assert ((out - one).abs().max() / scale).item() < 2e-6


@pytest.mark.parametrize("m,k,n", [(1000, 512, 512), (2048, 512, 2048), (128, 2048, 512), (96, 64, 128), (4100, 128, 256)])
def test_pipelined_128_row_kernel_is_bitwise_equal_to_the_256_row_kernel(hip, m, k, n):
    """Launches with fewer than 256 tiles of 256x128 run the 128-row form of the pipelined kernel: same six products in the
    same order per accumulator, so every epilogue agrees bit for bit with the 256-row kernel (forced by split_gemm_mi4 = 1)."""
    torch.manual_seed(m + k)
    x = torch.randn(m, k, device=DEV)
    w = torch.randn(n, k, device=DEV) * k ** -0.5
    b = torch.randn(n, device=DEV)
    g = torch.randn(n, device=DEV)
    r = torch.randn(m, n, device=DEV)
    pk = hip.pack_weight_bf16x3(w)
    for epi in ("none", "gelu", "scale_res"):
        extra = (g, r) if epi == "scale_res" else ()
        small = hip.linear_f32_split(x, pk, b, epi, *extra)
        hip.set_option("split_gemm_mi4", 1)
        try:
            big = hip.linear_f32_split(x, pk, b, epi, *extra)
        finally:
            hip.set_option("split_gemm_mi4", -1)
        assert torch.isfinite(small).all() and torch.equal(small, big), epi


@pytest.mark.parametrize("ks,pad,out_pad", [(3, 1, 1), (4, 1, 0), (2, 0, 0)])
@pytest.mark.parametrize("n,h,w,cin,cout,groups,bias,gelu", [(5, 8, 8, 1024, 256, 32, False, True), (3, 5, 7, 64, 128, 32, True, False),
                                                          (2, 16, 16, 128, 256, 32, False, True)])
def test_conv_transpose2d_groupnorm_fused_is_bitwise_the_two_pass_result(hip, ks, pad, out_pad, n, h, w, cin, cout, groups, bias, gelu):
    """ConvTranspose2d -> GroupNorm (-> GELU) with the statistics taken by the col2im gather (gdrnpp_deconv_col2im_gn_nhwc): the
    partials are those of gdrnpp_groupnorm_act_nhwc's own statistics pass (same partition, same order), so the result equals
    conv_transpose2d_f32_split + groupnorm_act bit for bit; and both sit at fp32 rounding from the fp64 layers."""
    torch.manual_seed(ks + n + cout)
    x = torch.randn(n, cin, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cin, cout, ks, ks, device="cuda") * cin ** -0.5
    b = torch.randn(cout, device="cuda") if bias else None
    gw, gb = torch.randn(cout, device="cuda"), torch.randn(cout, device="cuda")
    pk = hip.pack_deconv_weight_bf16x3(wt)
    fused = hip.conv_transpose2d_groupnorm_act(x, pk, b, ks, 2, pad, out_pad, gw, gb, groups, 1e-5, gelu=gelu)
    two = hip.groupnorm_act(hip.conv_transpose2d_f32_split(x, pk, b, ks, 2, pad, out_pad), gw, gb, groups, 1e-5, gelu=gelu)
    assert fused.is_contiguous(memory_format=torch.channels_last) and torch.equal(fused, two)
    ref = F.group_norm(F.conv_transpose2d(x.double(), wt.double(), None if b is None else b.double(), stride=2, padding=pad,
                                          output_padding=out_pad), groups, gw.double(), gb.double(), 1e-5)
    if gelu:
        ref = F.gelu(ref)
    assert ((fused.double() - ref).abs().max() / ref.abs().max()).item() < 5e-6


@pytest.mark.parametrize("b,k,rot_dim", [(128, 256, 6), (1, 256, 4), (7, 1000, 3), (33, 64, 9)])
def test_pnp_fc_heads_vs_fp64(hip, b, k, rot_dim):
    """fc_r | fc_t of Patch-PnP in one launch (conv_pnp_net.py:99-101,178-182) against the two nn.Linear in fp64, and against
    torch's fp32 layers at fp32 rounding."""
    torch.manual_seed(b + k)
    x = torch.randn(b, k, device="cuda")
    fc_r, fc_t = torch.nn.Linear(k, rot_dim).cuda(), torch.nn.Linear(k, 3).cuda()
    with torch.no_grad():
        fc_t.bias.add_(1.5)
        r, t = hip.pnp_fc_heads(x, fc_r.weight, fc_r.bias, fc_t.weight, fc_t.bias)
        r64, t64 = fc_r.double()(x.double()), fc_t.double()(x.double())
    assert r.shape == (b, rot_dim) and t.shape == (b, 3)
    assert (r.double() - r64).abs().max().item() < 2e-6 * max(1.0, r64.abs().max().item())
    assert (t.double() - t64).abs().max().item() < 2e-6 * max(1.0, t64.abs().max().item())
    r0, t0 = hip.pnp_fc_heads(x, fc_r.weight.float(), None, fc_t.weight.float(), None)     # bias is optional
    assert (r0.double() + fc_r.bias.double() - r64).abs().max().item() < 2e-6 * max(1.0, r64.abs().max().item())


def test_patch_pnp_tail_runs_on_this_library(hip):
    """The GDRNPP Patch-PnP tail (fc1 -> GELU -> fc2 -> GELU -> fc_r | fc_t) leaves no vendor-library launch: two split-K GEMMs with
    the GELU in their epilogues + gdrnpp_pnp_fc_heads; same numbers as the module graph."""
    from gdrnpp_bop2022_amd.gdrn_modeling.heads import ConvPnPNet
    torch.manual_seed(4)
    net = ConvPnPNet(nIn=69, featdim=128, rot_dim=6, num_regions=64, act="gelu").cuda().eval()
    for m in (net.fc1, net.fc2, net.fc_r, net.fc_t):
        torch.nn.init.normal_(m.weight, 0.0, m.in_features ** -0.5)
        torch.nn.init.normal_(m.bias, 0.0, 0.3)
    feat = torch.randn(128, 128, 8, 8, device="cuda")
    timer = hip.LaunchTimer()
    hip.set_launch_timer(timer)
    try:
        with torch.no_grad(), torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            r, t = net._fc_tail(feat)
            torch.cuda.synchronize()
    finally:
        hip.set_launch_timer(None)
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert [k for k in (rec[0] for rec in timer.records)] == ["linear_splitk", "linear_splitk"]
    assert not [n for n in names if n.startswith("Cijk") or "Gelu" in n or "gelu" in n], names
    with torch.no_grad():
        x = feat.flatten(1).double()
        h = F.gelu(net.fc2.double()(F.gelu(net.fc1.double()(x))))
        r64, t64 = net.fc_r.double()(h), net.fc_t.double()(h)
    assert (r.double() - r64).abs().max().item() < 5e-6 * r64.abs().max().item()
    assert (t.double() - t64).abs().max().item() < 5e-6 * max(1.0, t64.abs().max().item())


@pytest.mark.parametrize("rot_mode,t_mode,allo", [("rot6d", "centroid_z_rel", True), ("quat", "trans", False), ("lie_vec", "centroid_z_abs", True),
                                                  ("log_quat", "centroid_z_abs_z", True)])
def test_pnp_fc_heads_pose_equals_the_two_launches(hip, rot_mode, t_mode, allo):
    """gdrnpp_pnp_fc_heads_pose = gdrnpp_pnp_fc_heads followed by gdrnpp_pose_from_pred, bit for bit, for the ROT_TYPE / TRANS_TYPE
    families of GDRN_double_mask.py:162-200."""
    torch.manual_seed(3)
    b, k = 37, 256
    rot_dim = {"rot6d": 6, "quat": 4, "log_quat": 3, "lie_vec": 3}[rot_mode]
    x = torch.randn(b, k, device="cuda")
    w_r, b_r = torch.randn(rot_dim, k, device="cuda") * 0.1, torch.randn(rot_dim, device="cuda")
    w_t, b_t = torch.randn(3, k, device="cuda") * 0.05, torch.tensor([0.0, 0.0, 1.2], device="cuda")
    cams = torch.tensor([[1066.8, 0, 312.9, 0, 1067.5, 241.3, 0, 0, 1.0]], device="cuda").repeat(b, 1)
    centers = torch.rand(b, 2, device="cuda") * 400 + 100
    whs = torch.rand(b, 2, device="cuda") * 100 + 40
    rr = torch.rand(b, device="cuda") * 0.5 + 0.2
    need = t_mode.startswith("centroid_z") and t_mode != "centroid_z_abs"
    kw = dict(centers=centers if need else None, whs=whs if need else None, resize_ratios=rr if t_mode == "centroid_z_rel" else None,
              rot_mode=rot_mode, t_mode=t_mode, is_allo=allo)
    r1, t1 = hip.pnp_fc_heads(x, w_r, b_r, w_t, b_t)
    R1, T1 = hip.pose_from_pred(r1, t1, cams, **kw)
    r2, t2, R2, T2 = hip.pnp_fc_heads_pose(x, w_r, b_r, w_t, b_t, cams, **kw)
    assert torch.equal(r1, r2) and torch.equal(t1, t2) and torch.equal(R1, R2) and torch.equal(T1, T2)
    assert torch.isfinite(R2).all() and (R2 @ R2.transpose(1, 2) - torch.eye(3, device="cuda")).abs().max().item() < 1e-5
