"""Where the error of the three-product GEMM comes from: kernel output vs (a) fp64, (b) an fp64 evaluation of the SAME three products
of the same fp16 operands (isolates fp32 accumulation inside the MFMA chain), per K."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gdrnpp_bop2022_amd import hip_lib as hip  # noqa: E402

torch.manual_seed(0)
for m, k, n in [(2048, 128, 512), (2048, 512, 128), (2048, 1024, 256), (2048, 2048, 512), (2048, 4096, 1024)]:
    x = torch.randn(m, k, device="cuda")
    x[:, :3] *= 40.0
    w = torch.randn(n, k, device="cuda") * k ** -0.5
    pk = hip.pack_weight_f16x2(w)
    planes, inv = hip.unpack_weight_f16x2(pk)
    out3 = hip.linear_f32_split(x, pk, None)
    out6 = hip.linear_f32_split(x, hip.pack_weight_bf16x3(w), None)
    want = x.double() @ w.double().t()
    h = x.half()
    l = (x - h.float()).half()
    D = lambda a, b: a.double() @ b.double().t()  # noqa: E731
    emu = (D(h, planes[1]) + D(l, planes[0]) + D(h, planes[0])) * inv
    s = want.abs().max()
    f32 = torch.nn.functional.linear(x, w)
    print(f"K={k:5d} N={n:5d}: kernel-fp64 {((out3.double() - want).abs().max() / s).item():.2e}  emulation-fp64 "
          f"{((emu - want).abs().max() / s).item():.2e}  kernel-emulation {((out3.double() - emu).abs().max() / s).item():.2e}  "
          f"six products {((out6.double() - want).abs().max() / s).item():.2e}  torch fp32 {((f32.double() - want).abs().max() / s).item():.2e}")


def network_level():
    """The reference's recorded forward (tests/golden/net_golden_*.npz, 4 ROIs, fp32 PyTorch on CPU) against this library's three
    GEMM engines on the same parameters: six products, three products (every eligible layer forced onto them), and the vendor
    libraries' fp32 kernels (hip_layers off: MIOpen / hipBLASLt)."""
    import numpy as np

    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer
    from tests import netgolden as NG

    for ds in ("ycbv", "tless", "ycbvso"):
        fx = NG.load_fixture(ds)
        model, _ = build_model_optimizer(get_cfg(NG.cfg_name(ds), opts=["TEST.USE_DEPTH_REFINE=True"]))
        model.load_state_dict(NG.seeded_reference_state_dict(model, fx), strict=True)
        x = torch.from_numpy(NG.net_image()).cuda()
        kw = NG.forward_kwargs(fx, "cuda")
        for name in ("six products", "three products", "vendor fp32"):
            hip_layers.set_enabled(name != "vendor fp32")
            hip_layers.set_gemm_products(3 if name == "three products" else 6)
            hip.SPLIT2_MIN_TILES = 1 if name == "three products" else 256
            with torch.no_grad():
                out = {k: v.cpu().numpy().astype(np.float64) for k, v in model(x, **kw).items()}
            e = {k: np.abs(out[k] - fx[k]).max() / np.abs(fx[k]).max() for k in ("mask", "full_mask", "coor_x", "coor_y", "coor_z")}
            e["region"] = np.abs(out["region"][:, :, 1::4, 2::4] - fx["region_sub"]).max() / float(fx["region_absmax"])
            print(f"{ds:7s} {name:15s} maps (of scale): " + "  ".join(f"{k} {v:.1e}" for k, v in e.items()) +
                  f"   rot {np.abs(out['rot'] - fx['rot']).max():.1e}  trans {np.abs(out['trans'] - fx['trans']).max():.1e}")
        hip_layers.set_enabled(True)
        hip_layers.set_gemm_products(6)
        hip.SPLIT2_MIN_TILES = 256


network_level()
