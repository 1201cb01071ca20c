"""Outputs of this library's network path on the iteration-size fixtures (tests/golden/net_golden_{tless_b1024,ycbv_b512}.npz), for the
analysis of profiles/r06_large_parity.md: R, t and the Patch-PnP outputs of every ROI under the default arithmetic (three products)
and under the exact six-product form -> gpurun_out/large_parity_<ds>_b<b>.npz.   python tools/large_parity_dump.py tless 1024 [out_dir]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gdrnpp_bop2022_amd import hip_lib  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer  # noqa: E402
from tests import netgolden as NG  # noqa: E402

ds, b = sys.argv[1], int(sys.argv[2])
out_dir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out")
hip_lib.load()
fx = NG.load_fixture(f"{ds}_b{b}")
cfg = get_cfg(NG.cfg_name(ds), opts=["TEST.USE_DEPTH_REFINE=True"])
model, _ = build_model_optimizer(cfg)
model.load_state_dict(NG.seeded_reference_state_dict(model, fx), strict=True)
x = torch.from_numpy(NG.net_image(b, int(fx["image_seed"]))).cuda()
kw = NG.forward_kwargs(fx, "cuda")
rec = {}


def run(tag):
    with torch.no_grad():
        out = model(x, **kw)
        rot_, t_, _ = model.forward_maps(x, kw["roi_classes"], kw["roi_coord_2d"], None, kw["roi_extents"])
    rec.update({f"rot_{tag}": out["rot"].float().cpu().numpy(), f"trans_{tag}": out["trans"].float().cpu().numpy(),
                f"pred_rot__{tag}": rot_.cpu().numpy(), f"pred_t__{tag}": t_.cpu().numpy()})
    print(tag, "range words", hip_lib.split2_range_words(), flush=True)


run("x3")
with hip_layers.forced_gemm_products(6):
    run("x6")
hip_layers.set_fused_mlp_x3(False)
run("x3_unfused")
hip_layers.set_fused_mlp_x3(True)
os.makedirs(out_dir, exist_ok=True)
np.savez_compressed(os.path.join(out_dir, f"large_parity_{ds}_b{b}.npz"), **rec)
from tests.test_gpu_net_golden import check_iteration_size_outputs  # noqa: E402

lines = []
for tag, name in (("x3", "default path (three products, fused stage-0/1 MLPs)"), ("x6", "exact six-product form"), ("x3_unfused", "three products, MLPs unfused")):
    got = {k: rec[f"{k}_{tag}"] for k in ("rot", "trans", "pred_rot_", "pred_t_")}
    try:
        rep = check_iteration_size_outputs(fx, got, b, name)
        lines += [f"== {name}: passes =="] + rep
    except AssertionError as e:
        lines += [f"== {name}: FAILS == {e}"]
rep = check_iteration_size_outputs(fx, {k: fx[k] for k in ("rot", "trans", "pred_rot_", "pred_t_")}, b, "the reference's own fp32 forward")
lines += ["== the reference's own fp32 forward through the same bars: passes =="] + rep
text = "\n".join(lines)
print(text)
with open(os.path.join(out_dir, f"large_parity_{ds}_b{b}_all_forms.txt"), "w") as f:
    f.write(text + "\n")
