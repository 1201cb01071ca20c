"""Records and pre-refine network outputs under the shared-chip tile rule (three-product kernels from 128 tiles) against the one-stream
rule and against the exact six-product form, per column / per ROI."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from gdrnpp_bop2022_amd import hip_lib
from gdrnpp_bop2022_amd.gdrn_modeling import engine as E, hip_layers

args = B.parse(["--steps", "4", "--no-cpu-baseline", "--no-pmc"] + sys.argv[1:])
dev = torch.device("cuda", 0)
b = args.batch or 128
state = B.build_state(args, ["ycbv_convnext_a6"], True, "refine", b, 0, dev, 0)
m = [o for o in gc.get_objects() if isinstance(o, dict) and "model" in o and "batches" in o and "post" in o][0]
model, post, batches = m["model"], m["post"], m["batches"]
def run(rule, products=3):
    hip_lib.SPLIT2_SHARED_MIN_TILES = rule
    with hip_layers.forced_gemm_products(products):
        recs, nets = [], []
        for bt in batches:
            recs.append(E.inference_step(model, post, bt).clone())
            with torch.no_grad():
                out = model(bt["roi_img"], roi_classes=bt["roi_cls"], roi_cams=bt["roi_cam"], roi_whs=bt["roi_wh"], roi_centers=bt["roi_center"],
                            resize_ratios=bt["resize_ratio"], roi_coord_2d=bt.get("roi_coord_2d"), roi_extents=bt.get("roi_extent"))
            nets.append((out["rot"].clone().to(dev), out["trans"].clone()))
    hip_lib.SPLIT2_SHARED_MIN_TILES = 0
    return recs, nets
r_off, n_off = run(0); r_on, n_on = run(128); r_6, n_6 = run(0, 6)
for name, (ra, na), (rb, nb) in (("rule on vs rule off", (r_on, n_on), (r_off, n_off)), ("rule on vs six products", (r_on, n_on), (r_6, n_6)), ("rule off vs six products", (r_off, n_off), (r_6, n_6))):
    for k in range(2):
        d = (ra[k] - rb[k]).abs()
        dR, dt = d[:, :9].max(1).values, d[:, 9:12].max(1).values
        nR = (na[k][0] - nb[k][0]).abs().reshape(b, -1).max(1).values
        nt = (na[k][1] - nb[k][1]).abs().max(1).values
        w = int(dt.argmax())
        print(f"{name:26s} batch {k}: records max dR {float(dR.max()):.2e} max dt {float(dt.max()):.2e} m (ROI {w}; its network dt {float(nt[w]):.2e}, network dR {float(nR[w]):.2e}); "
              f"network outputs before the refine: max dR {float(nR.max()):.2e} max dt {float(nt.max()):.2e}; ROIs with record dt > 1e-4: {int((dt > 1e-4).sum())}")
