"""Which single kernel beside the 2x2-block upsample makes it go wrong: the conv3x3 three-product GEMM with the GroupNorm-statistics
epilogue, or the GroupNorm apply pass."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from gdrnpp_bop2022_amd import hip_lib
from gdrnpp_bop2022_amd.gdrn_modeling import heads, hip_layers

args = B.parse(["--steps", "4", "--no-cpu-baseline", "--no-pmc"] + sys.argv[1:])
dev = torch.device("cuda", 0)
state = B.build_state(args, ["ycbv_convnext_a6"], True, "refine", args.batch or 128, 0, dev, 0)
m = [o for o in gc.get_objects() if isinstance(o, dict) and "model" in o and "batches" in o and "post" in o][0]
model, batches = m["model"], m["batches"]
layers = list(model.geo_head_net.features)
cm = layers[3]
with torch.no_grad():
    feats = [model.backbone(batches[k]["roi_img"])[0].clone() for k in range(2)]
    x3 = [heads.run_features(layers[:3], f).clone() for f in feats]
    x5 = [heads.run_features(layers[3:5], x).clone() for x in x3]
    yref = hip_lib.upsample_bilinear2x(x5[0]).clone()
torch.cuda.synchronize()
lib = hip_lib.load()
x = x3[1]
n, cin, h, w = x.shape
conv, gn = cm.conv, cm.gn
cache = conv.__dict__.setdefault("_gdrnpp_cache", {})
w_pk, slot = hip_layers.x3_for(cache, "conv", conv.weight, hip_lib.pack_conv_weight_f16x2, n * h * w, conv.out_channels)
cout, groups = conv.out_channels, gn.num_groups
P = lib.gdrnpp_conv3x3_gnstats_partials(h, w)
y = torch.empty((n, cout, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
part = torch.empty((n, P, groups, 2), dtype=torch.float64, device=dev)
out = torch.empty_like(y)
def gemm():
    hip_lib._check(lib.gdrnpp_conv3x3_f32_split2(x.data_ptr(), w_pk.data_ptr(), None, y.data_ptr(), part.data_ptr(), n, h, w, cin, cout, groups, 0,
                                                 hip_lib._x3_flag_ptr(slot), hip_lib._stream()), "gemm")
def apply():
    hip_lib._check(lib.gdrnpp_groupnorm_apply_nhwc(y.data_ptr(), part.data_ptr(), P, gn.weight.data_ptr(), gn.bias.data_ptr(), out.data_ptr(), n, h * w, cout,
                                                   groups, float(gn.eps), 1, hip_lib._stream()), "apply")
gemm(); apply(); torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for s in streams: s.wait_stream(torch.cuda.current_stream())
def beside(name, comp, reps=8, n_up=6):
    worst, bad = 0.0, 0
    for rep in range(reps):
        with torch.cuda.stream(streams[1]):
            comp()
        with torch.cuda.stream(streams[0]):
            ys = [hip_lib.upsample_bilinear2x(x5[0]) for _ in range(n_up)]
        torch.cuda.synchronize()
        for yy in ys:
            d = (yy - yref).abs()
            worst = max(worst, float(d.max())); bad = max(bad, int((d > 0).sum()))
    print(f"upsample beside {name:40s} max diff {worst:.3e}  elements differing {bad}")
with torch.no_grad():
    beside("conv3x3 GEMM + GN statistics x4", lambda: [gemm() for _ in range(4)])
    beside("GroupNorm apply x12", lambda: [apply() for _ in range(12)])
    beside("conv2d GEMM without statistics x4", lambda: [hip_layers.conv2d(conv, x) for _ in range(4)])
    beside("torch GroupNorm x6", lambda: [torch.nn.functional.group_norm(y, groups) for _ in range(6)])
    beside("hip groupnorm_act (stats + apply) x6", lambda: [hip_layers.groupnorm_act(gn, None, y) for _ in range(6)])
