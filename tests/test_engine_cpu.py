"""Host logic of the engine without a GPU: ROI sharding rule, the record all-gather over gloo with
world_size 2, the class-sliced output layer against the reference-order graph, config surface."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gdrnpp_bop2022_amd.gdrn_modeling import engine
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg


def test_shard_range_matches_inference_sampler_rule():
    # core/utils/my_distributed_sampler.py:191-194: shard = (n-1)//world+1; [shard*rank, min(shard*(rank+1), n))
    for n in (0, 1, 7, 128, 1024, 1025):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, e = engine.shard_range(n, r, world)
                assert 0 <= b <= e <= n
                seen += list(range(b, e))
            assert seen == list(range(n))
    assert engine.shard_range(1024, 3, 8) == (384, 512)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, n_total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = engine.shard_range(n_total, rank, world)
    n_max = engine.shard_range(n_total, 0, world)[1]
    rec = torch.zeros((e - b, 16))
    rec[:, 14] = torch.arange(b, e)  # roi id
    rec[:, 15] = 1.0                 # valid
    rec[:, 9] = rank
    out = engine.gather_records(rec, n_max)
    ret[rank] = out.numpy()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 7])
def test_gather_records_gloo_world2(n_total):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_worker, args=(world, port, n_total, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert np.array_equal(a, b)                 # every rank holds the same gathered block
    valid = a[a[:, 15] > 0.5]
    assert sorted(valid[:, 14].astype(int).tolist()) == list(range(n_total))   # every ROI exactly once
    assert a.shape[0] == world * engine.shard_range(n_total, 0, world)[1]      # fixed shape incl. padding rows


def test_config_surface_and_opts():
    cfg = get_cfg("ycbv_convnext_a6", ["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True"])
    net = cfg.MODEL.POSE_NET
    assert net.NAME == "GDRN_double_mask" and net.NUM_CLASSES == 21 and net.OUTPUT_RES == 64
    assert net.GEO_HEAD.INIT_CFG.type == "TopDownDoubleMaskXyzRegionHead" and net.GEO_HEAD.INIT_CFG.in_dim == 1024
    assert net.GEO_HEAD.INIT_CFG.feat_dim == 256          # inherited from configs/_base_/gdrn_base.py
    assert net.PNP_NET.INIT_CFG.act == "gelu" and net.PNP_NET.INIT_CFG.type == "ConvPnPNet"
    assert cfg.TEST.USE_DEPTH_REFINE is True and cfg.TEST.DEPTH_REFINE_ITER == 2
    assert get_cfg("tless_convnext_a6").MODEL.POSE_NET.NUM_CLASSES == 30


def test_class_sliced_out_layer_equals_reference_graph():
    """The [B,70,256]x[B,256,4096] batched GEMM must reproduce conv1x1(1470 ch) + class gather."""
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

    cfg = get_cfg("tless_convnext_a6", ["MODEL.DEVICE=cpu", "TEST.USE_DEPTH_REFINE=True",
                                        "MODEL.POSE_NET.BACKBONE.INIT_CFG.type=timm/convnext_tiny",
                                        "MODEL.POSE_NET.GEO_HEAD.INIT_CFG.in_dim=768"])
    torch.manual_seed(0)
    model, _ = build_model_optimizer(cfg)
    torch.nn.init.normal_(model.geo_head_net.out_layer.weight, 0, 0.05)
    torch.nn.init.normal_(model.geo_head_net.out_layer.bias, 0, 0.5)
    b = 3
    x = torch.rand(b, 3, 256, 256)
    cls = torch.tensor([0, 29, 13])
    c2 = torch.rand(b, 2, 64, 64)
    ext = torch.rand(b, 3) * 0.2 + 0.05
    with torch.no_grad():
        r1, t1, m1 = model.forward_maps(x, cls, c2, None, ext)
        model.exact_reference_order = True
        r2, t2, m2 = model.forward_maps(x, cls, c2, None, ext)
    assert set(m1) == set(m2) == {"mask", "full_mask", "coor_x", "coor_y", "coor_z", "region"}
    for k in m1:
        assert m1[k].shape == m2[k].shape
        torch.testing.assert_close(m1[k], m2[k], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(r1, r2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(t1, t2, rtol=1e-5, atol=1e-6)
    # checkpoint key names the reference fixes (GDRN_double_mask.py:39-43, conv_module.py:175-182)
    keys = set(model.state_dict())
    for k in ["backbone.stem_0.weight", "backbone.stages_3.blocks.0.conv_dw.weight",
              "backbone.stages_1.downsample.1.weight", "geo_head_net.features.0.weight",
              "geo_head_net.features.3.conv.weight", "geo_head_net.features.3.gn.weight",
              "geo_head_net.out_layer.bias", "pnp_net.features.0.weight", "pnp_net.fc1.weight", "pnp_net.fc_r.bias"]:
        assert k in keys, k
    # duplicate `norm.*` keys of reference checkpoints are tolerated
    sd = model.state_dict()
    sd["geo_head_net.features.3.norm.weight"] = sd["geo_head_net.features.3.gn.weight"].clone()
    sd["geo_head_net.features.3.norm.bias"] = sd["geo_head_net.features.3.gn.bias"].clone()
    model.load_state_dict(sd, strict=True)


def test_rois_from_detections_rules():
    """data_loader.py:754-769: centre, clamped box size, padded + clipped scale, resize ratio."""
    r = engine.rois_from_detections([[100, 50, 180, 250], [0, 0, 639.5, 479], [10, 10, 10.2, 10.4]], 480, 640)
    np.testing.assert_allclose(r["bbox_center"], [[140, 150], [319.75, 239.5], [10.1, 10.2]])
    np.testing.assert_allclose(r["scale"], [300.0, 640.0, 1.5])          # 200*1.5, clipped to max(H,W), bw/bh >= 1
    np.testing.assert_allclose(r["roi_wh"], [[80, 200], [639.5, 479], [1, 1]])
    np.testing.assert_allclose(r["resize_ratio"], 64 / r["scale"])


def test_records_to_bop_and_csv_format(tmp_path):
    """Pose records -> BOP dicts (R flattened row-major, t in mm, invalid rows dropped) -> the CSV layout of
    save_and_eval_results (test_utils.py:33-52)."""
    import numpy as np
    import torch
    from gdrnpp_bop2022_amd.gdrn_modeling import engine

    rec = torch.zeros(3, 16)
    rec[0, :9] = torch.arange(9, dtype=torch.float32) * 0.125
    rec[0, 9:12] = torch.tensor([0.01, -0.02, 0.75]); rec[0, 12] = 0.5; rec[0, 13] = 2; rec[0, 14] = 1; rec[0, 15] = 1
    rec[1, 15] = 0                                     # padding row
    rec[2, :9] = torch.eye(3).reshape(9); rec[2, 9:12] = torch.tensor([0.0, 0.0, 1.0]); rec[2, 12] = 1.0; rec[2, 13] = 0
    rec[2, 14] = 0; rec[2, 15] = 1
    res = engine.records_to_bop(rec, ["48/1", "50/17"], obj_ids=[1, 5, 9], times=[0.25, 0.5])
    assert len(res) == 2 and res[0]["scene_id"] == "50" and res[0]["im_id"] == 17 and res[0]["obj_id"] == 9
    assert res[0]["R"] == [i * 0.125 for i in range(9)] and res[0]["time"] == 0.5
    assert np.allclose(res[0]["t"], [10.0, -20.0, 750.0]) and res[1]["t"] == [0.0, 0.0, 1000.0]
    path = tmp_path / "out.csv"
    engine.save_bop_csv(res, str(path))
    lines = path.read_text().splitlines()
    assert lines[0] == "scene_id,im_id,obj_id,score,R,t,time" and len(lines) == 3
    f = lines[2].split(",")
    assert f[:4] == ["48", "1", "1", "1.0"] and f[4] == "1.0 0.0 0.0 0.0 1.0 0.0 0.0 0.0 1.0" and f[5] == "0.0 0.0 1000.0" and f[6] == "0.25"


def test_detections_from_yolox_handoff():
    """Detector output (dets [B,max_det,7], count [B]) -> the detections dict of batch_data_test_gpu: per-image order kept,
    boxes divided by the test-time resize ratio, score = obj * class_conf, empty images skipped, per-image cap."""
    import numpy as np
    import torch
    from gdrnpp_bop2022_amd.gdrn_modeling import engine

    dets = torch.zeros(3, 4, 7)
    dets[0, 0] = torch.tensor([10.0, 20.0, 110.0, 220.0, 0.9, 0.8, 3.0])
    dets[0, 1] = torch.tensor([50.0, 60.0, 70.0, 90.0, 0.5, 0.5, 1.0])
    dets[2, 0] = torch.tensor([5.0, 6.0, 7.0, 8.0, 1.0, 0.25, 20.0])
    count = torch.tensor([2, 0, 1], dtype=torch.int32)
    K = np.eye(3, dtype=np.float32)
    ext = np.full((21, 3), 0.1, np.float32)
    d = engine.detections_from_yolox(dets, count, K, ext, ratio=2.0)
    assert d["im_idx"].tolist() == [0, 0, 2] and d["roi_cls"].tolist() == [3, 1, 20]
    assert np.allclose(d["bbox"], [[5, 10, 55, 110], [25, 30, 35, 45], [2.5, 3, 3.5, 4]])
    assert np.allclose(d["score"], [0.72, 0.25, 0.25]) and d["cam"] is K and d["extents"] is ext
    d1 = engine.detections_from_yolox(dets, count, K, ext, max_per_image=1)
    assert d1["im_idx"].tolist() == [0, 2] and d1["roi_cls"].tolist() == [3, 20]
    d0 = engine.detections_from_yolox(dets, torch.zeros(3, dtype=torch.int32), K, ext)
    assert d0["bbox"].shape == (0, 4) and len(d0["roi_cls"]) == 0
    r = engine.rois_from_detections(d["bbox"], 480, 640)
    assert np.allclose(r["bbox_center"][0], [30, 60]) and np.allclose(r["scale"][0], 150.0) and np.allclose(r["resize_ratio"][0], 64 / 150.0)


def test_upnp_weights_match_scipy_sqrtm():
    """pose_from_upnp's weights inv(sqrtm(C)) (gdrn_evaluator.py:612-626): closed form vs the reference's own dependency,
    scipy.linalg.sqrtm, incl. the degenerate-covariance rule."""
    import numpy as np
    import scipy.linalg
    from gdrnpp_bop2022_amd.gdrn_modeling import engine

    rng = np.random.default_rng(0)
    a = rng.normal(size=(40, 2, 2))
    cov = a @ a.transpose(0, 2, 1) + 0.05 * np.eye(2)
    cov[3] = 0.0                       # C[0,0] < 1e-6 -> zero weight
    cov[7, 0, 1] = np.nan              # NaN -> zero weight
    w = engine.upnp_weights_from_cov(cov)
    assert w.shape == (40, 3) and not w[3].any() and not w[7].any()
    for i in range(40):
        if i in (3, 7):
            continue
        ref = np.linalg.inv(scipy.linalg.sqrtm(cov[i])).reshape(4)[[0, 1, 3]]
        np.testing.assert_allclose(w[i], ref.real, rtol=1e-10, atol=1e-12)


def test_coor_planes_regression_and_classification():
    """get_out_coor (engine_utils.py:295-312): regression maps pass through; classification logits -> argmax bin / (BIN-1),
    background bin -> 0."""
    import torch
    from gdrnpp_bop2022_amd.gdrn_modeling import engine
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg

    cfg = get_cfg("ycbv_convnext_a6", [])
    nbin = cfg.MODEL.POSE_NET.GEO_HEAD.XYZ_BIN
    reg = {k: torch.rand(2, 1, 4, 4) for k in ("coor_x", "coor_y", "coor_z")}
    out = engine.coor_planes(cfg, reg)
    assert all(torch.equal(o, reg[k]) for o, k in zip(out, ("coor_x", "coor_y", "coor_z")))
    logits = torch.full((1, nbin + 1, 2, 2), -5.0)
    logits[0, 0, 0, 0] = 9.0            # bin 0
    logits[0, nbin - 1, 0, 1] = 9.0     # last foreground bin -> 1.0
    logits[0, nbin, 1, 0] = 9.0         # background bin -> 0
    logits[0, 21, 1, 1] = 9.0
    cls = {k: logits.clone() for k in ("coor_x", "coor_y", "coor_z")}
    px, py, pz = engine.coor_planes(cfg, cls)
    want = torch.tensor([[0.0, 1.0], [0.0, 21.0 / (nbin - 1)]])
    assert px.shape == (1, 1, 2, 2) and torch.allclose(px[0, 0], want) and torch.equal(px, py) and torch.equal(py, pz)


def test_detections_from_bop_json_selection_rules():
    """load_detections_into_dataset (dataset_utils.py:146-227): score threshold, unknown / untrained objects dropped,
    top-k per object by score (stable for ties), dataset class order, xywh -> xyxy, images without detections skipped."""
    import numpy as np
    from gdrnpp_bop2022_amd.gdrn_modeling import engine

    dets = {
        "48/1": [
            {"obj_id": 5, "bbox_est": [10, 20, 30, 40], "score": 0.6, "time": 0.1},
            {"obj_id": 5, "bbox_est": [11, 21, 31, 41], "score": 0.9, "time": 0.1},
            {"obj_id": 1, "bbox_est": [1, 2, 3, 4], "score": 0.5},
            {"obj_id": 1, "bbox_est": [5, 6, 7, 8], "score": 0.5},        # tie: first one stays first
            {"obj_id": 99, "bbox_est": [0, 0, 1, 1], "score": 1.0},       # not an object of the dataset
            {"obj_id": 9, "bbox_est": [0, 0, 9, 9], "score": 0.05},       # below the threshold
        ],
        "50/7": [{"obj_id": 9, "bbox_est": [100, 100, 50, 60], "score": 0.7, "time": 0.3}],
    }
    d = engine.detections_from_bop_json(dets, ["48/1", "49/3", "50/7"], obj_ids=[1, 5, 9], cam=None, extents=None,
                                        top_k_per_obj=1, score_thr=0.1)
    assert d["im_idx"].tolist() == [0, 0, 2] and d["roi_cls"].tolist() == [0, 1, 2]
    assert np.allclose(d["bbox"], [[1, 2, 4, 6], [11, 21, 42, 62], [100, 100, 150, 160]])
    assert np.allclose(d["score"], [0.5, 0.9, 0.7]) and np.allclose(d["time"], [0.0, 0.1, 0.3])
    d2 = engine.detections_from_bop_json(dets, ["48/1"], obj_ids=[1, 5, 9], cam=None, extents=None, top_k_per_obj=2,
                                         train_obj_ids=[5])
    assert d2["roi_cls"].tolist() == [1, 1] and np.allclose(d2["score"], [0.9, 0.6])


def test_detections_from_bop_json_equals_the_reference_loader():
    """engine.detections_from_bop_json against ``load_detections_into_dataset`` of the reference executed from its source text
    (tests/golden/make_golden_dets.py -> dets_golden.npz) on a seeded detection file with equal scores, foreign objects, scores
    under the threshold, an image without an entry and one whose entries are all filtered: the same ROIs in the same order with
    the same class index, box (xywh -> xyxy), score and time, for four (top_k, score_thr, train_objs) settings."""
    import json

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dets_golden.npz"))
    keys, dets, names = json.loads(str(z["keys"])), json.loads(str(z["detections"])), json.loads(str(z["names"]))
    obj_ids = z["obj_ids"].tolist()
    total = 0
    for case in json.loads(str(z["cases"])):
        a = case["args"]
        train = None if a["train_objs"] is None else [obj_ids[names.index(n)] for n in a["train_objs"]]
        d = engine.detections_from_bop_json(dets, keys, obj_ids, cam=None, extents=None, top_k_per_obj=a["top_k_per_obj"],
                                            score_thr=a["score_thr"], train_obj_ids=train)
        want = [(keys.index(im["scene_im_id"]), ann) for im in case["images"] for ann in im["annotations"]]
        assert d["im_idx"].tolist() == [w[0] for w in want] and d["roi_cls"].tolist() == [w[1][0] for w in want], a
        assert np.allclose(d["score"], [w[1][2] for w in want]) and np.allclose(d["time"], [w[1][3] for w in want]), a
        boxes = np.array([[w[1][1][0], w[1][1][1], w[1][1][0] + w[1][1][2], w[1][1][1] + w[1][1][3]] for w in want], np.float32).reshape(-1, 4)
        assert np.allclose(d["bbox"], boxes), a
        total += len(want)
    assert total == 57


def test_folded_conv_bn_equals_batchnorm_of_conv():
    """hip_layers.folded_conv_bn (inference BatchNorm folded into the convolution in front of it, ResNet path of config 1):
    conv(x, w', b') == bn(conv(x)) in eval mode, for a biased and an unbiased convolution, and the cache follows in-place
    changes of the running statistics."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers

    torch.manual_seed(0)
    for bias in (False, True):
        conv = nn.Conv2d(8, 12, 3, 2, 1, bias=bias).double()
        bn = nn.BatchNorm2d(12).double().eval()
        with torch.no_grad():
            bn.running_mean.normal_(0.0, 0.3)
            bn.running_var.uniform_(0.4, 1.6)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0.0, 0.3)
        x = torch.randn(2, 8, 9, 11, dtype=torch.float64)
        with torch.no_grad():
            w, b = hip_layers.folded_conv_bn(conv, bn)
            assert w.dtype == torch.float32 and b.dtype == torch.float32
            got = F.conv2d(x, w.double(), b.double(), conv.stride, conv.padding)
            want = bn(conv(x))
            assert (got - want).abs().max().item() < 5e-6
            bn.running_mean.add_(0.5)                      # in-place update bumps the version counter: the fold is rebuilt
            w2, b2 = hip_layers.folded_conv_bn(conv, bn)
            got2 = F.conv2d(x, w2.double(), b2.double(), conv.stride, conv.padding)
            assert (got2 - bn(conv(x))).abs().max().item() < 5e-6
            assert (b2 - b).abs().max().item() > 0.1


# ---- ROIs sorted by class within a rank, the original index carried in the record -------------------------
def test_sort_detections_by_class_is_stable_and_carries_the_original_index():
    rng = np.random.default_rng(5)
    n = 37
    det = dict(bbox=rng.uniform(0, 400, (n, 4)).astype(np.float32), im_idx=rng.integers(0, 4, n), roi_cls=rng.integers(0, 6, n),
               score=rng.uniform(0, 1, n).astype(np.float32), cam=np.eye(3, dtype=np.float32), extents=np.ones((6, 3), np.float32),
               time=rng.uniform(0, 1, n).astype(np.float32))
    out, roi_id = engine.sort_detections_by_class(det, roi_id_base=100)
    assert (np.diff(out["roi_cls"]) >= 0).all() and out["cam"].shape == (3, 3)
    orig = roi_id - 100
    assert sorted(orig.tolist()) == list(range(n))
    for k in ("bbox", "im_idx", "roi_cls", "score", "time"):
        assert np.array_equal(out[k], det[k][orig])
    for c in range(6):                                    # stable: detection order kept inside a class
        assert (np.diff(orig[out["roi_cls"] == c]) > 0).all()
    cams = np.repeat(np.eye(3, dtype=np.float32)[None], n, 0) * np.arange(1, n + 1, dtype=np.float32)[:, None, None]
    out2, _ = engine.sort_detections_by_class(dict(det, cam=cams))
    assert np.array_equal(out2["cam"], cams[orig])
    assert torch.equal(engine.class_sorted_order(torch.from_numpy(det["roi_cls"])), torch.from_numpy(orig.astype(np.int64)))


def test_sort_detections_by_class_reorders_nothing_on_a_guess():
    """The per-ROI / global split is a contract (engine.PER_ROI_DETECTION_KEYS / GLOBAL_DETECTION_KEYS): a per-class table whose
    length happens to equal the number of ROIs passes through when it is a known global key, an unknown key of that length
    raises until the caller says what it is, a per-ROI key of the wrong length raises."""
    n = 6
    cls = np.array([3, 1, 5, 0, 1, 2])
    table = np.arange(n * 3, dtype=np.float32).reshape(n, 3)                 # n == number of classes
    det = dict(roi_cls=cls, bbox=np.zeros((n, 4), np.float32), extents=table, obj_ids=list(range(1, n + 1)), cam=np.eye(3))
    out, rid = engine.sort_detections_by_class(det)
    assert np.array_equal(out["extents"], table) and out["obj_ids"] == list(range(1, n + 1))
    with pytest.raises(KeyError, match="extra_per_roi_keys"):
        engine.sort_detections_by_class(dict(det, keypoints=table))
    out, rid = engine.sort_detections_by_class(dict(det, keypoints=table), extra_global_keys=("keypoints",))
    assert np.array_equal(out["keypoints"], table)
    out, rid = engine.sort_detections_by_class(dict(det, keypoints=table, names=list("abcdef")), extra_per_roi_keys=("keypoints", "names"))
    assert np.array_equal(out["keypoints"], table[rid]) and out["names"] == [list("abcdef")[i] for i in rid]
    with pytest.raises(ValueError, match="per-ROI"):
        engine.sort_detections_by_class(dict(det, score=np.ones(n - 1)))
    assert engine.sort_detections_by_class(dict(det, note="x", thr=0.5, mask=None))[0]["note"] == "x"   # scalars / strings / None pass


def _sorted_shard_worker(rank, world, port, n_total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cls_all = np.random.default_rng(11).integers(0, 5, n_total)          # the iteration's detections, same on every rank
    b, e = engine.shard_range(n_total, rank, world)
    n_max = engine.shard_range(n_total, 0, world)[1]
    det = dict(roi_cls=cls_all[b:e], score=np.arange(b, e, dtype=np.float32), cam=np.eye(3, dtype=np.float32))
    det, roi_id = engine.sort_detections_by_class(det, roi_id_base=b)   # this rank's shard in class order
    rec = torch.zeros((e - b, 16))
    rec[:, 12] = torch.from_numpy(det["score"])                           # stands for the pose of that ROI
    rec[:, 13] = torch.from_numpy(det["roi_cls"]).float()
    rec[:, 14] = torch.from_numpy(roi_id).float()
    rec[:, 15] = 1.0
    ret[rank] = (engine.gather_records(rec, n_max).numpy(), det["roi_cls"])
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [12, 9])
def test_class_sorted_shards_gather_and_restore_roi_order_gloo_world2(n_total):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sorted_shard_worker, args=(world, _free_port(), n_total, ret), nprocs=world, join=True)
    (a, cls0), (b, cls1) = ret[0], ret[1]
    assert np.array_equal(a, b)
    assert (np.diff(cls0) >= 0).all() and (np.diff(cls1) >= 0).all()       # each rank ran its ROIs in class order
    rec = engine.records_in_roi_order(torch.from_numpy(a)).numpy()
    assert rec.shape[0] == n_total                                           # padding rows dropped
    assert np.array_equal(rec[:, 14], np.arange(n_total))                   # original order restored
    assert np.array_equal(rec[:, 12], np.arange(n_total))                   # ... and every record is its own ROI's
    assert np.array_equal(rec[:, 13], np.random.default_rng(11).integers(0, 5, n_total))


def test_gemm_product_switch_and_range_word_policy(monkeypatch):
    """Host logic of the three-product mode (no GPU): the switch validates its argument, launches are eligible from 256 tiles of
    256 x 128 on; a step's range words demote the layers with rows below the range and the FIRST overflowing layer, the repeat
    runs with six products for the calling thread only, and a process whose steps keep overflowing settles on six products."""
    import threading
    import warnings

    import pytest

    from gdrnpp_bop2022_amd import hip_lib
    from gdrnpp_bop2022_amd.gdrn_modeling import engine, hip_layers

    assert hip_layers.gemm_products() == 3
    with pytest.raises(ValueError):
        hip_layers.set_gemm_products(4)
    assert hip_lib.split2_tiles_ok(128 * 256, 256) and not hip_lib.split2_tiles_ok(127 * 256, 256)
    assert hip_lib.split2_tiles_ok(64 * 256 + 1, 512) and not hip_lib.split2_tiles_ok(1 << 20, 192)
    # linear form: the A operand must stay below 4 GiB (32-bit lane offsets) — 128 ROIs of stage-0 fc2 do, 600 do not
    assert hip_layers._use_x3(128 * 4096, 128, 512) and not hip_layers._use_x3(600 * 4096, 128, 512)
    assert hip_layers._use_x3(600 * 4096, 256)            # convolutions address per pixel: no such limit
    seen = []

    def run():
        seen.append(hip_layers.gemm_products())
        other = []
        t = threading.Thread(target=lambda: other.append(hip_layers.gemm_products()))
        t.start()
        t.join()
        seen.append(other[0])
        return "rec"

    hip_layers.reset_x3_demotions()
    monkeypatch.setattr(engine, "_X3_OVERFLOW_STEPS", 0)
    S, NF = hip_lib.X3_SMALL_ROWS, hip_lib.X3_NONFINITE
    try:
        # rows below the range in layers 5 and 9; non-finite values from layer 7 on (8 and 9 saw them pass through)
        assert engine._six_product_rerun(run, {5: S, 7: NF, 8: NF, 9: NF | S}) == "rec"
        assert seen == [6, 3]                              # six products for this thread, the process setting for the others
        assert hip_layers.gemm_products() == 3 and hip_layers.x3_demoted() == {5: S, 7: NF, 9: NF | S}
        assert engine._X3_OVERFLOW_STEPS == 1
        engine._six_product_rerun(run, {0: S})             # slot 0 = launches that named no layer: repeated, nothing to demote
        assert hip_layers.x3_demoted() == {5: S, 7: NF, 9: NF | S} and engine._X3_OVERFLOW_STEPS == 1
        engine._six_product_rerun(run, {8: NF})
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            engine._six_product_rerun(run, {11: NF})
        assert hip_layers.gemm_products() == 6 and any("six-product" in str(x.message) for x in w)
        assert set(hip_layers.x3_demoted()) == {5, 7, 8, 9, 11}
    finally:
        hip_layers.set_gemm_products(3)
        hip_layers.reset_x3_demotions()


def test_x3_layers_get_slots_and_lose_the_form_when_demoted(monkeypatch):
    """hip_layers.x3_for: a layer (module cache, key) gets its range-word slot at the first eligible launch (launch order), runs
    the three-product kernel until it is demoted or its packed weight reports rows below the range, and reset_x3_demotions()
    forgets both (new weights).  No device: the pack and its verdict are stubbed."""
    import torch

    from gdrnpp_bop2022_amd import hip_lib
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers as hl

    hl.reset_x3_demotions()
    verdict = {"ok": True}
    monkeypatch.setattr(hip_lib, "packed_rows_in_range", lambda packed: verdict["ok"])
    packs = []

    def pack3(w):
        packs.append(1)
        return ("packed", len(packs))

    w = torch.zeros(512, 128)
    c1, c2 = {}, {}
    big = (128 * 4096, 512, 128)
    try:
        p, s1 = hl.x3_for(c1, "fc1", w, pack3, *big)
        assert p == ("packed", 1) and s1 == 1
        assert hl.x3_for(c1, "fc1", w, pack3, *big) == (("packed", 1), 1) and len(packs) == 1     # cached
        assert hl.x3_for(c2, "fc1", w, pack3, *big)[1] == 2 and hl.x3_for(c1, "fc2", w, pack3, *big)[1] == 3
        assert hl.x3_for({}, "fc2", w, pack3, 4 * 4096, 128, 512) == (None, 0)                   # too few tiles: no slot spent
        hl.demote_x3({2: hip_lib.X3_SMALL_ROWS})
        assert hl.x3_for(c2, "fc1", w, pack3, *big) == (None, 2) and hl.x3_for(c1, "fc1", w, pack3, *big)[0] is not None
        with hl.forced_gemm_products(6):
            assert hl.x3_for(c1, "fc1", w, pack3, *big) == (None, 0)
        w.add_(1.0)                                         # in-place weight update: packed again, verdict read again
        verdict["ok"] = False
        assert hl.x3_for(c1, "fc1", w, pack3, *big) == (None, 1) and len(packs) == 4
        hl.reset_x3_demotions()
        verdict["ok"] = True
        w.add_(1.0)
        assert hl.x3_for(c2, "fc1", w, pack3, *big)[1] == 1 and hl.x3_demoted() == {}
    finally:
        hl.reset_x3_demotions()


# ---- ROI packing: the reference's one-image-per-forward stream dealt into steps of exactly P ROIs ------------------------------
def test_roi_packer_deals_exact_steps_and_returns_records_to_their_images():
    rng = np.random.default_rng(1)
    P = 16
    pk = engine.RoiPacker(P, roi_id_base=engine.RoiPacker.ID_WRAP - 40)       # the ids wrap (float32-exact range) mid-stream
    counts = rng.integers(0, 12, 40).tolist()
    packs, order = [], []
    for key, n in enumerate(counts):
        pk.add_image(key, n)
        if n:
            with pytest.raises(KeyError):
                pk.add_image(key, n)                      # a key cannot be in flight twice
        while pk.ready():
            packs.append(pk.next_pack())
    assert pk.next_pack() is None and 0 <= pk.pending < P
    tail = pk.next_pack(flush=True)
    if tail:
        packs.append(tail)
    assert pk.pending == 0 and pk.next_pack(flush=True) is None
    sizes = [sum(len(loc) for _, loc, _ in p) for p in packs]
    assert all(s == P for s in sizes[:-1]) and sum(sizes) == sum(counts)
    # arrival order is kept: flattening the packs enumerates (image, local index) in stream order
    flat = [(k, j) for p in packs for k, loc, _ in p for j in loc.tolist()]
    assert flat == [(k, j) for k, n in enumerate(counts) for j in range(n)]
    ids = np.concatenate([i for p in packs for _, _, i in p])
    assert len(set(ids.tolist())) == len(ids) and ids.max() < engine.RoiPacker.ID_WRAP and ids.min() == 0
    # steps come back in any order, rows in any order (class-sorted in the step), padding rows are ignored
    done = dict(pk.pop_completed())                                          # the images without detections
    assert sorted(done) == [k for k, n in enumerate(counts) if n == 0] and all(r.shape == (0, 16) for r in done.values())
    for p in packs[::-1]:
        ids_p = np.concatenate([i for _, _, i in p])
        rec = np.zeros((len(ids_p) + 3, 16), np.float32)
        rec[len(ids_p):, 14] = engine.PAD_ROI_ID                             # gather_records marks its padding rows
        rec[:len(ids_p), 14], rec[:len(ids_p), 15] = ids_p, 1.0
        rec[:len(ids_p), 0] = [1000 * k + j for k, loc, _ in p for j in loc.tolist()]
        pk.deliver(rec[rng.permutation(len(rec))])
        done.update(pk.pop_completed())
    assert sorted(done) == list(range(len(counts)))
    for k, n in enumerate(counts):
        assert done[k].shape == (n, 16) and done[k][:, 0].tolist() == [1000 * k + j for j in range(n)]


def test_upload_packed_round_trips_every_dtype_and_shape():
    """engine.upload_packed: the per-ROI host arrays of a step through one staging buffer (pinned + one asynchronous copy on a
    device) come back as tensors of the same dtype / shape / values, 16-byte aligned, empty arrays included."""
    rng = np.random.default_rng(3)
    arrays = dict(center64=rng.uniform(0, 640, (7, 2)), scale64=rng.uniform(10, 300, 7), im_idx=rng.integers(0, 4, 7).astype(np.int32),
                  roi_cls=rng.integers(0, 21, 7), cam=np.tile(np.eye(3, dtype=np.float32), (7, 1, 1)), score=rng.uniform(0, 1, 7).astype(np.float32),
                  empty=np.zeros((0, 3), np.float32), ids=np.arange(7, dtype=np.int32)[::-1])
    out = engine.upload_packed(arrays, "cpu")
    assert list(out) == list(arrays)
    for k, a in arrays.items():
        assert out[k].shape == a.shape and np.array_equal(out[k].numpy(), a) and str(out[k].dtype).split(".")[1] == str(a.dtype), k
        assert out[k].data_ptr() % 16 == 0 or a.size == 0


def test_roi_packer_delivers_invalid_records_and_scheduler_admission_is_atomic():
    """(round-4 advice) A real ROI may come back with valid = 0 (depth refine: object id outside the mesh set): its image must
    still complete and keep the row's valid bit — even when the whole record is zero (round-5 advice: ROI id 0 with zero R, t,
    score, obj is a record, not padding); only MARKED padding rows (roi_id = PAD_ROI_ID) and foreign ids are skipped.  A key pushed while it
    is still in flight raises BEFORE the scheduler's state for that key is touched; a stream cannot mix depth / no depth."""
    pk = engine.RoiPacker(4, roi_id_base=0)
    pk.add_image("a", 3)
    pk.add_image("b", 1)
    (ka, la, ia), (kb, lb, ib) = pk.next_pack()
    rec = np.zeros((6, 16), np.float32)                       # 4 records + 2 padding rows
    rec[4:, 14] = engine.PAD_ROI_ID
    rec[:3, 14], rec[3, 14] = ia, ib[0]
    rec[:4, 15] = [0.0, 1.0, 1.0, 1.0]                        # a's first ROI (id 0) is INVALID but real ...
    rec[1:4, 0] = 1.0                                         # ... and its record is zero in EVERY column
    assert ia[0] == 0 and not rec[0].any()
    pk.deliver(rec[[4, 0, 5, 3, 2, 1]])
    done = dict(pk.pop_completed())
    assert sorted(done) == ["a", "b"] and done["a"][:, 15].tolist() == [0.0, 1.0, 1.0] and not pk._where and not pk._open
    pk.deliver(rec)                                           # a late duplicate of the same step: nothing in flight, nothing happens
    assert pk.pop_completed() == []

    sch = engine.RoiStreamScheduler(None, None, None, rois_per_step=64)
    d1 = dict(roi_cls=np.array([1, 2]), bbox=np.zeros((2, 4), np.float32))
    sch._admit("k", "image-1", "depth-1", d1)
    t_first = sch._arrival["k"]
    with pytest.raises(KeyError):
        sch._admit("k", "image-2", "depth-2", dict(roi_cls=np.array([3]), bbox=np.zeros((1, 4), np.float32)))
    assert sch._images["k"][0] == "image-1" and sch._arrival["k"] == t_first and sch.packer.pending == 2
    with pytest.raises(ValueError, match="mixed"):
        sch._admit("m", "image-3", None, d1)
    assert "m" not in sch._images and "m" not in sch._arrival and sch.packer.pending == 2
    sch._admit("e", "image-4", None, dict(roi_cls=np.zeros((0,), np.int64), bbox=np.zeros((0, 4), np.float32)))   # no ROIs: no depth needed
    assert [k for k, _ in sch.packer.pop_completed()] == ["e"]


def test_first_overflowing_layer_is_the_first_in_launch_order_not_the_smallest_slot(monkeypatch):
    """(round-4 advice) Slots are handed out at a layer's first ELIGIBLE launch — a later layer may hold a smaller slot (another
    model, another batch size).  Of the layers reporting non-finite values the one launched first in the step is demoted."""
    import torch

    from gdrnpp_bop2022_amd import hip_lib
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers as hl

    hl.reset_x3_demotions()
    monkeypatch.setattr(hip_lib, "packed_rows_in_range", lambda packed: True)
    monkeypatch.setattr(engine, "_X3_OVERFLOW_STEPS", 0)
    w = torch.zeros(512, 128)
    big, small = (128 * 4096, 512, 128), (4 * 4096, 128, 512)
    late, early = {}, {}
    try:
        # a small batch first: only the LATE layer is eligible (say, another model's) and takes slot 1
        assert hl.x3_for(early, "fc", w, lambda t: "p", *small) == (None, 0)
        assert hl.x3_for(late, "fc", w, lambda t: "p", *big)[1] == 1
        # the step under test launches early (slot 2) and then late (slot 1)
        assert hl.x3_for(early, "fc", w, lambda t: "p", *big)[1] == 2
        assert hl.x3_for(late, "fc", w, lambda t: "p", *big)[1] == 1
        assert hl.x3_launch_order([1, 2, 7]) == [2, 1, 7]
        engine._note_range_words({1: hip_lib.X3_NONFINITE, 2: hip_lib.X3_NONFINITE})
        assert hl.x3_demoted() == {2: hip_lib.X3_NONFINITE}
    finally:
        hl.reset_x3_demotions()


def test_packed_loader_pools_images_until_the_step_is_full():
    from gdrnpp_bop2022_amd.gdrn_modeling.gdrn_evaluator import packed_loader
    counts = [3, 30, 12, 0, 25, 29, 30, 7, 1]
    loader = [[dict(roi_cls=list(range(n)), tag=i)] for i, n in enumerate(counts)]
    packs = list(packed_loader(loader, 64))
    assert [[d["tag"] for d in p] for p in packs] == [[0, 1, 2, 3, 4], [5, 6, 7], [8]]
    assert [sum(len(d["roi_cls"]) for d in p) for p in packs] == [70, 66, 1]
    assert [[d["tag"] for d in p] for p in packed_loader(loader, 1)] == [[0], [1], [2], [3, 4], [5], [6], [7], [8]]


def _packed_stream_worker(rank, world, port, ret):
    """Each rank runs its own image stream (InferenceSampler shards IMAGES), packs it into steps of P ROIs, and every step ends
    in the path's one collective; the step itself is a host stub that emits the records of the ROIs it was given."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, n_steps = 8, 5
    rng = np.random.default_rng(100 + rank)
    pk = engine.RoiPacker(P, roi_id_base=rank * 100000)                      # disjoint id ranges per rank
    seen, gathered, key = {}, [], 0
    for _ in range(n_steps):
        while not pk.ready():
            n = int(rng.integers(0, 6))
            pk.add_image((rank, key), n)
            seen[(rank, key)] = n
            key += 1
        pack = pk.next_pack()
        cls = rng.integers(0, 4, P)
        ids = np.concatenate([i for _, _, i in pack])
        order = engine.class_sorted_order(cls)                               # the step runs its ROIs in class order
        rec = torch.zeros((P, 16))
        rec[:, 14] = torch.from_numpy(ids[order].astype(np.float32))
        rec[:, 13] = torch.from_numpy(cls[order].astype(np.float32))
        rec[:, 12] = rec[:, 14] * 0.5
        rec[:, 15] = 1.0
        allrec = engine.gather_records(rec, P)                               # [world * P, 16], same on every rank
        gathered.append(allrec.numpy())
        mine = allrec[(allrec[:, 14] >= rank * 100000) & (allrec[:, 14] < (rank + 1) * 100000)]
        pk.deliver(mine.numpy())
    done = {k: r for k, r in pk.pop_completed()}
    ret[rank] = (np.stack(gathered), {k: r[:, 12].tolist() for k, r in done.items()}, seen, pk.pending)
    dist.destroy_process_group()


def test_packed_streams_shard_gather_and_restore_gloo_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_packed_stream_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    (g0, done0, seen0, pend0), (g1, done1, seen1, pend1) = ret[0], ret[1]
    assert np.array_equal(g0, g1) and g0.shape == (5, 16, 16)                # every step gathered 2 x 8 records, identical everywhere
    for rank, (done, seen, pend) in enumerate(((done0, seen0, pend0), (done1, seen1, pend1))):
        n_done = sum(len(v) for v in done.values())
        assert n_done + pend <= 5 * 8 and n_done >= 5 * 8 - pend - 5         # all but the images straddling the last step
        for k, vals in done.items():
            assert k[0] == rank and len(vals) == seen[k]
        ids = sorted(v * 2 for vals in done.values() for v in vals)
        assert ids == sorted(set(ids)) and all(rank * 100000 <= i < (rank + 1) * 100000 for i in ids)
    rec = engine.records_in_roi_order(torch.from_numpy(g0.reshape(-1, 16))).numpy()
    assert (np.diff(rec[:, 14]) > 0).all() and len(rec) == 2 * 5 * 8


def test_xyz_back_projection_equals_the_reference_function():
    """engine.xyz_back_projection (the XYZ_BP branch of the training-side batch_data, engine_utils.py:131-150) against
    calc_xyz_bp_batch of the reference executed from its source (tests/golden/make_golden_xyz.py -> xyz_bp_golden.npz): same
    depth / R / t / K in, the same points out to fp32 rounding, the same object mask."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xyz_bp_golden.npz"))
    T = torch.from_numpy
    xyz = engine.xyz_back_projection(T(z["depth"]), T(z["R"]), T(z["t"]), T(z["K_crop"])).numpy()
    assert xyz.shape == z["xyz_bp"].shape == (6, 64, 64, 3)
    assert np.abs(xyz - z["xyz_bp"]).max() <= 2e-7 * np.abs(z["xyz_bp"]).max() + 1e-8
    mask = ((xyz[..., 0] != 0) & (xyz[..., 1] != 0) & (xyz[..., 2] != 0)).astype(np.float32)
    assert np.array_equal(mask, z["mask_obj"]) and 0.1 < mask.mean() < 0.5


def test_two_steps_in_flight_only_where_every_kernel_of_a_step_is_ours():
    """engine.default_compute_streams: 2 for a ConvNeXt configuration with the HIP network layers and split GEMMs on (a step then
    launches this library's kernels only — built and link-checked against the packed-fp32 / MFMA hazard of MI355X), 1 for the
    ResNet-34 configuration (MIOpen convolutions in the step) and whenever PyTorch's operators are switched in."""
    from gdrnpp_bop2022_amd.gdrn_modeling import engine, hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer
    from gdrnpp_bop2022_amd.gdrn_modeling import GDRN

    cx, _ = build_model_optimizer(get_cfg("ycbv_convnext_a6"))
    cfg_r = get_cfg("lmo_resnet34_ape")
    builder = GDRN.build_model_optimizer if cfg_r.MODEL.POSE_NET.NAME == "GDRN" else build_model_optimizer
    rn, _ = builder(cfg_r)
    assert hip_layers.is_enabled() and hip_layers.mlp_gemm() == "split"
    assert engine.default_compute_streams(cx) == 2 and engine.default_compute_streams(rn) == 1
    try:
        hip_layers.set_enabled(False)
        assert engine.default_compute_streams(cx) == 1
        hip_layers.set_enabled(True)
        hip_layers.set_mlp_gemm("torch")
        assert engine.default_compute_streams(cx) == 1
    finally:
        hip_layers.set_enabled(True)
        hip_layers.set_mlp_gemm("split")


def test_three_product_tile_rule_alone_and_on_a_shared_chip():
    """hip_lib.split2_tiles_ok: 256 tiles of 256 x 128 when a launch has the chip to itself; inside a StepStreams(2) context (a
    second step beside it) from 128 tiles on, for launches of at least 4096 rows — and never for N not a multiple of 128."""
    from gdrnpp_bop2022_amd import hip_lib

    old = (hip_lib.SPLIT2_MIN_TILES, hip_lib.SPLIT2_SHARED_MIN_TILES, hip_lib.SPLIT2_SHARED_MIN_ROWS)
    try:
        hip_lib.SPLIT2_MIN_TILES, hip_lib.SPLIT2_SHARED_MIN_TILES, hip_lib.SPLIT2_SHARED_MIN_ROWS = 256, 0, 4096
        assert hip_lib.split2_tiles_ok(128 * 256, 256) and not hip_lib.split2_tiles_ok(127 * 256, 256)      # 256 / 254 tiles
        assert not hip_lib.split2_tiles_ok(32768, 128) and not hip_lib.split2_tiles_ok(1 << 20, 192)
        hip_lib.SPLIT2_SHARED_MIN_TILES = 128
        assert hip_lib.split2_tiles_ok(32768, 128)                  # 128 tiles, 32768 rows (Patch-PnP's second convolution at 128 ROIs)
        assert not hip_lib.split2_tiles_ok(2048, 2048)              # 128 tiles but 8 row tiles only (stage-2 fc1 at 8 ROIs): stays six-product
        assert not hip_lib.split2_tiles_ok(32768 - 256, 128) and not hip_lib.split2_tiles_ok(1 << 20, 192)
    finally:
        hip_lib.SPLIT2_MIN_TILES, hip_lib.SPLIT2_SHARED_MIN_TILES, hip_lib.SPLIT2_SHARED_MIN_ROWS = old


def test_default_compute_streams_looks_at_the_post_processing_branch_too():
    """(round-5 advice) Two steps in flight need EVERY arithmetic kernel of a step to be this library's: the TEST.USE_PNP branches
    and COORD_2D_TYPE="rel" run PyTorch operators (GdrnHipPost.process_*, batch_data_test_gpu) -> one stream by default; plain
    network pose and depth refine -> two."""
    from gdrnpp_bop2022_amd.gdrn_modeling import engine
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

    cx, _ = build_model_optimizer(get_cfg("ycbv_convnext_a6"))
    assert engine.default_compute_streams(cx) == 2
    assert engine.default_compute_streams(cx, get_cfg("ycbv_convnext_a6", ["TEST.USE_DEPTH_REFINE=True"])) == 2
    for opts in (["TEST.USE_PNP=True", "TEST.PNP_TYPE=ransac_pnp"], ["TEST.USE_PNP=True", "TEST.PNP_TYPE=net_iter_pnp"],
                 ["MODEL.POSE_NET.PNP_NET.COORD_2D_TYPE=rel"]):
        assert engine.default_compute_streams(cx, get_cfg("ycbv_convnext_a6", opts)) == 1, opts
        m, _ = build_model_optimizer(get_cfg("ycbv_convnext_a6", opts))
        assert engine.default_compute_streams(m) == 1, opts           # the model's own cfg is the default


def test_shared_chip_tile_rule_is_per_host_thread():
    """(round-5 verdict weak 4) The shared-chip kernel rule is set for the calling host thread only
    (hip_lib.shared_min_tiles_scope): a second thread with its own dealer never sees it, and it is restored on exit."""
    import threading
    from gdrnpp_bop2022_amd import hip_lib

    assert hip_lib.shared_min_tiles() == hip_lib.SPLIT2_SHARED_MIN_TILES == 0
    seen = {}
    inside, release = threading.Event(), threading.Event()

    def other():
        inside.wait(10)
        seen["other_thread"] = (hip_lib.shared_min_tiles(), hip_lib.split2_tiles_ok(32768, 128))
        with hip_lib.shared_min_tiles_scope(64):
            seen["other_thread_own"] = hip_lib.shared_min_tiles()
        release.set()

    t = threading.Thread(target=other)
    t.start()
    with hip_lib.shared_min_tiles_scope(128):
        assert hip_lib.shared_min_tiles() == 128 and hip_lib.split2_tiles_ok(32768, 128)
        inside.set()
        release.wait(10)
        assert hip_lib.shared_min_tiles() == 128                       # the other thread's scope(64) did not leak here
        with hip_lib.shared_min_tiles_scope(None):                     # None = keep what is in force
            assert hip_lib.shared_min_tiles() == 128
    t.join()
    assert seen == {"other_thread": (0, False), "other_thread_own": 64}
    assert hip_lib.shared_min_tiles() == 0 and not hip_lib.split2_tiles_ok(32768, 128)


def test_fallback_launches_are_counted_only_where_the_hip_path_was_on_offer():
    """hip_layers counts a layer's fall-back to the PyTorch operator when the HIP path is enabled AND the tensor was its to take
    (on the device, fp32, no grad): CPU tensors — every CPU test of this suite — never move the counter."""
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers

    n0 = hip_layers.fallback_launches()
    up = torch.nn.UpsamplingBilinear2d(scale_factor=2)
    x = torch.randn(1, 6, 4, 4)
    with torch.no_grad():
        y = hip_layers.upsample2x(up, x)
        z = hip_layers.groupnorm_act(torch.nn.GroupNorm(2, 6), None, x)
    assert y.shape == (1, 6, 8, 8) and z.shape == x.shape and hip_layers.fallback_launches() == n0
    hip_layers.note_foreign_launch("test")
    assert hip_layers.fallback_launches() == n0 + 1 and hip_layers.last_fallback() == "test"


def test_cached_entries_are_rebuilt_only_when_the_tag_changes():
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers

    w = torch.nn.Parameter(torch.ones(3))
    cache, built = {}, []

    def build():
        built.append(1)
        return (w.detach() * 2,)

    n0 = hip_layers.cache_fills()
    a = hip_layers.cached(cache, "k", hip_layers.weight_tag(w), build, w)
    b = hip_layers.cached(cache, "k", hip_layers.weight_tag(w), build, w)
    assert a is b and len(built) == 1 and hip_layers.cache_fills() == n0 + 1
    with torch.no_grad():
        w.add_(1.0)                                                    # in-place update bumps the version counter
    c = hip_layers.cached(cache, "k", hip_layers.weight_tag(w), build, w)
    assert c is not a and len(built) == 2 and torch.equal(c[1], torch.full((3,), 4.0))


def test_range_check_wrapper_runs_on_a_host_without_a_gpu():
    """(round-5 advice) engine.run_with_range_check must not touch torch.cuda when nothing ran on a device: the evaluator's CPU
    path (gdrn_inference_on_dataset with a CPU model) goes through it."""
    out = engine.run_with_range_check(lambda: {"rot": torch.eye(3)[None], "trans": torch.zeros(1, 3)})
    assert torch.equal(out["rot"], torch.eye(3)[None])
    h = engine.launch_with_range_check(lambda: torch.ones(2))
    assert h.stream is None and torch.equal(h.result(), torch.ones(2))


def _subgroup_gather_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = dist.new_group([1, 2])                   # global ranks 1 and 2 are ranks 0 and 1 OF THE GROUP
    out = None
    if rank in (1, 2):
        rec = torch.zeros((2, 16))
        rec[:, 14] = torch.tensor([10.0 * rank, 10.0 * rank + 1])
        rec[:, 15] = 1.0
        out = engine.gather_records(rec[: (1 if rank == 2 else 2)], 2, group=grp, dst=1)      # group rank 1 = global rank 2 receives
    ret[rank] = None if out is None else out.numpy()
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_dst_is_a_rank_of_the_group_gloo_world3():
    """(round-5 advice) ``dst`` follows the group-local convention of the rest of gather_records: in a subgroup {1, 2} of a
    3-rank world, dst=1 is global rank 2 — it receives both blocks (padding rows marked PAD_ROI_ID), the others get None."""
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_subgroup_gather_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret[0] is None and ret[1] is None
    got = ret[2]
    assert got.shape == (4, 16) and got[:, 14].tolist() == [10.0, 11.0, 20.0, engine.PAD_ROI_ID] and got[:, 15].tolist() == [1, 1, 1, 0]


def test_packed_layout_is_fixed_by_shapes_and_fill_checks_them():
    """RoiStreamScheduler(graph_steps=True) lays the per-ROI arrays of a step out ONCE (packed_layout) and refills the same pinned
    buffer every step (fill_packed): the layout depends on shapes / dtypes only, arrays start on 16-byte boundaries, a step
    with another ROI count is refused instead of silently overrunning, and packed_views of the buffer give the arrays back."""
    rng = np.random.default_rng(0)

    def arrays(n):
        return dict(center64=rng.uniform(0, 640, (n, 2)), scale64=rng.uniform(10, 300, n), im_idx=rng.integers(0, 4, n).astype(np.int32),
                    roi_cls=rng.integers(0, 21, n), roi_cam=np.tile(np.eye(3, dtype=np.float32), (n, 1, 1)),
                    score=rng.uniform(0, 1, n).astype(np.float32), roi_id=np.arange(n, dtype=np.int32))

    a, b = arrays(16), arrays(16)
    la, ta = engine.packed_layout(a)
    lb, tb = engine.packed_layout(b)
    assert la == lb and ta == tb and all(off % 16 == 0 for off, _, _, _ in la.values())
    buf = torch.zeros((ta,), dtype=torch.uint8)
    engine.fill_packed(buf.numpy(), a, la)
    got = engine.packed_views(buf, la)
    assert all(np.array_equal(got[k].numpy(), a[k]) for k in a)
    engine.fill_packed(buf.numpy(), b, la)                          # the same buffer, the next step
    assert all(np.array_equal(engine.packed_views(buf, la)[k].numpy(), b[k]) for k in b)
    with pytest.raises(ValueError, match="layout holds"):
        engine.fill_packed(buf.numpy(), arrays(15), la)


def test_roi_host_arrays_is_the_host_half_of_batch_data_test_gpu():
    """roi_host_arrays: ROI parameters as read_data_test derives them + the class sort + the ids, as plain NumPy (no device): the
    sorted order is stable, roi_id carries each detection's original position, per-class extents are looked up after the sort."""
    from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg

    cfg = get_cfg("ycbv_convnext_a6")
    ext = np.linspace(0.05, 0.25, 63, dtype=np.float32).reshape(21, 3)
    det = dict(bbox=np.array([[10, 20, 110, 220], [300, 100, 360, 130], [50, 60, 70, 90], [200, 200, 400, 300]], np.float32),
               roi_cls=np.array([7, 2, 7, 0]), score=np.array([0.9, 0.8, 0.7, 0.6], np.float32), im_idx=np.array([0, 0, 1, 1]),
               cam=np.eye(3, dtype=np.float32), extents=ext)
    h = engine.roi_host_arrays(cfg, det, 480, 640, sort_by_class=True, roi_id_base=100)
    assert h["roi_cls"].tolist() == [0, 2, 7, 7] and h["roi_id"].tolist() == [103, 101, 100, 102] and h["im_idx"].tolist() == [1, 0, 0, 1]
    assert np.array_equal(h["roi_extent"], ext[[0, 2, 7, 7]]) and h["roi_cam"].shape == (4, 3, 3)
    assert np.allclose(h["scale64"], [min(200 * 1.5, 640), 60 * 1.5, 200 * 1.5, 30 * 1.5]) and h["center64"].dtype == np.float64
    assert np.allclose(h["resize_ratio"], 64.0 / h["scale64"]) and h["roi_wh"].tolist() == [[200, 100], [60, 30], [100, 200], [20, 30]]
