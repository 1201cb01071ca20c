"""-m gpu: ROI packing between the reference's image loop and the step size the kernels want (engine.RoiStreamScheduler,
gdrn_evaluator.packed_loader).  A stream of images with 0 .. 30 detections each goes through the scheduler in steps of exactly
P ROIs (GPU crop -> forward -> depth refine -> records, resolved one launch late); every image gets the records the same image
gets when it is run on its own (the reference's schedule) — within the path's tolerance, since a lone image runs the
small-batch kernels (six products, split-K) and a packed step the large-batch ones."""
import numpy as np
import pytest
import torch

from gdrnpp_bop2022_amd import hip_lib, synthetic as S
from gdrnpp_bop2022_amd.gdrn_modeling import engine
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def setup(hip):
    cfg = get_cfg("ycbv_convnext_a6", opts=["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True"])
    torch.manual_seed(0)
    model, _ = build_model_optimizer(cfg)
    model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], 5), strict=True)
    with torch.no_grad():
        model.pnp_net.fc_t.bias.copy_(torch.tensor([0.0, 0.0, 1.5 * float(S.YCBV_K[0, 0]) * 0.19 / 64.0]))
    rng = np.random.default_rng(9)
    verts, faces, ext = S.make_models(21, rng, 2)
    post = engine.GdrnHipPost(cfg, hip_lib.MeshSet(verts, faces, DEV))
    g = torch.Generator(device=DEV).manual_seed(1)
    stream = []
    for i, n in enumerate([5, 0, 30, 17, 3, 22, 9, 30, 11, 1]):
        det = S.make_detections(max(n, 1), 21, ext, rng)
        x1y1 = det["roi_center"] - det["roi_wh"] / 2
        d = dict(bbox=np.concatenate([x1y1, x1y1 + det["roi_wh"]], 1).astype(np.float32)[:n], roi_cls=det["roi_cls"][:n],
                 score=det["score"][:n], cam=S.YCBV_K.astype(np.float32), extents=ext)
        img = torch.randint(0, 256, (S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=DEV, generator=g)
        dep = torch.rand((S.IM_H, S.IM_W), device=DEV, generator=g) + 0.5
        stream.append((f"48/{i}", img, dep, d))
    return cfg, model, post, stream


def _alone(cfg, model, post, img, dep, det):
    if len(det["roi_cls"]) == 0:
        return np.zeros((0, 16), np.float32)
    batch = engine.batch_data_test_gpu(cfg, img[None], dep[None], dict(det, im_idx=np.zeros(len(det["roi_cls"]), np.int64)))
    return engine.inference_step(model, post, batch, torch.arange(len(det["roi_cls"]), dtype=torch.int32, device=DEV)).cpu().numpy()


@pytest.mark.parametrize("P,in_flight", [(32, 2), (48, 1)])
def test_stream_scheduler_returns_every_image_its_own_records(setup, P, in_flight):
    cfg, model, post, stream = setup
    sch = engine.RoiStreamScheduler(cfg, model, post, rois_per_step=P, max_in_flight=in_flight)
    out = {}
    for key, img, dep, det in stream:
        for k, rec, sec in sch.push(key, img, dep, det):
            assert k not in out and sec >= 0.0
            out[k] = rec
    for k, rec, sec in sch.flush():
        out[k] = rec
    total = sum(len(d["roi_cls"]) for _, _, _, d in stream)
    assert sorted(out) == sorted(k for k, _, _, _ in stream) and sch.steps_launched == -(-total // P)
    for key, img, dep, det in stream:
        n = len(det["roi_cls"])
        rec = out[key]
        assert rec.shape == (n, 16)
        if n == 0:
            continue
        want = _alone(cfg, model, post, img, dep, det)
        assert np.array_equal(rec[:, 13], det["roi_cls"].astype(np.float32)) and np.array_equal(rec[:, 12], det["score"])
        assert (rec[:, 15] == 1).all() and np.isfinite(rec).all()
        assert np.abs(rec[:, :9] - want[:, :9]).max() <= 1e-4, key                    # R
        assert np.abs(rec[:, 9:12] - want[:, 9:12]).max() <= 1e-4, key               # t (metres)


def test_launch_next_runs_exact_steps_and_resolves_one_launch_late(setup):
    """bench.py's use: launch_next(feeder) launches exactly one full step per call; resolving it a call later returns that
    step's [P,16] records with every stream id exactly once."""
    import itertools

    cfg, model, post, stream = setup
    P = 32
    sch = engine.RoiStreamScheduler(cfg, model, post, rois_per_step=P)
    counter = itertools.count()
    feeder = ((f"{k}#{next(counter)}", i, d, det) for k, i, d, det in itertools.cycle(stream))
    pending, ids = None, []
    for _ in range(5):
        res = sch.launch_next(feeder)
        if pending is not None:
            ids.append(pending().cpu().numpy()[:, 14])
        pending = res
    ids.append(pending().cpu().numpy()[:, 14])
    assert sch.steps_launched == 5
    assert np.array_equal(np.sort(np.concatenate(ids)), np.arange(5 * P))


def test_host_fed_stream_equals_the_device_resident_one_bit_for_bit(setup):
    """The reference's loader hands over HOST arrays (data_loader.py:754-797): the same stream pushed from pinned host memory —
    copied on the scheduler's copy stream, the crop kernel ordered behind the copies by events — returns bit-identical records,
    the copies are timed, and part of their time lies under the previous step's kernels (device timeline)."""
    cfg, model, post, stream = setup
    P = 32

    def run(feed, **kw):
        sch = engine.RoiStreamScheduler(cfg, model, post, rois_per_step=P, **kw)
        out = {}
        for key, img, dep, det in feed:
            for k, rec, _ in sch.push(key, img, dep, det):
                out[k] = rec
        for k, rec, _ in sch.flush():
            out[k] = rec
        return sch, out

    _, want = run(stream)
    host = [(k, img.cpu().pin_memory(), dep.cpu().pin_memory(), det) for k, img, dep, det in stream]
    sch, got = run(host, time_h2d=True)
    assert sorted(got) == sorted(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    n_img = sum(1 for _, _, _, d in stream if len(d["roi_cls"]))
    assert sch.h2d_bytes == n_img * (S.IM_H * S.IM_W * 3 + S.IM_H * S.IM_W * 4)
    tl = sch.h2d_timeline()
    assert tl["images"] == n_img and tl["steps"] == sch.steps_launched and tl["h2d_ms"] > 0 and 0.0 <= tl["overlapped_frac"] <= 1.0
    assert not sch._h2d_ready and not sch._images                      # nothing of the stream is kept alive


def test_a_flagged_step_inside_the_scheduler_is_repeated_with_six_products(setup):
    """A three-product launch of a step leaves the fp16x2 range (one block's LayerNorm output shrunk to the 1e-3 scale): the
    scheduler's resolve repeats that step with six products on the compute stream and reads the REPEATED records back (the
    side-stream copy waits for the compute stream then, not for the step's first event) — every image gets exactly the records
    of the six-product mode; steps launched after the verdict run with the layer demoted and need no repeat."""
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers

    cfg, model, post, stream = setup
    P = 32

    def run():
        sch = engine.RoiStreamScheduler(cfg, model, post, rois_per_step=P)
        out = {}
        for key, img, dep, det in stream:
            for k, rec, _ in sch.push(key, img, dep, det):
                out[k] = rec
        for k, rec, _ in sch.flush():
            out[k] = rec
        return out

    blk = model.backbone.stages_0.blocks[1]
    w_ok, b_ok = blk.norm.weight.detach().clone(), blk.norm.bias.detach().clone()
    old = hip_layers.gemm_products()
    try:
        with torch.no_grad():
            blk.norm.weight.mul_(1e-3)
            blk.norm.bias.mul_(1e-3)
        hip_layers.set_gemm_products(6)
        want = run()
        hip_layers.set_gemm_products(3)
        hip_layers.reset_x3_demotions()
        reruns = engine.range_reruns()
        got = run()
        # the steps launched before the first verdict was read (up to max_in_flight + 1 = 3) run the layer on three products too
        assert 1 <= engine.range_reruns() - reruns <= 3 and len(hip_layers.x3_demoted()) >= 1
        assert sorted(got) == sorted(want)
        first = [k for k, _, _, d in stream if len(d["roi_cls"])][:2]                         # images of the flagged (first) step
        for k in first:
            assert np.array_equal(got[k], want[k]), k                                           # the repeat IS the six-product step
        for k in want:
            assert np.abs(got[k][:, :12] - want[k][:, :12]).max(initial=0.0) <= 1e-4, k
    finally:
        with torch.no_grad():
            blk.norm.weight.copy_(w_ok)
            blk.norm.bias.copy_(b_ok)
        hip_layers.set_gemm_products(old)
        hip_layers.reset_x3_demotions()
        engine._X3_OVERFLOW_STEPS = 0
