"""-m gpu: two independent steps in flight on two HIP streams (engine.StepStreams) give the records of the single-stream schedule,
bit for bit — and the kernel that first did not (round 5): the 2x2-block upsample, SLP-packed by hipcc into v_pk_add_f32 ...
op_sel:[0,1], returned wrong values in lanes 48..63 whenever its waves shared a SIMD with the split GEMM's f16 MFMAs
(tools/pk_hazard_probe.py, profiles/r05p_pk_hazard_probe.txt).  The library is built with -fno-slp-vectorize and checked at link
time (tools/check_isa_hazards.py, tests/test_capi.py); these tests are the behaviour."""
import numpy as np
import pytest
import torch

from gdrnpp_bop2022_amd import hip_lib, synthetic as S
from gdrnpp_bop2022_amd.gdrn_modeling import engine, hip_layers
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg
from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def same_kernels_on_one_and_two_streams():
    """StepStreams(2) lets a launch of 128..255 tiles take the three-product form (a second step shares the chip); these tests
    compare SCHEDULES bit for bit, so the one-stream runs choose their kernels by the same rule."""
    old = hip_lib.SPLIT2_SHARED_MIN_TILES
    hip_lib.SPLIT2_SHARED_MIN_TILES = hip_lib.SPLIT2_MIN_TILES // 2
    yield
    hip_lib.SPLIT2_SHARED_MIN_TILES = old


def test_upsample_beside_the_conv_gemm_of_another_stream_is_bitwise_the_serial_result(hip):
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(DEV)
    xg = torch.randn(64, 256, 32, 32, device=DEV).contiguous(memory_format=torch.channels_last)      # 256 tiles of 256 x 128: the three-product kernel
    xu = torch.randn(64, 256, 32, 32, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        n0 = hip_lib.x3_launch_count()
        yg_ref = hip_layers.conv2d(conv, xg).clone()
        assert hip_lib.x3_launch_count() > n0, "the companion must be the three-product (f16 MFMA) GEMM"
        yu_ref = hip_lib.upsample_bilinear2x(xu).clone()
        assert torch.equal(yu_ref, torch.nn.functional.interpolate(xu, scale_factor=2, mode="bilinear", align_corners=True)) or \
            float((yu_ref - torch.nn.functional.interpolate(xu, scale_factor=2, mode="bilinear", align_corners=True)).abs().max()) < 2e-6
        torch.cuda.synchronize()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        for rep in range(6):
            with torch.cuda.stream(sb):
                ygs = [hip_layers.conv2d(conv, xg) for _ in range(4)]
            with torch.cuda.stream(sa):
                yus = [hip_lib.upsample_bilinear2x(xu) for _ in range(6)]
            torch.cuda.synchronize()
            assert all(torch.equal(y, yu_ref) for y in yus), f"rep {rep}: upsample differs beside the GEMM"
            assert all(torch.equal(y, yg_ref) for y in ygs), f"rep {rep}: GEMM differs beside the upsample"


@pytest.fixture(scope="module")
def setup(hip):
    cfg = get_cfg("ycbv_convnext_a6", opts=["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True"])
    torch.manual_seed(0)
    model, _ = build_model_optimizer(cfg)
    model.load_state_dict(S.seeded_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()], 5), strict=True)
    with torch.no_grad():
        model.pnp_net.fc_t.bias.copy_(torch.tensor([0.0, 0.0, 1.5 * float(S.YCBV_K[0, 0]) * 0.19 / 64.0]))
    model = model.to(DEV).eval()
    rng = np.random.default_rng(11)
    verts, faces, ext = S.make_models(21, rng, 2)
    post = engine.GdrnHipPost(cfg, hip_lib.MeshSet(verts, faces, DEV))
    g = torch.Generator(device=DEV).manual_seed(2)
    b = 64                                       # 64 ROIs: the three-product kernels (256-tile rule) carry the backbone and the head
    batches = []
    for k in range(3):
        det = S.make_detections(b, 21, ext, rng)
        x1y1 = det["roi_center"] - det["roi_wh"] / 2
        d = dict(bbox=np.concatenate([x1y1, x1y1 + det["roi_wh"]], 1).astype(np.float32), roi_cls=det["roi_cls"], score=det["score"],
                 cam=S.YCBV_K.astype(np.float32), extents=ext, im_idx=np.zeros(b, np.int64))
        img = torch.randint(0, 256, (1, S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=DEV, generator=g)
        dep = torch.rand((1, S.IM_H, S.IM_W), device=DEV, generator=g) + 0.5
        batches.append(engine.batch_data_test_gpu(cfg, img, dep, d, sort_by_class=True))
    return cfg, model, post, batches


def _run(model, post, batches, n_steps, streams):
    dealer = engine.StepStreams(streams)
    out, prev = [], None
    for i in range(n_steps):
        with dealer.next():
            cur = engine.inference_step_async(model, post, batches[i % len(batches)])
        if prev is not None:
            out.append(prev.result().clone())
        prev = cur
    out.append(prev.result().clone())
    torch.cuda.synchronize()
    return out, dealer


def test_two_steps_in_flight_give_the_single_stream_records_bit_for_bit(setup):
    cfg, model, post, batches = setup
    n0 = hip_lib.x3_launch_count()
    one, _ = _run(model, post, batches, 9, 1)
    assert hip_lib.x3_launch_count() > n0, "the steps must run the three-product kernels (the MFMAs the hazard needs)"
    for rep in range(3):
        two, dealer = _run(model, post, batches, 9, 2)
        assert len(dealer.streams) == 2 and dealer.streams[0] != dealer.streams[1]
        for i, (a, c) in enumerate(zip(one, two)):
            assert torch.equal(a, c), f"rep {rep} step {i}: max |diff| {float((a - c).abs().max()):.3e}"
    ok = one[0][:, 15] > 0.5
    assert ok.all() and torch.isfinite(one[0]).all()
    assert torch.equal(one[0], one[3]) and not torch.equal(one[0], one[1])       # the same batch again / another batch


def test_handles_belong_to_their_stream_and_a_flagged_step_is_repeated_there(setup):
    """The range words are per stream: a step on stream A whose three-product launches report a row below the fp16x2 range (here:
    the word is set on A behind the step's kernels, where a launch would have set it) is repeated with six products ON A, while
    the step in flight on stream B is not (its words are clean) and keeps its records."""
    cfg, model, post, batches = setup
    one, _ = _run(model, post, batches, 2, 1)
    with hip_layers.forced_gemm_products(6):
        want = engine.inference_step(model, post, batches[0]).clone()
    dealer = engine.StepStreams(2)
    reruns0 = engine.range_reruns()
    step = torch.no_grad()(engine._step_closure(model, post, batches[0], batches[0]["roi_id"]))
    calls = []

    def flagged_step():
        out = step()
        calls.append(torch.cuda.current_stream())
        if len(calls) == 1:
            hip_lib._x3_flags()[0:1].fill_(hip_lib.X3_SMALL_ROWS)       # slot 0 (names no layer: nothing gets demoted); the current stream's words, stream-ordered behind the kernels
        return out

    try:
        with dealer.next():
            h_bad = engine.launch_with_range_check(flagged_step)
        with dealer.next():
            h_ok = engine.inference_step_async(model, post, batches[1])
        assert h_bad.stream == dealer.streams[0] and h_ok.stream == dealer.streams[1]
        r_ok = h_ok.result().clone()
        r_bad = h_bad.result().clone()          # resolved from the default stream: the repeat still goes to stream A
        torch.cuda.synchronize()
        assert h_bad.reran and not h_ok.reran and engine.range_reruns() == reruns0 + 1
        assert calls == [dealer.streams[0], dealer.streams[0]]
        assert torch.equal(r_ok, one[1])
        assert torch.equal(r_bad, want)                                   # the repeat IS the six-product step
        assert float((r_bad[:, :12] - one[0][:, :12]).abs().max()) <= 1e-4
        # stream A's words were cleared behind the flagged step: the next step there is clean
        with torch.cuda.stream(dealer.streams[0]):
            h = engine.inference_step_async(model, post, batches[1])
        assert torch.equal(h.result(), one[1]) and not h.reran
    finally:
        hip_layers.reset_x3_demotions()
        engine._X3_OVERFLOW_STEPS = 0


@pytest.mark.parametrize("host_fed", [False, True])
def test_scheduler_on_two_compute_streams_returns_the_single_stream_records(setup, host_fed):
    """RoiStreamScheduler(compute_streams=2): steps of a packed image stream alternate between two HIP streams (crop, forward,
    refine, records); every image gets bit for bit the records of the one-stream schedule — from device images of the caller's
    stream and from pinned host images copied on the scheduler's copy stream."""
    cfg, model, post, _ = setup
    rng = np.random.default_rng(5)
    g = torch.Generator(device=DEV).manual_seed(7)
    ext = np.asarray(S.make_models(21, np.random.default_rng(11), 2)[2])        # the fixture's extents (same seed)
    images = []
    for i, n in enumerate([20, 7, 30, 0, 25, 13, 30, 4, 18, 29, 11, 30, 2, 16]):
        det = S.make_detections(max(n, 1), 21, ext, rng)
        x1y1 = det["roi_center"] - det["roi_wh"] / 2
        d = dict(bbox=np.concatenate([x1y1, x1y1 + det["roi_wh"]], 1).astype(np.float32)[:n], roi_cls=det["roi_cls"][:n], score=det["score"][:n],
                 cam=S.YCBV_K.astype(np.float32), extents=ext)
        img = torch.randint(0, 256, (S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=DEV, generator=g)
        dep = torch.rand((S.IM_H, S.IM_W), device=DEV, generator=g) + 0.5
        if host_fed:
            img, dep = img.cpu().pin_memory(), dep.cpu().pin_memory()
        images.append((f"s/{i}", img, dep, d))

    def run(n_streams):
        sch = engine.RoiStreamScheduler(cfg, model, post, rois_per_step=64, max_in_flight=2, compute_streams=n_streams, device=torch.device(DEV, 0))
        out = {}
        for key, img, dep, det in images:
            for k, rec, _ in sch.push(key, img, dep, det):
                out[k] = rec
        for k, rec, _ in sch.flush():
            out[k] = rec
        torch.cuda.synchronize()
        return out, sch

    one, _ = run(1)
    for rep in range(2):
        two, sch = run(2)
        assert len(sch._dealer.streams) == 2 and sch.steps_launched == -(-sum(len(d["roi_cls"]) for _, _, _, d in images) // 64)
        assert sorted(two) == sorted(one) == sorted(k for k, _, _, _ in images)
        for k in one:
            assert np.array_equal(one[k], two[k]), (rep, k)
    assert all(np.isfinite(v).all() for v in one.values())


def test_steps_without_range_words_are_ordered_before_the_callers_stream_reads_them(setup):
    """Six-product steps (small batches, set_gemm_products(6)) carry no range words, so ``StepHandle.result()`` has nothing to wait
    for on the host: the caller's stream must then wait for the step ON THE DEVICE before it reads the records (here: ``.clone()``
    on the default stream right after ``result()``, with the next step already queued on the other compute stream)."""
    cfg, model, post, batches = setup
    with hip_layers.forced_gemm_products(6):
        n0 = hip_lib.x3_launch_count()
        one, _ = _run(model, post, batches, 7, 1)
        for rep in range(3):
            two, _ = _run(model, post, batches, 7, 2)
            for i, (a, c) in enumerate(zip(one, two)):
                assert torch.equal(a, c), f"rep {rep} step {i}"
        assert hip_lib.x3_launch_count() == n0, "six products were forced: no three-product launch, hence no range words"
    assert torch.isfinite(one[0]).all() and (one[0][:, 15] > 0.5).all()


def test_step_streams_are_chosen_so_that_they_really_overlap(hip):
    """HIP multiplexes streams over a few hardware queues: two streams on one queue run back to back and 'two steps in flight'
    silently becomes the one-stream schedule.  StepStreams probes a pair of spin kernels and draws from torch's stream pool until
    they overlap; a stream against itself shows what 'no overlap' measures."""
    dev = torch.device(DEV, 0)
    s = torch.cuda.Stream(dev)
    assert engine.streams_overlap_ratio(s, s) < 1.3
    for _ in range(6):          # later pairs of a process are the ones that used to collide
        d = engine.StepStreams(2, dev)
        assert len(d.streams) == 2 and d.streams[0] != d.streams[1]
        tried, ratio = d.overlap_probe[0]
        assert ratio > 1.6 and 1 <= tried <= 8, d.overlap_probe
        assert engine.streams_overlap_ratio(*d.streams) > 1.6
    assert engine.StepStreams(1, dev).streams == [None] and engine.StepStreams(1, dev).overlap_probe is None


@pytest.mark.parametrize("b,n_streams", [(8, 2), (64, 2), (16, 4)])
def test_two_hipgraphs_in_flight_give_the_eager_two_stream_records(setup, b, n_streams):
    """engine.GraphedStepStreams: one captured hipGraph per slot, slots dealt to the dealer's two HIP streams — two steps in
    flight without per-launch host work (the reference's own batch sizes are host-bound: 146 launches through ctypes per step).
    The graphs are captured with the dealer's shared-chip kernel rule, so their records equal the EAGER two-stream schedule's
    (and hence the one-stream one's) bit for bit: 64 ROIs (three-product kernels throughout the graphs, range words read back
    asynchronously) and 8 ROIs (mostly six-product kernels); also after new batches were copied into the slots' static buffers."""
    cfg, model, post, batches = setup
    bs = [{k: (v[:b].contiguous() if isinstance(v, torch.Tensor) and v.shape[:1] == batches[0]["roi_img"].shape[:1] else v) for k, v in bt.items()}
          for bt in batches]
    order = [0, 1, 0, 1, 2, 1, 2, 0]                       # slot = i % n_slots; from step 4 on other batches are loaded into the slots
    n_slots = n_streams                                    # (four streams: four hipGraphs in flight, engine.default_graph_streams)
    eager, _ = _run(model, post, [bs[k] for k in order], len(order), 2)
    n0 = hip_lib.x3_launch_count()
    gs = engine.GraphedStepStreams(model, post, [bs[order[j]] for j in range(n_slots)], compute_streams=n_streams)
    assert len(gs.dealer.streams) == n_streams and len({g.stream for g in gs.graphs}) == n_streams
    assert all(g.captures == 1 and g.foreign_launches == 0 for g in gs.graphs)
    assert b < 64 or (gs.graphs[0].uses_x3 and hip_lib.x3_launch_count() > n0)
    for rep in range(3):
        out, pend = [], []
        for i, k in enumerate(order):
            cur = gs.launch(i % n_slots, None if i < n_slots and rep == 0 else bs[k])
            pend.append(cur)
            if len(pend) >= n_slots:
                out.append(pend.pop(0).result())
        while pend:
            out.append(pend.pop(0).result())
        torch.cuda.synchronize()
        for i, (a, c) in enumerate(zip(eager, out)):
            assert torch.equal(a, c), f"rep {rep} step {i}: max |diff| {float((a - c).abs().max()):.3e}"
    assert all(g.captures == 1 for g in gs.graphs)
    # the synchronous single-graph form (GraphedInference.__call__) on the caller's stream gives the same records
    g1 = engine.GraphedInference(model, post, bs[0], bs[0]["roi_id"], warmup=1)
    assert torch.equal(g1(bs[2]), eager[4]) and torch.equal(g1(bs[0]), eager[0])


def test_a_graph_that_would_share_the_chip_refuses_foreign_kernels(hip):
    """GraphedInference(sharing=True) — what GraphedStepStreams builds when its dealer shares the chip — raises at capture when
    the step launched anything outside this library (here: Patch-PnP's ReLU / LeakyReLU activations as PyTorch operators)."""
    cfg = get_cfg("ycbv_convnext_a6", opts=["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True", "MODEL.POSE_NET.PNP_NET.INIT_CFG.act=relu"])
    torch.manual_seed(0)
    model, _ = build_model_optimizer(cfg)
    model = model.to(DEV).eval()
    rng = np.random.default_rng(11)
    verts, faces, ext = S.make_models(21, rng, 2)
    post = engine.GdrnHipPost(cfg, hip_lib.MeshSet(verts, faces, DEV))
    det = S.make_detections(8, 21, ext, rng)
    x1y1 = det["roi_center"] - det["roi_wh"] / 2
    d = dict(bbox=np.concatenate([x1y1, x1y1 + det["roi_wh"]], 1).astype(np.float32), roi_cls=det["roi_cls"], score=det["score"],
             cam=S.YCBV_K.astype(np.float32), extents=ext, im_idx=np.zeros(8, np.int64))
    g = torch.Generator(device=DEV).manual_seed(2)
    img = torch.randint(0, 256, (1, S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=DEV, generator=g)
    dep = torch.rand((1, S.IM_H, S.IM_W), device=DEV, generator=g) + 0.5
    batch = engine.batch_data_test_gpu(cfg, img, dep, d, sort_by_class=True)
    with pytest.raises(RuntimeError, match="outside this library"):
        engine.GraphedInference(model, post, batch, warmup=1, stream=torch.cuda.Stream(), sharing=True)
    g1 = engine.GraphedInference(model, post, batch, warmup=1)           # alone on a stream it is fine
    assert g1.foreign_launches > 0 and torch.isfinite(g1.replay()).all()


def test_a_graph_on_its_own_stream_that_leaves_the_range_is_repeated_there_and_captured_again(setup):
    """Two hipGraphs on two streams; before the third replay ONE layer's input shrinks to the 1e-3 scale (its LayerNorm affine is
    scaled in place): the graph whose replay trips SMALL_ROWS — read back asynchronously, looked at in GraphHandle.result() —
    returns the records of the six-product mode (bit for bit), demotes the layer and captures again ON ITS STREAM; the other graph,
    still holding the layer's three-product kernel, captures again before its next replay; after that both replay without repeats."""
    cfg, model, post, batches = setup
    blk = model.backbone.stages_2.blocks[9]
    reruns0 = engine.range_reruns()
    try:
        with torch.no_grad():
            gs = engine.GraphedStepStreams(model, post, batches[:2], compute_streams=2, warmup=1)
            healthy = [gs.launch(0).result().clone(), gs.launch(1).result().clone()]
            assert all(g.uses_x3 and g.captures == 1 for g in gs.graphs) and hip_layers.x3_demoted() == {}
            w_ok, b_ok = blk.norm.weight.clone(), blk.norm.bias.clone()
            blk.norm.weight.mul_(1e-3)
            blk.norm.bias.mul_(1e-3)
            with hip_layers.forced_gemm_products(6):
                want = [engine.inference_step(model, post, batches[k]).clone() for k in range(2)]
            h0, h1 = gs.launch(0), gs.launch(1)               # both in flight with the shrunken layer
            r0, r1 = h0.result().clone(), h1.result().clone()
            torch.cuda.synchronize()
            assert engine.range_reruns() >= reruns0 + 1 and len(hip_layers.x3_demoted()) >= 1
            assert torch.equal(r0, want[0]), "the flagged replay must return the six-product records"
            assert (r1[:, :12] - want[1][:, :12]).abs().max().item() <= 1e-4      # second graph: repeated too, or replayed after the demotion
            assert gs.graphs[0].captures == 2 and gs.graphs[0].stream == gs.dealer.streams[0]
            n_rer = engine.range_reruns()
            again = [gs.launch(0).result().clone(), gs.launch(1).result().clone()]
            torch.cuda.synchronize()
            assert engine.range_reruns() == n_rer and all(g.captures == 2 for g in gs.graphs)
            for k in range(2):
                assert (again[k][:, :12] - want[k][:, :12]).abs().max().item() <= 1e-4
            assert not torch.equal(again[0], healthy[0])
    finally:
        with torch.no_grad():
            blk.norm.weight.copy_(w_ok)
            blk.norm.bias.copy_(b_ok)
        hip_layers.reset_x3_demotions()
        engine._X3_OVERFLOW_STEPS = 0


def test_host_fed_images_report_when_their_pinned_source_may_be_overwritten(setup):
    """(round-5 advice) RoiStreamScheduler copies a host image with non_blocking=True from the caller's PINNED buffer;
    ``h2d_done_event(key)`` is what the caller waits for before recycling that buffer: overwriting it after the event gives the
    records of the untouched run, bit for bit."""
    cfg, model, post, _ = setup
    rng = np.random.default_rng(21)
    ext = np.asarray(S.make_models(21, np.random.default_rng(11), 2)[2])
    g = torch.Generator(device=DEV).manual_seed(4)
    dets, imgs = [], []
    for i in range(6):
        det = S.make_detections(24, 21, ext, rng)
        x1y1 = det["roi_center"] - det["roi_wh"] / 2
        dets.append(dict(bbox=np.concatenate([x1y1, x1y1 + det["roi_wh"]], 1).astype(np.float32), roi_cls=det["roi_cls"], score=det["score"],
                         cam=S.YCBV_K.astype(np.float32), extents=ext))
        imgs.append((torch.randint(0, 256, (S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=DEV, generator=g).cpu(),
                     (torch.rand((S.IM_H, S.IM_W), device=DEV, generator=g) + 0.5).cpu()))

    def run(recycle):
        sch = engine.RoiStreamScheduler(cfg, model, post, rois_per_step=64, device=torch.device(DEV, 0))
        buf_i, buf_d = torch.empty_like(imgs[0][0]).pin_memory(), torch.empty_like(imgs[0][1]).pin_memory()
        out = {}
        for i, (im, dp) in enumerate(imgs):
            if recycle:                                   # ONE pinned staging buffer for the whole stream
                buf_i.copy_(im); buf_d.copy_(dp)
                src = (buf_i, buf_d)
            else:
                src = (im.pin_memory(), dp.pin_memory())
            for k, rec, _ in sch.push(i, src[0], src[1], dets[i]):
                out[k] = rec
            if recycle:
                ev = sch.h2d_done_event(i)
                if ev is not None:
                    ev.synchronize()                      # ... now the next image may overwrite the buffer
        for k, rec, _ in sch.flush():
            out[k] = rec
        return out

    a, b = run(False), run(True)
    assert sorted(a) == sorted(b) == list(range(6))
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("host_fed", [False, True])
def test_scheduler_with_graph_steps_returns_the_eager_schedulers_records(setup, host_fed):
    """RoiStreamScheduler(graph_steps=True): every full step is a hipGraph replay of (GPU crop -> forward -> refine -> records) on
    static image / per-ROI buffers, four steps in flight; the short tail step of flush() runs eagerly.  Per image the records
    are those of the eager scheduler, bit for bit — from device images and from pinned host images; the slots are captured once."""
    cfg, model, post, _ = setup
    rng = np.random.default_rng(5)
    g = torch.Generator(device=DEV).manual_seed(7)
    ext = np.asarray(S.make_models(21, np.random.default_rng(11), 2)[2])
    images = []
    for i, n in enumerate([20, 7, 30, 0, 25, 13, 30, 4, 18, 29, 11, 30, 2, 16, 9, 22, 5, 27, 14, 30, 3, 19]):
        det = S.make_detections(max(n, 1), 21, ext, rng)
        x1y1 = det["roi_center"] - det["roi_wh"] / 2
        d = dict(bbox=np.concatenate([x1y1, x1y1 + det["roi_wh"]], 1).astype(np.float32)[:n], roi_cls=det["roi_cls"][:n], score=det["score"][:n],
                 cam=S.YCBV_K.astype(np.float32), extents=ext)
        img = torch.randint(0, 256, (S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=DEV, generator=g)
        dep = torch.rand((S.IM_H, S.IM_W), device=DEV, generator=g) + 0.5
        if host_fed:
            img, dep = img.cpu().pin_memory(), dep.cpu().pin_memory()
        images.append((f"s/{i}", img, dep, d))

    def run(graph_steps, n_streams):
        sch = engine.RoiStreamScheduler(cfg, model, post, rois_per_step=16, compute_streams=n_streams, graph_steps=graph_steps,
                                        device=torch.device(DEV, 0))
        out = {}
        for key, img, dep, det in images:
            for k, rec, _ in sch.push(key, img, dep, det):
                out[k] = rec
        for k, rec, _ in sch.flush():
            out[k] = rec
        torch.cuda.synchronize()
        return out, sch

    one, _ = run(False, 1)
    for rep in range(2):
        two, sch = run(True, 4)
        assert len(sch._dealer.streams) == 4 and sch.max_in_flight == 4
        slots = [s_ for s_ in sch._slots if s_ is not None]
        assert len(slots) >= 4 and all(s_["graph"].captures == 1 and s_["graph"].foreign_launches == 0 for s_ in slots)
        assert len({s_["graph"].stream for s_ in slots}) == 4
        assert sorted(two) == sorted(one) == sorted(k for k, _, _, _ in images)
        for k in one:
            assert np.array_equal(one[k], two[k]), (rep, k)
    assert all(np.isfinite(v).all() for v in one.values())
