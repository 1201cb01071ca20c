"""Inference engine: the reference's ``gdrn_inference_on_dataset`` + ``GDRN_Evaluator.process*`` hot loop
(core/gdrn_modeling/engine/gdrn_evaluator.py:155-239,461-573,575-585,668-809) re-scheduled for MI355X.

Reference schedule per image: forward -> D2H of all maps -> per-ROI Python loop (cv2.resize, vispy GL
render x2, NumPy compare) -> pickle all-gather.  Here, per batch of ROIs resident in HBM:

    forward (PyTorch-ROCm, no host sync)                                 a3
      -> gdrnpp_zoom_K                         (K_crop for the 64x64 maps)   a8.2
      -> gdrnpp_depth_refine                   (all iterations on chip)      a8 / a8.1
      -> gdrnpp_pack_pose_records              ([n,16] f32 records)          a13
      -> one RCCL all_gather of the fixed-shape records (multi-GPU only)

Nothing is copied to the host until the caller asks for the records.
"""
from __future__ import annotations

import threading

import torch
import torch.distributed as dist

from .. import hip_lib
from . import hip_layers


def shard_range(n: int, rank: int, world: int):
    """Contiguous ROI shard of rank ``rank`` — InferenceSampler's rule (core/utils/my_distributed_sampler.py:
    181-194) at ROI granularity: ceil(n/world)-sized blocks, the last ranks may get fewer (or none)."""
    shard = (n - 1) // world + 1 if n > 0 else 0
    begin = min(shard * rank, n)
    end = min(shard * (rank + 1), n)
    return begin, end


def coor_planes(cfg, out_dict: dict):
    """``get_out_coor`` (engine_utils.py:295-312) as three single-channel planes.  Regression heads (one channel per axis)
    pass through; the classification flavour (``XYZ_LOSS_TYPE`` CE / CE_coor: XYZ_BIN + 1 logits per axis) becomes
    argmax-bin / (XYZ_BIN - 1) with the background bin mapped to 0, as the reference does before any post-processing."""
    planes = [out_dict["coor_x"], out_dict["coor_y"], out_dict["coor_z"]]
    if all(p.shape[1] == 1 for p in planes):
        return [p.contiguous() for p in planes]
    nbin = int(cfg.MODEL.POSE_NET.GEO_HEAD.XYZ_BIN)
    hip_layers.note_foreign_launch("coor_planes: classification xyz decoded with torch argmax / where")
    out = []
    for p in planes:
        idx = torch.argmax(p, dim=1, keepdim=True)
        idx = torch.where(idx == nbin, torch.zeros_like(idx), idx)
        out.append((idx.to(torch.float32) / float(nbin - 1)).contiguous())
    return out


class GdrnHipPost:
    """Batched, device-resident replacement of ``GDRN_Evaluator.process / process_depth_refine``."""

    def __init__(self, cfg, meshes: "hip_lib.MeshSet | None" = None, z_near: float = 0.1, z_far: float = 100.0):
        self.cfg = cfg
        self.meshes = meshes
        self.z_near, self.z_far = z_near, z_far  # Renderer.set_cam defaults (render_vispy/renderer.py:126)
        net_cfg = cfg.MODEL.POSE_NET
        self.out_res = net_cfg.OUTPUT_RES
        mlt = net_cfg.LOSS_CFG.MASK_LOSS_TYPE
        if mlt == "L1":
            self.mask_type = 0
        elif mlt in ("BCE", "RW_BCE", "dice"):
            self.mask_type = 1
        elif mlt == "CE":        # two logits per pixel: get_out_mask takes the argmax (engine_utils.py:329-330)
            self.mask_type = 2
        else:
            raise NotImplementedError(f"MASK_LOSS_TYPE={mlt}")
        if cfg.TEST.USE_DEPTH_REFINE and meshes is None:
            raise ValueError("TEST.USE_DEPTH_REFINE needs the object meshes (gdrn_evaluator.py:64-84)")

    def mask_plane(self, out_dict: dict) -> torch.Tensor:
        """The mask map the kernels consume: raw logits for L1 / BCE (normalised / squashed inside the kernels), the argmax
        label for the CE flavour (``get_out_mask``, engine_utils.py:315-333)."""
        m = out_dict["mask"]
        if self.mask_type == 2:
            hip_layers.note_foreign_launch("GdrnHipPost.mask_plane: CE mask decoded with torch argmax")
            m = torch.argmax(m, dim=1, keepdim=True).to(torch.float32)
        return m.contiguous()

    def process_depth_refine(self, batch: dict, out_dict: dict) -> torch.Tensor:
        """-> refined translation f64[b,3]; rotation is unchanged (gdrn_evaluator.py:559-561)."""
        cfg = self.cfg
        b = out_dict["trans"].shape[0]
        K_crop = hip_lib.zoom_K(batch["roi_cam"].reshape(b, 9).contiguous(), batch["roi_center"].contiguous(),
                                batch["scale"].reshape(b).contiguous(), self.out_res)
        cx, cy, cz = coor_planes(cfg, out_dict)
        return hip_lib.depth_refine(
            self.meshes, batch["roi_cls"].to(torch.int32), cx, cy, cz, self.mask_plane(out_dict),
            batch["roi_depth"].contiguous(), K_crop, out_dict["rot"].reshape(b, 9).contiguous(),
            out_dict["trans"].contiguous(), res=self.out_res, iters=cfg.TEST.DEPTH_REFINE_ITER,
            threshold=cfg.TEST.DEPTH_REFINE_THRESHOLD, mask_type=self.mask_type,
            use_coor_z=bool(cfg.TEST.USE_COOR_Z_REFINE), z_near=self.z_near, z_far=self.z_far)

    def process_correspondences(self, batch: dict, out_dict: dict, max_num_points: int = -1, generator=None):
        """2D-3D correspondences for the PnP variants (gdrn_evaluator.py:115-153,255-311), all ROIs at once.
        ``max_num_points >= 4`` keeps a uniformly random subset of that size per ROI, in random order (:146-152; the
        reference shuffles with Python's unseeded ``random``, here a device permutation from ``generator``)."""
        imwh = torch.stack([batch["im_W"], batch["im_H"]], 1).float().contiguous()
        cx, cy, cz = coor_planes(self.cfg, out_dict)
        if max_num_points >= 4:
            hip_layers.note_foreign_launch("GdrnHipPost.process_correspondences(max_num_points): torch rand / argsort / gather")
            count, sel_idx, img_pts, mdl_pts, m = self.process_correspondences(batch, out_dict)
            b, hw = sel_idx.shape
            keys = torch.rand((b, hw), device=count.device, generator=generator)
            keys = torch.where(torch.arange(hw, device=count.device)[None] < count[:, None], keys, torch.full_like(keys, 2.0))
            order = torch.argsort(keys, dim=1)[:, :max_num_points]                   # the first `count` entries are a permutation
            take = lambda t: torch.gather(t, 1, order[..., None].expand(-1, -1, t.shape[2])).contiguous()  # noqa: E731
            pad = hw - order.shape[1]
            padded = lambda t: torch.cat([t, t.new_zeros((b, pad) + t.shape[2:])], 1).contiguous() if pad > 0 else t  # noqa: E731
            return (torch.clamp(count, max=max_num_points), padded(torch.gather(sel_idx, 1, order)), padded(take(img_pts)),
                    padded(take(mdl_pts)), m)
        return hip_lib.decode_correspondences(
            cx, cy, cz, self.mask_plane(out_dict), batch["roi_coord_2d"].contiguous(), batch["roi_extent"].contiguous(), imwh,
            mask_type=self.mask_type, mask_thr=self.cfg.MODEL.POSE_NET.GEO_HEAD.MASK_THR_TEST)

    def process_net_and_pnp(self, batch: dict, out_dict: dict):
        """``TEST.USE_PNP`` with ``PNP_TYPE="net_iter_pnp"`` (gdrn_evaluator.py:241-371, pnp_type="iter"): decode the
        maps, compact the 2D-3D correspondences and run the net-initialised LM for every ROI on the device."""
        b = out_dict["trans"].shape[0]
        count, _, img_pts, mdl_pts, _ = self.process_correspondences(batch, out_dict)
        return hip_lib.pnp_iter_from_correspondences(
            img_pts, mdl_pts, count, batch["roi_cam"].reshape(b, 9).contiguous(),
            out_dict["rot"].reshape(b, 9).contiguous(), out_dict["trans"].contiguous())

    def process_pnp_ransac(self, batch: dict, out_dict: dict, iters: int = 100, draws=None):
        """``TEST.USE_PNP`` with ``PNP_TYPE="ransac_pnp"`` (gdrn_evaluator.py:373-459): decode, compact, then
        ``misc.pnp_v2(..., method=EPNP, ransac=True, ransac_reprojErr=3, ransac_iter=100)`` for every ROI on the device.
        ROIs with fewer than 4 correspondences get the reference's sentinel pose -100 (:445-447); a RANSAC that finds no
        model leaves R = I, t = 0 (status 0).  Exactly 4 correspondences: one P3P solve like OpenCV's (csrc/epnp_ransac.hip,
        p3p_4points).  -> (R f32[b,3,3], t f32[b,3], status i32[b])."""
        b = out_dict["trans"].shape[0]
        count, _, img_pts, mdl_pts, _ = self.process_correspondences(batch, out_dict)
        R, t, _, status, _ = hip_lib.epnp_ransac(img_pts, mdl_pts, count, batch["roi_cam"].reshape(b, 9).contiguous(),
                                                 iters=iters, reproj_err=3.0, draws=draws)
        hip_layers.note_foreign_launch("GdrnHipPost.process_pnp_ransac: torch where / full_like around the RANSAC kernels")
        few = (count < 4).view(b, 1)
        R = torch.where(few.view(b, 1, 1), torch.full_like(R, -100.0), R)
        t = torch.where(few, torch.full_like(t, -100.0), t)
        return R, t, status

    def process_net_and_ransac(self, batch: dict, out_dict: dict, rot_only: bool = False, draws=None):
        """``PNP_TYPE="net_ransac_pnp"`` (gdrn_evaluator.py:241-371 with pnp_type "ransac"): solvePnPRansac(EPNP, reprojErr 3, 20
        iterations) on the correspondences (the extrinsic guess is ignored by EPnP); the translation falls back to the network's
        when it moved by more than 1 m (:347-351); fewer than 4 correspondences or no model: the network pose (:355-358).
        ``rot_only``: RANSAC rotation with the network's translation — what the NAME ``net_ransac_pnp_rot`` suggests; NOT what the
        reference does under that name (``process_net_and_rot_pnp`` below is), kept for callers that want it."""
        b = out_dict["trans"].shape[0]
        count, _, img_pts, mdl_pts, _ = self.process_correspondences(batch, out_dict)
        R, t, _, status, _ = hip_lib.epnp_ransac(img_pts, mdl_pts, count, batch["roi_cam"].reshape(b, 9).contiguous(),
                                                 iters=20, reproj_err=3.0, draws=draws)
        hip_layers.note_foreign_launch("GdrnHipPost.process_net_and_ransac: torch norm / where around the RANSAC kernels")
        R_net, t_net = out_dict["rot"].reshape(b, 3, 3).float(), out_dict["trans"].float()
        use = ((count >= 4) & (status == 1)).view(b, 1)
        far = (t - t_net).norm(dim=1, keepdim=True) > 1.0
        t = t_net if rot_only else torch.where(use & ~far, t, t_net)
        R = torch.where(use.view(b, 1, 1), R, R_net)
        return R, t

    def process_net_and_rot_pnp(self, batch: dict, out_dict: dict):
        """``PNP_TYPE="net_ransac_pnp_rot"`` exactly as the reference runs it: ``process`` passes pnp_type "ransac_rot"
        (gdrn_evaluator.py:171-173), and ``process_net_and_pnp`` only takes its RANSAC branch for ``pnp_type == "ransac"`` (:319)
        — "ransac_rot" falls through to the ITERATIVE solvePnP seeded with the network pose, after which the network's translation
        is kept (:341-348).  So: rotation of the net-initialised LM, translation of the network (pinned by eval_pnp_golden.npz)."""
        R, _ = self.process_net_and_pnp(batch, out_dict)
        return R, out_dict["trans"].float()

    def process(self, batch: dict, out_dict: dict, roi_ids: torch.Tensor | None = None) -> torch.Tensor:
        """-> pose records f32[b,16] = R(9) | t(3, metres) | score | obj | roi_id | valid."""
        if out_dict["trans"].shape[0] == 0:       # an image / a rank without detections: nothing to launch (the reference
            return torch.zeros((0, 16), dtype=torch.float32, device=out_dict["trans"].device)   # skips such images)
        if self.cfg.TEST.USE_PNP:      # gdrn_evaluator.py:165-176 (the PnP variants return without the depth refinement)
            pnp_type = self.cfg.TEST.PNP_TYPE.lower()
            if pnp_type == "ransac_pnp":
                R, t, _ = self.process_pnp_ransac(batch, out_dict)
            elif pnp_type == "net_iter_pnp":
                R, t = self.process_net_and_pnp(batch, out_dict)
            elif pnp_type == "net_ransac_pnp":
                R, t = self.process_net_and_ransac(batch, out_dict)
            elif pnp_type == "net_ransac_pnp_rot":
                R, t = self.process_net_and_rot_pnp(batch, out_dict)
            else:
                raise NotImplementedError(f"TEST.PNP_TYPE={self.cfg.TEST.PNP_TYPE}")
            b = t.shape[0]
            return hip_lib.pack_pose_records(
                R.reshape(b, 9).contiguous(), None, t.contiguous(),
                batch["score"].float().contiguous() if "score" in batch else None,
                batch["roi_cls"].to(torch.int32).contiguous(), roi_ids)
        b = out_dict["trans"].shape[0]
        if self.cfg.TEST.USE_DEPTH_REFINE:
            # zoom_K -> refine -> pack in ONE launch (the reference: batch_data_inference_roi + the per-ROI loop +
            # pose_prediction_to_json, gdrn_evaluator.py:461-573)
            cfg = self.cfg
            cx, cy, cz = coor_planes(cfg, out_dict)
            return hip_lib.refine_to_records(
                self.meshes, batch["roi_cls"].to(torch.int32), cx, cy, cz, self.mask_plane(out_dict),
                batch["roi_depth"].contiguous(), batch["roi_cam"].reshape(b, 9).contiguous(), batch["roi_center"].contiguous(),
                batch["scale"].reshape(b).contiguous(), out_dict["rot"].reshape(b, 9).contiguous(), out_dict["trans"].contiguous(),
                score=batch["score"].float().contiguous() if "score" in batch else None, roi_id=roi_ids, res=self.out_res,
                iters=cfg.TEST.DEPTH_REFINE_ITER, threshold=cfg.TEST.DEPTH_REFINE_THRESHOLD, mask_type=self.mask_type,
                use_coor_z=bool(cfg.TEST.USE_COOR_Z_REFINE), z_near=self.z_near, z_far=self.z_far)
        t_ref = None
        return hip_lib.pack_pose_records(
            out_dict["rot"].reshape(b, 9).contiguous(), t_ref, out_dict["trans"].contiguous(),
            batch["score"].float().contiguous() if "score" in batch else None,
            batch["roi_cls"].to(torch.int32).contiguous(), roi_ids)


def class_sorted_order(roi_cls):
    """within a rank ROIs run sorted by class (consecutive 256-row tiles of the class-sliced output layer
    then share a weight slice, consecutive refine workgroups a mesh), the original index travels in the record.  Returns the
    STABLE permutation ``order`` with ``roi_cls[order]`` non-decreasing — ROIs of one class keep their detection order, like
    the reference's per-object ordering of load_detections_into_dataset (dataset_utils.py:202-227)."""
    import numpy as np

    if isinstance(roi_cls, torch.Tensor):
        return torch.sort(roi_cls.reshape(-1), stable=True).indices
    return np.argsort(np.asarray(roi_cls).reshape(-1), kind="stable")


# The detections dict of ``batch_data_test_gpu`` — which entries are per ROI and which are not is a CONTRACT, not a guess:
PER_ROI_DETECTION_KEYS = ("bbox", "im_idx", "roi_cls", "score", "time", "roi_id", "det_id", "scene_im_id", "inst_id")
GLOBAL_DETECTION_KEYS = ("extents", "obj_ids", "model_points", "sym_infos")   # per class / per dataset: never permuted
_GLOBAL_DETECTION_KEYS = GLOBAL_DETECTION_KEYS      # the round-3 name


def sort_detections_by_class(detections: dict, roi_id_base: int = 0, extra_per_roi_keys=(), extra_global_keys=()):
    """-> (detections with every per-ROI entry permuted into class order, roi_id i32[n] = ``roi_id_base`` + the position the
    ROI had before).

    Per-ROI entries: ``PER_ROI_DETECTION_KEYS`` + ``extra_per_roi_keys`` (arrays, tensors or lists whose leading dimension is
    the number of ROIs — anything else under such a key raises) and ``cam`` when it is [n,3,3].  Passed through untouched:
    ``GLOBAL_DETECTION_KEYS`` + ``extra_global_keys``, a shared ``cam`` [3,3], scalars, strings, None.  Any OTHER entry whose
    leading dimension happens to equal the number of ROIs is ambiguous (a per-class table when n == number of classes?) and
    raises ``KeyError`` naming the two arguments that resolve it — nothing is reordered on a guess.  Sorting the DETECTIONS
    costs nothing on the device: the crop kernel reads its ROI parameters in the new order, no ROI tensor is ever permuted."""
    import numpy as np

    order = class_sorted_order(np.asarray(detections["roi_cls"]))
    out = dict(detections)
    n = len(order)
    per_roi = set(PER_ROI_DETECTION_KEYS) | set(extra_per_roi_keys)
    glob = set(GLOBAL_DETECTION_KEYS) | set(extra_global_keys)

    def lead(v):
        if isinstance(v, torch.Tensor):
            return v.shape[0] if v.dim() >= 1 else None
        if isinstance(v, (str, bytes)) or v is None or np.isscalar(v):
            return None
        if isinstance(v, (list, tuple)):
            return len(v)
        a = np.asarray(v)
        return a.shape[0] if a.ndim >= 1 else None

    def permuted(v):
        if isinstance(v, torch.Tensor):
            return v[torch.as_tensor(order, device=v.device)]
        if isinstance(v, (list, tuple)) and not isinstance(v, np.ndarray) and any(isinstance(e, (str, bytes)) for e in v):
            return [v[i] for i in order]
        return np.asarray(v)[order]

    for k, v in detections.items():
        if k in glob:
            continue
        if k == "cam":
            nd = v.dim() if isinstance(v, torch.Tensor) else np.asarray(v).ndim
            if nd == 3:
                if lead(v) != n:
                    raise ValueError(f"detections['cam'] is per ROI ([n,3,3]) but has {lead(v)} entries for {n} ROIs")
                out[k] = permuted(v)
            continue
        if k in per_roi:
            if v is None:
                continue
            if lead(v) != n:
                raise ValueError(f"detections[{k!r}] is a per-ROI entry but has leading dimension {lead(v)} for {n} ROIs")
            out[k] = permuted(v)
        elif lead(v) == n:
            raise KeyError(f"detections[{k!r}] has as many entries as there are ROIs ({n}) but is neither a known per-ROI key nor a "
                           "known global one: pass it in extra_per_roi_keys (to be permuted with the ROIs) or extra_global_keys")
    return out, (roi_id_base + order).astype(np.int32)


def records_in_roi_order(rec: torch.Tensor) -> torch.Tensor:
    """Valid records of a (gathered) block ordered by their ``roi_id`` column — undoes the per-rank class sort and drops the
    padding rows of ``gather_records``."""
    rec = rec[rec[:, 15] > 0.5]
    return rec[torch.sort(rec[:, 14], stable=True).indices]


_X3_OVERFLOW_STEPS = 0          # steps of this process whose three-product kernels overflowed the fp16 range
X3_OVERFLOW_STEPS_TO_GIVE_UP = 3
_RANGE_RERUNS = 0               # steps repeated with six products (either side of the range); bench.py reports it


def range_reruns() -> int:
    return _RANGE_RERUNS


def _note_range_words(words: dict) -> None:
    """What a step's non-zero range words ({slot: word}, hip_lib.split2_range_words) change for the steps to come:
      * every layer with rows below the range stays on the six-product kernels (hip_layers.demote_x3);
      * of the layers reporting non-finite values the FIRST in launch order does (the others saw its inf / NaN pass through);
      * a model whose activations overflow step after step is not paid for twice for ever: after X3_OVERFLOW_STEPS_TO_GIVE_UP
        such steps the process stays on six products (with a warning)."""
    global _X3_OVERFLOW_STEPS
    hip_layers.demote_x3({s_: w for s_, w in words.items() if w & hip_lib.X3_SMALL_ROWS})
    over = hip_layers.x3_launch_order(s_ for s_, w in words.items() if w & hip_lib.X3_NONFINITE)   # slot order is not launch order
    if over:
        _X3_OVERFLOW_STEPS += 1
        first = [s_ for s_ in over if s_ > 0][:1]
        hip_layers.demote_x3({s_: hip_lib.X3_NONFINITE for s_ in first})
        if _X3_OVERFLOW_STEPS >= X3_OVERFLOW_STEPS_TO_GIVE_UP and hip_layers.gemm_products() == 3:
            import warnings
            hip_layers.set_gemm_products(6)
            warnings.warn(f"{_X3_OVERFLOW_STEPS} steps overflowed the fp16 range of the three-product GEMM kernels: staying on the "
                          "six-product kernels (hip_layers.set_gemm_products(3) switches back)")


def _six_product_rerun(run, words: dict):
    """Repeat a step with the six-product kernels after its three-product launches reported ``words``; the calling host thread
    only (hip_layers.forced_gemm_products), other threads / streams keep their setting."""
    global _RANGE_RERUNS
    _RANGE_RERUNS += 1
    _note_range_words(words)
    with hip_layers.forced_gemm_products(6):
        return run()


class StepHandle:
    """A launched step whose range words have not been looked at yet.  ``result()`` waits for the step (one event), reads the
    words from pinned host memory and — if a three-product launch left the range — repeats the step with six products.  Between
    launch and ``result()`` the host is free: launch the next step first and the check costs no device idle time."""

    def __init__(self, run, out, host_words=None, event=None, stream=None, done=None):
        self._run, self._out, self._host, self._event = run, out, host_words, event
        self.stream = stream                    # the stream the step was launched on (a repeat goes to the same one)
        self._done = done                       # event behind the step on that stream, for a step WITHOUT range words (nothing to wait for on the host)
        self.reran = False                      # result() repeated the step with six products

    def result(self):
        """The step's output.  A handle belongs to the stream it was launched on (StepStreams deals consecutive steps to
        different ones): a repeat is issued there, and the output is marked as used by the CALLER's current stream, which may
        be another one (its memory is then not handed to a later step of the launch stream while the caller still reads it)."""
        if self._event is not None:
            self._event.synchronize()
            words = hip_lib.range_words_of(self._host)
            self._event = self._host = None
            if words:
                if self.stream is not None and self.stream != torch.cuda.current_stream():
                    with torch.cuda.stream(self.stream):
                        self._out = _six_product_rerun(self._run, words)
                        done = torch.cuda.Event()
                        done.record()
                    torch.cuda.current_stream().wait_event(done)
                else:
                    self._out = _six_product_rerun(self._run, words)
                self.reran = True
        self._run = None
        if self.stream is not None and self.stream != torch.cuda.current_stream():
            if self._done is not None:          # no host wait happened above: the caller's stream waits for the step on the device
                torch.cuda.current_stream().wait_event(self._done)
            if isinstance(self._out, torch.Tensor) and self._out.is_cuda:
                self._out.record_stream(torch.cuda.current_stream())
        self._done = None
        return self._out


def _on_device(out) -> bool:
    """Does ``out`` (a tensor, or a dict / sequence of them) live on a GPU?"""
    if isinstance(out, torch.Tensor):
        return out.is_cuda
    if isinstance(out, dict):
        return any(_on_device(v) for v in out.values())
    if isinstance(out, (list, tuple)):
        return any(_on_device(v) for v in out)
    return False


def launch_with_range_check(run) -> StepHandle:
    """``run()`` (a forward, or a whole step) under the contract of the three-product GEMM kernels, without waiting: if any of
    them was launched, the stream's range words are copied to pinned host memory behind the work (and cleared on the stream, so
    the next step starts from zero) and an event marks the copy; ``StepHandle.result()`` does the rest.  Under hipGraph capture
    the check is the graph owner's (GraphedInference.replay)."""
    n_x3 = hip_lib.x3_launch_count()
    out = run()
    if hip_lib.x3_launch_count() == n_x3 and not (torch.cuda.is_available() and _on_device(out)):
        return StepHandle(None, out)             # a CPU run (gdrn_inference_on_dataset supports one): no stream, no event, no range words
    if torch.cuda.is_current_stream_capturing():
        return StepHandle(None, out)
    st = torch.cuda.current_stream()
    if hip_lib.x3_launch_count() == n_x3:       # six-product kernels only (small batches, --gemm-products 6): no words, no host wait in result()
        done = torch.cuda.Event()
        done.record()
        return StepHandle(None, out, stream=st, done=done)
    words = hip_lib._x3_flags()              # this stream's words: steps in flight on other streams have their own
    host = torch.empty(words.shape, dtype=words.dtype, pin_memory=True)
    host.copy_(words, non_blocking=True)
    words.zero_()
    ev = torch.cuda.Event()
    ev.record()
    return StepHandle(run, out, host, ev, stream=st)


def run_with_range_check(run):
    """The synchronous form: ``run()``, then its range check (one stream sync when three-product kernels were launched)."""
    return launch_with_range_check(run).result()


run_with_overflow_check = run_with_range_check     # the round-3 name


def _step_closure(model, post: "GdrnHipPost", batch: dict, roi_ids):
    def run():
        out_dict = model(
            batch["roi_img"], roi_classes=batch["roi_cls"], roi_cams=batch["roi_cam"], roi_whs=batch["roi_wh"],
            roi_centers=batch["roi_center"], resize_ratios=batch["resize_ratio"],
            roi_coord_2d=batch.get("roi_coord_2d"), roi_coord_2d_rel=batch.get("roi_coord_2d_rel"),
            roi_extents=batch.get("roi_extent"))
        return post.process(batch, out_dict, roi_ids)
    return run


@torch.no_grad()
def inference_step_async(model, post: GdrnHipPost, batch: dict, roi_ids: torch.Tensor | None = None) -> StepHandle:
    """Launch one pass of the hot path over one batch of ROIs and return without waiting for the device: ``.result()`` gives
    the f32[b,16] records (after the range check of the three-product kernels).  ``batch`` must stay untouched until then."""
    if roi_ids is None:
        roi_ids = batch.get("roi_id")
    if batch["roi_img"].shape[0] == 0:            # empty shard (shard_range may give trailing ranks nothing): the caller
        return StepHandle(None, torch.zeros((0, 16), dtype=torch.float32, device=batch["roi_img"].device))   # still reaches gather_records
    run = torch.no_grad()(_step_closure(model, post, batch, roi_ids))
    dealer = getattr(_DEALER_TLS, "dealer", None)         # set by StepStreams.next() around the launches of one step
    if dealer is None or not dealer.sharing():
        return launch_with_range_check(run)
    n_foreign = getattr(_DEALER_TLS, "foreign_at_entry", hip_layers.fallback_launches())   # counted from the dealer context's entry: the
    handle = launch_with_range_check(run)                                                   # ROI preparation in front of the step is part of it
    if hip_layers.fallback_launches() != n_foreign:
        # the step launched kernels that are not this library's (a layer fell back to a PyTorch operator on its shape) while another
        # step may be running MFMAs on the other stream: foreign packed-fp32 code is exactly what MI355X gets wrong there
        # (profiles/r05p_two_stream_hazard.md).  The dealer stops sharing the chip — loudly — and this step is repeated alone.
        dealer.stop_sharing(f"{hip_layers.fallback_launches() - n_foreign} launch(es) outside this library, last: {hip_layers.last_fallback()}")
        torch.cuda.synchronize(dealer.device)
        handle = launch_with_range_check(run)
    return handle


def inference_step(model, post: GdrnHipPost, batch: dict, roi_ids: torch.Tensor | None = None) -> torch.Tensor:
    """One pass of the hot path over one batch of ROIs (the unit ``bench.py`` times).  ``roi_ids`` (or ``batch["roi_id"]``,
    set by ``batch_data_test_gpu(sort_by_class=True)``) = the global index each record carries."""
    return inference_step_async(model, post, batch, roi_ids).result()


def default_compute_streams(model, cfg=None) -> int:
    """How many compute streams consecutive steps of ``model`` may safely share the chip on: 2 when every arithmetic kernel of a
    step is this library's — code that is built and link-checked to hold no packed-fp32 instruction of the form MI355X gets wrong
    beside another stream's MFMAs (csrc/Makefile) — else 1.  Decided in two layers:
      * statically, here: ConvNeXt backbone, HIP network layers on, split GEMMs, and (``cfg`` = the model's own by default) a
        post-processing branch that is one launch of this library — plain network pose or depth refine; the ``TEST.USE_PNP``
        branches (torch elementwise ops around the PnP kernels, GdrnHipPost.process_*) and ``COORD_2D_TYPE="rel"`` (torch
        arithmetic in batch_data_test_gpu) run PyTorch operators and get 1, like the ResNet backbone (MIOpen convolutions);
      * dynamically, in ``inference_step_async``: every layer that falls back to a PyTorch operator ON ITS SHAPE (another input
        size, another norm) is counted (hip_layers.fallback_launches); a step that moved the counter inside a sharing dealer
        makes the dealer stop sharing and is repeated alone."""
    from .backbones import ConvNeXtFeatures

    cfg = getattr(model, "cfg", None) if cfg is None else cfg
    ours = (hip_layers.is_enabled() and hip_layers.mlp_gemm() == "split" and isinstance(getattr(model, "backbone", None), ConvNeXtFeatures)
            and not torch.is_autocast_enabled())
    if ours and cfg is not None:
        ours = not bool(cfg.TEST.USE_PNP) and cfg.MODEL.POSE_NET.PNP_NET.COORD_2D_TYPE != "rel"
    return 2 if ours else 1


def default_graph_streams(model, cfg=None) -> int:
    """Compute streams for the hipGraph form of the step (``GraphedStepStreams``): 4 where steps may share the chip at all
    (``default_compute_streams`` == 2) — a graph replay costs the host ~0.1 ms instead of ~3 ms of launches, so the host can keep
    FOUR steps in flight, which is what HIP offers hardware queues for (a fifth stream shares a queue with one of the four:
    profiles/r06_graph_streams.md; eager launches cannot feed more than two, profiles/r05r_compute_streams.txt) — else 1."""
    return 4 if default_compute_streams(model, cfg) > 1 else 1


_DEALER_TLS = threading.local()      # .dealer: the StepStreams whose next() context the calling host thread is inside


def streams_overlap_ratio(s0, s1, micros: int = 200) -> float:
    """(time of a spin kernel on s0 and then one on s1, each alone) / (time of both launched together): ~2 when the two streams
    execute concurrently, ~1 when they share a hardware queue and run back to back.  ~1 ms of device time.  The spin kernel is
    this library's (gdrnpp_debug_spin: one wave waiting on the wall clock)."""
    dev = s0.device
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(s0):
        hip_lib.spin(micros)               # warm both paths (first launch on a fresh stream creates its queue)
    with torch.cuda.stream(s1):
        hip_lib.spin(micros)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(s0):                  # alone
        ev[0].record()
        hip_lib.spin(micros)
        ev[1].record()
    torch.cuda.synchronize(dev)
    alone = ev[0].elapsed_time(ev[1])
    with torch.cuda.stream(s0):                  # together: s1's kernel is launched while s0's spins
        ev[2].record()
        hip_lib.spin(micros)
    with torch.cuda.stream(s1):
        hip_lib.spin(micros)
        ev[3].record()
    torch.cuda.synchronize(dev)
    together = ev[2].elapsed_time(ev[3])
    return 2.0 * alone / max(together, 1e-6)


class StepStreams:
    """Consecutive steps are independent (each batch its own ROIs, its own records), so they need not queue behind each other
    on ONE stream: dealt round-robin to ``n`` compute streams, the second step's GEMMs fill the chip while the first one is in
    its narrow tail (8x8 stage, Patch-PnP, pose heads, depth refine: launches of a few workgroups each) — two steps in flight
    instead of one.  Measured at the headline batch: 25.0 -> 23.3 ms per 128-ROI step, records bit-equal to the single-stream
    schedule (profiles/r05q_two_streams.txt).

        streams = StepStreams(2)
        with streams.next():
            handle = inference_step_async(model, post, batch)     # launched on the dealt stream; handle.result() from anywhere

    Everything a step allocates comes from its stream's pool and its range words are that stream's (hip_lib._x3_flags), so two
    steps share nothing but the read-only weights.  What must NOT share the chip with the split GEMMs is packed fp32 code with
    op_sel swizzles (a hardware hazard, csrc/Makefile): the library is built without it and checked at link time."""

    def __init__(self, n: int = 2, device=None, priorities=None, allow_foreign: bool = False):
        if n < 1:
            raise ValueError("StepStreams needs at least one stream")
        self.allow_foreign = bool(allow_foreign)   # A/B only: keep sharing the chip although a step launched kernels this library cannot check
        self.stopped_sharing = None                # reason, once a step with foreign launches made this dealer fall back to ONE stream
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        pr = list(priorities) if priorities is not None else [0] * n     # (A/B only: equal priorities are what was measured best)
        self.streams = [None]
        self.overlap_probe = None               # [(candidates tried, overlap ratio)] per stream after the first: what the choice was based on
        if n > 1:
            # HIP multiplexes its streams over a few hardware queues; two streams that land on the SAME queue run one after the
            # other, and "two steps in flight" silently becomes the one-stream schedule (measured: a pair drawn later in a
            # process gave 24.97 instead of 23.49 ms per step, profiles/r05last_two_stream_soak.txt).  So every further stream is
            # drawn from torch's pool until a pair of spin kernels really overlaps with the first one.
            self.streams = [torch.cuda.Stream(self.device, priority=int(pr[0]))]
            self.overlap_probe = []
            for i in range(1, n):
                best = None
                for attempt in range(8):
                    cand = torch.cuda.Stream(self.device, priority=int(pr[i % len(pr)]))
                    ratio = min(streams_overlap_ratio(s_, cand) for s_ in self.streams)
                    if best is None or ratio > best[1]:
                        best = (cand, ratio)
                    if ratio > 1.6:
                        break
                self.streams.append(best[0])
                self.overlap_probe.append((attempt + 1, round(best[1], 2)))
        self._i = 0
        self.sync_with_current()

    def sync_with_current(self) -> None:
        """Work queued on the caller's stream so far (weights, resident batches) is visible to every compute stream."""
        cur = torch.cuda.current_stream(self.device)
        for s_ in self.streams:
            if s_ is not None:
                s_.wait_stream(cur)

    def sharing(self) -> bool:
        """Are consecutive steps of this dealer really dealt to different streams (and must therefore launch nothing foreign)?"""
        return len(self.streams) > 1 and self.stopped_sharing is None and not self.allow_foreign

    def stop_sharing(self, reason: str) -> None:
        """From now on every step goes to the FIRST stream (one step at a time on the device): a step launched kernels that are
        not this library's.  Loud: a RuntimeWarning naming the launch."""
        import warnings

        if self.stopped_sharing is None:
            self.stopped_sharing = reason
            warnings.warn("StepStreams: two steps in flight switched OFF for this dealer — " + reason + ".  Foreign arithmetic kernels "
                          "must not run beside another stream's MFMAs on MI355X (packed-fp32 hazard); steps now queue on one stream.",
                          RuntimeWarning, stacklevel=3)

    def shared_min_tiles(self) -> int:
        """Tile count from which a launch takes the three-product 256-row form while this dealer's streams share the chip: a
        launch need not give every CU a workgroup when a second step runs beside it (hip_lib.split2_tiles_ok; 0 = rule off).
        Unchanged by ``stop_sharing``: the kernel choice (and with it every bit of the records) stays what it was."""
        n = len(self.streams)
        return hip_lib.SPLIT2_MIN_TILES // n if n > 1 else 0

    def shared_min_rows(self):
        """Fewest rows of a launch under that rule: the process default (4 096) for two streams — the eager schedule of rounds 5 / 6,
        whose records this keeps — and 2 048 from three streams on (four hipGraphs in flight: 8 / 16 / 32 ROIs 3 305 -> 3 592,
        4 533 -> 4 561, 5 107 -> 5 172 ROIs/s with 64 tiles, profiles/r06_graph_streams.md)."""
        return 2048 if len(self.streams) > 2 else None

    def next(self):
        """Context manager: the body's launches go to the next compute stream (with n = 1: the caller's current stream) and, with
        n > 1, choose their GEMM kernels for a shared chip (``shared_min_tiles``: 4 248 -> 4 525 ROIs/s at 32 ROIs, neutral at 8 and
        128, profiles/r05y_shared_chip_tile_rule.txt)."""
        idx = self._i % len(self.streams)
        self._i += 1
        return self.on(idx)

    def on(self, index: int):
        """Context manager like ``next()`` for a GIVEN stream of the dealer, without advancing the round-robin (a hipGraph slot is
        bound to the stream it was captured on)."""
        import contextlib

        s_ = self.streams[0 if self.stopped_sharing is not None else index % len(self.streams)]

        @contextlib.contextmanager
        def ctx():
            # the rule belongs to the calling HOST THREAD for the duration of this step's launches (hip_lib.shared_min_tiles_scope):
            # two threads with their own dealers do not see each other's; an explicit setting (tests, A/B runs, env var) wins
            explicit = hip_lib.shared_min_tiles() != 0
            rule, rows = (None, None) if explicit else (self.shared_min_tiles(), self.shared_min_rows())
            prev = getattr(_DEALER_TLS, "dealer", None), getattr(_DEALER_TLS, "foreign_at_entry", None)
            _DEALER_TLS.dealer, _DEALER_TLS.foreign_at_entry = self, hip_layers.fallback_launches()
            try:
                with hip_lib.shared_min_tiles_scope(rule, rows), torch.cuda.stream(s_):    # torch.cuda.stream(None) is a no-op context
                    yield s_
            finally:
                _DEALER_TLS.dealer, _DEALER_TLS.foreign_at_entry = prev
        return ctx()


PAD_ROI_ID = -1.0       # roi_id column of gather_records' padding rows


def gather_records(rec: torch.Tensor, n_local_max: int, group=None, dst: int | None = None, single_rank_collective: bool = False):
    """The one collective of the inference path (gdrn_evaluator.py:575-585 / my_comm.py:70-171): instead of
    pickling Python dicts into byte tensors (size all-gather + padded byte all-gather), every rank contributes a
    fixed-shape f32[n_local_max,16] block (``valid`` = 0 on padding rows) to ONE all_gather — 64 B per ROI,
    latency-bound on xGMI.  Returns f32[world*n_local_max,16] on every rank.

    Padding rows carry ``roi_id`` = ``PAD_ROI_ID`` (-1) and ``valid`` = 0.  ``dst`` is a rank of ``group`` (group-local).

    ``single_rank_collective``: run the collective even in a one-rank group (``bench.py --force-dist``: what a 1-GPU box can show
    of the path).  ``dst``: gather to that rank only (``my_comm.gather``, my_comm.py:119-171; the reference's ``evaluate`` lets only the main
    process go on to write the results, gdrn_evaluator.py:581-582): rank ``dst`` gets the block, every other rank ``None``."""
    if rec.shape[0] < n_local_max:
        pad = torch.zeros((n_local_max - rec.shape[0], 16), dtype=rec.dtype, device=rec.device)
        pad[:, 14] = PAD_ROI_ID                  # padding says so itself: no real ROI has a negative id
        rec = torch.cat([rec, pad], 0)
    if not (dist.is_available() and dist.is_initialized()):
        return rec
    world = dist.get_world_size(group)
    if world == 1 and not single_rank_collective:     # bench.py --force-dist sends a one-rank group's records through the collective
        return rec
    rec = rec.contiguous()
    if dst is not None:                                    # dst = a rank OF ``group`` (group-local, like every index of this function)
        mine = dist.get_rank(group) == dst
        parts = [torch.empty_like(rec) for _ in range(world)] if mine else None
        dst_global = dist.get_global_rank(group, dst) if group is not None else dst      # dist.gather's dst is a GLOBAL rank
        dist.gather(rec, parts, dst=dst_global, group=group)      # RCCL: world - 1 point-to-point receives on rank dst
        return torch.cat(parts, 0) if mine else None
    if dist.get_backend(group) == "gloo":  # CPU tests: list form
        parts = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(parts, rec, group=group)
        return torch.cat(parts, 0)
    out = torch.empty((world * n_local_max, 16), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec, group=group)  # RCCL ncclAllGather over xGMI
    return out


def records_to_bop(rec: torch.Tensor, scene_im_ids, obj_ids, times=None):
    """BOP result dicts as ``pose_prediction_to_json`` writes them (gdrn_evaluator.py:636-665): R flattened row-major
    (``to_list(rot)``), t in mm."""
    rec = rec.detach().cpu()
    results = []
    for r in rec:
        if r[15] < 0.5:
            continue
        i = int(r[14])
        scene_id, im_id = scene_im_ids[i].split("/")
        results.append({
            "scene_id": scene_id, "im_id": int(im_id), "obj_id": int(obj_ids[int(r[13])]), "score": float(r[12]),
            "R": r[:9].tolist(), "t": (1000.0 * r[9:12]).tolist(),
            "time": float(times[i]) if times is not None else -1.0,
        })
    return results


def xyz_back_projection(depth: torch.Tensor, ego_rot: torch.Tensor, trans: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """``calc_xyz_bp_batch(..., fmt="BHWC")`` (lib/pysixd/misc.py:412-448): rendered depth f32[b,h,w] -> object-space points
    f32[b,h,w,3] = R^T ((x - cx) z / fx, (y - cy) z / fy, z) - t), zero where the depth is zero; integer pixel coordinates like the
    reference.  Plain tensor arithmetic on whatever device the depth lives on; pinned by tests/golden/xyz_bp_golden.npz (the
    reference's function executed from its source)."""
    bs, h, w = depth.shape
    dev = depth.device
    gy, gx = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    X = gx.expand(bs, h, w) - K[:, 0, 2].view(bs, 1, 1)
    Y = gy.expand(bs, h, w) - K[:, 1, 2].view(bs, 1, 1)
    cam = torch.stack((X * depth / K[:, 0, 0].view(bs, 1, 1), Y * depth / K[:, 1, 1].view(bs, 1, 1), depth), dim=-1)
    mask = (depth != 0).to(depth).unsqueeze(-1)
    return torch.einsum("bij,bhwj->bhwi", ego_rot.transpose(1, 2), cam - trans.view(bs, 1, 1, 3)) * mask


def render_roi_xyz_batch(meshes: hip_lib.MeshSet, roi_cls, ego_rot, trans, roi_zoom_K, out_res: int = 64, xyz_bp: bool = False,
                         z_near: float = 0.25, z_far: float = 6.0):
    """Online XYZ targets of the training-side ``batch_data`` (engine_utils.py:131-172) in ONE launch instead of a Python
    loop of GL renders + CUDA-GL copies: object-space surface points per ROI pixel (``pc_obj_tensor[:, :, :3]``) or, with
    ``xyz_bp`` (``XYZ_BP``), the rendered depth back-projected through ``calc_xyz_bp_batch`` (lib/pysixd/misc.py:412-448;
    integer pixel coordinates like the reference).  Returns (roi_xyz f32[bs,res,res,3], roi_mask_obj f32[bs,res,res]):
    the mask is the reference's "all three coordinates non-zero" test.  z_near / z_far default to the EGL renderer's."""
    bs = ego_rot.shape[0]
    dev = ego_rot.device
    out = hip_lib.render_depth(meshes, roi_cls.to(torch.int32).contiguous(), roi_zoom_K.reshape(bs, 3, 3).contiguous().float(),
                               ego_rot.contiguous().float(), trans.contiguous().float(), out_res, z_near, z_far,
                               want_xyz=not xyz_bp)
    depth, xyz = (out, None) if xyz_bp else out
    roi_xyz = xyz_back_projection(depth, ego_rot.float(), trans.float(), roi_zoom_K.reshape(bs, 3, 3).float()) if xyz_bp else xyz
    roi_mask_obj = ((roi_xyz[..., 0] != 0) & (roi_xyz[..., 1] != 0) & (roi_xyz[..., 2] != 0)).to(torch.float32)
    return roi_xyz, roi_mask_obj


def upnp_weights_from_cov(covar) -> "np.ndarray":
    """Weights of ``GDRN_Evaluator.pose_from_upnp`` (gdrn_evaluator.py:612-629): W = inv(sqrtm(C)) per 2x2 keypoint
    covariance, returned as (w_xx, w_xy, w_yy) f32[pn,3]; degenerate covariances (C[0,0] < 1e-6 or NaN) get zero weight.
    The reference calls ``scipy.linalg.sqrtm``; a symmetric positive-definite 2x2 matrix has the closed form
    sqrtm(C) = (C + s I) / t with s = sqrt(det C), t = sqrt(trace C + 2 s) (Cayley-Hamilton), used here so that no per-keypoint
    SciPy call is needed.  tests/ pin it against scipy.linalg.sqrtm."""
    import numpy as np

    c = np.asarray(covar, np.float64).reshape(-1, 2, 2)
    bad = (c[:, 0, 0] < 1e-6) | np.isnan(c).any(axis=(1, 2))
    cs = np.where(bad[:, None, None], np.eye(2)[None], c)
    det = cs[:, 0, 0] * cs[:, 1, 1] - cs[:, 0, 1] * cs[:, 1, 0]
    s = np.sqrt(np.maximum(det, 0.0))
    t = np.sqrt(cs[:, 0, 0] + cs[:, 1, 1] + 2.0 * s)
    root = (cs + s[:, None, None] * np.eye(2)[None]) / t[:, None, None]
    inv = np.linalg.inv(root)
    inv[bad] = 0.0
    # the reference stacks float32 zeros with float64 inverses and keeps columns (0, 1, 3) of the flattened 2x2
    return inv.reshape(-1, 4)[:, (0, 1, 3)]


def pose_from_upnp(mean_pts2d, covar, points_3d, K, init_rt=None):
    """``GDRN_Evaluator.pose_from_upnp`` (gdrn_evaluator.py:612-629): covariance -> weights -> uncertainty-PnP (HIP, fp64)."""
    from ..core.csrc.uncertainty_pnp.un_pnp_utils import uncertainty_pnp

    return uncertainty_pnp(mean_pts2d, upnp_weights_from_cov(covar), points_3d, K, init_rt=init_rt)


def mask_rles(cfg, batch: dict, out_dict: dict, key: str = "mask", compressed: bool = True) -> list:
    """SAVE_RESULTS_ONLY instance masks (gdrn_evaluator.py:914-945): ``get_out_mask`` (engine_utils.py:315-333) on the raw
    ``out_dict[key]`` maps, boxes = roi_center -/+ scale/2, then paste + threshold + COCO RLE fused on the device
    (``gdrnpp_paste_masks_rle``).  Returns one ``{"counts", "size"}`` dict per ROI like ``binary_mask_to_rle``."""
    from ..lib.utils.mask_utils import rle_from_counts

    net_cfg = cfg.MODEL.POSE_NET
    raw = out_dict[key]
    loss_type = net_cfg.LOSS_CFG.MASK_LOSS_TYPE
    bs = raw.shape[0]
    if loss_type == "L1":                     # per-ROI (m - min) / (max - min), no epsilon (reference behaviour)
        flat = raw.reshape(bs, -1)
        mn, mx = flat.min(1).values.view(bs, 1, 1, 1), flat.max(1).values.view(bs, 1, 1, 1)
        prob = (raw - mn) / (mx - mn)
    elif loss_type in ("BCE", "RW_BCE", "dice"):
        prob = torch.sigmoid(raw)
    elif loss_type == "CE":
        prob = torch.argmax(raw, dim=1, keepdim=True).to(torch.float32)
    else:
        raise NotImplementedError(f"unknown mask loss type: {loss_type}")
    scale = batch["scale"].view(bs, 1).to(torch.float32)
    boxes = torch.cat([batch["roi_center"] - scale / 2, batch["roi_center"] + scale / 2], 1).contiguous()
    im_h, im_w = int(batch["im_H"][0]), int(batch["im_W"][0])
    counts = hip_lib.paste_masks_rle(prob[:, 0].contiguous(), boxes, im_h, im_w, float(net_cfg.GEO_HEAD.MASK_THR_TEST))
    return [rle_from_counts(c, im_h, im_w, compressed) for c in counts]


BOP_CSV_HEADER = "scene_id,im_id,obj_id,score,R,t,time"


def save_bop_csv(results, path: str) -> None:
    """The BOP results file of ``save_and_eval_results`` (core/gdrn_modeling/engine/test_utils.py:33-52): one header line,
    then one line per estimate with R (9 values) and t (3 values, mm) space-separated inside their comma fields, every
    value formatted with ``"{}".format`` like the reference's ``_to_str``."""
    keys = BOP_CSV_HEADER.split(",")

    def to_str(item):
        return " ".join("{}".format(e) for e in item) if isinstance(item, (list, tuple)) else "{}".format(item)

    with open(path, "w") as f:
        f.write(BOP_CSV_HEADER + "\n")
        for res in results:
            f.write(",".join(to_str(res[k]) for k in keys) + "\n")


class GraphHandle:
    """A replayed hipGraph whose range words have not been looked at yet (``GraphedInference.replay_async``).  ``result()``
    waits for the replay (one event), and — when a three-product kernel of the graph left its range — repeats the step eagerly
    with six products and has the graph captured again.  Must be resolved before the same graph is replayed again (the graph's
    static buffers are reused; ``replay_async`` resolves a forgotten handle itself)."""

    def __init__(self, owner, event, rec):
        self._owner, self._event, self._rec = owner, event, rec
        self.stream = owner.stream              # the stream the graph was replayed on (None: the caller's current one)
        self.reran = False                      # result() repeated the step eagerly with six products

    def result(self) -> torch.Tensor:
        if self._owner is not None:
            owner, self._owner = self._owner, None
            n0 = owner.reruns
            self._rec = owner._resolve(self._event, self._rec)
            self.reran = owner.reruns != n0
            self._event = None
        return self._rec


class GraphedInference:
    """The whole hot path (forward + HIP post-processing + record packing) captured once into a hipGraph and
    replayed per batch.  At the reference's own batch sizes (one image = a few to ~30 ROIs, gdrn_evaluator.py:702)
    the ~150 launches of a step are launch-bound (3.2 ms of host time per 8-ROI step through ctypes); a graph replay removes the
    per-launch host cost.  Shapes are fixed at capture time: batches are copied into static device buffers (pad the ROI
    dimension to the captured size; padded rows are ordinary ROIs whose records the caller ignores).

    ``stream``: the HIP stream the graph is captured and replayed on (default: the caller's current stream).  Two graphs on the
    two streams of a ``StepStreams`` dealer are TWO STEPS IN FLIGHT without any per-launch host work (``GraphedStepStreams``).
    ``shared_min_tiles``: the kernel rule of a shared chip (StepStreams.shared_min_tiles) the graph's launches are chosen by —
    captured with the rule of the eager two-stream schedule a graph replays the same kernels and gives the same bits.
    ``sharing``: the graph is going to run beside another stream's step: a captured step that launched anything outside this
    library raises (hip_layers.fallback_launches; the packed-fp32 hazard of MI355X, profiles/r05p_two_stream_hazard.md).

    Three-product kernels inside the graph write their range words to a buffer the graph owns; every replay copies it to pinned
    host memory behind the graph (asynchronously — ``replay_async`` returns at once, ``GraphHandle.result()`` looks).  When a
    layer left the range the step is repeated eagerly with six products, its records are copied into the static output, the
    layer is demoted and the graph is captured again with it on the six-product kernels — so a flagged layer is paid for once,
    not on every replay."""

    def __init__(self, model, post: GdrnHipPost, example_batch: dict, roi_ids: torch.Tensor | None = None,
                 warmup: int = 3, stream=None, shared_min_tiles=None, sharing: bool = False, shared_min_rows=None, body=None):
        self.model, self.post = model, post
        self.body = body                                      # callable(static) -> records: a step that is more than forward + post
        self.reruns = 0                                       # (RoiStreamScheduler(graph_steps=True): crop + forward + post)
        self.stream = stream                                  # None = whatever stream is current when replay is called
        self.shared_min_tiles, self.shared_min_rows, self.sharing = shared_min_tiles, shared_min_rows, bool(sharing)
        self.static = ({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in example_batch.items()} if body is None
                       else example_batch)                    # with a body the caller owns the static inputs and fills them itself
        self.roi_ids = roi_ids.clone() if roi_ids is not None else None
        self.captures = 0
        self._pending = None                                  # the unresolved GraphHandle of the latest replay
        dev = next(v.device for v in self.static.values() if isinstance(v, torch.Tensor))
        for _ in range(max(warmup, 1)):                       # MIOpen find, hipFuncSetAttribute, allocator warm-up, weight packing and the
            self._eager_pass()                                # first range verdicts (demotions) happen outside capture
        self.x3_flag = torch.zeros((hip_lib.X3_SLOTS,), dtype=torch.int32, device=dev)   # the graph's own range words
        self._host_words = torch.zeros((hip_lib.X3_SLOTS,), dtype=torch.int32, pin_memory=True)
        self._capture()

    def _on_stream(self):
        """Context: the graph's stream is current (no-op when the graph follows the caller's stream)."""
        return torch.cuda.stream(self.stream)

    @torch.no_grad()
    def _eager_pass(self):
        """One eager step on a side stream with the CURRENT demotions / products, range check included (it may demote further
        layers): everything a capture must not do — packing a weight for the first time (a block that left the fused MLP kernel
        has never packed its two unfused images), ``packed_rows_in_range``'s host read, hipFuncSetAttribute — happens here."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        if self.stream is not None:
            side.wait_stream(self.stream)
        with torch.cuda.stream(side), hip_lib.shared_min_tiles_scope(self.shared_min_tiles, self.shared_min_rows):
            run_with_range_check(torch.no_grad()(self._step))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def _step(self):
        """The work of one replay, as a callable of no arguments (eager passes, capture, the six-product repeat)."""
        if self.body is not None:
            return self.body(self.static)
        return _step_closure(self.model, self.post, self.static, self.roi_ids)()

    def _recapture(self):
        for _ in range(4):          # an eager pass may itself demote a layer: repeat until the set is stable
            before = (hip_layers.x3_demoted(), hip_layers.gemm_products())
            self._eager_pass()
            if (hip_layers.x3_demoted(), hip_layers.gemm_products()) == before:
                break
        self._capture()

    @torch.no_grad()
    def _capture(self):
        run = self._step
        self.graph = torch.cuda.CUDAGraph()
        n_x3, n_foreign = hip_lib.x3_launch_count(), hip_layers.fallback_launches()
        self.x3_flag.zero_()
        torch.cuda.synchronize()
        with hip_lib.x3_flag_scope(self.x3_flag), hip_lib.shared_min_tiles_scope(self.shared_min_tiles, self.shared_min_rows), \
                torch.cuda.graph(self.graph, stream=self.stream):
            self.records = run()
        self.uses_x3 = hip_lib.x3_launch_count() != n_x3     # the captured step holds three-product kernels
        self.foreign_launches = hip_layers.fallback_launches() - n_foreign
        if self.sharing and self.foreign_launches:
            raise RuntimeError(f"GraphedInference(sharing=True): the captured step launched {self.foreign_launches} kernel(s) outside this "
                               f"library (last: {hip_layers.last_fallback()}); such a graph must not run beside another stream's MFMAs on "
                               "MI355X — replay it on ONE stream (sharing=False)")
        self._demoted_at_capture = hip_layers.x3_demoted()
        self._products_at_capture = hip_layers.gemm_products()
        self.captures += 1

    @torch.no_grad()
    def replay_async(self) -> GraphHandle:
        """Replay on the static buffers and return without waiting for the device; ``.result()`` -> the records f32[b,16] — a copy
        of the graph's static output made on the graph's stream right behind the replay (64 B per ROI), so the next replay of this
        graph cannot overwrite what the caller still reads on another stream."""
        if self._pending is not None:
            self._pending.result()
        if self.uses_x3 and (hip_layers.x3_demoted() != self._demoted_at_capture or hip_layers.gemm_products() != self._products_at_capture):
            self._recapture()      # another step demoted a layer this graph still runs on three products
        with self._on_stream():
            self.graph.replay()
            if self.uses_x3:
                self._host_words.copy_(self.x3_flag, non_blocking=True)
            rec = self.records.clone()
            ev = torch.cuda.Event()
            ev.record()
        self._pending = GraphHandle(self, ev, rec)
        return self._pending

    @torch.no_grad()
    def _resolve(self, event, rec) -> torch.Tensor:
        self._pending = None
        caller = torch.cuda.current_stream()
        if self.uses_x3:
            event.synchronize()
            words = hip_lib.range_words_of(self._host_words)
            if words:
                with self._on_stream(), hip_lib.shared_min_tiles_scope(self.shared_min_tiles, self.shared_min_rows):
                    rec = _six_product_rerun(self._step, words)
                    self.reruns += 1
                    self._recapture()
                    self.records.copy_(rec)
                    event = torch.cuda.Event()
                    event.record()
        if self.stream is not None and self.stream != caller:
            caller.wait_event(event)            # the caller's stream reads the records behind the replay ...
            rec.record_stream(caller)           # ... and their memory (the graph stream's pool) is not handed out under it
        return rec

    @torch.no_grad()
    def replay(self) -> torch.Tensor:
        """Replay on the static buffers, then the range check of the graph's three-product kernels (synchronous form)."""
        return self.replay_async().result()

    @torch.no_grad()
    def load(self, batch: dict) -> None:
        """Copy a batch into the graph's static buffers on the graph's stream (behind the previous replay, which is resolved
        first: its inputs must not change under it)."""
        if self._pending is not None:
            self._pending.result()
        caller = torch.cuda.current_stream()
        with self._on_stream():
            if self.stream is not None and self.stream != caller:
                self.stream.wait_stream(caller)       # the batch was produced on the caller's stream
            for k, v in batch.items():
                if isinstance(v, torch.Tensor) and k in self.static:
                    self.static[k].copy_(v, non_blocking=True)
            if self.roi_ids is not None and isinstance(batch.get("roi_id"), torch.Tensor):
                self.roi_ids.copy_(batch["roi_id"], non_blocking=True)      # the ids the records carry travel with the batch

    @torch.no_grad()
    def __call__(self, batch: dict) -> torch.Tensor:
        self.load(batch)
        return self.replay()


class GraphedStepStreams:
    """Two hipGraphs in flight: the small-batch form of ``StepStreams``.  One ``GraphedInference`` per SLOT (a resident batch, or a
    static buffer batches are copied into), slots dealt round-robin to the dealer's compute streams and captured there with the
    dealer's shared-chip kernel rule — the kernels, and therefore every bit of the records, are those of the eager two-stream
    schedule; what disappears is the host's ~150 launches per step, which is what bounds 8-32 ROIs (the reference's own regime:
    one image per forward, gdrn_evaluator.py:697-750, demo/predictor_gdrn.py:133-143).

        gs = GraphedStepStreams(model, post, [batch0, batch1, batch2, batch3])   # default_graph_streams(model) = 4 streams
        h0 = gs.launch(0); h1 = gs.launch(1); ...                       # four steps in flight, ~0.1 ms of host time each
        rec0 = h0.result(); h2 = gs.launch(0, new_batch) ...            # (launching a slot again resolves its previous handle first)

    A model whose step launches kernels outside this library gets ONE stream (the static gate), and a capture that does so all
    the same raises (``GraphedInference(sharing=True)``)."""

    def __init__(self, model, post: GdrnHipPost, slot_batches, roi_ids=None, compute_streams=None, warmup: int = 2, device=None):
        slot_batches = list(slot_batches)
        if not slot_batches:
            raise ValueError("GraphedStepStreams needs at least one slot batch")
        dev = slot_batches[0]["roi_img"].device if device is None else torch.device(device)
        if isinstance(compute_streams, StepStreams):
            self.dealer = compute_streams
        else:
            n = default_graph_streams(model) if compute_streams is None else max(1, int(compute_streams))
            self.dealer = StepStreams(n, dev)
        streams = self.dealer.streams
        multi = len(streams) > 1
        rule, rows = ((self.dealer.shared_min_tiles(), self.dealer.shared_min_rows()) if multi and hip_lib.shared_min_tiles() == 0
                      else (None, None))
        ids = roi_ids if isinstance(roi_ids, (list, tuple)) else [roi_ids] * len(slot_batches)
        self.graphs = []
        for i, (bt, rid) in enumerate(zip(slot_batches, ids)):
            st = streams[i % len(streams)]      # None (StepStreams(1)): captured on a side stream, replayed on the caller's current one
            self.graphs.append(GraphedInference(model, post, bt, rid if rid is not None else bt.get("roi_id"), warmup=warmup, stream=st,
                                                shared_min_tiles=rule, shared_min_rows=rows, sharing=self.dealer.sharing()))

    def launch(self, slot: int, batch: dict | None = None) -> GraphHandle:
        g = self.graphs[slot % len(self.graphs)]
        if batch is not None:
            g.load(batch)
        return g.replay_async()


# --------------------------------------------------------------------------------------------------
# ROI preparation on the device (rows a1 + a2): detections -> ROI tensors, no CPU crop, no H2D of crops
# --------------------------------------------------------------------------------------------------
def rois_from_detections(bboxes_xyxy, im_H: int, im_W: int, dzi_pad_scale: float = 1.5, out_res: int = 64):
    """Per-detection ROI parameters exactly as read_data_test derives them (data_loader.py:754-769), float64 like
    the reference's NumPy/Python scalars: centre, (bw, bh) clamped to >= 1, scale = min(max(bw,bh)*DZI_PAD_SCALE,
    max(im_H, im_W)), resize_ratio = out_res / scale."""
    import numpy as np

    bb = np.asarray(bboxes_xyxy, np.float64).reshape(-1, 4)
    x1, y1, x2, y2 = bb[:, 0], bb[:, 1], bb[:, 2], bb[:, 3]
    center = np.stack([0.5 * (x1 + x2), 0.5 * (y1 + y2)], 1)
    bw = np.maximum(x2 - x1, 1)
    bh = np.maximum(y2 - y1, 1)
    scale = np.minimum(np.maximum(bh, bw) * dzi_pad_scale, max(im_H, im_W)) * 1.0
    return dict(bbox_center=center, scale=scale, roi_wh=np.stack([bw, bh], 1).astype(np.float32),
                resize_ratio=(out_res / scale))


def detections_from_yolox(dets: torch.Tensor, count: torch.Tensor, cam, extents, ratio: float = 1.0, max_per_image: int = 0) -> dict:
    """Hand-off from the detector to the pose path without the JSON file of the reference (dataset_utils.py:146-239):
    ``hip_lib.yolox_postprocess`` output (dets f32[B,max_det,7], count i32[B]) -> the ``detections`` dict of
    ``batch_data_test_gpu``.  Boxes are divided by ``ratio`` (YOLOX's test-time resize, predictor_yolo.py:170-176), the
    score is obj_conf * class_conf and the class column becomes ``roi_cls``; rows keep NMS order within an image."""
    import numpy as np

    counts = count.tolist()
    rows, im_idx = [], []
    for i, n in enumerate(counts):
        n = min(n, dets.shape[1], max_per_image or n)
        if n > 0:
            rows.append(dets[i, :n])
            im_idx += [i] * n
    if not rows:
        return dict(bbox=np.zeros((0, 4), np.float32), im_idx=np.zeros((0,), np.int64), roi_cls=np.zeros((0,), np.int64),
                    score=np.zeros((0,), np.float32), cam=cam, extents=extents)
    d = torch.cat(rows, 0).cpu().numpy()
    return dict(bbox=d[:, :4] / np.float32(ratio), im_idx=np.asarray(im_idx, np.int64), roi_cls=d[:, 6].astype(np.int64),
                score=d[:, 4] * d[:, 5], cam=cam, extents=extents)


def detections_from_bop_json(detections: dict, scene_im_ids, obj_ids, cam, extents, top_k_per_obj: int = 1,
                             score_thr: float = 0.0, train_obj_ids=None) -> dict:
    """The offline hand-off of the reference: a BOP detection file ``{scene_im_id: [{"obj_id", "bbox_est": [x, y, w, h],
    "score", "time"}]}`` -> the ``detections`` dict of ``batch_data_test_gpu``, with the selection rules of
    ``load_detections_into_dataset`` (core/utils/dataset_utils.py:146-227): drop score < score_thr and objects the model
    was not trained on, keep the ``top_k_per_obj`` highest scores per object (stable for ties), objects in the dataset's
    class order, images without detections skipped.  ``scene_im_ids[i]`` names image ``i`` of the batch; ``obj_ids`` is
    the dataset's object-id list in class order.  Also returns ``time`` (detector time per ROI) for the BOP results."""
    import numpy as np

    obj_ids = [int(o) for o in obj_ids]
    keep = set(obj_ids if train_obj_ids is None else [int(o) for o in train_obj_ids])
    bbox, im_idx, cls, score, times = [], [], [], [], []
    for i, key in enumerate(scene_im_ids):
        per_obj = {o: [] for o in obj_ids}
        for det in detections.get(key, []):
            o, sc = int(det["obj_id"]), float(det.get("score", 1.0))
            if sc < score_thr or o not in per_obj or o not in keep:
                continue
            per_obj[o].append((sc, det))
        for o in obj_ids:
            for sc, det in sorted(per_obj[o], key=lambda pair: pair[0], reverse=True)[:top_k_per_obj]:
                x, y, w, h = [float(v) for v in det["bbox_est"]]
                bbox.append([x, y, x + w, y + h])           # BoxMode.XYWH_ABS -> XYXY_ABS
                im_idx.append(i)
                cls.append(obj_ids.index(o))
                score.append(sc)
                times.append(float(det.get("time", 0.0)))
    return dict(bbox=np.asarray(bbox, np.float32).reshape(-1, 4), im_idx=np.asarray(im_idx, np.int64),
                roi_cls=np.asarray(cls, np.int64), score=np.asarray(score, np.float32), cam=cam, extents=extents,
                time=np.asarray(times, np.float32))


def packed_layout(arrays: dict):
    """Byte layout of ``upload_packed``'s staging buffer: -> ({key: (offset, nbytes, numpy dtype, shape)}, total bytes); every
    array starts on a 16-byte boundary, dict order."""
    import numpy as np

    lay, total = {}, 0
    for k, a in arrays.items():
        a = np.asarray(a)
        total = (total + 15) & ~15
        lay[k] = (total, a.nbytes, a.dtype, a.shape)
        total += a.nbytes
    return lay, max(total, 16)


def fill_packed(host_u8, arrays: dict, layout: dict) -> None:
    """Write ``arrays`` into a staging buffer (a uint8 NumPy view) laid out by ``packed_layout``."""
    import numpy as np

    for k, (off, nbytes, dt, shape) in layout.items():
        a = np.ascontiguousarray(arrays[k], dtype=dt)
        if a.shape != tuple(shape):
            raise ValueError(f"fill_packed: {k!r} has shape {a.shape}, the layout holds {tuple(shape)}")
        if nbytes:
            host_u8[off:off + nbytes] = a.reshape(-1).view(np.uint8)


def packed_views(dev_u8: torch.Tensor, layout: dict) -> dict:
    """Typed tensor views of a device copy of the staging buffer."""
    import numpy as np

    out = {}
    for k, (off, nbytes, dt, shape) in layout.items():
        tdt = torch.from_numpy(np.empty((0,), dt)).dtype
        out[k] = dev_u8[off:off + nbytes].view(tdt).reshape(tuple(shape))
    return out


def upload_packed(arrays: dict, dev) -> dict:
    """The small per-ROI host arrays of a step -> device tensors through ONE pinned staging buffer and ONE asynchronous copy on the
    current stream.  A ``torch.from_numpy(a).to(dev)`` per array is a blocking pageable copy queued behind everything already on
    the stream: the host would sit out the step that is still running there before it could prepare the next one."""
    dev = torch.device(dev)
    layout, total = packed_layout(arrays)
    host = torch.empty((total,), dtype=torch.uint8, pin_memory=dev.type == "cuda")
    fill_packed(host.numpy(), arrays, layout)
    return packed_views(host.to(dev, non_blocking=True), layout)


def roi_host_arrays(cfg, detections: dict, H: int, W: int, sort_by_class: bool = False, roi_id_base: int = 0, extra_per_roi_keys=(),
                    extra_global_keys=()) -> dict:
    """The HOST half of ``batch_data_test_gpu``: detections -> the per-ROI NumPy arrays of a step (ROI parameters exactly as
    read_data_test derives them, data_loader.py:754-769; class sort; ids), in the order ``upload_packed`` lays them out."""
    import numpy as np

    roi_id = None
    if sort_by_class:
        detections, roi_id = sort_detections_by_class(detections, roi_id_base, extra_per_roi_keys, extra_global_keys)
    if "roi_id" in detections:        # the caller's own ids (RoiStreamScheduler: global stream ids), permuted with the rest
        roi_id = np.asarray(detections["roi_id"], np.int32)
    r = rois_from_detections(detections["bbox"], H, W, cfg.INPUT.DZI_PAD_SCALE, cfg.MODEL.POSE_NET.OUTPUT_RES)
    n = len(r["scale"])
    cls = np.asarray(detections["roi_cls"], np.int64)
    cam = np.asarray(detections["cam"], np.float32)
    cam = np.repeat(cam[None], n, 0) if cam.ndim == 2 else cam
    host = dict(center64=r["bbox_center"], scale64=r["scale"], im_idx=np.asarray(detections["im_idx"], np.int32), roi_cls=cls, roi_cam=cam,
                roi_center=np.asarray(r["bbox_center"], np.float32), roi_wh=r["roi_wh"], scale=np.asarray(r["scale"], np.float32),
                resize_ratio=np.asarray(r["resize_ratio"], np.float32), roi_extent=np.asarray(detections["extents"], np.float32)[cls],
                score=np.asarray(detections.get("score", np.ones(n)), np.float32))
    if roi_id is not None:
        host["roi_id"] = np.asarray(roi_id, np.int32)
    return host


def batch_from_uploaded(cfg, images: torch.Tensor, depths, up: dict, dev=None) -> dict:
    """The DEVICE half: the uploaded per-ROI arrays (``upload_packed`` / ``packed_views`` of ``roi_host_arrays``) + the images ->
    GPU crops (``gdrnpp_crop_resize_roi``) and the batch dict ``GDRN_Net.forward`` / ``GdrnHipPost`` consume.  Launches and tensor
    views only — no host data: this half can be captured into a hipGraph (``RoiStreamScheduler(graph_steps=True)``)."""
    dev = dev or images.device
    net_cfg = cfg.MODEL.POSE_NET
    n_im, H, W, _ = images.shape
    n = up["scale"].shape[0]
    centers64, scales64 = up["center64"], up["scale64"]
    roi_img, roi_depth, roi_c2d = hip_lib.crop_resize_roi(
        images, depths, up["im_idx"], centers64, scales64,
        out_res=net_cfg.INPUT_RES, out_res_small=net_cfg.OUTPUT_RES, pixel_mean=cfg.MODEL.PIXEL_MEAN,
        pixel_std=cfg.MODEL.PIXEL_STD)
    batch = dict(
        roi_img=roi_img, roi_coord_2d=roi_c2d, roi_cls=up["roi_cls"], roi_cam=up["roi_cam"], roi_center=up["roi_center"],
        roi_wh=up["roi_wh"], scale=up["scale"], resize_ratio=up["resize_ratio"], roi_extent=up["roi_extent"], score=up["score"],
        im_H=torch.full((n,), float(H), device=dev), im_W=torch.full((n,), float(W), device=dev))
    if roi_depth is not None:
        batch["roi_depth"] = roi_depth
    if "roi_id" in up:
        batch["roi_id"] = up["roi_id"]
    if net_cfg.PNP_NET.COORD_2D_TYPE == "rel":
        # data_loader.py:799-804: (bbox_center - roi_coord_2d * (im_W, im_H)) / scale, float64 like NumPy, stored float32
        hip_layers.note_foreign_launch("batch_data_test_gpu: COORD_2D_TYPE='rel' computed with torch operators")
        wh = torch.tensor([float(W), float(H)], dtype=torch.float64, device=dev).view(1, 2, 1, 1)
        batch["roi_coord_2d_rel"] = ((centers64.view(n, 2, 1, 1) - roi_c2d.double() * wh) / scales64.view(n, 1, 1, 1)).float()
    return batch


def batch_data_test_gpu(cfg, images: torch.Tensor, depths, detections: dict, device=None, sort_by_class: bool = False,
                        roi_id_base: int = 0, extra_per_roi_keys=(), extra_global_keys=()) -> dict:
    """``read_data_test`` + ``batch_data_test`` (data_loader.py:647-818, engine_utils.py:213-241) with the crops made
    on the GPU.  images u8[n_im,H,W,3] (BGR, device), depths f32[n_im,H,W] or None, detections:
    {"bbox": [n,4] xyxy, "im_idx": [n], "roi_cls": [n], "score": [n], "cam": [n,3,3] or [3,3], "extents": [C,3]}.
    Returns the batch dict ``GDRN_Net.forward`` / ``GdrnHipPost`` consume (all tensors on the device).
    ``sort_by_class``: ROIs are laid out in class order and ``batch["roi_id"]`` = ``roi_id_base`` + the
    detection's original position (or the caller's ``detections["roi_id"]``), which ``inference_step`` writes into the records
    (``records_in_roi_order`` restores it).  = ``roi_host_arrays`` -> ``upload_packed`` (one pinned buffer, one asynchronous copy:
    the host never waits for the stream) -> ``batch_from_uploaded``."""
    dev = device or images.device
    n_im, H, W, _ = images.shape
    host = roi_host_arrays(cfg, detections, H, W, sort_by_class, roi_id_base, extra_per_roi_keys, extra_global_keys)
    return batch_from_uploaded(cfg, images, depths, upload_packed(host, dev), dev)


# --------------------------------------------------------------------------------------------------
# ROI packing: the reference's image loop (one image per forward, gdrn_evaluator.py:702, data_loader.py:901 batch_size=1)
# feeds the network 3-30 ROIs at a time; the kernels of this library reach their rate from ~128 ROIs per step on.  The packer
# sits between the two: ROIs of consecutive images are dealt into steps of EXACTLY ``rois_per_step`` (an image's ROIs may
# straddle two steps), every ROI carries a stream-wide id into its record, and records are dealt back to their images.
# --------------------------------------------------------------------------------------------------
class RoiPacker:
    """Host-side bookkeeping of the packing (no device, no tensors): which ROI of which image goes into which step, and which
    images are complete once a step's records are back.  ROI ids wrap at 2^24 (they travel as float32 in the records)."""

    ID_WRAP = 1 << 24

    def __init__(self, rois_per_step: int, roi_id_base: int = 0):
        import collections

        if rois_per_step < 1:
            raise ValueError("rois_per_step must be positive")
        self.rois_per_step = int(rois_per_step)
        self._queue = collections.deque()      # [key, n, next local index] of images with ROIs not yet dealt into a step
        self._pending = 0
        self._next_id = int(roi_id_base) % self.ID_WRAP
        self._where = {}                       # roi id -> (key, local index) of ROIs dealt into a step whose records are not back
        self._open = {}                        # key -> [n, records f32[n,16], number still missing]
        self._done = []

    def add_image(self, key, n_rois: int) -> None:
        import numpy as np

        if key in self._open:
            raise KeyError(f"image key {key!r} is already in flight")
        n = int(n_rois)
        if n == 0:
            self._done.append((key, np.zeros((0, 16), np.float32)))      # the reference skips images without detections
            return
        self._open[key] = [n, np.full((n, 16), np.nan, np.float32), n]
        self._queue.append([key, n, 0])
        self._pending += n

    @property
    def pending(self) -> int:
        return self._pending

    def ready(self) -> bool:
        return self._pending >= self.rois_per_step

    def next_pack(self, flush: bool = False):
        """-> [(key, local indices i64[k], roi ids i32[k]), ...] covering exactly ``rois_per_step`` ROIs in arrival order (fewer
        only with ``flush`` = the tail of the stream), or None when there is nothing to launch yet."""
        import numpy as np

        if self._pending == 0 or (not flush and not self.ready()):
            return None
        want = min(self.rois_per_step, self._pending)
        pack = []
        while want > 0:
            ent = self._queue[0]
            key, n, nxt = ent
            k = min(want, n - nxt)
            local = np.arange(nxt, nxt + k, dtype=np.int64)
            ids = ((self._next_id + np.arange(k, dtype=np.int64)) % self.ID_WRAP).astype(np.int32)
            for j, i in zip(local.tolist(), ids.tolist()):
                self._where[i] = (key, j)
            self._next_id = (self._next_id + k) % self.ID_WRAP
            pack.append((key, local, ids))
            ent[2] += k
            if ent[2] == n:
                self._queue.popleft()
            want -= k
            self._pending -= k
        return pack

    def last_roi_dealt(self, key) -> bool:
        """True once every ROI of image ``key`` has been dealt into a step (its pixels are no longer needed)."""
        return all(e[0] != key for e in self._queue)

    def deliver(self, records) -> None:
        """Records f32[m,16] of one step (any order) -> their images.  A record is delivered when its id is one this packer dealt
        and is still waiting for — whatever its ``valid`` column says: the refine kernel marks a ROI whose object id lies outside
        the mesh set invalid, and that ROI's image must still complete (the row keeps valid = 0 for the consumer; a record that
        is zero in every column is still the record of ROI id 0).  The only rows skipped are ``gather_records``' padding
        (roi_id = PAD_ROI_ID < 0: marked, not guessed) and ids that are not in flight."""
        import numpy as np

        rec = np.asarray(records, np.float32).reshape(-1, 16)
        for r in rec:
            if not r[14] >= 0:                   # padding (or a NaN id): never a ROI of this stream
                continue
            rid = int(r[14])
            if rid not in self._where:
                continue
            key, j = self._where.pop(rid)
            ent = self._open[key]
            ent[1][j] = r
            ent[2] -= 1
            if ent[2] == 0:
                self._done.append((key, ent[1]))
                del self._open[key]

    def pop_completed(self):
        """-> [(key, records f32[n,16] in the image's own detection order), ...] of the images completed since the last call."""
        done, self._done = self._done, []
        return done


def h2d_overlap(copies, steps, detail: bool = False) -> dict:
    """Device timeline of host-to-device copies against compute: ``copies`` = (start, end) timing events on copy streams,
    ``steps`` = (start, end) timing events around the steps' kernels on the compute stream (of one or several schedulers feeding
    the same device); one clock (elapsed time from the first copy's start).  -> h2d_ms (summed copy durations), overlapped_ms
    (the part of them during which some step's kernels were executing) and overlapped_frac = overlapped_ms / h2d_ms."""
    torch.cuda.synchronize()
    out = {"h2d_ms": sum(a.elapsed_time(b) for a, b in copies), "overlapped_ms": 0.0, "overlapped_frac": None,
           "images": len(copies), "steps": len(steps)}
    if copies and steps:
        origin = copies[0][0]
        busy, merged = sorted((origin.elapsed_time(a), origin.elapsed_time(b)) for a, b in steps), []
        for s0, s1 in busy:                      # union of the step intervals
            if merged and s0 <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], s1)
            else:
                merged.append([s0, s1])
        for a, b in copies:
            c0, c1 = origin.elapsed_time(a), origin.elapsed_time(b)
            for s0, s1 in merged:
                lo, hi = max(c0, s0), min(c1, s1)
                if hi > lo:
                    out["overlapped_ms"] += hi - lo
        if out["h2d_ms"] > 0:
            out["overlapped_frac"] = min(1.0, out["overlapped_ms"] / out["h2d_ms"])
        if detail:
            out["steps_ms"] = busy
            out["copies_ms"] = [(origin.elapsed_time(a), origin.elapsed_time(b)) for a, b in copies]
    return out


class RoiStreamScheduler:
    """detections -> pose records for a STREAM of images, at the step size the kernels want.

        push(key, image u8[H,W,3] (device, BGR), depth f32[H,W] | None, detections)      one image and its detections
          -> the packer deals ROIs into steps of exactly ``rois_per_step``; every full step is launched at once:
             GPU crop (gdrnpp_crop_resize_roi, ROIs class-sorted within the step) -> inference_step_async
          -> steps are resolved one launch late (the range check of the three-product kernels and the 8 KB record copy never
             stall the device), their records dealt back to the images
          -> returns the images that became complete: [(key, records f32[n,16] in detection order, seconds since push)]
        flush()  launches the tail (a short step) and returns everything still open.

    ``detections`` = the dict of ``batch_data_test_gpu`` for ONE image (bbox [n,4] xyxy, roi_cls [n], score [n], cam [3,3],
    extents [C,3]).  All images of a stream share H x W (a BOP dataset's resolution).

    Host-fed streams (the reference's loader hands over HOST arrays, data_loader.py:754-797, and ``batch_data_test`` moves the
    ROI crops to the device, engine_utils.py:213-241): ``image`` / ``depth`` may be CPU tensors — pinned, or the copy is not
    asynchronous.  They are copied to the device on the scheduler's own copy stream the moment they are admitted (the FULL
    image once, 0.9 + 1.2 MB, not a 1 MB crop per ROI), an event per image orders the step's crop kernel behind its copies, and
    since admission runs one step ahead of the device the copies overlap the previous step's kernels.  ``time_h2d=True``
    brackets every image's copies with timing events (``h2d_ms()``).

    ``graph_steps``: every FULL step replays a captured hipGraph of (GPU crop -> forward -> post-processing) instead of ~150 eager
    launches — for small ``rois_per_step`` (the reference's own regime: a few ROIs at a time, low latency), where the host's launches
    bound the eager schedule; the graphs sit in 2 x streams slots with static image / per-ROI buffers, ``default_graph_streams``
    (4) steps in flight; records bit-equal to the eager scheduler with the same kernel rule; the tail step of ``flush`` runs eagerly.

    ``compute_streams`` (default: ``default_compute_streams(model)`` = 2 for the ConvNeXt configurations, whose every kernel is
    this library's): consecutive steps are launched on alternating HIP streams (``StepStreams``), so that with
    ``max_in_flight`` >= 2 two steps really are in flight on the device — one step's narrow tail under the next one's GEMMs —
    instead of queued behind each other; 1 = everything on the caller's current stream (rounds 1-4); a ``StepStreams`` object =
    that dealer, shared by several schedulers of one device (bench.py's seven-dataset stream)."""

    def __init__(self, cfg, model, post: GdrnHipPost, rois_per_step: int = 128, max_in_flight: int = 2, roi_id_base: int = 0,
                 device=None, time_h2d: bool = False, compute_streams=None, graph_steps: bool = False):
        import collections

        self.device = device
        self.graph_steps = bool(graph_steps)    # full steps replay a captured hipGraph (crop + forward + post): small rois_per_step
        self._slots = []                        # graph slots: static inputs + GraphedInference, bound to a compute stream each
        if compute_streams is None:
            compute_streams = default_graph_streams(model) if graph_steps else default_compute_streams(model)
        if isinstance(compute_streams, StepStreams):     # shared with other schedulers feeding the same device
            self._n_compute, self._dealer = len(compute_streams.streams), compute_streams
        else:
            self._n_compute = max(1, int(compute_streams))
            self._dealer = None                 # StepStreams, made at the first launch (the device is known then)
        self._copy_stream = None
        self._h2d_ready = {}                    # key -> event: the image's pixels are on the device
        self._time_h2d = bool(time_h2d)
        self._h2d_timing = []                   # (start, end) events per image, copy stream
        self._step_timing = []                  # (start, end) events per step, compute stream (time_h2d only)
        self.h2d_bytes = 0

        self.cfg, self.model, self.post = cfg, model, post
        self.packer = RoiPacker(rois_per_step, roi_id_base)
        self.max_in_flight = max(1, int(max_in_flight))
        if self.graph_steps:                    # a graph replay costs the host ~0.1 ms: as many steps in flight as there are streams
            self.max_in_flight = max(self.max_in_flight, self._n_compute)
        self._images = {}                       # key -> (image, depth, detections, arrival time)
        self._arrival = {}
        self._in_flight = collections.deque()   # (StepHandle, batch, done event) — the batch stays alive for a six-product repeat
        self._with_depth = None                 # fixed by the first image that has ROIs
        self._d2h_stream = None                 # side stream of the 8 KB record copies
        self.latencies = collections.deque(maxlen=1 << 16)   # seconds from push to completed records, per image (newest 65 536)
        self.steps_launched = 0

    # -- one step ----------------------------------------------------------------------------------
    def _launch(self, pack) -> None:
        import numpy as np

        def per_roi(key, loc, name, dtype, default=None):
            d = self._images[key][2]
            a = np.asarray(d[name] if name in d else default(len(d["roi_cls"])), dtype)
            return a.reshape((len(d["roi_cls"]),) + a.shape[1:])[loc]

        def cams(key, loc):
            c = np.asarray(self._images[key][2]["cam"], np.float32)
            return np.broadcast_to(c, (len(loc), 3, 3)) if c.ndim == 2 else c[loc]

        keys = [k for k, _, _ in pack]
        if self._dealer is None:
            dev = self.device if self.device is not None else self._images[keys[0]][0].device
            self._dealer = StepStreams(self._n_compute, dev)
        caller = torch.cuda.current_stream(self._dealer.device)
        n_rois = sum(len(loc) for _, loc, _ in pack)
        if self.graph_steps and n_rois == self.packer.rois_per_step:
            n_slots = 2 * self._n_compute       # twice the streams: consecutive steps alternate streams, a slot is reused only after
            k = self.steps_launched % n_slots   # max_in_flight (<= streams) younger steps were launched, i.e. after it was resolved
            with self._dealer.on(k):
                self._launch_graph_step(k, pack, keys, per_roi, cams, caller)
        else:                                   # eager (the default; in graph mode: the short tail step of flush())
            with self._dealer.next():           # this step's crop, forward and post-processing: the next compute stream
                self._launch_on_current_stream(pack, keys, per_roi, cams, caller)
        for k in keys:                          # pixels are only read by the crop kernel just enqueued
            if self.packer.last_roi_dealt(k):
                del self._images[k]
                self._h2d_ready.pop(k, None)

    def _step_detections(self, pack, keys, per_roi, cams) -> dict:
        import numpy as np

        return dict(
            bbox=np.concatenate([per_roi(k, loc, "bbox", np.float32) for k, loc, _ in pack]),
            roi_cls=np.concatenate([per_roi(k, loc, "roi_cls", np.int64) for k, loc, _ in pack]),
            score=np.concatenate([per_roi(k, loc, "score", np.float32, np.ones) for k, loc, _ in pack]),
            im_idx=np.concatenate([np.full(len(loc), i, np.int64) for i, (_, loc, _) in enumerate(pack)]),
            roi_id=np.concatenate([ids for _, _, ids in pack]),
            cam=np.concatenate([cams(k, loc) for k, loc, _ in pack]),
            extents=self._images[keys[0]][2]["extents"])

    def _launch_graph_step(self, k, pack, keys, per_roi, cams, caller) -> None:
        """A full step as a hipGraph replay: the step's images are copied into the slot's static image block, its per-ROI arrays
        through the slot's pinned buffer into the slot's packed device buffer (one asynchronous copy), then the slot's graph —
        GPU crop, forward, post-processing, records — is replayed on the slot's stream.  The first use of a slot captures it."""
        dev = self._dealer.device
        cur = torch.cuda.current_stream(dev)
        if cur != caller:
            cur.wait_stream(caller)
        for key in keys:
            ev = self._h2d_ready.get(key)
            if ev is not None:
                cur.wait_event(ev)
        im0, dp0 = self._images[keys[0]][0], self._images[keys[0]][1]
        H, W = int(im0.shape[0]), int(im0.shape[1])
        det = self._step_detections(pack, keys, per_roi, cams)
        host = roi_host_arrays(self.cfg, det, H, W, sort_by_class=True)
        while len(self._slots) <= k:
            self._slots.append(None)
        slot = self._slots[k]
        if slot is None:
            P = self.packer.rois_per_step       # a step of P ROIs touches at most P images
            layout, total = packed_layout(host)
            slot = dict(images=torch.zeros((P, H, W, 3), dtype=torch.uint8, device=dev),
                        depths=torch.zeros((P, H, W), dtype=torch.float32, device=dev) if self._with_depth else None,
                        packed=torch.zeros((total,), dtype=torch.uint8, device=dev),
                        pinned=torch.zeros((total,), dtype=torch.uint8, pin_memory=True), layout=layout, graph=None)
            self._slots[k] = slot
        if slot["graph"] is not None and slot["graph"]._pending is not None:
            slot["graph"]._pending.result()     # (cannot happen with max_in_flight <= streams; the pinned buffer must be free)
        for i, key in enumerate(keys):          # device-to-device copies on the slot's stream (the images were produced / copied elsewhere)
            im, dp = self._images[key][0], self._images[key][1]
            im.record_stream(cur)
            slot["images"][i].copy_(im, non_blocking=True)
            if slot["depths"] is not None:
                dp.record_stream(cur)
                slot["depths"][i].copy_(dp, non_blocking=True)
        fill_packed(slot["pinned"].numpy(), host, slot["layout"])
        slot["packed"].copy_(slot["pinned"], non_blocking=True)
        if self._time_h2d:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
        if slot["graph"] is None:
            cfg, model, post = self.cfg, self.model, self.post

            def body(static):
                up = packed_views(static["packed"], slot["layout"])
                batch = batch_from_uploaded(cfg, static["images"], static["depths"], up, dev)
                return _step_closure(model, post, batch, batch["roi_id"])()

            multi = len(self._dealer.streams) > 1
            rule, rows = ((self._dealer.shared_min_tiles(), self._dealer.shared_min_rows()) if multi and hip_lib.shared_min_tiles() == 0
                          else (None, None))
            slot["graph"] = GraphedInference(model, post, dict(images=slot["images"], depths=slot["depths"], packed=slot["packed"]), None,
                                             warmup=2, stream=cur, shared_min_tiles=rule, shared_min_rows=rows,
                                             sharing=self._dealer.sharing(), body=body)
        handle = slot["graph"].replay_async()
        done = torch.cuda.Event()
        done.record()
        self._in_flight.append((handle, None, done))
        if self._time_h2d:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self._step_timing.append((t0, t1))
        self.steps_launched += 1

    def _launch_on_current_stream(self, pack, keys, per_roi, cams, caller) -> None:
        import numpy as np

        cur = torch.cuda.current_stream(self._dealer.device)
        if cur != caller:
            cur.wait_stream(caller)             # device images handed over by the caller were produced on ITS stream
        for k in keys:                          # host-fed images: the crop kernel waits for their copies (device-side wait)
            ev = self._h2d_ready.get(k)
            if ev is not None:
                cur.wait_event(ev)
        for k in keys:                          # allocated on the caller's / the copy stream, read by this stream's crop kernel:
            for t in self._images[k][:2]:       # their memory must not be handed out again before that kernel has run
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
        images = torch.stack([self._images[k][0] for k in keys])
        depths = torch.stack([self._images[k][1] for k in keys]) if self._with_depth else None
        det = self._step_detections(pack, keys, per_roi, cams)
        if self._time_h2d:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
        batch = batch_data_test_gpu(self.cfg, images, depths, det, sort_by_class=True)
        handle = inference_step_async(self.model, self.post, batch)
        done = torch.cuda.Event()               # everything of this step, on the compute stream
        done.record()
        self._in_flight.append((handle, batch, done))
        if self._time_h2d:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self._step_timing.append((t0, t1))
        self.steps_launched += 1

    def _resolve_oldest(self):
        """Records of the OLDEST step in flight -> their images.  The 8 KB device-to-host copy runs on a side stream behind that
        step's own event: a ``rec.cpu()`` on the compute stream would queue behind the NEWER steps already launched there and
        stall the host until they finish — no step would ever be prepared while another runs (measured: every image copy of a
        host-fed stream landed in the idle gap between two steps, profiles/r05d_h2d_timeline_before_fix.txt)."""
        handle, _batch, done = self._in_flight.popleft()
        rec = handle.result()                   # waits for that step's range-word event only (and repeats a flagged step)
        if rec.is_cuda:
            if self._d2h_stream is None:
                # high priority = the other hardware-queue pool: a default-priority stream may share a queue with a compute stream, and
                # a record copy queued there behind the NEWEST step's kernels would hold the host until that step is done
                self._d2h_stream = torch.cuda.Stream(device=rec.device, priority=-1)
            compute = handle.stream if handle.stream is not None else torch.cuda.current_stream(rec.device)   # the step's own stream (looked up OUTSIDE the side stream's context)
            with torch.cuda.stream(self._d2h_stream):
                if handle.reran:                # a six-product repeat ran on the step's stream just now: its records are the newest work there
                    self._d2h_stream.wait_stream(compute)
                else:
                    self._d2h_stream.wait_event(done)
                host = rec.to("cpu", non_blocking=False)
            rec.record_stream(self._d2h_stream)
        else:
            host = rec
        self.packer.deliver(host.numpy())
        return rec

    def _finished(self):
        import time

        now = time.perf_counter()
        done = [(k, r, now - self._arrival.pop(k)) for k, r in self.packer.pop_completed()]
        self.latencies.extend(lat for _, r, lat in done if len(r))       # push -> records back on the host, images with ROIs
        return done

    def _admit(self, key, image, depth, detections) -> None:
        import time

        n = len(detections["roi_cls"])
        if n and self._with_depth is not None and (depth is not None) != self._with_depth:
            raise ValueError("RoiStreamScheduler: a stream is either with depth or without, not mixed "
                             f"(image {key!r} {'has' if depth is not None else 'lacks'} a depth map)")
        self.packer.add_image(key, n)           # raises for a key still in flight BEFORE any state of that image is touched
        if n:
            if self._with_depth is None:
                self._with_depth = depth is not None
            if isinstance(image, torch.Tensor) and image.device.type == "cpu":
                image, depth = self._to_device(key, image, depth)
            self._images[key] = (image, depth, detections)
        self._arrival[key] = time.perf_counter()

    def _to_device(self, key, image, depth):
        """Host image (+ depth) -> device on the copy stream; the event is what ``_launch`` waits for."""
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        if self._copy_stream is None:
            # high priority = a hardware queue of its own: a default-priority stream may be mapped onto the compute stream's queue
            # (HIP multiplexes its streams over a few hardware queues) and its copies would then wait for the step in front of them —
            # measured: 3.5 % of the copy time under compute with a default stream (profiles/r05b_bench_stream_hostfed.json)
            self._copy_stream = torch.cuda.Stream(device=dev, priority=-1)
        with torch.cuda.stream(self._copy_stream):
            if self._time_h2d:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record()
            image_d = image.to(dev, non_blocking=True)
            depth_d = depth.to(dev, non_blocking=True) if depth is not None else None
            ev = torch.cuda.Event(enable_timing=self._time_h2d)
            ev.record()
        self._h2d_ready[key] = ev               # (_launch marks the tensors as used by the compute stream that crops them)
        self.h2d_bytes += image.numel() * image.element_size() + (depth.numel() * depth.element_size() if depth is not None else 0)
        if self._time_h2d:
            self._h2d_timing.append((t0, ev))
        return image_d, depth_d

    def h2d_done_event(self, key):
        """The event behind the host-to-device copies of image ``key`` (a host-fed image admitted by ``push`` / ``launch_next``), or
        None once every ROI of the image has been dealt into a step (the copies are long done then) or for a device image.  The
        copies are asynchronous (``non_blocking``) reads of the caller's PINNED buffers: a caller that recycles those buffers must
        ``event.synchronize()`` (or make its producer stream wait for it) before overwriting them."""
        return self._h2d_ready.get(key)

    def h2d_ms(self, reset: bool = True) -> float:
        """Summed device-side duration of the host-to-device copies admitted so far (``time_h2d=True``), in ms."""
        return self.h2d_timeline(reset)["h2d_ms"]

    def h2d_timeline(self, reset: bool = True) -> dict:
        """Device timeline of this scheduler's copies against its steps (``time_h2d=True``), see ``h2d_overlap``."""
        out = h2d_overlap(self._h2d_timing, self._step_timing)
        if reset:
            self._h2d_timing, self._step_timing = [], []
        return out

    # -- the stream --------------------------------------------------------------------------------
    def push(self, key, image: torch.Tensor, depth, detections: dict):
        self._admit(key, image, depth, detections)
        while self.packer.ready():
            self._launch(self.packer.next_pack())
            while len(self._in_flight) > self.max_in_flight:
                self._resolve_oldest()
        return self._finished()

    def launch_next(self, feeder):
        """bench.py's step: pull (key, image, depth, detections) tuples from ``feeder`` until one more step is launched;
        returns a callable that resolves the OLDEST step in flight (-> its records f32[rois_per_step,16] on the device).  Call
        it one launch late and the host never waits for the device."""
        while not self.packer.ready():
            self._admit(*next(feeder))
        self._launch(self.packer.next_pack())

        def resolve():
            rec = self._resolve_oldest()
            self._finished()
            return rec
        return resolve

    def flush(self):
        while self.packer.pending:
            self._launch(self.packer.next_pack(flush=True))
        while self._in_flight:
            self._resolve_oldest()
        return self._finished()
