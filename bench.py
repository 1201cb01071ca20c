#!/usr/bin/env python
"""bench.py — ROIs/sec of the GDRNPP hot path on MI355X (BASELINE.json metric).

One "step" = one call of the product entry point ``engine.inference_step`` on one batch of synthetic ROIs already
resident in HBM (+ the one collective of the path when N > 1):
    GDRN_Net forward (ConvNeXt-B + geometry head + Patch-PnP, fp32)  ->  K_crop  ->  fast depth refine
    (render + compare, 2 iterations, HIP)  ->  pose records  ->  (N>1) one RCCL all-gather of the f32[n,16] records.
Two distinct batches per model alternate step by step.  Consecutive steps are independent and are dealt to ``--compute-streams``
HIP streams (default 2, engine.StepStreams): two steps are in flight on the device, one's narrow tail under the other's GEMMs;
records are bit-equal to the one-stream schedule (tests/test_gpu_streams2.py) and ``single_stream_mode`` reports that schedule
from the same process.  A step's records are resolved (range check of the three-product GEMM kernels, engine.StepHandle) after
the next ``compute_streams`` steps have been launched, all of them inside the timed region (K steps launched, K resolved).  Timing = the reference's protocol (gdrn_evaluator.py:697-706,
748-750): host perf_counter, device synchronised (and ranks barriered) on both sides, warm-up steps discarded, MAX over ranks.

Workloads (``--workload``; index into BASELINE.json ``configs``):
  refine (default, N < 8)  configs[2]  YCB-V convnext_a6 + fast depth refine, 128 ROIs per GPU — the metric's configuration
  tless  (default, N = 8)  configs[3]  T-LESS 30 objects, 1024 ROIs per iteration ROI-sharded (128 per rank) + refine + gather
  rgb                      configs[1]  YCB-V convnext_a6, 64 ROIs, RGB-only Patch-PnP
  lmo_upnp                 configs[0]  LM-O ape, 32 ROIs, ResNet-34 forward + uncertainty-PnP (9 keypoints per ROI, HIP LM)
  bop7                     configs[4]  BOP-7 mixed stream (lmo/ycbv/tless/icbin/hb/itodd/tudl models cycled per step) + refine
  stream                   (configs[2] fed the reference's way) a stream of 480x640 images with 3-30 detections each, packed by
                           engine.RoiStreamScheduler into steps of exactly 128 ROIs, GPU crop inside the step: ROIs/s AND images/s
  bop7_stream              configs[4]  the same image-stream feed for all seven BOP datasets (one stream + model each, cycled per step);
                           with --host-fed every image starts in pinned host memory: detections -> pose -> refine end to end

  --with-yolox-post (stream workloads): every image's detections are produced inside the timed loop by gdrnpp_yolox_postprocess from a
  seeded YOLOX head output [8400, 5 + C] (decode + class-aware NMS on the device, eight images per launch) and handed to the
  scheduler without the reference's JSON file; the line reports yolox_post.ms_per_image (device) and the host's share.

  --host-fed (stream workloads): every image starts in pinned host memory (the reference's loader hands over host arrays) and is
  copied once on the scheduler's copy stream, one step ahead of the device; the line reports h2d_ms_per_step, h2d_overlapped_frac
  (device timeline of copy vs step events) and the resident-pool rate of the same process.

Multi-GPU: ``python bench.py --gpus N`` spawns N ranks by itself (one process per GPU, RCCL); under
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` it joins the launcher's ranks instead.
ROIs are sharded (every rank owns its contiguous block of the global ROI ids, weak scaling); the only collective is the
all-gather of the pose records (``--gather-to-rank0``: a gather to the main process, as the reference's evaluate() only lets that
one write), and rank 0 checks that the gathered block holds every ROI id exactly once.  ``--force-dist`` runs the same path with a
one-rank process group (what a 1-GPU box can show of it); the line's ``collective`` block is read from the process group itself.

Rank 0 prints ONE JSON line (driver contract) carrying two extra objects:
  roofline      the dominant kernel family, the split GEMMs (MFMA-bound): MFMA flops executed / summed launch durations,
                every launch bracketed by HIP events on the launch stream in a SEPARATE pass of the same steps right after
                the timed region (the timed region itself carries no per-launch events); ``traffic`` = HBM-side bytes per launch
                measured IN this run: two child runs of this command under ``rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace``
                (``--no-pmc``: the committed figure instead); ``roofline_other_kernels``: the refine kernel and the other
                hand-written kernels against the HBM / vector roofline.
  parity_in_run records of the timed batches under the headline arithmetic vs the exact six-product form (max |dR|, max |dt|), and the
                refine stage's t of 16 ROIs vs the CPU oracle (computed in the cpu_baseline child)
  cpu_baseline  oracle/cpu_baseline.py in a child process on the host cores (N = 1 only): refine stage 1 thread and all
                cores, uncertainty-PnP, decode, the reference's own compiled FPS / NN-distance / flow sources, and the network
                forward with PyTorch's CPU operators (configs[0]: ResNet-34 forward + uncertainty-PnP end to end).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gdrnpp_bop2022_amd import synthetic as S  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.config import get_cfg  # noqa: E402
from gdrnpp_bop2022_amd.gdrn_modeling.engine import class_sorted_order, gather_records, shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3  # same guide: fp32-input MFMA (1/16 of bf16)
# `traffic` is NOT measured by this process (PMC counters need rocprofv3 around it): it is the per-launch figure of the
# builder's own counter passes over this same command, committed with the profile they came from
PMC_SOURCE = ("profiles/pmc_traffic.json: builder-run rocprofv3 --pmc passes over this command (FETCH_SIZE / WRITE_SIZE, "
              "corrected per MI355X_MICROARCH.md), not re-measured in this run")

WORKLOADS = {  # name -> (index into BASELINE.json configs, cfg names, ROIs per GPU, refine, label)
    "lmo_upnp": (0, ["lmo_resnet34_ape"], 32, False, "LM-O ape, ResNet-34 forward + uncertainty-PnP (pn = 9 keypoints per ROI)"),
    "rgb": (1, ["ycbv_convnext_a6"], 64, False, "YCB-V convnext_a6, RGB-only Patch-PnP"),
    "refine": (2, ["ycbv_convnext_a6"], 128, True, "YCB-V convnext_a6 + fast depth refine (render-compare)"),
    "tless": (3, ["tless_convnext_a6"], 128, True, "T-LESS 30 objects convnext_a6 + fast depth refine, ROI-sharded"),
    "bop7": (4, [f"{d}_convnext_a6" for d in ("lmo", "ycbv", "tless", "icbin", "hb", "itodd", "tudl")], 128, True,
             "BOP-7 mixed stream (lmo/ycbv/tless/icbin/hb/itodd/tudl convnext_a6 models cycled per step) + fast depth refine"),
    # not a BASELINE.json config (index None): the class-agnostic head of the reference's single-object config families
    # not a BASELINE.json config of its own: configs[2]'s model fed by an image stream through the ROI packer
    "stream": (None, ["ycbv_convnext_a6"], 128, True,
               "YCB-V convnext_a6 + fast depth refine on a stream of 480x640 images (3-30 detections each) packed into 128-ROI steps, GPU crop in the step"),
    "bop7_stream": (4, [f"{d}_convnext_a6" for d in ("lmo", "ycbv", "tless", "icbin", "hb", "itodd", "tudl")], 128, True,
                    "BOP-7 mixed stream end to end: per dataset a stream of 480x640 images (3-30 detections each) -> ROI packer -> GPU crop -> "
                    "convnext_a6 forward -> fast depth refine -> records; the seven datasets' models cycled per 128-ROI step"),
    "ycbv_so": (None, ["ycbv_convnext_so"], 128, True,
                "YCB-V single-object convnext (configs/gdrn/ycbvSO/*, class-agnostic head) + fast depth refine"),
}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="auto", choices=["auto"] + sorted(WORKLOADS),
                   help="auto = refine (BASELINE configs[2]) below 8 GPUs, tless (config 4: 1024 ROIs over 8 ranks) at 8")
    p.add_argument("--batch", type=int, default=0, help="ROIs per GPU per step (0 = the workload's batch)")
    p.add_argument("--graph", action="store_true", help="replay the whole step from a captured hipGraph")
    p.add_argument("--with-crop", action="store_true",
                   help="start each step from full images: GPU ROI crop-resize (row a1) feeds the forward")
    p.add_argument("--subdiv", type=int, default=4, help="icosphere subdivision of the synthetic meshes (4 = 2562V/5120F)")
    p.add_argument("--random-init", action="store_true", help="A/B: PyTorch default initialisation instead of the seeded O(1) parameters")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=20.0, help="host wall time spent on the CPU baseline stages")
    p.add_argument("--no-roofline-pass", action="store_true", help="skip the per-launch event pass after the timed region")
    p.add_argument("--no-pmc", action="store_true",
                   help="do not measure roofline.traffic live (two rocprofv3 --pmc child runs of this command, ~1-2 min); the committed "
                        "profiles/pmc_traffic.json figure is reported instead")
    p.add_argument("--pmc-child", action="store_true", help="(internal) the run rocprofv3 wraps: steps only, nothing after the timed region")
    p.add_argument("--pmc-timeout", type=float, default=240.0, help="seconds allowed per rocprofv3 counter pass")
    p.add_argument("--exact-reference-order", action="store_true",
                   help="run the reference's full 1470-channel output layer + gather instead of the class-sliced one")
    p.add_argument("--no-hip-layers", action="store_true", help="A/B: run the memory-bound network layers with PyTorch ops")
    p.add_argument("--no-conv-gn-fusion", action="store_true", help="A/B: GroupNorm statistics in their own pass instead of the conv epilogue")
    p.add_argument("--mlp-gemm", choices=["split", "torch"], default="split",
                   help="GEMM engine of the ConvNeXt MLPs / head convolutions: split = exact 3-way bf16 operand split on the "
                        "bf16 matrix cores (fp32-accurate); torch = hipBLASLt / MIOpen fp32 + separate elementwise kernels")
    p.add_argument("--gemm-products", type=int, choices=[6, 3], default=3,
                   help="partial products per fp32 product in the split GEMMs (hip_layers.set_gemm_products): 3 = the library default "
                        "(fp16x2 operand split where the batch is large enough, overflow detected per step and repeated with 6), "
                        "6 = bf16x3 everywhere (exact to 2^-26)")
    p.add_argument("--no-fused-mlp", action="store_true", help="A/B: stage-0 ConvNeXt MLPs as two three-product launches instead of the fused kernel")
    p.add_argument("--fused-mlp-max-c", type=int, default=256, help="A/B: widest ConvNeXt block on the fused MLP kernel (128 = stage 0 only, 256 = stages 0 and 1)")
    p.add_argument("--fused-mlp-min-rows", type=int, default=None, help="A/B: fewest pixels of a block for the fused MLP kernel (default 32768)")
    p.add_argument("--no-f16x2-rows", action="store_true", help="A/B: fp32 tensors between dwconv+LN / fc1 / fc2 of a ConvNeXt block instead of the pre-split f16x2-rows hand-over")
    p.add_argument("--compute-streams", type=int, default=0,
                   help="HIP streams consecutive steps are dealt to (engine.StepStreams): 2 = two independent steps in flight on the device, "
                        "1 = the single-stream schedule (reported beside the headline as single_stream_mode); 0 (default) = "
                        "engine.default_compute_streams: 2 for the ConvNeXt configurations (every kernel of a step is this library's), "
                        "1 for the ResNet-34 one (MIOpen kernels in the step)")
    p.add_argument("--stream-priorities", default="", help="A/B: HIP priorities of the compute streams, e.g. -1,0 (default: all 0)")
    p.add_argument("--no-other-mode-line", action="store_true",
                   help="skip the extra measurement of the other --gemm-products setting after the timed region")
    p.add_argument("--opt", action="append", default=[], metavar="NAME=INT",
                   help="library tuning switch (gdrnpp_set_option), e.g. --opt split_gemm_glds=0 for A/B measurements")
    p.add_argument("--host-fed", action="store_true",
                   help="stream workload: images / depth maps start in PINNED HOST memory (the reference's loader hands over host arrays, "
                        "data_loader.py:754-797); hipMemcpyAsync on a copy stream, one step ahead of the device, inside the timed loop")
    p.add_argument("--with-yolox-post", action="store_true",
                   help="stream workloads: the detections of every image come out of gdrnpp_yolox_postprocess INSIDE the timed loop — a seeded "
                        "YOLOX head output f32[8400, 5 + C] per image (resident in HBM, as the detector network leaves it) -> decode + "
                        "class-aware NMS on the device -> engine.detections_from_yolox -> the scheduler (det/yolox/utils/boxes.py:34-74, "
                        "demo/predictor_yolo.py:84-165): configs[4]'s detections -> pose -> refine leg from the detector's raw output on")
    p.add_argument("--gather-to-rank0", action="store_true",
                   help="gather the records to rank 0 only (dist.gather; the reference lets only the main process write, "
                        "gdrn_evaluator.py:581-582) instead of the all-gather")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise the process group even with one rank and send the records through the collective: the closest a "
                        "1-GPU box gets to the N > 1 path (RCCL loads, rendezvous on 127.0.0.1, all_gather_into_tensor on the device)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --stub-step: CPU test of the launch path")
    p.add_argument("--stub-step", action="store_true",
                   help="(tests) replace the GPU step by a host stub that emits this rank's records: exercises spawn, "
                        "rendezvous, sharding, gather, id check and the JSON line without a device")
    return p.parse_args(argv)


def algorithmic_bytes_refine(b, iters, n_verts, n_faces):
    """SURVEY.md §8(d) row a8, per ROI: (3+1) maps x 64^2 x 4 B + the used quarter of the 256^2 depth crop
    (64^2 x 4 x 4 B) + K,R,t (84 B) + iters x (12 V + 12 F) mesh bytes + 12 B written."""
    per_roi = 65536 + 65536 + 84 + iters * (12 * n_verts + 12 * n_faces) + 12
    return b * per_roi, per_roi


def pmc_traffic_live(args, wname, b):
    """HBM-side bytes per launch of the split-GEMM family and of the refine kernel, measured NOW: two child runs of this very
    command under ``rocprofv3 --pmc FETCH_SIZE --kernel-trace`` / ``--pmc WRITE_SIZE --kernel-trace`` (separate passes: the two
    counters do not fit the TCC's four slots together, MI355X_MICROARCH.md "rocprofv3 PMC slots"), no other trace domain.  Units
    KiB; gfx950 correction of that guide's HBM section: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> x2;
    WRITE_SIZE x1 (calibrated on fills, profiles/pmc_traffic.json).  Returns (dict | None, note)."""
    import csv
    import glob
    import shutil

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--no-pmc", "--steps", "2", "--warmup", "1", "--workload", wname,
             "--batch", str(b), "--gemm-products", str(args.gemm_products), "--mlp-gemm", args.mlp_gemm, "--subdiv", str(args.subdiv)]
    for flag, on in (("--no-fused-mlp", args.no_fused_mlp), ("--no-f16x2-rows", args.no_f16x2_rows), ("--random-init", args.random_init),
                     ("--with-crop", args.with_crop), ("--host-fed", args.host_fed), ("--no-hip-layers", args.no_hip_layers),
                     ("--with-yolox-post", args.with_yolox_post)):
        if on:
            child.append(flag)
    for o in args.opt:
        child += ["--opt", o]
    raw = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gdrnpp_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=args.pmc_timeout)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {r.returncode}: {r.stderr[-300:]}"
            per = {"gemm": [], "refine": []}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != counter:
                        continue
                    kn = row["Kernel_Name"]
                    if ("gemm_split" in kn or "mlp_fused_x3" in kn) and "reduce" not in kn and "pack" not in kn:
                        per["gemm"].append(float(row["Counter_Value"]))
                    elif "depth_refine_kernel" in kn:
                        per["refine"].append(float(row["Counter_Value"]))
            if not per["gemm"]:
                return None, f"rocprofv3 --pmc {counter}: no split-GEMM rows in the counter file"
            raw[counter] = per
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} did not finish in {args.pmc_timeout:.0f} s"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    mean = lambda v: sum(v) / len(v) if v else None  # noqa: E731
    out = {"seconds": time.perf_counter() - t0}
    for k in ("gemm", "refine"):
        f_, w_ = mean(raw["FETCH_SIZE"][k]), mean(raw["WRITE_SIZE"][k])
        if f_ is not None and w_ is not None:
            out[k] = {"traffic_bytes_per_launch": (2.0 * f_ + w_) * 1024.0, "fetch_kib_raw_mean": f_, "write_kib_raw_mean": w_,
                      "launches_fetch_pass": len(raw["FETCH_SIZE"][k]), "launches_write_pass": len(raw["WRITE_SIZE"][k])}
    return out, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around child runs of "
                 "this command (--steps 2 --warmup 1, same workload and switches); KiB units, FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md), WRITE_SIZE x1")


def resolve_workload(args):
    name = args.workload
    if name == "auto":
        name = "tless" if args.gpus >= 8 else "refine"
    return name


# ------------------------------------------------------------------------------------------------------------------
# launch: --gpus N without a launcher spawns the ranks itself
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawned(rank, world, port, argv):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worker(parse(argv))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not args.stub_step:
            have = torch.cuda.device_count()
            if have < args.gpus:
                raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible on this node")
        import torch.multiprocessing as tmp
        tmp.spawn(_spawned, args=(args.gpus, _free_port(), list(argv)), nprocs=args.gpus, join=True)
        return
    worker(args)


# ------------------------------------------------------------------------------------------------------------------
def worker(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the two must agree")
    if args.stub_step:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a HIP device"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" IS RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")

    wname = resolve_workload(args)
    cfg_no, cfg_names, b_default, refine, label = WORKLOADS[wname]
    b = args.batch or b_default
    n_global = world * b
    lo, hi = shard_range(n_global, rank, world)          # this rank's contiguous block of the global ROI ids
    assert hi - lo == b
    roi_ids = torch.arange(lo, hi, dtype=torch.int32, device=dev)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    if args.stub_step:
        state = None

        def step(i):
            rec = torch.zeros((b, 16), dtype=torch.float32)
            rec[:, 14] = roi_ids.float()
            rec[:, 15] = 1.0
            return rec
    else:
        state = build_state(args, cfg_names, refine, wname, b, rank, dev, lo)
        step = state["step"]

    launch = state["launch"] if state is not None else (lambda i: (lambda: step(i)))
    dst = 0 if args.gather_to_rank0 else None
    depth = state["compute_streams"] if state is not None else max(1, args.compute_streams)   # launches ahead of the oldest unresolved step

    def run_steps(n):
        """n steps, each resolved (range check + gather) after the next `depth` ones have been launched: with two compute streams
        two steps stay in flight on the device while the host reads the oldest one's verdict."""
        rec, pend = None, []
        for i in range(n):
            pend.append(launch(i))
            if len(pend) > depth:
                rec = gather_records(pend.pop(0)(), b, dst=dst, single_rank_collective=args.force_dist)
        while pend:
            rec = gather_records(pend.pop(0)(), b, dst=dst, single_rank_collective=args.force_dist)
        return rec

    n_warm = max(args.warmup, 1) * len(cfg_names) * 2      # MIOpen find, weight packing, first range verdicts, both batches of every model
    if args.graph and state is not None:                  # ... and every hipGraph slot captured (the scheduler's graph steps: 2 x streams slots)
        n_warm = max(n_warm, 2 * state["compute_streams"] + 2)
    run_steps(n_warm)
    sync()
    if state is not None and state.get("after_warmup"):
        state["after_warmup"]()
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    rec = run_steps(args.steps)
    sync()
    if use_dist:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # the gathered block holds every ROI id of the iteration exactly once, on every rank (on rank 0 with --gather-to-rank0)
    if dst is not None and rank != dst:
        assert rec is None
    else:
        ids = rec[:, 14][rec[:, 15] > 0.5].to(torch.int64).cpu().numpy()
        if wname.endswith("stream"):     # stream ids keep counting: the last step holds n_global distinct consecutive ids per rank block
            assert len(ids) == n_global and len(set(ids.tolist())) == n_global, "gathered records: ROI ids not distinct"
        else:
            assert len(ids) == n_global and np.array_equal(np.sort(ids), np.arange(n_global)), "gathered records: ROI ids not a permutation"

    gather_ms, collective = None, None
    if use_dist:                                            # the collective alone, after the timed region
        local = torch.zeros((b, 16), dtype=torch.float32, device=dev)
        local[:, 14] = roi_ids.float()
        for _ in range(3):
            gather_records(local, b, dst=dst, single_rank_collective=args.force_dist)
        sync()
        g0 = time.perf_counter()
        for _ in range(20):
            gather_records(local, b, dst=dst, single_rank_collective=args.force_dist)
        sync()
        gather_ms = (time.perf_counter() - g0) / 20 * 1e3
        # what the process group itself reports: a SCALE run proves from this that RCCL really saw N ranks
        collective = {"op": "gather(dst=0)" if dst is not None else "all_gather_into_tensor", "backend": dist.get_backend(),
                      "world_size_seen": dist.get_world_size(), "rank0_device": str(dev), "payload": "f32[rois_per_gpu,16]",
                      "bytes_per_rank": b * 16 * 4, "bytes_received_per_rank": (world - 1) * b * 16 * 4,
                      "calls_per_step": 1, "ms_alone": gather_ms}

    extras = {}
    if args.pmc_child:          # the run rocprofv3 wraps: the steps above are all it is for
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        print(json.dumps({"pmc_child": True, "ms_per_step": dt / args.steps * 1e3}), flush=True)
        return
    if state is not None:
        extras = state["measure_after"](rank == 0 and world == 1)
    if state is not None and rank == 0 and world == 1 and not args.no_pmc and not args.graph and extras.get("roofline"):
        live, note = pmc_traffic_live(args, wname, b)
        rl = extras["roofline"]
        if live is not None:
            key = "gemm" if rl.get("bound") == "mfma" else "refine"
            if key in live:
                rl["traffic_committed_file"] = rl.get("traffic")
                rl["traffic"] = live[key]["traffic_bytes_per_launch"]
                rl["traffic_source"] = note
                rl["traffic_detail"] = live[key]
                if rl.get("algorithmic_bytes_per_launch"):
                    rl["traffic_over_algorithmic"] = rl["traffic"] / rl["algorithmic_bytes_per_launch"]
            for other in extras.get("roofline_other_kernels") or []:
                if other and "depth_refine" in str(other.get("kernel")) and "refine" in live:
                    other["traffic_committed_file"] = other.get("traffic")
                    other["traffic"] = live["refine"]["traffic_bytes_per_launch"]
                    other["traffic_source"] = note + "; maps / depth of the alternating batches partly L2 / Infinity-Cache resident (the committed figure evicted them)"
            extras["pmc_live_seconds"] = live["seconds"]
        else:
            rl["traffic_live_error"] = note
    if state is not None and world == 1 and not args.no_other_mode_line and args.mlp_gemm == "split" and not args.graph \
            and not args.no_hip_layers:
        # the same K steps once more with the other split-GEMM setting (reported beside the headline, never as `value`)
        other = 6 if args.gemm_products == 3 else 3
        extras["six_product_mode" if other == 6 else "three_product_mode"] = state["other_mode_line"](other, args.steps, n_global, sync)
    rl = extras.get("roofline") if extras else None
    if rl and rl.get("bound") == "mfma" and rl.get("flops_per_launch"):
        # the matrix-core rate of the WHOLE step as timed (every kernel of it, two steps in flight): MFMA flops per step / step time
        tf = rl["flops_per_launch"] * rl["launches_per_step"] / (dt / args.steps) / 1e12
        rl["whole_step"] = {"mfma_tflops": tf, "frac_of_peak": tf / rl["peak"], "gemm_share_of_step_alone": rl["ms_per_step"] / (dt / args.steps * 1e3),
                            "note": "MFMA flops of one step / ms_per_step of the timed region; gemm_share_of_step_alone = summed GEMM launch time "
                                    "measured alone / the timed step time (can exceed what a single stream could hold: the rest of the step runs beside them)"}
    n_cs = state["compute_streams"] if state is not None else 1
    if state is not None and world == 1 and not args.no_other_mode_line and n_cs > 1 and state["single_stream_line"] is not None:
        extras["single_stream_mode"] = state["single_stream_line"](args.steps, n_global, sync, rec)
    if rank == 0:
        metric = ("ROIs/sec (GDRNPP fwd + PnP + depth refine), 256x256 crops" if refine else
                  "ROIs/sec (GDRNPP fwd + uncertainty-PnP), 256x256 crops" if wname == "lmo_upnp" else
                  "ROIs/sec (GDRNPP fwd + Patch-PnP, RGB only), 256x256 crops")
        product_path = args.mlp_gemm == "split" and not args.no_hip_layers and not args.stub_step
        if not product_path:      # hip_layers keeps PyTorch's operators selectable for A/B measurements and CPU graph tests: never the headline
            metric += " [A/B run: " + ("host stub" if args.stub_step else "PyTorch-operator layers") + ", NOT the product path]"
        line = {
            "metric": metric, "product_path": product_path, "value": n_global * args.steps / dt, "unit": "ROIs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",   # fp32 in, fp32 accumulate, fp32 out; how the operands enter the matrix cores:
            "arithmetic": (("fp16x2 operand split (22 bits) on the fp16 matrix cores, fp32 accumulate; range checked per launch on both sides, "
                            "out-of-range layers repeated / kept on the exact bf16x3 form" if args.gemm_products == 3 else
                            "bf16x3 exact operand split (six partial products) on the bf16 matrix cores, fp32 accumulate")
                           if args.mlp_gemm == "split" else "fp32 vendor kernels (hipBLASLt / MIOpen)"),
            "data": "synthetic (seeded ROIs sorted by class within the rank, two alternating batches per model, ellipsoid meshes "
                    "2562V/5120F, " + ("PyTorch default-init weights" if args.random_init else
                                        "seeded O(1) parameters = synthetic.seeded_state_dict, the parity tests' set") +
                    ", t-head bias = z_rel prior)",
            "config": {
                "workload": f"{label}, batch={b} ROIs/GPU" + (f", {n_global} ROIs per iteration over {world} ranks" if world > 1 else ""),
                "baseline_config_index": cfg_no, "workload_key": wname, "global_batch": n_global, "rois_per_gpu": b,
                "roi_prep_on_gpu": bool(args.with_crop) or wname.endswith("stream"), "host_fed": bool(args.host_fed), "hipgraph": bool(args.graph), "input_res": 256, "output_res": 64,
                "parallelism": f"roi-shard x{world}",
                "compute_streams": n_cs,
                "compute_stream_overlap_probe": (state["overlap_probe"]() if state is not None and state.get("overlap_probe") else None),   # [(pool streams tried, overlap ratio of two spin kernels: ~2 = concurrent)]
                "steps_in_flight": ("consecutive steps dealt round-robin to %d HIP streams (engine.StepStreams): independent batches, "
                                                             "records bit-equal to the single-stream schedule (tests/test_gpu_streams2.py)" % n_cs) if n_cs > 1 else "one stream",
                "collective": (("gather(dst=0)" if dst is not None else "all_gather") + " f32[n,16] pose records") if use_dist else None,
                "class_sliced_out_layer": not args.exact_reference_order, "rois_class_sorted_within_rank": True,
                "parameters": "default-init" if args.random_init else "seeded O(1)", "hip_network_layers": not args.no_hip_layers,
                "mlp_gemm": args.mlp_gemm, "gemm_products": args.gemm_products, "fused_mlp": (not args.no_fused_mlp) and args.fused_mlp_max_c, "f16x2_rows": not args.no_f16x2_rows,
                "gemm_numerics": ("fp32 operands, fp32 accumulation, fp32 results; operands enter the fp16 matrix cores as two fp16 values "
                                  "(22 significant bits), three partial products; measured error against fp64 = that of the six-product "
                                  "bf16x3 form and below hipBLASLt's fp32 GEMM on the same operands, network outputs at the same distance "
                                  "from the reference's recorded forward (profiles/r03y_split2_accuracy.txt, tests/test_gpu_split2.py); "
                                  "both sides of the fp16x2 range checked on the device by every launch (range word per layer), a flagged step is "
                                  "repeated with six products and the layer kept there; six_product_mode = the same steps with the exact form"
                                  if args.gemm_products == 3 else
                                  "fp32 operands split exactly into three bf16 values, six partial products, fp32 accumulation (exact to 2^-26)"),
                "library_options": args.opt,
                "timed_entry_point": (("engine.RoiStreamScheduler(graph_steps=True).launch_next (static-buffer fill + hipGraph replay of GPU crop + forward + post) + engine.gather_records"
                                       if args.graph and wname == "stream" else
                                       "engine.RoiStreamScheduler.launch_next (GPU crop + inference_step_async) + engine.gather_records")
                                      if wname.endswith("stream") else
                                      "engine.GraphedStepStreams.launch / GraphHandle.result + engine.gather_records" if args.graph else
                                      "engine.inference_step_async / StepHandle.result + engine.gather_records"),
                "stub_step": bool(args.stub_step)},
            "gather_ms": gather_ms, "collective": collective,
        }
        line.update(extras)
        if "stream" in line:
            line["stream"]["images_per_s"] = line["value"] / line["stream"]["rois_per_image_mean"]
        line.setdefault("roofline", None)
        line.setdefault("cpu_baseline", None)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
def build_state(args, cfg_names, refine, wname, b, rank, dev, lo):
    from gdrnpp_bop2022_amd import hip_lib
    from gdrnpp_bop2022_amd.gdrn_modeling import hip_layers
    from gdrnpp_bop2022_amd.gdrn_modeling import engine as E
    from gdrnpp_bop2022_amd.gdrn_modeling.engine import GdrnHipPost, GraphedInference, inference_step, inference_step_async
    from gdrnpp_bop2022_amd.gdrn_modeling.GDRN_double_mask import build_model_optimizer

    hip_lib.load()
    for o in args.opt:
        k_, v_ = o.split("=")
        hip_lib.set_option(k_, int(v_))
    torch.backends.cudnn.benchmark = True  # MIOpen find mode during warm-up
    hip_layers.set_enabled(not args.no_hip_layers)
    hip_layers.set_conv_gn_fused(not args.no_conv_gn_fusion)
    hip_layers.set_mlp_gemm(args.mlp_gemm)
    hip_layers.set_gemm_products(args.gemm_products)
    hip_layers.set_fused_mlp_x3(not args.no_fused_mlp, args.fused_mlp_max_c, args.fused_mlp_min_rows)
    hip_layers.set_f16x2_rows(not args.no_f16x2_rows)
    opts = ["TEST.USE_DEPTH_REFINE=True", "INPUT.WITH_DEPTH=True"] if refine else []

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def make_batch(cfg, rng, ext, meshes, K=S.YCBV_K):
        C = cfg.MODEL.POSE_NET.NUM_CLASSES
        det = S.make_detections(b, C, ext, rng, K=K)
        # ROIs sorted by class within the rank, the record carries the ROI's global id
        order = class_sorted_order(det["roi_cls"])
        det = {k_: v_[order] for k_, v_ in det.items()}
        batch = dict(
            roi_id=T((lo + order).astype(np.int32)),
            roi_img=torch.rand(b, 3, 256, 256, device=dev), roi_cls=T(det["roi_cls"]), roi_cam=T(det["roi_cam"]),
            roi_wh=T(det["roi_wh"]), roi_center=T(det["roi_center"]), resize_ratio=T(det["resize_ratio"]),
            roi_coord_2d=T(S.coord2d_roi(det["roi_center"], det["scale"])), roi_extent=T(det["roi_extent"]),
            scale=T(det["scale"]), score=T(det["score"]), im_W=T(det["im_W"]), im_H=T(det["im_H"]))
        K_crop = S.zoom_K_np(det["roi_cam"], det["roi_center"], det["scale"], 64)
        if refine:
            # sensor depth: HIP render of the GT pose at 64^2, nearest x4 to 256^2, + N(0, 2 mm), 5 % holes (§8d)
            depth = hip_lib.render_depth(meshes, T(det["roi_cls"].astype(np.int32)), T(K_crop), T(det["R_gt"]), T(det["t_gt"]), 64)
            big = depth.repeat_interleave(4, 1).repeat_interleave(4, 2)
            g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
            noisy = torch.where(big > 0, big + 0.002 * torch.randn(big.shape, device=dev, generator=g), big)
            drop = torch.rand(big.shape, device=dev, generator=g) < 0.05
            batch["roi_depth"] = torch.where(drop, torch.zeros_like(noisy), noisy)[:, None].contiguous()
        if args.with_crop:
            g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
            batch["images"] = torch.randint(0, 256, (16, S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=dev, generator=g)
            batch["depths"] = torch.rand((16, S.IM_H, S.IM_W), device=dev, generator=g) + 0.3
            batch["im_idx"] = torch.from_numpy(rng.integers(0, 16, b).astype(np.int32)).to(dev)
            batch["center64"] = torch.from_numpy(det["roi_center"].astype(np.float64)).to(dev)
            batch["scale64"] = torch.from_numpy(det["scale"].astype(np.float64)).to(dev)
        return batch, det, K_crop

    models = []
    for di, name in enumerate(cfg_names):
        cfg = get_cfg(name, opts)
        torch.manual_seed(20220925)  # identical weights on every rank; the data below is per-rank
        rng = np.random.default_rng(20220925 + 3 + rank + 100 * di)
        model, _ = build_model_optimizer(cfg, is_test=True)
        model.exact_reference_order = bool(args.exact_reference_order)
        # No checkpoint exists offline.  The parameters are the seeded O(1) set of the parity tests (synthetic.seeded_param:
        # fan-in scaled weights, norm scales 1 +- 0.2, ConvNeXt layer scale 0.4 +- 0.2 — NOT timm's 1e-6 initial layer scale,
        # which mutes every MLP), so the timed operands are the ones tests/test_gpu_headline_shapes.py checks.
        if not args.random_init:
            model.load_state_dict(S.seeded_state_dict([(k_, tuple(v_.shape)) for k_, v_ in model.state_dict().items()], 20220925),
                                  strict=True)
        # Such weights predict t ~ 0 (object at the camera centre), which no trained model does and which would make
        # every triangle straddle the camera plane.  Set the translation head's bias to the dataset prior of the
        # scale-invariant depth z_rel = t_z / resize_ratio (pose_from_pred_centroid_z.py:84-90): poses land in the frustum.
        with torch.no_grad():
            model.pnp_net.fc_t.bias.copy_(torch.tensor([0.0, 0.0, 1.5 * float(S.YCBV_K[0, 0]) * 0.19 / 64.0]))
        C = cfg.MODEL.POSE_NET.NUM_CLASSES
        if wname == "lmo_upnp":
            ext = np.array([[0.076, 0.078, 0.092]], np.float32)     # ape-sized ellipsoid
            sv, sf = S.icosphere(args.subdiv)
            verts, faces = [(sv * ext[0] / 2).astype(np.float32)], [sf]
        else:
            verts, faces, ext = S.make_models(C, np.random.default_rng(20220925 + di), subdiv=args.subdiv)
        meshes = hip_lib.MeshSet(verts, faces, device=dev)
        post = GdrnHipPost(cfg, meshes if refine else None)
        pair = [make_batch(cfg, rng, ext, meshes, K=S.LMO_K if wname == "lmo_upnp" else S.YCBV_K) for _ in range(2)]
        models.append(dict(cfg=cfg, model=model, post=post, batches=[p[0] for p in pair], dets=[p[1] for p in pair],
                           K_crops=[p[2] for p in pair], meshes=meshes, verts=verts, faces=faces, C=C, graphs={}, ext=ext))

    if args.compute_streams <= 0:        # default: two steps in flight only where every kernel of a step is this library's
        default_streams = E.default_graph_streams if (args.graph and not args.with_crop and wname not in ("lmo_upnp", "bop7_stream")) \
            else E.default_compute_streams            # hipGraph replays leave the host free to keep four steps in flight
        args.compute_streams = min(default_streams(m_["model"], m_["cfg"]) for m_ in models)
    else:                                # an explicit --compute-streams N is an A/B request: the dealer keeps sharing whatever a step launches
        args.allow_foreign_streams = True

    stream = None
    if wname in ("stream", "bop7_stream"):
        # the reference's feed: one image at a time (data_loader.py:901, batch_size = 1), 3-30 detections each.  64 distinct images
        # + detections resident in HBM, cycled; every pushed image gets a fresh key, ROI ids keep counting (per-rank id block).
        import itertools
        counter = itertools.count()
        YOLOX_GROUP = 8
        yolox = dict(stream=torch.cuda.Stream(dev, priority=-1), events=[], host_s=0.0, images=0, dets=0) if args.with_yolox_post else None
        subs = []                 # one image stream + scheduler per model ("stream": YCB-V; "bop7_stream": the seven BOP datasets)
        sched_streams = E.StepStreams(max(1, args.compute_streams), dev, allow_foreign=bool(getattr(args, "allow_foreign_streams", False)))      # one dealer for all of them: consecutive steps alternate
        for di, m_ in enumerate(models):
            rng = np.random.default_rng(20220925 + 17 + rank + 1000 * di)
            g = torch.Generator(device=dev).manual_seed(20220925 + rank + 1000 * di)
            pool = []
            for _ in range(64 if len(models) == 1 else 24):
                n = int(rng.integers(3, 31))
                det = S.make_detections(n, m_["C"], m_["ext"], rng)
                x1y1 = det["roi_center"] - det["roi_wh"] / 2
                pool.append((torch.randint(0, 256, (S.IM_H, S.IM_W, 3), dtype=torch.uint8, device=dev, generator=g),
                             torch.rand((S.IM_H, S.IM_W), device=dev, generator=g) + 0.3,
                             dict(bbox=np.concatenate([x1y1, x1y1 + det["roi_wh"]], 1).astype(np.float32), roi_cls=det["roi_cls"],
                                  score=det["score"], cam=S.YCBV_K.astype(np.float32), extents=m_["ext"])))
            # --host-fed: the very same images, but every push starts from PINNED HOST memory (the reference's loader hands over host
            # arrays); the scheduler copies them on its copy stream, one step ahead of the device
            host_pool = [(im.cpu().pin_memory(), dp.cpu().pin_memory(), dt_) for im, dp, dt_ in pool] if args.host_fed else None
            heads = None
            if args.with_yolox_post:
                # what the detector network leaves in HBM for each image: f32[8400, 5 + C] = (cx, cy, w, h, obj, class scores); every
                # synthetic detection is predicted by four jittered confident anchors (NMS keeps one), the other anchors are clutter
                # below the confidence threshold.  Groups of YOLOX_GROUP images are post-processed by ONE launch.
                A = 8400
                hd = np.zeros((len(pool), A, 5 + m_["C"]), np.float32)
                hd[..., 0] = rng.uniform(0, S.IM_W, (len(pool), A)); hd[..., 1] = rng.uniform(0, S.IM_H, (len(pool), A))
                hd[..., 2:4] = rng.uniform(10, 60, (len(pool), A, 2)); hd[..., 4] = rng.uniform(0, 0.2, (len(pool), A))
                hd[..., 5:] = rng.uniform(0, 0.5, (len(pool), A, m_["C"]))
                for pi_, (_, _, dt_) in enumerate(pool):
                    bb = dt_["bbox"]
                    slots = rng.permutation(A // 4)[:len(bb)] * 4
                    for (x1, y1, x2, y2), cl, sc, a0 in zip(bb, dt_["roi_cls"], dt_["score"], slots):
                        for j in range(4):
                            row = hd[pi_, a0 + j]
                            row[:4] = (0.5 * (x1 + x2) + 0.3 * j, 0.5 * (y1 + y2) - 0.3 * j, (x2 - x1) + 0.5 * j, (y2 - y1) - 0.4 * j)
                            row[4] = 0.97 - 0.01 * j
                            row[5:] = 0.01
                            row[5 + int(cl)] = min(0.99, max(0.6, float(sc)))
                heads = torch.from_numpy(hd).to(dev)

            def make_feeder(src, heads=heads, C=m_["C"], ext=m_["ext"]):
                if heads is None:
                    return ((next(counter), im, dp, dt_) for im, dp, dt_ in itertools.cycle(src))

                def gen():
                    G = YOLOX_GROUP
                    for g0 in itertools.cycle(range(0, len(src), G)):
                        idx = list(range(g0, min(g0 + G, len(src))))
                        t_host = time.perf_counter()
                        with torch.cuda.stream(yolox["stream"]):      # its own (high-priority) stream: the host waits for THIS launch only,
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # not for the step in flight
                            e0.record()
                            dets, count = hip_lib.yolox_postprocess(heads[idx[0]:idx[-1] + 1], C, 0.5, 0.45, max_det=64)
                            e1.record()
                            d = E.detections_from_yolox(dets, count, S.YCBV_K.astype(np.float32), ext)     # one read-back per group
                        yolox["events"].append((e0, e1, len(idx)))
                        yolox["host_s"] += time.perf_counter() - t_host
                        yolox["images"] += len(idx)
                        yolox["dets"] += len(d["roi_cls"])
                        for j, pi_ in enumerate(idx):
                            sel = d["im_idx"] == j
                            yield (next(counter), src[pi_][0], src[pi_][1],
                                   dict(bbox=d["bbox"][sel], roi_cls=d["roi_cls"][sel], score=d["score"][sel], cam=d["cam"], extents=d["extents"]))
                return gen()

            def make_sched(m_=m_, timed=args.host_fed):
                return E.RoiStreamScheduler(m_["cfg"], m_["model"], m_["post"], rois_per_step=b, roi_id_base=lo, device=dev, time_h2d=timed,
                                            compute_streams=sched_streams, graph_steps=bool(args.graph) and wname == "stream")
            subs.append(dict(sched=make_sched(), feeder=make_feeder(host_pool if args.host_fed else pool), pool=pool, make_sched=make_sched,
                             make_feeder=make_feeder))
        stream = dict(subs=subs, counter=counter, h2d_bytes_warmup=0, yolox=yolox,
                      rois_per_image=float(np.mean([len(p[2]["roi_cls"]) for s_ in subs for p in s_["pool"]])))

    upnp = None
    if wname == "lmo_upnp":   # PVNet-style pose of config 1: 8 FPS keypoints + centre, noisy projections, cov^-1/2 weights
        m0 = models[0]
        kp_idx = hip_lib.fps(T(m0["verts"][0])[None], 8, init_center=True).cpu().numpy()[0]
        kpts = np.concatenate([m0["verts"][0][kp_idx], m0["verts"][0].mean(0, keepdims=True)], 0).astype(np.float64)
        rng = np.random.default_rng(20220925 + 1 + rank)
        K = S.LMO_K.astype(np.float64)
        upnp = []
        for det in m0["dets"]:
            cam = np.einsum("bij,kj->bki", det["R_gt"].astype(np.float64), kpts) + det["t_gt"][:, None].astype(np.float64)
            uv = cam[..., :2] / cam[..., 2:] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
            p2 = uv + rng.normal(0, 1.0, uv.shape)
            w = np.stack([rng.uniform(0.5, 1.4, (b, 9)), rng.uniform(-0.2, 0.2, (b, 9)), rng.uniform(0.5, 1.4, (b, 9))], -1)
            from scipy.spatial.transform import Rotation
            rv = Rotation.from_matrix(det["R_gt"].astype(np.float64)).as_rotvec()
            init = np.concatenate([rv, det["t_gt"].astype(np.float64)], 1) + rng.uniform(0, 0.05, (b, 6))
            upnp.append(dict(p2=T(p2), p3=T(np.repeat(kpts[None], b, 0)), w=T(w), K=T(np.repeat(K.reshape(1, 9), b, 0)), init=T(init)))

    prios = [int(v) for v in args.stream_priorities.split(",")] if args.stream_priorities else None
    allow_foreign = bool(getattr(args, "allow_foreign_streams", False))
    dealer = {"streams": E.StepStreams(1 if stream is not None else max(1, args.compute_streams), dev, prios, allow_foreign=allow_foreign)}

    def set_compute_streams(n):
        torch.cuda.synchronize(dev)
        dealer["streams"] = E.StepStreams(n, dev, prios, allow_foreign=allow_foreign)
        for m_ in models:                # hipGraphs were captured on the old dealer's streams
            m_["graphs"].clear()

    @torch.no_grad()
    def prepared(m, k):
        """The batch of step (model m, parity k), through the GPU ROI crop when --with-crop."""
        bt = m["batches"][k]
        if not args.with_crop:
            return bt
        roi_img, roi_depth, roi_c2d = hip_lib.crop_resize_roi(bt["images"], bt["depths"], bt["im_idx"], bt["center64"], bt["scale64"])
        return dict(bt, roi_img=roi_img, roi_coord_2d=roi_c2d)

    @torch.no_grad()
    def launch(i):
        """Launch step i, return the callable that resolves it (-> records f32[b,16])."""
        m = models[i % len(models)]
        k = (i // len(models)) % 2
        if stream is not None:
            s_ = stream["subs"][i % len(stream["subs"])]
            return s_["sched"].launch_next(s_["feeder"])
        if args.graph and not args.with_crop and upnp is None:
            if "gs" not in m["graphs"]:      # one graph per slot (a resident batch), slots dealt to the dealer's streams: n hipGraphs in flight
                n_st = len(dealer["streams"].streams)
                n_slots = 2 * ((n_st + 1) // 2) if len(models) == 1 else 2          # slot j holds batch j % 2
                m["graphs"]["gs"] = E.GraphedStepStreams(m["model"], m["post"], [m["batches"][j % 2] for j in range(n_slots)], compute_streams=dealer["streams"])
                m["graphs"]["n_slots"] = n_slots
            slot = i % m["graphs"]["n_slots"] if len(models) == 1 else k            # (one model: i % 2 == k, so slot % 2 == k)
            return m["graphs"]["gs"].launch(slot).result   # inputs already live in the graphs' static buffers (resident in HBM)
        with dealer["streams"].next():       # consecutive steps on alternating compute streams (engine.StepStreams); the crop too
            h = inference_step_async(m["model"], m["post"], prepared(m, k))     # records carry batch["roi_id"]
        if upnp is None:
            return h.result

        def resolve():
            rec = h.result()
            u = upnp[k]
            rt = hip_lib.uncertainty_pnp_batched(u["p2"], u["p3"], u["w"], u["K"], u["init"])
            rec[:, 9:12] = rt[:, 3:6].float()   # the PVNet-style pose replaces the direct translation in the records
            return rec
        return resolve

    def step(i):
        return launch(i)()

    def run_pipelined(n):
        pend, out = [], None
        depth = len(dealer["streams"].streams)
        for i in range(n):
            pend.append(launch(i))
            if len(pend) > depth:
                out = pend.pop(0)()
        while pend:
            out = pend.pop(0)()
        return out

    def other_mode_line(products, steps, n_rois, sync):
        try:
            hip_layers.set_gemm_products(products)
            hip_lib.split2_range_words(reset=True)
            reruns0 = E.range_reruns()
            run_pipelined(max(args.warmup, 3) * len(models) * 2)      # the same warm-up as the headline's
            sync()
            t0 = time.perf_counter()
            run_pipelined(steps)
            sync()
            dt = time.perf_counter() - t0
            return {"gemm_products": products, "value": n_rois * steps / dt, "unit": "ROIs/s", "ms_per_step": dt / steps * 1e3,
                    "steps": steps, "range_reruns": E.range_reruns() - reruns0,
                    "note": ("hip_layers.set_gemm_products(6): every split GEMM on the six-product bf16x3 kernels (exact to 2^-26)"
                             if products == 6 else
                             "hip_layers.set_gemm_products(3): ConvNeXt MLPs / 3x3 convolutions / deconv GEMM on the fp16x2 three-product "
                             "kernels; the per-step range check is inside the timing")}
        except Exception as e:  # the headline line must not depend on the extra measurement
            return {"gemm_products": products, "error": repr(e)}
        finally:
            hip_layers.set_gemm_products(args.gemm_products)

    def single_stream_line(steps, n_rois, sync, headline_rec=None):
        """The same K steps on ONE compute stream (the schedule of rounds 1-4), reported beside the headline; the records of its
        last step against those of the timed region's last step (the same batch): two steps in flight must not change a bit."""
        try:
            set_compute_streams(1)
            run_pipelined(4 * len(models))
            sync()
            t0 = time.perf_counter()
            rec1 = run_pipelined(steps)
            sync()
            dt = time.perf_counter() - t0
            out = {"compute_streams": 1, "value": n_rois * steps / dt, "unit": "ROIs/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
                   "note": "every step queued behind the previous one on one HIP stream, every launch sized to fill the chip by itself"}
            if headline_rec is not None and torch.is_tensor(rec1) and rec1.shape == headline_rec.shape:
                # scheduling alone must not change a bit: the last step of the timed region once more on ONE stream, with the kernel
                # choice of the shared chip (StepStreams.shared_min_tiles) — the same kernels, one stream instead of two
                period = 2 * len(models)
                rule = (hip_lib.SPLIT2_MIN_TILES // args.compute_streams if args.compute_streams > 1 else 0) or None   # = StepStreams(n).shared_min_tiles() ...
                with hip_lib.shared_min_tiles_scope(rule, 2048 if args.compute_streams > 2 else None):                   # ... and .shared_min_rows()
                    rec_same = run_pipelined(((steps - 1) % period) + 1)
                    sync()
                out["last_step_records_bit_equal_to_timed_region"] = bool(torch.equal(rec_same, headline_rec.to(rec_same.device)))
            return out
        except Exception as e:  # the headline line must not depend on the extra measurement
            return {"compute_streams": 1, "error": repr(e)}
        finally:
            set_compute_streams(max(1, args.compute_streams))

    def after_warmup():
        if stream is not None:
            for s_ in stream["subs"]:
                s_["sched"].latencies.clear()
        if stream is not None and stream.get("yolox") is not None:      # the warm-up's post-processing is not the timed region's
            torch.cuda.synchronize()
            stream["yolox"].update(events=[], host_s=0.0, images=0, dets=0)
        if stream is not None and args.host_fed:       # copies of the warm-up steps are not the timed region's
            for s_ in stream["subs"]:
                s_["sched"].h2d_timeline(reset=True)
            stream["h2d_bytes_warmup"] = sum(s_["sched"].h2d_bytes for s_ in stream["subs"])

    @torch.no_grad()
    def parity_in_run():
        """Parity INSIDE the driver run (round-4 verdict item 2): the records of the two timed batches under the headline
        arithmetic (three products, fused MLPs, rows hand-over) against the exact six-product form on the very same tensors."""
        m = models[0]
        dR, dt_, n = 0.0, 0.0, 0
        reruns0 = E.range_reruns()
        recs = []
        for k in range(2):
            bt = prepared(m, k)
            r3 = inference_step(m["model"], m["post"], bt).clone()
            with hip_layers.forced_gemm_products(6):
                r6 = inference_step(m["model"], m["post"], bt).clone()
            assert torch.equal(r3[:, 12:], r6[:, 12:])                      # score | obj | roi_id | valid: not arithmetic
            ok = (r3[:, 15] > 0.5)
            dR = max(dR, float((r3[ok, :9] - r6[ok, :9]).abs().max()))
            dt_ = max(dt_, float((r3[ok, 9:12] - r6[ok, 9:12]).abs().max()))
            n += int(ok.sum())
            recs.append(r3)
        return recs, {"compared": "records of the timed batches: headline arithmetic (--gemm-products %d) vs the exact six-product bf16x3 form, same tensors" % args.gemm_products,
                      "max_abs_dR": dR, "max_abs_dt_m": dt_, "n_rois": n, "range_reruns": E.range_reruns() - reruns0,
                      "tolerance": "north_star: R / t within 1e-4"}

    def measure_after(do_cpu):
        out = {"range_check": {"steps_repeated_with_six_products": E.range_reruns(),
                               "layers_kept_on_six_products": len(hip_layers.x3_demoted()),
                               "note": "whole process (warm-up included): a layer whose A rows sat below 2^-4 rms or that overflowed "
                                       "the fp16 range is repeated once and then stays on the bf16x3 kernels"}}
        d_ = sched_streams if stream is not None else dealer["streams"]
        out["two_stream_guard"] = {"launches_outside_this_library": hip_layers.fallback_launches(), "last": hip_layers.last_fallback(),
                                   "dealer_stopped_sharing": d_.stopped_sharing, "allow_foreign": d_.allow_foreign,
                                   "note": "whole process: hip_layers counts every layer that fell back to a PyTorch operator while the HIP path was on; "
                                           "a step that moves the counter inside a sharing dealer is repeated alone and the dealer drops to one stream"}
        if stream is not None:
            lat = np.sort(np.concatenate([np.asarray(s_["sched"].latencies, np.float64) for s_ in stream["subs"]] or [np.zeros(0)])) * 1e3
            out["stream"] = {"images_per_s": None, "rois_per_image_mean": stream["rois_per_image"],
                             "image_latency_ms": ({"p50": float(lat[len(lat) // 2]), "p95": float(lat[int(0.95 * (len(lat) - 1))]), "max": float(lat[-1]),
                                                   "images": int(len(lat)),
                                                   "note": "push() -> the image's records back on the host, timed region only: packing delay (waiting for "
                                                           "enough ROIs to fill a step) + the steps in flight ahead of it + its own step"} if len(lat) else None),
                             "images_pushed": next(stream["counter"]), "rois_per_step": b,
                             "note": "value / rois_per_image_mean = images per second; ROI-granular packing, an image's ROIs may straddle two steps"}
        if do_cpu and not args.graph:
            # what THIS box does with the instruction form the library is built without (round-5 verdict item 2c): the raw probe of
            # tools/probe (every op_sel form of v_pk_add / v_pk_mul_f32 against the scalar instruction) beside the product's
            # three-product GEMM on another stream.  Diagnostic code from tools/, never the product path; skipped when absent.
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import pk_hazard as PK
                lib_ = PK.load_probe(build=False)
                if lib_ is not None:
                    torch.cuda.synchronize()
                    alone = PK.raw_probe_beside(lib_, None, dev, reps=1)
                    beside = PK.raw_probe_beside(lib_, PK.product_gemm_companion(dev), dev, reps=2)
                    out["pk_hazard_probe"] = {"wrong_results": beside["wrong_results"], "wrong_results_alone": alone["wrong_results"],
                                              "lanes": beside["lanes"], "v_pk_add_f32": beside["v_pk_add_f32"], "v_pk_mul_f32": beside["v_pk_mul_f32"],
                                              "note": "raw probe (tools/probe/pk_probe2.hip) beside the product's three-product GEMM on another stream: wrong "
                                                      "packed-fp32 results on this box; the product library holds no such instruction "
                                                      "(tools/check_isa_hazards.py at link time, tests/test_gpu_stream_guard.py)"}
                else:
                    out["pk_hazard_probe"] = {"wrong_results": None, "note": "tools/probe/libpk_probe.so not built"}
            except Exception as e:  # the headline line must not depend on a diagnostic
                out["pk_hazard_probe"] = {"wrong_results": None, "error": repr(e)}
        if stream is not None and stream.get("yolox") is not None:
            y_ = stream["yolox"]
            torch.cuda.synchronize()
            dev_ms = sum(a_.elapsed_time(b_) for a_, b_, _ in y_["events"])
            out["yolox_post"] = {"images": y_["images"], "detections": y_["dets"], "images_per_launch": YOLOX_GROUP,
                                 "ms_per_image": dev_ms / max(y_["images"], 1), "ms_per_step": dev_ms / args.steps,
                                 "host_ms_per_image": 1e3 * y_["host_s"] / max(y_["images"], 1), "host_ms_per_step": 1e3 * y_["host_s"] / args.steps,
                                 "anchors": 8400, "conf_thre": 0.5, "nms_thre": 0.45,
                                 "note": "inside the timed loop: gdrnpp_yolox_postprocess (decode + sort + class-aware NMS) on a seeded YOLOX head "
                                         "output f32[8, 8400, 5 + C] per launch, on its own high-priority stream; ms_per_image = device time of "
                                         "the launch / 8; host_ms = launch + the one read-back of (dets, count) + engine.detections_from_yolox "
                                         "(the hand-off that replaces the reference's detection JSON, dataset_utils.py:146-239)"}
        m = models[0]
        bt, det, K_crop, cfg = m["batches"][0], m["dets"][0], m["K_crops"][0], m["cfg"]

        @torch.no_grad()
        def fwd():
            pb = prepared(m, 0)
            return m["model"](pb["roi_img"], roi_classes=pb["roi_cls"], roi_cams=pb["roi_cam"], roi_whs=pb["roi_wh"],
                              roi_centers=pb["roi_center"], resize_ratios=pb["resize_ratio"],
                              roi_coord_2d=pb["roi_coord_2d"], roi_extents=pb["roi_extent"])

        def timed(fn, n=5):
            fn()
            torch.cuda.synchronize()
            s = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - s) / n * 1e3

        o = fwd()
        with torch.no_grad():
            out["stages_ms"] = {"forward": timed(fwd), "post_processing": timed(lambda: m["post"].process(bt, o, bt["roi_id"]))}
        if args.mlp_gemm == "split" and not args.no_hip_layers and upnp is None:
            try:
                _, out["parity_in_run"] = parity_in_run()
            except Exception as e:   # the headline line must not depend on the extra measurement
                out["parity_in_run"] = {"error": repr(e)}
        if stream is not None and args.host_fed:
            # (a) the copies of the timed steps, device-side; (b) the same steps once more from the HBM-resident pool, same box, same process
            subs = stream["subs"]
            tl = E.h2d_overlap([c for s_ in subs for c in s_["sched"]._h2d_timing], [t for s_ in subs for t in s_["sched"]._step_timing],
                               detail=bool(os.environ.get("GDRNPP_H2D_DEBUG")))
            if "steps_ms" in tl:
                sys.stderr.write("H2D_TIMELINE " + json.dumps({"steps_ms": tl.pop("steps_ms")[:8], "copies_ms": tl.pop("copies_ms")[:70]}) + "\n")
            h2d_ms = tl["h2d_ms"]
            h2d_bytes = sum(s_["sched"].h2d_bytes for s_ in subs) - stream["h2d_bytes_warmup"]
            for s_ in subs:
                s_["sched"].flush()
            subs2 = [dict(sched=s_["make_sched"](timed=False), feeder=s_["make_feeder"](s_["pool"])) for s_ in subs]
            pend2, k2, depth2 = [], 0, max(1, args.compute_streams)

            def run2(n):
                nonlocal k2
                for _ in range(n):
                    s2 = subs2[k2 % len(subs2)]
                    k2 += 1
                    pend2.append(s2["sched"].launch_next(s2["feeder"]))
                    if len(pend2) > depth2:
                        pend2.pop(0)()
            run2(max(args.warmup, 3) * len(subs2))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run2(args.steps)
            while pend2:
                pend2.pop(0)()
            torch.cuda.synchronize()
            ms_res = (time.perf_counter() - t0) / args.steps * 1e3
            for s2 in subs2:
                s2["sched"].flush()
            out["host_fed"] = {"h2d_ms_per_step": h2d_ms / args.steps, "h2d_overlapped_frac": tl["overlapped_frac"],
                               "h2d_overlapped_ms_per_step": tl["overlapped_ms"] / args.steps, "h2d_bytes_per_step": h2d_bytes / args.steps,
                               "h2d_gbs": (h2d_bytes / 1e9) / (h2d_ms * 1e-3) if h2d_ms > 0 else None,
                               "resident_pool_ms_per_step": ms_res, "resident_pool_rois_per_s": b * 1000.0 / ms_res,
                               "note": "h2d_* = device-side duration of the hipMemcpyAsync copies of the TIMED steps (copy-stream events; "
                                       "full 480x640 u8 image + f32 depth per pushed image, pinned host memory); h2d_overlapped_frac = the share "
                                       "of that copy time during which a step's kernels were executing on the compute stream (events of both "
                                       "streams on one clock); resident_pool_* = the same "
                                       "number of steps run again right after from the HBM-resident pool; the line's ms_per_step vs "
                                       "resident_pool_ms_per_step is what the host feed costs end to end"}

        if not args.no_roofline_pass and not args.graph:
            # per-launch HIP events on the launch stream over the same steps, outside the timed region
            gemm_timer = hip_lib.LaunchTimer()
            ev_pairs = []
            hip_lib.set_launch_timer(gemm_timer)
            hip_lib.set_refine_event_sink(ev_pairs if refine else None)
            for i in range(args.steps):
                step(i)
            hip_lib.set_launch_timer(None)
            hip_lib.set_refine_event_sink(None)
            torch.cuda.synchronize()
            roofline, refine_roofline = None, None
            hbm_records = [r for r in gemm_timer.records if r[0].startswith("hbm:")]
            gemm_records = [r for r in gemm_timer.records if not r[0].startswith("hbm:")]
            hbm_rooflines = []
            for kind in sorted({r[0] for r in hbm_records}):   # memory-bound network kernels: algorithmic bytes / event time
                rs = [r for r in hbm_records if r[0] == kind]
                t_k = sum(r[2].elapsed_time(r[3]) for r in rs)
                gbs = sum(r[4] for r in rs) / (t_k * 1e-3) / 1e9
                entry = dict(kernel=kind[4:], bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=gbs / HBM_PEAK_GBS, traffic=None, launch_ms=t_k / len(rs),
                             launches_per_step=len(rs) / args.steps, ms_per_step=t_k / args.steps,
                             bytes_per_launch=sum(r[4] for r in rs) / len(rs))
                if kind == "hbm:dwconv7_ln":
                    # 49 taps x 2 flops per element on the vector ALUs (+ ~10 for the LayerNorm): the kernel is bound by its fp32 FMA
                    # stream, not by the bytes it moves (profiles/r03_dwconv_dissection.txt) — report it against the VALU peak
                    fl = sum(r[4] for r in rs) / 8.0 * (2.0 * 49.0 + 10.0)
                    tf = fl / (t_k * 1e-3) / 1e12
                    entry.update(bound="valu", achieved=tf, peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / F32_MFMA_PEAK_TFLOPS,
                                 flops_per_launch=fl / len(rs), hbm_gbs=gbs, hbm_frac=gbs / HBM_PEAK_GBS,
                                 note="fp32 vector peak 157.3 TFLOP/s (MI355X_MICROARCH.md); hbm_* = the same launches against the HBM roofline")
                hbm_rooflines.append(entry)
            if gemm_records:
                fl = sum(r[1] for r in gemm_records)
                ms_all = sum(r[2].elapsed_time(r[3]) for r in gemm_records)
                n_l = len(gemm_records)
                prods = lambda r: 3.0 if r[0].endswith(hip_lib.X3) else 6.0  # noqa: E731  (MFMA flops per fp32-equivalent flop)
                mfma_fl = sum(prods(r) * r[1] for r in gemm_records)
                bf16_tflops = mfma_fl / (ms_all * 1e-3) / 1e12
                by_kind = {}
                for kind in sorted({r[0] for r in gemm_records}):
                    rs = [r for r in gemm_records if r[0] == kind]
                    t_k = sum(r[2].elapsed_time(r[3]) for r in rs)
                    by_kind[kind] = dict(launches_per_step=len(rs) / args.steps, ms_per_step=t_k / args.steps,
                                         fp32_equiv_tflops=sum(r[1] for r in rs) / (t_k * 1e-3) / 1e12)
                # the same events by launch shape (kind + algorithmic bytes identify a layer shape): in-situ time per launch
                by_shape = []
                for key in sorted({(r[0], int(r[4]), int(r[1])) for r in gemm_records}):
                    rs = [r for r in gemm_records if (r[0], int(r[4]), int(r[1])) == key]
                    t_k = sum(r[2].elapsed_time(r[3]) for r in rs)
                    by_shape.append(dict(kind=key[0], algorithmic_mb=round(key[1] / 1e6, 1), gflop_fp32=round(key[2] / 1e9, 1),
                                         launches_per_step=len(rs) / args.steps, us_per_launch=round(t_k / len(rs) * 1e3, 1)))
                g_traffic = None  # HBM-side bytes per launch from the committed PMC passes (same workload only)
                pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                if os.path.exists(pmc) and b == 128 and wname == "refine" and args.mlp_gemm == "split":
                    g_traffic = json.load(open(pmc)).get("gemm_split_kernel" if args.gemm_products == 6 else "gemm_split_kernel_x3",
                                                         {}).get("traffic_bytes_per_launch")
                roofline = dict(kernel="gemm_split2_pipe_kernel + gemm_split_*_kernel (all split-GEMM launches)" if any(r[0].endswith(hip_lib.X3) for r in gemm_records) else "gemm_split_kernel", bound="mfma", achieved=bf16_tflops, peak=BF16_MFMA_PEAK_TFLOPS,
                                unit="TFLOP/s", frac=bf16_tflops / BF16_MFMA_PEAK_TFLOPS, traffic=g_traffic,
                                traffic_source=None if g_traffic is None else PMC_SOURCE,
                                algorithmic_bytes_per_launch=sum(r[4] for r in gemm_records) / n_l,
                                launch_ms=ms_all / n_l, launches_per_step=n_l / args.steps, ms_per_step=ms_all / args.steps,
                                flops_per_launch=mfma_fl / n_l, fp32_equiv_tflops=fl / (ms_all * 1e-3) / 1e12,
                                fp32_mfma_peak_tflops=F32_MFMA_PEAK_TFLOPS, by_kind=by_kind, by_shape=by_shape,
                                measured_in="separate event pass of the same steps after the timed region, ONE step at a time (each launch has the chip to itself; "
                                            "in the timed region two steps share it and a launch takes longer while the step takes less: whole_step below)",
                                note="bf16 MFMA flops executed = 6 x fp32-equivalent flops (exact 3-way operand split, six "
                                     "partial products, fp32 accumulate)" if args.gemm_products == 6 else
                                     "MFMA flops executed = 3 x fp32-equivalent flops in the *_x3 kinds (two-way fp16 operand split, "
                                     "three partial products, fp32 accumulate; same 2.5 PFLOP/s dense peak), 6 x in the others")
            if ev_pairs:
                ms = [a.elapsed_time(bb) for a, bb in ev_pairs]
                mean_ms = float(np.mean(ms))
                nv, nf = len(m["verts"][0]), len(m["faces"][0])
                bytes_launch, per_roi = algorithmic_bytes_refine(b, cfg.TEST.DEPTH_REFINE_ITER, nv, nf)
                achieved = bytes_launch / (mean_ms * 1e-3) / 1e9
                traffic = None
                pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                if os.path.exists(pmc) and b == 128 and args.subdiv == 4 and wname == "refine":
                    traffic = json.load(open(pmc)).get("traffic_bytes_per_launch")
                refine_roofline = dict(kernel=hip_lib.refine_kernel_name(), bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS,
                                       unit="GB/s", frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                                       traffic_source=None if traffic is None else PMC_SOURCE, launch_ms=mean_ms,
                                       bytes_per_launch=bytes_launch, bytes_per_roi=per_roi, rois_per_launch=b)
            out["roofline"] = roofline or refine_roofline
            out["roofline_other_kernels"] = ([refine_roofline] if (refine_roofline and roofline) else []) + hbm_rooflines
            if upnp is not None:      # configs[0]: the uncertainty-PnP launch of the step against the HBM roofline (it is latency / fp64-VALU bound)
                u = upnp[0]
                ms_u = timed(lambda: hip_lib.uncertainty_pnp_batched(u["p2"], u["p3"], u["w"], u["K"], u["init"]), n=20)
                by = b * (9 * (16 + 24 + 24) + 72 + 48 + 48)
                out["roofline_other_kernels"].append(dict(
                    kernel="upnp_kernel (gdrnpp_uncertainty_pnp_batched, pn = 9)", bound="latency / fp64 valu", achieved=by / (ms_u * 1e-3) / 1e9,
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=by / (ms_u * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None, launch_ms=ms_u, bytes_per_launch=by,
                    problems_per_launch=b, note="one wave per problem, ~10 dependent fp64 LM iterations; host-timed launch incl. the wrapper"))

        if do_cpu and refine and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            pad_v = max(len(v) for v in m["verts"])
            pad_f = max(len(f) for f in m["faces"])
            vv = np.zeros((len(m["verts"]), pad_v, 3), np.float32)
            ff = np.zeros((len(m["faces"]), pad_f, 3), np.int32)
            for k_, (v, f) in enumerate(zip(m["verts"], m["faces"])):
                vv[k_, :len(v)], ff[k_, :len(f)] = v, f
            with tempfile.TemporaryDirectory() as tmp:
                path = os.path.join(tmp, "batch.npz")
                np.savez(path, roi_cls=det["roi_cls"], K_crop=K_crop, roi_depth=bt["roi_depth"].cpu().numpy(),
                         iters=cfg.TEST.DEPTH_REFINE_ITER, thr=cfg.TEST.DEPTH_REFINE_THRESHOLD, verts=vv, faces=ff,
                         n_verts=np.array([len(v) for v in m["verts"]]), n_faces=np.array([len(f) for f in m["faces"]]),
                         coord2d=bt["roi_coord_2d"].cpu().numpy(), extent=det["roi_extent"],
                         **{k_: o[k_].detach().cpu().numpy() for k_ in ("mask", "coor_x", "coor_y", "coor_z", "rot", "trans")})
                with torch.no_grad():
                    rec_gpu = m["post"].process(bt, o, bt["roi_id"]).cpu().numpy()      # the GPU records of the very tensors the child gets
                r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--inputs", path, "--seconds", str(args.cpu_seconds),
                                    "--parity-sample", "16", "--forward-cfg", cfg_names[0], "--forward-seconds", str(max(3.0, args.cpu_seconds / 4))],
                                   cwd=ROOT, capture_output=True, text=True)
            if r.returncode == 0:
                out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
                ps = out["cpu_baseline"].pop("refine_parity_sample", None)
                if ps is not None and isinstance(out.get("parity_in_run"), dict):
                    idx = np.asarray(ps["idx"], int)
                    d = np.abs(rec_gpu[idx, 9:12].astype(np.float64) - np.asarray(ps["t"], np.float64))
                    out["parity_in_run"]["refine_vs_oracle"] = {
                        "n_rois": int(len(idx)), "max_abs_dt_m": float(d.max()), "roi_index": idx.tolist(), "oracle": ps["oracle"],
                        "tolerance_m": 1e-5, "compared": "t of gdrnpp_refine_to_records vs the CPU oracle on the same maps / depth / pose, "
                                                         "computed in this run's cpu_baseline child"}
                out["cpu_baseline"]["note"] = (
                    "value = the depth-refine stage ONLY (row a8, oracle port, 1 thread): the stage the reference runs on the host. "
                    "The network forward - 99.8 % of the GPU step - runs on the GPU in the reference too; stages.forward_cpu_torch is the "
                    "same network with PyTorch's CPU operators on all host cores (the like-for-like CPU figure of that stage); "
                    "value is not comparable with the line's ROIs/s, see stages for the other CPU-side ops")
            else:
                out["cpu_baseline"] = dict(value=None, error=r.stderr[-400:])
        elif do_cpu and upnp is not None and not args.no_cpu_baseline:
            # BASELINE configs[0] ("CPU uncertainty-PnP only"): ResNet-34 forward with PyTorch's CPU operators + uncertainty-PnP (pn = 9)
            # on the host cores, SURVEY.md §8(d)(iii)
            torch.cuda.synchronize()
            r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--inputs", os.devnull, "--upnp-only", "--forward-cfg", cfg_names[0],
                                "--seconds", str(args.cpu_seconds)], cwd=ROOT, capture_output=True, text=True)
            out["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else dict(value=None, error=r.stderr[-400:])
        return out

    return dict(step=step, launch=launch, measure_after=measure_after, other_mode_line=other_mode_line, after_warmup=after_warmup,
                single_stream_line=single_stream_line if stream is None else None, set_compute_streams=set_compute_streams,
                overlap_probe=(lambda: (sched_streams if stream is not None else dealer["streams"]).overlap_probe),
                compute_streams=len(dealer["streams"].streams) if stream is None else max(1, args.compute_streams))


if __name__ == "__main__":
    main()
