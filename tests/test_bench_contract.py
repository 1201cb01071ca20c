"""bench.py contract pieces that can be checked without a GPU: the algorithmic-byte model of SURVEY.md §8(d) and the
shape of the JSON line (validated on the committed result of the last GPU run, profiles/*.json)."""
import glob
import importlib.util
import json
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_of_the_refine_kernel():
    b = _bench()
    total, per_roi = b.algorithmic_bytes_refine(128, 2, 2562, 5120)
    assert per_roi == 65536 + 65536 + 84 + 2 * (12 * 2562 + 12 * 5120) + 12 == 315536
    assert total == 128 * per_roi == 40388608


def test_committed_bench_lines_follow_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_refine_b128.json")))
    assert files, "no committed bench line under profiles/"
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, (f, k)
        assert d["unit"] == "ROIs/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
        assert d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
        r = d["roofline"]
        assert (r["bound"], r["unit"]) in (("hbm", "GB/s"), ("mfma", "TFLOP/s"))
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1 and "traffic" in r
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
        if f.split("/")[-1].startswith("r01"):
            continue
        st = c["stages"]                                  # round 2 on: one entry per stage, 1 thread and all cores
        assert st["refine_1thread"]["cores"] == 1 and st["refine_allcores"]["cores"] == c["host_cores_available"]
        assert st["refine_allcores"]["value"] > st["refine_1thread"]["value"]
        assert {"upnp_pn9_1thread", "upnp_pn4096_1thread", "fps_reference_1thread", "nnd_reference_1thread"} <= set(st)
        assert any("cv2 unavailable" == v.get("note") for v in st.values())
        assert d["config"]["timed_entry_point"].startswith("engine.inference_step")
        assert abs(d["value"] - d["n_gpus"] * d["config"]["rois_per_gpu"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]


def test_round5_lines_carry_in_run_parity_and_live_traffic():
    """Round 5: the driver line proves more by itself — `parity_in_run` (three- vs six-product records of the timed batches and the
    refine stage against the oracle, both inside the north_star tolerances), `roofline.traffic` measured by rocprofv3 in the run,
    the CPU forward leg, and (host-fed stream lines) the copy / compute timeline."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05*_bench_refine_b128.json")))
    assert files, "no committed round-5 bench line under profiles/"
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d["parity_in_run"]
        assert p["n_rois"] == 2 * d["config"]["rois_per_gpu"] and p["max_abs_dR"] <= 1e-4 and p["max_abs_dt_m"] <= 1e-4 and p["range_reruns"] == 0
        o = p["refine_vs_oracle"]
        assert o["n_rois"] == 16 and o["max_abs_dt_m"] <= o["tolerance_m"] == 1e-5
        r = d["roofline"]
        assert r["traffic_source"].startswith("measured in this run") and r["traffic_detail"]["launches_fetch_pass"] > 0
        assert 1.0 < r["traffic_over_algorithmic"] < 1.6
        fwd = d["cpu_baseline"]["stages"]["forward_cpu_torch"]
        assert fwd["unit"] == "ROIs/s" and fwd["value"] > 0 and fwd["cores"] >= 1
        if "compute_streams" in d["config"]:       # from r05t on: two steps in flight (engine.StepStreams), the one-stream schedule beside it
            one = d["single_stream_mode"]
            assert d["config"]["compute_streams"] == 2 and one["compute_streams"] == 1 and one["steps"] == d["steps"]
            assert 1.0 < d["value"] / one["value"] < 1.15
            assert one.get("last_step_records_bit_equal_to_timed_region", True) is True     # (r05v on: compared in the run)
            ws = r["whole_step"]
            assert abs(ws["mfma_tflops"] - r["flops_per_launch"] * r["launches_per_step"] / (d["ms_per_step"] * 1e-3) / 1e12) < 1e-6 * ws["mfma_tflops"]
            assert ws["frac_of_peak"] < r["frac"] < 1
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r05*_bench_*stream_hostfed.json"))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        h = d["host_fed"]
        assert d["config"]["host_fed"] is True and h["h2d_ms_per_step"] > 0 and h["h2d_bytes_per_step"] > 1e6
        assert d["value"] >= 0.95 * h["resident_pool_rois_per_s"]          # verdict r4 item 3: >= 0.95 of the resident-pool rate
        if not os.path.basename(f).startswith(("r05a", "r05b", "r05c")):    # from r05e on the host really is a step ahead: copies under compute
            assert h["h2d_overlapped_frac"] >= 0.5


def test_round6_lines_carry_the_two_stream_guard_the_hazard_probe_and_the_detector_leg():
    """Round 6: the driver line says by itself that nothing outside this library ran beside the MFMAs (`two_stream_guard`), what the
    raw packed-fp32 probe does on the box of the run (`pk_hazard_probe`), how a host core fares on the ROI crops (`crop_resize_port`
    stage); the configs[4] stream line starts from the detector's raw output (`yolox_post`); the small-batch lines hold four
    hipGraphs in flight."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06*_bench_refine_b128.json")))
    assert files, "no committed round-6 bench line under profiles/"
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        g = d["two_stream_guard"]
        assert g["launches_outside_this_library"] == 0 and g["dealer_stopped_sharing"] is None and g["allow_foreign"] is False
        assert d["config"]["compute_streams"] == 2 and d["single_stream_mode"]["last_step_records_bit_equal_to_timed_region"] is True
        pk = d["pk_hazard_probe"]
        assert pk["wrong_results_alone"] == 0 and pk["wrong_results"] is not None
        if pk["wrong_results"]:
            assert pk["lanes"][0] >= 48 and all("op_sel:[0,1]" in x for x in pk["v_pk_add_f32"] + pk["v_pk_mul_f32"])
        p_ = d["parity_in_run"]
        assert p_["max_abs_dR"] <= 1e-4 and p_["max_abs_dt_m"] <= 1e-4 and p_["refine_vs_oracle"]["max_abs_dt_m"] <= 1e-5
        crop = d["cpu_baseline"]["stages"]["crop_resize_port_1thread"]
        assert crop["kind"] == "port" and crop["cores"] == 1 and crop["value"] > 0
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r06*_bench_*stream_hostfed_yolox.json"))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        y = d["yolox_post"]
        assert y["images"] > 0 and y["detections"] > 3 * y["images"] and 0 < y["ms_per_image"] < 1.0 and 0 < y["host_ms_per_image"] < 2.0
        assert d["config"]["host_fed"] is True and d["value"] >= 0.95 * d["host_fed"]["resident_pool_rois_per_s"]
        assert d["two_stream_guard"]["dealer_stopped_sharing"] is None
    small = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06*_small_batch.jsonl")))
    assert small
    for f in small:
        rows = [json.loads(l) for l in open(f) if l.startswith("{")]
        by = {(r["config"]["rois_per_gpu"], r["config"]["hipgraph"]): r for r in rows}
        for b in (8, 16):
            assert by[(b, True)]["config"]["compute_streams"] == 4
            assert by[(b, True)]["value"] >= 1.2 * by[(b, False)]["value"]        # round-5 verdict item 3's bar, same box


def test_gpus_flag_spawns_ranks_and_gathers_every_roi_once():
    """`python bench.py --gpus 2` without a launcher: spawn, rendezvous on 127.0.0.1, contiguous ROI shards, one all-gather
    of the records, ROI-id permutation check, MAX-over-ranks timing, one JSON line from rank 0 — on CPU through the gloo
    backend with the GPU step stubbed (the same code path the RCCL run takes)."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub-step",
                        "--steps", "3", "--warmup", "1", "--batch", "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 10 and d["config"]["rois_per_gpu"] == 5
    assert d["scaling"] == "weak" and d["gather_ms"] > 0 and d["config"]["parallelism"] == "roi-shard x2"
    assert abs(d["value"] - 10 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]


def test_eight_ranks_1024_tless_roi_ids_and_the_collective_block():
    """BASELINE configs[3] launch path without hardware: `bench.py --gpus 8` (auto workload = tless, 128 ROIs per rank = 1024 per
    iteration) through gloo with the GPU step stubbed — 8 spawned ranks, contiguous shards, ONE all-gather per step, the
    1024-id permutation check on every rank, and the `collective` block filled from the process group itself
    (dist.get_backend / dist.get_world_size), which is what a hardware SCALE run will prove RCCL's world size with."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--stub-step",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["config"]["workload_key"] == "tless" and d["config"]["baseline_config_index"] == 3
    assert d["config"]["global_batch"] == 1024 and d["config"]["rois_per_gpu"] == 128 and d["scaling"] == "weak"
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world_size_seen"] == 8 and c["op"] == "all_gather_into_tensor"
    assert c["bytes_per_rank"] == 128 * 64 and c["bytes_received_per_rank"] == 7 * 128 * 64 and c["calls_per_step"] == 1
    assert abs(d["value"] - 1024 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]


def test_gather_to_rank0_mirrors_the_main_process_only_write():
    """--gather-to-rank0: dist.gather instead of the all-gather (the reference's evaluate() returns on every rank but the main
    one, gdrn_evaluator.py:581-582): rank 0 holds every ROI id once, the other ranks hold nothing."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub-step", "--gather-to-rank0",
                        "--steps", "2", "--warmup", "1", "--batch", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["collective"]["op"] == "gather(dst=0)" and d["collective"]["world_size_seen"] == 2 and d["config"]["global_batch"] == 14
    assert d["config"]["collective"].startswith("gather(dst=0)")


def test_force_dist_runs_the_collective_with_one_rank():
    """--force-dist: a one-rank process group (what a 1-GPU box can show of the N > 1 path: profiles/r05k_*): the records go through
    the collective, the line carries the collective block."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--backend", "gloo", "--stub-step", "--steps", "2",
                        "--warmup", "1", "--batch", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["collective"]["world_size_seen"] == 1 and d["collective"]["backend"] == "gloo" and d["gather_ms"] > 0
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r05[ku]_bench_rccl_one_rank_*.json"))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        assert d["collective"]["backend"] == "nccl" and d["collective"]["rank0_device"] == "cuda:0" and d["product_path"] is True


def test_world_size_must_match_gpus_flag():
    import subprocess
    import sys

    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-step"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_steps_resolved_two_launches_late_still_gather_every_step_once():
    """The resolve queue of the two-stream schedule (a step's records are gathered after the next TWO have been launched) on the CPU
    launch path: K steps launched, K gathered, the last gathered block holds every ROI id once on both ranks."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub-step", "--compute-streams", "2",
                        "--steps", "5", "--warmup", "1", "--batch", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["config"]["global_batch"] == 12 and d["collective"]["calls_per_step"] == 1
    assert abs(d["value"] - 12 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
