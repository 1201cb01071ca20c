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
        assert abs(d["value"] - d["n_gpus"] * d["config"]["rois_per_gpu"] * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
