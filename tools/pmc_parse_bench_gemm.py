"""Mean FETCH_SIZE / WRITE_SIZE per split-GEMM launch (gemm_split*_kernel and the fused mlp_fused_x3_kernel: the launches of bench.py's roofline) from rocprofv3 --pmc passes over bench.py (JSON on stdout).
usage: pmc_parse_bench_gemm.py <fetch_dir> <write_dir>   (units KiB; gfx950: FETCH_SIZE x2 for wide streaming reads)"""
import csv, glob, json, sys

def collect(d, counter):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ("gemm_split" in r["Kernel_Name"] or "mlp_fused_x3" in r["Kernel_Name"]) and "reduce" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals

f, w = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"launches_fetch": len(f), "launches_write": len(w), "fetch_kib_raw_mean": sum(f) / len(f), "write_kib_raw_mean": sum(w) / len(w)}
out["traffic_bytes_per_launch"] = (2.0 * out["fetch_kib_raw_mean"] + out["write_kib_raw_mean"]) * 1024.0
print(json.dumps(out, indent=1))
