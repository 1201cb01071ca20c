"""Matrix-pipe utilisation of the split-GEMM kernels from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass over
bench.py (JSON on stdout): per kernel variant the launch count and mean busy / active ratio.
usage: pmc_mfma_busy.py <dir>
SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs (32 cycles per 32x32x16 bf16 MFMA, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE
is summed over the 8 XCDs (checked: value / 8 / kernel time = the shader clock) — busy fraction = busy / (active / 8 * 1024)."""
import collections, csv, glob, json, re, sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_split" not in n or "reduce" in n:
            continue
        m = re.search(r"(gemm_split\w*kernel<[^>]*>)", n)
        acc[m.group(1) if m else n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    busy, act = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", []), cs.get("GRBM_GUI_ACTIVE", [])
    if busy and act and len(busy) == len(act):
        out[k] = dict(launches=len(busy), mfma_busy_frac=sum(b / (a / 8.0 * 1024.0) for b, a in zip(busy, act)) / len(busy),
                      mean_active_cycles_per_xcd=sum(act) / len(act) / 8.0)
print(json.dumps(out, indent=1))
