"""Matrix-pipe busy fraction and shader clock of the pipelined GEMM launches of tools/gemm_occupancy_probe.py, grouped by grid
size (256 workgroups = one per CU, 512 = two per CU, ...), from rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
--kernel-trace.  usage: pmc_occupancy_probe.py <dir>"""
import collections, csv, glob, re, sys

cnt = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_split_pipe" in r["Kernel_Name"]:
            d = cnt[r["Dispatch_Id"]]
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["wgs"] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
            d["epi"] = re.search(r"kernel<(\d)", r["Kernel_Name"]).group(1)
dur = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(list)
for k, d in cnt.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and k in dur:
        act = d["GRBM_GUI_ACTIVE"] / 8.0
        acc[(d["epi"], d["wgs"])].append((d["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024.0), act / dur[k], dur[k] / 1e3))
print("epilogue workgroups launches busy clock_GHz us")
for k in sorted(acc):
    v = acc[k]
    print(k[0], k[1], len(v), round(sum(x[0] for x in v) / len(v), 3), round(sum(x[1] for x in v) / len(v), 3), round(sum(x[2] for x in v) / len(v), 1))
