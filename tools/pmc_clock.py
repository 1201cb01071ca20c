"""Effective shader clock per kernel from one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass: pmc_clock.py <dir> <substr>"""
import csv, glob, sys, collections
cnt = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[r["Dispatch_Id"]] = (r["Kernel_Name"][:50], float(r["Counter_Value"]), r)
dur = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(list)
for d, (k, v, r) in cnt.items():
    if d in dur:
        agg[k].append((v, dur[d]))
    elif "Start_Timestamp" in r:
        agg[k].append((v, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for k, l in agg.items():
    g = sum(a for a, _ in l) / len(l); t = sum(b for _, b in l) / len(l)
    print(f"{k}: {len(l)} launches, GRBM_GUI_ACTIVE {g:.0f}, {t:.1f} us -> {g / t / 1e3:.2f} GHz if one counter, {g / 8 / t / 1e3:.2f} GHz if summed over 8 XCDs")
