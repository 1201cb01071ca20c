"""Parse rocprofv3 --pmc counter_collection CSVs -> per-kernel mean counter values (JSON on stdout)."""
import csv, glob, json, sys, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = ("stream_read_16B" if "stream_read_kernel<4>" in name else "stream_read_4B" if "stream_read_kernel<1>"
                   in name else "depth_refine_kernel" if "depth_refine_kernel" in name else None)
            if key:
                res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: {c: sum(v) / len(v) for c, v in cs.items()} | {"n": len(next(iter(cs.values())))} for k, cs in res.items()}, indent=1))
