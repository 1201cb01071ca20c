import csv, glob, sys, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_split" in n:
            key = "fc1_gelu" if "Li1E" in n or "<1>" in n else "fc2_res"
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in res.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
