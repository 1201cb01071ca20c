"""Sum rocprofv3 --pmc counters per kernel name substring: pmc_any_kernel.py <dir> <substr>"""
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            key = r["Kernel_Name"][:60]
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
