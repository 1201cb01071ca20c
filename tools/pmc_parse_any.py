"""usage: pmc_parse_any.py <dir> <kernel-name-substring> : mean counter values per matching kernel launch."""
import csv, glob, sys, collections
res = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            res[(r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for (c, g), v in sorted(res.items()):
    print(f"{c:32s} grid={g:>10s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
