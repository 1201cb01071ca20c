"""FETCH_SIZE / WRITE_SIZE of the split-GEMM launches of a bench step, grouped by (kernel instance, grid size) so that the
traffic of every GEMM shape can be put next to its algorithmic bytes (markdown on stdout).
usage: pmc_gemm_by_shape.py <fetch_dir> <write_dir>   (units KiB; gfx950: FETCH_SIZE x2 for wide streaming reads)"""
import collections, csv, glob, re, sys


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm_split" not in n or "reduce" in n or r["Counter_Name"] != counter:
                continue
            m = re.search(r"(gemm_split\w*kernel<[^>]*>)", n)
            acc[(m.group(1) if m else n[:60], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
    return acc


f, w = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
print("| kernel | workgroups | launches | fetch MB (x2 corrected) | write MB |")
print("|---|---|---|---|---|")
tot_f = tot_w = n = 0
for k in sorted(f, key=lambda k: -sum(f[k])):
    fm = 2.0 * sum(f[k]) / len(f[k]) * 1024 / 1e6
    wm = sum(w.get(k, [0])) / max(1, len(w.get(k, [0]))) * 1024 / 1e6
    print(f"| `{k[0]}` | {k[1]} | {len(f[k])} | {fm:.1f} | {wm:.1f} |")
    tot_f += 2.0 * sum(f[k]) * 1024 / 1e6
    tot_w += sum(w.get(k, [0])) * 1024 / 1e6
    n += len(f[k])
print(f"\nall {n} launches: fetch {tot_f / n:.1f} MB, write {tot_w / n:.1f} MB per launch")
