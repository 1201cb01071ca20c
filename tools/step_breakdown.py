"""Per-kernel time of the LAST timed bench step from a rocprofv3 --kernel-trace CSV (markdown on stdout).

usage: step_breakdown.py <dir with *_kernel_trace.csv> [title]
A step is delimited by two consecutive depth_refine_kernel launches (one per step); the last pair that encloses a full
forward is taken (bench.py ends with a few post-processing-only timing launches, which enclose nothing)."""
import collections, csv, glob, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "depth_refine_kernel" in r["Kernel_Name"]]
pair = max(range(len(marks) - 1), key=lambda i: (50 < marks[i + 1] - marks[i] < 400, i))   # last pair with ONE forward in between
lo, hi = marks[pair] + 1, marks[pair + 1] + 1
step = rows[lo:hi]
wall = (int(step[-1]["End_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0.0, 0])
for r in step:
    a = agg[r["Kernel_Name"]]
    a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a[1] += 1
tot = sum(v[0] for v in agg.values())
print(f"# {sys.argv[2] if len(sys.argv) > 2 else 'steady-state step'}\n")
print(f"step wall {wall:.3f} ms, {len(step)} kernels, sum of kernel time {tot:.3f} ms\n")
print("| ms | calls | kernel |\n|---|---|---|")
for k, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if ms >= 0.05:
        print(f"| {ms:.3f} | {n} | `{k[:150]}` |")
refine = [r for r in rows if "depth_refine_kernel" in r["Kernel_Name"]]
print("\ndepth_refine_kernel launches (us):", [f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}" for r in refine])
