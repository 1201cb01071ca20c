"""Per-kernel time inside the last <frac> of a rocprofv3 --kernel-trace run (steady state of a bench without a marker kernel).
usage: tail_window_breakdown.py <trace_dir> [frac=0.3]"""
import collections, csv, glob, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
lo = t1 - (t1 - t0) * frac
acc, cnt = collections.Counter(), collections.Counter()
for r in rows:
    if int(r["Start_Timestamp"]) >= lo:
        acc[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cnt[r["Kernel_Name"]] += 1
tot = sum(acc.values())
print(f"window {(t1 - lo) / 1e6:.2f} ms, kernel time {tot / 1e6:.2f} ms, {sum(cnt.values())} launches")
for k, v in acc.most_common(22):
    print(f"{v / 1e6:8.3f} ms {100 * v / tot:5.1f} % {cnt[k]:5d}  {k[:100]}")
