"""Per-launch durations (us) of the GEMM / dwconv kernels in the last full step of a rocprofv3 --kernel-trace csv dir (argv[1]),
in launch order — to compare in-situ launches with isolated timings of the same shapes."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "depth_refine_kernel" in r["Kernel_Name"]]
pair = max(range(len(marks) - 1), key=lambda i: (50 < marks[i + 1] - marks[i] < 400, i))
seq = rows[marks[pair] + 1:marks[pair + 1] + 1]
out = []
for r in seq:
    n = r["Kernel_Name"]
    short = ("fc1" if "pipe_kernel<1" in n else "fc2" if "pipe_kernel<2" in n else "out" if "pipe_kernel<0" in n else
             "conv3x3" if "glds_kernel<0, 1>" in n else "dw" if "dwconv7" in n else "g128" if "gemm_split_kernel<" in n else None)
    if short:
        out.append(f"{short}:{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}")
print(" ".join(out))
