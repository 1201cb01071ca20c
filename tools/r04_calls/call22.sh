#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04w
rm -rf $O; mkdir -p $O
cd $R
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p -- python tools/wide_tile_clock.py > /dev/null 2> $O/p.err
python - <<'PY' > $O/wide.txt
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r04w/p"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_split2_pipe" in r["Kernel_Name"]:
            import re
            key=re.search(r"gemm_split2_pipe_kernel<[^>]*>", r["Kernel_Name"]).group(0)
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"]=="GRBM_GUI_ACTIVE": agg[key]["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,d in agg.items():
    m={c:sum(v)/len(v) for c,v in d.items()}
    cyc=m["GRBM_GUI_ACTIVE"]/8
    print(f"{k}: {len(d['dur_us'])} launches  {m['dur_us']:.1f} us  clock {cyc/m['dur_us']/1e3:.2f} GHz  MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc:.3f}  wave-cycles {m['SQ_WAVE_CYCLES']*4/1e6:.1f} M  wait_any {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f}  wait_inst {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}  active {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
PY
rm -rf $O/p
cat $O/wide.txt; tail -2 $O/p.err
