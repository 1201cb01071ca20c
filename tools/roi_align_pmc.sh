#!/bin/bash
# Counters of gdrnpp_roi_align on the ops microbenchmark's workload (separate --pmc passes, kernel-trace only): where do its cycles go?
#   bash tools/roi_align_pmc.sh <out_dir> [variant]
cd /tmp && export TMPDIR=/tmp
OUT=$1; V=${2:-0}
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/ra_run.py <<PY
import sys, os
sys.path.insert(0, "$R")
sys.path.insert(0, "$R/tools")
import numpy as np, torch
from gdrnpp_bop2022_amd import hip_lib, synthetic as S
hip_lib.load()   # (the variant argument selected A/B builds of round 6; the library now holds the kept form only)
dev = torch.device("cuda", 0); rng = np.random.default_rng(20220925); b = 128
verts, faces, ext = S.make_models(21, np.random.default_rng(1), 2)
det = S.make_detections(b, 21, ext, rng)
x = torch.rand(16, 3, 480, 640, device=dev)
rois = torch.from_numpy(np.concatenate([rng.integers(0, 16, (b, 1)), det["roi_center"] - det["scale"][:, None] / 2, det["roi_center"] + det["scale"][:, None] / 2], 1).astype(np.float32)).to(dev)
for _ in range(6): hip_lib.roi_align(x, rois, 256)
torch.cuda.synchronize()
PY
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  D=$(mktemp -d /tmp/ra_pmc_XXXX)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python /tmp/ra_run.py > $D.out 2> $D.err || { echo "pass failed: $C"; tail -3 $D.err; }
  python $R/tools/pmc_parse_any.py $D roi_align_kernel
done > $OUT/roi_align_pmc_v$V.txt 2>&1
