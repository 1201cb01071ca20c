#!/bin/bash
# A/B builds of the library: tools/build_variant.sh <name> <file.hip> "<extra hipcc flags>" [<file2.hip> "<flags2>"]
# -> _ab/<name>/libgdrnpp_hip.so = the default objects with the named translation units recompiled with the extra flags.
# Select it with GDRNPP_HIP_LIB=_ab/<name>/libgdrnpp_hip.so (hip_lib.LIB_PATH).  _ab/ is git-ignored and travels with gpurun.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/gdrnpp_bop2022_amd/csrc
name=$1; shift
mkdir -p $R/_ab/$name
make -C $C -j16 > /dev/null
objs=""
declare -A over
while [ $# -gt 0 ]; do
  f=$1; flags=$2; shift 2
  base=${f%.hip}
  extra=""
  case $base in gemm_split_pipe|gemm_split2_pipe|gemm_mlp_fused) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value \
      $extra $flags -c $C/$f -o $R/_ab/$name/$base.o
  over[$base]=1
done
for o in $C/*.o; do
  b=$(basename $o .o)
  if [ -n "${over[$b]}" ]; then objs="$objs $R/_ab/$name/$b.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/_ab/$name/libgdrnpp_hip.so
echo "built _ab/$name/libgdrnpp_hip.so"
