#!/bin/bash
# Build a variant of the library for same-box A/B runs: recompiles ONE source with extra flags and links it with the product's other
# objects.   tools/build_variant.sh <name> <source.hip> "<extra flags>"   ->  tools/_ab/lib_<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; src=$2; extra=$3
mkdir -p $R/tools/_ab/obj_$name
base=$(basename $src .hip)
eflags=""
case $base in
  block_fused) eflags="-fno-honor-nans -DTANTE_MFMA_SETPRIO -mllvm -amdgpu-sched-strategy=max-ilp";;
  block_sliced) eflags="-fno-honor-nans -DTANTE_MFMA_SETPRIO -DFS_PRIO=1";;
  block_bwd|block_bwd_fs) eflags="-fno-honor-nans -DTANTE_MFMA_SETPRIO -DBT_PRIO";;
  head_fused|head_enc|enc_fused|operators|spectral_dft) eflags="-fno-honor-nans";;
  pointwise) eflags="";;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast $eflags $extra -c $R/tante_amd/csrc/$base.hip -o $R/tools/_ab/obj_$name/$base.o
objs=""
for o in $R/tante_amd/lib/*.o; do
  if [ "$(basename $o)" == "$base.o" ]; then objs="$objs $R/tools/_ab/obj_$name/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_ab/lib_$name.so $objs -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
echo $R/tools/_ab/lib_$name.so
