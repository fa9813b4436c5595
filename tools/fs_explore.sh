#!/bin/bash
# One box, one call: the block kernel's A/B of launch forms on the product library and on timing-experiment builds (tools/_ab/lib_<name>.so,
# wrong results on purpose), then the phase stamps of the diagnostic build in both forms.  gpurun -- 'bash tools/fs_explore.sh name ...'
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
echo "== product"; timeout -k 10 120 python tools/fs_ab.py --rounds 5 --iters 30 2>/dev/null
for v in "$@"; do
  echo "== $v"; TANTE_LIB=$R/tools/_ab/lib_$v.so timeout -k 10 120 python tools/fs_ab.py --rounds 5 --iters 30 2>/dev/null
done
echo "== product again"; timeout -k 10 120 python tools/fs_ab.py --rounds 5 --iters 30 2>/dev/null
if [ -f $R/tools/_ab/libtante_ablate.so ]; then
  echo "== stamps L=32 unpaired"; timeout -k 10 120 python tools/fs_stamps.py 32 0 1 2>/dev/null
  echo "== stamps L=32 paired"; timeout -k 10 120 python tools/fs_stamps.py 32 0 2 2>/dev/null
fi
