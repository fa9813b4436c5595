set -u
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in product he_r8 he_r12; do
  if [ "$v" == "product" ]; then lib=$PWD/tante_amd/lib/libtante_hip.so; else lib=$PWD/tools/_ab/lib_$v.so; fi
  TANTE_LIB=$lib timeout -k 10 100 python tools/head_enc_time.py 2>/dev/null | tee -a gpurun_out/r4_ring.log
done
TANTE_HEAD_TILES=1 timeout -k 10 100 python tools/head_enc_time.py 2>/dev/null | tee -a gpurun_out/r4_ring.log
done
