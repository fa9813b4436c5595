set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_round4.py -x -q -m gpu > gpurun_out/r4_t4.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_t4.log
tail -3 gpurun_out/r4_t4.log
for i in 1 2; do
for v in "TANTE_HEAD_TILES=1 TANTE_HEAD_L2_HANDOFF=1" "TANTE_HEAD_TILES=1 TANTE_HEAD_L2_HANDOFF=0" "TANTE_HEAD_TILES=2 TANTE_HEAD_L2_HANDOFF=1"; do
  env $v timeout -k 10 100 python tools/head_enc_time.py 2>/dev/null | sed "s/^/$v /" | tee -a gpurun_out/r4_l2.log
done
done
TANTE_HEAD_TILES=1 timeout -k 10 300 python tools/head_enc_stamps.py > gpurun_out/r4_stamps4.log 2>&1
grep -E "^\s+\[(2[6-9]|3[0-4])\]|un-stamped" gpurun_out/r4_stamps4.log
for i in 1 2; do
  for v in "TANTE_HEAD_TILES=1" "TANTE_HEAD_ENC=0"; do
    env $v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'frames/s', d['value'], 'ms', d['ms_per_step'], 'block us', d['roofline']['avg_launch_us'])" | tee -a gpurun_out/r4_ab4.log
  done
done
