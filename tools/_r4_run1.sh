set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_round4.py -x -q -m gpu > gpurun_out/r4_t1.log 2>&1; echo "tests rc=$?" >> gpurun_out/r4_t1.log
tail -5 gpurun_out/r4_t1.log
for i in 1 2; do
  for v in 1 0; do
    TANTE_HEAD_ENC=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HEAD_ENC=$v', 'frames/s', d['value'], 'ms', d['ms_per_step'], 'block us', d['roofline']['avg_launch_us'])" | tee -a gpurun_out/r4_ab1.log
  done
done
timeout -k 10 300 python tools/rollout_aten_profile.py > gpurun_out/r4_aten_profile.log 2>&1
tail -40 gpurun_out/r4_aten_profile.log
