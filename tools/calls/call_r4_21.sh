# round 4, call 21: the collective path after the bucket rule change (in-line captured gathers: one bucket for a short run)
source tools/gpu_step.sh
step 900 gpurun_out/r4_21_tests.log python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu
tail -5 gpurun_out/r4_21_tests.log
GTOP_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4_21_dist1.log 2>&1
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r4_21_dist1.log") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["ms_per_step_gpu"], r["config"]["gather"], r["config"]["buckets"], r["collective"]["collective_exposed_us"])
PY
