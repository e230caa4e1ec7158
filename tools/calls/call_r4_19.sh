# round 4, call 19: the collective path's whole-region graph (forked all-gathers) at RCCL world size 1, and the N = 1 line
source tools/gpu_step.sh
step 900 gpurun_out/r4_19_tests.log python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu
tail -5 gpurun_out/r4_19_tests.log
GTOP_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4_19_dist1.log 2>&1
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r4_19_dist1.log") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["ms_per_step_gpu"], r["config"]["gather"], r["config"]["buckets"])
print(json.dumps(r["collective"])[:1200])
PY
