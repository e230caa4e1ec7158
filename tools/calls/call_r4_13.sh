# round 4, call 13: optimizer tests with the precision-specific switch points, then the bench with its extras (large-batch optimizer)
source tools/gpu_step.sh
step 900 gpurun_out/r4_13_tests.log python3 -m pytest tests/test_optimizer.py tests/test_capi.py tests/test_gpu_fuzz.py -x -q -m gpu
tail -3 gpurun_out/r4_13_tests.log
step 600 gpurun_out/r4_13_bench.log python3 bench.py --gpus 1 --steps 20 --warmup 5
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r4_13_bench.log") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step_gpu"], r["roofline"]["frac"])
print(json.dumps(r["extras"]["optimizer"], indent=1)[:3000])
PY
