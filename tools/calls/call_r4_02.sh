# round 4, call 2: corner records — the whole GPU suite, then the bench line with its extras
source tools/gpu_step.sh
step 1000 gpurun_out/r4_02_tests.log python3 -m pytest tests -x -q -m gpu
tail -5 gpurun_out/r4_02_tests.log
step 400 gpurun_out/r4_02_bench.log python3 bench.py --no-cpu-baseline
grep '^{' gpurun_out/r4_02_bench.log > gpurun_out/r4_02_bench.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4_02_bench.json"))
print("value %.4g ms/step host %.5f gpu %.5f" % (d["value"], d["ms_per_step"], d["ms_per_step_gpu"]), d["roofline"]["launch_us"], d["roofline"]["frac_by_source"])
for w in d["extras"]["workloads"]:
    print(w["workload"], w.get("us_per_launch"), w.get("roofline", {}).get("frac"), w["parity"])
print(d["extras"].get("optimizer", {}).get("seconds"))
PY
