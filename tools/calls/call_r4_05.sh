# round 4, call 5: the fp32 bodies with exact cell selection (positions and cell index in double, set-up in double):
# the four seeds that failed round 3's hunt under the restored all-rows criterion, the fp32 tests, a hunt of 1 000 more
# seeds, the bench line with extras (configs[2] time)
source tools/gpu_step.sh
export GTOP_FUZZ_EXTRA=4000 GTOP_FUZZ_BASE=500000
step 300 gpurun_out/r4_05_four_seeds.log python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "test_random_draw_fp32 and (501702 or 502714 or 504816 or 503657)"
tail -3 gpurun_out/r4_05_four_seeds.log
unset GTOP_FUZZ_EXTRA GTOP_FUZZ_BASE
step 900 gpurun_out/r4_05_tests.log python3 -m pytest tests -x -q -m gpu
tail -4 gpurun_out/r4_05_tests.log
GTOP_FUZZ_EXTRA=4000 GTOP_FUZZ_BASE=500000 timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "fp32" -n 5 > gpurun_out/r4_05_fuzz_f32.log 2>&1; tail -6 gpurun_out/r4_05_fuzz_f32.log
step 400 gpurun_out/r4_05_bench.log python3 bench.py --no-cpu-baseline
grep '^{' gpurun_out/r4_05_bench.log > gpurun_out/r4_05_bench.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4_05_bench.json"))
print("value %.4g ms/step host %.5f gpu %.5f" % (d["value"], d["ms_per_step"], d["ms_per_step_gpu"]), d["roofline"]["frac_by_source"])
for w in d["extras"]["workloads"]:
    print(w["workload"], w.get("us_per_launch"), w.get("roofline", {}).get("frac"), w["parity"])
print(d["extras"].get("optimizer", {}))
PY
