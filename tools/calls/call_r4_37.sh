# round 4, call 37: the bench with its extras (per-pass figure of the optimizer loop)
source tools/gpu_step.sh
step 600 gpurun_out/r4_37_bench.log python3 bench.py --gpus 1 --steps 20 --warmup 5
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r4_37_bench.log") if l.startswith("{")][-1])
o = r["extras"]["optimizer"]
print(r["value"], r["ms_per_step_gpu"], r["roofline"]["frac"])
print({k: o[k] for k in ("seconds", "per_pass_us", "evals_per_s_inside_the_loop", "inside_the_loop_note")})
print(o["large_batch"])
PY
