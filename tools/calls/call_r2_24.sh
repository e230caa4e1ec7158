source tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
prof() {  # tag, bench args...
  local tag=$1; shift
  rm -rf gpurun_out/kprof_$tag
  step 300 gpurun_out/kprof_$tag.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$tag -- python3 bench.py --no-extras --no-cpu-baseline "$@"
  grep -o '"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' gpurun_out/kprof_$tag.log | tr '\n' ' '; echo
}
prof default_B1024_f64 --steps 2000
bash tools/pmc_collect.sh B1024_f64
step 400 gpurun_out/bench_default.json python bench.py --steps 20 --warmup 5
tail -1 gpurun_out/bench_default.json | cut -c1-300
step 400 gpurun_out/bench_2000.json python bench.py --steps 2000 --no-cpu-baseline
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/bench_2000.json") if l.startswith("{")][-1])
print("K=2000 value %.4g  us/step %.3f  roofline" % (r["value"], r["ms_per_step"] * 1e3), r["roofline"]["frac"], r["roofline"].get("avg_launch_us"))
for w in r["extras"]["workloads"]:
    print("  ", w["workload"][:60], "%.2f us" % w["us_per_launch"], "frac %.3f" % w["roofline_frac"], w.get("parity", {}).get("ok"))
print("  optimizer", r["extras"]["optimizer"]["seconds"], "esdf_build_s", r.get("esdf_build_s"))
PY
