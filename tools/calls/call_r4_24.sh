# round 4, call 24: what the gather costs at RCCL world size 1: point-to-point stores (push) against RCCL's captured all-gather
source tools/gpu_step.sh 2>/dev/null || true
for steps in 20 200; do
for mode in "GTOP_BENCH_GATHER=push" "GTOP_BENCH_GATHER=library"; do
  env GTOP_BENCH_FORCE_DIST=1 $mode timeout -k 10 300 python3 bench.py --gpus 1 --steps $steps --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4_24_tmp.log 2>&1
  python3 - "$mode" $steps <<'PY'
import json, sys
r = json.loads([l for l in open("gpurun_out/r4_24_tmp.log") if l.startswith("{")][-1])
c = r["collective"]
print(f"{sys.argv[1]:28s} steps {sys.argv[2]:>4s} buckets {r['config']['buckets']}: host {c['elapsed_s_max']*1e6:7.1f} us, device span {c['gpu_elapsed_s_by_rank'][0]*1e6:7.1f} us; "
      f"kernels only host {c['kernels_only_elapsed_s_max']*1e6:7.1f} us; exposed {c['collective_exposed_us']:6.1f} us  [{c['gather_impl']}: {r['config']['gather'][:50]}]")
PY
done; done > gpurun_out/r4_24_modes.txt 2>&1
cat gpurun_out/r4_24_modes.txt
timeout -k 10 300 env GTOP_BENCH_BACKEND=gloo GTOP_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 --batch 1024 > gpurun_out/r4_24_two.log 2>&1
python3 - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r4_24_two.log") if l.startswith("{")][-1])
c = r["collective"]
print("two processes on one card:", r["value"], r["ms_per_step"], c["gather_impl"], c["elapsed_s_by_rank"], c["collective_exposed_us"], r["config"]["gather"])
PY
