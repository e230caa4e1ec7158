# round 4, call 20: the collective path at RCCL world size 1 in its three forms (side-stream gathers / graph per bucket / host-side
# gathers), 20 and 200 steps: host time of the region, device span, what the gathers add
source tools/gpu_step.sh
for steps in 20 200; do
for mode in "GTOP_BENCH_SIDE_STREAM=1" "GTOP_BENCH_SIDE_STREAM=0" "GTOP_BENCH_CAPTURE_GATHER=0"; do
  env GTOP_BENCH_FORCE_DIST=1 $mode timeout -k 10 300 python3 bench.py --gpus 1 --steps $steps --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4_20_tmp.log 2>&1
  python3 - "$mode" $steps <<'PY'
import json, sys
r = json.loads([l for l in open("gpurun_out/r4_20_tmp.log") if l.startswith("{")][-1])
c = r["collective"]
print(f"{sys.argv[1]:28s} steps {sys.argv[2]:>4s} buckets {r['config']['buckets']}: host {c['elapsed_s_max']*1e6:7.1f} us, device span {c['gpu_elapsed_s_by_rank'][0]*1e6:7.1f} us; "
      f"kernels only host {c['kernels_only_elapsed_s_max']*1e6:7.1f} us, device {c['kernels_only_gpu_elapsed_s_by_rank'][0]*1e6:7.1f} us; exposed {c['collective_exposed_us']:6.1f} us  [{r['config']['gather'][:40]}]")
PY
done; done > gpurun_out/r4_20_modes.txt 2>&1
cat gpurun_out/r4_20_modes.txt
