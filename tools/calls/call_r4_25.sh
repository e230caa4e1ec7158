# round 4, call 25: repeatability of the 20-step collective region at RCCL world size 1 (push / library), three runs each
source tools/gpu_step.sh
for rep in 1 2 3; do
for mode in "GTOP_BENCH_GATHER=push" "GTOP_BENCH_GATHER=library"; do
  env GTOP_BENCH_FORCE_DIST=1 $mode timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r4_25_tmp.log 2>&1
  python3 - "$mode" <<'PY'
import json, sys
r = json.loads([l for l in open("gpurun_out/r4_25_tmp.log") if l.startswith("{")][-1])
c = r["collective"]
print(f"{sys.argv[1]:28s} host {c['elapsed_s_max']*1e6:7.1f} us, device span {c['gpu_elapsed_s_by_rank'][0]*1e6:7.1f} us; kernels only host {c['kernels_only_elapsed_s_max']*1e6:7.1f} us device {c['kernels_only_gpu_elapsed_s_by_rank'][0]*1e6:7.1f}; exposed {c['collective_exposed_us']:6.1f} us [{c['gather_impl']}] probe {r['roofline']['launch_us']['probe']:.2f}")
PY
done; done > gpurun_out/r4_25_modes.txt 2>&1
cat gpurun_out/r4_25_modes.txt
