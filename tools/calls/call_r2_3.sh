source tools/gpu_step.sh
show() { python -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); print('$2', '%.3f us' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], 'value %.3e' % d['value'], 'ms/step %.5f' % d['ms_per_step'], d['config']['launch'])
"; }
i=0
for envs in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "AMD_DIRECT_DISPATCH=0" "GPU_MAX_HW_QUEUES=1"; do
  i=$((i+1))
  step 120 gpurun_out/env_$i.json env $envs python bench.py --no-extras --no-cpu-baseline --steps 1000
  show gpurun_out/env_$i.json "$envs"
  step 120 gpurun_out/env20_$i.json env $envs python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5
  show gpurun_out/env20_$i.json "$envs steps20"
done
step 120 gpurun_out/env_nograph.json python bench.py --no-extras --no-cpu-baseline --steps 1000 --no-graph
show gpurun_out/env_nograph.json "eager"
for envs in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "== launch_floor2 $envs"; env $envs tools/ubench/launch_floor2 | head -3
done
