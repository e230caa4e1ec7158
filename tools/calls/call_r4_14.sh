# round 4, call 14: the fixed host-side cost of a 20-step timed region (graph launch call + completion) under the
# runtime's wait / dispatch knobs — each setting in its own process
source tools/gpu_step.sh
run() { echo "=== $*"; env "$@" timeout -k 10 200 python3 tools/region_overhead.py 20 2>&1 | grep "K=20"; }
{
run A=1
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run ROC_ACTIVE_WAIT_TIMEOUT=100000
run ROC_CPU_WAIT_FOR_SIGNAL=1
run ROC_CPU_WAIT_FOR_SIGNAL=0
run ROC_SYSTEM_SCOPE_SIGNAL=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run GPU_MAX_HW_QUEUES=1
run A=1
} > gpurun_out/r4_14_region.txt 2>&1
cat gpurun_out/r4_14_region.txt
