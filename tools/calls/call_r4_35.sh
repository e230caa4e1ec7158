# round 4, call 35: the per-launch floor inside a graph under the runtime's cache-flush / kernarg knobs (device span of the
# 20-step graph: tools/region_overhead.py)
source tools/gpu_step.sh
run() { echo "=== $*"; env "$@" timeout -k 10 200 python3 tools/region_overhead.py 20 2>&1 | grep "K=20"; }
{
run A=1
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run AMD_OPT_FLUSH=3
run ROC_USE_FGS_KERNARG=0
run ROC_USE_FGS_KERNARG=1
run ROC_SKIP_KERNEL_ARG_COPY=1
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run AMD_DIRECT_DISPATCH=0
run A=1
} > gpurun_out/r4_35_knobs.txt 2>&1
cat gpurun_out/r4_35_knobs.txt
