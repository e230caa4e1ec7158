# round 3, after the launch rule moved fp64 batches from 12 288 to two trajectories per wavefront: the B = 16 384 fp64 profile again
source tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/kprof_B16384_f64
step 300 gpurun_out/kprof_B16384_f64.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_B16384_f64 -- python3 bench.py --no-extras --no-cpu-baseline --batch 16384 --dtype f64 --steps 500
grep -o '"probe": [0-9.]*' gpurun_out/kprof_B16384_f64.log | head -2
bash tools/pmc_collect.sh B16384_f64 --batch 16384 --dtype f64
