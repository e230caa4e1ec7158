# round 3: PMC passes for the two remaining bench workloads, the variant timings, the post-processing kernels
source tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/pmc_collect.sh B16384_f64 --batch 16384 --dtype f64
bash tools/pmc_collect.sh B8192_m12_g400_f64 --batch 8192 --segments 12 --grid 400 --density 0.04
step 300 gpurun_out/r3_variant_times.txt python3 tools/variant_times.py
rm -rf gpurun_out/f4prof
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f4prof -- python3 tools/f4_time.py > gpurun_out/f4.log 2>&1
grep "edt_query\|trajectory_stats" gpurun_out/f4.log
