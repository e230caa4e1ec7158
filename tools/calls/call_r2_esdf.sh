#!/bin/bash
# ESDF build: parity tests, then kernel times at 200^3 / 400^3
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
source tools/gpu_step.sh
step 600 gpurun_out/esdf_tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu -k "esdf or sdf_map" && \
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esdf_prof -- python3 tools/esdf_time.py > gpurun_out/esdf_time.log 2>&1
tail -3 gpurun_out/esdf_tests.log; cat gpurun_out/esdf_time.log | grep grid
f=$(ls -t gpurun_out/esdf_prof/*/*kernel_stats.csv | head -1); cat $f | cut -c1-200
