source tools/gpu_step.sh
step 600 gpurun_out/pytest10.log python -m pytest tests/test_optimizer.py tests/test_capi.py -q -m gpu
tail -15 gpurun_out/pytest10.log
