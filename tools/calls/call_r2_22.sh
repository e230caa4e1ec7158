source tools/gpu_step.sh
step 900 gpurun_out/pytest22.log python -m pytest tests/test_optimizer.py tests/test_rendezvous.py tests/test_cpp_shim.py tests/test_gpu_api.py -q -m gpu
tail -8 gpurun_out/pytest22.log
step 300 gpurun_out/opt_time.txt python tools/opt_time.py
cat gpurun_out/opt_time.txt
