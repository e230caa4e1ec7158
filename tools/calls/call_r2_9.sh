source tools/gpu_step.sh
step 600 gpurun_out/pytest9.log python -m pytest tests/test_rendezvous.py tests/test_analytic.py tests/test_optimizer.py -q -m gpu
tail -5 gpurun_out/pytest9.log
step 300 gpurun_out/host_api_rate.txt python tools/host_api_rate.py 1 1024
cat gpurun_out/host_api_rate.txt
