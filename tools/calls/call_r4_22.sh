# round 4, call 22: probe — two processes mapping each other's device buffers (torch CUDA-IPC) and writing into them
source tools/gpu_step.sh
step 300 gpurun_out/r4_22_ipc.log python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 tools/proto/ipc_probe.py
tail -12 gpurun_out/r4_22_ipc.log
