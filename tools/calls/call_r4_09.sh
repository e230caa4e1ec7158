# round 4, call 9: the launch rule's switch points after the corner records / loads-first changes (samples per lane
# pinned to 3 and 6 beside the auto rule), fp64 and fp32, and the small-batch rule for 7..12 segments
source tools/gpu_step.sh
step 600 gpurun_out/r4_09_spl_f64.txt python3 tools/spl_compare.py 2048,6,f64 3072,6,f64 4096,6,f64 6144,6,f64 8192,6,f64 10240,6,f64 12288,6,f64 16384,6,f64
cat gpurun_out/r4_09_spl_f64.txt
step 600 gpurun_out/r4_09_spl_f32.txt python3 tools/spl_compare.py 2048,6,f32 4096,6,f32 6144,6,f32 8192,6,f32 12288,6,f32 16384,6,f32
cat gpurun_out/r4_09_spl_f32.txt
step 600 gpurun_out/r4_09_spl_m12.txt python3 tools/spl_compare.py 256,12,f64 512,12,f64 1024,12,f64 1536,12,f64 2048,12,f64 256,12,f32 512,12,f32 1024,12,f32
cat gpurun_out/r4_09_spl_m12.txt
