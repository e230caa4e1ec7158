# round 4, call 8: A/B on one box — the throughput bodies with the sample's four corner loads issued together in front of
# the velocity / speed arithmetic (NEW) against the compiler's own order (base: two dependent round trips per sample)
source tools/gpu_step.sh
export ROWS="16384,6,f64 4096,6,f64 8192,12,f64 16384,6,f32 65536,6,f64 8192,24,f64" LIBS="base NEW"
bash tools/ab_bisect.sh > gpurun_out/r4_08_ab_200.txt 2>&1
python3 tools/ab_table.py gpurun_out/r4_08_ab_200.txt
export GTOP_GRID=400 ROWS="8192,12,f64 16384,6,f64"
bash tools/ab_bisect.sh > gpurun_out/r4_08_ab_400.txt 2>&1
python3 tools/ab_table.py gpurun_out/r4_08_ab_400.txt
