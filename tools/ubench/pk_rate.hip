// micro-benchmark: issue rate of scalar f32 fma vs packed f32 fma vs f64 fma on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float *out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a3}, p5 = {a5, a7}, p6 = {a0, a2}, p7 = {a4, a6};
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  const float c = 1.0001f, e = 0.5f;
  const float2v cc = {c, c}, ee = {e, e};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0 = a0 * c + e; a1 = a1 * c + e; a2 = a2 * c + e; a3 = a3 * c + e; a4 = a4 * c + e; a5 = a5 * c + e; a6 = a6 * c + e; a7 = a7 * c + e; }
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { p0 = p0 * cc + ee; p1 = p1 * cc + ee; p2 = p2 * cc + ee; p3 = p3 * cc + ee; p4 = p4 * cc + ee; p5 = p5 * cc + ee; p6 = p6 * cc + ee; p7 = p7 * cc + ee; }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) { d0 = d0 * 1.0001 + 0.5; d1 = d1 * 1.0001 + 0.5; d2 = d2 * 1.0001 + 0.5; d3 = d3 * 1.0001 + 0.5; d4 = d4 * 1.0001 + 0.5; d5 = d5 * 1.0001 + 0.5; d6 = d6 * 1.0001 + 0.5; d7 = d7 * 1.0001 + 0.5; }
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, int wavesPerSimd) {
  float *out; hipMalloc(&out, 1 << 26);
  const int iters = 2000, blocks = 256 * 4 * wavesPerSimd / 4, threads = 256;  // 4 waves per block -> 1 per SIMD per block
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = (double)iters * 64;   // 64 fma instructions per iteration per wave
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-10s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wavesPerSimd, ms, cyc / (instr_per_wave * wavesPerSimd));
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>("f32 fma", w); run<1>("pk f32 fma", w); run<2>("f64 fma", w); }
  return 0;
}
