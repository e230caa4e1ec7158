// micro-benchmark: duration of a (nearly) empty kernel at the bench's launch shape, back to back in a hipGraph
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(double *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && out) out[0] = 1.0; }
__global__ void k_load_store(const double *in, double *out, int n) {
  extern __shared__ double sm[];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += blockDim.x) sm[i] = in[(size_t)b * n + i];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[(size_t)b * n + i] = sm[n - 1 - i] + 1.0;
}
int main() {
  double *in, *out; hipMalloc(&in, 1 << 24); hipMalloc(&out, 1 << 24); hipMemset(in, 0, 1 << 24);
  hipStream_t s; hipStreamCreate(&s);
  for (int mode = 0; mode < 2; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 50; ++i) {
      if (mode == 0) k_empty<<<1024, 192, 0, s>>>(out);
      else k_load_store<<<1024, 192, 45 * 8, s>>>(in, out, 45);
    }
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f us per kernel (1024 blocks x 192 threads, 50-node graph)\n", mode ? "load->LDS->store" : "empty", ms * 1e3 / (20 * 50));
  }
  return 0;
}
