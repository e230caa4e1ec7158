// micro-benchmark: duration of an empty kernel for several launch shapes of 1024 wavefronts-worth of work,
// back to back in a 50-node hipGraph (the launch floor under the bench's graph replay)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(double *out) { if (threadIdx.x == 0 && blockIdx.x == 0 && out) out[0] = 1.0; }
int main() {
  double *out; hipMalloc(&out, 1 << 20);
  hipStream_t s; hipStreamCreate(&s);
  const int shapes[][3] = {{1024, 64, 0}, {1024, 64, 10240}, {512, 128, 0}, {256, 256, 0}, {128, 512, 0}, {1024, 192, 0}, {64, 64, 0}, {1, 64, 0}};
  for (auto &sh : shapes) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), sh[2], s, out);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%4d blocks x %3d threads, %5d B LDS: %.3f us per kernel\n", sh[0], sh[1], sh[2], ms * 1e3 / (20 * 50));
  }
  return 0;
}
