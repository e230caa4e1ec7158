// gtop_push.hip — collecting the ranks' results without a collective launch: the all-gather as point-to-point stores.
//
// the batch shards over the GPUs with no data-path exchange; the only communication is the all-gather of
// the per-trajectory costs (and optionally gradients) "so that every rank holds all costs".  Each rank's rows are
// written by that rank alone, into a slot that is its own in every rank's buffer — so the gather needs no protocol:
// ONE kernel on the producing GPU stores its rows into every destination (its own buffer and the peers', mapped into
// this process: hipIpcOpenMemHandle across processes, peer access within one), each destination over its own xGMI link,
// 16 bytes per lane, whole 128-byte lines per 8 lanes.  A library all-gather of a few hundred KB costs a launch, a
// ring's worth of synchronisation steps and a completion handshake; this costs one small kernel behind the last
// evaluation of the bucket.  When the destinations' owners may read: after any synchronisation that orders their read
// behind this kernel's completion (the closing barrier of a timed region; an event the owner waits on).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gtop_kernels.h"

namespace {

typedef unsigned int gtop_u4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) push_rows_kernel(const unsigned char *__restrict__ src, size_t bytes, GtopPushDsts d,
                                                        unsigned long long *minmax) {
  // optionally the device-clock stamp of gtop_device_clock_stamp, taken as the kernel starts: "behind the last
  // evaluation, in front of the gather" without a kernel of its own (a node of a graph costs its 1.7 us floor)
  if (minmax && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const unsigned long long t = wall_clock64();
    atomicMin(minmax, t);
    atomicMax(minmax + 1, t);
  }
  unsigned char *dst = static_cast<unsigned char *>(d.p[blockIdx.y]);   // one destination per grid row: the links work side by side
  const size_t nvec = bytes >> 4;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride)
    reinterpret_cast<gtop_u4 *>(dst)[i] = reinterpret_cast<const gtop_u4 *>(src)[i];
  if (blockIdx.x == 0 && threadIdx.x < (bytes & 15)) dst[(nvec << 4) + threadIdx.x] = src[(nvec << 4) + threadIdx.x];
}

}  // namespace

hipError_t gtop_launch_push_rows(const void *src, size_t bytes, const GtopPushDsts &dsts, int n_dsts,
                                 unsigned long long *minmax, hipStream_t stream) {
  if (bytes == 0 || n_dsts <= 0) return minmax ? gtop_launch_clock_stamp(minmax, stream) : hipSuccess;
  if (n_dsts > GTOP_PUSH_MAX_DSTS) return hipErrorInvalidValue;
  // (16-byte accesses: every pointer 16-byte aligned — the C-ABI checks)
  size_t blocks = ((bytes >> 4) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;      // per destination; a grid-stride loop covers the rest
  hipLaunchKernelGGL(push_rows_kernel, dim3((unsigned)blocks, (unsigned)n_dsts), dim3(256), 0, stream,
                     static_cast<const unsigned char *>(src), bytes, dsts, minmax);
  return hipGetLastError();
}
