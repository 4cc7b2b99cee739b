// On-device loop-closure decision for a 1-vs-N sweep on gfx950: only one 16-byte record leaves the GPU.
//
// Reference (demo/demo3_lcd.py:117-120):  overlaps, yaws = infer_multiple(idx, reference_idx)
//                                         if np.max(overlaps) > overlap_thres: return reference_idx[np.argmax(overlaps)]
// np.argmax returns the FIRST maximum; the reduction below keeps (value, position) pairs and prefers the
// smaller position on ties in every step, so the result does not depend on the reduction tree.
// N is at most a few 1e5 scores (4 B each): one 1024-thread workgroup streams them in a few microseconds, which
// keeps the whole decision in one launch with no inter-block ordering to get right.
#include "ovn_internal.h"

namespace {

constexpr int BM_THREADS = 1024;

struct Best {
  float v;
  int k;
};

__device__ __forceinline__ Best better(Best a, Best b) {
  // larger value wins; equal values: smaller position wins; k < 0 marks "nothing yet"
  if (b.k < 0) return a;
  if (a.k < 0) return b;
  if (b.v > a.v || (b.v == a.v && b.k < a.k)) return b;
  return a;
}

__global__ __launch_bounds__(BM_THREADS) void best_match_kernel(const float* __restrict__ overlap,
                                                                const int32_t* __restrict__ yaw,
                                                                const int32_t* __restrict__ ids, int n, float threshold,
                                                                int index_offset, int32_t* __restrict__ out) {
  __shared__ float sv[BM_THREADS / 64];
  __shared__ int sk[BM_THREADS / 64];
  Best b = {0.f, -1};
  for (int k = threadIdx.x; k < n; k += BM_THREADS) {
    const float v = overlap[k];
    if (v == v) b = better(b, Best{v, k});  // NaN scores never win
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o;
    o.v = __shfl_down(b.v, off, 64);
    o.k = __shfl_down(b.k, off, 64);
    b = better(b, o);
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = b.v;
    sk[threadIdx.x >> 6] = b.k;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < BM_THREADS / 64; ++w) b = better(b, Best{sv[w], sk[w]});
    // ONE 16-byte store: a host that polls word 3 of a record in pinned host memory (engine.best_match(host=True)) sees the four
    // words appear together
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 rec = {-1, 0, 0, 0};
    if (b.k >= 0) rec = (i32x4){ids ? ids[b.k] : b.k + index_offset, __float_as_int(b.v), yaw ? yaw[b.k] : 0, b.v > threshold ? 1 : 0};
    *reinterpret_cast<i32x4*>(out) = rec;
  }
}

}  // namespace

int ovn_best_match_forward(const float* overlap, const int32_t* yaw, const int32_t* ids, int n, float threshold,
                           int index_offset, int32_t* out, hipStream_t stream) {
  hipLaunchKernelGGL(best_match_kernel, dim3(1), dim3(BM_THREADS), 0, stream, overlap, yaw, ids, n, threshold,
                     index_offset, out);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
