// Ground-truth overlap between a frame and N reference scans for gfx950 (the label producer next to the hot path).
//
// Reference (src/utils/com_overlap_yaw.py:28-46): every reference scan is moved into the current frame,
//     p_world = pose_ref . p,   p_cur = inv(pose_cur) . p_world            (two float64 matrix products, :37-39)
// range-projected IN FLOAT64 (range_projection on load_vertex's float64 points, utils.py:59-134,217-230), and
//     overlap = #{ pixels : ref_range > 0 and |ref_range - cur_range| < 1 } / #{ cur_range > 0 }.
// Only the range image is needed, so "nearest point wins" is a 32-bit atomicMin over the float32 bits of the depth
// (the image is float32, utils.py:120-121; positive floats order like their bit patterns).  HBM-bound: 16 B per point in,
// one atomic per kept point; one workgroup row per scan so that ragged scans need no host-side padding.
// Compiled with -ffp-contract=off: the fused multiply-adds below are explicit.
#include <math.h>

#include "ovn_internal.h"

namespace {

constexpr unsigned EMPTY = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void gt_fill_kernel(unsigned* __restrict__ img, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    img[i] = EMPTY;
}

// dot of a pose row with (x, y, z, 1) the way a BLAS dgemm micro-kernel accumulates it: k = 0..3, fused
__device__ __forceinline__ double row_dot(const double* __restrict__ r, double x, double y, double z) {
  double acc = r[0] * x;
  acc = fma(r[1], y, acc);
  acc = fma(r[2], z, acc);
  acc = fma(r[3], 1.0, acc);
  return acc;
}

__global__ __launch_bounds__(256) void gt_scatter_kernel(const float* __restrict__ points, const int64_t* __restrict__ offsets,
                                                         const double* __restrict__ ref_poses,
                                                         const double* __restrict__ inv_cur_pose, int H, int W, double fov_down_abs,
                                                         double fov, double max_range, unsigned* __restrict__ img) {
  const int s = blockIdx.y;
  const long long beg = offsets[s], end = offsets[s + 1];
  const long long i = beg + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= end) return;
  const f32x4 p = *reinterpret_cast<const f32x4*>(points + 4 * i);
  double x = (double)p[0], y = (double)p[1], z = (double)p[2];
  if (ref_poses) {
    const double* P = ref_poses + 16 * (long long)s;
    const double wx = row_dot(P, x, y, z), wy = row_dot(P + 4, x, y, z), wz = row_dot(P + 8, x, y, z);
    x = wx;
    y = wy;
    z = wz;
  }
  if (inv_cur_pose) {
    const double cx = row_dot(inv_cur_pose, x, y, z), cy = row_dot(inv_cur_pose + 4, x, y, z),
                 cz = row_dot(inv_cur_pose + 8, x, y, z);
    x = cx;
    y = cy;
    z = cz;
  }
  const double depth = sqrt((x * x + y * y) + z * z);
  if (!(depth > 0.0 && depth < max_range)) return;
  const double yaw = -atan2(y, x);
  const double pitch = asin(z / depth);
  double px = 0.5 * (yaw / 3.141592653589793 + 1.0);
  double py = 1.0 - (pitch + fov_down_abs) / fov;
  px = floor(px * (double)W);
  py = floor(py * (double)H);
  px = fmax(0.0, fmin((double)(W - 1), px));
  py = fmax(0.0, fmin((double)(H - 1), py));
  const int pix = (int)py * W + (int)px;
  atomicMin(img + (long long)s * H * W + pix, __float_as_uint((float)depth));
}

// uint image -> float range image in place (-1 = empty)
__global__ __launch_bounds__(256) void gt_finish_kernel(unsigned* __restrict__ img, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned v = img[i];
    if (v == EMPTY) img[i] = __float_as_uint(-1.0f);
  }
}

// counts[s] = #{ref > 0 and |ref - cur| < 1}; counts[n] = #{cur > 0} (block n)
__global__ __launch_bounds__(256) void gt_count_kernel(const float* __restrict__ ref, const float* __restrict__ cur, int npix,
                                                       int n, int32_t* __restrict__ counts) {
  __shared__ int part[4];
  const int s = blockIdx.x;
  int c = 0;
  if (s < n) {
    const float* r = ref + (long long)s * npix;
    for (int i = threadIdx.x; i < npix; i += 256) {
      const float v = r[i];
      if (v > 0.0f && fabsf(v - cur[i]) < 1.0f) ++c;
    }
  } else {
    for (int i = threadIdx.x; i < npix; i += 256)
      if (cur[i] > 0.0f) ++c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[s] = part[0] + part[1] + part[2] + part[3];
}

}  // namespace

int ovn_gt_range_forward(const float* points, const int64_t* offsets, int n_scans, long long max_points, const double* ref_poses,
                         const double* inv_cur_pose, int H, int W, double fov_up_deg, double fov_down_deg, double max_range,
                         float* range_out, hipStream_t stream) {
  const long long total = (long long)n_scans * H * W;
  if (total == 0) return OVN_OK;
  unsigned* img = reinterpret_cast<unsigned*>(range_out);
  hipLaunchKernelGGL(gt_fill_kernel, dim3(1024), dim3(256), 0, stream, img, total);
  if (max_points > 0) {
    const double up = fov_up_deg / 180.0 * 3.141592653589793, down = fov_down_deg / 180.0 * 3.141592653589793;
    const double fov = fabs(down) + fabs(up);
    dim3 grid((unsigned)((max_points + 255) / 256), (unsigned)n_scans);
    hipLaunchKernelGGL(gt_scatter_kernel, grid, dim3(256), 0, stream, points, offsets, ref_poses, inv_cur_pose, H, W, fabs(down),
                       fov, max_range, img);
  }
  hipLaunchKernelGGL(gt_finish_kernel, dim3(1024), dim3(256), 0, stream, img, total);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

int ovn_gt_count_forward(const float* ref_ranges, const float* cur_range, int n, int npix, int32_t* counts, hipStream_t stream) {
  hipLaunchKernelGGL(gt_count_kernel, dim3(n + 1), dim3(256), 0, stream, ref_ranges, cur_range, npix, n, counts);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
