// Correlation (yaw) head of OverlapNet for gfx950.
//
// Reference: src/two_heads/generateNet.py:327-354 -> NormalizedCorrelation2D(normalize='none')
// (NormalizedCorrelation2D.py:43-109) over RangePadding2D(padding=W/2) (RangePadding2D.py:31-38):
//     corr[k] = sum_{j<360} sum_{c<128} l[(k + j + 180) mod 360, c] * r[j, c],     k in [0,360)
// and the post-processing of Infer (infer.py:158): yaw = 180 - argmax_k corr[k], first maximum wins.
//
// Direct form on the fp32 matrix cores: the Gram matrix G = l r^T (360x360, K = 128) is produced
// 16 query columns at a time with v_mfma_f32_16x16x4_f32 and immediately folded along its wrapped
// diagonals, corr[k] += G[(k + j + 180) mod 360, j].  Thread k owns corr[k]; panels and the columns
// inside a panel are added in a fixed order, so the result (and the argmax) is deterministic.
// One workgroup (8 waves) = one pair; the candidate feature volume is read from HBM exactly once.
#include "ovn_internal.h"

namespace {

constexpr int FW = OVN_FEAT_W;
constexpr int FC = OVN_FEAT_C;
constexpr int GS_STRIDE = 17;  // floats per Gram row in LDS (16 + 1: column reads hit distinct banks)

__global__ __launch_bounds__(512) void corr_head_kernel(const float* __restrict__ feats_l,
                                                        const int32_t* __restrict__ lidx,
                                                        const float* __restrict__ feats_r,
                                                        const int32_t* __restrict__ ridx, int32_t* __restrict__ yaw,
                                                        float* __restrict__ corr) {
  __shared__ float gs[FW * GS_STRIDE];
  __shared__ float red_v[8];
  __shared__ int red_i[8];

  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lrow = lane & 15;
  const int g = lane >> 4;

  const float* L = feats_l + (long long)(lidx ? lidx[pair] : pair) * OVN_FEAT_ELEMS;
  const float* R = feats_r + (long long)(ridx ? ridx[pair] : 0) * OVN_FEAT_ELEMS;

  // A operand: rows i = 48*wave + 16*t + lrow, channels 32g..32g+31 (same slice as the Delta kernel)
  f32x4 lreg[3][8];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int i = 48 * wave + 16 * t + lrow;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      lreg[t][q] = (i < FW) ? *reinterpret_cast<const f32x4*>(L + i * FC + 32 * g + 4 * q)
                            : (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  float partial = 0.f;
  for (int j0 = 0; j0 < FW; j0 += 16) {
    // B operand: column j = j0 + lrow of r^T, channels 32g..32g+31
    int j = j0 + lrow;
    if (j > FW - 1) j = FW - 1;  // last panel: columns 360..367 are padding, masked below
    f32x4 rreg[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) rreg[q] = *reinterpret_cast<const f32x4*>(R + j * FC + 32 * g + 4 * q);

    f32x4 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 3; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(lreg[t][q][e], rreg[q][e], acc[t], 0, 0, 0);

    // G tile -> LDS.  C/D: lane holds column lrow (= j - j0), rows 4g..4g+3.
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 48 * wave + 16 * t + 4 * g + r;
        if (i < FW) gs[i * GS_STRIDE + lrow] = acc[t][r];
      }
    __syncthreads();
    if (tid < FW) {
      const int jn = (FW - j0 < 16) ? (FW - j0) : 16;
      int row = tid + j0 + FW / 2;
      row -= (row >= FW) ? FW : 0;
      row -= (row >= FW) ? FW : 0;
      for (int jl = 0; jl < jn; ++jl) {
        partial += gs[row * GS_STRIDE + jl];
        ++row;
        if (row == FW) row = 0;
      }
    }
    __syncthreads();
  }

  if (corr && tid < FW) corr[(long long)pair * FW + tid] = partial;

  // argmax with first-maximum-wins (np.argmax semantics, infer.py:158)
  float bv = (tid < FW) ? partial : -INFINITY;
  int bi = (tid < FW) ? tid : 0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_down(bv, off, 64);
    const int oi = __shfl_down(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if (lane == 0) {
    red_v[wave] = bv;
    red_i[wave] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    float v = red_v[0];
    int i = red_i[0];
    for (int w = 1; w < 8; ++w)
      if (red_v[w] > v || (red_v[w] == v && red_i[w] < i)) {
        v = red_v[w];
        i = red_i[w];
      }
    yaw[pair] = FW / 2 - i;
  }
}

}  // namespace

int ovn_corr_forward(const float* feats_l, const int32_t* lidx, const float* feats_r, const int32_t* ridx, int n,
                     int32_t* yaw, float* corr, hipStream_t stream) {
  hipLaunchKernelGGL(corr_head_kernel, dim3(n), dim3(512), 0, stream, feats_l, lidx, feats_r, ridx, yaw, corr);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
