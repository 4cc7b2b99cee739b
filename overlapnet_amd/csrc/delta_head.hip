// Delta (overlap) head of OverlapNet for gfx950: DeltaLayer + c_conv1 + c_conv2 fused in one kernel,
// plus the Dense(1)+sigmoid tail.  (c_conv3 runs on the generic conv kernel of conv_f32.hip.)
//
// Reference: src/two_heads/generateNet.py:15-61 (DeltaLayer) and :64-116 (head).
//   diff[i,j,c] = |L[i,c] - R[j,c]|                       (360 x 360 x 128 per pair, 66 MB in the reference)
//   o1[i,jb,o]  = b1[o] + sum_{dj<15,c} diff[i,15jb+dj,c] * W1[dj,c,o]         c_conv1, linear
//   o2[ib,jb,p] = relu(b2[p] + sum_{di<15,o} o1[15ib+di,jb,o] * W2[di,o,p])    c_conv2
// The diff tensor is never materialised: each lane keeps its slice of L in registers for the whole
// pair and forms |L - R| on the fly as the A operand of v_mfma_f32_16x16x4_f32 (exact fp32).
//
// One workgroup (8 waves) = one pair.  For each of the 24 column groups jb:
//   GEMM1  (360 x 1920) x (1920 x 64): wave w owns rows 48w..48w+47 (3 tiles of 16), all 64 outputs.
//          K = (dj, c) is walked dj-major; within a dj, lane group g = lane>>4 covers channels
//          32g..32g+31, so a lane needs exactly L[i, 32g..32g+31] (32 registers per row tile) and the
//          matching R row comes from LDS as a broadcast ds_read_b128.  W1 is pre-permuted to that order.
//   o1 (+b1) goes to LDS laid out as the [24][960] A matrix of GEMM2 (row ib = 15 consecutive i rows).
//   GEMM2  (24 x 960) x (960 x 128): 2 x 8 tiles of 16x16 over the 8 waves, + b2, ReLU, store o2.
#include "ovn_internal.h"

namespace {

constexpr int FW = OVN_FEAT_W;        // 360
constexpr int FC = OVN_FEAT_C;        // 128
constexpr int S = OVN_S;              // 15
constexpr int G = OVN_G;              // 24
constexpr int O1 = OVN_C1_OUT;        // 64
constexpr int O2 = OVN_C2_OUT;        // 128
constexpr int K2 = S * O1;            // 960
constexpr int O1S_STRIDE = K2 + 4;    // 964 floats: odd number of 16-B slots -> conflict-free b128 rows
constexpr int LDS_FLOATS = G * O1S_STRIDE + S * FC;

// W1p[dj][sq][nt][lane][e] = W1[dj][c = 32*(lane>>4) + 4*sq + e][o = 16*nt + (lane&15)]
__global__ void delta_prep_w1_kernel(const float* __restrict__ w1, float* __restrict__ w1p) {
  const int total = S * 8 * 4 * 64 * 4;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 3;
    const int lane = (idx >> 2) & 63;
    const int nt = (idx >> 8) & 3;
    const int sq = (idx >> 10) & 7;
    const int dj = idx >> 13;
    const int c = 32 * (lane >> 4) + 4 * sq + e;
    const int o = 16 * nt + (lane & 15);
    w1p[idx] = w1[(dj * FC + c) * O1 + o];  // Keras (1,15,128,64) flattened
  }
}

__global__ __launch_bounds__(512) void delta_c12_kernel(const float* __restrict__ feats_l,
                                                        const int32_t* __restrict__ lidx,
                                                        const float* __restrict__ feats_r,
                                                        const int32_t* __restrict__ ridx,
                                                        const float* __restrict__ w1p, const float* __restrict__ b1,
                                                        const float* __restrict__ w2p, const float* __restrict__ b2,
                                                        float* __restrict__ o2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* o1s = smem;                     // [24][964]
  float* rs = smem + G * O1S_STRIDE;     // [15][128]

  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lrow = lane & 15;
  const int g = lane >> 4;

  const float* L = feats_l + (long long)(lidx ? lidx[pair] : pair) * OVN_FEAT_ELEMS;
  const float* R = feats_r + (long long)(ridx ? ridx[pair] : 0) * OVN_FEAT_ELEMS;

  // this lane's slice of L: rows 48*wave + 16*t + lrow, channels 32g..32g+31
  f32x4 lreg[3][8];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int i = 48 * wave + 16 * t + lrow;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      lreg[t][q] = (i < FW) ? *reinterpret_cast<const f32x4*>(L + i * FC + 32 * g + 4 * q)
                            : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }

  for (int jb = 0; jb < G; ++jb) {
    __syncthreads();  // previous group's GEMM2 has finished reading o1s / rs
    if (tid < S * FC / 4)
      *reinterpret_cast<f32x4*>(rs + 4 * tid) = *reinterpret_cast<const f32x4*>(R + jb * S * FC + 4 * tid);
    __syncthreads();

    f32x4 acc[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int dj = 0; dj < S; ++dj) {
      const float* wrow = w1p + dj * 8192 + lane * 4;
      const float* rrow = rs + dj * FC + 32 * g;
#pragma unroll
      for (int sq = 0; sq < 8; ++sq) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(rrow + 4 * sq);
        f32x4 bw[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bw[nt] = *reinterpret_cast<const f32x4*>(wrow + (sq * 4 + nt) * 256);
        f32x4 d[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const f32x4 lv = lreg[t][sq];
          d[t][0] = fabsf(lv[0] - rv[0]);
          d[t][1] = fabsf(lv[1] - rv[1]);
          d[t][2] = fabsf(lv[2] - rv[2]);
          d[t][3] = fabsf(lv[3] - rv[3]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[t][e], bw[nt][e], acc[t][nt], 0, 0, 0);
      }
    }

    // o1 (+ bias) -> LDS in GEMM2's A layout.  C/D: lane holds column lrow, rows 4g..4g+3 of the tile.
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int o = 16 * nt + lrow;
      const float bv = b1[o];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 48 * wave + 16 * t + 4 * g + r;
          if (i < FW) {
            const int ib = i / S;
            const int di = i - ib * S;
            o1s[ib * O1S_STRIDE + di * O1 + o] = acc[t][nt][r] + bv;
          }
        }
      }
    }
    __syncthreads();

    // GEMM2: wave -> m-tile (wave&1), n-tiles 2*(wave>>1) and +1
    {
      const int mt = wave & 1;
      const int ntp = wave >> 1;
      int ib = 16 * mt + lrow;
      if (ib > G - 1) ib = G - 1;  // rows 24..31 of the second tile are padding
      const float* arow = o1s + ib * O1S_STRIDE + 4 * g;
      const float* wcol = w2p + (2 * ntp) * 256 + lane * 4;
      f32x4 acc2[2];
      acc2[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int kc = 0; kc < K2 / 16; ++kc) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * kc);
        const f32x4 bv0 = *reinterpret_cast<const f32x4*>(wcol + kc * (8 * 256));
        const f32x4 bv1 = *reinterpret_cast<const f32x4*>(wcol + kc * (8 * 256) + 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv0[e], acc2[0], 0, 0, 0);
          acc2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv1[e], acc2[1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int p = 16 * (2 * ntp + q) + lrow;
        const float bv = b2[p];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ib2 = 16 * mt + 4 * g + r;
          if (ib2 < G) {
            const float v = fmaxf(acc2[q][r] + bv, 0.0f);
            o2[(((long long)pair * G + ib2) * G + jb) * O2 + p] = v;
          }
        }
      }
    }
  }
}

// logit[n] = bd + <o3[n,:], wd>, overlap = sigmoid(logit).  Flatten order (H,W,C) == o3's NHWC layout
// (generateNet.py:112-114).  One workgroup per pair, fixed reduction order (deterministic).
__global__ __launch_bounds__(256) void dense_sigmoid_kernel(const float* __restrict__ o3, const float* __restrict__ wd,
                                                            const float* __restrict__ bd, float* __restrict__ overlap,
                                                            float* __restrict__ logit) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const f32x4* x = reinterpret_cast<const f32x4*>(o3 + (long long)n * OVN_DENSE_IN);
  const f32x4* w = reinterpret_cast<const f32x4*>(wd);
  float s = 0.f;
  for (int i = threadIdx.x; i < OVN_DENSE_IN / 4; i += 256) {
    const f32x4 a = x[i];
    const f32x4 b = w[i];
    s += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float z = ((red[0] + red[1]) + (red[2] + red[3])) + bd[0];
    if (logit) logit[n] = z;
    overlap[n] = 1.0f / (1.0f + expf(-z));
  }
}

}  // namespace

int ovn_delta_prepare_w1(const float* c1_kernel_dev, float** w1p_out, hipStream_t stream) {
  const size_t elems = (size_t)S * FC * O1;
  OVN_HIP_CHECK(hipMalloc((void**)w1p_out, elems * sizeof(float)));
  hipLaunchKernelGGL(delta_prep_w1_kernel, dim3(120), dim3(256), 0, stream, c1_kernel_dev, *w1p_out);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

int ovn_delta_c12_forward(const ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r,
                          const int32_t* ridx, int n, float* o2, hipStream_t stream) {
  const size_t lds = (size_t)LDS_FLOATS * sizeof(float);
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(delta_c12_kernel), lds);
  if (rc) return rc;
  hipLaunchKernelGGL(delta_c12_kernel, dim3(n), dim3(512), lds, stream, feats_l, lidx, feats_r, ridx, ctx->w1p,
                     ctx->b1, ctx->c2.wp, ctx->c2.bias, o2);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

int ovn_dense_sigmoid_forward(const ovn_ctx* ctx, const float* o3, int n, float* overlap, float* logit,
                              hipStream_t stream) {
  hipLaunchKernelGGL(dense_sigmoid_kernel, dim3(n), dim3(256), 0, stream, o3, ctx->wd, ctx->bd, overlap, logit);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
