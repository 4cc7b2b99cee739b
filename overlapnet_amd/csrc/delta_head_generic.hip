// DeltaLayer + c_conv1 for ANY conv1NetworkHead_conv1size s (generateNet.py:15-61, :88-99), fp32, for gfx950.
//
// The shipped network.yml leaves the key at its default (15), and the two fast Delta paths (delta_head_f16x3.hip, delta_head.hip)
// are tiled around it: 24 column groups of 15.  The reference builds the head for any s (Conv2D(64, (1, s), strides (1, s)) on the
// 360 x 360 x 128 difference tensor, then Conv2D(128, (s, 1), strides (s, 1)), generateNet.py:96-106; 'valid' padding: G = 360 // s
// groups, a remainder of columns / rows is dropped), so a configuration with another s must load and run here too.  This is
// the generality path, not the benchmarked one: plain fp32 FMAs, the difference tensor still never materialised.
//
//   out1[pair][i][jb][o] = b1[o] + sum_{dj < s} sum_{c < 128} | l[i][c] - r[s jb + dj][c] | W1[dj][c][o]      i < 360, jb < G, o < 64
//
// is written as the NHWC image (n, 360, G, 64); c_conv2 (s x 1, stride (s, 1)) and c_conv3 then run through the generic fp32
// implicit-GEMM kernel (conv_f32.hip) and the Dense layer through dense_sigmoid_kernel.
// Workgroup = (pair, jb, block of 64 rows i); thread = (output channel o, 16 rows): the 64 l rows and the s r rows sit in LDS (all
// lanes of a wave read the same address: broadcasts), one coalesced 256-byte weight row per (dj, c) from L2.
#include "ovn_internal.h"

namespace {

constexpr int FW = OVN_FEAT_W;    // 360
constexpr int FC = OVN_FEAT_C;    // 128
constexpr int O1 = OVN_C1_OUT;    // 64
constexpr int IB = 64;            // rows i per workgroup
constexpr int RPT = 16;           // rows per thread

__global__ __launch_bounds__(256) void delta_c1_generic_kernel(const float* __restrict__ feats_l, const int32_t* __restrict__ lidx,
                                                               const float* __restrict__ feats_r, const int32_t* __restrict__ ridx,
                                                               const float* __restrict__ w1, const float* __restrict__ b1, int s, int G,
                                                               float* __restrict__ out1) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* ll = gsm;                  // [IB][128]
  float* rl = gsm + IB * FC;        // [s][128]
  const int nib = (FW + IB - 1) / IB;
  int bid = blockIdx.x;
  const int iblk = bid % nib;
  bid /= nib;
  const int jb = bid % G;
  const int pair = bid / G;
  const int tid = threadIdx.x;
  const int o = tid & (O1 - 1);
  const int ig = tid >> 6;
  const float* L = feats_l + (long long)(lidx ? lidx[pair] : pair) * OVN_FEAT_ELEMS;
  const float* R = feats_r + (long long)(ridx ? ridx[pair] : 0) * OVN_FEAT_ELEMS;
  const int i0 = iblk * IB;
  for (int e = tid; e < IB * FC / 4; e += 256) {
    const int row = e / (FC / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i0 + row < FW) v = *reinterpret_cast<const f32x4*>(L + (size_t)(i0 + row) * FC + 4 * (e - row * (FC / 4)));
    *reinterpret_cast<f32x4*>(ll + 4 * e) = v;
  }
  for (int e = tid; e < s * FC / 4; e += 256) *reinterpret_cast<f32x4*>(rl + 4 * e) = *reinterpret_cast<const f32x4*>(R + (size_t)s * jb * FC + 4 * e);
  __syncthreads();
  float acc[RPT];
#pragma unroll
  for (int u = 0; u < RPT; ++u) acc[u] = 0.f;
  const float* lrow = ll + (RPT * ig) * FC;
  for (int dj = 0; dj < s; ++dj) {
    const float* wrow = w1 + (size_t)dj * FC * O1 + o;
    const float* rrow = rl + dj * FC;
#pragma unroll 2
    for (int c = 0; c < FC; c += 4) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(rrow + c);
      const float w0 = wrow[(c + 0) * O1], w1v = wrow[(c + 1) * O1], w2 = wrow[(c + 2) * O1], w3 = wrow[(c + 3) * O1];
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const f32x4 lv = *reinterpret_cast<const f32x4*>(lrow + u * FC + c);
        float a = acc[u];
        a = fmaf(fabsf(lv[0] - rv[0]), w0, a);    // fixed order over (dj, c): deterministic
        a = fmaf(fabsf(lv[1] - rv[1]), w1v, a);
        a = fmaf(fabsf(lv[2] - rv[2]), w2, a);
        a = fmaf(fabsf(lv[3] - rv[3]), w3, a);
        acc[u] = a;
      }
    }
  }
  const float bv = b1[o];
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const int i = i0 + RPT * ig + u;
    if (i < FW) out1[(((size_t)pair * FW + i) * G + jb) * O1 + o] = acc[u] + bv;   // c_conv1 is linear (generateNet.py:96-99)
  }
}

// logit[n] = bd + <o3[n,:], wd>, overlap = sigmoid(logit) for a Dense input of `dense_in` floats (a multiple of 4); Flatten order
// (H, W, C) == o3's NHWC layout (generateNet.py:112-114).  One workgroup per pair, fixed reduction order.
__global__ __launch_bounds__(256) void dense_sigmoid_any_kernel(const float* __restrict__ o3, const float* __restrict__ wd,
                                                                const float* __restrict__ bd, long long dense_in,
                                                                float* __restrict__ overlap, float* __restrict__ logit) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  const f32x4* x = reinterpret_cast<const f32x4*>(o3 + (long long)n * dense_in);
  const f32x4* w = reinterpret_cast<const f32x4*>(wd);
  float s = 0.f;
  for (long long i = threadIdx.x; i < dense_in / 4; i += 256) {
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

// Bytes of scratch per pair of the general path: out1 (360, G, 64) | o2 (G, G, 128) | o3 (G - 2, G - 2, 256), each 256-byte aligned
size_t ovn_delta_generic_pair_bytes(int G) {
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  return al((size_t)FW * G * O1 * 4) + al((size_t)G * G * OVN_C2_OUT * 4) + al((size_t)(G - 2) * (G - 2) * OVN_C3_OUT * 4);
}

// The whole Delta head for n pairs at conv1size s = ctx->head_s (any value with 360 // s >= 3); scratch: n * pair_bytes
int ovn_delta_generic_forward(const ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r,
                              const int32_t* ridx, int n, void* scratch, float* overlap, float* logit, hipStream_t stream) {
  const int s = ctx->head_s, G = ctx->head_g;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  float* out1 = static_cast<float*>(scratch);
  float* o2 = reinterpret_cast<float*>(static_cast<char*>(scratch) + al((size_t)n * FW * G * O1 * 4));
  float* o3 = reinterpret_cast<float*>(reinterpret_cast<char*>(o2) + al((size_t)n * G * G * OVN_C2_OUT * 4));
  const size_t lds = ((size_t)IB * FC + (size_t)s * FC) * sizeof(float);
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(delta_c1_generic_kernel), lds);
  if (rc) return rc;
  const int nib = (FW + IB - 1) / IB;
  hipLaunchKernelGGL(delta_c1_generic_kernel, dim3((unsigned)(nib * G * n)), dim3(256), lds, stream, feats_l, lidx, feats_r, ridx,
                     ctx->w1raw, ctx->b1, s, G, out1);
  OVN_HIP_CHECK(hipGetLastError());
  int oh = 0, ow = 0;
  rc = ovn_conv_forward(ctx->c2, out1, n, FW, G, o2, &oh, &ow, stream);          // (n, 360, G, 64) -> (n, G, G, 128), s x 1 / stride (s, 1)
  if (rc) return rc;
  OVN_REQUIRE(oh == G && ow == G, OVN_ERR_STATE, "general Delta head: c_conv2 produced %dx%d, expected %dx%d", oh, ow, G, G);
  rc = ovn_conv_forward(ctx->c3, o2, n, G, G, o3, &oh, &ow, stream);              // -> (n, G - 2, G - 2, 256)
  if (rc) return rc;
  const long long dense_in = (long long)(G - 2) * (G - 2) * OVN_C3_OUT;
  hipLaunchKernelGGL(dense_sigmoid_any_kernel, dim3(n), dim3(256), 0, stream, o3, ctx->wd, ctx->bd, dense_in, overlap, logit);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
