// Valid-padded NHWC convolution + bias (+ReLU) as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate) for gfx950.
//
// Computes what Keras' Conv2D(padding='valid') computes for every layer of the reference leg
// (src/two_heads/generateNet.py:161-214) and for c_conv3 of the Delta head (:108-110):
//     out[n,oh,ow,co] = act( bias[co] + sum_{kh,kw,ci} in[n, oh*sh+kh, ow*sw+kw, ci] * W[kh,kw,ci,co] )
//
// GEMM view: M = n*OH*OW output pixels, N = Cout, K = KH*KW*Cin with k = (kh, kw, ci) flattened in
// exactly the Keras kernel order, so W is the row-major [K][Cout] matrix as stored.  Because the
// input is channels-last, the (kw, ci) part of a K index is ONE contiguous run of KW*Cin floats in
// HBM for a given output pixel and kh -- A-tile rows are gathered with 16-byte loads, no im2col.
//
// Tiling (one workgroup = 4 waves): each wave owns WM x WN tiles of 16x16; the K loop advances 16 at
// a time; A (BM x 16) and B (16 x BN, pre-arranged on the host side of the API in fragment order) are
// double-buffered in LDS and every lane fetches its four k-steps of a tile with ONE ds_read_b128.
// K is permuted inside each 16-chunk (lane group g takes k = 4g..4g+3) identically for A and B,
// which leaves the sum unchanged.
#include "ovn_internal.h"

namespace {

constexpr int KC = 16;         // K elements per chunk
constexpr int A_STRIDE = 20;   // floats per A row in LDS (16 + 4 pad; keeps 16-B alignment)

struct ConvArgs {
  const float* in;
  const float* wp;
  const float* bias;
  float* out;
  int H, W, Cin, OH, OW, Cout, SH, SW;
  int K, nkc, KWC, rowstride;
  long long M;
  int relu;
  int out_cols;  // stored output channels = output row stride (<= Cout)
};

__global__ void conv_prep_kernel(const float* __restrict__ w, float* __restrict__ wp, int K, int nkc, int Cout) {
  const int NT = Cout / 16;
  const long long total = (long long)nkc * NT * 256;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    const long long t = e >> 8;
    const int nt = (int)(t % NT);
    const int kc = (int)(t / NT);
    const int k = kc * KC + 4 * (lane >> 4) + s;
    const int n = nt * 16 + (lane & 15);
    wp[e] = (k < K) ? w[(long long)k * Cout + n] : 0.0f;
  }
}

template <int WM, int WN, int WAVES_M, int WAVES_N, bool VEC4>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_mfma_f32_kernel(ConvArgs a) {
  constexpr int NTHREADS = 64 * WAVES_M * WAVES_N;
  constexpr int BM = 16 * WM * WAVES_M;
  constexpr int BN = 16 * WN * WAVES_N;
  constexpr int A_ROWS_PER_THREAD = (BM * 4) / NTHREADS;            // float4 slots of A per thread
  constexpr int B_VEC = BN * 4;                                     // float4 slots of B per chunk
  constexpr int B_PER_THREAD = (B_VEC + NTHREADS - 1) / NTHREADS;
  static_assert((BM * 4) % NTHREADS == 0, "A tile must divide evenly");

  __shared__ __attribute__((aligned(16))) float As[2][BM * A_STRIDE];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * KC];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int lrow = lane & 15;
  const int g = lane >> 4;

  const long long m0 = (long long)blockIdx.x * BM;
  const int nt0 = blockIdx.y * (BN / 16);
  const int NT = a.Cout / 16;

  // staging assignment for A: slot = tid + r*NTHREADS -> (row = slot/4, kq = slot%4)
  long long abase[A_ROWS_PER_THREAD];
#pragma unroll
  for (int r = 0; r < A_ROWS_PER_THREAD; ++r) {
    const int slot = tid + r * NTHREADS;
    long long m = m0 + (slot >> 2);
    if (m >= a.M) m = a.M - 1;  // clamped rows are computed and thrown away
    const int ow = (int)(m % a.OW);
    const long long t2 = m / a.OW;
    const int oh = (int)(t2 % a.OH);
    const long long nb = t2 / a.OH;
    abase[r] = ((nb * a.H + (long long)oh * a.SH) * a.W + (long long)ow * a.SW) * a.Cin;
  }

  f32x4 areg[A_ROWS_PER_THREAD];
  f32x4 breg[B_PER_THREAD];

  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int r = 0; r < A_ROWS_PER_THREAD; ++r) {
      const int slot = tid + r * NTHREADS;
      const int k = kc * KC + 4 * (slot & 3);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (VEC4) {
        if (k < a.K) {
          const int kh = k / a.KWC;
          const int x = k - kh * a.KWC;
          v = *reinterpret_cast<const f32x4*>(a.in + abase[r] + (long long)kh * a.rowstride + x);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = k + e;
          if (kk < a.K) {
            const int kh = kk / a.KWC;
            const int x = kk - kh * a.KWC;
            v[e] = a.in[abase[r] + (long long)kh * a.rowstride + x];
          }
        }
      }
      areg[r] = v;
    }
    const float* wsrc = a.wp + ((long long)kc * NT + nt0) * 256;
#pragma unroll
    for (int r = 0; r < B_PER_THREAD; ++r) {
      const int slot = tid + r * NTHREADS;
      if (slot < B_VEC) breg[r] = *reinterpret_cast<const f32x4*>(wsrc + 4 * slot);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int r = 0; r < A_ROWS_PER_THREAD; ++r) {
      const int slot = tid + r * NTHREADS;
      *reinterpret_cast<f32x4*>(&As[buf][(slot >> 2) * A_STRIDE + 4 * (slot & 3)]) = areg[r];
    }
#pragma unroll
    for (int r = 0; r < B_PER_THREAD; ++r) {
      const int slot = tid + r * NTHREADS;
      if (slot < B_VEC) *reinterpret_cast<f32x4*>(&Bs[buf][4 * slot]) = breg[r];
    }
  };

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  int cur = 0;
  for (int kc = 0; kc < a.nkc; ++kc) {
    const bool more = (kc + 1 < a.nkc);
    if (more) load_chunk(kc + 1);

    f32x4 af[WM], bf[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
      af[i] = *reinterpret_cast<const f32x4*>(&As[cur][((wave_m * WM + i) * 16 + lrow) * A_STRIDE + 4 * g]);
#pragma unroll
    for (int j = 0; j < WN; ++j)
      bf[j] = *reinterpret_cast<const f32x4*>(&Bs[cur][((wave_n * WN + j) * 64 + lane) * 4]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);

    if (more) store_chunk(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: C/D layout of the 16x16 tile -- lane holds column (lane&15), rows 4*(lane>>4)+j
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = (nt0 + wave_n * WN + j) * 16 + lrow;
    const float bv = a.bias[n];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long m = m0 + (wave_m * WM + i) * 16 + 4 * g + r;
        if (m < a.M) {
          float v = acc[i][j][r] + bv;
          if (a.relu) v = fmaxf(v, 0.0f);
          if (n < a.out_cols) a.out[m * a.out_cols + n] = v;
        }
      }
    }
  }
}

template <int WM, int WN, int WAVES_M, int WAVES_N>
int launch_conv(const ConvArgs& a, bool vec4, hipStream_t stream) {
  constexpr int BM = 16 * WM * WAVES_M;
  constexpr int BN = 16 * WN * WAVES_N;
  dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)(a.Cout / BN));
  dim3 block(64 * WAVES_M * WAVES_N);
  if (vec4)
    hipLaunchKernelGGL((conv_mfma_f32_kernel<WM, WN, WAVES_M, WAVES_N, true>), grid, block, 0, stream, a);
  else
    hipLaunchKernelGGL((conv_mfma_f32_kernel<WM, WN, WAVES_M, WAVES_N, false>), grid, block, 0, stream, a);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

}  // namespace

int ovn_conv_prepare(OvnConvLayer* L, const float* kernel_dev, const float* bias_dev, hipStream_t stream) {
  OVN_REQUIRE(L->cout % 16 == 0, OVN_ERR_ARG, "layer %s: cout=%d must be a multiple of 16", L->name.c_str(), L->cout);
  L->K = L->kh * L->kw * L->cin;
  L->nkc = (L->K + KC - 1) / KC;
  const size_t wp_elems = (size_t)L->nkc * (L->cout / 16) * 256;
  OVN_HIP_CHECK(hipMalloc((void**)&L->wp, wp_elems * sizeof(float)));
  OVN_HIP_CHECK(hipMalloc((void**)&L->bias, (size_t)L->cout * sizeof(float)));
  hipLaunchKernelGGL(conv_prep_kernel, dim3(256), dim3(256), 0, stream, kernel_dev, L->wp, L->K, L->nkc, L->cout);
  OVN_HIP_CHECK(hipGetLastError());
  OVN_HIP_CHECK(hipMemcpyAsync(L->bias, bias_dev, (size_t)L->cout * sizeof(float), hipMemcpyDeviceToDevice, stream));
  OVN_HIP_CHECK(hipStreamSynchronize(stream));
  return OVN_OK;
}

void ovn_conv_release(OvnConvLayer* L) {
  if (L->wp) (void)hipFree(L->wp);
  if (L->bias) (void)hipFree(L->bias);
  if (L->wp_h) (void)hipFree(L->wp_h);
  L->wp_h = nullptr;
  if (L->wp_h16) (void)hipFree(L->wp_h16);
  L->wp_h16 = nullptr;
  L->wp = nullptr;
  L->bias = nullptr;
}

int ovn_conv_forward(const OvnConvLayer& L, const float* in, int nb, int h, int w, float* out, int* oh_out,
                     int* ow_out, hipStream_t stream) {
  OVN_REQUIRE(L.wp != nullptr, OVN_ERR_STATE, "layer %s has no weights", L.name.c_str());
  OVN_REQUIRE(h >= L.kh && w >= L.kw, OVN_ERR_ARG, "layer %s: input %dx%d smaller than kernel", L.name.c_str(), h, w);
  ConvArgs a;
  a.in = in;
  a.wp = L.wp;
  a.bias = L.bias;
  a.out = out;
  a.H = h;
  a.W = w;
  a.Cin = L.cin;
  a.OH = (h - L.kh) / L.sh + 1;
  a.OW = (w - L.kw) / L.sw + 1;
  a.Cout = L.cout;
  a.SH = L.sh;
  a.SW = L.sw;
  a.K = L.K;
  a.nkc = L.nkc;
  a.KWC = L.kw * L.cin;
  a.rowstride = w * L.cin;
  a.M = (long long)nb * a.OH * a.OW;
  a.relu = L.relu;
  a.out_cols = L.out_cols > 0 ? L.out_cols : L.cout;
  if (oh_out) *oh_out = a.OH;
  if (ow_out) *ow_out = a.OW;
  if (a.M == 0) return OVN_OK;
  // 16-byte gathers need every (pixel, kh) run to start on a 16-B boundary and K runs to split on 4
  const bool vec4 = (L.cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
  switch (L.cout) {
    case 16: return launch_conv<2, 1, 4, 1>(a, vec4, stream);
    case 32: return launch_conv<2, 2, 4, 1>(a, vec4, stream);
    case 64: return launch_conv<2, 4, 4, 1>(a, vec4, stream);
    default:
      if (L.cout % 128 == 0) return launch_conv<2, 4, 2, 2>(a, vec4, stream);
      return launch_conv<2, 1, 4, 1>(a, vec4, stream);
  }
}
