// Valid-padded NHWC convolution + bias (+ReLU) as an implicit GEMM on the fp16 matrix cores with the scaled 3-term
// split ("f16x3": s x = hi + lo in fp16 with a power-of-two scale s, a*w ~ ah*wh + al*wh + ah*wl, fp32 accumulate:
// 22 significand bits per operand, the error of an fp32 evaluation -- see delta_head_f16x3.hip) for gfx950.
//
// Same contract, GEMM view and gather scheme as conv_f32.hip (reference: Keras Conv2D(padding='valid'),
// generateNet.py:161-214 for the leg); used where fp32 matrix-core time dominates.
// The K loop advances 32 at a time (one v_mfma_f32_16x16x32_f16 step).  A-tile values are scaled and split into hi/lo
// fp16 ONCE when they are staged into LDS (8 values per thread per chunk, amortised over all Cout columns);
// weights are scaled, split and laid out in fragment order when the layer is registered.
// Scales: the weights' is static (max |W| -> 2^14); the activations' comes from `in_max`, the float bits of the largest
// |input| of the call, which the PRODUCER of the tensor leaves in device memory (every kernel here folds its outputs into
// `out_max` with one atomicMax per workgroup; the first layer's input is scanned by ovn_absmax_forward) -- no host round trip.
#include <stdlib.h>

#include "ovn_internal.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 32;          // K elements per chunk
constexpr int A_STRIDE = 40;    // fp16 elements per A row in LDS (32 + 8 pad: 80 B = 5 slots, odd -> no conflicts)

struct ConvArgsB {
  const float* in;
  const _Float16* wp;   // [nkc][Cout/16][hi,lo][64][8], scaled by sw
  const float* bias;
  float* out;
  const unsigned* in_max;   // [scan] float bits of max |input| of every scan (device), written by the producer of `in`
  unsigned* out_max;        // NULL, or [scan]: where this layer folds max |output| of every scan (atomicMax on the float bits)
  float sw;                 // power-of-two scale of wp
  int H, W, Cin, OH, OW, Cout, SH, SW;
  int K, nkc, KWC, rowstride;
  long long M;
  int relu;
};

// (s d0, s d1) -> hi, lo as packed fp16 pairs.  hi = fp16_rtz (v_cvt_pkrtz_f16_f32: one instruction per pair), lo = fp16_rne of the
// exact remainder, through v_fma_mixlo/hi_f16 (`one` = 1.0f in a register keeps the compiler from folding the fma into a subtraction
// that needs two more conversions, see delta_head_f16x3.hip).
__device__ __forceinline__ void split_pair_f16(float d0, float d1, float s, float one, unsigned& hi_pk, unsigned& lo_pk) {
  const float x0 = d0 * s, x1 = d1 * s;
  const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  f16x2 l;
  l[0] = (_Float16)__builtin_fmaf(x0, one, -(float)h[0]);
  l[1] = (_Float16)__builtin_fmaf(x1, one, -(float)h[1]);
  hi_pk = __builtin_bit_cast(unsigned, h);
  lo_pk = __builtin_bit_cast(unsigned, l);
}

// Scaled fp16 hi/lo fragments (f16x3 arithmetic), same fragment order: sw * W = hi + lo, both rounded to nearest.
__global__ void conv_prep_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int K, int nkc, int Cout, float sw) {
  const int NT = Cout / 16;
  const long long total = (long long)nkc * NT * 512;  // (hi, lo) pairs
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    const long long t = e >> 9;
    const int nt = (int)(t % NT);
    const int kc = (int)(t / NT);
    const int k = kc * KC + 8 * (lane >> 4) + s;
    const int n = nt * 16 + (lane & 15);
    const float v = (k < K) ? sw * w[(long long)k * Cout + n] : 0.0f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const long long base = (((long long)kc * NT + nt) * 2) * 512 + lane * 8 + s;
    wp[base] = hi;
    wp[base + 512] = lo;
  }
}

// The same for the pixel-major strip kernel (conv_strip.hip, cin 4 / 16): K order (ky, kx padded to 16 taps, c), i.e. K step
// ks = ky * (cin / 2) + kh covers taps (32 / cin) kh .. of kernel row ky; taps >= kw get zero weights.
__global__ void conv_prep_f16_pad16_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int KH, int KW, int Cin, int Cout,
                                           float sw) {
  const int NT = Cout / 16;
  const int ksr = Cin / 2;                 // K steps per kernel row = 16 taps * Cin / 32
  const int tps = 32 / Cin;
  const long long total = (long long)KH * ksr * NT * 512;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    const long long t = e >> 9;
    const int nt = (int)(t % NT);
    const int ks = (int)(t / NT);
    const int ky = ks / ksr, kh = ks - ky * ksr;
    const int kk = 8 * (lane >> 4) + s;                 // position inside the 32-deep step
    const int tap = tps * kh + kk / Cin, c = kk % Cin;
    const int n = nt * 16 + (lane & 15);
    const float v = (tap < KW) ? sw * w[(((long long)ky * KW + tap) * Cin + c) * Cout + n] : 0.0f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const long long base = (((long long)ks * NT + nt) * 2) * 512 + lane * 8 + s;
    wp[base] = hi;
    wp[base + 512] = lo;
  }
}

// out[0] = max |w[i]|, i < n; one workgroup
__global__ __launch_bounds__(1024) void conv_absmax_kernel(const float* __restrict__ w, long long n, float* __restrict__ out) {
  __shared__ float red[16];
  float m = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    out[0] = m;
  }
}

// The tiles are passed as __restrict__ pointers so that the compiler keeps treating the three LDS regions as
// disjoint when they are carved out of one dynamic allocation (without it the ds_write/ds_read streams serialise).
template <int WM, int WN, int WAVES_M, int WAVES_N, bool VEC4>
__device__ __forceinline__ void conv_mfma_f16x3_body(const ConvArgsB& a, float one, _Float16* __restrict__ Ah, _Float16* __restrict__ Al,
                                                      unsigned char* __restrict__ Bs) {
  constexpr int NTHREADS = 64 * WAVES_M * WAVES_N;
  constexpr int BM = 16 * WM * WAVES_M;
  constexpr int BN = 16 * WN * WAVES_N;
  constexpr int ATILE = BM * A_STRIDE;   // fp16 elements per A buffer
  constexpr int BTILE = BN * 128;        // bytes per B buffer
  constexpr int A_SLOTS = (BM * 4) / NTHREADS;                 // 8-float slots of A per thread per chunk
  constexpr int B_VEC = BN * 8;                                // 16-byte slots of B per chunk (hi + lo)
  constexpr int B_PER_THREAD = (B_VEC + NTHREADS - 1) / NTHREADS;
  static_assert((BM * 4) % NTHREADS == 0, "A tile must divide evenly");


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

  // A tile may span two scans; every ROW is scaled by the power of two that fits its own scan's largest input (a row's K-sum
  // only sees that scan) and the epilogue divides it out per row: a scan's result does not depend on what else is in the batch
  long long abase[A_SLOTS];
  float s_row[A_SLOTS];
#pragma unroll
  for (int r = 0; r < A_SLOTS; ++r) {
    const int slot = tid + r * NTHREADS;
    long long m = m0 + (slot >> 2);
    if (m >= a.M) m = a.M - 1;
    const int ow = (int)(m % a.OW);
    const long long t2 = m / a.OW;
    const int oh = (int)(t2 % a.OH);
    const long long nb = t2 / a.OH;
    abase[r] = ((nb * a.H + (long long)oh * a.SH) * a.W + (long long)ow * a.SW) * a.Cin;
    s_row[r] = ovn_pow2_scale_for(__uint_as_float(a.in_max[nb * OVN_ACTMAX_STRIDE]));
  }

  f32x4 areg[A_SLOTS][2];
  f32x4 breg[B_PER_THREAD];

  // VEC4: position of this thread's 8 k inside the kernel window, kept incrementally (chunks are visited in order, k
  // advances by KC per chunk): ax = offset inside the kernel row, aoff = kh * rowstride.  No integer division in the loop.
  int ax = 0, aoff = 0;
  if (VEC4) {
    const int k0 = 8 * (tid & 3);
    const int kh0 = k0 / a.KWC;
    ax = k0 - kh0 * a.KWC;
    aoff = kh0 * a.rowstride;
  }

  auto load_chunk = [&](int kc) {
    if (VEC4) {
      const bool wrap1 = ax + 4 >= a.KWC;  // the second group of 4 may start on the next kernel row
      const int x1 = ax + 4 - (wrap1 ? a.KWC : 0), off1 = aoff + (wrap1 ? a.rowstride : 0);
      const int k = kc * KC + 8 * (tid & 3);
#pragma unroll
      for (int r = 0; r < A_SLOTS; ++r) {   // NTHREADS % 4 == 0: every slot of a thread shares (slot & 3)
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (k < a.K) v0 = *reinterpret_cast<const f32x4*>(a.in + abase[r] + aoff + ax);
        if (k + 4 < a.K) v1 = *reinterpret_cast<const f32x4*>(a.in + abase[r] + off1 + x1);
        areg[r][0] = v0;
        areg[r][1] = v1;
      }
      ax += KC;
      if (a.KWC >= KC) {  // one conditional step is enough (no loop, no branch: two selects)
        const bool wrap = ax >= a.KWC;
        ax -= wrap ? a.KWC : 0;
        aoff += wrap ? a.rowstride : 0;
      } else {
        while (ax >= a.KWC) {
          ax -= a.KWC;
          aoff += a.rowstride;
        }
      }
    }
    if (!VEC4) {  // scalar gather (Cin not a multiple of 4: the first leg layer at C = 1, 5, ...)
#pragma unroll
      for (int r = 0; r < A_SLOTS; ++r) {
        const int slot = tid + r * NTHREADS;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int k = kc * KC + 8 * (slot & 3) + 4 * hh;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kk = k + e;
            if (kk < a.K) {
              const int kh = kk / a.KWC;
              const int x = kk - kh * a.KWC;
              v[e] = a.in[abase[r] + (long long)kh * a.rowstride + x];
            }
          }
          areg[r][hh] = v;
        }
      }
    }
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wp) + ((long long)kc * NT + nt0) * 2048;
#pragma unroll
    for (int r = 0; r < B_PER_THREAD; ++r) {
      const int slot = tid + r * NTHREADS;
      if (slot < B_VEC) breg[r] = *reinterpret_cast<const f32x4*>(wsrc + 16 * slot);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int r = 0; r < A_SLOTS; ++r) {
      const int slot = tid + r * NTHREADS;
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;
      split_pair_f16(areg[r][0][0], areg[r][0][1], s_row[r], one, h0, l0);
      split_pair_f16(areg[r][0][2], areg[r][0][3], s_row[r], one, h1, l1);
      split_pair_f16(areg[r][1][0], areg[r][1][1], s_row[r], one, h2, l2);
      split_pair_f16(areg[r][1][2], areg[r][1][3], s_row[r], one, h3, l3);
      const int off = (slot >> 2) * A_STRIDE + 8 * (slot & 3);
      *reinterpret_cast<u32x4*>(&Ah[buf * ATILE + off]) = (u32x4){h0, h1, h2, h3};
      *reinterpret_cast<u32x4*>(&Al[buf * ATILE + off]) = (u32x4){l0, l1, l2, l3};
    }
#pragma unroll
    for (int r = 0; r < B_PER_THREAD; ++r) {
      const int slot = tid + r * NTHREADS;
      if (slot < B_VEC) *reinterpret_cast<f32x4*>(&Bs[buf * BTILE + 16 * slot]) = breg[r];
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

    f16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int off = ((wave_m * WM + i) * 16 + lrow) * A_STRIDE + 8 * g;
      ah[i] = *reinterpret_cast<const f16x8*>(&Ah[cur * ATILE + off]);
      al[i] = *reinterpret_cast<const f16x8*>(&Al[cur * ATILE + off]);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int off = (((wave_n * WN + j) * 2) * 64 + lane) * 16;
      bh[j] = *reinterpret_cast<const f16x8*>(&Bs[cur * BTILE + off]);
      bl[j] = *reinterpret_cast<const f16x8*>(&Bs[cur * BTILE + off + 1024]);
    }
    // term-major so that consecutive MFMAs never chain on the same accumulator
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);

    if (more) store_chunk(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: the BM rows of the workgroup lie in at most two scans (a scan has at least BM output rows in every layer this kernel
  // serves; checked by the launcher)
  const long long rows_per_scan = (long long)a.OH * a.OW;
  const long long scan_lo = m0 / rows_per_scan;
  float vmax[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const long long mt0 = m0 + (wave_m * WM + i) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long m = mt0 + 4 * g + r;
      if (m < a.M) {
        const long long scan = m / rows_per_scan;
        const float inv = 1.0f / (ovn_pow2_scale_for(__uint_as_float(a.in_max[scan * OVN_ACTMAX_STRIDE])) * a.sw);
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int n = (nt0 + wave_n * WN + j) * 16 + lrow;
          float v = fmaf(acc[i][j][r], inv, a.bias[n]);
          if (a.relu) v = fmaxf(v, 0.0f);
          a.out[m * a.Cout + n] = v;
          mx = fmaxf(mx, fabsf(v));
        }
        if (scan == scan_lo) vmax[0] = fmaxf(vmax[0], mx);
        else vmax[1] = fmaxf(vmax[1], mx);
      }
    }
  }
  if (a.out_max) {   // kernel-uniform: every thread takes part in the two workgroup reductions (the LDS tiles are free: the K loop
                     // ended with a barrier)
    float* red = reinterpret_cast<float*>(Ah);
    ovn_fold_absmax_wg(vmax[0], a.out_max + scan_lo * OVN_ACTMAX_STRIDE, red);
    __syncthreads();
    ovn_fold_absmax_wg(vmax[1], a.out_max + (scan_lo + 1) * OVN_ACTMAX_STRIDE, red);   // all zero unless the rows straddle two scans: no atomic then
  }
}

// static LDS (<= 64 KB): the common tiles
template <int WM, int WN, int WAVES_M, int WAVES_N, bool VEC4>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_mfma_f16x3_kernel(ConvArgsB a, float one) {
  constexpr int BM = 16 * WM * WAVES_M;
  constexpr int BN = 16 * WN * WAVES_N;
  __shared__ __attribute__((aligned(16))) _Float16 Ah[2 * BM * A_STRIDE];
  __shared__ __attribute__((aligned(16))) _Float16 Al[2 * BM * A_STRIDE];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * BN * 128];
  conv_mfma_f16x3_body<WM, WN, WAVES_M, WAVES_N, VEC4>(a, one, Ah, Al, Bs);
}

// dynamic LDS: [Ah 2 x BM x A_STRIDE][Al same][Bs 2 x BN x 128 B] (the 128 x 128 tile needs 72 KB > the 64 KB static limit)
template <int WM, int WN, int WAVES_M, int WAVES_N, bool VEC4>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_mfma_f16x3_dyn_kernel(ConvArgsB a, float one) {
  constexpr int BM = 16 * WM * WAVES_M;
  extern __shared__ __attribute__((aligned(16))) unsigned char conv_smem[];
  conv_mfma_f16x3_body<WM, WN, WAVES_M, WAVES_N, VEC4>(a, one, reinterpret_cast<_Float16*>(conv_smem),
                                                       reinterpret_cast<_Float16*>(conv_smem) + 2 * BM * A_STRIDE,
                                                       conv_smem + 4 * BM * A_STRIDE * sizeof(_Float16));
}

template <int WM, int WN, int WAVES_M, int WAVES_N>
int launch_conv_b(const ConvArgsB& a, bool vec4, hipStream_t stream) {
  constexpr int BM = 16 * WM * WAVES_M;
  constexpr int BN = 16 * WN * WAVES_N;
  dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)(a.Cout / BN));
  dim3 block(64 * WAVES_M * WAVES_N);
  constexpr size_t lds = 4 * (size_t)BM * A_STRIDE * sizeof(_Float16) + 2 * (size_t)BN * 128;
  if constexpr (lds > 64 * 1024) {
    int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(conv_mfma_f16x3_dyn_kernel<WM, WN, WAVES_M, WAVES_N, true>), lds);
    if (!rc) rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(conv_mfma_f16x3_dyn_kernel<WM, WN, WAVES_M, WAVES_N, false>), lds);
    if (rc) return rc;
    if (vec4)
      hipLaunchKernelGGL((conv_mfma_f16x3_dyn_kernel<WM, WN, WAVES_M, WAVES_N, true>), grid, block, lds, stream, a, 1.0f);
    else
      hipLaunchKernelGGL((conv_mfma_f16x3_dyn_kernel<WM, WN, WAVES_M, WAVES_N, false>), grid, block, lds, stream, a, 1.0f);
  } else {
    if (vec4)
      hipLaunchKernelGGL((conv_mfma_f16x3_kernel<WM, WN, WAVES_M, WAVES_N, true>), grid, block, 0, stream, a, 1.0f);
    else
      hipLaunchKernelGGL((conv_mfma_f16x3_kernel<WM, WN, WAVES_M, WAVES_N, false>), grid, block, 0, stream, a, 1.0f);
  }
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

}  // namespace

int ovn_conv_prepare_f16x3(OvnConvLayer* L, const float* kernel_dev, hipStream_t stream) {
  OVN_REQUIRE(L->cout % 16 == 0, OVN_ERR_ARG, "layer %s: cout=%d must be a multiple of 16", L->name.c_str(), L->cout);
  const int K = L->kh * L->kw * L->cin;
  const int nkc = (K + KC - 1) / KC;
  float* dmax = nullptr;
  OVN_HIP_CHECK(hipMalloc((void**)&dmax, sizeof(float)));
  hipLaunchKernelGGL(conv_absmax_kernel, dim3(1), dim3(1024), 0, stream, kernel_dev, (long long)K * L->cout, dmax);
  float hmax = 0.f;
  hipError_t e = hipMemcpyAsync(&hmax, dmax, sizeof(float), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dmax);
  OVN_HIP_CHECK(e);
  L->sw_h = ovn_pow2_scale_for(hmax);
  const size_t elems = (size_t)nkc * (L->cout / 16) * 1024;  // hi + lo
  OVN_HIP_CHECK(hipMalloc(&L->wp_h, elems * sizeof(_Float16)));
  hipLaunchKernelGGL(conv_prep_f16_kernel, dim3(256), dim3(256), 0, stream, kernel_dev, reinterpret_cast<_Float16*>(L->wp_h), K,
                     nkc, L->cout, L->sw_h);
  if ((L->cin == 4 || L->cin == 16) && L->kw <= 16) {
    const size_t e16 = (size_t)L->kh * (L->cin / 2) * (L->cout / 16) * 1024;
    OVN_HIP_CHECK(hipMalloc(&L->wp_h16, e16 * sizeof(_Float16)));
    hipLaunchKernelGGL(conv_prep_f16_pad16_kernel, dim3(64), dim3(256), 0, stream, kernel_dev, reinterpret_cast<_Float16*>(L->wp_h16),
                       L->kh, L->kw, L->cin, L->cout, L->sw_h);
  }
  OVN_HIP_CHECK(hipGetLastError());
  OVN_HIP_CHECK(hipStreamSynchronize(stream));
  return OVN_OK;
}

// out_max[scan] = max |x| over the `per_scan` floats of every scan (float bits; zero the words first).  Grid (blocks, scans).
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long per_scan, unsigned* __restrict__ out_max) {
  const float* xs = x + (long long)blockIdx.y * per_scan;
  float m = 0.f;
  if ((reinterpret_cast<uintptr_t>(xs) & 15) == 0) {
    const long long n4 = per_scan >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(xs);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      const f32x4 v = x4[i];
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (per_scan & 3)) m = fmaxf(m, fabsf(xs[4 * n4 + threadIdx.x]));
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_scan; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(xs[i]));
  }
  __shared__ float red[16];
  ovn_fold_absmax_wg(m, out_max + (size_t)blockIdx.y * OVN_ACTMAX_STRIDE, red);
}

int ovn_absmax_forward(const float* x, int n_scans, long long per_scan, unsigned* out_max, hipStream_t stream) {
  if (n_scans <= 0 || per_scan <= 0) return OVN_OK;
  const long long want = (per_scan / 4 + 255) / 256;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 64 ? 64 : want));
  hipLaunchKernelGGL(absmax_kernel, dim3(grid, (unsigned)n_scans), dim3(256), 0, stream, x, per_scan, out_max);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

int ovn_conv_forward_f16x3(const OvnConvLayer& L, const float* in, int nb, int h, int w, float* out, int* oh_out,
                           int* ow_out, const unsigned* in_max, unsigned* out_max, hipStream_t stream) {
  OVN_REQUIRE(L.wp_h != nullptr && L.bias != nullptr, OVN_ERR_STATE, "layer %s has no f16x3 weights", L.name.c_str());
  OVN_REQUIRE(in_max != nullptr, OVN_ERR_ARG, "layer %s: f16x3 arithmetic needs the input maximum", L.name.c_str());
  OVN_REQUIRE(h >= L.kh && w >= L.kw, OVN_ERR_ARG, "layer %s: input %dx%d smaller than kernel", L.name.c_str(), h, w);
  ConvArgsB a;
  a.in = in;
  a.wp = reinterpret_cast<const _Float16*>(L.wp_h);
  a.bias = L.bias;
  a.out = out;
  a.in_max = in_max;
  a.out_max = out_max;
  a.sw = L.sw_h;
  a.H = h;
  a.W = w;
  a.Cin = L.cin;
  a.OH = (h - L.kh) / L.sh + 1;
  a.OW = (w - L.kw) / L.sw + 1;
  a.Cout = L.cout;
  a.SH = L.sh;
  a.SW = L.sw;
  a.K = L.kh * L.kw * L.cin;
  a.nkc = (a.K + KC - 1) / KC;
  a.KWC = L.kw * L.cin;
  a.rowstride = w * L.cin;
  a.M = (long long)nb * a.OH * a.OW;
  a.relu = L.relu;
  if (oh_out) *oh_out = a.OH;
  if (ow_out) *ow_out = a.OW;
  if (a.M == 0) return OVN_OK;
  {  // the leg's layer shapes: input strip resident in LDS (conv_strip.hip), for EVERY call size -- same per-accumulator summation
     // order as the generic kernel below and per-scan scales in both, so which of the two runs never shows in the result
    const int took = ovn_conv_strip_try(L, in, nb, nb, h, w, out, in_max, out_max, stream);
    if (took < 0) return -took;
    if (took > 0) return OVN_OK;
  }
  OVN_REQUIRE((long long)a.OH * a.OW >= 128, OVN_ERR_ARG, "layer %s: fewer than 128 output positions per image", L.name.c_str());
  const bool vec4 = (L.cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
  switch (L.cout) {
    case 16: return launch_conv_b<2, 1, 4, 1>(a, vec4, stream);
    case 32: return launch_conv_b<2, 2, 4, 1>(a, vec4, stream);
    case 64: return launch_conv_b<2, 4, 4, 1>(a, vec4, stream);
    default:
      // many output rows (c_conv3 of a sweep: 495 k rows): 128 x 128 tile, 8 waves of 32 x 64 -- each staged weight tile
      // (16 KB through the ~79 B/clk LDS store path) is shared by twice as many rows and a barrier covers twice the
      // MFMAs; 0.90 vs 1.08 ms for 1024 pairs.  72 KB of LDS, hence the dynamic-LDS kernel.
      static const long long big_m = getenv("OVN_CONV_BIG_M") ? atoll(getenv("OVN_CONV_BIG_M")) : 32ll * 1024;
      if (L.cout % 128 == 0 && a.M >= big_m) return launch_conv_b<2, 4, 4, 2>(a, vec4, stream);
      if (L.cout % 128 == 0) return launch_conv_b<2, 4, 2, 2>(a, vec4, stream);
      return launch_conv_b<2, 1, 4, 1>(a, vec4, stream);
  }
}
