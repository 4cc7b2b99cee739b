// Which |L-R| split is cheapest NEXT TO the MFMAs it feeds?  (round 2; same harness as ubench3.hip)
// Per iteration and wave: the A operands of 8 elements per lane (L in registers, R from LDS at a moving address) and the MFMAs
// they feed.  Variants F:
//   10  bf16: sub, sub, and, and, sub|.|, sub|.|, perm, cvt_pk_bf16            (8 VALU / element pair; round-1 kernel) + 12 bf16 MFMA
//   11  fp16: sub, sub, cvt_pkrtz |.|, fma_mixlo, fma_mixhi                    (5)                                   + 12 f16 MFMA
//   12  fp16: sub, sub, cvt_pkrtz |.|, cvt_f32_f16 x2, sub x2, cvt_pk_f16      (8)                                   + 12 f16 MFMA
//   13  fp16 min form: operands pre-split and packed (hi << 16 | lo), v_min_u32 per element, 2 v_perm per pair (4)   + 12 f16 MFMA
//   14  as 11 with the RNE v_cvt_pk_f16_f32 for hi                                                                  + 12 f16 MFMA
//   15  fp16 min form, K-interleaved operands: v_min_u32 per element only (2 per pair), 4 product terms              + 16 f16 MFMA
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/experiments/ubench5.hip -o tools/bin/ubench5
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int F>
__device__ __forceinline__ void split_pair_f(float l0, float l1, float r0, float r1, float one, unsigned& hi_pk, unsigned& lo_pk) {
  if (F == 10) {
    const float d0 = l0 - r0, d1 = l1 - r1;
    const unsigned h0 = __float_as_uint(d0) & 0x7fff0000u, h1 = __float_as_uint(d1) & 0x7fff0000u;
    const float q0 = fabsf(d0) - __uint_as_float(h0), q1 = fabsf(d1) - __uint_as_float(h1);
    hi_pk = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    bf16x2_t lp;
    lp[0] = (__bf16)q0;
    lp[1] = (__bf16)q1;
    lo_pk = __builtin_bit_cast(unsigned, lp);
  } else if (F == 11 || F == 14) {
    const float d0 = l0 - r0, d1 = l1 - r1;
    f16x2 h;
    if (F == 11) {
      h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(fabsf(d0), fabsf(d1)));
    } else {
      h[0] = (_Float16)fabsf(d0);
      h[1] = (_Float16)fabsf(d1);
    }
    f16x2 lo;
    lo[0] = (_Float16)__builtin_fmaf(fabsf(d0), one, -(float)h[0]);
    lo[1] = (_Float16)__builtin_fmaf(fabsf(d1), one, -(float)h[1]);
    hi_pk = __builtin_bit_cast(unsigned, h);
    lo_pk = __builtin_bit_cast(unsigned, lo);
  } else if (F == 12) {
    const float d0 = l0 - r0, d1 = l1 - r1;
    const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(fabsf(d0), fabsf(d1)));
    f16x2 lo;
    lo[0] = (_Float16)(fabsf(d0) - (float)h[0]);
    lo[1] = (_Float16)(fabsf(d1) - (float)h[1]);
    hi_pk = __builtin_bit_cast(unsigned, h);
    lo_pk = __builtin_bit_cast(unsigned, lo);
  } else if (F == 13) {
    const unsigned m0 = min(__float_as_uint(l0), __float_as_uint(r0)), m1 = min(__float_as_uint(l1), __float_as_uint(r1));
    hi_pk = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    lo_pk = __builtin_amdgcn_perm(m1, m0, 0x05040100u);
  } else {  // 15: the words themselves are the operands
    hi_pk = min(__float_as_uint(l0), __float_as_uint(r0));
    lo_pk = min(__float_as_uint(l1), __float_as_uint(r1));
  }
}
template <int F>
__device__ __forceinline__ void make_a_f(const f32x4& l0, const f32x4& l1, const f32x4& r0, const f32x4& r1, float one, u32x4& ah, u32x4& al) {
  unsigned h[4], q[4];
  split_pair_f<F>(l0[0], l0[1], r0[0], r0[1], one, h[0], q[0]);
  split_pair_f<F>(l0[2], l0[3], r0[2], r0[3], one, h[1], q[1]);
  split_pair_f<F>(l1[0], l1[1], r1[0], r1[1], one, h[2], q[2]);
  split_pair_f<F>(l1[2], l1[3], r1[2], r1[3], one, h[3], q[3]);
  ah = (u32x4){h[0], h[1], h[2], h[3]};
  al = (u32x4){q[0], q[1], q[2], q[3]};
}

// MF: 0 none, 1 = MFMAs on.  VA: 0 none, 1 = split then MFMAs (compiler order).  W waves per SIMD.
template <int MF, int VA, int W, int F>
__global__ __launch_bounds__(256 * W) void k(const float* __restrict__ in, float* __restrict__ out, int iters, float one) {
  __shared__ __attribute__((aligned(16))) float rs[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256 * W) rs[i] = in[i];
  const f32x4 l0 = *reinterpret_cast<const f32x4*>(in + tid * 8), l1 = *reinterpret_cast<const f32x4*>(in + tid * 8 + 4);
  u32x4 b[8];
  for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(in + 64 * i + lane * 4));
  f32x4 acc[4] = {};
  u32x4 ah = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, al = ah, vsum = {0, 0, 0, 0};
  __syncthreads();
  const float* rp = rs + 8 * (lane >> 4);
  f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
  constexpr int NM = (F == 15) ? 16 : 12;
  for (int it = 0; it < iters; ++it) {
    const float* rn = rs + ((it + 1) & 63) * 64 + 8 * (lane >> 4);
    const f32x4 n0 = *reinterpret_cast<const f32x4*>(rn), n1 = *reinterpret_cast<const f32x4*>(rn + 4);
    u32x4 nh = ah, nl = al;
    if (VA == 1) {
      make_a_f<F>(l0, l1, r0, r1, one, nh, nl);
      if (MF == 0) { vsum ^= nh; vsum ^= nl; }
    }
    if (VA == 0) __builtin_amdgcn_sched_barrier(0);
    if (VA == 2) {
      // software pipeline: the operands of iteration it+1 are formed (16 VALU) BETWEEN the 12 MFMAs of iteration it
      u32x4 ch = ah, cl = al;                 // current operands (formed during the previous iteration)
      make_a_f<F>(l0, l1, n0, n1, one, nh, nl);   // next operands from the prefetched R words
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const u32x4 a = (m >= 4 && m < 8) ? cl : ch;
        const u32x4 bb = b[(m < 8 ? 0 : 4) + (m & 3)];
        acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bb), acc[m & 3], 0, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // 2 VALU
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
      }
      ah = nh; al = nl;
    }
    if (MF == 1 && VA != 2) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const u32x4 a = (NM == 16) ? ((m & 4) ? nl : nh) : ((m >= 4 && m < 8) ? nl : nh);
        const u32x4 bb = b[((NM == 16) ? (m < 8 ? 0 : 4) : (m < 8 ? 0 : 4)) + (m & 3)];
        if (F == 10)
          acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), acc[m & 3], 0, 0, 0);
        else
          acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bb), acc[m & 3], 0, 0, 0);
      }
    }
    r0 = n0; r1 = n1;
  }
  out[blockIdx.x * 256 * W + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + __uint_as_float(vsum[0] ^ vsum[1] ^ vsum[2] ^ vsum[3] ^ ah[0] ^ al[1]);
}

template <int MF, int VA, int W, int F>
void run(const char* name, const float* in, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MF, VA, W, F>), dim3(256), dim3(256 * W), 0, 0, in, out, 100, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MF, VA, W, F>), dim3(256), dim3(256 * W), 0, 0, in, out, iters, 1.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("W=%d %-52s %8.1f ns/iter/SIMD = %6.1f per wave-iter\n", W, name, ms * 1e6 / iters, ms * 1e6 / iters / W);
}

template <int W>
void all(const float* in, float* out) {
  run<1, 0, W, 10>("12 x mfma16x16x32 bf16 only", in, out);
  run<1, 0, W, 11>("12 x mfma16x16x32 f16 only", in, out);
  run<1, 0, W, 15>("16 x mfma16x16x32 f16 only", in, out);
  run<0, 1, W, 10>("F10 bf16 perm split only (32 VALU)", in, out);
  run<1, 1, W, 10>("F10 split + 12 bf16 mfma", in, out);
  run<0, 1, W, 11>("F11 f16 pkrtz + fma_mix split only (20 VALU)", in, out);
  run<1, 1, W, 11>("F11 split + 12 f16 mfma", in, out);
  run<0, 1, W, 12>("F12 f16 pkrtz + cvt/sub/cvt_pk split only (32 VALU)", in, out);
  run<1, 1, W, 12>("F12 split + 12 f16 mfma", in, out);
  run<0, 1, W, 13>("F13 min_u32 + 2 perm only (16 VALU)", in, out);
  run<1, 1, W, 13>("F13 + 12 f16 mfma", in, out);
  run<0, 1, W, 14>("F14 f16 cvt_pk RNE + fma_mix split only (20 VALU)", in, out);
  run<1, 1, W, 14>("F14 split + 12 f16 mfma", in, out);
  run<0, 1, W, 15>("F15 min_u32 only (8 VALU)", in, out);
  run<1, 1, W, 15>("F15 + 16 f16 mfma", in, out);
  run<1, 2, W, 13>("F13 software-pipelined: 12 x (mfma + 1-2 VALU of the next operands)", in, out);
}

int main() {
  float *in, *out;
  CHECK(hipMalloc(&in, 65536 * 4));
  CHECK(hipMalloc(&out, 256 * 1024 * 4));
  static float h[65536];
  for (int i = 0; i < 65536; ++i) h[i] = 0.001f * (i % 977) + 0.5f;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  all<2>(in, out);
  all<1>(in, out);
  if (getenv("UB5_ALL")) all<3>(in, out);
  return 0;
}
