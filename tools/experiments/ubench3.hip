// Can the |L-R| bf16 split (VALU) hide behind the MFMAs that consume it?  Register/LDS-only loop, gfx950.
// Per iteration and wave: one A fragment pair (36 VALU instructions, inputs: L in registers, R from LDS at a
// moving address so nothing is loop-invariant) and the 12 x 16x16x32 (or 6 x 32x32x16) MFMAs it feeds.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/experiments/ubench3.hip -o tools/bin/ubench3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct SplitTmp { float d0, d1, q0, q1; unsigned h0, h1; };
__device__ __forceinline__ void split_stage(int stage, float a0, float b0, float a1, float b1, SplitTmp& s, unsigned& hi_pk, unsigned& lo_pk) {
  if (stage == 0) {
    s.d0 = a0 - b0; s.d1 = a1 - b1; s.h0 = __float_as_uint(s.d0) & 0x7fff0000u;
  } else if (stage == 1) {
    s.h1 = __float_as_uint(s.d1) & 0x7fff0000u;
    s.q0 = fabsf(s.d0) - __uint_as_float(s.h0);
    s.q1 = fabsf(s.d1) - __uint_as_float(s.h1);
  } else {
    hi_pk = (s.h0 >> 16) | s.h1;
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 lp; lp[0] = (__bf16)s.q0; lp[1] = (__bf16)s.q1;
    lo_pk = __builtin_bit_cast(unsigned, lp);
  }
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// F: 1 = v_perm pack of hi; 2 = hi by RNE cvt_pk with |.| modifiers; 3 = F1 + packed-f32 subtract; 5 = F2 + packed-f32 subtract
template <int F>
__device__ __forceinline__ void split_pair_f(float l0, float l1, float r0, float r1, unsigned& hi_pk, unsigned& lo_pk) {
  float d0, d1;
  if (F == 6) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"((f32x2){l0, l1}), "v"((f32x2){r0, r1}));
    d0 = d[0]; d1 = d[1];
  } else if (F == 3 || F == 5) {
    const f32x2 d = (f32x2){l0, l1} - (f32x2){r0, r1};
    d0 = d[0]; d1 = d[1];
  } else {
    d0 = l0 - r0; d1 = l1 - r1;
  }
  float q0, q1;
  if (F == 1 || F == 3 || F == 6) {
    const unsigned h0 = __float_as_uint(d0) & 0x7fff0000u, h1 = __float_as_uint(d1) & 0x7fff0000u;
    q0 = fabsf(d0) - __uint_as_float(h0);
    q1 = fabsf(d1) - __uint_as_float(h1);
    hi_pk = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  } else {
    bf16x2_t hp;
    hp[0] = (__bf16)fabsf(d0);
    hp[1] = (__bf16)fabsf(d1);
    hi_pk = __builtin_bit_cast(unsigned, hp);
    q0 = fabsf(d0) - __uint_as_float(hi_pk << 16);
    q1 = fabsf(d1) - __uint_as_float(hi_pk & 0xffff0000u);
  }
  bf16x2_t lp;
  lp[0] = (__bf16)q0;
  lp[1] = (__bf16)q1;
  lo_pk = __builtin_bit_cast(unsigned, lp);
}
template <int F>
__device__ __forceinline__ void make_a_f(const f32x4& l0, const f32x4& l1, const f32x4& r0, const f32x4& r1, u32x4& ah, u32x4& al) {
  unsigned h[4], q[4];
  split_pair_f<F>(l0[0], l0[1], r0[0], r0[1], h[0], q[0]);
  split_pair_f<F>(l0[2], l0[3], r0[2], r0[3], h[1], q[1]);
  split_pair_f<F>(l1[0], l1[1], r1[0], r1[1], h[2], q[2]);
  split_pair_f<F>(l1[2], l1[3], r1[2], r1[3], h[3], q[3]);
  ah = (u32x4){h[0], h[1], h[2], h[3]};
  al = (u32x4){q[0], q[1], q[2], q[3]};
}
__device__ __forceinline__ void make_a(const f32x4& l0, const f32x4& l1, const f32x4& r0, const f32x4& r1, u32x4& ah, u32x4& al) {
  SplitTmp s;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const f32x4 lv = p < 2 ? l0 : l1, rv = p < 2 ? r0 : r1;
    unsigned hw = 0, lw = 0;
#pragma unroll
    for (int st = 0; st < 3; ++st) split_stage(st, lv[2 * (p & 1)], rv[2 * (p & 1)], lv[2 * (p & 1) + 1], rv[2 * (p & 1) + 1], s, hw, lw);
    ah[p] = hw;
    al[p] = lw;
  }
}

// MF: 0 none, 1 = 12 x mfma16x16x32, 2 = 6 x mfma32x32x16.  VA: 0 none, 1 = split then MFMAs (compiler order), 2 = one split stage after each MFMA.
template <int MF, int VA, int W, int F = 0>
__global__ __launch_bounds__(256 * W) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) float rs[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256 * W) rs[i] = in[i];
  const f32x4 l0 = *reinterpret_cast<const f32x4*>(in + tid * 8), l1 = *reinterpret_cast<const f32x4*>(in + tid * 8 + 4);
  bf16x8 b[8];
  for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(in + 64 * i + lane * 4));
  f32x4 acc[4] = {};
  f32x16 acc32[2] = {};
  u32x4 ah = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, al = ah, vsum = {0, 0, 0, 0};
  __syncthreads();
  const float* rp = rs + 8 * (lane >> 4);
  f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
  for (int it = 0; it < iters; ++it) {
    const float* rn = rs + ((it + 1) & 63) * 64 + 8 * (lane >> 4);
    const f32x4 n0 = *reinterpret_cast<const f32x4*>(rn), n1 = *reinterpret_cast<const f32x4*>(rn + 4);
    u32x4 nh = ah, nl = al;
    if (VA == 1) {
      if (F == 0) make_a(l0, l1, r0, r1, nh, nl); else make_a_f<F>(l0, l1, r0, r1, nh, nl);
      if (MF == 0) { vsum ^= nh; vsum ^= nl; }
    }
    const bf16x8 ahb = __builtin_bit_cast(bf16x8, VA == 1 ? nh : ah), alb = __builtin_bit_cast(bf16x8, VA == 1 ? nl : al);
    if (VA == 2 || VA == 0) __builtin_amdgcn_sched_barrier(0);
    SplitTmp st;
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      if (MF == 1) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((m >= 4 && m < 8) ? alb : ahb, b[(m < 8 ? 0 : 4) + (m & 3)], acc[m & 3], 0, 0, 0);
      } else if (MF == 2 && (m & 1) == 0) {
        const int q = m >> 1;
        acc32[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((q == 2 || q == 3) ? alb : ahb, b[(q < 4 ? 0 : 4) + (q & 1)], acc32[q & 1], 0, 0, 0);
      }
      if (VA == 2) {
        const int pr = m / 3;
        const f32x4 lv = pr < 2 ? l0 : l1, rv = pr < 2 ? r0 : r1;
        unsigned hw = 0, lw = 0;
        split_stage(m % 3, lv[2 * (pr & 1)], rv[2 * (pr & 1)], lv[2 * (pr & 1) + 1], rv[2 * (pr & 1) + 1], st, hw, lw);
        if (m % 3 == 2) {
          nh[pr] = hw;
          nl[pr] = lw;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (VA == 2) {
      if (MF == 0) { vsum ^= nh; vsum ^= nl; }
      ah = nh; al = nl;
    }
    r0 = n0; r1 = n1;
  }
  out[blockIdx.x * 256 * W + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc32[0][0] + acc32[1][5] + __uint_as_float(vsum[0] ^ vsum[1] ^ vsum[2] ^ vsum[3] ^ ah[0] ^ al[1]);
}

template <int MF, int VA, int W, int F = 0>
void run(const char* name, const float* in, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MF, VA, W, F>), dim3(256), dim3(256 * W), 0, 0, in, out, 100);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MF, VA, W, F>), dim3(256), dim3(256 * W), 0, 0, in, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("W=%d %-46s %8.1f ns/iter/SIMD = %6.1f per wave-iter\n", W, name, ms * 1e6 / iters, ms * 1e6 / iters / W);
}

template <int W>
void all(const float* in, float* out) {
  run<1, 0, W>("12 x mfma16x16x32 only", in, out);
  run<2, 0, W>("6 x mfma32x32x16 only", in, out);
  run<0, 1, W>("split only (36 VALU)", in, out);
  run<1, 1, W>("split, then 12 x mfma16", in, out);
  run<1, 2, W>("12 x (mfma16 + 3 VALU of the next split)", in, out);
  run<2, 1, W>("split, then 6 x mfma32", in, out);
  run<2, 2, W>("6 x (mfma32 + 6 VALU of the next split)", in, out);
}

int main() {
  float *in, *out;
  CHECK(hipMalloc(&in, 65536 * 4));
  CHECK(hipMalloc(&out, 256 * 1024 * 4));
  static float h[65536];
  for (int i = 0; i < 65536; ++i) h[i] = 0.001f * (i % 977) + 0.5f;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  if (getenv("UB3_ALL")) {
    all<1>(in, out);
    all<2>(in, out);
    all<3>(in, out);
    all<4>(in, out);
  }
  run<0, 1, 2, 0>("F0 split only", in, out);
  run<1, 1, 2, 0>("F0 split + 12 mfma16", in, out);
  run<0, 1, 2, 1>("F1 (perm pack) split only", in, out);
  run<1, 1, 2, 1>("F1 split + 12 mfma16", in, out);
  run<0, 1, 2, 2>("F2 (RNE hi by cvt_pk) split only", in, out);
  run<1, 1, 2, 2>("F2 split + 12 mfma16", in, out);
  run<0, 1, 2, 6>("F6 (perm + asm v_pk_add_f32) split only", in, out);
  run<1, 1, 2, 6>("F6 split + 12 mfma16", in, out);
  return 0;
}
