// Round 3: one K = 32 step of the contraction kernel per wave, in the two MFMA shapes (see ubench6.hip for the naive comparison,
// whose 32x32 arm was limited by having only two accumulator chains).
//   A: 6 operand formations (3 row tiles of 16 x 2 column groups) -> 72 x mfma_f32_16x16x32_f16 on 24 accumulators (96 registers)
//   B: 6 operand formations (3 row tiles of 32 x 2 K halves)      -> 36 x mfma_f32_32x32x16_f16 on  6 accumulators (96 registers)
// both with 8 weight-fragment reads + 4 packed-R reads (ds_read_b128) per step, L words in registers (24 / 48).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/experiments/ubench7.hip -o tools/bin/ubench7
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void make_a(const u32x4& l0, const u32x4& l1, const u32x4& r0, const u32x4& r1, f16x8& ah, f16x8& al) {
  u32x4 h, q;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const unsigned m0 = min(l0[2 * p], r0[2 * p]), m1 = min(l0[2 * p + 1], r0[2 * p + 1]);
    h[p] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    q[p] = __builtin_amdgcn_perm(m1, m0, 0x05040100u);
    const unsigned n0 = min(l1[2 * p], r1[2 * p]), n1 = min(l1[2 * p + 1], r1[2 * p + 1]);
    h[2 + p] = __builtin_amdgcn_perm(n1, n0, 0x07060302u);
    q[2 + p] = __builtin_amdgcn_perm(n1, n0, 0x05040100u);
  }
  ah = __builtin_bit_cast(f16x8, h);
  al = __builtin_bit_cast(f16x8, q);
}

template <int BIG, int W>
__global__ __launch_bounds__(256 * W) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned rs[24576];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 24576; i += 256 * W) rs[i] = __float_as_uint(in[i]) & 0x7fff7fffu;
  __syncthreads();
  float s = 0.f;
  if (BIG == 0) {
    u32x4 la[3][2];
    for (int t = 0; t < 3; ++t)
      for (int h = 0; h < 2; ++h) la[t][h] = *reinterpret_cast<const u32x4*>(rs + (tid * 6 + t * 2 + h) * 4);
    f32x4 acc[2][3][4] = {};
    for (int it = 0; it < iters; ++it) {
      const unsigned* wb = rs + 8192 + ((it & 7) * 8) * 256 + lane * 4;
      const unsigned* rr = rs + ((it & 15) * 4) * 64 + 8 * (lane >> 4);
      f16x8 bh[4], bl[4];
      u32x4 rw[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bh[nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb + (2 * nt) * 256));
        bl[nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb + (2 * nt + 1) * 256));
      }
      rw[0] = *reinterpret_cast<const u32x4*>(rr);
      rw[1] = *reinterpret_cast<const u32x4*>(rr + 4);
      rw[2] = *reinterpret_cast<const u32x4*>(rr + 2048);
      rw[3] = *reinterpret_cast<const u32x4*>(rr + 2052);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f16x8 ah, al;
          make_a(la[t][0], la[t][1], rw[2 * j], rw[2 * j + 1], ah, al);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], acc[j][t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nt], acc[j][t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], acc[j][t][nt], 0, 0, 0);
        }
    }
    for (int j = 0; j < 2; ++j)
      for (int t = 0; t < 3; ++t)
        for (int nt = 0; nt < 4; ++nt) s += acc[j][t][nt][0] + acc[j][t][nt][3];
  } else {
    u32x4 la[3][2][2];   // [tile of 32 rows][K half][2 x 4 words]
    for (int t = 0; t < 3; ++t)
      for (int kh = 0; kh < 2; ++kh)
        for (int h = 0; h < 2; ++h) la[t][kh][h] = *reinterpret_cast<const u32x4*>(rs + ((tid * 12 + (t * 2 + kh) * 2 + h) * 4) % 8192);
    f32x16 acc[3][2] = {};
    for (int it = 0; it < iters; ++it) {
      const unsigned* wb = rs + 8192 + ((it & 7) * 8) * 256 + lane * 4;
      const unsigned* rr = rs + ((it & 15) * 4) * 64 + 8 * (lane >> 5);
      f16x8 bh[2][2], bl[2][2];   // [K half][n-tile]
      u32x4 rw[2][2];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          bh[kh][nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb + ((kh * 2 + nt) * 2) * 256));
          bl[kh][nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb + ((kh * 2 + nt) * 2 + 1) * 256));
        }
        rw[kh][0] = *reinterpret_cast<const u32x4*>(rr + kh * 16);
        rw[kh][1] = *reinterpret_cast<const u32x4*>(rr + kh * 16 + 4);
      }
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          f16x8 ah, al;
          make_a(la[t][kh][0], la[t][kh][1], rw[kh][0], rw[kh][1], ah, al);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[kh][nt], acc[t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[kh][nt], acc[t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[kh][nt], acc[t][nt], 0, 0, 0);
        }
    }
    for (int t = 0; t < 3; ++t)
      for (int nt = 0; nt < 2; ++nt)
        for (int i = 0; i < 16; ++i) s += acc[t][nt][i];
  }
  out[blockIdx.x * 256 * W + tid] = s;
}

template <int BIG, int W>
void run(const char* name, const float* in, float* out) {
  const int iters = 8000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<BIG, W>), dim3(256), dim3(256 * W), 0, 0, in, out, 100);
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<BIG, W>), dim3(256), dim3(256 * W), 0, 0, in, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tf = 72.0 * 16384.0 * iters * W * 4 * 256 / (ms * 1e-3) / 1e12;
    printf("W=%d %-44s %8.1f ns/step/SIMD  %7.1f TF executed (%.0f %% of 2.5 PF)\n", W, name, ms * 1e6 / iters, tf, tf / 25.0);
  }
}

int main() {
  float *in, *out;
  CHECK(hipMalloc(&in, 65536 * 4));
  CHECK(hipMalloc(&out, 256 * 1024 * 4));
  static float h[65536];
  for (int i = 0; i < 65536; ++i) h[i] = 0.001f * ((i * 7919) % 977) + 0.5f;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  run<0, 2>("72 x 16x16x32, step of the kernel", in, out);
  run<1, 2>("36 x 32x32x16, same step", in, out);
  run<0, 1>("72 x 16x16x32, step of the kernel", in, out);
  run<1, 1>("36 x 32x32x16, same step", in, out);
  return 0;
}
