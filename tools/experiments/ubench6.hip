// Round 3: would the contraction kernel gain from v_mfma_f32_32x32x16_f16 instead of v_mfma_f32_16x16x32_f16?
// Same harness as ubench5.hip, min-form operand formation (F13: 8 v_min_u32 + 8 v_perm per 8 packed words and column group) next to
//   A: 12 x mfma_f32_16x16x32_f16 (4 accumulators of 4 registers)        = 196,608 FLOP
//   B:  6 x mfma_f32_32x32x16_f16 (2 accumulators of 16 registers)       = 196,608 FLOP
// each alone, with the VALU, and with 8 LDS fragment reads (ds_read_b128) per iteration as in the kernel.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/experiments/ubench6.hip -o tools/bin/ubench6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void make_a(const u32x4& l0, const u32x4& l1, const u32x4& r0, const u32x4& r1, u32x4& ah, u32x4& al) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const unsigned m0 = min(l0[2 * p], r0[2 * p]), m1 = min(l0[2 * p + 1], r0[2 * p + 1]);
    ah[p] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    al[p] = __builtin_amdgcn_perm(m1, m0, 0x05040100u);
    const unsigned n0 = min(l1[2 * p], r1[2 * p]), n1 = min(l1[2 * p + 1], r1[2 * p + 1]);
    ah[2 + p] = __builtin_amdgcn_perm(n1, n0, 0x07060302u);
    al[2 + p] = __builtin_amdgcn_perm(n1, n0, 0x05040100u);
  }
}

// BIG: 0 = 16x16x32, 1 = 32x32x16.  VA: operand formation on.  LD: B fragments re-read from LDS every iteration (8 x ds_read_b128).
template <int BIG, int VA, int LD, int W>
__global__ __launch_bounds__(256 * W) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned rs[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 256 * W) rs[i] = __float_as_uint(in[i]);
  const u32x4 l0 = *reinterpret_cast<const u32x4*>(in + tid * 8), l1 = *reinterpret_cast<const u32x4*>(in + tid * 8 + 4);
  u32x4 b[8];
  for (int i = 0; i < 8; ++i) b[i] = *reinterpret_cast<const u32x4*>(in + 64 * i + lane * 4);
  f32x4 acc[4] = {};
  f32x16 big[2] = {};
  u32x4 ah = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, al = ah;
  __syncthreads();
  const unsigned* rp = rs + 8 * (lane >> 4);
  u32x4 r0 = *reinterpret_cast<const u32x4*>(rp), r1 = *reinterpret_cast<const u32x4*>(rp + 4);
  for (int it = 0; it < iters; ++it) {
    const unsigned* rn = rs + ((it + 1) & 63) * 64 + 8 * (lane >> 4);
    const u32x4 n0 = *reinterpret_cast<const u32x4*>(rn), n1 = *reinterpret_cast<const u32x4*>(rn + 4);
    if (LD) {
      const unsigned* wb = rs + 4096 + ((it & 7) * 8) * 256 + lane * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i] = *reinterpret_cast<const u32x4*>(wb + i * 256);
    }
    u32x4 nh = ah, nl = al;
    if (VA) make_a(l0, l1, r0, r1, nh, nl);
    if (BIG == 0) {
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const u32x4 a = (m >= 4 && m < 8) ? nl : nh;
        const u32x4 bb = b[(m < 8 ? 0 : 4) + (m & 3)];
        acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bb), acc[m & 3], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 6; ++m) {   // (ah, bh) (al, bh) (ah, bl) x 2 n-tiles
        const u32x4 a = (m == 2 || m == 3) ? nl : nh;
        const u32x4 bb = b[(m < 4 ? 0 : 4) + (m & 1)];
        big[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bb), big[m & 1], 0, 0, 0);
      }
    }
    r0 = n0;
    r1 = n1;
    if (!VA) { ah[0] ^= n0[0]; }
  }
  float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  for (int i = 0; i < 16; ++i) s += big[0][i] + big[1][i];
  out[blockIdx.x * 256 * W + tid] = s + __uint_as_float(ah[0] ^ al[1]);
}

template <int BIG, int VA, int LD, int W>
void run(const char* name, const float* in, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<BIG, VA, LD, W>), dim3(256), dim3(256 * W), 0, 0, in, out, 100);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<BIG, VA, LD, W>), dim3(256), dim3(256 * W), 0, 0, in, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double tf = 196608.0 * iters * W * 4 * 256 / (ms * 1e-3) / 1e12;
  printf("W=%d %-64s %8.1f ns/iter/SIMD  %7.1f TF executed\n", W, name, ms * 1e6 / iters, tf);
}

template <int W>
void all(const float* in, float* out) {
  run<0, 0, 0, W>("12 x 16x16x32 only", in, out);
  run<1, 0, 0, W>(" 6 x 32x32x16 only", in, out);
  run<0, 1, 0, W>("12 x 16x16x32 + min-form operands (16 VALU)", in, out);
  run<1, 1, 0, W>(" 6 x 32x32x16 + min-form operands (16 VALU)", in, out);
  run<0, 1, 1, W>("12 x 16x16x32 + operands + 8 ds_read_b128", in, out);
  run<1, 1, 1, W>(" 6 x 32x32x16 + operands + 8 ds_read_b128", in, out);
}

int main() {
  float *in, *out;
  CHECK(hipMalloc(&in, 65536 * 4));
  CHECK(hipMalloc(&out, 256 * 1024 * 4));
  static float h[65536];
  for (int i = 0; i < 65536; ++i) h[i] = 0.001f * ((i * 7919) % 977) + 0.5f;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  all<2>(in, out);
  all<1>(in, out);
  return 0;
}
