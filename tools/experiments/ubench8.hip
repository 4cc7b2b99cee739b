// Round 5: would a K-COMPACTED contraction step pay?  In a 1-vs-N sweep the query's packed words r'[j][c] are shared by all candidates and
// ~48 % of them are exactly zero (ReLU features): min(l', 0) = 0 for every candidate row, so those K entries could be dropped from the
// contraction once per query (MFMA count x 0.53-0.57).  The price: the L words of a step are no longer 8 fixed channels per lane held in
// registers for 15 taps -- they must be GATHERED per step from the wave's L slice in LDS (channel list of the step: 8 byte indices per
// lane group), and the two column groups of a pass no longer share their W1 fragments (each has its own compacted weight rows).
//   dense : the kernel's step as in ubench7 (6 operand formations from register L words, 8 B-fragment reads, 72 MFMAs)
//   sparse: two compacted column-group steps = 72 MFMAs: per step 1 index read + 16 VALU + 24 ds_read_b32 (L gather from a [channel][row]
//           slice) + 2 R reads + 8 B-fragment reads + 3 operand formations + 36 MFMAs
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/experiments/ubench8.hip -o tools/bin/ubench8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
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

// LDS (words): [0, 12288) L slices of the 8 waves, [channel 32][row 48]; [12288, 16384) R words; [16384, 32768) B fragments (8 steps x 8 KB);
// [32768, 33280) channel-index table: 16 steps x 4 lane groups x 8 bytes
template <int SPARSE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned rs[33280];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32768; i += 512) rs[i] = __float_as_uint(in[i & 65535]) & 0x7fff7fffu;
  for (int i = tid; i < 512; i += 512) rs[32768 + i] = ((i * 7) & 31) | (((i * 11 + 3) & 31) << 8) | (((i * 13 + 5) & 31) << 16) | (((i * 17 + 9) & 31) << 24);
  __syncthreads();
  float s = 0.f;
  f32x4 acc[2][3][4] = {};
  if (SPARSE == 0) {
    u32x4 la[3][2];
    for (int t = 0; t < 3; ++t)
      for (int h = 0; h < 2; ++h) la[t][h] = *reinterpret_cast<const u32x4*>(rs + ((tid * 6 + t * 2 + h) * 4) % 12288);
    for (int it = 0; it < iters; ++it) {
      const unsigned* wb = rs + 16384 + ((it & 7) * 8) * 256 + lane * 4;
      const unsigned* rr = rs + 12288 + ((it & 15) * 4) * 32 + 8 * (lane >> 4);
      f16x8 bh[4], bl[4];
      u32x4 rw[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bh[nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb + (2 * nt) * 256));
        bl[nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb + (2 * nt + 1) * 256));
      }
      rw[0] = *reinterpret_cast<const u32x4*>(rr);
      rw[1] = *reinterpret_cast<const u32x4*>(rr + 4);
      rw[2] = *reinterpret_cast<const u32x4*>(rr + 1024);
      rw[3] = *reinterpret_cast<const u32x4*>(rr + 1028);
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
  } else {
    // this lane's row of tile 0 in the wave's [channel][row] slice: word offset wave * 1536 + lrow; tiles 16 rows apart; channels 48 words apart
    const unsigned* lmine = rs + wave * 1536 + (lane & 15);
    const int g = lane >> 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int step = (2 * it + j) & 15;
        const unsigned* wb = rs + 16384 + ((step & 7) * 8) * 256 + lane * 4;
        const unsigned* rr = rs + 12288 + (step * 4) * 32 + 8 * g;
        const u32x2 ix = *reinterpret_cast<const u32x2*>(rs + 32768 + (step * 4 + g) * 2);
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          bh[nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb + (2 * nt) * 256));
          bl[nt] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(wb + (2 * nt + 1) * 256));
        }
        const u32x4 r0 = *reinterpret_cast<const u32x4*>(rr), r1 = *reinterpret_cast<const u32x4*>(rr + 4);
        unsigned lw[3][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned c = __builtin_amdgcn_ubfe(e < 4 ? ix[0] : ix[1], 8 * (e & 3), 5);
          const unsigned* p = lmine + c * 48;
#pragma unroll
          for (int t = 0; t < 3; ++t) lw[t][e] = p[16 * t];
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          f16x8 ah, al;
          const u32x4 l0 = {lw[t][0], lw[t][1], lw[t][2], lw[t][3]}, l1 = {lw[t][4], lw[t][5], lw[t][6], lw[t][7]};
          make_a(l0, l1, r0, r1, ah, al);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], acc[j][t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nt], acc[j][t][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], acc[j][t][nt], 0, 0, 0);
        }
      }
    }
  }
  for (int j = 0; j < 2; ++j)
    for (int t = 0; t < 3; ++t)
      for (int nt = 0; nt < 4; ++nt) s += acc[j][t][nt][0] + acc[j][t][nt][3];
  out[blockIdx.x * 512 + tid] = s;
}

template <int SPARSE>
void run(const char* name, const float* in, float* out) {
  const int iters = 8000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<SPARSE>), dim3(256), dim3(512), 0, 0, in, out, 100);
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<SPARSE>), dim3(256), dim3(512), 0, 0, in, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tf = 72.0 * 16384.0 * iters * 2 * 4 * 256 / (ms * 1e-3) / 1e12;
    printf("%-58s %8.1f ns per 72 MFMAs per SIMD-pair  %7.1f TF executed (%.0f %% of 2.5 PF)\n", name, ms * 1e6 / iters, tf, tf / 25.0);
  }
}

int main() {
  float *in, *out;
  CHECK(hipMalloc(&in, 65536 * 4));
  CHECK(hipMalloc(&out, 256 * 512 * 4));
  static float h[65536];
  for (int i = 0; i < 65536; ++i) h[i] = 0.001f * ((i * 7919) % 977) + 0.5f;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  run<0>("dense step (L words in registers, shared B fragments)", in, out);
  run<1>("2 compacted steps (L gathered from LDS, own B fragments)", in, out);
  run<0>("dense step (L words in registers, shared B fragments)", in, out);
  run<1>("2 compacted steps (L gathered from LDS, own B fragments)", in, out);
  return 0;
}
