// Loop-structure experiments for the 12-wave Delta kernel (gfx950): the real step body (split + 6 MFMA 32x32x16)
// with its operand traffic switched on piece by piece.  768 threads (3 waves/SIMD), one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench2.hip -o tools/bin/ubench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void split_pair(float d0, float d1, unsigned& hi_pk, unsigned& lo_pk) {
  const unsigned h0 = __float_as_uint(d0) & 0x7fff0000u;
  const unsigned h1 = __float_as_uint(d1) & 0x7fff0000u;
  const float l0 = fabsf(d0) - __uint_as_float(h0);
  const float l1 = fabsf(d1) - __uint_as_float(h1);
  hi_pk = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 lp;
  lp[0] = (__bf16)l0;
  lp[1] = (__bf16)l1;
  lo_pk = __builtin_bit_cast(unsigned, lp);
}
__device__ __forceinline__ void make_a(const f32x4& l0, const f32x4& l1, const f32x4& r0, const f32x4& r1, bf16x8& ah, bf16x8& al) {
  unsigned h0, h1, h2, h3, q0, q1, q2, q3;
  split_pair(l0[0] - r0[0], l0[1] - r0[1], h0, q0);
  split_pair(l0[2] - r0[2], l0[3] - r0[3], h1, q1);
  split_pair(l1[0] - r1[0], l1[1] - r1[1], h2, q2);
  split_pair(l1[2] - r1[2], l1[3] - r1[3], h3, q3);
  ah = __builtin_bit_cast(bf16x8, (u32x4){h0, h1, h2, h3});
  al = __builtin_bit_cast(bf16x8, (u32x4){q0, q1, q2, q3});
}

constexpr int STEP_BYTES = 4096;
// USE_B: B fragments from the LDS ring (prefetched one step ahead); USE_R: R fragment from LDS (prefetched);
// SPC: steps per ring chunk (barrier + publication every SPC steps; 0 = never); STAGE: 0 none, 1 global->VGPR->LDS, 2 LDS-DMA
template <bool USE_B, bool USE_R, int SPC, int STAGE>
__global__ __launch_bounds__(768) void loopk(const float* __restrict__ in, const unsigned char* __restrict__ w, float* __restrict__ out,
                                              long long* __restrict__ cycles, int njb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CHUNK = (SPC > 0 ? SPC : 3) * STEP_BYTES;
  constexpr int PER_THREAD = CHUNK / (768 * 16);  // 16-B pieces per thread per chunk
  unsigned char* ring = smem;                 // 3 chunks
  float* rs = reinterpret_cast<float*>(smem + 3 * CHUNK);
  const int tid = threadIdx.x, lane = tid & 63, kh = lane >> 5;
  for (int i = tid; i < 3 * CHUNK / 4; i += 768) reinterpret_cast<float*>(ring)[i] = in[(i * 7) % 12000];
  for (int i = tid; i < 15 * 128; i += 768) rs[i] = in[(i * 3) % 12000];
  f32x4 la[2] = {*reinterpret_cast<const f32x4*>(in + tid * 8), *reinterpret_cast<const f32x4*>(in + tid * 8 + 4)};
  f32x16 acc[2];
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0; acc[1][r] = 0; }
  bf16x8 bq[4];
  for (int i = 0; i < 4; ++i) bq[i] = __builtin_bit_cast(bf16x8, la[i & 1]);
  f32x4 rq0 = la[1], rq1 = la[0];
  f32x4 pf[PER_THREAD > 0 ? PER_THREAD : 1];
  for (int i = 0; i < (PER_THREAD > 0 ? PER_THREAD : 1); ++i) pf[i] = la[0];
  int cb = 0, chunk = 0;
  constexpr int NCH = 120 / (SPC > 0 ? SPC : 3);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int jb = 0; jb < njb; ++jb) {
#pragma unroll 1
    for (int sl = 0; sl < 8; ++sl) {
      constexpr int SP = (SPC > 0 ? SPC : 3);
#pragma unroll 1
      for (int c = 0; c < 15 / SP; ++c) {
        int nb = cb + 1; if (nb == 3) nb = 0;
        int wb = cb + 2; if (wb >= 3) wb -= 3;
        int c2 = chunk + 2; if (c2 >= NCH) c2 -= NCH;
        if (STAGE == 1) {
#pragma unroll
          for (int q = 0; q < PER_THREAD; ++q) pf[q] = *reinterpret_cast<const f32x4*>(w + (size_t)c2 * CHUNK + (q * 768 + tid) * 16);
        } else if (STAGE == 2) {
#pragma unroll
          for (int q = 0; q < PER_THREAD; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + (size_t)c2 * CHUNK + (q * 768 + tid) * 16),
                                             (__attribute__((address_space(3))) void*)(ring + wb * CHUNK + (q * 768 + (tid & ~63)) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int hh = 0; hh < SP; ++hh) {
          const int dj = c * SP + hh;
          bf16x8 bn[4] = {bq[0], bq[1], bq[2], bq[3]};
          if (USE_B) {
            const unsigned char* bp = (hh + 1 < SP) ? ring + cb * CHUNK + (hh + 1) * STEP_BYTES + lane * 16 : ring + nb * CHUNK + lane * 16;
            bn[0] = *reinterpret_cast<const bf16x8*>(bp);
            bn[1] = *reinterpret_cast<const bf16x8*>(bp + 1024);
            bn[2] = *reinterpret_cast<const bf16x8*>(bp + 2048);
            bn[3] = *reinterpret_cast<const bf16x8*>(bp + 3072);
          }
          f32x4 rn0 = rq0, rn1 = rq1;
          if (USE_R) {
            const float* rr = rs + ((dj + 1 < 15) ? (dj + 1) * 128 + 16 * sl : 16 * ((sl + 1) & 7)) + 8 * kh;
            rn0 = *reinterpret_cast<const f32x4*>(rr);
            rn1 = *reinterpret_cast<const f32x4*>(rr + 4);
          }
          bf16x8 ah, al;
          make_a(la[0], la[1], rq0, rq1, ah, al);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[0], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[2], acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[0], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[2], acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[1], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[3], acc[1], 0, 0, 0);
          bq[0] = bn[0]; bq[1] = bn[1]; bq[2] = bn[2]; bq[3] = bn[3];
          rq0 = rn0; rq1 = rn1;
          if (!USE_R) la[0][0] += 0.5f;
        }
        if (SPC > 0) {
          if (STAGE == 1) {
#pragma unroll
            for (int q = 0; q < PER_THREAD; ++q) *reinterpret_cast<f32x4*>(ring + wb * CHUNK + (q * 768 + tid) * 16) = pf[q];
          }
          __syncthreads();
          cb = nb;
          ++chunk; if (chunk == NCH) chunk = 0;
        }
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 768 + tid] = acc[0][0] + acc[1][7] + rq0[0] + (float)bq[0][0] + la[0][0] + pf[0][0];
  if (blockIdx.x == 0 && lane == 0) cycles[tid >> 6] = t1 - t0;
}

template <bool USE_B, bool USE_R, int SPC, int STAGE>
void run(const char* name, const float* in, const unsigned char* w, float* out, long long* cyc) {
  const int njb = 4;
  constexpr int CHUNK = (SPC > 0 ? SPC : 3) * STEP_BYTES;
  const size_t lds = 3 * CHUNK + 15 * 128 * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(loopk<USE_B, USE_R, SPC, STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((loopk<USE_B, USE_R, SPC, STAGE>), dim3(256), dim3(768), lds, 0, in, w, out, cyc, njb);
  CHECK(hipDeviceSynchronize());
  long long h[12];
  CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double avg = 0;
  for (int i = 0; i < 12; ++i) avg += (double)h[i];
  avg /= 12.0 * njb * 120;
  printf("%-66s %7.1f clk per step per wave (%6.1f per SIMD)\n", name, avg, avg / 3);
}

int main() {
  float *in, *out;
  unsigned char* w;
  long long* cyc;
  CHECK(hipMalloc(&in, 12288 * 4));
  CHECK(hipMalloc(&out, 256 * 768 * 4));
  CHECK(hipMalloc(&cyc, 64 * 8));
  CHECK(hipMalloc(&w, 120 * 4096));
  float h[12288];
  for (int i = 0; i < 12288; ++i) h[i] = 0.001f * (i % 977) + 0.5f;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  CHECK(hipMemset(w, 0x3c, 120 * 4096));
  run<false, false, 0, 0>("core: split + 6 mfma32", in, w, out, cyc);
  run<true, false, 0, 0>("+ B from LDS (4 x b128, one step ahead)", in, w, out, cyc);
  run<true, true, 0, 0>("+ R from LDS (2 x b128 broadcast, one step ahead)", in, w, out, cyc);
  run<true, true, 3, 0>("+ barrier every 3 steps", in, w, out, cyc);
  run<true, true, 3, 1>("+ stage 12 KB chunk via VGPRs every 3 steps", in, w, out, cyc);
  run<true, true, 3, 2>("  (same, LDS-DMA instead of VGPR staging)", in, w, out, cyc);
  run<true, true, 15, 0>("barrier every 15 steps (slice-sized chunk)", in, w, out, cyc);
  run<true, true, 15, 1>("  + stage 60 KB chunk via VGPRs every 15 steps", in, w, out, cyc);
  run<true, true, 15, 2>("  + stage 60 KB chunk via LDS-DMA every 15 steps", in, w, out, cyc);
  run<false, true, 0, 0>("core + R only", in, w, out, cyc);
  return 0;
}
