// Micro-benchmarks of the instruction mixes the Delta kernel is built from (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench.hip -o /tmp/ubench && /tmp/ubench
// Every kernel runs ITER iterations of an unrolled body in each wave and reports the wall cycles per body
// (s_memtime) averaged over the waves of the first workgroup, for 1, 2 and 3 waves per SIMD (block = 256/512/768
// threads, one workgroup per CU).
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

enum { T_FSUB = 0, T_AND, T_SUBABS, T_PERM, T_CVT, T_SPLIT, T_MFMA16, T_MFMA32, T_MIX16, T_MIX32, T_MIX32_SEP, T_LDSB, T_LDSBR, T_LDSBR_BAR, T_LDSBR_BAR_PF, T_COUNT };
static const char* NAMES[] = {"64 x v_sub_f32", "64 x v_and_b32", "64 x v_sub_f32 |a|", "64 x v_perm_b32", "32 x v_cvt_pk_bf16_f32",
                              "split 8 elems (make_a)", "12 x mfma16x16x32", "6 x mfma32x32x16",
                              "make_a + 12 mfma16 (3 terms x 4 ct)", "make_a + 6 mfma32 (3 terms x 2 ct)",
                              "make_a(next) + 6 mfma32 independent",
                              "make_a + 6 mfma32 + 4 B ds_read_b128 (prefetched)", "  ... + 2 R ds_read_b128 (broadcast)",
                              "  ... + barrier & 16B LDS write every 3 bodies", "  ... + 16B global load every 3 bodies"};

template <int TEST>
__global__ void ubench(const float* __restrict__ in, float* __restrict__ out, long long* __restrict__ cycles, int iters) {
  const int tid = threadIdx.x;
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = in[tid * 16 + i];
  f32x4 l0 = {x[0], x[1], x[2], x[3]}, l1 = {x[4], x[5], x[6], x[7]};
  f32x4 r0 = {x[8], x[9], x[10], x[11]}, r1 = {x[12], x[13], x[14], x[15]};
  bf16x8 bh = __builtin_bit_cast(bf16x8, l0), bl = __builtin_bit_cast(bf16x8, l1);
  f32x4 acc16[4];
  for (int i = 0; i < 4; ++i) acc16[i] = (f32x4){0, 0, 0, 0};
  f32x16 acc32[2];
  for (int r = 0; r < 16; ++r) { acc32[0][r] = 0; acc32[1][r] = 0; }
  bf16x8 ah = bh, al = bl;
  __shared__ __attribute__((aligned(16))) unsigned char ring[3 * 12288];
  __shared__ __attribute__((aligned(16))) float rsh[15 * 128];
  for (int i = tid; i < 3 * 12288 / 4; i += blockDim.x) reinterpret_cast<float*>(ring)[i] = in[(i * 7) % (768 * 16)];
  for (int i = tid; i < 15 * 128; i += blockDim.x) rsh[i] = in[(i * 3) % (768 * 16)];
  const int lane = tid & 63;
  const int kh = lane >> 5;
  bf16x8 bq[4];
  for (int i = 0; i < 4; ++i) bq[i] = bh;
  f32x4 rq0 = r0, rq1 = r1;
  f32x4 pf = l0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (TEST == T_FSUB) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = x[i] - 1.0009765625f;
    } else if (TEST == T_AND) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) { x[i] = __uint_as_float(__float_as_uint(x[i]) & 0x7fff0000u); asm volatile("" : "+v"(x[i])); }
    } else if (TEST == T_SUBABS) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fabsf(x[i]) - 1.0009765625f;
    } else if (TEST == T_PERM) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) { x[i] = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(x[i]), __float_as_uint(x[(i + 1) & 15]), 0x07060302u)); asm volatile("" : "+v"(x[i])); }
    } else if (TEST == T_CVT) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          bf16x2 p;
          p[0] = (__bf16)x[i];
          p[1] = (__bf16)x[i + 1];
          unsigned u = __builtin_bit_cast(unsigned, p);
          asm volatile("" : "+v"(u));
          x[i] = __uint_as_float(u);
        }
    } else if (TEST == T_SPLIT) {
      make_a(l0, l1, r0, r1, ah, al);
      asm volatile("" : "+v"(ah), "+v"(al));
      l0 = __builtin_bit_cast(f32x4, ah);
    } else if (TEST == T_MFMA16) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc16[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t == 1 ? al : ah, t == 2 ? bl : bh, acc16[n], 0, 0, 0);
    } else if (TEST == T_MFMA32) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc32[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 1 ? al : ah, t == 2 ? bl : bh, acc32[n], 0, 0, 0);
    } else if (TEST == T_MIX16) {
      make_a(l0, l1, r0, r1, ah, al);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc16[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t == 1 ? al : ah, t == 2 ? bl : bh, acc16[n], 0, 0, 0);
      l0[0] += 0.5f;  // new values every iteration
    } else if (TEST == T_MIX32) {
      make_a(l0, l1, r0, r1, ah, al);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc32[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 1 ? al : ah, t == 2 ? bl : bh, acc32[n], 0, 0, 0);
      l0[0] += 0.5f;
    } else if (TEST == T_MIX32_SEP) {
      // MFMAs use the fragments of the PREVIOUS iteration, the split of the next one is independent of them
      bf16x8 nh, nl;
      make_a(l0, l1, r0, r1, nh, nl);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc32[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t == 1 ? al : ah, t == 2 ? bl : bh, acc32[n], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
      ah = nh;
      al = nl;
      l0[0] += 0.5f;
    } else if (TEST >= T_LDSB) {
      const int step = it % 3, slot = (it / 3) % 3, dj = it % 15, sl = (it / 15) & 7;
      bf16x8 bn[4];
      const unsigned char* bp = ring + slot * 12288 + step * 4096 + lane * 16;
      bn[0] = *reinterpret_cast<const bf16x8*>(bp);
      bn[1] = *reinterpret_cast<const bf16x8*>(bp + 1024);
      bn[2] = *reinterpret_cast<const bf16x8*>(bp + 2048);
      bn[3] = *reinterpret_cast<const bf16x8*>(bp + 3072);
      f32x4 rn0 = rq0, rn1 = rq1;
      if (TEST >= T_LDSBR) {
        const float* rr = rsh + dj * 128 + 16 * sl + 8 * kh;
        rn0 = *reinterpret_cast<const f32x4*>(rr);
        rn1 = *reinterpret_cast<const f32x4*>(rr + 4);
      }
      if (TEST >= T_LDSBR_BAR_PF && step == 0) pf = *reinterpret_cast<const f32x4*>(in + ((it * 64 + tid) * 4) % (768 * 16 - 4));
      make_a(l0, l1, rq0, rq1, ah, al);
      acc32[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[0], acc32[0], 0, 0, 0);
      acc32[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[2], acc32[1], 0, 0, 0);
      acc32[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[0], acc32[0], 0, 0, 0);
      acc32[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[2], acc32[1], 0, 0, 0);
      acc32[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[1], acc32[0], 0, 0, 0);
      acc32[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[3], acc32[1], 0, 0, 0);
      bq[0] = bn[0]; bq[1] = bn[1]; bq[2] = bn[2]; bq[3] = bn[3];
      rq0 = rn0; rq1 = rn1;
      if (TEST >= T_LDSBR_BAR && step == 2) {
        *reinterpret_cast<f32x4*>(ring + ((slot + 2) % 3) * 12288 + (tid % 768) * 16) = pf;
        __syncthreads();
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  s += l0[0] + l0[1] + (float)ah[0] + (float)al[1] + pf[0] + rq0[1] + (float)bq[0][0];
  for (int i = 0; i < 4; ++i) s += acc16[i][0] + acc16[i][3];
  s += acc32[0][0] + acc32[1][5];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (blockIdx.x == 0 && (tid & 63) == 0) cycles[tid >> 6] = t1 - t0;
}

template <int TEST>
void run(const float* in, float* out, long long* cyc) {
  const int iters = 2000;
  for (int wps = 1; wps <= 3; ++wps) {
    const int block = 256 * wps;
    hipLaunchKernelGGL(ubench<TEST>, dim3(256), dim3(block), 0, 0, in, out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    long long h[12];
    CHECK(hipMemcpy(h, cyc, sizeof(long long) * 4 * wps, hipMemcpyDeviceToHost));
    double avg = 0;
    for (int i = 0; i < 4 * wps; ++i) avg += (double)h[i];
    avg /= (4 * wps) * (double)iters;
    printf("%-42s waves/SIMD %d: %8.1f clk per body per wave  (%7.1f clk per body per SIMD)\n", NAMES[TEST], wps, avg, avg / wps);
  }
}

int main() {
  float *in, *out;
  long long* cyc;
  CHECK(hipMalloc(&in, 768 * 16 * 4));
  CHECK(hipMalloc(&out, 256 * 768 * 4));
  CHECK(hipMalloc(&cyc, 64 * 8));
  float h[768 * 16];
  for (int i = 0; i < 768 * 16; ++i) h[i] = 0.001f * (i % 977) + 0.5f;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  run<T_FSUB>(in, out, cyc);
  run<T_AND>(in, out, cyc);
  run<T_SUBABS>(in, out, cyc);
  run<T_PERM>(in, out, cyc);
  run<T_CVT>(in, out, cyc);
  run<T_SPLIT>(in, out, cyc);
  run<T_MFMA16>(in, out, cyc);
  run<T_MFMA32>(in, out, cyc);
  run<T_MIX16>(in, out, cyc);
  run<T_MIX32>(in, out, cyc);
  run<T_MIX32_SEP>(in, out, cyc);
  run<T_LDSB>(in, out, cyc);
  run<T_LDSBR>(in, out, cyc);
  run<T_LDSBR_BAR>(in, out, cyc);
  run<T_LDSBR_BAR_PF>(in, out, cyc);
  return 0;
}
