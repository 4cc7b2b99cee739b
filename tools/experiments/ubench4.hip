// How many ds_read_b128 per MFMA can a wave pair sustain?  The step of conv_strip / c3_dense in isolation: NR conflict-free
// ds_read_b128 (issued one step ahead, sched_barrier-pinned) + NM v_mfma_f32_16x16x32_bf16, 8 waves per workgroup
// (2 per SIMD), one workgroup per CU.    hipcc --offload-arch=gfx950 -O3 tools/experiments/ubench4.hip -o tools/bin/ubench4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NR, int NM>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 512) reinterpret_cast<float*>(smem)[i] = in[i & 4095];
  __syncthreads();
  const unsigned char* base = smem + lane * 16;
  bf16x8 f[2][NR];
  f32x4 acc[8] = {};
  const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(in + lane * 4));
#pragma unroll
  for (int r = 0; r < NR; ++r) f[0][r] = *reinterpret_cast<const bf16x8*>(base + r * 1024);
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int off = ((it + u + 1) & 15) * 64;
#pragma unroll
      for (int r = 0; r < NR; ++r) f[u ^ 1][r] = *reinterpret_cast<const bf16x8*>(base + ((r * 1024 + off) & 0xffff));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NM; ++m)
        acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[u][m % (NR > 0 ? NR : 1)], b, acc[m & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + tid] = s;
}

template <int NR, int NM>
void run(const float* in, float* out) {
  const int iters = 20000;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<NR, NM>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<NR, NM>), dim3(256), dim3(512), 65536, 0, in, out, 100);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NR, NM>), dim3(256), dim3(512), 65536, 0, in, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double ns = ms * 1e6 / iters;
  printf("%2d ds_read_b128 + %2d mfma per step: %7.1f ns/step = %5.2f ns per mfma per SIMD (2 waves)\n", NR, NM, ns, ns / (2.0 * NM));
}

int main() {
  float *in, *out;
  CHECK(hipMalloc(&in, 65536 * 4)); CHECK(hipMalloc(&out, 256 * 512 * 4));
  static float h[65536];
  for (int i = 0; i < 65536; ++i) h[i] = 0.001f * (i % 977) + 0.5f;
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  run<1, 21>(in, out);
  run<4, 21>(in, out);
  run<7, 21>(in, out);
  run<14, 21>(in, out);
  run<14, 42>(in, out);
  run<8, 24>(in, out);
  run<22, 66>(in, out);
  run<11, 66>(in, out);
  return 0;
}
