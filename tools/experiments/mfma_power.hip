// What does the 1.4 kW package cap buy in fp16 MFMAs?  A bare v_mfma_f32_16x16x32_f16 loop on register operands (no LDS, no memory),
// 2 waves per SIMD on every CU, run for a few seconds while rocm-smi is sampled; random operands (mode 0) and all-zero operands (mode 1).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_power.hip -o tools/bin/mfma_power ; tools/bin/mfma_power [mode] [seconds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(512) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[i][e] = (_Float16)in[(tid * 64 + i * 8 + e) & 65535];
      b[i][e] = (_Float16)in[(tid * 64 + 32 + i * 8 + e) & 65535];
    }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[tid] = s;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  float* h = (float*)malloc(65536 * sizeof(float));
  srand(1);
  for (int i = 0; i < 65536; ++i) h[i] = mode ? 0.0f : (float)(rand() % 2001 - 1000) / 500.0f;
  float *din, *dout;
  CHECK(hipMalloc(&din, 65536 * 4));
  CHECK(hipMalloc(&dout, 256 * 8 * 512 * 4));
  CHECK(hipMemcpy(din, h, 65536 * 4, hipMemcpyHostToDevice));
  const int blocks = 256, iters = 20000;   // one 8-wave workgroup per CU = 2 waves per SIMD; 16 x 20000 MFMAs per wave per launch
  hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, din, dout, 100);
  CHECK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  int sampled = 0;
  while (el < secs) {
    for (int q = 0; q < 8; ++q) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
    launches += 8;
    if (sampled < 4 && el > 0.5 + 0.6 * sampled) {   // rocm-smi while the queue is full
      (void)!system("rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -2 | head -1");
      ++sampled;
    }
    CHECK(hipDeviceSynchronize());
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double flops = (double)launches * blocks * 8 * 16.0 * iters * 16384.0;
  printf("mode %d (%s operands): %.1f TF executed over %.2f s (%.0f %% of 2.5 PF)\n", mode, mode ? "zero" : "random", flops / el / 1e12, el,
         100.0 * flops / el / 2.5e15);
  return 0;
}
