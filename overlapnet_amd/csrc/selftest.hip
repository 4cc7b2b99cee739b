// Self-test of the MFMA fragment layout assumptions of this library (gfx950):
//   v_mfma_f32_16x16x4_f32:  A lane l -> A[row = l&15][k = l>>4],  B lane l -> B[k = l>>4][col = l&15],
//                            C/D lane l, reg r -> C[row = 4*(l>>4) + r][col = l&15].
// One wave multiplies an asymmetric 16x8 by 8x16 integer-valued pair (exact in fp32) and the host checks
// every element, so a transposed or permuted layout cannot pass.
#include "ovn_internal.h"

namespace {

__global__ void mfma_probe_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < 8; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * 8 + k0 + g], B[(k0 + g) * 16 + r], acc, 0, 0, 0);
  for (int j = 0; j < 4; ++j) C[(4 * g + j) * 16 + r] = acc[j];
}

}  // namespace

int ovn_mfma_selftest(hipStream_t stream) {
  float hA[16 * 8], hB[8 * 16], hC[256], ref[256];
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 8; ++k) hA[i * 8 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 8; ++k)
    for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)((k * 5 + j * 2 + k * j) % 13 - 6);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float s = 0.f;
      for (int k = 0; k < 8; ++k) s += hA[i * 8 + k] * hB[k * 16 + j];
      ref[i * 16 + j] = s;
    }
  float *dA = nullptr, *dB = nullptr, *dC = nullptr;
  OVN_HIP_CHECK(hipMalloc((void**)&dA, sizeof(hA)));
  OVN_HIP_CHECK(hipMalloc((void**)&dB, sizeof(hB)));
  OVN_HIP_CHECK(hipMalloc((void**)&dC, sizeof(hC)));
  OVN_HIP_CHECK(hipMemcpyAsync(dA, hA, sizeof(hA), hipMemcpyHostToDevice, stream));
  OVN_HIP_CHECK(hipMemcpyAsync(dB, hB, sizeof(hB), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, stream, dA, dB, dC);
  OVN_HIP_CHECK(hipGetLastError());
  OVN_HIP_CHECK(hipMemcpyAsync(hC, dC, sizeof(hC), hipMemcpyDeviceToHost, stream));
  OVN_HIP_CHECK(hipStreamSynchronize(stream));
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dC);
  for (int e = 0; e < 256; ++e)
    if (hC[e] != ref[e]) {
      ovn_set_error("mfma_f32_16x16x4f32 layout self-test failed at C[%d][%d]: got %g want %g", e / 16, e % 16,
                    (double)hC[e], (double)ref[e]);
      return OVN_ERR_STATE;
    }
  return OVN_OK;
}
