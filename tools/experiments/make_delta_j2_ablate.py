#!/usr/bin/env python3
"""Generates tools/experiments/delta_j2_ablate.hip from the product kernel (timing-only ablation variants).
   python tools/experiments/make_delta_j2_ablate.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize \
       tools/experiments/delta_j2_ablate.hip -o tools/bin/delta_j2_ablate
ABL bits: 256 no GEMM2, 512 no o1 split/store, 1 no W1 staging, 2 no slice barriers, 8 no B LDS reads (constant fragments), 16 no split (fake A from L + R),
32 no MFMA, 64 no epilogue/GEMM2, 128 no L reloads.  (Making R or L loop-invariant is NOT a valid ablation: the
compiler hoists the split out of the step loop.)"""
import os
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
src = open(os.path.join(root, "overlapnet_amd/csrc/delta_head_bf16x3.hip")).read()


def rep(s, old, new, count=1):
    assert old in s, old[:60]
    return s.replace(old, new, count)


k0 = src.index("template <int T, int NW, bool DMA>")
k1 = src.index("#undef OVN_LOAD_L")
kern = src[k0:k1]
pre = src[src.index("typedef __bf16 bf16x8"):k0]
kern = rep(kern, "template <int T, int NW, bool DMA>", "template <int T, int NW, bool DMA, int ABL>")
kern = rep(kern, '''      } else {                                                                                                    \\
        _Pragma("unroll") for (int q = 0; q < PFN; ++q)                                                           \\
            pf[q] = *reinterpret_cast<const f32x4*>(src + (q * NT_ + tid) * 16);                                  \\
      }''', '''      } else if (!(ABL & 1)) {                                                                                    \\
        _Pragma("unroll") for (int q = 0; q < PFN; ++q)                                                           \\
            pf[q] = *reinterpret_cast<const f32x4*>(src + (q * NT_ + tid) * 16);                                  \\
      }''')
kern = rep(kern, '''      if (!DMA) {                                                                                                 \\
        unsigned char* dstw''', '''      if (!DMA && !(ABL & 1)) {                                                                                   \\
        unsigned char* dstw''')
kern = rep(kern, '''      __syncthreads();                                                                                            \\
      cur ^= 1;''', '''      if (!(ABL & 2)) __syncthreads();                                                                            \\
      cur ^= 1;''')
kern = rep(kern, '''          bh[nt] = *reinterpret_cast<const bf16x8*>(wbuf + ((nt * 2 + 0) * 64 + lane) * 16);                      \\
          bl[nt] = *reinterpret_cast<const bf16x8*>(wbuf + ((nt * 2 + 1) * 64 + lane) * 16);                      \\''',
           '''          if (ABL & 8) { bh[nt] = fakeb; bl[nt] = fakeb; } else {                                                 \\
          bh[nt] = *reinterpret_cast<const bf16x8*>(wbuf + ((nt * 2 + 0) * 64 + lane) * 16);                      \\
          bl[nt] = *reinterpret_cast<const bf16x8*>(wbuf + ((nt * 2 + 1) * 64 + lane) * 16); }                    \\''')
for grp in ("ra", "rb"):
    kern = rep(kern, "          make_a(LX[t][0], LX[t][1], %s0, %s1, ah, al);" % (grp, grp) + " " * 59 + "\\",
               "          if (ABL & 16) { ah = __builtin_bit_cast(bf16x8, LX[t][0] + %s0); al = __builtin_bit_cast(bf16x8, LX[t][1] + %s1); } else \\\n"
               "          make_a(LX[t][0], LX[t][1], %s0, %s1, ah, al);" % (grp, grp, grp, grp) + " " * 59 + "\\")
kern = rep(kern, "#define OVN_TILE_MFMA(J, T, AH, AL)" + " " * 71 + "\\", '''#define OVN_TILE_MFMA(J, T, AH, AL)                                                                       \\
  if (ABL & 32) {                                                                                          \\
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                     \\
      f32x4 x = __builtin_bit_cast(f32x4, AH), y = __builtin_bit_cast(f32x4, AL), z = __builtin_bit_cast(f32x4, bh[nt]), u = __builtin_bit_cast(f32x4, bl[nt]); \\
      acc[J][T][nt][nt] += x[nt] + y[nt] + z[0] + u[0];                                                    \\
    }                                                                                                      \\
  } else                                                                                                   \\
  OVN_TILE_MFMA_REAL(J, T, AH, AL)
#define OVN_TILE_MFMA_REAL(J, T, AH, AL)                                                                  \\
  {                                                                                                        \\''')
kern = rep(kern, "      acc[J][T][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, bl[nt], acc[J][T][nt], 0, 0, 0);\n",
           "      acc[J][T][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, bl[nt], acc[J][T][nt], 0, 0, 0);        \\\n  }\n")
for sl in ("s1", "s2", "s3"):
    kern = rep(kern, "    OVN_LOAD_L(la, %s)\n" % sl, "    if (!(ABL & 128)) OVN_LOAD_L(la, %s)\n" % sl)
kern = rep(kern, "    OVN_SLICE(la, s3)\n    OVN_LOAD_L(la, s0)\n", "    OVN_SLICE(la, s3)\n    if (!(ABL & 128)) OVN_LOAD_L(la, s0)\n")
kern = rep(kern, '''#pragma unroll
    for (int j = 0; j < 2; ++j) {
    const int jb = 2 * jb2 + j;''', '''    if (ABL & 64) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) sum += acc[j][t][nt][0] + acc[j][t][nt][1] + acc[j][t][nt][2] + acc[j][t][nt][3];
      if (sum == 12345.678f) o2[pair] = sum;
      continue;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
    const int jb = 2 * jb2 + j;''')
kern = rep(kern, "  int cur = 0;\n", "  int cur = 0;\n  bf16x8 fakeb = __builtin_bit_cast(bf16x8, la[1][0]);\n")
kern = rep(kern, "    if (wave < 8) {\n      const int ib0 = lrow;", "    if (wave < 8 && !(ABL & 256)) {\n      const int ib0 = lrow;")
kern = rep(kern, "    {\n      float bv[4];", "    if (!(ABL & 512)) {\n      float bv[4];")
kern = rep(kern, "    DST[u][0] = *reinterpret_cast<const bf16x8*>(wk);                              \\\n    DST[u][1] = *reinterpret_cast<const bf16x8*>(wk + 512);                        \\",
           "    if (ABL & 1024) { DST[u][0] = fakeb; DST[u][1] = fakeb; } else {               \\\n    DST[u][0] = *reinterpret_cast<const bf16x8*>(wk);                              \\\n    DST[u][1] = *reinterpret_cast<const bf16x8*>(wk + 512); }                      \\")
kern = rep(kern, "    af[SLOT][0] = *reinterpret_cast<const bf16x8*>(a0h + 32 * ks_);                \\",
           "    if (ABL & 2048) { af[SLOT][0] = fakeb; af[SLOT][1] = fakeb; af[SLOT][2] = fakeb; af[SLOT][3] = fakeb; } else \\\n    af[SLOT][0] = *reinterpret_cast<const bf16x8*>(a0h + 32 * ks_);                \\")
host = r'''
#undef OVN_LOAD_L
#undef OVN_SLICE
#undef OVN_TILE_MFMA
}  // namespace

#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ABL>
int run(const char* what, int n, const float* fl, const float* fr, const __bf16* w1, const float* b1, const __bf16* w2, const float* b2, float* o2) {
  auto k = delta_c12_bf16x3_j2_kernel<3, 8, false, ABL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(n), dim3(512), LDS_BYTES, 0, fl, nullptr, fr, nullptr, w1, b1, w2, b2, o2, 1, 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(n), dim3(512), LDS_BYTES, 0, fl, nullptr, fr, nullptr, w1, b1, w2, b2, o2, 1, 1);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("ABL=%3d  %-58s %.3f ms\n", ABL, what, ms / 5);
  return 0;
}

int main() {
  const int n = 1024;
  float *fl, *fr, *b1, *b2, *o2; __bf16 *w1, *w2;
  const size_t fe = (size_t)n * 360 * 128;
  CK(hipMalloc(&fl, fe * 4)); CK(hipMalloc(&fr, 360 * 128 * 4)); CK(hipMalloc(&b1, 64 * 4)); CK(hipMalloc(&b2, 128 * 4));
  CK(hipMalloc(&o2, (size_t)n * 24 * 24 * 128 * 4));
  const size_t w1e = (size_t)60 * 4 * 2 * 64 * 8, w2e = (size_t)30 * 8 * 2 * 64 * 8;
  CK(hipMalloc(&w1, w1e * 2)); CK(hipMalloc(&w2, w2e * 2));
  std::vector<float> h(fe);
  unsigned s = 1;
  for (size_t i = 0; i < fe; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) * (1.0f / 16777216.0f); }
  CK(hipMemcpy(fl, h.data(), fe * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(fr, h.data(), 360 * 128 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b1, h.data(), 64 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b2, h.data(), 128 * 4, hipMemcpyHostToDevice));
  std::vector<unsigned short> hw(w1e > w2e ? w1e : w2e);
  for (size_t i = 0; i < hw.size(); ++i) { s = s * 1664525u + 1013904223u; hw[i] = 0x3c00 | ((s >> 12) & 0x1ff) | ((s >> 3) & 0x8000); }
  CK(hipMemcpy(w1, hw.data(), w1e * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w2, hw.data(), w2e * 2, hipMemcpyHostToDevice));
#define RUN(A, W) if (run<A>(W, n, fl, fr, w1, b1, w2, b2, o2)) return 1;
  RUN(0, "baseline")
  RUN(64, "no epilogue/GEMM2")
  RUN(64 + 1, "no epilogue, no W1 staging")
  RUN(64 + 1 + 2, "no epilogue, no staging, no slice barriers")
  RUN(64 + 1 + 2 + 128, "... + no L reloads")
  RUN(64 + 8, "no epilogue, no B LDS reads")
  RUN(64 + 16, "no epilogue, no split")
  RUN(64 + 32, "no epilogue, no MFMA")
  RUN(256, "no GEMM2 (o1 split + stores + barriers kept)")
  RUN(1024, "GEMM2 without W2 loads (constant fragments)")
  RUN(2048, "GEMM2 without o1 LDS reads (first fragment only)")
  RUN(1024 + 2048, "GEMM2 MFMAs only")
  return 0;
}
'''
out = ("// Timing ablations of the shipped two-group Delta kernel, generated from overlapnet_amd/csrc/delta_head_bf16x3.hip by\n"
       "// tools/experiments/make_delta_j2_ablate.py (results are WRONG by construction, only the timings mean anything).\n"
       "#include <hip/hip_runtime.h>\n#include <stdint.h>\n#define OVN_FEAT_W 360\n#define OVN_FEAT_C 128\n#define OVN_S 15\n"
       "#define OVN_G 24\n#define OVN_C1_OUT 64\n#define OVN_C2_OUT 128\n#define OVN_FEAT_ELEMS (360 * 128)\n"
       "typedef float f32x4 __attribute__((ext_vector_type(4)));\n" + pre + kern + host)
open(os.path.join(root, "tools/experiments/delta_j2_ablate.hip"), "w").write(out)
print("wrote tools/experiments/delta_j2_ablate.hip")
