// The right-volume linear term of the Delta head, A2raw[ksl][jb][o] = partial sums over K slice ksl of
// sum_{dj,c} R[15 jb + dj][c] W1[dj][c][o] (generateNet.py:96-100 applied to the query alone), as ONE WAVE's task on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32: an fp32 FMA chain).  64 tasks per right volume: 8 K slices x (2 m-tiles of 16 jb x 4 n-tiles
// of 16 o).  Shared by delta_a2_kernel (delta_head_f16x3.hip: grid (volumes, slices), wave = tile) and by the yaw kernel of small
// sweeps (corr_spectral.hip), whose launch carries the 64 tasks in 22 extra workgroups: the two are independent, and a single-pair
// query has one launch less in its chain.  Rows 15 jb .. 15 jb + 14 of R are contiguous, so the A operand of output row jb is simply
// R[1920 jb + k].  The slices are summed in a fixed order by the prepare / query kernels.
#pragma once
#include "ovn_internal.h"

constexpr int OVN_A2_KSPLIT = 8;                          // K slices per right volume
constexpr int OVN_A2_ELEMS = OVN_G * OVN_C1_OUT;          // floats of A2 per right volume and slice [jb][o]
constexpr int OVN_A2_TASKS = OVN_A2_KSPLIT * 8;           // wave tasks per right volume

#ifdef __HIPCC__
// task = ksl * 8 + tile, tile = 4 mt + nt; `lane` = the calling lane of a whole wave; a2raw_v: this right volume's [8][24][64] block
__device__ __forceinline__ void ovn_delta_a2_task(const float* __restrict__ R, const float* __restrict__ w1raw,
                                                  float* __restrict__ a2raw_v, int task, int lane) {
  constexpr int K1 = OVN_S * OVN_FEAT_C, O1 = OVN_C1_OUT, G = OVN_G;
  constexpr int KS = K1 / 4 / OVN_A2_KSPLIT;   // 60 k-steps of 4 per slice
  static_assert(K1 % (4 * OVN_A2_KSPLIT) == 0 && KS % 4 == 0, "a slice is whole groups of 16 k");
  const int ksl = task >> 3, mt = (task >> 2) & 1, nt = task & 3;
  const int lrow = lane & 15, g = lane >> 4;
  const int jb = 16 * mt + lrow;
  // K order inside a slice: step (j, e) takes k = 16 j + 4 g + e from lane group g -- a lane's four consecutive k are ONE 16-byte load
  // of its row (the four lane groups of a row read 64 contiguous bytes); k = 4 ks + g, one float per lane and step, made every load
  // instruction touch 16 rows x 4 bytes at 7.5 KB strides and the kernel address-bound (14 us in front of every sweep).  Any K order
  // serves as long as A and B agree; chain e sums its 15 steps in order, the chains are combined in one fixed order.
  const float* arow = R + (size_t)(jb < G ? jb : G - 1) * K1 + 4 * KS * ksl + 4 * g;
  const float* bcol = w1raw + (size_t)(4 * KS * ksl + 4 * g) * O1 + 16 * nt + lrow;
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // independent chains
  f32x4 av[KS / 4];
  float bv[KS / 4][4];
#pragma unroll
  for (int j = 0; j < KS / 4; ++j) {
    av[j] = *reinterpret_cast<const f32x4*>(arow + 16 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[j][e] = bcol[(size_t)(16 * j + e) * O1];
  }
#pragma unroll
  for (int j = 0; j < KS / 4; ++j) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], bv[j][e], acc[e], 0, 0, 0);
  }
  const f32x4 s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * mt + 4 * g + r;
    if (row < G) a2raw_v[(size_t)ksl * OVN_A2_ELEMS + row * O1 + 16 * nt + lrow] = s[r];
  }
}
#endif
