// Delta head (DeltaLayer + c_conv1 + c_conv2 fused), 3-term bf16 split, 12-wave schedule for gfx950.
//
// Same mathematics as delta_head_bf16x3.hip (reference generateNet.py:15-61, :96-106); different mapping onto
// the CU, chosen from the measurements of the 8-wave version (instruction-issue bound at 2 waves per SIMD):
//   * 12 waves per workgroup (3 per SIMD), wave w owns ONE 32-row tile (rows 32w..32w+31 of the 360 x 64 c_conv1
//     output) and both 32-column tiles: v_mfma_f32_32x32x16_bf16, half as many matrix instructions per flop;
//   * K = (channel slice of 16, R row dj) walked slice-major: a lane keeps 8 floats of L per slice (ping-pong
//     register sets, next slice prefetched from L2), so the kernel needs ~130 VGPRs and B / R fragments can be
//     fetched from LDS one MFMA step AHEAD of their use;
//   * W1 (hi, lo; pre-permuted) streams through LDS one whole channel slice (15 steps, 60 KB) at a time, written by
//     LDS-DMA (global_load_lds) into the buffer the previous slice just released: ONE barrier per 15 steps.
//     (Measured on the 3 x 12 KB ring this replaced: a barrier every 3 steps re-aligned the phases of the three
//     waves of each SIMD and cost more than the whole split arithmetic -- tools/ubench2.hip.)  The two 60 KB
//     buffers only fit because the hi/lo image of o1 for GEMM2 aliases the second one: it is written after the
//     last slice has been consumed and read before the next column group refills that buffer.
// GEMM2 (c_conv2, 24 x 960 x 128) and the hi/lo LDS image of o1 are as in the 8-wave kernel (waves 0..7).
#include <stdlib.h>

#include "ovn_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int FW = OVN_FEAT_W;        // 360
constexpr int FC = OVN_FEAT_C;        // 128
constexpr int S = OVN_S;              // 15
constexpr int G = OVN_G;              // 24
constexpr int O1 = OVN_C1_OUT;        // 64
constexpr int O2 = OVN_C2_OUT;        // 128
constexpr int K2 = S * O1;            // 960
constexpr int O1_STRIDE = K2 + 8;     // bf16 per o1 row in LDS (1936 B = 121 slots of 16 B: odd)
constexpr int NWAVES = 12;
constexpr int NTHREADS = 64 * NWAVES; // 768
constexpr int NSLICE = 8;             // channel slices of 16
constexpr int STEP_BYTES = 4096;      // [ct(2)][hi/lo][lane(64)][8 bf16]
constexpr int SLICE_BYTES = S * STEP_BYTES;                 // 61440 = 768 threads x 5 x 16 B
constexpr int DMA_PER_THREAD = SLICE_BYTES / (NTHREADS * 16);
// LDS map: [rs 7680][buffer A 61440][buffer B 61440 + pad 32768]; o1 hi/lo (2 x 46464) aliases buffer B + pad
constexpr int RS_BYTES = S * FC * 4;
constexpr int O1_BYTES = G * O1_STRIDE * 2;                 // 46464 per hi / lo image
constexpr int BUFB_OFF = RS_BYTES + SLICE_BYTES;
constexpr size_t LDS_BYTES = (size_t)BUFB_OFF + 2 * O1_BYTES;
static_assert(2 * O1_BYTES >= SLICE_BYTES, "o1 image must cover buffer B");
static_assert(LDS_BYTES <= 163840, "LDS budget");
static_assert(SLICE_BYTES % (NTHREADS * 16) == 0, "slice must split evenly over the workgroup");

// |d0|, |d1| -> packed bf16 pairs: hi = |d| truncated to bf16 (the AND also strips the sign),
// lo = bf16_rne(|d| - hi) with |d| - hi exact in fp32  =>  hi + lo = |d| to ~2^-17 relative.
__device__ __forceinline__ void split_pair(float d0, float d1, unsigned& hi_pk, unsigned& lo_pk) {
  const unsigned h0 = __float_as_uint(d0) & 0x7fff0000u;
  const unsigned h1 = __float_as_uint(d1) & 0x7fff0000u;
  const float l0 = fabsf(d0) - __uint_as_float(h0);
  const float l1 = fabsf(d1) - __uint_as_float(h1);
  hi_pk = __builtin_amdgcn_perm(h1, h0, 0x07060302u);  // {h1[31:16], h0[31:16]}
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 lp;
  lp[0] = (__bf16)l0;
  lp[1] = (__bf16)l1;
  lo_pk = __builtin_bit_cast(unsigned, lp);
}

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

// W1q[u = s*15 + dj (120)][ct(2)][hl(2)][lane(64)][e(8)] = W1[dj][c = 16 s + 8 (lane>>5) + e][o = 32 ct + (lane&31)]
__global__ void delta_prep_w1_w12_kernel(const float* __restrict__ w1, __bf16* __restrict__ w1q) {
  const int total = NSLICE * S * 2 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 7;
    const int lane = (idx >> 3) & 63;
    const int ct = (idx >> 9) & 1;
    const int u = idx >> 10;
    const int s = u / S;
    const int dj = u - s * S;
    const int c = 16 * s + 8 * (lane >> 5) + e;
    const int o = 32 * ct + (lane & 31);
    __bf16 hi, lo;
    split_bf16(w1[(dj * FC + c) * O1 + o], hi, lo);
    const size_t base = (((size_t)u * 2 + ct) * 2) * 512 + lane * 8 + e;
    w1q[base] = hi;
    w1q[base + 512] = lo;
  }
}

__global__ __launch_bounds__(NTHREADS) void delta_c12_bf16x3_w12_kernel(
    const float* __restrict__ feats_l, const int32_t* __restrict__ lidx, const float* __restrict__ feats_r,
    const int32_t* __restrict__ ridx, const __bf16* __restrict__ w1q, const float* __restrict__ b1,
    const __bf16* __restrict__ w2p, const float* __restrict__ b2, float* __restrict__ o2, int abl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* rs = reinterpret_cast<float*>(smem_raw);
  unsigned char* bufa = smem_raw + RS_BYTES;
  unsigned char* bufb = smem_raw + BUFB_OFF;
  __bf16* o1h = reinterpret_cast<__bf16*>(bufb);            // aliases buffer B (+ pad)
  __bf16* o1l = reinterpret_cast<__bf16*>(bufb + O1_BYTES);

  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lr = lane & 31;   // row of the 32x32 tile this lane feeds / column it receives
  const int kh = lane >> 5;   // k-group of the 32x32x16 MFMA

  const float* L = feats_l + (long long)(lidx ? lidx[pair] : pair) * OVN_FEAT_ELEMS;
  const float* R = feats_r + (long long)(ridx ? ridx[pair] : 0) * OVN_FEAT_ELEMS;

  const int irow = 32 * wave + lr;
  const int loff = (irow < FW) ? irow * FC + 8 * kh : -1;
  f32x4 la[2], lb[2], lt[2];  // even / odd channel slices, and the slice in flight from L2
#define OVN_LOAD_L(DST, SL)                                                        \
  if (loff >= 0) {                                                                 \
    DST[0] = *reinterpret_cast<const f32x4*>(L + loff + 16 * (SL));                \
    DST[1] = *reinterpret_cast<const f32x4*>(L + loff + 16 * (SL) + 4);            \
  } else {                                                                         \
    DST[0] = (f32x4){0.f, 0.f, 0.f, 0.f};                                          \
    DST[1] = (f32x4){0.f, 0.f, 0.f, 0.f};                                          \
  }
  OVN_LOAD_L(la, 0)
  OVN_LOAD_L(lb, 1)

  const unsigned char* w1bytes = reinterpret_cast<const unsigned char*>(w1q);
  // LDS-DMA of one W1 slice: 5 x 16 B per thread, destination = wave-uniform base + lane * 16
#define OVN_DMA_SLICE(DSTBUF, SL)                                                                                  \
  if (!(abl & 1)) _Pragma("unroll") for (int q = 0; q < DMA_PER_THREAD; ++q) {                                                     \
    __builtin_amdgcn_global_load_lds(                                                                              \
        (const __attribute__((address_space(1))) void*)(w1bytes + (size_t)(SL)*SLICE_BYTES + (q * NTHREADS + tid) * 16), \
        (__attribute__((address_space(3))) void*)((DSTBUF) + (q * NTHREADS + (tid & ~63)) * 16), 16, 0, 0);        \
  }
  OVN_DMA_SLICE(bufa, 0)

  // B fragments of one step: [ct][hi/lo] -> 4 x ds_read_b128
#define OVN_LOAD_B(DST, BUF, STEP)                                                                               \
  {                                                                                                              \
    const unsigned char* bp = (BUF) + (STEP)*STEP_BYTES + lane * 16;                                             \
    DST[0] = *reinterpret_cast<const bf16x8*>(bp);        /* ct 0 hi */                                           \
    DST[1] = *reinterpret_cast<const bf16x8*>(bp + 1024); /* ct 0 lo */                                           \
    DST[2] = *reinterpret_cast<const bf16x8*>(bp + 2048); /* ct 1 hi */                                           \
    DST[3] = *reinterpret_cast<const bf16x8*>(bp + 3072); /* ct 1 lo */                                           \
  }

  // One channel slice SL (15 MFMA steps) with the L slice in LX and its W1 slice in BUF (published).
  // LASTSL: the step after this slice belongs to the next column group, whose R rows are not in LDS yet.
#define OVN_SLICE(LX, SL, BUF, LASTSL)                                                                             \
  {                                                                                                                \
    bf16x8 bq[4];                                                                                                  \
    OVN_LOAD_B(bq, BUF, 0)                                                                                         \
    _Pragma("unroll 1") for (int c5 = 0; c5 < S / 3; ++c5)                                                         \
    _Pragma("unroll") for (int hh = 0; hh < 3; ++hh) {                                                             \
      const int dj = 3 * c5 + hh;                                                                                  \
      bf16x8 bn[4] = {bq[0], bq[1], bq[2], bq[3]};                                                                 \
      if (dj + 1 < S && !(abl & 8)) { OVN_LOAD_B(bn, BUF, dj + 1) }                                                \
      f32x4 rn0 = rq0, rn1 = rq1;                                                                                  \
      if (dj + 1 < S) {                                                                                            \
        const float* rr = rs + (dj + 1) * FC + 16 * (SL) + 8 * kh;                                                 \
        rn0 = *reinterpret_cast<const f32x4*>(rr);                                                                 \
        rn1 = *reinterpret_cast<const f32x4*>(rr + 4);                                                             \
      } else if (!(LASTSL)) {                                                                                      \
        const float* rr = rs + 16 * ((SL) + 1) + 8 * kh;                                                           \
        rn0 = *reinterpret_cast<const f32x4*>(rr);                                                                 \
        rn1 = *reinterpret_cast<const f32x4*>(rr + 4);                                                             \
      }                                                                                                            \
      unsigned h0, h1, h2, h3, q0, q1, q2, q3;                                                                     \
      split_pair(LX[0][0] - rq0[0], LX[0][1] - rq0[1], h0, q0);                                                    \
      split_pair(LX[0][2] - rq0[2], LX[0][3] - rq0[3], h1, q1);                                                    \
      split_pair(LX[1][0] - rq1[0], LX[1][1] - rq1[1], h2, q2);                                                    \
      split_pair(LX[1][2] - rq1[2], LX[1][3] - rq1[3], h3, q3);                                                    \
      const bf16x8 ah = __builtin_bit_cast(bf16x8, (u32x4){h0, h1, h2, h3});                                       \
      const bf16x8 al = __builtin_bit_cast(bf16x8, (u32x4){q0, q1, q2, q3});                                       \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[0], acc[0], 0, 0, 0);                                \
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[2], acc[1], 0, 0, 0);                                \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[0], acc[0], 0, 0, 0);                                \
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[2], acc[1], 0, 0, 0);                                \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[1], acc[0], 0, 0, 0);                                \
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[3], acc[1], 0, 0, 0);                                \
      bq[0] = bn[0];                                                                                               \
      bq[1] = bn[1];                                                                                               \
      bq[2] = bn[2];                                                                                               \
      bq[3] = bn[3];                                                                                               \
      rq0 = rn0;                                                                                                   \
      rq1 = rn1;                                                                                                   \
    }                                                                                                              \
  }

  f32x4 rq0, rq1;    // R fragment of the step about to run (fetched one step ahead)

  for (int jb = 0; jb < G; ++jb) {
    // previous group's GEMM2 is done with the o1 image (buffer B) and rs; slice 0 (DMA above / during the
    // previous group's last slice) has landed in buffer A
    __syncthreads();
    if (tid < S * FC / 4)
      *reinterpret_cast<f32x4*>(rs + 4 * tid) = *reinterpret_cast<const f32x4*>(R + jb * S * FC + 4 * tid);
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[0][r] = 0.f;
      acc[1][r] = 0.f;
    }
    rq0 = *reinterpret_cast<const f32x4*>(rs + 8 * kh);
    rq1 = *reinterpret_cast<const f32x4*>(rs + 8 * kh + 4);

#pragma unroll 1
    for (int sp = 0; sp < NSLICE / 2; ++sp) {
      const int s0 = 2 * sp, s1 = 2 * sp + 1;
      // even slice from buffer A while the odd slice streams into buffer B.  The L slice needed two slices
      // from now is fetched into `lt` at the START of the slice (a load issued right before the barrier would be
      // drained by the barrier's vmcnt(0) and expose a full L2 round trip per slice).
      OVN_DMA_SLICE(bufb, s1)
      {
        const int sn = (s0 + 2) & (NSLICE - 1);
        OVN_LOAD_L(lt, sn)
      }
      OVN_SLICE(la, s0, bufa, false)
      la[0] = lt[0];
      la[1] = lt[1];
      if (!(abl & 2)) __syncthreads();  // slice s1 landed and is visible; everyone is done with buffer A
      // odd slice from buffer B while the next even slice (of this or the next column group) streams into A
      {
        const int sn = (s1 + 1) & (NSLICE - 1);
        if (s1 != NSLICE - 1 || jb + 1 < G) { OVN_DMA_SLICE(bufa, sn) }
      }
      {
        const int sn = (s1 + 2) & (NSLICE - 1);
        OVN_LOAD_L(lt, sn)
      }
      OVN_SLICE(lb, s1, bufb, (s1 == NSLICE - 1))
      lb[0] = lt[0];
      lb[1] = lt[1];
      if (!(abl & 2)) __syncthreads();  // slice landed in A; everyone is done with buffer B (the o1 image may overwrite it)
    }

    if (abl & 4) continue;  // timing ablation: no epilogue / GEMM2
    // o1 (+ bias) -> LDS as hi/lo bf16 in GEMM2's A layout.
    // 32x32 C/D: lane holds column lr, rows (r&3) + 8*(r>>2) + 4*kh of the tile, r = 0..15.
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int o = 32 * ct + lr;
      const float bv = b1[o];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (i < FW) {
          const int ib = i / S;
          const int di = i - ib * S;
          __bf16 h, l;
          split_bf16(acc[ct][r] + bv, h, l);
          o1h[ib * O1_STRIDE + di * O1 + o] = h;
          o1l[ib * O1_STRIDE + di * O1 + o] = l;
        }
      }
    }
    __syncthreads();

    // GEMM2 (24 x 960) x (960 x 128) on waves 0..7 with 16x16x32 tiles: wave -> m-tile (wave&1), n-tiles 2*(wave>>1), +1
    if (wave < 8) {
      const int lrow = lane & 15;
      const int g = lane >> 4;
      const int mt = wave & 1;
      const int ntp = wave >> 1;
      int ib = 16 * mt + lrow;
      if (ib > G - 1) ib = G - 1;
      const __bf16* ahp = o1h + ib * O1_STRIDE + 8 * g;
      const __bf16* alp = o1l + ib * O1_STRIDE + 8 * g;
      const __bf16* wcol = w2p + ((size_t)(2 * ntp) * 2) * 512 + lane * 8;
      f32x4 acc2[2];
      acc2[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int ks = 0; ks < K2 / 32; ++ks) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ahp + 32 * ks);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(alp + 32 * ks);
        const __bf16* wk = wcol + (size_t)ks * (8 * 2 * 512);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wk + (q * 2 + 0) * 512);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(wk + (q * 2 + 1) * 512);
          acc2[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc2[q], 0, 0, 0);
          acc2[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc2[q], 0, 0, 0);
          acc2[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc2[q], 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int p = 16 * (2 * ntp + q) + lrow;
        const float bv = b2[p];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ib2 = 16 * mt + 4 * g + r;
          if (ib2 < G) o2[(((long long)pair * G + ib2) * G + jb) * O2 + p] = fmaxf(acc2[q][r] + bv, 0.0f);
        }
      }
    }
  }
#undef OVN_SLICE
#undef OVN_LOAD_B
#undef OVN_LOAD_L
#undef OVN_DMA_SLICE
}

}  // namespace

int ovn_delta_prepare_w12(const float* c1_kernel_dev, void** w1q_out, hipStream_t stream) {
  const size_t elems = (size_t)S * FC * O1 * 2;  // hi + lo
  OVN_HIP_CHECK(hipMalloc(w1q_out, elems * sizeof(__bf16)));
  hipLaunchKernelGGL(delta_prep_w1_w12_kernel, dim3(240), dim3(256), 0, stream, c1_kernel_dev,
                     reinterpret_cast<__bf16*>(*w1q_out));
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

int ovn_delta_c12_w12_forward(const ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r,
                              const int32_t* ridx, int n, float* o2, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    OVN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(delta_c12_bf16x3_w12_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set = true;
  }
  static int abl = -1;  // timing ablations for kernel analysis only (results are wrong when non-zero)
  if (abl < 0) {
    const char* e = getenv("OVN_W12_ABLATE");
    abl = e ? atoi(e) : 0;
  }
  hipLaunchKernelGGL(delta_c12_bf16x3_w12_kernel, dim3(n), dim3(NTHREADS), LDS_BYTES, stream, feats_l, lidx, feats_r, ridx,
                     reinterpret_cast<const __bf16*>(ctx->w1q_bf), ctx->b1, reinterpret_cast<const __bf16*>(ctx->w2p_bf),
                     ctx->c2.bias, o2, abl);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
