// Delta head: DeltaLayer + c_conv1 + c_conv2 fused, on the bf16 matrix cores with a 3-term split
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate) for gfx950.  Reference: generateNet.py:15-61 (DeltaLayer), :96-106.
//
// Arithmetic: every fp32 operand x is written as hi + lo (bf16 each) and a*w is evaluated as
// a_hi*w_hi + a_lo*w_hi + a_hi*w_lo: three MFMAs at the bf16 rate (16x the fp32 matrix rate) instead of one fp32 MFMA.
// The dropped a_lo*w_lo term and the rounding of lo are ~2^-17 relative per product; sums are fp32.  The overlap
// tolerance of the north star (1e-4 after the sigmoid) is checked against the fp64 oracle in tests/test_gpu_parity.py.
//
// Work decomposition: one workgroup (8 waves) = one pair; wave w owns rows 48w..48w+47 (3 MFMA row tiles) of the
// 360 x 64 c_conv1 output of TWO column groups jb, jb+1 at a time: the K walk of c_conv1 is shared by the two groups, so
// every W1 fragment read from LDS, every staged W1 chunk, every barrier and every L slice load serves 24 MFMAs per row
// tile.  K = (c, dj) is walked channel-slice-major: an MFMA step covers 32 channels (lane group g = lane>>4 takes
// channels 32g + 8s .. 32g + 8s + 7 for slice s = 0..3) of one R row dj, and the 15 rows dj of a slice are consecutive
// steps -- a lane needs only 8 floats of L per row tile at a time.  |L-R| is formed and split on the VALU (32
// instructions per fragment pair; the matrix pipe hides only ~40 % of them, tools/experiments/ubench3.hip, which is
// what bounds this kernel).  W1 (hi and lo, pre-permuted to this order) streams through a double-buffered 2 x 24 KB LDS
// window shared by the 8 waves; o1 goes to LDS as hi/lo bf16 in GEMM2's [24][960] A layout; GEMM2 (c_conv2) reads its
// weights straight from L2, software-pipelined.  Earlier schedules and their measurements: tools/experiments/.
#include <stdlib.h>

#include "ovn_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int FW = OVN_FEAT_W;        // 360
constexpr int FC = OVN_FEAT_C;        // 128
constexpr int S = OVN_S;              // 15
constexpr int G = OVN_G;              // 24
constexpr int O1 = OVN_C1_OUT;        // 64
constexpr int O2 = OVN_C2_OUT;        // 128
constexpr int K2 = S * O1;            // 960
constexpr int O1_STRIDE = K2 + 8;     // bf16 elements per o1 row in LDS: 1936 B = 121 16-B slots (odd)
constexpr int STEPS_PER_CHUNK = 3;    // MFMA steps per W1 window chunk; 5 chunks = one 15-step channel slice
constexpr int NCHUNK = 4 * S / STEPS_PER_CHUNK;   // 20 chunks per column group
constexpr int STEP_BYTES = 8192;      // [nt(4)][hi/lo][lane(64)][8 bf16]
constexpr int CHUNK_BYTES = STEPS_PER_CHUNK * STEP_BYTES;
constexpr size_t LDS_BYTES = 2 * (size_t)G * O1_STRIDE * 2 + 2 * (size_t)S * FC * 4 + 2 * CHUNK_BYTES;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));

// (l0 - r0, l1 - r1) -> |.| as packed bf16 pairs (element 0 in the low half).  hi = |d| truncated to bf16 (one AND
// also strips the sign), lo = bf16_rne(|d| - hi): |d| - hi is exact in fp32, so hi + lo carries |d| to ~2^-17 relative.
// 7 VALU instructions per pair: the split competes with the MFMAs for issue slots (tools/experiments/ubench3.hip:
// an MFMA hides only ~40 % of the VALU time next to it), so every instruction counts: the two subtractions go through
// one packed-fp32 add (L and R pairs sit in aligned register pairs) and the hi halves are packed by one v_perm_b32.
__device__ __forceinline__ void split_pair(f32x2 l, f32x2 r, unsigned& hi_pk, unsigned& lo_pk) {
  const f32x2 d = l - r;
  const unsigned h0 = __float_as_uint(d[0]) & 0x7fff0000u;
  const unsigned h1 = __float_as_uint(d[1]) & 0x7fff0000u;
  const float l0 = fabsf(d[0]) - __uint_as_float(h0);
  const float l1 = fabsf(d[1]) - __uint_as_float(h1);
  hi_pk = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 lp;
  lp[0] = (__bf16)l0;
  lp[1] = (__bf16)l1;
  lo_pk = __builtin_bit_cast(unsigned, lp);
}

// A fragments (hi, lo) of one 16-row tile for one MFMA step: 8 values |L - R| per lane.
__device__ __forceinline__ void make_a(const f32x4& l0, const f32x4& l1, const f32x4& r0, const f32x4& r1, bf16x8& ah,
                                       bf16x8& al) {
  unsigned h0, h1, h2, h3, q0, q1, q2, q3;
  split_pair((f32x2){l0[0], l0[1]}, (f32x2){r0[0], r0[1]}, h0, q0);
  split_pair((f32x2){l0[2], l0[3]}, (f32x2){r0[2], r0[3]}, h1, q1);
  split_pair((f32x2){l1[0], l1[1]}, (f32x2){r1[0], r1[1]}, h2, q2);
  split_pair((f32x2){l1[2], l1[3]}, (f32x2){r1[2], r1[3]}, h3, q3);
  ah = __builtin_bit_cast(bf16x8, (u32x4){h0, h1, h2, h3});
  al = __builtin_bit_cast(bf16x8, (u32x4){q0, q1, q2, q3});
}

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

// W1p[u = s*15 + dj][nt(4)][hl(2)][lane(64)][e(8)]: W1[dj][c = 32*(lane>>4) + 8*s + e][o = 16*nt + (lane&15)]
__global__ void delta_prep_w1_bf16_kernel(const float* __restrict__ w1, __bf16* __restrict__ w1p) {
  const int total = S * 4 * 4 * 64 * 8;  // (hi, lo) pairs
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 7;
    const int lane = (idx >> 3) & 63;
    const int nt = (idx >> 9) & 3;
    const int u = idx >> 11;  // 0..59
    const int s = u / S;
    const int dj = u - s * S;
    const int c = 32 * (lane >> 4) + 8 * s + e;
    const int o = 16 * nt + (lane & 15);
    __bf16 hi, lo;
    split_bf16(w1[(dj * FC + c) * O1 + o], hi, lo);
    const size_t base = (((size_t)u * 4 + nt) * 2) * 512 + lane * 8 + e;
    w1p[base] = hi;
    w1p[base + 512] = lo;
  }
}

// W2p[ks(30)][nt(8)][hl(2)][lane(64)][e(8)]: W2[k(k')][p = 16*nt + (lane&15)], k' = 32*ks + 8*(lane>>4) + e.
// GEMM2 walks its K axis in the order k' = di*64 + 4*(o & 15) + (o >> 4) instead of k = di*64 + o: the four c_conv1
// n-tiles a lane holds after GEMM1 (o = lrow, 16+lrow, 32+lrow, 48+lrow) are then adjacent in the o1 image, so the
// epilogue stores 8 bytes per (row, hi/lo) instead of four 2-byte pieces.  Any K order works as long as A and B agree.
__global__ void delta_prep_w2_bf16_kernel(const float* __restrict__ w2, __bf16* __restrict__ w2p) {
  const int total = (K2 / 32) * 8 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 7;
    const int lane = (idx >> 3) & 63;
    const int nt = (idx >> 9) & 7;
    const int ks = idx >> 12;
    const int kp = 32 * ks + 8 * (lane >> 4) + e;
    const int m = kp & 63;
    const int k = (kp & ~63) + 16 * (m & 3) + (m >> 2);
    const int p = 16 * nt + (lane & 15);
    __bf16 hi, lo;
    split_bf16(w2[k * O2 + p], hi, lo);
    const size_t base = (((size_t)ks * 8 + nt) * 2) * 512 + lane * 8 + e;
    w2p[base] = hi;
    w2p[base + 512] = lo;
  }
}

template <int T, int NW, bool DMA>
__global__ __launch_bounds__(64 * NW) void delta_c12_bf16x3_j2_kernel(const float* __restrict__ feats_l,
                                                               const int32_t* __restrict__ lidx,
                                                               const float* __restrict__ feats_r,
                                                               const int32_t* __restrict__ ridx,
                                                               const __bf16* __restrict__ w1p,
                                                               const float* __restrict__ b1,
                                                               const __bf16* __restrict__ w2p,
                                                               const float* __restrict__ b2, float* __restrict__ o2, int rot, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* o1h = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* o1l = o1h + G * O1_STRIDE;
  float* rs = reinterpret_cast<float*>(o1l + G * O1_STRIDE);
  unsigned char* wst = reinterpret_cast<unsigned char*>(rs + 2 * S * FC);  // 2 x 24 KB window

  // nsplit > 1 (small sweeps): the 12 column-group passes of a pair are spread over nsplit workgroups, so that a handful of
  // pairs still fills the chip (a pair's latency drops from 1.4 ms to 1.4 / nsplit ms; no work is duplicated)
  const int pair = blockIdx.x / nsplit;
  const int part = blockIdx.x - pair * nsplit;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lrow = lane & 15;
  const int g = lane >> 4;

  const float* L = feats_l + (long long)(lidx ? lidx[pair] : pair) * OVN_FEAT_ELEMS;
  const float* R = feats_r + (long long)(ridx ? ridx[pair] : 0) * OVN_FEAT_ELEMS;

  // this lane's slice of L for channel slice s: rows 48*wave + 16*t + lrow, channels 32g + 8s .. +7
  constexpr int NT_ = 64 * NW;                       // threads
  constexpr int PFN = CHUNK_BYTES / (NT_ * 16);      // 16-byte window pieces per thread per chunk
  static_assert(CHUNK_BYTES % (NT_ * 16) == 0 && T * NW * 16 >= FW, "bad tiling");
  int lrow_off[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int i = 16 * T * wave + 16 * t + lrow;
    lrow_off[t] = (i < FW) ? i * FC + 32 * g : -1;
  }
  f32x4 la[T][2];  // even / odd channel slices ping-pong (no register rotation)
#define OVN_LOAD_L(DST, SL)                                                                              \
  _Pragma("unroll") for (int t = 0; t < T; ++t) {                                                        \
    if (lrow_off[t] >= 0) {                                                                              \
      DST[t][0] = *reinterpret_cast<const f32x4*>(L + lrow_off[t] + 8 * (SL));                           \
      DST[t][1] = *reinterpret_cast<const f32x4*>(L + lrow_off[t] + 8 * (SL) + 4);                       \
    } else {                                                                                             \
      DST[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                           \
      DST[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                           \
    }                                                                                                    \
  }
  // Workgroups walk the channel slices (and with them the W1 stream) in rotated order: the 32 CUs of an XCD then
  // touch every W1 line several times per column-group period instead of in one burst, which keeps the 1 MB of
  // weights resident in the 4 MB L2 under the private L / o2 streams (LRU thrash otherwise: 18.7 GB/launch of misses).
  const int s0 = rot ? ((pair >> 3) & 3) : 0;
  const int s1 = (s0 + 1) & 3, s2 = (s0 + 2) & 3, s3 = (s0 + 3) & 3;
  OVN_LOAD_L(la, s0)

  // W1 chunk 0 -> LDS buffer 0 (every column group walks the same 20 chunks, so the window just wraps)
  const unsigned char* w1bytes = reinterpret_cast<const unsigned char*>(w1p);
  f32x4 pf[PFN];
#pragma unroll
  for (int q = 0; q < PFN; ++q) {
    pf[q] = *reinterpret_cast<const f32x4*>(w1bytes + (size_t)(5 * s0) * CHUNK_BYTES + (q * NT_ + tid) * 16);
    *reinterpret_cast<f32x4*>(wst + (q * NT_ + tid) * 16) = pf[q];
  }
  int cur = 0;
  int chunk = 5 * s0;  // running chunk index 0..19 (cyclic), 5 chunks per slice

  // 12 MFMAs of one row tile; term-major so consecutive MFMAs never chain on one accumulator
#define OVN_TILE_MFMA(J, T, AH, AL)                                                                       \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                         \
      acc[J][T][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, bh[nt], acc[J][T][nt], 0, 0, 0);        \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                         \
      acc[J][T][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AL, bh[nt], acc[J][T][nt], 0, 0, 0);        \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                         \
      acc[J][T][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, bl[nt], acc[J][T][nt], 0, 0, 0);
  // One channel slice SL (15 MFMA steps = 5 window chunks) with the L slice held in LX.
#define OVN_SLICE(LX, SL)                                                                                         \
  {                                                                                                               \
    for (int c5 = 0; c5 < S / STEPS_PER_CHUNK; ++c5) {                                                            \
      const int nxt = (chunk + 1 == NCHUNK) ? 0 : chunk + 1;                                                      \
      const unsigned char* src = w1bytes + (size_t)nxt * CHUNK_BYTES;                                             \
      if (DMA) {                                                                                                  \
        _Pragma("unroll") for (int q = 0; q < PFN; ++q) __builtin_amdgcn_global_load_lds(                         \
            (const __attribute__((address_space(1))) void*)(src + (q * NT_ + tid) * 16),                         \
            (__attribute__((address_space(3))) void*)(wst + (cur ^ 1) * CHUNK_BYTES + (q * NT_ + (tid & ~63)) * 16), 16, 0, 0); \
      } else {                                                                                                    \
        _Pragma("unroll") for (int q = 0; q < PFN; ++q)                                                           \
            pf[q] = *reinterpret_cast<const f32x4*>(src + (q * NT_ + tid) * 16);                                  \
      }                                                                                                           \
      _Pragma("unroll") for (int h = 0; h < STEPS_PER_CHUNK; ++h) {                                               \
        const int dj = c5 * STEPS_PER_CHUNK + h;                                                                  \
        const unsigned char* wbuf = wst + cur * CHUNK_BYTES + h * STEP_BYTES;                                     \
        const float* rrow = rs + dj * FC + 32 * g + 8 * (SL);                                                     \
        const f32x4 ra0 = *reinterpret_cast<const f32x4*>(rrow);                                                  \
        const f32x4 ra1 = *reinterpret_cast<const f32x4*>(rrow + 4);                                              \
        const f32x4 rb0 = *reinterpret_cast<const f32x4*>(rrow + S * FC);                                         \
        const f32x4 rb1 = *reinterpret_cast<const f32x4*>(rrow + S * FC + 4);                                     \
        bf16x8 bh[4], bl[4];                                                                                      \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                        \
          bh[nt] = *reinterpret_cast<const bf16x8*>(wbuf + ((nt * 2 + 0) * 64 + lane) * 16);                      \
          bl[nt] = *reinterpret_cast<const bf16x8*>(wbuf + ((nt * 2 + 1) * 64 + lane) * 16);                      \
        }                                                                                                         \
        _Pragma("unroll") for (int t = 0; t < T; ++t) {                                                           \
          bf16x8 ah, al;                                                                                          \
          make_a(LX[t][0], LX[t][1], ra0, ra1, ah, al);                                                           \
          OVN_TILE_MFMA(0, t, ah, al)                                                                             \
          make_a(LX[t][0], LX[t][1], rb0, rb1, ah, al);                                                           \
          OVN_TILE_MFMA(1, t, ah, al)                                                                             \
        }                                                                                                         \
      }                                                                                                           \
      if (!DMA) {                                                                                                 \
        unsigned char* dstw = wst + (cur ^ 1) * CHUNK_BYTES;                                                      \
        _Pragma("unroll") for (int q = 0; q < PFN; ++q)                                                           \
            *reinterpret_cast<f32x4*>(dstw + (q * NT_ + tid) * 16) = pf[q];                                       \
      }                                                                                                           \
      __syncthreads();                                                                                            \
      cur ^= 1;                                                                                                   \
      chunk = nxt;                                                                                                \
    }                                                                                                             \
  }

  for (int jb2 = part * (G / 2) / nsplit; jb2 < (part + 1) * (G / 2) / nsplit; ++jb2) {
    __syncthreads();  // previous pass's GEMM2 is done with o1h/o1l and rs; W window write above is visible
    for (int i4 = tid; i4 < 2 * S * FC / 4; i4 += NT_)
      *reinterpret_cast<f32x4*>(rs + 4 * i4) = *reinterpret_cast<const f32x4*>(R + jb2 * 2 * S * FC + 4 * i4);
    __syncthreads();

    f32x4 acc[2][T][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // single L register set (the second accumulator set took the ping-pong's registers): each slice load is exposed
    OVN_SLICE(la, s0)
    OVN_LOAD_L(la, s1)
    OVN_SLICE(la, s1)
    OVN_LOAD_L(la, s2)
    OVN_SLICE(la, s2)
    OVN_LOAD_L(la, s3)
    OVN_SLICE(la, s3)
    OVN_LOAD_L(la, s0)

#pragma unroll
    for (int j = 0; j < 2; ++j) {
    const int jb = 2 * jb2 + j;
    if (j == 1) __syncthreads();  // GEMM2 of the first group is done with the o1 image
    // o1 (+ bias) -> LDS as hi/lo bf16 in GEMM2's A layout (K order k' = di*64 + 4*lrow + nt, see the W2 prep kernel).
    // C/D: lane holds column lrow of every n-tile, rows 4g..4g+3: one 8-byte store for the 4 hi parts, one for the lo parts.
    {
      float bv[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bv[nt] = b1[16 * nt + lrow];
#pragma unroll
      for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * T * wave + 16 * t + 4 * g + r;
          if (i < FW) {
            const int ib = i / S;
            const int di = i - ib * S;
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 h4, l4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              __bf16 h, l;
              split_bf16(acc[j][t][nt][r] + bv[nt], h, l);
              h4[nt] = h;
              l4[nt] = l;
            }
            *reinterpret_cast<bf16x4*>(o1h + ib * O1_STRIDE + di * O1 + 4 * lrow) = h4;
            *reinterpret_cast<bf16x4*>(o1l + ib * O1_STRIDE + di * O1 + 4 * lrow) = l4;
          }
        }
      }
    }
    __syncthreads();

    // GEMM2 (24 x 960) x (960 x 128): wave w owns output columns 16w..16w+15 for BOTH 16-row m-tiles, so every
    // W2 fragment is fetched from L2 by exactly one wave of the workgroup (491 KB per column group, not 2x that).
    if (wave < 8) {
      const int ib0 = lrow;                                   // m-tile 0: rows 0..15
      const int ib1 = (16 + lrow > G - 1) ? G - 1 : 16 + lrow;  // m-tile 1: rows 16..23 (+ 8 padding rows)
      const __bf16* a0h = o1h + ib0 * O1_STRIDE + 8 * g;
      const __bf16* a0l = o1l + ib0 * O1_STRIDE + 8 * g;
      const __bf16* a1h = o1h + ib1 * O1_STRIDE + 8 * g;
      const __bf16* a1l = o1l + ib1 * O1_STRIDE + 8 * g;
      const __bf16* wcol = w2p + ((size_t)wave * 2) * 512 + lane * 8;
      // one accumulator per (m-tile, split term): six independent MFMA chains per k-step instead of two -- with only two
      // accumulators every MFMA waited on the one issued two before it (this wave's whole GEMM2 is 2 x 1 tiles)
      f32x4 acc2t[2][3];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc2t[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int ks0 = rot ? 6 * ((pair >> 3) % 5) : 0;  // rotated start of the W2 walk, same reason as s0
      // W2 fragments come straight from L2 (491 KB per column group, no LDS left to stage them): the K walk is
      // software-pipelined in batches of GB k-steps, batch b+1 in flight while batch b feeds the matrix pipe
      constexpr int GB = 3, NB = K2 / 32 / GB;
      static_assert(NB % 2 == 0, "the batch loop is unrolled by two");
      bf16x8 wq0[GB][2], wq1[GB][2];
      auto ksof = [&](int kk) { const int ks = kk + ks0; return ks >= K2 / 32 ? ks - K2 / 32 : ks; };
#define OVN_W2_LOAD(DST, B)                                                        \
  _Pragma("unroll") for (int u = 0; u < GB; ++u) {                                 \
    const __bf16* wk = wcol + (size_t)ksof((B) * GB + u) * (8 * 2 * 512);          \
    DST[u][0] = *reinterpret_cast<const bf16x8*>(wk);                              \
    DST[u][1] = *reinterpret_cast<const bf16x8*>(wk + 512);                        \
  }
// o1 fragments are read from LDS one k-step ahead of the MFMAs that consume them
#define OVN_W2_READ_A(SLOT, B, U)                                                  \
  {                                                                                \
    const int ks_ = ksof((B) * GB + (U));                                          \
    af[SLOT][0] = *reinterpret_cast<const bf16x8*>(a0h + 32 * ks_);                \
    af[SLOT][1] = *reinterpret_cast<const bf16x8*>(a0l + 32 * ks_);                \
    af[SLOT][2] = *reinterpret_cast<const bf16x8*>(a1h + 32 * ks_);                \
    af[SLOT][3] = *reinterpret_cast<const bf16x8*>(a1l + 32 * ks_);                \
  }
#define OVN_W2_COMPUTE(SRC, B)                                                     \
  {                                                                                \
    bf16x8 af[2][4];                                                               \
    OVN_W2_READ_A(0, B, 0)                                                         \
    _Pragma("unroll") for (int u = 0; u < GB; ++u) {                               \
      if (u + 1 < GB) OVN_W2_READ_A((u + 1) & 1, B, u + 1)                         \
      __builtin_amdgcn_sched_barrier(0);                                           \
      acc2t[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][0], SRC[u][0], acc2t[0][0], 0, 0, 0); \
      acc2t[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][2], SRC[u][0], acc2t[1][0], 0, 0, 0); \
      acc2t[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][1], SRC[u][0], acc2t[0][1], 0, 0, 0); \
      acc2t[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][3], SRC[u][0], acc2t[1][1], 0, 0, 0); \
      acc2t[0][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][0], SRC[u][1], acc2t[0][2], 0, 0, 0); \
      acc2t[1][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][2], SRC[u][1], acc2t[1][2], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                           \
    }                                                                              \
  }
      OVN_W2_LOAD(wq0, 0)
#pragma unroll 1
      for (int b = 0; b < NB; b += 2) {
        OVN_W2_LOAD(wq1, b + 1)
        OVN_W2_COMPUTE(wq0, b)
        if (b + 2 < NB) {
          OVN_W2_LOAD(wq0, b + 2)
        }
        OVN_W2_COMPUTE(wq1, b + 1)
      }
#undef OVN_W2_LOAD
#undef OVN_W2_COMPUTE
#undef OVN_W2_READ_A
      f32x4 acc2[2];
      acc2[0] = (acc2t[0][0] + acc2t[0][1]) + acc2t[0][2];
      acc2[1] = (acc2t[1][0] + acc2t[1][1]) + acc2t[1][2];
      const int p = 16 * wave + lrow;
      const float bv = b2[p];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ib2 = 16 * mt + 4 * g + r;
          if (ib2 < G) o2[(((long long)pair * G + ib2) * G + jb) * O2 + p] = fmaxf(acc2[mt][r] + bv, 0.0f);
        }
      }
    }
    }
  }
}

#undef OVN_LOAD_L
#undef OVN_SLICE
#undef OVN_TILE_MFMA
}  // namespace

int ovn_delta_c12_bf16x3_forward(const ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r,
                                 const int32_t* ridx, int n, float* o2, hipStream_t stream) {
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(delta_c12_bf16x3_j2_kernel<3, 8, false>), LDS_BYTES);
  if (rc) return rc;
  // divisors of the 12 passes: time ~ rounds of workgroups over the 256 CUs x 1/d of a pair's work; the smallest d within
  // 5 % of the best (big sweeps keep d = 1: one workgroup per pair, W1 window and R rows set up once)
  int nsplit = 1;
  {
    double best = 1e30;
    for (const int d : {1, 2, 3, 4, 6, 12}) {
      const double cost = (double)(((long long)n * d + 255) / 256) / d;
      if (cost < best) best = cost;
    }
    for (const int d : {1, 2, 3, 4, 6, 12}) {
      const double cost = (double)(((long long)n * d + 255) / 256) / d;
      if (cost <= 1.05 * best) {
        nsplit = d;
        break;
      }
    }
  }
  hipLaunchKernelGGL((delta_c12_bf16x3_j2_kernel<3, 8, false>), dim3(n * nsplit), dim3(512), LDS_BYTES, stream, feats_l, lidx,
                     feats_r, ridx, reinterpret_cast<const __bf16*>(ctx->w1p_bf), ctx->b1,
                     reinterpret_cast<const __bf16*>(ctx->w2p_bf), ctx->c2.bias, o2, 1, nsplit);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

int ovn_delta_prepare_bf16x3(const float* c1_kernel_dev, const float* c2_kernel_dev, void** w1p_out, void** w2p_out,
                             hipStream_t stream) {
  const size_t w1_elems = (size_t)S * FC * O1 * 2;   // hi + lo
  const size_t w2_elems = (size_t)K2 * O2 * 2;
  OVN_HIP_CHECK(hipMalloc(w1p_out, w1_elems * sizeof(__bf16)));
  OVN_HIP_CHECK(hipMalloc(w2p_out, w2_elems * sizeof(__bf16)));
  hipLaunchKernelGGL(delta_prep_w1_bf16_kernel, dim3(240), dim3(256), 0, stream, c1_kernel_dev,
                     reinterpret_cast<__bf16*>(*w1p_out));
  hipLaunchKernelGGL(delta_prep_w2_bf16_kernel, dim3(240), dim3(256), 0, stream, c2_kernel_dev,
                     reinterpret_cast<__bf16*>(*w2p_out));
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
