// Delta head: DeltaLayer + c_conv1 + c_conv2 on the fp16 matrix cores with a scaled 3-term split
// (v_mfma_f32_16x16x32_f16, fp32 accumulate) for gfx950.  Reference: generateNet.py:15-61 (DeltaLayer), :96-106.
//
// Arithmetic ("f16x3"): every fp32 operand x is scaled by a power of two s (exact) and written as hi + lo, both fp16:
// 11 + 11 significand bits, i.e. x*s is carried to 2^-21 relative, or to 2^-25 ABSOLUTE in scaled units once lo falls below
// the fp16 normal range (scales put the largest operand at 2^13..2^14, so that floor is < 2^-38 of it).  a*w is evaluated
// as a_hi*w_hi + a_lo*w_hi + a_hi*w_lo: three MFMAs at the fp16 rate (16x the fp32 matrix rate); the dropped a_lo*w_lo term
// is 2^-22 relative.  Products and sums are fp32 inside the MFMA, the scales are divided out of the fp32 accumulators.
//
// The DeltaLayer in MIN FORM.  c_conv1 needs sum_k |l - r| w.  Forming and splitting |l - r| costs 5-8 VALU instructions per
// element pair next to the 12 MFMAs they feed, and the matrix pipe hides only part of that (tools/experiments/ubench3.hip,
// ubench5.hip).  Instead, with l' = l + c, r' = r + c >= 0 (c = -min(0, smallest value of the pair)):
//     |l - r| = l' + r' - 2 min(l', r')
//   * min COMMUTES with the split: for x >= 0 the word P(x) = fp16_rtz(x) << 16 | fp16_rne(x - hi) orders like x, hence
//     P(min(l', r')) = min_u32(P(l'), P(r')).  Both volumes are packed once per pair by the prepare kernel; the inner loop is
//     ONE v_min_u32 per element plus two v_perm_b32 per element pair that separate the hi and lo halves: 4 full-rate VALU per
//     pair (100 ns per 12 MFMAs in ubench5.hip; the 12 MFMAs alone take 94).
//   * the l' and r' terms are LINEAR, and so is c_conv1 itself (activation='linear', generateNet.py:96-99), so they are
//     pushed through c_conv2 and added to its output per pair:
//       o2pre[ib][jb][p] = b2[p] + TT[ib][p] + AA[jb][p] - 2 sum_{di,o} M[15 ib + di][jb][o] W2[di][o][p]
//       M[i][jb][o]  = sum_{dj,c} min(l'[i][c], r'[15 jb + dj][c]) W1[dj][c][o]          <- delta_c1_f16x3_kernel (99.4 % of the work)
//       TT[ib][p]    = sum_{di,o} (b1[o] + sum_c l'[15 ib + di][c] Ws[c][o]) W2[di][o][p]  <- prepare kernel (Ws = W1 summed over its taps)
//       AA[jb][p]    = sum_o (sum_{dj,c} r'[15 jb + dj][c] W1[dj][c][o]) W2s[o][p]        <- a2 + prepare kernels (W2s = W2 summed over its taps)
//   Numerics: the three terms are ~1.5x larger than their sum, so the fp32 accumulation error is that of an fp32 evaluation
//   times ~2-3 (1e-6 of the largest c_conv1 output; tests/test_parity_sweep.py compares every pair of the benchmark sweep with fp64).
//
// Three kernels per call (plus delta_a2_kernel for the right volumes):
//   delta_prepare_split_kernel  per pair: value range -> shift and scales, both volumes packed into the word streams below, TT, AA
//   delta_c1_f16x3_kernel       the min-term contraction; NO epilogue phase: it stores -2 M (scaled, fp32) and goes on with the
//                               next rows; registers hold the 96 accumulators, the pass's L words and the operand fragments, and
//                               every operand stream (W1 window, packed R words, packed L slices) arrives by LDS-DMA
//                               (global_load_lds_dwordx4): no staging registers, no ds_write pass, no exposed global-load latency
//   delta_c2_f16x3_kernel       c_conv2 as a streaming GEMM over the (n 576) x 960 matrix of -2 M rows (2.2 MB per pair through HBM)
// The one-kernel predecessor (o1 image in LDS, c_conv2 as an epilogue phase of every pass: 5.56 ms per 1024 pairs against
// 4.2 + 0.6 + 0.21 here; its epilogue phase cost 0.93 ms for 0.25 ms of MFMAs, exposed L loads 0.43, W1 register staging 0.35)
// is in the history (tools/experiments/delta_head_f16x3_fused.hip up to the round-3 tree, commit 170ae22).
//
// Scales: weights statically (max |W| -> 2^14), features per PAIR (max over both volumes -> 2^14, so a pair's result does
// not depend on the other pairs of the call), -2 M by the bound 2 span max_o sum|W1[.,o]|, T by |b1|max + span max_o sum|W1[.,o]|.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "ovn_internal.h"
#include "delta_a2.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int FW = OVN_FEAT_W;        // 360
constexpr int FC = OVN_FEAT_C;        // 128
constexpr int S = OVN_S;              // 15
constexpr int G = OVN_G;              // 24
constexpr int O1 = OVN_C1_OUT;        // 64
constexpr int O2 = OVN_C2_OUT;        // 128
constexpr int K1 = S * FC;            // 1920
constexpr int K2 = S * O1;            // 960
constexpr int STEP_BYTES = 8192;      // W1 fragments of one MFMA step: [nt(4)][hi/lo][lane(64)][8 fp16]
constexpr int NWAVE = 8;
constexpr int A2_ELEMS = OVN_A2_ELEMS;             // floats of A2 per right volume and K slice [jb][o] (delta_a2.h)
constexpr int A2_KSPLIT = OVN_A2_KSPLIT;           // K slices (workgroups) per right volume in delta_a2_kernel

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// P(x0), P(x1) for two scaled non-negative values: word = fp16_rtz(x) << 16 | fp16_rne(x - hi).
__device__ __forceinline__ void pack_pair(float x0, float x1, unsigned& w0, unsigned& w1) {
  const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  f16x2 lo;
  lo[0] = (_Float16)(x0 - (float)h[0]);
  lo[1] = (_Float16)(x1 - (float)h[1]);
  const unsigned hp = __builtin_bit_cast(unsigned, h), lp = __builtin_bit_cast(unsigned, lo);
  w0 = __builtin_amdgcn_perm(hp, lp, 0x05040100u);   // h0 << 16 | l0
  w1 = __builtin_amdgcn_perm(hp, lp, 0x07060302u);   // h1 << 16 | l1
}

__device__ __forceinline__ u32x4 pack4(const f32x4& v, float sa, float csa) {
  u32x4 w;
  unsigned a, b;
  pack_pair(fmaf(v[0], sa, csa), fmaf(v[1], sa, csa), a, b);
  w[0] = a;
  w[1] = b;
  pack_pair(fmaf(v[2], sa, csa), fmaf(v[3], sa, csa), a, b);
  w[2] = a;
  w[3] = b;
  return w;
}

// R words of the PACKED last slice of a compacted walk (TailPack, below) for all 24 column groups: item = (column group jb, step u,
// lane group gq) -> 8 words [jb][slice][u][gq][0..7], entry k = 8 gq + e = tap (u T + k / n) of channel chan[32 slice + k]; pad
// entries are zero words.  512 threads.
struct TailPack;
__device__ __forceinline__ int tail_tap(const TailPack& t, int u, int k);
template <class TP>
__device__ __forceinline__ void pack_tail_slice(const float* __restrict__ Rf, const unsigned char* chan, const TP& tp, unsigned* __restrict__ Pr,
                                                float sa, float csa, int tid) {
  const int items = OVN_G * tp.steps * 4;
  for (int it = tid; it < items; it += 512) {
    const int gq = it & 3, u = (it >> 2) % tp.steps, jb = (it >> 2) / tp.steps;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int tap = tail_tap(tp, u, 8 * gq + e);
      const float v = Rf[(size_t)(OVN_S * jb + (tap < 0 ? 0 : tap)) * OVN_FEAT_C + chan[32 * tp.sl + 8 * gq + e]];
      x[e] = tap < 0 ? 0.0f : v;        // a pad entry packs to the zero word (a packed slice exists only without a shift: csa = 0)
    }
    unsigned* dst = Pr + ((jb * 4 + tp.sl) * OVN_S + u) * 32 + gq * 8;
    *reinterpret_cast<u32x4*>(dst) = pack4((f32x4){x[0], x[1], x[2], x[3]}, sa, csa);
    *reinterpret_cast<u32x4*>(dst + 4) = pack4((f32x4){x[4], x[5], x[6], x[7]}, sa, csa);
  }
}

// (x0, x1), already scaled -> packed fp16 hi pair (rtz) and lo pair (rne of the exact remainder; `one` = 1.0f in a register keeps
// the fma from being folded into a subtraction that needs two more conversions)
__device__ __forceinline__ void split_pair(float x0, float x1, float one, unsigned& hi_pk, unsigned& lo_pk) {
  const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  f16x2 l;
  l[0] = (_Float16)__builtin_fmaf(x0, one, -(float)h[0]);
  l[1] = (_Float16)__builtin_fmaf(x1, one, -(float)h[1]);
  hi_pk = __builtin_bit_cast(unsigned, h);
  lo_pk = __builtin_bit_cast(unsigned, l);
}

// A fragments (hi, lo) of min(l, r) for one 16-row tile and one MFMA step: 8 packed words per lane each side.
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

// out[0] = max |w|, out[1] = max over columns n of sum_k |w[k][n]| for a row-major [K][N] matrix; one workgroup.
__global__ __launch_bounds__(256) void delta_wstats_kernel(const float* __restrict__ w, int K, int N, float* __restrict__ out) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  float amax = 0.f, cmax = 0.f;
  for (int n = tid; n < N; n += 256) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
      const float v = fabsf(w[(size_t)k * N + n]);
      s += v;
      amax = fmaxf(amax, v);
    }
    cmax = fmaxf(cmax, s);
  }
  red[tid] = amax;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  const float a = red[0];
  __syncthreads();
  red[tid] = cmax;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  if (tid == 0) {
    out[0] = a;
    out[1] = red[0];
  }
}

// w1sum[c][o] = sum_dj W1[dj][c][o] (fp64 accumulation, rounded once) and w1col[o] = sum_{dj,c} W1[dj][c][o].
__global__ __launch_bounds__(256) void delta_w1sum_kernel(const float* __restrict__ w1, float* __restrict__ w1sum, float* __restrict__ w1col) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < FC * O1) {
    double s = 0.0;
    for (int dj = 0; dj < S; ++dj) s += (double)w1[(size_t)dj * FC * O1 + idx];
    w1sum[idx] = (float)s;
  }
  if (idx < O1) {
    double s = 0.0;
    for (int k = 0; k < K1; ++k) s += (double)w1[(size_t)k * O1 + idx];
    w1col[idx] = (float)s;
  }
}

// W1p[u = s*15 + dj][nt(4)][hl(2)][lane(64)][e(8)]: sw * W1[dj][c = 32*(lane>>4) + 8*s + e][o = 16*nt + (lane&15)]
__global__ void delta_prep_w1_f16_kernel(const float* __restrict__ w1, _Float16* __restrict__ w1p, float sw) {
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
    _Float16 hi, lo;
    split_f16(sw * w1[(dj * FC + c) * O1 + o], hi, lo);
    const size_t base = (((size_t)u * 4 + nt) * 2) * 512 + lane * 8 + e;
    w1p[base] = hi;
    w1p[base + 512] = lo;
  }
}

// W2p[ks(30)][nt(8)][hl(2)][lane(64)][e(8)]: sw * W2[k(k')][p = 16*nt + (lane&15)], k' = 32*ks + 8*(lane>>4) + e.
// GEMM2 walks its K axis in the order k' = di*64 + 4*(o & 15) + (o >> 4) instead of k = di*64 + o: the four c_conv1
// n-tiles a lane holds after GEMM1 (o = lrow, 16+lrow, 32+lrow, 48+lrow) are then adjacent in the o1 image, so the
// epilogue stores 8 bytes per (row, hi/lo) instead of four 2-byte pieces.  Any K order works as long as A and B agree.
__global__ void delta_prep_w2_f16_kernel(const float* __restrict__ w2, _Float16* __restrict__ w2p, float sw) {
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
    _Float16 hi, lo;
    split_f16(sw * w2[k * O2 + p], hi, lo);
    const size_t base = (((size_t)ks * 8 + nt) * 2) * 512 + lane * 8 + e;
    w2p[base] = hi;
    w2p[base + 512] = lo;
  }
}

// A2raw[v][ksl][jb][o] for right volume v (v = ridx[b] if ridx else 0): grid (volumes, K slices), wave = tile; the task itself is
// ovn_delta_a2_task (delta_a2.h), which the yaw kernel of small 1-vs-N sweeps runs in extra workgroups of its own launch instead.
__global__ __launch_bounds__(512) void delta_a2_kernel(const float* __restrict__ feats_r, const int32_t* __restrict__ ridx,
                                                       const float* __restrict__ w1raw, float* __restrict__ a2raw) {
  const int b = blockIdx.x, ksl = blockIdx.y;
  const float* R = feats_r + (long long)(ridx ? ridx[b] : 0) * OVN_FEAT_ELEMS;
  ovn_delta_a2_task(R, w1raw, a2raw + (size_t)b * A2_KSPLIT * A2_ELEMS, ksl * 8 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
}

// WsP[ks(4)][nt(4)][hl(2)][lane(64)][e(8)]: sws * Ws[c = 32 ks + 8 (lane>>4) + e][o = 16 nt + (lane&15)], Ws = W1 summed over its taps
__global__ void delta_prep_ws_f16_kernel(const float* __restrict__ w1sum, _Float16* __restrict__ wsp, float sws) {
  const int total = 4 * 4 * 64 * 8;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 7;
    const int lane = (idx >> 3) & 63;
    const int nt = (idx >> 9) & 3;
    const int ks = idx >> 11;
    _Float16 hi, lo;
    split_f16(sws * w1sum[(32 * ks + 8 * (lane >> 4) + e) * O1 + 16 * nt + (lane & 15)], hi, lo);
    const size_t base = (((size_t)ks * 4 + nt) * 2) * 512 + lane * 8 + e;
    wsp[base] = hi;
    wsp[base + 512] = lo;
  }
}

constexpr int TT_STRIDE = K2 + 8;                   // fp16 elements per row of the T image in LDS (1936 B: 16-B reads of 16 rows spread over all banks)
constexpr int LIN_ELEMS = 2 * G * O2;               // per pair: TT + b2 [24][128], AA [24][128]
constexpr size_t O1RAW_ELEMS = (size_t)G * FW * O1; // per pair: 552,960 floats = 3 tiles of 192 (jb, ib) rows x 960
constexpr int C2_TILE_ROWS = 192;                   // rows (jb % 8, ib) of one c_conv2 workgroup; o1raw is stored [tile][k-step][row][32]
constexpr size_t PREP_SPLIT_LDS = 2 * (size_t)G * TT_STRIDE * sizeof(_Float16) + ((size_t)G * O1 + 2 * NWAVE) * sizeof(float) + (size_t)FC * O1 * 2 * sizeof(_Float16);

// Where the contraction and c_conv2 kernels find a pair's operands: the per-pair scratch filled by the prepare kernel, or -- for a
// sweep over candidates with a Delta cache (ovn_delta_cache) -- the candidate's cached row and the query's shared operands.
struct DeltaDesc {
  const unsigned* pl;   // packed L words, channel-major [c(128)][i(360)]
  const unsigned* pr;   // packed R words [jb(24)][sc(4)][dj(15)][g(4)][8] in the order of the pair's (compacted) K walk
  const float* tt;      // TT + b2 [24][128]
  const float* aa;      // AA [24][128]
};
static_assert(sizeof(DeltaDesc) == 32, "DeltaDesc is read as two 16-byte words");

// Delta cache row of one candidate (OVN_DELTA_CACHE_ELEMS floats): packed words at the candidate's OWN scale (valid for a pair
// whenever the query's largest value does not reach the next power of two and nothing is negative), TT + b2, {max, min}.
constexpr int DC_PL = 0;                             // [46080] words
constexpr int DC_TT = OVN_FEAT_ELEMS;                // [24 * 128] floats
constexpr int DC_META = DC_TT + G * O2;              // {max L, min L, 0, 0}
static_assert(DC_META + 4 <= OVN_DELTA_CACHE_ELEMS, "Delta cache row too small");
constexpr int QV = 8;                                // packed versions of the query: scales sa_q, sa_q / 2, ... sa_q / 128
constexpr int QGW = 24;                              // further workgroups of delta_query_kernel that only help gather the W1 fragments
// per-query operands shared by every cached pair of a sweep: [QV][46080] packed words | AA [24][128] | {max R, min R, 0, 0}
constexpr size_t QBLOCK_WORDS = (size_t)QV * OVN_FEAT_ELEMS + G * O2 + 4;

// ---- dead-channel compaction of a query (round 5) ---------------------------------------------------------------------------------
// min(l', r') = 0 for EVERY candidate row wherever the query's word r'[j][c] is 0, and a feature channel that is 0 in all 360 columns
// of the query (a quarter of the 128 channels under the benchmark's weights: ReLU outputs) drops out of the contraction altogether --
// for every column group, every tap and every candidate of the sweep.  The K walk then covers ns < 4 slices of 32 LIVE channels
// (dead ones pad the last slice): W1 fragments gathered once per query for those channels, the query's words packed in the same
// order, and the candidates' L words -- stored CHANNEL-MAJOR ([c][360], in cache rows and scratch alike) -- fetched by the same list.
// Exact (the dropped products are exact zeros); only the grouping of K into MFMA steps changes.  The list keeps the live channels in
// the order of the uncompacted walk (slice s, lane group g, element e <-> channel 32 g + 8 s + e), so a query without dead channels
// -- and any pair that needs a shift (negative values: r' = r + c has no zeros) -- walks exactly the K of rounds 2-4.
constexpr int NPAIR = G / 2;               // column-group pairs (the passes of the contraction kernel)
constexpr int LIVE_WORDS = 4 + FC / 4 + 4; // {largest slice count, live channels, 0, 0} | 128 channel bytes: position 32 s + 8 g + e of the
                                           // compacted walk | slices to walk for column groups 2 p, 2 p + 1 (12 bytes, + 4 of padding)
constexpr int CHAN_BYTES = FC + 16;        // the table in LDS: channel bytes + per-pair slice counts + the packed last slice (below)
// The LAST slice of a compacted walk, when it holds n <= 16 live channels, is packed TAP-MAJOR: a step then carries T = 32 / n taps of
// those n channels (entry k of step u = tap u T + k / n of channel k % n) and the slice has ceil(15 / T) steps (rounded up to whole
// chunks of 3) instead of 15 -- a query with 99 live channels walks 3 x 15 + 3 steps, not 4 x 15.  Table bytes FC + 12 .. FC + 15:
// {steps of the packed slice (15: nothing is packed), its index (255: none), T, n}; the table's 32 positions of that slice are
// written in ENTRY order (channel of entry k), so the L slice the contraction kernel fetches by the table already is the A operand's
// order and the kernel itself only needs the step count.  Pad entries (k >= T n, tap >= 15) carry zero R words and zero weights.
constexpr int TB_STEPS = FC + 12, TB_SLICE = FC + 13, TB_T = FC + 14, TB_N = FC + 15;
struct TailPack {
  int sl, steps, T, n;   // sl < 0: no packed slice
};
__device__ __forceinline__ TailPack tail_of(const unsigned char* chan) {
  TailPack t;
  t.sl = chan[TB_SLICE] == 255 ? -1 : (int)chan[TB_SLICE];
  t.steps = chan[TB_STEPS];
  t.T = chan[TB_T];
  t.n = chan[TB_N];
  return t;
}
// tap of entry k of step u of the packed slice, or -1 for a pad entry
__device__ __forceinline__ int tail_tap(const TailPack& t, int u, int k) {
  const int tap = u * t.T + k / t.n;
  return (k < t.T * t.n && tap < S) ? tap : -1;
}
__device__ __forceinline__ int ident_chan(int pos) { return 32 * ((pos >> 3) & 3) + 8 * (pos >> 5) + (pos & 7); }
__device__ __forceinline__ unsigned ident_chan_word(int w) {
  return (unsigned)ident_chan(4 * w) | (unsigned)ident_chan(4 * w + 1) << 8 | (unsigned)ident_chan(4 * w + 2) << 16 | (unsigned)ident_chan(4 * w + 3) << 24;
}
// channel table of a workgroup -> LDS (`use`: workgroup-uniform); returns the number of slices.  Caller synchronises.
__device__ __forceinline__ int load_chan_table(const unsigned* __restrict__ live, bool use, unsigned char* chan_s, int tid) {
  if (tid < FC / 4) reinterpret_cast<unsigned*>(chan_s)[tid] = use ? live[4 + tid] : ident_chan_word(tid);
  else if (tid < FC / 4 + 3) reinterpret_cast<unsigned*>(chan_s)[tid] = use ? live[4 + tid] : 0x04040404u;
  else if (tid == FC / 4 + 3) reinterpret_cast<unsigned*>(chan_s)[tid] = use ? live[4 + tid] : (15u | 255u << 8 | 1u << 16 | 32u << 24);
  return use ? (int)live[0] : 4;
}

// Channel list of a query from its flags alive2[p][ch] (channel ch is non-zero somewhere in the 30 query columns of column groups 2 p,
// 2 p + 1; `shifted`: the volume has a negative value -> everything counts as alive).  The channels are ordered by the NUMBER OF
// PAIRS p they are alive in (descending; ties and the dead ones in walk order -- a query alive everywhere gets the identity), and pair p
// walks only up to the last position that is alive in it: nsp[p] = ceil(that / 32) slices.  One list, one set of gathered weights, and
// a column-group pair whose own live channels fit fewer slices than the query's walks fewer (the benchmark's query: 94 live channels,
// 3 slices everywhere; under the trained-like weights 99 live -> 4 slices as a whole, but 3 in ten of its twelve pairs).
// Every thread of the (512-thread) workgroup must call it.  `scr`: 2 FC + 32 ints of LDS.
// Leaves chan_s[0 .. 127], chan_s[FC + p] and the packed-slice bytes; returns the number of live channels.
__device__ __forceinline__ int build_chan_list(const int (*alive2)[FC], bool shifted, unsigned char* chan_s, int* scr, int tid) {
  int* cnt_s = scr;            // [FC] pairs position q is alive in
  int* nmax = scr + FC;        // [2 waves][16]: walk length of pair p = 0 .. 11; [12] live channels
  int* rank_s = scr + FC + 32; // [FC] rank of position q, summed over the four quarters of the comparison range
  const int pos = tid & (FC - 1), quarter = tid >> 7;   // 512 threads: position x quarter of the positions it is compared with
  const int ch = ident_chan(pos);
  int cnt = 0;
  if (tid < FC) {
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) cnt += (shifted || alive2[p][ch] != 0) ? 1 : 0;
    cnt_s[tid] = cnt;
    rank_s[tid] = 0;
  }
  __syncthreads();
  cnt = cnt_s[pos];
  {
    // (the 128 comparisons of a position as 4 x 32 over the workgroup, reads batched: as one rolled loop on 128 threads every LDS read
    //  waited for the previous one -- 6 us of a kernel that stands in front of every sweep)
    int part = 0;
#pragma unroll 8
    for (int k = 0; k < FC / 4; ++k) {
      const int q = (FC / 4) * quarter + k;
      const int cq = cnt_s[q];
      part += (cq > cnt || (cq == cnt && q < pos)) ? 1 : 0;
    }
    if (part) atomicAdd(&rank_s[pos], part);
  }
  __syncthreads();
  if (tid < FC) {   // (waves 0 and 1, whole)
    const int rank = rank_s[tid];
    chan_s[rank] = (unsigned char)ch;
    // walk length of every pair and the live count as WAVE reductions (shuffles), one LDS word per wave and quantity: written as LDS
    // atomics the compiler's atomic optimiser turns each of the 13 into a serial loop over the 64 lanes -- 17 of the kernel's 35 us
    int v[NPAIR + 1];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) v[p] = (shifted || alive2[p][ch] != 0) ? rank + 1 : 0;
    v[NPAIR] = cnt > 0 ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) v[p] = max(v[p], __shfl_xor(v[p], off, 64));
      v[NPAIR] += __shfl_xor(v[NPAIR], off, 64);
    }
    if ((tid & 63) == 0) {
#pragma unroll
      for (int p = 0; p <= NPAIR; ++p) nmax[16 * (tid >> 6) + p] = v[p];
    }
  }
  __syncthreads();
  const int nlive = nmax[NPAIR] + nmax[16 + NPAIR];
  // the packed last slice (TailPack above): live channels hold ranks 0 .. nlive - 1
  const int nsl = nlive > 0 ? (nlive + 31) / 32 : 1;
  const int n_last = nlive - 32 * (nsl - 1);
  int t_steps = S, t_T = 1;
  if (!shifted && nlive > 0 && n_last <= 16) {
    t_T = 32 / n_last;
    t_steps = 3 * (((S + t_T - 1) / t_T + 2) / 3);
  }
  const bool packed = t_steps < S;
  if (tid < 16) {
    int ns = tid < NPAIR ? (max(nmax[tid], nmax[16 + tid]) + 31) / 32 : 0;
    unsigned char v = (unsigned char)(tid < NPAIR ? (ns < 1 ? 1 : ns) : 4);
    if (tid == TB_STEPS - FC) v = (unsigned char)t_steps;
    if (tid == TB_SLICE - FC) v = (unsigned char)(packed ? nsl - 1 : 255);
    if (tid == TB_T - FC) v = (unsigned char)t_T;
    if (tid == TB_N - FC) v = (unsigned char)(packed ? n_last : 32);
    chan_s[FC + tid] = v;
  }
  if (packed && tid >= 64 && tid < 96) {   // entry order: position k of the slice <- the channel of entry k (k % n)
    const int k = tid - 64;
    if (k >= n_last) chan_s[32 * (nsl - 1) + k] = chan_s[32 * (nsl - 1) + k % n_last];
  }
  __syncthreads();
  return nlive;
}

// One workgroup per query: which channels are alive, in walk order (1-vs-N sweeps WITHOUT Delta cache rows: with them the query
// kernel below builds the list itself).
__global__ __launch_bounds__(512) void delta_live_kernel(const float* __restrict__ feats_r, unsigned* __restrict__ live) {
  __shared__ int alive2[NPAIR][FC];
  __shared__ int neg_s;
  __shared__ int scr[2 * FC + 32];
  __shared__ __attribute__((aligned(16))) unsigned char chan_s[CHAN_BYTES];
  const int tid = threadIdx.x, c = tid & (FC - 1), part = tid >> 7;
  for (int i = tid; i < NPAIR * FC; i += 512) (&alive2[0][0])[i] = 0;
  if (tid == 0) neg_s = 0;
  __syncthreads();
  int ng = 0;
  for (int j = part; j < FW; j += 4) {
    const float v = feats_r[(size_t)j * FC + c];
    if (v != 0.0f) alive2[j / (2 * S)][c] = 1;
    ng |= (v < 0.0f);
  }
  if (ng) neg_s = 1;
  __syncthreads();
  const int nlive = build_chan_list(alive2, neg_s != 0, chan_s, scr, tid);
  if (tid == 0) {
    int nsm = 1;
    for (int p = 0; p < NPAIR; ++p) nsm = chan_s[FC + p] > nsm ? chan_s[FC + p] : nsm;
    live[0] = (unsigned)nsm;
    live[1] = (unsigned)nlive;
    live[2] = live[3] = 0u;
  }
  if (tid < FC / 4 + 4) live[4 + tid] = reinterpret_cast<const unsigned*>(chan_s)[tid];
}

// 8 consecutive elements (one lane's 16 bytes) of the compacted W1 fragments: out[sc][dj][nt][hi,lo][lane][0..7] <- w1p[...] of the
// channels chan[32 sc + 8 (lane >> 4) + e]
__device__ __forceinline__ void w1c_gather8(const _Float16* __restrict__ w1p, const unsigned char* chan, int idx8, _Float16* __restrict__ w1c) {
  const int lane = idx8 & 63, hl = (idx8 >> 6) & 1, nt = (idx8 >> 7) & 3;
  const int step = idx8 >> 9;                   // sc * 15 + dj
  const int sc = step / S, dj = step - sc * S;
  const unsigned char* cp = chan + 32 * sc + 8 * (lane >> 4);
  const TailPack tp = tail_of(chan);
  f16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = cp[e];
    const int tap = (sc == tp.sl) ? tail_tap(tp, dj, 8 * (lane >> 4) + e) : dj;   // packed slice: `dj` is its step
    const int src_step = ((ch & 31) >> 3) * S + (tap < 0 ? 0 : tap);
    const _Float16 w = w1p[((((size_t)src_step * 4 + nt) * 2 + hl) * 64 + (lane & 15) + 16 * (ch >> 5)) * 8 + (ch & 7)];
    v[e] = tap < 0 ? (_Float16)0.0f : w;
  }
  *reinterpret_cast<f16x8*>(w1c + (size_t)idx8 * 8) = v;
}

// W1 fragments of the compacted walk (the no-cache route; see delta_live_kernel)
__global__ __launch_bounds__(256) void delta_w1c_kernel(const _Float16* __restrict__ w1p, const unsigned* __restrict__ live,
                                                        _Float16* __restrict__ w1c) {
  const int ns = (int)live[0];
  const unsigned char* chan = reinterpret_cast<const unsigned char*>(live + 4);
  const int total8 = ns * S * 4 * 2 * 64;
  for (int idx8 = blockIdx.x * 256 + threadIdx.x; idx8 < total8; idx8 += gridDim.x * 256) w1c_gather8(w1p, chan, idx8, w1c);
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  // LDS-DMA: lane l's 16 bytes land at lds_wave_base + 16 l (the base is wave-uniform: it goes through M0)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Per pair: value range -> shift and scales; both volumes packed ONCE into the word streams the c_conv1 kernel DMAs into LDS
//   pl[pair][c(128)][i(360)]              L words, CHANNEL-major: the contraction kernel fetches 32 rows of the channels its walk names
//   pr[pair][jb(24)][s(4)][dj(15)][g(4)][8] R words in the order of the K walk: a chunk of 3 taps of a column group is 384 contiguous bytes
// and the linear terms lin[pair] = {TT + b2 [24][128], AA [24][128]} (fp16 MFMA on the T image in LDS / plain FMAs).
// scales[2 pair] = {sa, -2 s1r / (sa sw1), s1r, 1 / (s1r sw2)}; s1r = scale of -2 M, bounded by 2 span max_o sum |W1[., o]|.
// One workgroup per pair and one workgroup per CU (129 KB of LDS), so nothing hides a memory round trip: every phase issues ALL of
// a thread's loads before it uses the first (the left volume stays in registers from the range scan to the packing; a loop of
// load / use / load costs one round trip per iteration and made this kernel 0.33 ms per 1024 pairs instead of 0.21).
//
// BUILD = true is the same kernel run per CANDIDATE to fill its Delta cache row (ovn_delta_cache): packed L words at the candidate's
// own scale, TT + b2 and {max, min} -- everything of a pair's preparation that does not depend on the query as long as no value is
// negative and the query's largest value stays below the candidate's next power of two.  BUILD = false with `dcache` given (1-vs-N
// sweeps) checks exactly that condition per pair and, when it holds, only writes the pair's scales and its operand descriptor
// (cache row + the query's shared block); otherwise it prepares the pair in scratch as before.  Both routes give the same bits:
// the scales are functions of the power-of-two bucket of the pair's largest value, never of the value itself.
template <bool BUILD>
__global__ __launch_bounds__(512) void delta_prepare_split_kernel(
    const float* __restrict__ feats_l, const int32_t* __restrict__ lidx, const float* __restrict__ feats_r,
    const int32_t* __restrict__ ridx, const _Float16* __restrict__ wsp, const float* __restrict__ w1col,
    const float* __restrict__ b1, const float* __restrict__ a2raw, const _Float16* __restrict__ w2p,
    const float* __restrict__ w2sum, const float* __restrict__ b2, float sw1, float sw2, float sws, float w1_colsum, float b1_absmax,
    float one, f32x4* __restrict__ scales, unsigned* __restrict__ o2max, unsigned* __restrict__ pl, unsigned* __restrict__ pr,
    float* __restrict__ lin, DeltaDesc* __restrict__ desc, const float* __restrict__ dcache, const unsigned* __restrict__ qblock,
    float* __restrict__ cache_out, const unsigned* __restrict__ live) {
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  __shared__ __attribute__((aligned(16))) unsigned char chan_p[CHAN_BYTES];
  // T image, scaled fp16 hi / lo: T[15 ib + di][o] at [ib][di * 64 + 4 (o & 15) + (o >> 4)] (the K order of W2p)
  _Float16* Th = reinterpret_cast<_Float16*>(psm);
  _Float16* Tlo = Th + G * TT_STRIDE;
  float* A2l = reinterpret_cast<float*>(Tlo + G * TT_STRIDE);   // [24][64]
  float* red = A2l + G * O1;                                    // [2][NWAVE]
  _Float16* wsl = reinterpret_cast<_Float16*>(red + 2 * NWAVE); // Ws fragments [ks(4)][nt(4)][hl(2)][lane(64)][8]: 32 KB
  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, g = lane >> 4;
  const long long cand = lidx ? lidx[pair] : pair;
  if (!BUILD && dcache) {   // workgroup-uniform: is the candidate's cache row valid for this query?
    const float* row = dcache + cand * OVN_DELTA_CACHE_ELEMS;
    const float* qm = reinterpret_cast<const float*>(qblock + (size_t)QV * OVN_FEAT_ELEMS + G * O2);
    const float mxl = row[DC_META], mnl = row[DC_META + 1], mxr = qm[0], mnr = qm[1];
    const float sa = ovn_pow2_scale_for(mxl);
    const float saq = ovn_pow2_scale_for(mxr);
    int eq = 0, el = 0;
    (void)frexpf(saq, &eq);
    (void)frexpf(sa, &el);
    const int ver = eq - el;   // the query version packed with sa: saq / 2^ver
    if (mnl >= 0.0f && mnr >= 0.0f && ovn_pow2_scale_for(fmaxf(mxl, mxr)) == sa && ver >= 0 && ver < QV) {
      if (tid == 0) {
        const float bc = 16384.0f / sa;
        const float s1r = ovn_pow2_scale_for(2.0f * bc * w1_colsum);
        scales[2 * pair] = (f32x4){sa, -2.0f * s1r / (sa * sw1), s1r, 1.0f / (s1r * sw2)};
        scales[2 * pair + 1] = (f32x4){0.f, 0.f, fmaxf(mxl, mxr), 1.f};
        o2max[pair] = 0u;
        DeltaDesc d;
        d.pl = reinterpret_cast<const unsigned*>(row) + DC_PL;
        d.pr = qblock + (size_t)ver * OVN_FEAT_ELEMS;
        d.tt = row + DC_TT;
        d.aa = reinterpret_cast<const float*>(qblock + (size_t)QV * OVN_FEAT_ELEMS);
        desc[pair] = d;
      }
      return;
    }
  }
  const float* Lf = feats_l + cand * OVN_FEAT_ELEMS;
  const float* Rf = BUILD ? Lf : feats_r + (long long)(ridx ? ridx[pair] : 0) * OVN_FEAT_ELEMS;   // BUILD: no right volume
  const f32x4* R4 = reinterpret_cast<const f32x4*>(Rf);

  // ---- all loads of the first phase in flight at once ----
  // L in MFMA A-fragment order: wave w, row tiles 3w .. 3w+2; a lane owns row lrow of the tile and channels 32 ks + 8 g .. + 7 of
  // each 32-channel MFMA step, i.e. main-kernel lane group ks of slice g.  96 registers, kept until the words are packed.
  f32x4 lv[3][4][2];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int i = 48 * wave + 16 * t + lrow;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (i < FW) {
        const float* src = Lf + (size_t)i * FC + 32 * ks + 8 * g;
        lv[t][ks][0] = *reinterpret_cast<const f32x4*>(src);
        lv[t][ks][1] = *reinterpret_cast<const f32x4*>(src + 4);
      } else {
        lv[t][ks][0] = lv[t][ks][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  // Ws fragments -> LDS (4 x 16 B per thread), A2 K slices (3 elements x 8 slices per thread)
  f32x4 wsv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wsv[q] = reinterpret_cast<const f32x4*>(wsp)[tid + 512 * q];
  float a2v[3][A2_KSPLIT] = {};
  if (!BUILD) {
    const float* src = a2raw + (size_t)(ridx ? pair : 0) * A2_KSPLIT * A2_ELEMS;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int k = 0; k < A2_KSPLIT; ++k) a2v[u][k] = src[(size_t)k * A2_ELEMS + tid + 512 * u];
  }
  // R: range scan now (12 x 32 B per thread, two batches of loads), packed in a second pass once the scale is known
  constexpr int R_ITEMS = OVN_FEAT_ELEMS / 8;   // 5760 = 11.25 x 512
  float mx = -3.0e38f, mn = 3.0e38f;
#pragma unroll
  for (int half = 0; half < (BUILD ? 0 : 2); ++half) {
    f32x4 rv[6][2];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int i8 = tid + 512 * (6 * half + k);
      if (i8 < R_ITEMS) {
        rv[k][0] = R4[2 * i8];
        rv[k][1] = R4[2 * i8 + 1];
      } else {
        rv[k][0] = rv[k][1] = R4[0];
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        mx = fmaxf(mx, fmaxf(fmaxf(rv[k][h][0], rv[k][h][1]), fmaxf(rv[k][h][2], rv[k][h][3])));
        mn = fminf(mn, fminf(fminf(rv[k][h][0], rv[k][h][1]), fminf(rv[k][h][2], rv[k][h][3])));
      }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (48 * wave + 16 * t + lrow < FW) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 a = lv[t][ks][h];
          mx = fmaxf(mx, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
          mn = fminf(mn, fminf(fminf(a[0], a[1]), fminf(a[2], a[3])));
        }
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mx = fmaxf(mx, __shfl_down(mx, off, 64));
    mn = fminf(mn, __shfl_down(mn, off, 64));
  }
  if (lane == 0) {
    red[wave] = mx;
    red[NWAVE + wave] = mn;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) reinterpret_cast<f32x4*>(wsl)[tid + 512 * q] = wsv[q];
  __syncthreads();
  mx = red[0];
  mn = red[NWAVE];
#pragma unroll
  for (int w = 1; w < NWAVE; ++w) {
    mx = fmaxf(mx, red[w]);
    mn = fminf(mn, red[NWAVE + w]);
  }
  const float c = (mn < 0.0f) ? -mn : 0.0f;       // shift that makes both volumes non-negative
  const float span = mx + c;                       // largest shifted value
  const float sa = ovn_pow2_scale_for(span);
  // every further scale is a function of the power-of-two BUCKET of span (bc = 2^14 / sa >= span), not of span itself: a candidate's
  // cached products (BUILD) are then bit for bit what this kernel computes for any pair in which the candidate sets the bucket
  const float bc = 16384.0f / sa;
  const float s1r = ovn_pow2_scale_for(2.0f * bc * w1_colsum);
  const float sT = ovn_pow2_scale_for(b1_absmax + bc * w1_colsum);   // |T| <= |b1| + span max_o sum |W1[., o]|
  const float csa = c * sa;
  float* crow = BUILD ? cache_out + (size_t)pair * OVN_DELTA_CACHE_ELEMS : nullptr;
  if (tid == 0) {
    if (BUILD) {
      *reinterpret_cast<f32x4*>(crow + DC_META) = (f32x4){mx, mn, 0.f, 0.f};
    } else {
      scales[2 * pair] = (f32x4){sa, -2.0f * s1r / (sa * sw1), s1r, 1.0f / (s1r * sw2)};
      scales[2 * pair + 1] = (f32x4){csa, c, span, (c == 0.0f) ? 1.f : 0.f};   // [3]: no shift -> the query's compacted K walk applies
      o2max[pair] = 0u;
      DeltaDesc d;
      d.pl = pl + (size_t)pair * OVN_FEAT_ELEMS;
      d.pr = pr + (size_t)pair * OVN_FEAT_ELEMS;
      d.tt = lin + (size_t)pair * LIN_ELEMS;
      d.aa = lin + (size_t)pair * LIN_ELEMS + G * O2;
      desc[pair] = d;
    }
  }
  // A2[jb][o] of this pair (true units): sum of the K slices of delta_a2_kernel (fixed order) + c (column sums of W1)
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    float v = a2v[u][0];
#pragma unroll
    for (int k = 1; k < A2_KSPLIT; ++k) v += a2v[u][k];
    const int i = tid + 512 * u;
    A2l[i] = v + c * w1col[i & (O1 - 1)];
  }
  // R words in the order of the pair's K walk (second read of R: L2 hits): the query's compacted channel list when the pair has no
  // shift (1-vs-N sweeps; `live` is NULL for indexed pairs), the plain slice order otherwise
  if (!BUILD) {
    const bool compact = live != nullptr && c == 0.0f;
    const int ns = load_chan_table(live, compact, chan_p, tid);
    __syncthreads();
    unsigned* Pr = pr + (size_t)pair * OVN_FEAT_ELEMS;
    const TailPack tp = tail_of(chan_p);
    if (tp.sl >= 0) pack_tail_slice(Rf, chan_p, tp, Pr, sa, csa, tid);
#pragma unroll 2
    for (int k = 0; k < 12; ++k) {
      const int i8 = tid + 512 * k;
      if (i8 < R_ITEMS) {
        const int jrow = i8 >> 4, sc = (i8 >> 2) & 3, gq = i8 & 3;
        if (sc < ns && sc != tp.sl) {
          const int jb = jrow / S, dj = jrow - jb * S;
          const float* rrow = Rf + (size_t)jrow * FC;
          const unsigned char* ch = chan_p + 32 * sc + 8 * gq;
          f32x4 v0, v1;
          if (!compact) {   // 8 consecutive channels 32 gq + 8 sc ..
            v0 = *reinterpret_cast<const f32x4*>(rrow + 32 * gq + 8 * sc);
            v1 = *reinterpret_cast<const f32x4*>(rrow + 32 * gq + 8 * sc + 4);
          } else {
            v0 = (f32x4){rrow[ch[0]], rrow[ch[1]], rrow[ch[2]], rrow[ch[3]]};
            v1 = (f32x4){rrow[ch[4]], rrow[ch[5]], rrow[ch[6]], rrow[ch[7]]};
          }
          unsigned* dst = Pr + ((jb * 4 + sc) * S + dj) * 32 + gq * 8;
          *reinterpret_cast<u32x4*>(dst) = pack4(v0, sa, csa);
          *reinterpret_cast<u32x4*>(dst + 4) = pack4(v1, sa, csa);
        }
      }
    }
  }
  // L words (from the registers) + T = b1 + (L + c) Ws on the fp16 matrix cores, Ws fragments from LDS
  unsigned* P = BUILD ? reinterpret_cast<unsigned*>(crow) + DC_PL : pl + (size_t)pair * OVN_FEAT_ELEMS;
  const float inv_t = 1.0f / (sa * sws);
  float add[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) add[nt] = b1[16 * nt + lrow];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int i = 48 * wave + 16 * t + lrow;
    if (16 * (3 * wave + t) >= FW) continue;   // wave-uniform: the 24th row tile does not exist
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 w0, w1;
      if (i < FW) {
        w0 = pack4(lv[t][ks][0], sa, csa);
        w1 = pack4(lv[t][ks][1], sa, csa);
        unsigned* dst = P + (size_t)(32 * ks + 8 * g) * FW + i;   // channel-major: channels 32 ks + 8 g .. + 7 of row i
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dst[(size_t)e * FW] = w0[e];
          dst[(size_t)(4 + e) * FW] = w1[e];
        }
      } else {
        w0 = w1 = (u32x4){0u, 0u, 0u, 0u};
      }
      u32x4 h, q;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        h[p] = __builtin_amdgcn_perm(w0[2 * p + 1], w0[2 * p], 0x07060302u);
        q[p] = __builtin_amdgcn_perm(w0[2 * p + 1], w0[2 * p], 0x05040100u);
        h[2 + p] = __builtin_amdgcn_perm(w1[2 * p + 1], w1[2 * p], 0x07060302u);
        q[2 + p] = __builtin_amdgcn_perm(w1[2 * p + 1], w1[2 * p], 0x05040100u);
      }
      const f16x8 ah = __builtin_bit_cast(f16x8, h), al = __builtin_bit_cast(f16x8, q);
      const _Float16* wk = wsl + (size_t)ks * (4 * 2 * 512) + lane * 8;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f16x8 bh = *reinterpret_cast<const f16x8*>(wk + (nt * 2) * 512), bl = *reinterpret_cast<const f16x8*>(wk + (nt * 2 + 1) * 512);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[nt], 0, 0, 0);
      }
    }
    // the words hold (L + c) sa, so acc / (sa sws) = (L + c) Ws already includes the shift term
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 48 * wave + 16 * t + 4 * g + r;
      if (row < FW) {
        const int ib = row / S;
        unsigned h0, h1, l0, l1;
        split_pair(fmaf(acc[0][r], inv_t, add[0]) * sT, fmaf(acc[1][r], inv_t, add[1]) * sT, one, h0, l0);
        split_pair(fmaf(acc[2][r], inv_t, add[2]) * sT, fmaf(acc[3][r], inv_t, add[3]) * sT, one, h1, l1);
        const int off = ib * TT_STRIDE + (row - ib * S) * O1 + 4 * lrow;
        *reinterpret_cast<uint2*>(Th + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(Tlo + off) = make_uint2(l0, l1);
      }
    }
  }
  // first batch of W2 fragments for TT and this thread's column of W2s for AA: in flight across the barrier
  constexpr int TT_BATCH = 10;                                      // k-steps per batch of weight-fragment loads (80 registers)
  const _Float16* wk2 = w2p + ((size_t)wave * 2) * 512 + lane * 8;   // [ks][nt(8)][hl][lane][8], this wave's n-tile
  f16x8 bfr[TT_BATCH][2];
#pragma unroll
  for (int u = 0; u < TT_BATCH; ++u) {
    bfr[u][0] = *reinterpret_cast<const f16x8*>(wk2 + (size_t)u * (8 * 2 * 512));
    bfr[u][1] = *reinterpret_cast<const f16x8*>(wk2 + (size_t)u * (8 * 2 * 512) + 512);
  }
  __syncthreads();
  float* lp = BUILD ? crow + DC_TT : lin + (size_t)pair * LIN_ELEMS;
  // TT = T[24 x 960] W2[960 x 128] on the fp16 matrix cores (3-term split against W2p); wave = n-tile, both m-tiles
  {
    const int r1 = (16 + lrow < G) ? 16 + lrow : G - 1;
    const _Float16* ah0 = Th + lrow * TT_STRIDE + 8 * g;
    const _Float16* al0 = Tlo + lrow * TT_STRIDE + 8 * g;
    const _Float16* ah1 = Th + r1 * TT_STRIDE + 8 * g;
    const _Float16* al1 = Tlo + r1 * TT_STRIDE + 8 * g;
    f32x4 acc[2][3] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
    for (int kb = 0; kb < K2 / 32; kb += TT_BATCH) {
#pragma unroll
      for (int u = 0; u < TT_BATCH; ++u) {
        const int ks = kb + u;
        const f16x8 bh = bfr[u][0], bl = bfr[u][1];
        const f16x8 a0h = *reinterpret_cast<const f16x8*>(ah0 + 32 * ks), a0l = *reinterpret_cast<const f16x8*>(al0 + 32 * ks);
        const f16x8 a1h = *reinterpret_cast<const f16x8*>(ah1 + 32 * ks), a1l = *reinterpret_cast<const f16x8*>(al1 + 32 * ks);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, bh, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, bh, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0l, bh, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l, bh, acc[1][1], 0, 0, 0);
        acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, bl, acc[0][2], 0, 0, 0);
        acc[1][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, bl, acc[1][2], 0, 0, 0);
        if (ks + TT_BATCH < K2 / 32) {   // refill the slot for the next batch
          bfr[u][0] = *reinterpret_cast<const f16x8*>(wk2 + (size_t)(ks + TT_BATCH) * (8 * 2 * 512));
          bfr[u][1] = *reinterpret_cast<const f16x8*>(wk2 + (size_t)(ks + TT_BATCH) * (8 * 2 * 512) + 512);
        }
      }
    }
    const float bv = b2[16 * wave + lrow];
    const float inv_tt = 1.0f / (sT * sw2);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 v = (acc[mt][0] + acc[mt][1]) + acc[mt][2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ib = 16 * mt + 4 * g + r;
        if (ib < G) lp[ib * O2 + 16 * wave + lrow] = fmaf(v[r], inv_tt, bv);
      }
    }
  }
  // AA[jb][p] = sum_o A2[jb][o] W2s[o][p]: thread = (p, 6 consecutive jb); its column of W2s in registers (64 loads in flight)
  if (!BUILD) {
    const int p = tid & (O2 - 1), jb0 = 6 * (tid >> 7);
    float wcol[O1];
#pragma unroll
    for (int o = 0; o < O1; ++o) wcol[o] = w2sum[o * O2 + p];
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < O1; ++o)
#pragma unroll
      for (int j = 0; j < 6; ++j) s[j] = fmaf(A2l[(jb0 + j) * O1 + o], wcol[o], s[j]);
#pragma unroll
    for (int j = 0; j < 6; ++j) lp[G * O2 + (jb0 + j) * O2 + p] = s[j];
  }
}

// The query's side of a cached 1-vs-N sweep, once per call: workgroup v packs the query's words at scale sa_q / 2^v (the version a
// candidate whose own scale is that much coarser pairs with); workgroup 0 also leaves AA = (R W1) W2s and {max R, min R}.
// Same arithmetic, in the same order, as the R / AA parts of delta_prepare_split_kernel with shift c = 0.
__global__ __launch_bounds__(512) void delta_query_kernel(const float* __restrict__ feats_r, const float* __restrict__ a2raw,
                                                          const float* __restrict__ w2sum, unsigned* __restrict__ qblock,
                                                          unsigned* __restrict__ live_out, const _Float16* __restrict__ w1p,
                                                          _Float16* __restrict__ w1c) {
  __shared__ float red[2 * NWAVE];
  __shared__ float A2l[G * O1];
  __shared__ int alive2[NPAIR][FC];
  __shared__ int scr[2 * FC + 32];
  __shared__ __attribute__((aligned(16))) unsigned char chan_q[CHAN_BYTES];
  const int ver = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f32x4* R4 = reinterpret_cast<const f32x4*>(feats_r);
  constexpr int R_ITEMS = OVN_FEAT_ELEMS / 8;
  f32x4 rv[12][2];
  float mx = -3.0e38f, mn = 3.0e38f;
  for (int i = tid; i < NPAIR * FC; i += 512) (&alive2[0][0])[i] = 0;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int i8 = tid + 512 * k;
    if (i8 < R_ITEMS) {
      rv[k][0] = R4[2 * i8];
      rv[k][1] = R4[2 * i8 + 1];
    } else {
      rv[k][0] = rv[k][1] = R4[0];
    }
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      mx = fmaxf(mx, fmaxf(fmaxf(rv[k][h][0], rv[k][h][1]), fmaxf(rv[k][h][2], rv[k][h][3])));
      mn = fminf(mn, fminf(fminf(rv[k][h][0], rv[k][h][1]), fminf(rv[k][h][2], rv[k][h][3])));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mx = fmaxf(mx, __shfl_down(mx, off, 64));
    mn = fminf(mn, __shfl_down(mn, off, 64));
  }
  if (lane == 0) {
    red[wave] = mx;
    red[NWAVE + wave] = mn;
  }
  __syncthreads();   // (also: alive2[][] cleared)
  // item i8 = (row i8 / 16, channels 8 (i8 % 16) ..): flag the channels that are non-zero in the row's column-group pair
  if (live_out != nullptr) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int i8 = tid + 512 * k;
      if (i8 < R_ITEMS) {
        int* arow = &alive2[(i8 >> 4) / (2 * S)][8 * (i8 & 15)];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (rv[k][0][e] != 0.0f) arow[e] = 1;
          if (rv[k][1][e] != 0.0f) arow[4 + e] = 1;
        }
      }
    }
  }
  mx = red[0];
  mn = red[NWAVE];
#pragma unroll
  for (int w = 1; w < NWAVE; ++w) {
    mx = fmaxf(mx, red[w]);
    mn = fminf(mn, red[NWAVE + w]);
  }
  __syncthreads();
  const float sa = ldexpf(ovn_pow2_scale_for(mx), -(ver < QV ? ver : 0));
  unsigned* Pr = qblock + (size_t)(ver < QV ? ver : 0) * OVN_FEAT_ELEMS;
  // The query's K walk: live channels only (every workgroup builds the same list; workgroup 0 leaves it in `live_out` for the
  // prepare and contraction kernels), plain order when the sweep does not compact (live_out NULL) or the volume has a negative value
  int ns = 4;
  if (live_out != nullptr) {
    const int nlive = build_chan_list(alive2, mn < 0.0f, chan_q, scr, tid);
    ns = 1;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) ns = chan_q[FC + p] > ns ? chan_q[FC + p] : ns;   // the largest walk: what is packed and gathered
    if (ver == 0) {
      if (tid == 0) {
        live_out[0] = (unsigned)ns;
        live_out[1] = (unsigned)nlive;
        live_out[2] = live_out[3] = 0u;
      }
      if (tid < FC / 4 + 4) live_out[4 + tid] = reinterpret_cast<const unsigned*>(chan_q)[tid];
    }
    // this workgroup's share of the W1 fragments gathered for the list (workgroups QV .. QV + QGW - 1 do nothing else)
    const int total8 = ns * S * 4 * 2 * 64;
    for (int idx8 = ver * 512 + tid; idx8 < total8; idx8 += gridDim.x * 512) w1c_gather8(w1p, chan_q, idx8, w1c);
    if (ver >= QV && ver != (int)gridDim.x - 1) return;
  } else {
    (void)load_chan_table(nullptr, false, chan_q, tid);
    __syncthreads();
  }
  if (ver < QV) {
    // two phases: ALL 96 channel-gathered loads of a thread first (addresses clamped for the items it will not store: the loads stay
    // unconditional and in flight together), then packing and stores -- as one load / use loop unrolled by 2 this paid six L2 round
    // trips in a row, a third of the kernel's 39 us in front of a single-pair sweep
    f32x4 gv[12][2];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int i8 = tid + 512 * k;
      const int i8c = i8 < R_ITEMS ? i8 : R_ITEMS - 1;
      const int jrow = i8c >> 4, sc = (i8c >> 2) & 3, gq = i8c & 3;
      const float* rrow = feats_r + (size_t)jrow * FC;
      const unsigned char* ch = chan_q + 32 * (sc < ns ? sc : 0) + 8 * gq;
      gv[k][0] = (f32x4){rrow[ch[0]], rrow[ch[1]], rrow[ch[2]], rrow[ch[3]]};
      gv[k][1] = (f32x4){rrow[ch[4]], rrow[ch[5]], rrow[ch[6]], rrow[ch[7]]};
    }
    const TailPack tp = tail_of(chan_q);
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int i8 = tid + 512 * k;
      if (i8 < R_ITEMS) {
        const int jrow = i8 >> 4, sc = (i8 >> 2) & 3, gq = i8 & 3;
        if (sc < ns && sc != tp.sl) {
          const int jb = jrow / S, dj = jrow - jb * S;
          unsigned* dst = Pr + ((jb * 4 + sc) * S + dj) * 32 + gq * 8;
          *reinterpret_cast<u32x4*>(dst) = pack4(gv[k][0], sa, 0.0f);
          *reinterpret_cast<u32x4*>(dst + 4) = pack4(gv[k][1], sa, 0.0f);
        }
      }
    }
    // the packed last slice (TailPack): step u of column group jb holds the taps u T .. u T + T - 1 of its n channels
    if (tp.sl >= 0) pack_tail_slice(feats_r, chan_q, tp, Pr, sa, 0.0f, tid);
  }
  // AA and {max, min} are left by the LAST workgroup of the launch (a gather helper when the sweep compacts): workgroup 0's packing and
  // this dot product run side by side instead of one after the other
  if (ver != (int)gridDim.x - 1) return;
  float* aa = reinterpret_cast<float*>(qblock + (size_t)QV * OVN_FEAT_ELEMS);
  if (tid == 0) *reinterpret_cast<f32x4*>(aa + G * O2) = (f32x4){mx, mn, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int i = tid + 512 * u;
    float v = a2raw[i];
#pragma unroll
    for (int k = 1; k < A2_KSPLIT; ++k) v += a2raw[(size_t)k * A2_ELEMS + i];
    A2l[i] = v;   // (+ c * w1col with c = 0)
  }
  __syncthreads();
  {
    const int p = tid & (O2 - 1), jb0 = 6 * (tid >> 7);
    float wcol[O1];
#pragma unroll
    for (int o = 0; o < O1; ++o) wcol[o] = w2sum[o * O2 + p];
    float sacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < O1; ++o)
#pragma unroll
      for (int j = 0; j < 6; ++j) sacc[j] = fmaf(A2l[(jb0 + j) * O1 + o], wcol[o], sacc[j]);
#pragma unroll
    for (int j = 0; j < 6; ++j) aa[(jb0 + j) * O2 + p] = sacc[j];
  }
}

// W2s[o][p] = sum_di W2[di][o][p] (fp64 accumulation, rounded once)
__global__ __launch_bounds__(256) void delta_w2sum_kernel(const float* __restrict__ w2, float* __restrict__ w2sum) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < O1 * O2) {
    double s = 0.0;
    for (int di = 0; di < S; ++di) s += (double)w2[(size_t)di * O1 * O2 + idx];
    w2sum[idx] = (float)s;
  }
}

// c_conv1's min-term contraction, M[i][jb][o] = sum_{dj,c} min(l'[i][c], r'[15 jb + dj][c]) W1[dj][c][o].  One workgroup of 8 waves = one
// pair (or 1/nsplit of its passes).  A pair is 360 x 24 (row i, column group jb) combinations = 540 MFMA row tiles exactly, but 360
// rows are 22.5 tiles: the kernel of rounds 2-5 (wave = 48 rows x two column groups per pass, twelve passes; in the history up to
// the round-5 tree) padded every column group to 24 tiles -- 6.7 % of its MFMAs, in the busiest SIMD of every pass.  Here a pass is
// RT = 2 row tiles (32 rows) x ALL 24 column groups: wave w holds column groups 3 w .. 3 w + 2 of both row tiles (96 accumulator
// registers), eleven such passes cover rows 0 .. 351 without a padded slot, and ONE short pass takes the last 8 rows of all 24
// column groups as 12 tiles of (8 rows x 2 column groups), 3 per SIMD: 11.25 pass-times instead of 12 (SQ_INSTS_MFMA per 1024 pairs
// at 3 slices: 3.185e8 -> 2.986e8; same-box A/B 3.05-3.09 -> 2.92 ms, profiles/r6_c1_transposed_ab.txt).  With it:
//   * L (the candidate's words, pair-specific, from HBM): a pass needs 32 rows of every walked channel -- 4 KB per slice, SHARED by the
//     eight waves, fetched ONCE per pair (184 KB) instead of once per pass (12 x 184 KB); nothing past row 359 is ever read;
//   * R (the query's words, shared by every pair of the sweep: L2-resident): all 24 column groups, streamed chunk by chunk beside the
//     W1 window (9 KB per chunk of 3 steps) instead of two column groups resident per pass;
//   * the slice counts of the dead-channel compaction: a pass walks the LARGEST count of the query; a slice beyond a column group's own
//     count adds exact zeros to it (its channels there are dead in that group's query columns), so a wave skips a slice's MFMAs when
//     none of its slots walks it -- one wave-uniform test per chunk.
// LDS (76 KB): W1 window 2 x 24 KB | R chunk 2 x 9 KB | L slice 2 x 4.25 KB, all filled by LDS-DMA one chunk / one slice ahead; the
// chunk barrier's vmcnt(0) finds them landed.  Per-accumulator order as in rounds 2-5 (slices cyclically from the slot's rotation,
// taps 0 .. 14, hi hi / lo hi / hi lo): the results have the bits of the round-5 kernel wherever every column group walks the same
// slices.  Output: o1raw = -2 M s1r in the streaming order of the c_conv2 kernel (tile of 192 (jb, ib) rows, k-step major; K order of
// W2p: o' = 4 (o & 15) + (o >> 4)), 16 bytes per lane and accumulator row.  RT = 1 (16 rows x 24 column groups, 22 + 1 passes = 23
// workgroups per pair) and RT = 0 (every pass a short one: 8 rows x 24 column groups, 45 workgroups per pair with 3 tiles per SIMD
// each) serve a handful of pairs (the single-pair latency of demo2 / gated demo3 queries): same order, same bits.
constexpr int T_LBLK = 8 * 32 + 16;                       // words of one L block: [8 positions][32 rows] + 16 of padding (bank spread)
constexpr int T_LSL_WORDS = 4 * T_LBLK;                   // one L slice (32 positions x 32 rows): 4,352 B
constexpr int T_TAIL_ROW0 = (FW / 16) * 16;               // 352: first row of the short pass
static_assert(FW - T_TAIL_ROW0 == 8 && G % 3 == 0 && G / 3 == NWAVE, "pass geometry");
// SPC = MFMA steps (taps) per chunk: W1 fragments SPC x 8 KB + R words [jb(24)][tap(SPC)][g(4)][8] = SPC x 3 KB per chunk
constexpr size_t t_lds_bytes(int spc) { return 2 * (size_t)spc * STEP_BYTES + 2 * (size_t)G * spc * 32 * 4 + 2 * (size_t)T_LSL_WORDS * 4; }

template <int RT, int T_SPC>
__global__ __launch_bounds__(512) void delta_c1_f16x3_kernel(const DeltaDesc* __restrict__ desc, const _Float16* __restrict__ w1p,
                                                             const f32x4* __restrict__ scales, float* __restrict__ o1raw, int rot,
                                                             int nsplit, int pair0, const int32_t* __restrict__ lidx,
                                                             const unsigned* __restrict__ live, const _Float16* __restrict__ w1c) {
  constexpr int NFULL = RT ? (FW / 16) / RT : 0; // full passes: 11 (RT 2) / 22 (RT 1); RT 0: none -- EVERY pass is a short one
  constexpr int NPASS = RT ? NFULL + 1 : FW / 8; // + the short pass over rows 352 .. 359 (RT 0: 45 passes of 8 rows)
  constexpr int T_CHB = T_SPC * STEP_BYTES;      // W1 fragments of a chunk
  constexpr int T_CPS = S / T_SPC;               // chunks per channel slice
  constexpr int T_RCH_WORDS = G * T_SPC * 32;    // R words of a chunk
  constexpr int PFN = T_CHB / (512 * 16);        // W1 DMA instructions per lane and chunk
  constexpr int RPJ = T_SPC * 8;                 // 16-byte pieces of an R chunk per column group
  constexpr int RFN = (G * RPJ + 511) / 512;     // R DMA instructions per lane and chunk (the last one partial, whole waves)
  static_assert(S % T_SPC == 0 && T_CHB % (512 * 16) == 0 && T_SPC == 3 && (G * RPJ) % 64 == 0, "bad chunking (the packed slice has 3, 6 or 9 steps)");
  __shared__ __attribute__((aligned(16))) unsigned char chan_s[CHAN_BYTES];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* wst = smem_raw;                                                   // [2][T_CHB]
  unsigned* rbuf = reinterpret_cast<unsigned*>(smem_raw + 2 * T_CHB);              // [2][T_RCH_WORDS]
  unsigned* lbuf = rbuf + 2 * T_RCH_WORDS;                                         // [2][T_LSL_WORDS]

  const int pair = blockIdx.x / nsplit;
  const int part = blockIdx.x - pair * nsplit;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15;
  const int g = lane >> 4;

  const unsigned* L = desc[pair].pl;     // channel-major [c][360]: per-pair scratch or the candidate's cache row
  const unsigned* Rw = desc[pair].pr;    // [jb][slice][tap][g][8]: per-pair scratch or the query's shared words
  const float krow = scales[2 * pair][1];
  const bool compact = live != nullptr && scales[2 * pair + 1][3] != 0.0f;   // workgroup-uniform
  (void)load_chan_table(live, compact, chan_s, tid);
  const unsigned char* w1bytes = reinterpret_cast<const unsigned char*>(compact ? w1c : w1p);
  __syncthreads();   // channel table

  // slices walked: by the workgroup (the largest count of any column-group pair) and by this wave's slots
  int nsm = 1;
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) nsm = max(nsm, (int)chan_s[FC + p]);
  nsm = __builtin_amdgcn_readfirstlane(nsm);
  int ns_full[3], ns_tail[2];
#pragma unroll
  for (int j = 0; j < 3; ++j) ns_full[j] = __builtin_amdgcn_readfirstlane((int)chan_s[FC + ((3 * wave + j) >> 1)]);
  // short pass: waves 0 .. 3 take two tiles (column groups 4 w .. 4 w + 3), waves 4 .. 7 one (16 + 2 (w - 4), + 1): three per SIMD
  // chunks of a slice: 5, or fewer for the packed last slice of a compacted walk (TailPack: 3, 6 or 9 steps)
  const int pk_sl = __builtin_amdgcn_readfirstlane(chan_s[TB_SLICE] == 255 ? -1 : (int)chan_s[TB_SLICE]);
  const int pk_cps = __builtin_amdgcn_readfirstlane((int)chan_s[TB_STEPS] / T_SPC);
  const int tail_jb0 = __builtin_amdgcn_readfirstlane(wave < 4 ? 4 * wave : 16 + 2 * (wave - 4));
  const int tail_slots = __builtin_amdgcn_readfirstlane(wave < 4 ? 2 : 1);
#pragma unroll
  for (int j = 0; j < 2; ++j) ns_tail[j] = __builtin_amdgcn_readfirstlane((int)chan_s[FC + ((tail_jb0 + 2 * j) >> 1)]);

  // rotation of the K walk by the CANDIDATE's slot in the left pool (see the header of ovn_heads): a pair's bits depend neither on the
  // chunking of the sweep nor on the order of an index list, and a 32-aligned shard reproduces the unsharded sweep
  const int slot = lidx ? lidx[pair] : pair0 + pair;
  const int s0 = (rot ? ((slot >> 3) & 3) : 0) % nsm;
  const int p_begin = part * NPASS / nsplit, p_end = (part + 1) * NPASS / nsplit;

#define OVN_DMA_W(CH, BUF)                                                                        \
  _Pragma("unroll") for (int q = 0; q < PFN; ++q)                                                 \
      glds16(w1bytes + (size_t)(CH) * T_CHB + (q * 512 + tid) * 16, wst + (BUF) * T_CHB + (q * 512 + wave * 64) * 16);
  // piece idx = q 512 + tid of an R chunk: column group idx / RPJ, 16 bytes idx % RPJ of its T_SPC taps (contiguous in the packed volume)
#define OVN_DMA_R(SL, C5, BUF)                                                                    \
  _Pragma("unroll") for (int q = 0; q < RFN; ++q) {                                               \
    const int idx = q * 512 + tid;                                                                \
    const int jb_ = idx / RPJ, pc_ = idx - RPJ * jb_;                                             \
    if (q * 512 + wave * 64 < G * RPJ)                                                            \
      glds16(Rw + jb_ * K1 + ((SL) * S + T_SPC * (C5)) * 32 + pc_ * 4, rbuf + (BUF) * T_RCH_WORDS + (q * 512 + wave * 64) * 4); \
  }
  // L slice SL of the pass that starts at row ROW0 and holds ROWS rows: block q (wave q < 4) = positions 8 q .. 8 q + 7, a lane moves
  // rows 4 (lane & 7) .. + 3 of position 8 q + (lane >> 3): 16 contiguous bytes of the channel-major volume, nothing past row 359
#define OVN_DMA_L(SL, ROW0, ROWS, BUF)                                                            \
  if (wave < 4 && 4 * (lane & 7) < (ROWS))                                                        \
    glds16(L + (size_t)chan_s[(SL) * 32 + 8 * wave + (lane >> 3)] * FW + (ROW0) + 4 * (lane & 7),  \
           lbuf + (BUF) * T_LSL_WORDS + wave * T_LBLK);

  int cur = 0, lcur = 0;
  {
    const bool tail0 = RT == 0 || p_begin == NFULL;
    OVN_DMA_W(T_CPS * s0, 0)
    OVN_DMA_R(s0, 0, 0)
    OVN_DMA_L(s0, tail0 ? (RT ? T_TAIL_ROW0 : 8 * p_begin) : 16 * RT * p_begin, tail0 ? 8 : 16 * RT, 0)
  }
  __syncthreads();

  // One pass.  TAIL = false: NT = RT row tiles x NJ = 3 column groups per wave; TAIL = true: one tile row (8 rows, both halves of the
  // 16 lanes-rows: column groups jb, jb + 1) x NJ = 2 slots.
  auto run_pass = [&](auto slots_tag, int pass) {   // tag: 0 = a full pass, 1 / 2 = a short pass of this wave's one / two tiles
    constexpr bool TAIL = decltype(slots_tag)::value != 0;
    constexpr int NT = TAIL ? 1 : (RT ? RT : 1);
    constexpr int NJ = TAIL ? decltype(slots_tag)::value : 3;
    const bool has_next = pass + 1 < p_end;
    const bool next_tail = RT == 0 || pass + 1 == NFULL;
    const int row0_n = next_tail ? (RT ? T_TAIL_ROW0 : 8 * (pass + 1)) : 16 * RT * (pass + 1);
    const int rows_n = next_tail ? 8 : 16 * RT;
    const int row0 = TAIL ? (RT ? T_TAIL_ROW0 : 8 * pass) : 16 * RT * pass;
    f32x4 acc[NJ][NT][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // word offset of this lane's R words inside a chunk, per slot: column group x 96 + 8 g (TAIL: the lane's half picks the group)
    int rofs[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rofs[j] = (TAIL ? tail_jb0 + 2 * j + (lrow >> 3) : 3 * wave + j) * (T_SPC * 32) + 8 * g;
    u32x4 la[NT][2];
#pragma unroll 1
    for (int q4 = 0; q4 < nsm; ++q4) {
      const int sl = (s0 + q4 >= nsm) ? s0 + q4 - nsm : s0 + q4;
      const bool last_slice = q4 + 1 == nsm;
      const int sl_n = last_slice ? s0 : ((sl + 1 == nsm) ? 0 : sl + 1);
      {
        const unsigned* lw = lbuf + lcur * T_LSL_WORDS + g * T_LBLK + (TAIL ? (lrow & 7) : lrow);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            la[t][0][e] = lw[e * 32 + 16 * t];
            la[t][1][e] = lw[(4 + e) * 32 + 16 * t];
          }
      }
      bool act[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) act[j] = TAIL ? (sl < ns_tail[j]) : (sl < ns_full[j]);
      bool any = false;
#pragma unroll
      for (int j = 0; j < NJ; ++j) any = any || act[j];
      // A slice beyond a column group's own walk adds exact zeros to it (its channels there are dead in the query's columns of that
      // group), so skipping is only ever an optimisation: a wave skips the MFMAs of a slice when NONE of its slots walks it -- one
      // wave-uniform test per chunk; a test per slot would end the scheduling region at every slot
      const int cps = (sl == pk_sl) ? pk_cps : T_CPS;
#pragma unroll 1
      for (int c5 = 0; c5 < cps; ++c5) {
        {   // the next chunk of the walk (the first one of the next pass after the last): W1 fragments + R words, one chunk ahead
          const int sl_x = (c5 + 1 < cps) ? sl : sl_n, c5_x = (c5 + 1 < cps) ? c5 + 1 : 0;
          OVN_DMA_W(T_CPS * sl_x + c5_x, cur ^ 1)
          OVN_DMA_R(sl_x, c5_x, cur ^ 1)
        }
        if (c5 == 0) {   // the next L slice (into the other buffer): of this pass, or the first of the next pass
          if (!last_slice) {
            OVN_DMA_L(sl_n, row0, TAIL ? 8 : 16 * RT, lcur ^ 1)
          } else if (has_next) {
            OVN_DMA_L(sl_n, row0_n, rows_n, lcur ^ 1)
          }
        }
        if (any) {
          const unsigned char* wcur = wst + cur * T_CHB;
          const unsigned* rcur = rbuf + cur * T_RCH_WORDS;
          u32x4 rw[NJ][2];
          f16x8 bh[2][4], bl[2][4];
          f16x8 ah[2], al[2];
#define OVN_READ_B(SET, H)                                                                          \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                \
    bh[SET][nt] = *reinterpret_cast<const f16x8*>(wcur + (H) * STEP_BYTES + ((nt * 2 + 0) * 64 + lane) * 16); \
    bl[SET][nt] = *reinterpret_cast<const f16x8*>(wcur + (H) * STEP_BYTES + ((nt * 2 + 1) * 64 + lane) * 16); \
  }
#define OVN_READ_R(J, H)                                                                            \
  {                                                                                                 \
    rw[J][0] = *reinterpret_cast<const u32x4*>(rcur + rofs[J] + (H) * 32);                           \
    rw[J][1] = *reinterpret_cast<const u32x4*>(rcur + rofs[J] + (H) * 32 + 4);                       \
  }
          // The chunk as a chain of slot regions (q = step x slot, slot = (row tile, column group)), fenced from each other: region q
          // forms the operands of slot q + 1 (16 VALU) next to the 12 MFMAs of slot q, requests a column group's R words of the next
          // step once its last tile of this step has taken them, and the next step's W1 fragments at the first slot of a step.
          // Unfenced, the scheduler hoists every operand formation and fragment read of a step to its top: 344 spilled registers.
          constexpr int NS = NT * NJ;
          OVN_READ_B(0, 0)
#pragma unroll
          for (int j = 0; j < NJ; ++j) OVN_READ_R(j, 0)
          make_a(la[0][0], la[0][1], rw[0][0], rw[0][1], ah[0], al[0]);
#pragma unroll
          for (int q = 0; q < T_SPC * NS; ++q) {
            const int h = q / NS, k = q - NS * h, t = k / NJ, j = k - NJ * t;
            __builtin_amdgcn_sched_barrier(0);
            if (k == 0 && h + 1 < T_SPC) OVN_READ_B((h + 1) & 1, h + 1)
            if (t == NT - 1 && h + 1 < T_SPC) OVN_READ_R(j, h + 1)
            if (q + 1 < T_SPC * NS) {
              const int k1 = (q + 1) % NS, t1 = k1 / NJ, j1 = k1 - NJ * t1;
              make_a(la[t1][0], la[t1][1], rw[j1][0], rw[j1][1], ah[(q + 1) & 1], al[(q + 1) & 1]);
            }
            {
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q & 1], bh[h & 1][nt], acc[j][t][nt], 0, 0, 0);
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q & 1], bh[h & 1][nt], acc[j][t][nt], 0, 0, 0);
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) acc[j][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q & 1], bl[h & 1][nt], acc[j][t][nt], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#undef OVN_READ_B
#undef OVN_READ_R
        }
        __syncthreads();   // vmcnt(0): the DMAs issued above have landed; every wave is done with buffers `cur`
        cur ^= 1;
      }
      lcur ^= 1;
    }
    // -2 M s1r -> o1raw in the streaming order of delta_c2_f16x3_kernel: [tile of 192 rows = (pair, jb / 8)][k-step 2 di + (o' >> 5)]
    // [row = (jb % 8) 24 + ib][o' & 31], o' = 4 lrow + nt: 16 bytes per lane.  The stores drain behind the next pass's first chunk.
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lr = 4 * g + r;
          const int jb = TAIL ? tail_jb0 + 2 * j + (lr >> 3) : 3 * wave + j;
          const int i = TAIL ? row0 + (lr & 7) : row0 + 16 * t + lr;
          const int ib = i / S, di = i - ib * S;
          float* dst = o1raw + ((size_t)pair * 3 + (jb >> 3)) * (C2_TILE_ROWS * K2) + (size_t)(2 * di + (lrow >> 3)) * (C2_TILE_ROWS * 32) +
                       ((jb & 7) * G + ib) * 32 + 4 * (lrow & 7);
          const f32x4 v = {acc[j][t][0][r] * krow, acc[j][t][1][r] * krow, acc[j][t][2][r] * krow, acc[j][t][3][r] * krow};
          // streaming stores: the 2.2 MB per pair pass through L2 without displacing the W1 / R lines
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
        }
    }
  };

  // (the short pass as two instantiations, picked per wave: a slot test inside the chunk would end its scheduling regions, an idle
  //  second slot in waves 4 .. 7 would be 4 more tiles per pair and the busiest SIMD at 4 instead of 3; every wave meets the same barriers)
  for (int pass = p_begin; pass < p_end; ++pass) {
    bool full = false;
    if constexpr (RT != 0) full = pass < NFULL;
    if constexpr (RT != 0) {
      if (full) run_pass(std::integral_constant<int, 0>{}, pass);
    }
    if (!full) {
      if (tail_slots == 2) run_pass(std::integral_constant<int, 2>{}, pass);
      else run_pass(std::integral_constant<int, 1>{}, pass);
    }
  }
#undef OVN_DMA_L
#undef OVN_DMA_W
#undef OVN_DMA_R
}

// c_conv2 on the stored -2 M rows: a (n 576) x 960 x 128 GEMM, HBM-bound (3840 B read per row, 0.2 ms of MFMA per 1024 pairs).
// Workgroup = 64 MT consecutive rows (row = (pair 24 + jb) 24 + ib; a tile may straddle two pairs), 4 waves x 16 MT rows x all 128
// columns.  The kernel is bound by load latency x bytes in flight: the A rows are read straight into registers TWO k-steps ahead
// (each element is used by one wave only) and split to fp16 hi/lo there; W2's k-step slabs (16 KB, shared by the 4 waves) go
// through registers into a double-buffered LDS window, loaded BEFORE the A rows of the same step so that waiting for them
// (in-order return) never waits for the younger A loads.  Plain loads only: next to an LDS-DMA the compiler drains vmcnt(0)
// at every barrier, which cuts the prefetch distance to one step (measured 0.60 ms; one step = the loaded HBM latency, ~3 us).
// Epilogue: acc / (s1r sw2) + TT[ib] + AA[jb] + b2, ReLU, per-pair maximum.
// Template: MT m-tiles per wave (workgroup = 64 MT rows), A rows ASLOTS - 1 k-steps ahead, W2 slabs WSETS k-steps ahead in registers.
// <3, 3, 1> is the sweep kernel described above (its latency cover is other workgroups: two per CU, 768+ per launch).
// <1, 6, 3> serves a handful of pairs (9 workgroups per pair, acc 32 registers): with nothing else on the CU, a step costs
// the memory round trip of the operands it waits for, so they travel 5 (A) and 3 (W2) steps ahead -- 44 -> ~17 us for one pair
// (the single-pair latency of demo2 / gated demo3 queries).  Same per-accumulator order (k-step, hi hi / lo hi / hi lo): same bits.
template <int MT, int ASLOTS, int WSETS>
__global__ __launch_bounds__(256, 2) void delta_c2_f16x3_kernel(const float* __restrict__ o1raw, const _Float16* __restrict__ w2p,
                                                                         const DeltaDesc* __restrict__ desc, const f32x4* __restrict__ scales,
                                                                         float* __restrict__ o2, unsigned* __restrict__ o2max, float one) {
  __shared__ __attribute__((aligned(16))) unsigned char wb[2][16384];
  constexpr int NKS = K2 / 32;   // 30
  constexpr int WG_ROWS = 64 * MT;
  static_assert(C2_TILE_ROWS % WG_ROWS == 0 && 6 % ASLOTS == 0 && 6 % WSETS == 0 && NKS % 6 == 0 && ASLOTS >= 2, "bad c_conv2 tiling");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, g = lane >> 4;
  const int wg_row0 = blockIdx.x * WG_ROWS;                      // global row (pair 576 + jb 24 + ib) of the workgroup's first row
  const int row0 = wg_row0 + 16 * MT * wave;                     // ... of this wave's first m-tile
  // o1raw[tile of 192 rows][ks][row in tile][32]: one k-step of the tile is 24 KB contiguous, a wave's m-tile 2 KB of it
  const int tile = wg_row0 / C2_TILE_ROWS, in_tile = wg_row0 - tile * C2_TILE_ROWS;
  const float* abase = o1raw + (size_t)tile * (C2_TILE_ROWS * K2) + (in_tile + 16 * MT * wave + lrow) * 32 + 8 * g;
  const unsigned char* w2bytes = reinterpret_cast<const unsigned char*>(w2p) + tid * 16;

  f32x4 araw[ASLOTS][MT][2];   // [k-step mod ASLOTS][m-tile][half]
  f32x4 wr[WSETS][4];          // [k-step mod WSETS]
#define OVN_LOAD_A(SLOT, KS)                                                                 \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                        \
    araw[SLOT][mt][0] = *reinterpret_cast<const f32x4*>(abase + (KS) * (C2_TILE_ROWS * 32) + mt * 512);       \
    araw[SLOT][mt][1] = *reinterpret_cast<const f32x4*>(abase + (KS) * (C2_TILE_ROWS * 32) + mt * 512 + 4);   \
  }
#define OVN_LOAD_W(SET, KS) \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) wr[SET][q] = *reinterpret_cast<const f32x4*>(w2bytes + (size_t)(KS) * 16384 + q * 4096);
#define OVN_STORE_W(SET, BUF) \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&wb[BUF][q * 4096 + tid * 16]) = wr[SET][q];

  f32x4 acc[MT][8];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: W2 slabs of steps 0 .. WSETS - 1 (each ahead of the A rows requested after it), A rows of steps 0 .. ASLOTS - 2
#pragma unroll
  for (int k = 0; k < WSETS; ++k) OVN_LOAD_W(k, k)
#pragma unroll
  for (int k = 0; k < ASLOTS - 1; ++k) OVN_LOAD_A(k, k)
  OVN_STORE_W(0, 0)
  __syncthreads();

  // J = k-step modulo 6 (compile time): register slots and the LDS buffer follow from it
#define OVN_C2_STEP(J, KS)                                                                                        \
  {                                                                                                               \
    /* set J % WSETS held step KS's slab, stored to LDS at the end of the previous step: free for step KS + WSETS */ \
    if ((KS) + WSETS < NKS) OVN_LOAD_W((J) % WSETS, (KS) + WSETS)                                                 \
    if ((KS) + ASLOTS - 1 < NKS) OVN_LOAD_A(((J) + ASLOTS - 1) % ASLOTS, (KS) + ASLOTS - 1)                       \
    __builtin_amdgcn_sched_barrier(0);   /* the scheduler otherwise sinks the loads next to their first use */      \
    f16x8 ah[MT], al[MT];                                                                                         \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                           \
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;                                                                    \
      split_pair(araw[(J) % ASLOTS][mt][0][0], araw[(J) % ASLOTS][mt][0][1], one, h0, l0);                        \
      split_pair(araw[(J) % ASLOTS][mt][0][2], araw[(J) % ASLOTS][mt][0][3], one, h1, l1);                        \
      split_pair(araw[(J) % ASLOTS][mt][1][0], araw[(J) % ASLOTS][mt][1][1], one, h2, l2);                        \
      split_pair(araw[(J) % ASLOTS][mt][1][2], araw[(J) % ASLOTS][mt][1][3], one, h3, l3);                        \
      ah[mt] = __builtin_bit_cast(f16x8, (u32x4){h0, h1, h2, h3});                                                \
      al[mt] = __builtin_bit_cast(f16x8, (u32x4){l0, l1, l2, l3});                                                \
    }                                                                                                             \
    const unsigned char* wcur = &wb[(J) & 1][lane * 16];                                                          \
    _Pragma("unroll") for (int nt = 0; nt < 8; ++nt) {                                                            \
      const f16x8 bh = *reinterpret_cast<const f16x8*>(wcur + (nt * 2) * 1024);                                   \
      const f16x8 bl = *reinterpret_cast<const f16x8*>(wcur + (nt * 2 + 1) * 1024);                               \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, acc[mt][nt], 0, 0, 0);                 \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh, acc[mt][nt], 0, 0, 0);                 \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, acc[mt][nt], 0, 0, 0);                 \
    }                                                                                                             \
    if ((KS) + 1 < NKS) OVN_STORE_W(((J) + 1) % WSETS, ((J) + 1) & 1)                                             \
    __syncthreads();                                                                                              \
  }
#pragma unroll 1
  for (int ks = 0; ks < NKS; ks += 6) {
    OVN_C2_STEP(0, ks)
    OVN_C2_STEP(1, ks + 1)
    OVN_C2_STEP(2, ks + 2)
    OVN_C2_STEP(3, ks + 3)
    OVN_C2_STEP(4, ks + 4)
    OVN_C2_STEP(5, ks + 5)
  }
#undef OVN_C2_STEP
#undef OVN_LOAD_A
#undef OVN_LOAD_W
#undef OVN_STORE_W

  // a workgroup's 192 rows belong to ONE pair (576 = 3 x 192): its scale and the pointers to its linear terms are wave-uniform
  // (scalar loads, issued here and long landed when the epilogue needs them)
  static_assert((G * G) % C2_TILE_ROWS == 0, "a c_conv2 workgroup must not straddle two pairs");
  const int pair = blockIdx.x / (G * G / WG_ROWS);
  const float inv2 = scales[2 * pair][3];
  // (pointers loaded from memory are generic to the compiler: as flat loads they would force vmcnt(0) lgkmcnt(0) waits into the
  // K loop's prefetch chain -- 0.60 -> 0.67 ms; say that they are global)
  typedef const __attribute__((address_space(1))) float* gfloat_p;
  const gfloat_p tt = (gfloat_p)desc[pair].tt;
  const gfloat_p aa = (gfloat_p)desc[pair].aa;
  // The linear terms of output row (mt, r) are loaded one row AHEAD of the row being stored: tt / aa are not `restrict` to the
  // compiler (they come out of a descriptor), so it keeps every load behind the o2 stores that precede it in program order -- written
  // naively that is one exposed L2 round trip per element (96 per lane: 0.60 -> 0.66 ms for the kernel)
  float vmax = 0.f;
  float lv[2][8];
  auto load_lin = [&](int q, float* dst8) {   // q = 4 mt + r
    const int rr = row0 + 16 * (q >> 2) - pair * (G * G) + 4 * g + (q & 3);
    const int jb = rr / G, ib = rr - jb * G;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) dst8[nt] = tt[ib * O2 + 16 * nt + lrow] + aa[jb * O2 + 16 * nt + lrow];
  };
  load_lin(0, lv[0]);
#pragma unroll
  for (int q = 0; q < 4 * MT; ++q) {
    if (q + 1 < 4 * MT) load_lin(q + 1, lv[(q + 1) & 1]);
    const int mt = q >> 2, r = q & 3;
    const int rr = row0 + 16 * mt - pair * (G * G) + 4 * g + r;
    const int jb = rr / G, ib = rr - jb * G;
    float* dst = o2 + (((size_t)pair * G + ib) * G + jb) * O2 + lrow;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const float v = fmaxf(fmaf(acc[mt][nt][r], inv2, lv[q & 1][nt]), 0.0f);
      dst[16 * nt] = v;
      vmax = fmaxf(vmax, v);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
  if (lane == 0 && vmax > 0.f) atomicMax(o2max + pair, __float_as_uint(vmax));
}

}  // namespace

size_t ovn_delta_f16x3_scratch_bytes(int n, bool per_pair_right) {
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  return al((size_t)n * 8 * sizeof(float)) + al((size_t)n * sizeof(unsigned)) + 2 * al((size_t)n * OVN_FEAT_ELEMS * sizeof(unsigned)) +
         al((size_t)n * LIN_ELEMS * sizeof(float)) + al((size_t)(per_pair_right ? n : 1) * A2_KSPLIT * A2_ELEMS * sizeof(float)) +
         al((size_t)n * O1RAW_ELEMS * sizeof(float)) + al((size_t)n * sizeof(DeltaDesc)) + al(QBLOCK_WORDS * sizeof(unsigned)) +
         al(LIVE_WORDS * sizeof(unsigned)) + al((size_t)S * FC * O1 * 2 * sizeof(_Float16));
}

static int pick_nsplit(int n) {
  // divisors of the 12 passes (11 of two row tiles + the short one): time ~ rounds of workgroups over the 256 CUs x 1/d of a pair's
  // work; the smallest d within 5 % of the best (big sweeps keep d = 1: one workgroup per pair).  d = 23 (ONE row tile per pass, 22 + 1
  // passes, one per workgroup) only for <= 10 pairs and d = 45 (8-row passes) for <= 5, where the workgroups still fit ONE round: such
  // a workgroup walks the whole K / W1 stream like a two-tile pass does, so it does not cost a fraction of one and loses as soon as it
  // adds a round
  if (n <= 5) return 45;
  double best = 1e30;
  auto cost = [n](int d) { return (double)(((long long)n * d + 255) / 256) / d; };
  auto allowed = [n](int d) { return d < 23 || n <= 10; };
  for (const int d : {1, 2, 3, 4, 6, 12, 23})
    if (allowed(d) && cost(d) < best) best = cost(d);
  for (const int d : {1, 2, 3, 4, 6, 12, 23})
    if (allowed(d) && cost(d) <= 1.05 * best) return d;
  return 1;
}

// a2 + prepare (profile class delta_prep), c_conv1 contraction (delta_c12), c_conv2 GEMM (delta_c2)
// where ovn_delta_c12_f16x3_forward keeps A2raw inside its scratch (the yaw launch of a small sweep fills it: a2_done)
float* ovn_delta_f16x3_a2raw(void* scratch, int n) {
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  char* p = static_cast<char*>(scratch);
  p += al((size_t)n * 8 * sizeof(float)) + al((size_t)n * sizeof(unsigned)) + 2 * al((size_t)n * OVN_FEAT_ELEMS * sizeof(unsigned)) +
       al((size_t)n * LIN_ELEMS * sizeof(float));
  return reinterpret_cast<float*>(p);
}

int ovn_delta_c12_f16x3_forward(ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r,
                                const int32_t* ridx, int n, void* scratch, unsigned** o2max_out, float* o2, hipStream_t stream,
                                int pair0, const float* dcache_l, bool a2_done) {
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(delta_prepare_split_kernel<false>), PREP_SPLIT_LDS);
  if (rc) return rc;
  if (ridx) dcache_l = nullptr;   // the cache serves the 1-vs-N form (one query against many cached candidates)
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  char* p = static_cast<char*>(scratch);
  f32x4* scales = reinterpret_cast<f32x4*>(p);
  p += al((size_t)n * 8 * sizeof(float));
  unsigned* o2max = reinterpret_cast<unsigned*>(p);
  p += al((size_t)n * sizeof(unsigned));
  unsigned* pl = reinterpret_cast<unsigned*>(p);
  p += al((size_t)n * OVN_FEAT_ELEMS * sizeof(unsigned));
  unsigned* pr = reinterpret_cast<unsigned*>(p);
  p += al((size_t)n * OVN_FEAT_ELEMS * sizeof(unsigned));
  float* lin = reinterpret_cast<float*>(p);
  p += al((size_t)n * LIN_ELEMS * sizeof(float));
  float* a2raw = reinterpret_cast<float*>(p);
  p += al((size_t)(ridx ? n : 1) * A2_KSPLIT * A2_ELEMS * sizeof(float));
  float* o1raw = reinterpret_cast<float*>(p);
  p += al((size_t)n * O1RAW_ELEMS * sizeof(float));
  DeltaDesc* desc = reinterpret_cast<DeltaDesc*>(p);
  p += al((size_t)n * sizeof(DeltaDesc));
  unsigned* qblock = reinterpret_cast<unsigned*>(p);
  p += al(QBLOCK_WORDS * sizeof(unsigned));
  unsigned* live_buf = reinterpret_cast<unsigned*>(p);
  p += al(LIVE_WORDS * sizeof(unsigned));
  _Float16* w1c = reinterpret_cast<_Float16*>(p);
  // 1-vs-N sweeps (one query for all pairs): the query's live-channel list and the W1 fragments gathered for it; indexed pairs walk
  // the plain K (every pair has its own right volume)
  const unsigned* live = (ridx || !ctx->head_compact) ? nullptr : live_buf;
  ctx->dbg_live = live;   // ovn_head_walk_stats: the K walk of the most recent sweep (its last chunk)
  *o2max_out = o2max;
  const int nsplit = pick_nsplit(n);   // 45 / 23: 8-row / one-row-tile passes, one per workgroup, chosen for <= 5 / <= 10 pairs
  {
    OvnProfScope ps(ctx, OVN_K_DELTA_PREP, stream);
    if (!a2_done)
      hipLaunchKernelGGL(delta_a2_kernel, dim3(ridx ? n : 1, A2_KSPLIT), dim3(512), 0, stream, feats_r, ridx, ctx->w1raw, a2raw);
    if (dcache_l) {   // the query kernel also builds the live-channel list and gathers the W1 fragments for it
      hipLaunchKernelGGL(delta_query_kernel, dim3(live ? QV + QGW : QV), dim3(512), 0, stream, feats_r, a2raw, ctx->w2sum, qblock, live ? live_buf : nullptr,
                         reinterpret_cast<const _Float16*>(ctx->w1p_h), w1c);
    } else if (live) {
      hipLaunchKernelGGL(delta_live_kernel, dim3(1), dim3(512), 0, stream, feats_r, live_buf);
      hipLaunchKernelGGL(delta_w1c_kernel, dim3(240), dim3(256), 0, stream, reinterpret_cast<const _Float16*>(ctx->w1p_h), live, w1c);
    }
    hipLaunchKernelGGL(delta_prepare_split_kernel<false>, dim3(n), dim3(512), PREP_SPLIT_LDS, stream, feats_l, lidx, feats_r, ridx,
                       reinterpret_cast<const _Float16*>(ctx->wsp_h), ctx->w1col, ctx->b1, a2raw,
                       reinterpret_cast<const _Float16*>(ctx->w2p_h), ctx->w2sum, ctx->c2.bias, ctx->hs.sw1, ctx->hs.sw2, ctx->hs.sws,
                       ctx->hs.w1_colsum, ctx->hs.b1_absmax, 1.0f, scales, o2max, pl, pr, lin, desc, dcache_l, qblock, (float*)nullptr, live);
  }
  {
    OvnProfScope ps(ctx, OVN_K_DELTA, stream);
    if (nsplit == 45) {   // up to five pairs: 8-row passes, one per workgroup
      rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(delta_c1_f16x3_kernel<0, 3>), t_lds_bytes(3));
      if (rc) return rc;
      hipLaunchKernelGGL((delta_c1_f16x3_kernel<0, 3>), dim3(n * nsplit), dim3(512), t_lds_bytes(3), stream, desc,
                         reinterpret_cast<const _Float16*>(ctx->w1p_h), scales, o1raw, 1, nsplit, pair0, lidx, live, w1c);
    } else if (nsplit == 23) {   // a handful of pairs: one row tile per pass, one pass per workgroup
      rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(delta_c1_f16x3_kernel<1, 3>), t_lds_bytes(3));
      if (rc) return rc;
      hipLaunchKernelGGL((delta_c1_f16x3_kernel<1, 3>), dim3(n * nsplit), dim3(512), t_lds_bytes(3), stream, desc,
                         reinterpret_cast<const _Float16*>(ctx->w1p_h), scales, o1raw, 1, nsplit, pair0, lidx, live, w1c);
    } else {
      rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(delta_c1_f16x3_kernel<2, 3>), t_lds_bytes(3));
      if (rc) return rc;
      hipLaunchKernelGGL((delta_c1_f16x3_kernel<2, 3>), dim3(n * nsplit), dim3(512), t_lds_bytes(3), stream, desc,
                         reinterpret_cast<const _Float16*>(ctx->w1p_h), scales, o1raw, 1, nsplit, pair0, lidx, live, w1c);
    }
  }
  {
    OvnProfScope ps(ctx, OVN_K_DELTA_C2, stream);
    const int total_rows = n * G * G;
    if (n <= 28)   // 9 workgroups of 64 rows per pair, operands several k-steps ahead (a handful of pairs: latency, not throughput)
      hipLaunchKernelGGL((delta_c2_f16x3_kernel<1, 6, 3>), dim3(total_rows / 64), dim3(256), 0, stream, o1raw,
                         reinterpret_cast<const _Float16*>(ctx->w2p_h), desc, scales, o2, o2max, 1.0f);
    else
      hipLaunchKernelGGL((delta_c2_f16x3_kernel<3, 3, 1>), dim3(total_rows / C2_TILE_ROWS), dim3(256), 0, stream, o1raw,
                         reinterpret_cast<const _Float16*>(ctx->w2p_h), desc, scales, o2, o2max, 1.0f);
  }
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

// ovn_head_walk_stats: {largest slice count, live channels, slices of passes 0 .. 11, compacted?, 0} of the most recent sweep
int ovn_delta_walk_stats(ovn_ctx* ctx, int32_t* out16, hipStream_t stream) {
  for (int i = 0; i < 16; ++i) out16[i] = 0;
  if (ctx->dbg_live == nullptr) {   // indexed pairs, fp32 mode, compaction off: every pass walks the 4 slices of all 128 channels
    out16[0] = 4;
    out16[1] = FC;
    for (int p = 0; p < NPAIR; ++p) out16[2 + p] = 4;
    return OVN_OK;
  }
  unsigned h[LIVE_WORDS];
  OVN_HIP_CHECK(hipMemcpyAsync(h, ctx->dbg_live, sizeof(h), hipMemcpyDeviceToHost, stream));
  OVN_HIP_CHECK(hipStreamSynchronize(stream));
  out16[0] = (int32_t)h[0];
  out16[1] = (int32_t)h[1];
  const unsigned char* nsp = reinterpret_cast<const unsigned char*>(h + 4) + FC;
  for (int p = 0; p < NPAIR; ++p) out16[2 + p] = nsp[p];
  out16[14] = 1;
  out16[15] = nsp[TB_STEPS - FC] < S ? (int32_t)nsp[TB_STEPS - FC] : 0;   // steps of the packed last slice (0: none)
  return OVN_OK;
}

// Delta cache rows of n feature volumes (ovn_delta_cache): delta_prepare_split_kernel<true>, one workgroup per volume.
int ovn_delta_cache_forward(ovn_ctx* ctx, const float* feats, int n, float* cache, hipStream_t stream) {
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(delta_prepare_split_kernel<true>), PREP_SPLIT_LDS);
  if (rc) return rc;
  hipLaunchKernelGGL(delta_prepare_split_kernel<true>, dim3(n), dim3(512), PREP_SPLIT_LDS, stream, feats, (const int32_t*)nullptr,
                     (const float*)nullptr, (const int32_t*)nullptr, reinterpret_cast<const _Float16*>(ctx->wsp_h), ctx->w1col, ctx->b1,
                     (const float*)nullptr, reinterpret_cast<const _Float16*>(ctx->w2p_h), ctx->w2sum, ctx->c2.bias, ctx->hs.sw1,
                     ctx->hs.sw2, ctx->hs.sws, ctx->hs.w1_colsum, ctx->hs.b1_absmax, 1.0f, (f32x4*)nullptr, (unsigned*)nullptr,
                     (unsigned*)nullptr, (unsigned*)nullptr, (float*)nullptr, (DeltaDesc*)nullptr, (const float*)nullptr,
                     (const unsigned*)nullptr, cache, (const unsigned*)nullptr);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

// Weight statistics (host copies: this call synchronises `stream`, like every weight registration), the tap-summed c_conv1
// kernel of the linear terms, and the scaled fp16 hi/lo fragments of c_conv1 / c_conv2.
int ovn_delta_prepare_f16x3(ovn_ctx* ctx, const float* c1_kernel_dev, const float* c1_bias_dev, const float* c2_kernel_dev,
                            hipStream_t stream) {
  OvnHeadScales* hs = &ctx->hs;
  float* stats = nullptr;
  OVN_HIP_CHECK(hipMalloc((void**)&stats, 6 * sizeof(float)));
  hipLaunchKernelGGL(delta_wstats_kernel, dim3(1), dim3(256), 0, stream, c1_kernel_dev, K1, O1, stats);
  hipLaunchKernelGGL(delta_wstats_kernel, dim3(1), dim3(256), 0, stream, c2_kernel_dev, K2, O2, stats + 2);
  hipLaunchKernelGGL(delta_wstats_kernel, dim3(1), dim3(256), 0, stream, c1_bias_dev, 1, O1, stats + 4);
  float h[6];
  hipError_t e = hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(stats);
  OVN_HIP_CHECK(e);
  hs->sw1 = ovn_pow2_scale_for(h[0]);
  hs->w1_colsum = h[1];
  hs->sw2 = ovn_pow2_scale_for(h[2]);
  hs->b1_absmax = h[4];
  const size_t w1_elems = (size_t)S * FC * O1 * 2;   // hi + lo
  const size_t w2_elems = (size_t)K2 * O2 * 2;
  OVN_HIP_CHECK(hipMalloc(&ctx->w1p_h, w1_elems * sizeof(_Float16)));
  OVN_HIP_CHECK(hipMalloc(&ctx->w2p_h, w2_elems * sizeof(_Float16)));
  OVN_HIP_CHECK(hipMalloc((void**)&ctx->w1raw, (size_t)K1 * O1 * sizeof(float)));
  OVN_HIP_CHECK(hipMalloc((void**)&ctx->w1sum, (size_t)FC * O1 * sizeof(float)));
  OVN_HIP_CHECK(hipMalloc((void**)&ctx->w1col, (size_t)O1 * sizeof(float)));
  OVN_HIP_CHECK(hipMalloc(&ctx->wsp_h, (size_t)FC * O1 * 2 * sizeof(_Float16)));
  OVN_HIP_CHECK(hipMalloc((void**)&ctx->w2sum, (size_t)O1 * O2 * sizeof(float)));
  hipLaunchKernelGGL(delta_w2sum_kernel, dim3((O1 * O2 + 255) / 256), dim3(256), 0, stream, c2_kernel_dev, ctx->w2sum);
  OVN_HIP_CHECK(hipMemcpyAsync(ctx->w1raw, c1_kernel_dev, (size_t)K1 * O1 * sizeof(float), hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(delta_w1sum_kernel, dim3((FC * O1 + 255) / 256), dim3(256), 0, stream, c1_kernel_dev, ctx->w1sum, ctx->w1col);
  {   // scale of the tap-summed kernel (up to 15x the largest single weight)
    float* st = nullptr;
    OVN_HIP_CHECK(hipMalloc((void**)&st, 2 * sizeof(float)));
    hipLaunchKernelGGL(delta_wstats_kernel, dim3(1), dim3(256), 0, stream, ctx->w1sum, FC, O1, st);
    float hws[2];
    hipError_t e2 = hipMemcpyAsync(hws, st, sizeof(hws), hipMemcpyDeviceToHost, stream);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(stream);
    (void)hipFree(st);
    OVN_HIP_CHECK(e2);
    hs->sws = ovn_pow2_scale_for(hws[0]);
  }
  hipLaunchKernelGGL(delta_prep_ws_f16_kernel, dim3(32), dim3(256), 0, stream, ctx->w1sum, reinterpret_cast<_Float16*>(ctx->wsp_h), hs->sws);
  hipLaunchKernelGGL(delta_prep_w1_f16_kernel, dim3(240), dim3(256), 0, stream, c1_kernel_dev,
                     reinterpret_cast<_Float16*>(ctx->w1p_h), hs->sw1);
  hipLaunchKernelGGL(delta_prep_w2_f16_kernel, dim3(240), dim3(256), 0, stream, c2_kernel_dev,
                     reinterpret_cast<_Float16*>(ctx->w2p_h), hs->sw2);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
