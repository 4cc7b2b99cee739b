// Correlation (yaw) head in spectral form for gfx950: HBM-bound by construction.
//
// Reference op (NormalizedCorrelation2D.py:43-109 with normalize='none', RangePadding2D.py:31-38, infer.py:158):
//     corr[k] = sum_j sum_c l[(k + j + 180) mod 360, c] * r[j, c],   yaw = 180 - argmax_k corr[k]
// is, per channel, a circular cross-correlation of length 360.  With X^[f,c] = sum_i x[i,c] e^{-2 pi i f i / 360}:
//     C^[f]   = sum_c L^[f,c] * conj(R^[f,c])                      (correlation theorem, r real)
//     corr[k] = (1/360) sum_f C^[f] e^{+2 pi i f (k+180)/360}     (real part; Hermitian half f = 0..180 is enough)
// The direct form costs 33.2 MFLOP per pair against 184 KB read (180 FLOP/B: matrix-core bound, corr_head.hip);
// this form costs 0.45 MFLOP per pair against 188 KB read (2.4 FLOP/B) once every scan's spectrum is cached next
// to its feature volume -- the sweep then streams candidate spectra at HBM speed.
//
//   spectrum layout per scan: [c = 0..127][368] floats: Re X^[f,c] at [f], Im X^[f,c] at [184 + f], f = 0..180,
//   columns 181..183 and 365..367 are zero padding (keeps every row 16-byte aligned for float4 loads).
//   * ovn_spectrum:  the DFT is a dense contraction with a constant 360 x 368 twiddle matrix: dft_f16x3_kernel (scaled fp16
//     hi/lo split on the fp16 matrix cores, default) or, in the fp32 head mode, the generic fp32 conv kernel (a 360x1 'valid'
//     convolution over the (360,128) feature image).
//   * spectral_corr_kernel: ONE launch per sweep, one workgroup per pair: spectral product (the candidate spectrum streamed once),
//     inverse transform of the 181 Hermitian coefficients shifted by W/2 in fp64 registers, first-maximum argmax -- nothing but
//     yaw (and, on request, the 360 correlation values) leaves the kernel.
#include <math.h>

#include <vector>

#include "ovn_internal.h"
#include "delta_a2.h"

namespace {

constexpr int FW = OVN_FEAT_W;     // 360
constexpr int FC = OVN_FEAT_C;     // 128
constexpr int NF = FW / 2 + 1;     // 181 non-redundant frequencies
constexpr int IM_OFF = 184;        // start of the imaginary block (16-byte aligned)
constexpr int SW = OVN_SPEC_W;     // 368 floats per spectrum row
constexpr int FQ = IM_OFF / 4;     // 46 frequency quads
constexpr int CG = 4;              // channel groups of 32
constexpr int PROD_THREADS = 192;  // 184 working threads
constexpr int SWP = 384;           // SW padded with zero filters to 3 x 128 so the conv kernel can use its 64 x 128 tile

// Yaw head of one pair in ONE launch: spectral product -> Hermitian inverse transform shifted by W/2 (RangePadding2D) -> first
// maximum (infer.py:158).  One workgroup per pair.
//   phase 1 (HBM-bound): thread = (4 consecutive frequencies, 32 channels); the candidate spectrum is streamed once (188,416 B), the
//     query spectrum comes from L2; partial sums of the 4 channel groups are combined in LDS in a fixed order.
//   phase 2: corr[k] = sum_f wf/360 (Re C^[f] cos(2 pi f (k+180)/360) - Im C^[f] sin(2 pi f (k+180)/360)), wf = 1 for f = 0, 180, else 2.
//     With a_f = (-1)^f wf/360 Re C^[f], b_f = (-1)^f wf/360 Im C^[f] (the shift by 180 bins is the sign (-1)^f) and
//     E = sum_f a_f cos(f x), O = sum_f b_f sin(f x), x = 2 pi k / 360, split by the parity of f (Ee, Eo, Oe, Oo):
//         corr[k] = (Ee + Eo) - (Oe + Oo)      corr[360 - k] = (Ee + Eo) + (Oe + Oo)
//         corr[180 - k] = (Ee - Eo) + (Oe - Oo)      corr[180 + k] = (Ee - Eo) - (Oe - Oo)
//     so k = 0..90 gives all 360 bins.  Thread (k, half h) walks 90 / 92 frequencies with the Chebyshev recurrence
//     cos((f+1)x) = 2 cos x cos(fx) - cos((f-1)x) (same for sin) in FP64, started from an exact fp64 table: four DFMAs per (k, f),
//     no table lookups in the loop, no LDS bank conflicts, and the sums are exact to fp64 rounding (the fp32 contraction this
//     replaces carried 181 fp32 roundings per bin).  The two halves are added in a fixed order.
//   phase 3: first maximum over the 360 bins (lowest index wins ties) by a fixed-order wave + LDS reduction; yaw = 180 - argmax.
constexpr int KH = 96;             // threads per frequency half in phase 2 (k = 0..90 active)
constexpr int NK = FW / 4 + 1;     // 91 values of k
constexpr int F_SPLIT = 90;        // half 0: f = 0..89, half 1: f = 90..180 (+ the zero pad 181)

// A2 = true (small 1-vs-N sweeps): workgroups n_pairs .. n_pairs + 21 of the launch run the 64 wave tasks of the Delta head's
// right-volume term for the sweep's query instead (delta_a2.h) -- independent of the yaw head, one launch less in a single query's chain.
constexpr int A2_WGS = (OVN_A2_TASKS + PROD_THREADS / 64 - 1) / (PROD_THREADS / 64);   // 22
template <bool A2>
__global__ __launch_bounds__(PROD_THREADS) void spectral_corr_kernel(const float* __restrict__ spec_l,
                                                                    const int32_t* __restrict__ lidx,
                                                                    const float* __restrict__ spec_r,
                                                                    const int32_t* __restrict__ ridx,
                                                                    const double* __restrict__ tw64,   // [2][360]: cos, sin(2 pi m / 360)
                                                                    int32_t* __restrict__ yaw, float* __restrict__ corr_out,
                                                                    int n_pairs, const float* __restrict__ a2_feats_r,
                                                                    const float* __restrict__ a2_w1raw, float* __restrict__ a2raw) {
  __shared__ float part[CG][2][IM_OFF];
  __shared__ __attribute__((aligned(16))) double ab[NF + 1][2];
  __shared__ double half1[NK][4];
  __shared__ float rv[3];
  __shared__ int ri[3];
  if (A2 && (int)blockIdx.x >= n_pairs) {   // workgroup-uniform
    const int task = ((int)blockIdx.x - n_pairs) * (PROD_THREADS / 64) + (int)(threadIdx.x >> 6);
    if (task < OVN_A2_TASKS) ovn_delta_a2_task(a2_feats_r, a2_w1raw, a2raw, task, (int)(threadIdx.x & 63));
    return;
  }
  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const float* L = spec_l + (long long)(lidx ? lidx[pair] : pair) * OVN_SPEC_ELEMS;
  const float* R = spec_r + (long long)(ridx ? ridx[pair] : 0) * OVN_SPEC_ELEMS;
  // table entries of this thread's recurrence start: issued before the streaming loop, used after it
  const int k = tid < KH ? tid : tid - KH;
  const int half = tid < KH ? 0 : 1;
  const bool idft_thread = k < NK;
  const int f0 = half ? F_SPLIT : 0;
  double c0 = 0.0, cm = 0.0, s0 = 0.0, sm = 0.0, twoc = 0.0;
  if (idft_thread) {
    const int m0 = (f0 * k) % FW;                    // angle index of f0
    const int m1 = (m0 + FW - k) % FW;               // ... of f0 - 1
    c0 = tw64[m0];
    s0 = tw64[FW + m0];
    cm = tw64[m1];
    sm = tw64[FW + m1];
    twoc = 2.0 * tw64[k];
  }
  if (tid < FQ * CG) {
    const int fq = tid % FQ;
    const int cg = tid / FQ;
    f32x4 sre = {0.f, 0.f, 0.f, 0.f}, sim = {0.f, 0.f, 0.f, 0.f};
    const float* lrow = L + (cg * 32) * SW + 4 * fq;
    const float* rrow = R + (cg * 32) * SW + 4 * fq;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) {
      const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(lrow + c * SW));            // Re L^ (read once)
      const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(lrow + c * SW + IM_OFF));   // Im L^
      const f32x4 p = *reinterpret_cast<const f32x4*>(rrow + c * SW);            // Re R^
      const f32x4 q = *reinterpret_cast<const f32x4*>(rrow + c * SW + IM_OFF);   // Im R^
      sre += a * p + b * q;   // (a + ib)(p - iq)
      sim += b * p - a * q;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      part[cg][0][4 * fq + e] = sre[e];
      part[cg][1][4 * fq + e] = sim[e];
    }
  }
  __syncthreads();
  if (tid < NF + 1) {
    double a = 0.0, b = 0.0;
    if (tid < NF) {
      const float re = (part[0][0][tid] + part[1][0][tid]) + (part[2][0][tid] + part[3][0][tid]);
      const float im = (part[0][1][tid] + part[1][1][tid]) + (part[2][1][tid] + part[3][1][tid]);
      const double wf = ((tid == 0 || tid == FW / 2) ? 1.0 : 2.0) / (double)FW * ((tid & 1) ? -1.0 : 1.0);
      a = wf * (double)re;
      b = wf * (double)im;
    }
    ab[tid][0] = a;   // entry 181 is a zero pad: both halves walk an even number of frequencies
    ab[tid][1] = b;
  }
  __syncthreads();
  // phase 2: both halves start on an even frequency and walk whole (even, odd) pairs: half 0 f = 0..89, half 1 f = 90..181 (181 = pad)
  double ee = 0.0, eo = 0.0, oe = 0.0, oo = 0.0;
  if (idft_thread) {
    const int npairs = half ? (NF + 1 - F_SPLIT) / 2 : F_SPLIT / 2;   // 46 : 45
    double cc = c0, cp = cm, sc = s0, sp = sm;                         // cos / sin of f x and of (f - 1) x
    for (int j = 0; j < npairs; ++j) {
      const int f = f0 + 2 * j;
      const double a0 = ab[f][0], b0 = ab[f][1], a1 = ab[f + 1][0], b1 = ab[f + 1][1];
      const double cn = __builtin_fma(twoc, cc, -cp), sn = __builtin_fma(twoc, sc, -sp);     // (f + 1) x
      ee = __builtin_fma(a0, cc, ee);
      oe = __builtin_fma(b0, sc, oe);
      eo = __builtin_fma(a1, cn, eo);
      oo = __builtin_fma(b1, sn, oo);
      cp = cn;
      sp = sn;
      cc = __builtin_fma(twoc, cn, -cc);                                                      // (f + 2) x
      sc = __builtin_fma(twoc, sn, -sc);
    }
    if (half) {
      half1[k][0] = ee;
      half1[k][1] = eo;
      half1[k][2] = oe;
      half1[k][3] = oo;
    }
  }
  __syncthreads();
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  if (!half && idft_thread) {
    ee += half1[k][0];
    eo += half1[k][1];
    oe += half1[k][2];
    oo += half1[k][3];
    const double es = ee + eo, ed = ee - eo, os = oe + oo, od = oe - oo;
    // bins k, 360 - k, 180 - k, 180 + k (each bin exactly once over k = 0..90), in increasing index order per candidate list
    const float v0 = (float)(es - os), v1 = (float)(es + os), v2 = (float)(ed + od), v3 = (float)(ed - od);
    const int i0 = k, i1 = FW - k, i2 = FW / 2 - k, i3 = FW / 2 + k;
    const bool u0 = true, u1 = (k >= 1), u2 = (k <= NK - 2), u3 = (k >= 1 && k <= NK - 2);
    float* co = corr_out ? corr_out + (long long)pair * FW : nullptr;
#define OVN_TAKE(U, V, I)                                   \
  if (U) {                                                  \
    if (co) co[I] = V;                                      \
    if (V > bv || (V == bv && I < bi)) {                    \
      bv = V;                                               \
      bi = I;                                               \
    }                                                       \
  }
    OVN_TAKE(u0, v0, i0)
    OVN_TAKE(u1, v1, i1)
    OVN_TAKE(u2, v2, i2)
    OVN_TAKE(u3, v3, i3)
#undef OVN_TAKE
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_down(bv, off, 64);
    const int oi = __shfl_down(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if ((tid & 63) == 0) {
    rv[tid >> 6] = bv;
    ri[tid >> 6] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    float v = rv[0];
    int i = ri[0];
    for (int w = 1; w < 2; ++w)   // bins live in threads 0..90: waves 0 and 1
      if (rv[w] > v || (rv[w] == v && ri[w] < i)) {
        v = rv[w];
        i = ri[w];
      }
    yaw[pair] = FW / 2 - (i == 0x7fffffff ? 0 : i);   // every bin NaN: bin 0, like np.argmax's first NaN
  }
}

// ---- forward DFT on the fp16 matrix cores (f16x3 arithmetic, see delta_head_f16x3.hip) -------------------------------------------
// spectra[scan][c][col] = sum_i X[scan][i][c] T[i][col]: per scan a (128 x 360) x (360 x 368) product whose A operand is the
// TRANSPOSE of the feature image.  Workgroup = (scan, half of the channels): the 360 x 64 slab is read in items of 4 channels x
// 8 rows (all of a thread's loads in flight), scaled by a power of two from its own largest |value|, split into fp16 hi / lo and
// written transposed into LDS with 16-byte stores ([channel][i], rows 784 B apart: the 16 rows of an A-fragment read start in 16
// different bank groups);
// 8 waves x (4 m-tiles x 3 n-tiles), 12 K steps of 32 (i >= 360: zero twiddles against zeroed LDS columns), twiddle fragments
// straight from L2 one step ahead.  The generic fp32 conv kernel gathers the strided operand element by element: 85 us per
// 128 scans against ~15 here.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int DFT_CH = 64;                 // channels per workgroup
constexpr int DFT_KS = (FW + 31) / 32;     // 12 K steps
constexpr int DFT_LD = 392;                // fp16 elements per LDS row (>= 384; 196 dwords = 4 mod 64)
constexpr int DFT_NT = SWP / 16;           // 24 n-tiles
constexpr size_t DFT_LDS = 2 * (size_t)DFT_CH * DFT_LD * sizeof(_Float16) + 64;

// NTW n-tiles per wave: 3 (one workgroup covers all 24 n-tiles) or 1 (three workgroups per slab, blockIdx.y: a handful of scans
// then still fills 6 x as many CUs; the slab staging is repeated, the K loop is a third as long)
template <int NTW>
__global__ __launch_bounds__(512) void dft_f16x3_kernel(const float* __restrict__ feats, const _Float16* __restrict__ tw,
                                                       float sT, float one, float* __restrict__ spectra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  _Float16* ah = reinterpret_cast<_Float16*>(dsm);
  _Float16* al = ah + DFT_CH * DFT_LD;
  float* red = reinterpret_cast<float*>(al + DFT_CH * DFT_LD);
  const int scan = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, g = lane >> 4;
  const float* X = feats + (size_t)scan * OVN_FEAT_ELEMS + DFT_CH * half;

  // slab -> registers.  Item = (4 channels, 8 consecutive rows i): 16 x 45 = 720 items, two per thread (the second round is
  // partial); a lane then owns 8 consecutive K positions of 4 channel rows = one 16-byte LDS write per row and image.
  constexpr int NITEM = (DFT_CH / 4) * (FW / 8);   // 720
  f32x4 v[2][8];
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int item = tid + 512 * k;
    const int cq = item & 15, ib = (item < NITEM) ? (item >> 4) : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[k][j] = *reinterpret_cast<const f32x4*>(X + (size_t)(8 * ib + j) * FC + 4 * cq);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[k][j][0]), fabsf(v[k][j][1])), fmaxf(fabsf(v[k][j][2]), fabsf(v[k][j][3]))));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if (lane == 0) red[wave] = m;
  // zero the K padding columns i = 360 .. 383 of every row (64 rows x 24 columns x hi/lo = 3 x 16 B per row and image)
  if (tid < DFT_CH * 3 * 2) {
    const int img = tid / (DFT_CH * 3), r = (tid / 3) % DFT_CH, q = tid % 3;
    *reinterpret_cast<f32x4*>((img ? al : ah) + r * DFT_LD + FW + 8 * q) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
  const float sx = ovn_pow2_scale_for(m);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int item = tid + 512 * k;
    if (item < NITEM) {
      const int cq = item & 15, ib = item >> 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // channel 4 cq + e: K positions 8 ib .. 8 ib + 7
        unsigned hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float x0 = v[k][j][e] * sx, x1 = v[k][j + 1][e] * sx;
          const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
          f16x2 lp;
          lp[0] = (_Float16)__builtin_fmaf(x0, one, -(float)hp[0]);
          lp[1] = (_Float16)__builtin_fmaf(x1, one, -(float)hp[1]);
          hw[j >> 1] = __builtin_bit_cast(unsigned, hp);
          lw[j >> 1] = __builtin_bit_cast(unsigned, lp);
        }
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4v*>(ah + (4 * cq + e) * DFT_LD + 8 * ib) = (u32x4v){hw[0], hw[1], hw[2], hw[3]};
        *reinterpret_cast<u32x4v*>(al + (4 * cq + e) * DFT_LD + 8 * ib) = (u32x4v){lw[0], lw[1], lw[2], lw[3]};
      }
    }
  }
  __syncthreads();

  f32x4 acc[4][NTW];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // twiddle fragments [ks][nt(24)][hi,lo][lane][8]; this wave's n-tiles nt0 .. nt0 + NTW - 1
  const int nt0 = NTW * (8 * blockIdx.y + wave);
  const _Float16* tsrc = tw + ((size_t)nt0 * 2) * 512 + lane * 8;
  // an L2 round trip is ~3 K steps long (36 MFMAs each): fragments of three steps ahead are in flight (ring of 4 slots)
  f16x8 bq[4][NTW][2];
#define DFT_LOAD_B(SLOT, KS)                                                                     \
  _Pragma("unroll") for (int j = 0; j < NTW; ++j) {                                              \
    const _Float16* q = tsrc + (size_t)(KS) * (DFT_NT * 2 * 512) + j * 1024;                     \
    bq[SLOT][j][0] = *reinterpret_cast<const f16x8*>(q);                                         \
    bq[SLOT][j][1] = *reinterpret_cast<const f16x8*>(q + 512);                                   \
  }
  DFT_LOAD_B(0, 0)
  DFT_LOAD_B(1, 1)
  DFT_LOAD_B(2, 2)
#pragma unroll
  for (int ks = 0; ks < DFT_KS; ++ks) {
    if (ks + 3 < DFT_KS) DFT_LOAD_B((ks + 3) & 3, ks + 3)
    f16x8 fh[4], fl[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      fh[mt] = *reinterpret_cast<const f16x8*>(ah + (16 * mt + lrow) * DFT_LD + 32 * ks + 8 * g);
      fl[mt] = *reinterpret_cast<const f16x8*>(al + (16 * mt + lrow) * DFT_LD + 32 * ks + 8 * g);
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[mt], bq[ks & 3][j][0], acc[mt][j], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[mt], bq[ks & 3][j][0], acc[mt][j], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[mt], bq[ks & 3][j][1], acc[mt][j], 0, 0, 0);
    }
  }
#undef DFT_LOAD_B
  // C/D layout: lane holds column lrow of its n-tile, rows (channels) 4g .. 4g+3 of each m-tile
  const float inv = 1.0f / (sx * sT);
  float* out = spectra + (size_t)scan * OVN_SPEC_ELEMS + (size_t)(DFT_CH * half) * SW;
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int col = 16 * (nt0 + j) + lrow;
    if (col < SW) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(16 * mt + 4 * g + r) * SW + col] = acc[mt][j][r] * inv;
    }
  }
}

int upload_layer(OvnConvLayer* L, const std::vector<float>& w, hipStream_t stream, bool f16 = false) {
  float* dw = nullptr;
  float* db = nullptr;
  OVN_HIP_CHECK(hipMalloc((void**)&dw, w.size() * sizeof(float)));
  OVN_HIP_CHECK(hipMalloc((void**)&db, (size_t)L->cout * sizeof(float)));
  OVN_HIP_CHECK(hipMemcpyAsync(dw, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  OVN_HIP_CHECK(hipMemsetAsync(db, 0, (size_t)L->cout * sizeof(float), stream));
  int rc = ovn_conv_prepare(L, dw, db, stream);  // synchronises the stream
  if (rc == OVN_OK && f16) rc = ovn_conv_prepare_f16x3(L, dw, stream);   // + scaled fp16 hi/lo fragments (dft_f16x3_kernel)
  (void)hipFree(dw);
  (void)hipFree(db);
  return rc;
}

}  // namespace

// Constant twiddle layers (built once per context).
int ovn_spectral_prepare(ovn_ctx* ctx, hipStream_t stream) {
  const double w0 = 2.0 * 3.14159265358979323846 / FW;
  {  // forward DFT as a (360,1,1,368) 'valid' convolution over the (360,128,1) feature image
    OvnConvLayer& L = ctx->dft;
    L = OvnConvLayer();
    L.name = "dft360";
    L.kh = FW;
    L.kw = 1;
    L.cin = 1;
    L.cout = SWP;
    L.out_cols = SW;
    L.sh = 1;
    L.sw = 1;
    L.relu = 0;
    std::vector<float> w((size_t)FW * SWP, 0.f);
    for (int i = 0; i < FW; ++i)
      for (int f = 0; f < NF; ++f) {
        const double ang = w0 * (double)((long long)f * i % FW);
        w[(size_t)i * SWP + f] = (float)cos(ang);
        w[(size_t)i * SWP + IM_OFF + f] = (float)(-sin(ang));
      }
    int rc = upload_layer(&L, w, stream, true);
    if (rc) return rc;
  }
  {  // exact fp64 table cos / sin (2 pi m / 360), m = 0..359: start values of the inverse transform's recurrences (spectral_corr_kernel)
    std::vector<double> t(2 * FW);
    for (int m = 0; m < FW; ++m) {
      // octant reduction so that the table is exactly symmetric (cos 90 deg = 0, sin 180 deg = 0, ...)
      const int q = m / 90, r = m % 90;
      const double ang = w0 * (double)r;
      const double c = (r == 0) ? 1.0 : cos(ang), sn = (r == 0) ? 0.0 : sin(ang);
      const double cs[4] = {c, -sn, -c, sn}, ss[4] = {sn, c, -sn, -c};
      t[m] = cs[q];
      t[FW + m] = ss[q];
    }
    if (ctx->tw64) (void)hipFree(ctx->tw64);
    ctx->tw64 = nullptr;
    OVN_HIP_CHECK(hipMalloc((void**)&ctx->tw64, t.size() * sizeof(double)));
    OVN_HIP_CHECK(hipMemcpy(ctx->tw64, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  return OVN_OK;
}

int ovn_spectrum_forward(ovn_ctx* ctx, const float* feats, int n, float* spectra, hipStream_t stream) {
  if (ctx->head_mode != 0) {   // f16x3 arithmetic (default)
    int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(dft_f16x3_kernel<3>), DFT_LDS);
    if (rc) return rc;
    rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(dft_f16x3_kernel<1>), DFT_LDS);
    if (rc) return rc;
    if (n <= 64)   // up to 128 slabs: three workgroups per slab
      hipLaunchKernelGGL(dft_f16x3_kernel<1>, dim3(2 * n, 3), dim3(512), DFT_LDS, stream, feats,
                         reinterpret_cast<const _Float16*>(ctx->dft.wp_h), ctx->dft.sw_h, 1.0f, spectra);
    else
      hipLaunchKernelGGL(dft_f16x3_kernel<3>, dim3(2 * n), dim3(512), DFT_LDS, stream, feats,
                         reinterpret_cast<const _Float16*>(ctx->dft.wp_h), ctx->dft.sw_h, 1.0f, spectra);
    OVN_HIP_CHECK(hipGetLastError());
    return OVN_OK;
  }
  int oh = 0, ow = 0;
  // input viewed as (n, H=360, W=128, C=1): out (n, 1, 128, 368) = spectra (n, 128, 368)
  return ovn_conv_forward(ctx->dft, feats, n, FW, FC, spectra, &oh, &ow, stream);
}

int ovn_corr_spectral_forward(ovn_ctx* ctx, const float* spec_l, const int32_t* lidx, const float* spec_r,
                              const int32_t* ridx, int n, int32_t* yaw, float* corr, hipStream_t stream, const float* a2_feats_r,
                              float* a2raw) {
  if (a2_feats_r != nullptr && a2raw != nullptr)
    hipLaunchKernelGGL(spectral_corr_kernel<true>, dim3(n + A2_WGS), dim3(PROD_THREADS), 0, stream, spec_l, lidx, spec_r, ridx,
                       ctx->tw64, yaw, corr, n, a2_feats_r, ctx->w1raw, a2raw);
  else
    hipLaunchKernelGGL(spectral_corr_kernel<false>, dim3(n), dim3(PROD_THREADS), 0, stream, spec_l, lidx, spec_r, ridx, ctx->tw64, yaw,
                       corr, n, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
