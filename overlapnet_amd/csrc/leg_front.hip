// The first two leg layers fused for the network.yml input (C = 4: depth + normals), f16x3 arithmetic, for gfx950:
//   s_conv1  5 x 15, stride (2, 2),  4 -> 16, ReLU   (generateNet.py:161-165)     64 x 900 x 4  -> 30 x 443 x 16
//   s_conv2  3 x 15, stride (2, 1), 16 -> 32, ReLU   (generateNet.py:167-171)     30 x 443 x 16 -> 14 x 429 x 32
//
// As two strip kernels (conv_strip.hip) these layers are bound by HBM, not by the matrix pipe (profiles/r3_leg_pmc.md: 30 % / 52 %
// busy, waves parked 56-62 % of the time): each moves its whole input and output -- 0.92 + 0.85 MB and 0.85 + 0.77 MB per scan -- for
// 0.4 GFLOP.  Here the 850 KB s_conv1 activation never leaves the CU: a workgroup owns 2 output rows x 144 pixels of s_conv2, computes
// the 5 x 160 s_conv1 outputs they need from a 13-row x 342-pixel input strip (staged, scaled by its own maximum and split ONCE
// into LDS, as conv_strip_small_kernel does), rescales that tile by its own maximum, splits it into the LDS space the input strip
// occupied, and runs s_conv2 on it.  Halo recompute of s_conv1: 5 rows for 4 (stride 2), 160 pixels for 144: + 39 % of 0.13 GFLOP.
// HBM per scan: 0.92 MB in (x 1.6 with the row / pixel halo, mostly absorbed by L2) + 0.77 MB out instead of 3.4 MB.
// Both stages keep the per-accumulator order of the unfused kernels (tap-major K walk, hi hi / lo hi / hi lo); the scale of the
// intermediate tile is a power of two taken from the tile itself (like leg_tail.hip), so a scan's result depends on that scan alone.
#include "ovn_internal.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// FRONT_ABL (timing-only builds of tools/experiments/front_ablate.sh, never the product): 1 strip of constants instead of the global
// loads, 2 no stage A MFMAs, 4 no stage B MFMAs, 8 no output stores, 16 no strip staging at all, 32 no intermediate-tile exchange
#ifndef FRONT_ABL
#define FRONT_ABL 0
#endif

namespace {

constexpr int C0 = 4, C1 = 16, C2 = 32;
constexpr int KH1 = 5, S1 = 2;                 // s_conv1: 5 x 15 (kernel row padded to 16 taps), stride 2 x 2
constexpr int KH2 = 3, SH2 = 2;                // s_conv2: 3 x 15 (padded to 16), stride 2 x 1
constexpr int TW2 = 144;                       // s_conv2 output pixels per workgroup
constexpr int RB = 2;                          // s_conv2 output rows per workgroup
constexpr int R1 = SH2 * (RB - 1) + KH2;       // 5 s_conv1 rows
constexpr int PIXM = TW2 + 16;                 // 160 s_conv1 pixels per row (taps 0..15 of the last output pixel)
constexpr int KHS1 = S1 * (R1 - 1) + KH1;      // 13 input rows
constexpr int PIXA1 = S1 * (PIXM - 1) + 16 + 8;   // 342 input pixels per row (+ slack for the 8-element fragment reads)
constexpr int IN_ELEMS = KHS1 * PIXA1 * C0;    // 17,784 fp16 per image (hi or lo)
constexpr int MID_ELEMS = R1 * PIXM * C1;      // 12,800 fp16 per image
static_assert(2 * MID_ELEMS <= 2 * IN_ELEMS, "the intermediate tile reuses the input strip's LDS");
constexpr size_t FRONT_LDS = 2 * (size_t)IN_ELEMS * sizeof(_Float16);   // 71,136 B: two workgroups per CU
// 4 waves per workgroup, two workgroups per CU (two waves per SIMD, 256 registers each):
// stage A: a wave owns COLUMNS of the 5 x 10 s_conv1 m-tiles (columns w, w + 4, w + 8): the five output rows of a column read the
//   same 13 input rows (row r feeds output row ry with kernel row ky = r - 2 ry), so an A fragment pair read from LDS feeds up to
//   nine MFMAs instead of three; the 10 K steps of s_conv1's weights (8 taps x 4 channels each) stay in 80 registers
// stage B: a wave owns m-tiles w, w + 4, .. of the 2 x 9 s_conv2 m-tiles and BOTH n-tiles: an A pair feeds six MFMAs
constexpr int NWF = 4;                          // waves
constexpr int NTHR = 64 * NWF;
constexpr int MTRA = PIXM / 16;                 // 10 s_conv1 m-tile columns
constexpr int COLS = (MTRA + NWF - 1) / NWF;    // 3 column slots per wave
constexpr int NKA = KH1 * 2;                    // 10 K steps of stage A
constexpr int MTRB = TW2 / 16, MTB = RB * MTRB; // 9 per row, 18 s_conv2 m-tiles
constexpr int MTHB = (MTB + NWF - 1) / NWF;     // 5 m-tile slots per wave
constexpr int NKB = KH2 * 8;                    // 24 K steps of stage B: 2 taps x 16 channels each

struct FrontArgs {
  const float* in;          // (nb, H, W, 4)
  const _Float16* wp1;      // s_conv1 fragments in the padded-tap order (OvnConvLayer::wp_h16)
  const float* b1;
  const _Float16* wp2;
  const float* b2;
  float* out;               // (nb, OH2, OW2, 32)
  unsigned* out_max;        // [scan] maxima of the s_conv2 output (scale of s_conv3), or NULL
  float sw1, sw2, one;
  int H, W, OH1, OW1, OH2, OW2, XT;
};

#define FRONT_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)

// s_conv1 for NC columns of m-tiles (c0, c0 + NWF, ..): accumulators acc[column][output row], input rows walked once
template <int NC>
__device__ __forceinline__ void front_stage_a(const _Float16* __restrict__ sh, const _Float16* __restrict__ sl, int c0, int lrow, int g,
                                              const f16x8 (&wh)[NKA], const f16x8 (&wl)[NKA], f32x4 (&acc)[NC][R1]) {
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int ry = 0; ry < R1; ++ry) acc[c][ry] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int cbase[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) cbase[c] = (S1 * (16 * (c0 + NWF * c) + lrow)) * C0 + 8 * g;
#pragma unroll
  for (int r = 0; r < KHS1; ++r) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      f16x8 fh[NC], fl[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        fh[c] = *reinterpret_cast<const f16x8*>(sh + cbase[c] + (r * PIXA1 + 8 * kh) * C0);
        fl[c] = *reinterpret_cast<const f16x8*>(sl + cbase[c] + (r * PIXA1 + 8 * kh) * C0);
      }
      // output rows fed by input row r: ky = r - 2 ry in [0, 5); per accumulator the K steps arrive in the order ks = 2 ky + kh
      // ascending (r ascending) and, within a step, hi hi / lo hi / hi lo -- the order of the unfused kernel
#pragma unroll
      for (int ry = 0; ry < R1; ++ry) {
        const int ky = r - S1 * ry;
        if (ky >= 0 && ky < KH1 && !(FRONT_ABL & 2)) {
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c][ry] = FRONT_MFMA(fh[c], wh[2 * ky + kh], acc[c][ry]);
        }
      }
#pragma unroll
      for (int ry = 0; ry < R1; ++ry) {
        const int ky = r - S1 * ry;
        if (ky >= 0 && ky < KH1 && !(FRONT_ABL & 2)) {
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c][ry] = FRONT_MFMA(fl[c], wh[2 * ky + kh], acc[c][ry]);
        }
      }
#pragma unroll
      for (int ry = 0; ry < R1; ++ry) {
        const int ky = r - S1 * ry;
        if (ky >= 0 && ky < KH1 && !(FRONT_ABL & 2)) {
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c][ry] = FRONT_MFMA(fh[c], wl[2 * ky + kh], acc[c][ry]);
        }
      }
    }
  }
}

__global__ __launch_bounds__(NTHR, 2) void leg_front_kernel(FrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char front_smem[];
  __shared__ float wg_red[16];
  _Float16* sh = reinterpret_cast<_Float16*>(front_smem);
  _Float16* sl = sh + IN_ELEMS;
  _Float16* mh = sh;                       // the intermediate tile takes over the strip's space once every wave is done with it
  _Float16* ml = sh + MID_ELEMS;
  const float one = a.one;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, g = lane >> 4;

  int bid = blockIdx.x;
  const int xt = bid % a.XT;
  bid /= a.XT;
  const int ohb = (a.OH2 + RB - 1) / RB;
  const int oy2 = RB * (bid % ohb);            // first s_conv2 row of the block
  const int b = bid / ohb;
  const int x0 = xt * TW2;                     // first s_conv2 pixel = first s_conv1 pixel of the tile
  const int tw = (a.OW2 - x0 < TW2) ? a.OW2 - x0 : TW2;
  const int oy1 = SH2 * oy2;                   // first s_conv1 row
  const int iy0 = S1 * oy1, ix0 = S1 * x0;     // first input row / pixel
  const int pixv = (a.W - ix0 < PIXA1) ? a.W - ix0 : PIXA1;

  // s_conv1's weights: all 10 K steps, requested before the strip (both round trips overlap)
  f16x8 wh[NKA], wl[NKA];
  {
    const _Float16* wbase1 = a.wp1 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < NKA; ++ks) {
      wh[ks] = *reinterpret_cast<const f16x8*>(wbase1 + (size_t)ks * (2 * 512));
      wl[ks] = *reinterpret_cast<const f16x8*>(wbase1 + (size_t)ks * (2 * 512) + 512);
    }
  }

  // ---- input strip -> LDS, scaled by its own maximum and split once (zero outside the image) ----
  float s_in = 1.0f;
  if (!(FRONT_ABL & 16)) {
    constexpr int TOTAL = KHS1 * PIXA1;        // one float4 (4 channels) per pixel
    constexpr int ITERS = (TOTAL + NTHR - 1) / NTHR;   // 18
    f32x4 v[ITERS];
#pragma unroll
    for (int u = 0; u < ITERS; ++u) {
      const int i = tid + u * NTHR;
      v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < TOTAL) {
        const int row = i / PIXA1, pix = i - row * PIXA1;
        if (FRONT_ABL & 1) v[u] = (f32x4){a.one, a.sw1, a.one, a.sw2};
        else if (pix < pixv && iy0 + row < a.H)
          v[u] = *reinterpret_cast<const f32x4*>(a.in + (((long long)b * a.H + iy0 + row) * a.W + ix0 + pix) * C0);
      }
    }
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < ITERS; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if (lane == 0) wg_red[wave] = m;
    __syncthreads();
    m = wg_red[0];
#pragma unroll
    for (int w = 1; w < NWF; ++w) m = fmaxf(m, wg_red[w]);
    s_in = ovn_pow2_scale_for(m);
#pragma unroll
    for (int u = 0; u < ITERS; ++u) {
      const int i = tid + u * NTHR;
      if (i < TOTAL) {
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const float x0f = v[u][e] * s_in, x1f = v[u][e + 1] * s_in;
          const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0f, x1f));
          h[e] = hp[0];
          h[e + 1] = hp[1];
          l[e] = (_Float16)__builtin_fmaf(x0f, one, -(float)hp[0]);
          l[e + 1] = (_Float16)__builtin_fmaf(x1f, one, -(float)hp[1]);
        }
        *reinterpret_cast<f16x4*>(sh + 4 * i) = h;           // [row][pix][4] is the linear order of i
        *reinterpret_cast<f16x4*>(sl + 4 * i) = l;
      }
    }
  }
  __syncthreads();   // strip complete

  // ---- stage A: s_conv1 on the 5 x 160 tile, by columns of m-tiles (wave, wave + 4 together; wave + 8 for waves 0 and 1) ----
  float va[COLS][R1][4];
  float amax = 0.f;
  {
    const float inv = 1.0f / (s_in * a.sw1);
    const float bv = a.b1[lrow];
    // bias + ReLU; positions outside the s_conv1 image are zero (finite, and out of the tile maximum)
    auto finish = [&](const f32x4& acc, int col, int ry, float (&dst)[4]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 16 * col + 4 * g + r;
        const bool ok = oy1 + ry < a.OH1 && x0 + p < a.OW1;
        const float v = ok ? fmaxf(fmaf(acc[r], inv, bv), 0.0f) : 0.0f;
        dst[r] = v;
        amax = fmaxf(amax, v);
      }
    };
    {
      f32x4 acc[2][R1];
      front_stage_a<2>(sh, sl, wave, lrow, g, wh, wl, acc);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ry = 0; ry < R1; ++ry) finish(acc[c][ry], wave + NWF * c, ry, va[c][ry]);
    }
    if (wave + 2 * NWF < MTRA) {   // wave-uniform
      f32x4 acc[1][R1];
      front_stage_a<1>(sh, sl, wave + 2 * NWF, lrow, g, wh, wl, acc);
#pragma unroll
      for (int ry = 0; ry < R1; ++ry) finish(acc[0][ry], wave + 2 * NWF, ry, va[2][ry]);
    } else {
#pragma unroll
      for (int ry = 0; ry < R1; ++ry)
#pragma unroll
        for (int r = 0; r < 4; ++r) va[2][ry][r] = 0.f;
    }
  }
  // s_conv2's first weight steps travel under the tile exchange below
  const _Float16* wbase2 = a.wp2 + lane * 8;
  f16x8 bq[3][2][2];   // [ring slot][n-tile][hi, lo]
#define FRONT_LOAD_B2(SLOT, KS)                                                                         \
  {                                                                                                     \
    const _Float16* q = wbase2 + (size_t)(KS) * (2 * 2 * 512);                                          \
    bq[SLOT][0][0] = *reinterpret_cast<const f16x8*>(q);                                                \
    bq[SLOT][0][1] = *reinterpret_cast<const f16x8*>(q + 512);                                          \
    bq[SLOT][1][0] = *reinterpret_cast<const f16x8*>(q + 1024);                                         \
    bq[SLOT][1][1] = *reinterpret_cast<const f16x8*>(q + 1536);                                         \
  }
  FRONT_LOAD_B2(0, 0)
  FRONT_LOAD_B2(1, 1)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_down(amax, off, 64));
  if (lane == 0) wg_red[8 + wave] = amax;
  __syncthreads();   // every wave has finished reading the input strip
  amax = wg_red[8];
#pragma unroll
  for (int w = 1; w < NWF; ++w) amax = fmaxf(amax, wg_red[8 + w]);
  const float s_mid = ovn_pow2_scale_for(amax);
  // intermediate tile -> LDS as [row][pixel][16 channels] hi / lo: lane = channel lrow of pixels 4 g .. 4 g + 3 of each m-tile
#pragma unroll
  for (int c = 0; c < COLS; ++c) {
    const int col = wave + NWF * c;
    if (col < MTRA && (!(FRONT_ABL & 32) || amax == 123.456f)) {
#pragma unroll
      for (int ry = 0; ry < R1; ++ry) {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const float x0f = va[c][ry][r] * s_mid, x1f = va[c][ry][r + 1] * s_mid;
          const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0f, x1f));
          const _Float16 l0 = (_Float16)__builtin_fmaf(x0f, one, -(float)hp[0]);
          const _Float16 l1 = (_Float16)__builtin_fmaf(x1f, one, -(float)hp[1]);
          const int o = (ry * PIXM + 16 * col + 4 * g + r) * C1 + lrow;
          mh[o] = hp[0];
          mh[o + C1] = hp[1];
          ml[o] = l0;
          ml[o + C1] = l1;
        }
      }
    }
  }
  __syncthreads();   // tile complete

  // ---- stage B: s_conv2 on the tile.  m-tile t = wave + 4 i: row t / 9, pixels 16 (t % 9) ..; both n-tiles ----
  {
    int aoff[MTHB];
#pragma unroll
    for (int i = 0; i < MTHB; ++i) {
      int t = wave + NWF * i;
      t = t < MTB ? t : 0;
      const int ry = t / MTRB, mt = t - ry * MTRB;
      aoff[i] = (ry * SH2 * PIXM + 16 * mt + lrow) * C1 + 8 * g;
    }
    const bool last = wave + NWF * (MTHB - 1) < MTB;   // wave-uniform: does slot MTHB - 1 hold a tile?
    f32x4 acc[MTHB][2];
#pragma unroll
    for (int i = 0; i < MTHB; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < ((FRONT_ABL & 4) ? 1 : NKB); ++ks) {
      const int ky = ks >> 3, kh = ks & 7;
      const int toff = (ky * PIXM + 2 * kh) * C1;
      if (ks + 2 < NKB) FRONT_LOAD_B2((ks + 2) % 3, ks + 2)
      f16x8 fh[MTHB], fl[MTHB];
#pragma unroll
      for (int i = 0; i < MTHB; ++i) {
        fh[i] = *reinterpret_cast<const f16x8*>(mh + aoff[i] + toff);
        fl[i] = *reinterpret_cast<const f16x8*>(ml + aoff[i] + toff);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int i = 0; i < MTHB; ++i)
          if (i + 1 < MTHB || last) acc[i][j] = FRONT_MFMA(fh[i], bq[ks % 3][j][0], acc[i][j]);
#pragma unroll
        for (int i = 0; i < MTHB; ++i)
          if (i + 1 < MTHB || last) acc[i][j] = FRONT_MFMA(fl[i], bq[ks % 3][j][0], acc[i][j]);
#pragma unroll
        for (int i = 0; i < MTHB; ++i)
          if (i + 1 < MTHB || last) acc[i][j] = FRONT_MFMA(fh[i], bq[ks % 3][j][1], acc[i][j]);
      }
    }
    const float inv = 1.0f / (s_mid * a.sw2);
    float vmax = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = 16 * j + lrow;
      const float bv = a.b2[n];
#pragma unroll
      for (int i = 0; i < MTHB; ++i) {
        const int t = wave + NWF * i;
        const int ry = t / MTRB, mt = t - ry * MTRB;
        float* orow = a.out + (((long long)b * a.OH2 + oy2 + ry) * a.OW2 + x0) * C2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 16 * mt + 4 * g + r;
          if (p < tw && t < MTB && oy2 + ry < a.OH2) {
            const float v = fmaxf(fmaf(acc[i][j][r], inv, bv), 0.0f);
            if (!(FRONT_ABL & 8) || v == 123.456f) orow[(long long)p * C2 + n] = v;
            vmax = fmaxf(vmax, v);
          }
        }
      }
    }
    if (a.out_max) ovn_fold_absmax_wg(vmax, a.out_max + (size_t)b * OVN_ACTMAX_STRIDE, wg_red);   // kernel-uniform condition
  }
}
#undef FRONT_LOAD_B2
#undef FRONT_MFMA

bool is_layer(const OvnConvLayer& L, int kh, int kw, int cin, int cout, int sh, int sw) {
  return L.relu && L.kh == kh && L.kw == kw && L.cin == cin && L.cout == cout && L.sh == sh && L.sw == sw && L.wp_h16 != nullptr;
}

}  // namespace

// True when layers `first`, `first + 1` of the leg are s_conv1 / s_conv2 of the C = 4 network and the input has their geometry.
bool ovn_leg_front_matches(const ovn_ctx* ctx, size_t first, int h, int w) {
  if (first + 1 >= ctx->leg.size()) return false;
  if (!is_layer(ctx->leg[first], KH1, 15, C0, C1, S1, S1) || !is_layer(ctx->leg[first + 1], KH2, 15, C1, C2, SH2, 1)) return false;
  const int oh1 = (h - KH1) / S1 + 1, ow1 = (w - 15) / S1 + 1;
  return h >= KH1 && w >= 15 && oh1 >= KH2 && ow1 >= 15;
}

// images (nb, h, w, 4) -> s_conv2 output (nb, oh2, ow2, 32); out_max: zeroed per-scan words for the maxima of the output, or NULL
int ovn_leg_front_forward(const ovn_ctx* ctx, size_t first, const float* in, int nb, int h, int w, float* out, int* oh_out, int* ow_out,
                          unsigned* out_max, hipStream_t stream) {
  const OvnConvLayer& L1 = ctx->leg[first];
  const OvnConvLayer& L2 = ctx->leg[first + 1];
  FrontArgs a;
  a.in = in;
  a.wp1 = reinterpret_cast<const _Float16*>(L1.wp_h16);
  a.b1 = L1.bias;
  a.wp2 = reinterpret_cast<const _Float16*>(L2.wp_h16);
  a.b2 = L2.bias;
  a.out = out;
  a.out_max = out_max;
  a.sw1 = L1.sw_h;
  a.sw2 = L2.sw_h;
  a.one = 1.0f;
  a.H = h;
  a.W = w;
  a.OH1 = (h - KH1) / S1 + 1;
  a.OW1 = (w - 15) / S1 + 1;
  a.OH2 = (a.OH1 - KH2) / SH2 + 1;
  a.OW2 = a.OW1 - 15 + 1;
  a.XT = (a.OW2 + TW2 - 1) / TW2;
  *oh_out = a.OH2;
  *ow_out = a.OW2;
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(leg_front_kernel), FRONT_LDS);
  if (rc) return rc;
  const long long wgs = (long long)nb * ((a.OH2 + RB - 1) / RB) * a.XT;
  hipLaunchKernelGGL(leg_front_kernel, dim3((unsigned)wgs), dim3(NTHR), FRONT_LDS, stream, a);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
