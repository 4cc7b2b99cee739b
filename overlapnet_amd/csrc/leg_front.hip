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
// stage A: 8 waves x MTHA m-tiles of the 5 x 10 s_conv1 tiles, one n-tile (16 channels)
constexpr int MTRA = PIXM / 16, MTA = R1 * MTRA, MTHA = (MTA + 7) / 8;   // 10, 50, 7
constexpr int NKA = KH1 * 2;                   // 10 K steps: 8 taps x 4 channels each
// stage B: 2 n-tiles x 4 wave rows, 2 x 9 m-tiles
constexpr int MTRB = TW2 / 16, MTB = RB * MTRB, MTHB = (MTB + 3) / 4;    // 9, 18, 5
constexpr int NKB = KH2 * 8;                   // 24 K steps: 2 taps x 16 channels each

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

__global__ __launch_bounds__(512, 4) void leg_front_kernel(FrontArgs a) {   // 4 waves per SIMD = two workgroups per CU: at most 128 VGPRs
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

  // weight fragments travel three K steps ahead of their MFMAs (an L2 round trip is longer than one step); the first steps of
  // stage A are requested before the strip, those of stage B before stage A's epilogue
  f16x8 bq[3][2];
  const _Float16* wbase1 = a.wp1 + lane * 8;
  const _Float16* wbase2 = a.wp2 + (size_t)__builtin_amdgcn_readfirstlane(wave & 1) * (2 * 512) + lane * 8;
#define FRONT_LOAD_B1(SLOT, KS)                                                              \
  {                                                                                          \
    bq[SLOT][0] = *reinterpret_cast<const f16x8*>(wbase1 + (size_t)(KS) * (2 * 512));        \
    bq[SLOT][1] = *reinterpret_cast<const f16x8*>(wbase1 + (size_t)(KS) * (2 * 512) + 512);  \
  }
#define FRONT_LOAD_B2(SLOT, KS)                                                                  \
  {                                                                                              \
    bq[SLOT][0] = *reinterpret_cast<const f16x8*>(wbase2 + (size_t)(KS) * (2 * 2 * 512));        \
    bq[SLOT][1] = *reinterpret_cast<const f16x8*>(wbase2 + (size_t)(KS) * (2 * 2 * 512) + 512);  \
  }
  FRONT_LOAD_B1(0, 0)
  FRONT_LOAD_B1(1, 1)

  // ---- input strip -> LDS, scaled by its own maximum and split once (zero outside the image) ----
  float s_in;
  {
    constexpr int TOTAL = KHS1 * PIXA1;        // one float4 (4 channels) per pixel
    constexpr int ITERS = (TOTAL + 511) / 512; // 9
    f32x4 v[ITERS];
#pragma unroll
    for (int u = 0; u < ITERS; ++u) {
      const int i = tid + u * 512;
      v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < TOTAL) {
        const int row = i / PIXA1, pix = i - row * PIXA1;
        if (pix < pixv && iy0 + row < a.H)
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
    for (int w = 1; w < 8; ++w) m = fmaxf(m, wg_red[w]);
    s_in = ovn_pow2_scale_for(m);
#pragma unroll
    for (int u = 0; u < ITERS; ++u) {
      const int i = tid + u * 512;
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

  // ---- stage A: s_conv1 on the 5 x 160 tile.  m-tile t = 7 wave + i: row t / 10, pixels 16 (t % 10) .. ----
  float va[MTHA][4];
  float amax = 0.f;
  {
    int aoff[MTHA];
#pragma unroll
    for (int i = 0; i < MTHA; ++i) {
      int t = wave * MTHA + i;
      t = t < MTA ? t : 0;
      const int ry = t / MTRA, mt = t - ry * MTRA;
      aoff[i] = (ry * S1 * PIXA1 + S1 * (16 * mt + lrow)) * C0 + 8 * g;
    }
    f32x4 acc[MTHA];
#pragma unroll
    for (int i = 0; i < MTHA; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKA; ++ks) {
      const int ky = ks >> 1, kh = ks & 1;
      const int toff = (ky * PIXA1 + 8 * kh) * C0;
      if (ks + 2 < NKA) FRONT_LOAD_B1((ks + 2) % 3, ks + 2)
      const f16x8 bh = bq[ks % 3][0], bl = bq[ks % 3][1];
      f16x8 fh[MTHA], fl[MTHA];
#pragma unroll
      for (int i = 0; i < MTHA; ++i) {
        fh[i] = *reinterpret_cast<const f16x8*>(sh + aoff[i] + toff);
        fl[i] = *reinterpret_cast<const f16x8*>(sl + aoff[i] + toff);
      }
#pragma unroll
      for (int i = 0; i < MTHA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], bh, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MTHA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[i], bh, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MTHA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], bl, acc[i], 0, 0, 0);
    }
    FRONT_LOAD_B2(0, 0)
    FRONT_LOAD_B2(1, 1)
    // bias + ReLU; positions outside the s_conv1 image are zero (finite, and out of the tile maximum)
    const float inv = 1.0f / (s_in * a.sw1);
    const float bv = a.b1[lrow];
#pragma unroll
    for (int i = 0; i < MTHA; ++i) {
      const int t = wave * MTHA + i;
      const int ry = t / MTRA, mt = t - ry * MTRA;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 16 * mt + 4 * g + r;
        const bool ok = t < MTA && oy1 + ry < a.OH1 && x0 + p < a.OW1;
        const float v = ok ? fmaxf(fmaf(acc[i][r], inv, bv), 0.0f) : 0.0f;
        va[i][r] = v;
        amax = fmaxf(amax, v);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_down(amax, off, 64));
  if (lane == 0) wg_red[8 + wave] = amax;
  __syncthreads();   // every wave has finished reading the input strip
  amax = wg_red[8];
#pragma unroll
  for (int w = 1; w < 8; ++w) amax = fmaxf(amax, wg_red[8 + w]);
  const float s_mid = ovn_pow2_scale_for(amax);
  // intermediate tile -> LDS as [row][pixel][16 channels] hi / lo: lane = channel lrow of pixels 4 g .. 4 g + 3 of each m-tile
#pragma unroll
  for (int i = 0; i < MTHA; ++i) {
    const int t = wave * MTHA + i;
    if (t < MTA) {
      const int ry = t / MTRA, mt = t - ry * MTRA;
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        _Float16 h0, h1, l0, l1;
        const float x0f = va[i][r] * s_mid, x1f = va[i][r + 1] * s_mid;
        const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0f, x1f));
        h0 = hp[0];
        h1 = hp[1];
        l0 = (_Float16)__builtin_fmaf(x0f, one, -(float)hp[0]);
        l1 = (_Float16)__builtin_fmaf(x1f, one, -(float)hp[1]);
        const int o = (ry * PIXM + 16 * mt + 4 * g + r) * C1 + lrow;
        mh[o] = h0;
        mh[o + C1] = h1;
        ml[o] = l0;
        ml[o + C1] = l1;
      }
    }
  }
  __syncthreads();   // tile complete

  // ---- stage B: s_conv2 on the tile.  waves = 2 n-tiles x 4 rows of m-tiles; m-tile t = 5 wm + i: row t / 9, pixels 16 (t % 9) .. ----
  {
    const int wn = wave & 1, wm = wave >> 1;
    int aoff[MTHB];
#pragma unroll
    for (int i = 0; i < MTHB; ++i) {
      int t = wm * MTHB + i;
      t = t < MTB ? t : 0;
      const int ry = t / MTRB, mt = t - ry * MTRB;
      aoff[i] = (ry * SH2 * PIXM + 16 * mt + lrow) * C1 + 8 * g;
    }
    f32x4 acc[MTHB];
#pragma unroll
    for (int i = 0; i < MTHB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKB; ++ks) {
      const int ky = ks >> 3, kh = ks & 7;
      const int toff = (ky * PIXM + 2 * kh) * C1;
      if (ks + 2 < NKB) FRONT_LOAD_B2((ks + 2) % 3, ks + 2)
      const f16x8 bh = bq[ks % 3][0], bl = bq[ks % 3][1];
      f16x8 fh[MTHB], fl[MTHB];
#pragma unroll
      for (int i = 0; i < MTHB; ++i) {
        fh[i] = *reinterpret_cast<const f16x8*>(mh + aoff[i] + toff);
        fl[i] = *reinterpret_cast<const f16x8*>(ml + aoff[i] + toff);
      }
#pragma unroll
      for (int i = 0; i < MTHB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], bh, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MTHB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[i], bh, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MTHB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], bl, acc[i], 0, 0, 0);
    }
    const float inv = 1.0f / (s_mid * a.sw2);
    const int n = 16 * wn + lrow;
    const float bv = a.b2[n];
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < MTHB; ++i) {
      const int t = wm * MTHB + i;
      const int ry = t / MTRB, mt = t - ry * MTRB;
      float* orow = a.out + (((long long)b * a.OH2 + oy2 + ry) * a.OW2 + x0) * C2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 16 * mt + 4 * g + r;
        if (p < tw && t < MTB && oy2 + ry < a.OH2) {
          const float v = fmaxf(fmaf(acc[i][r], inv, bv), 0.0f);
          orow[(long long)p * C2 + n] = v;
          vmax = fmaxf(vmax, v);
        }
      }
    }
    if (a.out_max) ovn_fold_absmax_wg(vmax, a.out_max + (size_t)b * OVN_ACTMAX_STRIDE, wg_red);   // kernel-uniform condition
  }
}

#undef FRONT_LOAD_B1
#undef FRONT_LOAD_B2

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
  hipLaunchKernelGGL(leg_front_kernel, dim3((unsigned)wgs), dim3(512), FRONT_LDS, stream, a);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
