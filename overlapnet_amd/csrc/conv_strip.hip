// KH x KW / stride (SH,1) convolutions + bias + ReLU with the input strip resident in LDS, for gfx950 (f16x3 arithmetic: scaled
// fp16 hi/lo split, see conv_f16x3.hip / delta_head_f16x3.hip).
//
// Reference: the leg layers s_conv3 .. s_conv10 (generateNet.py:173-214): 3x15 and 3x12 with stride (2,1) to 64 channels,
// 2x9 stride (2,1) and the 1xKW layers to 128 channels -- 70 % of the batched leg's time in the generic implicit-GEMM
// kernel (conv_f16x3.hip), which gathers and splits every input element once per tap that touches it (up to 22x) and
// pushes it through the slow LDS store path behind a barrier per 32-deep K chunk.  Here a workgroup owns TW output pixels
// of one output row: the KH input rows it needs ((TW + KW - 1) pixels x CIN channels each) are loaded and split ONCE into
// an LDS-resident hi/lo strip and every tap is a compile-time address offset into it (the K walk is fully unrolled).
// Strip layout: one plane per group of 8 channels, [pixel][8 fp16 = 16 B], planes a multiple of 256 B apart: the 16
// lanes that ds_read_b128 services together (8 lanes of channel group g, 8 of g+1, consecutive pixels) then cover all 64
// banks exactly once (a pixel-major layout with any padding gives 2-way conflicts).  No barrier in the K loop (KH x KW x
// CIN/32 steps of 32 channels, exactly the chunk order of the pre-split weights [kc][nt][hi,lo][lane][8]).  NW (8 or 4) waves = NT
// n-tiles x NW/NT groups of m-tiles; weight fragments straight from L2 three steps deep; A fragments for step s+1 are read
// while step s feeds the matrix pipe.  Same per-accumulator summation order as the generic kernel: bit-identical results.
// Kernels in this file: conv_strip_kernel (one n-tile per wave; calls of a few scans and the 1 x KW layers of non-standard nets),
// conv_strip2_kernel (two n-tiles per wave, ROWS output rows per workgroup: batched s_conv3 / s_conv3a / s_conv4),
// conv_strip_small_kernel (CIN 4 / 16: s_conv1 / s_conv2 when the fused front kernel does not apply).
#include <stdlib.h>

#include <utility>

#include "ovn_internal.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// STRIP_ABL (timing-only builds of tools/experiments/strip_ablate.sh, never the product): 1 no strip staging at all, 2 no K loop,
// 4 staging without the global loads (split + LDS writes of constants)
#ifndef STRIP_ABL
#define STRIP_ABL 0
#endif

namespace {

struct StripArgs {
  const float* in;
  const _Float16* wp;
  const float* bias;
  float* out;
  const unsigned* in_max;   // [scan] float bits of max |input| of every scan of the launch (device)
  unsigned* out_max;        // NULL or [scan]: where max |output| of every scan is folded
  float sw, one;            // weight scale; 1.0f (keeps v_fma_mix selectable, see conv_f16x3.hip)
  int H, W, OH, OW, XT;   // input rows / cols, output rows / cols, x tiles per output row
};

template <int CIN, int KH, int KW, int TW, int NT, int NW = 8>
struct StripCfg {
  static constexpr int PIX = TW + KW - 1;              // input pixels per strip row
  static constexpr int PLANE = (KH * PIX * 8 + 127) / 128 * 128;  // fp16 elements per 8-channel plane (multiple of 256 B)
  static constexpr int NPL = CIN / 8;                  // planes
  static constexpr int MT = (TW + 15) / 16;            // m-tiles per workgroup
  static constexpr int MSPLIT = NW / NT;               // wave groups along M (NW waves = NT n-tiles x MSPLIT)
  static constexpr int MTH = (MT + MSPLIT - 1) / MSPLIT;   // m-tiles per wave
  static constexpr int CC = CIN / 32;                  // 32-channel chunks per tap
  static constexpr int NK = KH * KW * CC;              // K steps
  static constexpr size_t LDS_BYTES = 2 * (size_t)NPL * PLANE * sizeof(_Float16) + 1024;   // + slack for padded-row reads
};

// WPS: waves per SIMD the register budget must allow; with WPS 4 (128 registers) the A fragments are read at their own step instead of
// one step ahead (the other waves of the SIMD cover the LDS round trip)
template <int CIN, int KH, int SH, int KW, int TW, int NT, int NW, int WPS>
__global__ __launch_bounds__(64 * NW, WPS) void conv_strip_kernel(StripArgs a) {
  constexpr int ABUF = WPS >= 4 ? 1 : 2;
  typedef StripCfg<CIN, KH, KW, TW, NT, NW> C;
  static_assert(NW % NT == 0, "waves = n-tiles x groups of m-tiles");
  constexpr int COUT = 16 * NT;
  constexpr int PLANE = C::PLANE, PIX = C::PIX, MTH = C::MTH, CC = C::CC, NK = C::NK;
  extern __shared__ __attribute__((aligned(16))) unsigned char strip_smem[];
  __shared__ float wg_red[16];
  _Float16* sh = reinterpret_cast<_Float16*>(strip_smem);
  _Float16* sl = sh + C::NPL * PLANE;
  const float one = a.one;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lrow = lane & 15;
  const int g = lane >> 4;
  const int wn = wave % NT;    // n-tile
  const int wm = wave / NT;    // group of m-tiles

  int bid = blockIdx.x;
  const int xt = bid % a.XT;
  bid /= a.XT;
  const int oy = bid % a.OH;
  const int b = bid / a.OH;
  const int x0 = xt * TW;
  const int tw = (a.OW - x0 < TW) ? a.OW - x0 : TW;           // valid output pixels of this tile
  const int pixv = (a.W - x0 < PIX) ? a.W - x0 : PIX;         // valid input pixels per strip row
  // power-of-two scale from the largest |input| of THIS scan (left by the producing layer): a scan's result does not depend on
  // what else is in the batch
  const float s_in = ovn_pow2_scale_for(__uint_as_float(a.in_max[(size_t)b * OVN_ACTMAX_STRIDE]));
  const float inv = 1.0f / (s_in * a.sw);

  // weights [kc][nt(NT)][hi,lo][lane][8]: wave-uniform base (scalar) + one per-lane offset
  const _Float16* wbase = a.wp + (size_t)__builtin_amdgcn_readfirstlane(wn) * (2 * 512);
  const int wlane = lane * 8;
  f16x8 bq[3][2];
#define STRIP_LOAD_B(SLOT, KS)                                                     \
  {                                                                                \
    const _Float16* q = wbase + (size_t)(KS) * (NT * 2 * 512);                       \
    bq[SLOT][0] = *reinterpret_cast<const f16x8*>(q + wlane);                     \
    bq[SLOT][1] = *reinterpret_cast<const f16x8*>(q + 512 + wlane);               \
  }
  STRIP_LOAD_B(0, 0)   // requested ahead of the strip: both round trips overlap
  STRIP_LOAD_B(1, 1)

  // ---- strip -> LDS, split once ----
  // Lane <-> element order of the staging: a wave instruction covers 8 consecutive strip pixels x 4 planes (32 channels = one
  // 128-byte line per pixel); lane = (plane pg = lane / 16, pixel pl = (lane / 2) % 8, half h = lane % 2 of the plane's 8 channels).
  // The 16 lanes a ds_write_b64 services together then write 128 CONTIGUOUS bytes of one plane (all 32 banks once).  With
  // consecutive lanes on consecutive channel groups of one pixel (rounds 2-4) they hit 4 / 8 / 16 planes at the same bank offset --
  // planes are a multiple of 256 B apart -- i.e. 4- / 8- / 16-way write conflicts (CIN 32 / 64 / 128) that held the LDS pipe the
  // co-resident workgroups' fragment reads need: 9 / 39 / 24 % on top of the K loops' read cycles for s_conv3 / s_conv3a / s_conv4
  // (profiles/r3_leg_pmc.md: conflict cycles 36-109 % of the LDS-active cycles).  Same values into the same slots: same bits.
  if (!(STRIP_ABL & 1)) {
    constexpr int NQ = KH * PIX;                              // strip pixels (rows are consecutive in a plane)
    constexpr int QG = (NQ + 7) / 8;                          // groups of 8 pixels
    constexpr int PQ = C::NPL / 4;                            // quads of planes
    constexpr int UNITS = QG * PQ;                            // wave instructions of the whole strip
    constexpr int ITERS = (UNITS + NW - 1) / NW;
    static_assert(C::NPL % 4 == 0, "planes are staged four at a time");
    const int h4 = 4 * (lane & 1), pl = (lane >> 1) & 7, pg = lane >> 4;
    // all loads of a batch are issued before the first is consumed (a rolled loop would pay one memory round trip
    // per iteration: the compiler cannot overlap iterations it does not see)
    constexpr int BATCH = 6;
#pragma unroll 1
    for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
      f32x4 v[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int unit = (it0 + u) * NW + wave;
        const int qg = unit / PQ, pq = unit - qg * PQ;
        const int q = 8 * qg + pl;
        const int row = q / PIX, pix = q - row * PIX;
        const int c = (4 * pq + pg) * 8 + h4;
        v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (unit < UNITS && q < NQ) {
          if (STRIP_ABL & 4) v[u] = (f32x4){a.one, a.sw, a.one, a.sw};
          else if (pix < pixv) v[u] = *reinterpret_cast<const f32x4*>(a.in + (((long long)b * a.H + SH * oy + row) * a.W + x0 + pix) * CIN + c);
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int unit = (it0 + u) * NW + wave;
        const int qg = unit / PQ, pq = unit - qg * PQ;
        const int q = 8 * qg + pl;
        if (unit < UNITS && q < NQ) {
          f16x4 h, l;
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const float x0 = v[u][e] * s_in, x1 = v[u][e + 1] * s_in;
            const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
            h[e] = hp[0];
            h[e + 1] = hp[1];
            l[e] = (_Float16)__builtin_fmaf(x0, one, -(float)hp[0]);
            l[e + 1] = (_Float16)__builtin_fmaf(x1, one, -(float)hp[1]);
          }
          const int o = (4 * pq + pg) * PLANE + q * 8 + h4;
          *reinterpret_cast<f16x4*>(sh + o) = h;
          *reinterpret_cast<f16x4*>(sl + o) = l;
        }
      }
    }
  }

  // Fragment addresses: one per-lane base (pixel lrow of the wave's first m-tile, channel group g); m-tile i is 16 pixels
  // = 256 B further, a tap (ky, kx) and a channel chunk are compile-time offsets because the K loop is fully unrolled
  // -> no address arithmetic in the loop.  (Rows of the last, partly padded m-tile read a few pixels past the tile; those
  // accumulators are never stored, and the allocation has slack for the last plane.)
  const _Float16* ah_base = sh + g * PLANE + (16 * wm * MTH + lrow) * 8;
  const _Float16* al_base = sl + g * PLANE + (16 * wm * MTH + lrow) * 8;
  f32x4 acc[MTH];
#pragma unroll
  for (int i = 0; i < MTH; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int LBUF = ABUF;
  f16x8 fh[ABUF][MTH], fl[LBUF][MTH];
#define STRIP_TOFF(KS)                                                             \
    constexpr int tap_ = (KS) / CC;                                                \
    constexpr int ky_ = tap_ / KW;                                                 \
    constexpr int toff_ = (ky_ * PIX + (tap_ - ky_ * KW)) * 8 + 4 * PLANE * ((KS) - tap_ * CC);
#define STRIP_READ_AH(BUF, KS, M)                                                  \
  {                                                                                \
    STRIP_TOFF(KS)                                                                 \
    _Pragma("unroll") for (int i = 0; i < (M); ++i) fh[BUF][i] = *reinterpret_cast<const f16x8*>(ah_base + toff_ + i * 128); \
  }
#define STRIP_READ_AL(BUF, KS, M)                                                  \
  {                                                                                \
    STRIP_TOFF(KS)                                                                 \
    _Pragma("unroll") for (int i = 0; i < (M); ++i) fl[BUF][i] = *reinterpret_cast<const f16x8*>(al_base + toff_ + i * 128); \
  }
#define STRIP_MFMA(BUF, LB, SLOT, M)                                                                           \
  _Pragma("unroll") for (int i = 0; i < (M); ++i)                                                              \
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[BUF][i], bq[SLOT][0], acc[i], 0, 0, 0);             \
  _Pragma("unroll") for (int i = 0; i < (M); ++i)                                                              \
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[LB][i], bq[SLOT][0], acc[i], 0, 0, 0);              \
  _Pragma("unroll") for (int i = 0; i < (M); ++i)                                                              \
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[BUF][i], bq[SLOT][1], acc[i], 0, 0, 0);
  __syncthreads();  // strip complete
  // fully unrolled K walk (compile-time k) over the first M m-tiles of the wave: 3 weight slots x 2 fragment buffers, everything one
  // step (A) / two steps (B) ahead
  auto kwalk = [&]<int M>(std::integral_constant<int, M>) {
    if (ABUF == 2) {
      STRIP_READ_AH(0, 0, M)
      if (LBUF == 2) STRIP_READ_AL(0, 0, M)
    }
    [&]<int... K>(std::integer_sequence<int, K...>) {
      (([&] {
         if constexpr (K + 2 < NK) STRIP_LOAD_B((K + 2) % 3, K + 2)
         if constexpr (ABUF == 2 && LBUF == 2) {
           if constexpr (K + 1 < NK) {
             STRIP_READ_AH((K + 1) & 1, K + 1, M)
             STRIP_READ_AL((K + 1) & 1, K + 1, M)
           }
         } else if constexpr (ABUF == 2) {
           STRIP_READ_AL(0, K, M)
           if constexpr (K + 1 < NK) STRIP_READ_AH((K + 1) & 1, K + 1, M)
         } else {
           STRIP_READ_AH(0, K, M)
           STRIP_READ_AL(0, K, M)
         }
         __builtin_amdgcn_sched_barrier(0);
         STRIP_MFMA(K & (ABUF - 1), K & (LBUF - 1), K % 3, M)
         __builtin_amdgcn_sched_barrier(0);
       }()),
       ...);
    }(std::make_integer_sequence<int, NK>{});
  };
  if (!(STRIP_ABL & 2)) {
    if constexpr (C::MSPLIT == 1) {
      // one wave per n-tile: the LAST tile of a row holds fewer m-tiles than the others -- its K walk is instantiated for that count
      // instead of running empty m-tiles (rounds 3-4 ran 28 m-tiles per s_conv3 / s_conv3a row where 26 are needed: + 8 % / + 11 %
      // issued MFMAs; batched calls now take conv_strip2_kernel below, this serves the narrow tiles of calls of a few scans).  Same
      // order per accumulator: same bits.
      const int mw = (tw + 15) >> 4;   // workgroup-uniform
      [&]<int... Ms>(std::integer_sequence<int, Ms...>) {
        ((mw == MTH - Ms ? (kwalk(std::integral_constant<int, MTH - Ms>{}), 0) : 0), ...);
      }(std::make_integer_sequence<int, MTH>{});
    } else {
      kwalk(std::integral_constant<int, MTH>{});
    }
  }
#undef STRIP_LOAD_B
#undef STRIP_TOFF
#undef STRIP_READ_AH
#undef STRIP_READ_AL
#undef STRIP_MFMA

  // ---- epilogue: bias + ReLU.  C/D layout: lane holds output channel lrow of its n-tile, rows 4g..4g+3 of each m-tile
  const int n = 16 * wn + lrow;
  const float bv = a.bias[n];
  float* orow = a.out + (((long long)b * a.OH + oy) * a.OW + x0) * COUT;
  float vmax = 0.f;
#pragma unroll
  for (int i = 0; i < MTH; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = 16 * (wm * MTH + i) + 4 * g + r;
      if (p < tw) {
        const float v = fmaxf(fmaf(acc[i][r], inv, bv), 0.0f);
        orow[(long long)p * COUT + n] = v;
        vmax = fmaxf(vmax, v);
      }
    }
  }
  if (a.out_max) ovn_fold_absmax_wg(vmax, a.out_max + (size_t)b * OVN_ACTMAX_STRIDE, wg_red);   // wave-uniform condition: every thread calls it
}

// ---- batched calls: ROWS output rows per workgroup, NTW n-tiles per wave ------------------------------------------------------------
// The chip runs these kernels AT ITS POWER CAP (tools/experiments/power_probe.py: 1.33 kW of 1.4 kW at 1.95 GHz under the batched
// leg), so what a kernel costs is the energy of its MFMAs plus the energy of feeding them.  conv_strip_kernel above feeds 3 MFMAs per
// pair of 1-KB LDS fragment reads (one n-tile per wave); the fused tail (2 n-tiles per wave) 6, the contraction kernel of the head 12.
// Here a wave owns NTW n-tiles and ALL m-tiles of ONE of the workgroup's ROWS output rows (waves = NT / NTW x ROWS): an A pair feeds 3
// NTW MFMAs, and no m-tile slot is padded (the earlier two-n-tiles-per-wave builds split the m-tiles of one row over wave groups and
// lost to the padded slots, tools/experiments/README.md round 3).  Consecutive output rows share KH - SH input rows: the strip is
// KH + SH (ROWS - 1) rows instead of KH ROWS.  Staging, scales and the per-accumulator order are those of conv_strip_kernel: same bits.
template <int CIN, int KH, int SH, int KW, int TW, int NT, int NTW, int ROWS>
struct Strip2Cfg {
  static constexpr int PIX = TW + KW - 1;
  static constexpr int KHS = KH + SH * (ROWS - 1);            // strip rows
  static constexpr int PLANE = (KHS * PIX * 8 + 127) / 128 * 128;
  static constexpr int NPL = CIN / 8;
  static constexpr int MT = (TW + 15) / 16;
  static constexpr int NG = NT / NTW;                         // wave groups along N
  static constexpr int NW = NG * ROWS;
  static constexpr int CC = CIN / 32;
  static constexpr int NK = KH * KW * CC;
  static constexpr size_t LDS_BYTES = 2 * (size_t)NPL * PLANE * sizeof(_Float16) + 1024;
  static_assert(NT % NTW == 0 && NPL % 4 == 0, "n-tiles per wave / planes staged four at a time");
};

template <int CIN, int KH, int SH, int KW, int TW, int NT, int NTW, int ROWS, int WPS>
__global__ __launch_bounds__(64 * (NT / NTW) * ROWS, WPS) void conv_strip2_kernel(StripArgs a) {
  typedef Strip2Cfg<CIN, KH, SH, KW, TW, NT, NTW, ROWS> C;
  constexpr int NW = C::NW, NG = C::NG;
  constexpr int COUT = 16 * NT;
  constexpr int PLANE = C::PLANE, PIX = C::PIX, MT = C::MT, CC = C::CC, NK = C::NK, KHS = C::KHS;
  extern __shared__ __attribute__((aligned(16))) unsigned char strip_smem[];
  __shared__ float wg_red[16];
  _Float16* sh = reinterpret_cast<_Float16*>(strip_smem);
  _Float16* sl = sh + C::NPL * PLANE;
  const float one = a.one;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lrow = lane & 15;
  const int g = lane >> 4;
  const int wq = wave % NG;    // group of NTW n-tiles
  const int wr = wave / NG;    // output row of the block

  int bid = blockIdx.x;
  const int xt = bid % a.XT;
  bid /= a.XT;
  const int ohb = (a.OH + ROWS - 1) / ROWS;
  const int oy0 = ROWS * (bid % ohb);
  const int b = bid / ohb;
  const int x0 = xt * TW;
  const int tw = (a.OW - x0 < TW) ? a.OW - x0 : TW;
  const int pixv = (a.W - x0 < PIX) ? a.W - x0 : PIX;
  const float s_in = ovn_pow2_scale_for(__uint_as_float(a.in_max[(size_t)b * OVN_ACTMAX_STRIDE]));
  const float inv = 1.0f / (s_in * a.sw);

  const _Float16* wbase = a.wp + (size_t)__builtin_amdgcn_readfirstlane(wq * NTW) * (2 * 512) + lane * 8;
  f16x8 bq[3][NTW][2];
#define STRIP2_LOAD_B(SLOT, KS)                                                    \
  {                                                                                \
    const _Float16* q = wbase + (size_t)(KS) * (NT * 2 * 512);                      \
    _Pragma("unroll") for (int j = 0; j < NTW; ++j) {                              \
      bq[SLOT][j][0] = *reinterpret_cast<const f16x8*>(q + j * 1024);             \
      bq[SLOT][j][1] = *reinterpret_cast<const f16x8*>(q + j * 1024 + 512);       \
    }                                                                              \
  }
  STRIP2_LOAD_B(0, 0)
  STRIP2_LOAD_B(1, 1)

  // ---- strip -> LDS, split once (8 pixels x 4 planes per wave instruction, see conv_strip_kernel) ----
  {
    constexpr int NQ = KHS * PIX;
    constexpr int QG = (NQ + 7) / 8;
    constexpr int PQ = C::NPL / 4;
    constexpr int UNITS = QG * PQ;
    constexpr int ITERS = (UNITS + NW - 1) / NW;
    const int h4 = 4 * (lane & 1), pl = (lane >> 1) & 7, pg = lane >> 4;
    constexpr int BATCH = 6;
#pragma unroll 1
    for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
      f32x4 v[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int unit = (it0 + u) * NW + wave;
        const int qg = unit / PQ, pq = unit - qg * PQ;
        const int q = 8 * qg + pl;
        const int row = q / PIX, pix = q - row * PIX;
        const int c = (4 * pq + pg) * 8 + h4;
        v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (unit < UNITS && q < NQ && pix < pixv && SH * oy0 + row < a.H)
          v[u] = *reinterpret_cast<const f32x4*>(a.in + (((long long)b * a.H + SH * oy0 + row) * a.W + x0 + pix) * CIN + c);
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int unit = (it0 + u) * NW + wave;
        const int qg = unit / PQ, pq = unit - qg * PQ;
        const int q = 8 * qg + pl;
        if (unit < UNITS && q < NQ) {
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
          const int o = (4 * pq + pg) * PLANE + q * 8 + h4;
          *reinterpret_cast<f16x4*>(sh + o) = h;
          *reinterpret_cast<f16x4*>(sl + o) = l;
        }
      }
    }
  }

  const _Float16* ah_base = sh + g * PLANE + (wr * SH * PIX + lrow) * 8;
  const _Float16* al_base = sl + g * PLANE + (wr * SH * PIX + lrow) * 8;
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 fh[2][MT], fl[1][MT];   // hi fragments one step ahead, lo fragments at their own step (ahead of the next step's hi reads)
#define STRIP2_TOFF(KS)                                                            \
    constexpr int tap_ = (KS) / CC;                                                \
    constexpr int ky_ = tap_ / KW;                                                 \
    constexpr int toff_ = (ky_ * PIX + (tap_ - ky_ * KW)) * 8 + 4 * PLANE * ((KS) - tap_ * CC);
#define STRIP2_READ_AH(BUF, KS, M)                                                 \
  {                                                                                \
    STRIP2_TOFF(KS)                                                                \
    _Pragma("unroll") for (int i = 0; i < (M); ++i) fh[BUF][i] = *reinterpret_cast<const f16x8*>(ah_base + toff_ + i * 128); \
  }
#define STRIP2_READ_AL(KS, M)                                                      \
  {                                                                                \
    STRIP2_TOFF(KS)                                                                \
    _Pragma("unroll") for (int i = 0; i < (M); ++i) fl[0][i] = *reinterpret_cast<const f16x8*>(al_base + toff_ + i * 128); \
  }
  __syncthreads();  // strip complete
  auto kwalk = [&]<int M>(std::integral_constant<int, M>) {
    STRIP2_READ_AH(0, 0, M)
    [&]<int... K>(std::integer_sequence<int, K...>) {
      (([&] {
         if constexpr (K + 2 < NK) STRIP2_LOAD_B((K + 2) % 3, K + 2)
         STRIP2_READ_AL(K, M)
         if constexpr (K + 1 < NK) STRIP2_READ_AH((K + 1) & 1, K + 1, M)
         __builtin_amdgcn_sched_barrier(0);
#pragma unroll
         for (int j = 0; j < NTW; ++j) {
#pragma unroll
           for (int i = 0; i < M; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[K & 1][i], bq[K % 3][j][0], acc[i][j], 0, 0, 0);
#pragma unroll
           for (int i = 0; i < M; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[0][i], bq[K % 3][j][0], acc[i][j], 0, 0, 0);
#pragma unroll
           for (int i = 0; i < M; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[K & 1][i], bq[K % 3][j][1], acc[i][j], 0, 0, 0);
         }
         __builtin_amdgcn_sched_barrier(0);
       }()),
       ...);
    }(std::make_integer_sequence<int, NK>{});
  };
  {
    const int mw = (tw + 15) >> 4;   // m-tiles this tile really holds (workgroup-uniform)
    [&]<int... Ms>(std::integer_sequence<int, Ms...>) {
      ((mw == MT - Ms ? (kwalk(std::integral_constant<int, MT - Ms>{}), 0) : 0), ...);
    }(std::make_integer_sequence<int, MT>{});
  }
#undef STRIP2_LOAD_B
#undef STRIP2_TOFF
#undef STRIP2_READ_AH
#undef STRIP2_READ_AL

  // ---- epilogue: bias + ReLU; lane holds output channel lrow of each of its n-tiles, rows 4g..4g+3 of each m-tile
  const int oy = oy0 + wr;
  float vmax = 0.f;
  if (oy < a.OH) {
    float* orow = a.out + (((long long)b * a.OH + oy) * a.OW + x0) * COUT;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = 16 * (wq * NTW + j) + lrow;
      const float bv = a.bias[n];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 16 * i + 4 * g + r;
          if (p < tw) {
            const float v = fmaxf(fmaf(acc[i][j][r], inv, bv), 0.0f);
            orow[(long long)p * COUT + n] = v;
            vmax = fmaxf(vmax, v);
          }
        }
      }
    }
  }
  if (a.out_max) ovn_fold_absmax_wg(vmax, a.out_max + (size_t)b * OVN_ACTMAX_STRIDE, wg_red);   // kernel-uniform condition: every thread calls it
}

// ---- few input channels (s_conv1: 4, s_conv2: 16) --------------------------------------------------------------------------
// With CIN < 32 an MFMA K step of 32 spans TPS = 32 / CIN consecutive taps kx of one kernel row (KW is padded to 16 taps with
// zero weights: s_conv1 5 x 15 x 4 -> 10 steps, s_conv2 3 x 15 x 16 -> 24 steps), and the strip is kept PIXEL-major in LDS
// ([row][pixel][CIN] fp16, hi and lo): the 8 consecutive k of lane group g are then 16 contiguous bytes at pixel
// SW * m + TPS * kh + 8 g / CIN, channel 8 g % CIN -- the 64 lanes of a fragment read touch 19 (CIN 4, SW 2) / 34 (CIN 16)
// consecutive 16-byte slots, conflict-free.  The generic implicit-GEMM kernel re-gathers and re-splits every input element for
// each of the ~19 (s_conv1) / ~22 (s_conv2) output positions that use it and, with only 1 / 2 n-tiles to amortise a split over,
// ran these two layers at 3 % of the MFMA rate: 46 % of the batched leg's time.
template <int CIN, int KH, int SH, int KW, int SW, int TW, int NT, int ROWS>
struct SmallCfg {
  static constexpr int TPS = 32 / CIN;                       // taps per K step
  static constexpr int KSR = 16 / TPS;                       // K steps per kernel row (taps padded to 16)
  static constexpr int NK = KH * KSR;
  static constexpr int PIXA = SW * (TW - 1) + 16 + 8;        // strip pixels per row: last m-tile's last pixel + 16 taps (+ slack)
  static constexpr int KHS = KH + SH * (ROWS - 1);           // strip rows: ROWS output rows share KH - SH input rows with their neighbour
  static constexpr int MTR = TW / 16;                        // m-tiles per output row
  static constexpr int MT = ROWS * MTR;
  static constexpr int MSPLIT = 8 / NT;
  static constexpr int MTH = (MT + MSPLIT - 1) / MSPLIT;
  static constexpr size_t LDS_BYTES = 2 * (size_t)KHS * PIXA * CIN * sizeof(_Float16);
  static_assert(TW % 16 == 0 && 32 % CIN == 0 && CIN % 4 == 0 && KW <= 16, "tile / channel constraints");
};

// ROWS consecutive output rows per workgroup: with stride SH < KH they share input rows, so the strip (the part of these
// kernels that is not hidden: 52 MFMAs per wave against a 38 KB strip at ROWS = 1) is KH + SH (ROWS - 1) rows instead of KH ROWS.
// s_conv1: 3 rows per block (9 strip rows for 15), s_conv2: 2 (5 for 6): batched leg 6.50 -> 6.05 ms per 1025 scans, almost all
// of it from s_conv2; same K order per accumulator as ROWS = 1, bit-identical results.
// OWN: the input scale comes from the strip's own largest |value| (all of the strip is in registers before it is split anyway)
// instead of the call-wide maximum in *a.in_max -- the first layer then needs no absmax pass over the images (0.33 ms per 1025
// scans), and a scan's result does not depend on the other scans of the call.
template <int CIN, int KH, int SH, int KW, int SW, int TW, int NT, int ROWS, bool OWN>
__global__ __launch_bounds__(512) void conv_strip_small_kernel(StripArgs a) {
  typedef SmallCfg<CIN, KH, SH, KW, SW, TW, NT, ROWS> C;
  constexpr int COUT = 16 * NT;
  constexpr int PIXA = C::PIXA, MTH = C::MTH, NK = C::NK, TPS = C::TPS, KSR = C::KSR, KHS = C::KHS, MTR = C::MTR;
  extern __shared__ __attribute__((aligned(16))) unsigned char strip_smem[];
  __shared__ float wg_red[16];
  _Float16* sh = reinterpret_cast<_Float16*>(strip_smem);
  _Float16* sl = sh + KHS * PIXA * CIN;
  float s_in = 1.0f;   // OWN: from the strip itself, below; else from the scan's input maximum once the scan index is known
  const float one = a.one;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lrow = lane & 15;
  const int g = lane >> 4;
  const int wn = wave % NT;
  const int wm = wave / NT;

  int bid = blockIdx.x;
  const int xt = bid % a.XT;
  bid /= a.XT;
  const int ohb = (a.OH + ROWS - 1) / ROWS;                    // row blocks per image
  const int oy = ROWS * (bid % ohb);                           // first output row of the block
  const int b = bid / ohb;
  if (!OWN) s_in = ovn_pow2_scale_for(__uint_as_float(a.in_max[(size_t)b * OVN_ACTMAX_STRIDE]));
  const int x0 = xt * TW;                                      // first output pixel of the tile
  const int tw = (a.OW - x0 < TW) ? a.OW - x0 : TW;
  const int px0 = SW * x0;                                     // first input pixel of the strip
  const int pixv = (a.W - px0 < PIXA) ? a.W - px0 : PIXA;      // valid input pixels per strip row

  // ---- strip -> LDS, scaled and split once ----
  {
    constexpr int Q = CIN / 4;
    constexpr int TOTAL = KHS * PIXA * Q;
    constexpr int ITERS = (TOTAL + 511) / 512;
    constexpr int BATCH = OWN ? ITERS : 4;   // OWN: the whole strip in one batch (its maximum is needed before the first split)
    static_assert(!OWN || ITERS <= 12, "strip too large to hold in registers");
#pragma unroll 1
    for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
      f32x4 v[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int i = tid + (it0 + u) * 512;
        v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (i < TOTAL) {
          const int row = i / (PIXA * Q);
          const int r = i - row * (PIXA * Q);
          const int pix = r / Q;
          const int c = 4 * (r - pix * Q);
          if (pix < pixv && SH * oy + row < a.H)   // rows past the image belong to output rows past OH (odd row counts)
            v[u] = *reinterpret_cast<const f32x4*>(a.in + (((long long)b * a.H + SH * oy + row) * a.W + px0 + pix) * CIN + c);
        }
      }
      if (OWN) {   // workgroup maximum through the (not yet written) start of the lo image
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < BATCH; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
        float* red = reinterpret_cast<float*>(sl);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = red[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
        __syncthreads();   // everyone has read the scratch before the lo image overwrites it
        s_in = ovn_pow2_scale_for(m);
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int i = tid + (it0 + u) * 512;
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
          *reinterpret_cast<f16x4*>(sh + 4 * i) = h;           // [row][pix][CIN] is exactly the linear order of i
          *reinterpret_cast<f16x4*>(sl + 4 * i) = l;
        }
      }
    }
  }
  const float inv = 1.0f / (s_in * a.sw);

  // per-lane fragment base: output pixel lrow of the wave's first m-tile, k offset of lane group g
  const int goff = ((8 * g) / CIN) * CIN + (8 * g) % CIN;       // = 8 g: pixel-major makes (pixel, channel) linear
  // m-tile t = wm MTH + i covers output row t / MTR of the block, pixels 16 (t % MTR) ..; tiles past the block re-read tile 0
  int aoff[MTH];
#pragma unroll
  for (int i = 0; i < MTH; ++i) {
    int t = wm * MTH + i;
    t = t < C::MT ? t : 0;
    const int ry = t / MTR, mt = t - ry * MTR;
    aoff[i] = (ry * SH * PIXA + SW * (16 * mt + lrow)) * CIN + goff;
  }
  f32x4 acc[MTH];
#pragma unroll
  for (int i = 0; i < MTH; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const _Float16* wbase = a.wp + (size_t)__builtin_amdgcn_readfirstlane(wn) * (2 * 512) + lane * 8;
  __syncthreads();  // strip complete
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    const int ky = ks / KSR, kh = ks - ky * KSR;
    const int toff = (ky * PIXA + TPS * kh) * CIN;
    const f16x8 bh = *reinterpret_cast<const f16x8*>(wbase + (size_t)ks * (NT * 2 * 512));
    const f16x8 bl = *reinterpret_cast<const f16x8*>(wbase + (size_t)ks * (NT * 2 * 512) + 512);
    f16x8 fh[MTH], fl[MTH];
#pragma unroll
    for (int i = 0; i < MTH; ++i) {
      fh[i] = *reinterpret_cast<const f16x8*>(sh + aoff[i] + toff);
      fl[i] = *reinterpret_cast<const f16x8*>(sl + aoff[i] + toff);
    }
#pragma unroll
    for (int i = 0; i < MTH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], bh, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MTH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[i], bh, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MTH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], bl, acc[i], 0, 0, 0);
  }

  const int n = 16 * wn + lrow;
  const float bv = a.bias[n];
  float vmax = 0.f;
#pragma unroll
  for (int i = 0; i < MTH; ++i) {
    const int t = wm * MTH + i;
    const int ry = t / MTR, mt = t - ry * MTR;
    float* orow = a.out + (((long long)b * a.OH + oy + ry) * a.OW + x0) * COUT;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = 16 * mt + 4 * g + r;
      if (p < tw && t < C::MT && oy + ry < a.OH) {
        const float v = fmaxf(fmaf(acc[i][r], inv, bv), 0.0f);
        orow[(long long)p * COUT + n] = v;
        vmax = fmaxf(vmax, v);
      }
    }
  }
  if (a.out_max) ovn_fold_absmax_wg(vmax, a.out_max + (size_t)b * OVN_ACTMAX_STRIDE, wg_red);   // wave-uniform condition: every thread calls it
}

template <int CIN, int KH, int SH, int KW, int SW, int TW, int NT, int ROWS, bool OWN>
int launch_strip_small(const OvnConvLayer& L, const float* in, int nb, long long call_nb, int h, int w, float* out, const unsigned* in_max,
                       unsigned* out_max, hipStream_t stream, bool* took) {
  typedef SmallCfg<CIN, KH, SH, KW, SW, TW, NT, ROWS> C;
  StripArgs a;
  a.in = in;
  a.wp = reinterpret_cast<const _Float16*>(L.wp_h16);
  a.bias = L.bias;
  a.out = out;
  a.in_max = in_max;
  a.out_max = out_max;
  a.sw = L.sw_h;
  a.one = 1.0f;
  a.H = h;
  a.W = w;
  a.OH = (h - KH) / SH + 1;
  a.OW = (w - KW) / SW + 1;
  a.XT = (a.OW + TW - 1) / TW;
  const long long wgs = (long long)nb * ((a.OH + ROWS - 1) / ROWS) * a.XT;
  *took = true;   // every call size takes this kernel: the arithmetic of a scan must not depend on the size of its batch
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(conv_strip_small_kernel<CIN, KH, SH, KW, SW, TW, NT, ROWS, OWN>), C::LDS_BYTES);
  if (rc) return rc;
  hipLaunchKernelGGL((conv_strip_small_kernel<CIN, KH, SH, KW, SW, TW, NT, ROWS, OWN>), dim3((unsigned)wgs), dim3(512), C::LDS_BYTES, stream, a);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

template <int CIN, int KH, int SH, int KW, int TW, int NT, int NW = 8, int WPS = 2>
int launch_strip(const OvnConvLayer& L, const float* in, int nb, long long call_nb, int h, int w, float* out, const unsigned* in_max,
                 unsigned* out_max, hipStream_t stream, bool* took) {
  typedef StripCfg<CIN, KH, KW, TW, NT, NW> C;
  StripArgs a;
  a.in = in;
  a.wp = reinterpret_cast<const _Float16*>(L.wp_h);
  a.bias = L.bias;
  a.out = out;
  a.in_max = in_max;
  a.out_max = out_max;
  a.sw = L.sw_h;
  a.one = 1.0f;
  a.H = h;
  a.W = w;
  a.OH = (h - KH) / SH + 1;
  a.OW = w - KW + 1;
  a.XT = (a.OW + TW - 1) / TW;
  const long long wgs = (long long)nb * a.OH * a.XT;
  *took = true;   // every call size takes this kernel (one scan: OH x XT workgroups, still faster than the generic kernel's serial K walk)
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(conv_strip_kernel<CIN, KH, SH, KW, TW, NT, NW, WPS>), C::LDS_BYTES);
  if (rc) return rc;
  hipLaunchKernelGGL((conv_strip_kernel<CIN, KH, SH, KW, TW, NT, NW, WPS>), dim3((unsigned)wgs), dim3(64 * NW), C::LDS_BYTES, stream, a);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

template <int CIN, int KH, int SH, int KW, int TW, int NT, int NTW, int ROWS, int WPS>
int launch_strip2(const OvnConvLayer& L, const float* in, int nb, int h, int w, float* out, const unsigned* in_max, unsigned* out_max,
                  hipStream_t stream, bool* took) {
  typedef Strip2Cfg<CIN, KH, SH, KW, TW, NT, NTW, ROWS> C;
  StripArgs a;
  a.in = in;
  a.wp = reinterpret_cast<const _Float16*>(L.wp_h);
  a.bias = L.bias;
  a.out = out;
  a.in_max = in_max;
  a.out_max = out_max;
  a.sw = L.sw_h;
  a.one = 1.0f;
  a.H = h;
  a.W = w;
  a.OH = (h - KH) / SH + 1;
  a.OW = w - KW + 1;
  a.XT = (a.OW + TW - 1) / TW;
  const long long wgs = (long long)nb * ((a.OH + ROWS - 1) / ROWS) * a.XT;
  *took = true;
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(conv_strip2_kernel<CIN, KH, SH, KW, TW, NT, NTW, ROWS, WPS>), C::LDS_BYTES);
  if (rc) return rc;
  hipLaunchKernelGGL((conv_strip2_kernel<CIN, KH, SH, KW, TW, NT, NTW, ROWS, WPS>), dim3((unsigned)wgs), dim3(64 * C::NW), C::LDS_BYTES, stream, a);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

}  // namespace

constexpr int SMALL_NB = 4;   // calls of at most this many scans take the narrow-tile instantiations below
static int pad_rows(int ow, int tw) { return (ow + tw - 1) / tw * tw - ow; }   // padded output pixels per row with tiles of tw

// True when ovn_conv_strip_try will run this layer with a kernel that scales its input by the strip's own maximum (s_conv1 at
// C = 4 in batched calls): the caller then skips the absmax pass over the layer's input.
bool ovn_conv_strip_own_scale(const OvnConvLayer& L, long long call_nb, int h, int w) {
  if (!(L.relu && L.wp_h != nullptr && L.wp_h16 != nullptr) || h < L.kh || w < L.kw) return false;
  if (!(L.kh == 5 && L.kw == 15 && L.cin == 4 && L.cout == 16 && L.sh == 2 && L.sw == 2)) return false;
  (void)call_nb;   // the choice must not depend on the size of the call
  return true;
}

// Returns 1 when the layer / call was taken (result in out), 0 when the caller should use the generic kernel, < 0 on error.
int ovn_conv_strip_try(const OvnConvLayer& L, const float* in, int nb, long long call_nb, int h, int w, float* out,
                       const unsigned* in_max, unsigned* out_max, hipStream_t stream) {
  if (!(L.relu && L.wp_h != nullptr) || (reinterpret_cast<uintptr_t>(in) & 15) != 0) return 0;
  if (h < L.kh || w < L.kw) return 0;
  bool took = false;
  int rc = OVN_OK;
  if (L.wp_h16 != nullptr) {   // few input channels: pixel-major strips, taps padded to 16
    if (L.kh == 5 && L.kw == 15 && L.cin == 4 && L.cout == 16 && L.sh == 2 && L.sw == 2)          // s_conv1 at C = 4
      rc = launch_strip_small<4, 5, 2, 15, 2, 224, 1, 3, true>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took);
    else if (L.kh == 3 && L.kw == 15 && L.cin == 16 && L.cout == 32 && L.sh == 2 && L.sw == 1)    // s_conv2
      rc = launch_strip_small<16, 3, 2, 15, 1, 144, 2, 2, false>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took);
    else
      return 0;
    if (rc) return -rc;
    return took ? 1 : 0;
  }
  if (L.sw != 1) return 0;
  const int key = ((L.kh * 100 + L.kw) * 1000 + L.cin) * 1000 + L.cout;   // kh, kw, cin, cout
  if (L.kh > 1 && L.sh != 2) return 0;
  if (L.kh == 1 && L.sh != 1) return 0;
  switch (key) {
    // s_conv3 / s_conv3a / s_conv4, batched calls: conv_strip2_kernel -- four waves, each TWO n-tiles and all m-tiles of one of the
    // workgroup's output rows (an LDS fragment pair feeds 6 MFMAs instead of 3; the chip runs these kernels at its power cap, what
    // counts is the energy per MFMA incl. its operand traffic): batched leg 4.70 -> 4.58 ms per 1025 scans on one box, same bits.
    // A call of a few scans -- the query of a loop-closure step -- takes conv_strip_kernel with narrow tiles (one n-tile per wave,
    // more workgroups, each with less to do): these kernels scale by the SCAN's maximum, so neither the tile width nor the kernel
    // changes a bit
    case ((3 * 100 + 15) * 1000 + 32) * 1000 + 64:   // s_conv3
      // 96-pixel tiles: 6 + 6 + 6 + 6 + 2 = the 26 m-tiles a 415-pixel row needs; two rows per workgroup share a 5-row strip (72 KB: two per CU)
      rc = (nb <= SMALL_NB) ? launch_strip<32, 3, 2, 15, 52, 4, 4>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took)
                            : launch_strip2<32, 3, 2, 15, 96, 4, 2, 2, 2>(L, in, nb, h, w, out, in_max, out_max, stream, &took);
      break;
    case ((3 * 100 + 12) * 1000 + 64) * 1000 + 64:   // s_conv3a
      // 48-pixel tiles: 8 x 3 + 2 = the 26 m-tiles of a 404-pixel row; both output rows in one workgroup (5-row strip, 79 KB: two per CU)
      rc = (nb <= SMALL_NB) ? launch_strip<64, 3, 2, 12, 32, 4, 4>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took)
                            : launch_strip2<64, 3, 2, 12, 48, 4, 2, 2, 2>(L, in, nb, h, w, out, in_max, out_max, stream, &took);
      break;
    // the 128-channel layers: tiles of 80 or 96 pixels (5 / 6 exact m-tiles), whichever wastes fewer padded rows of the row
    case ((2 * 100 + 9) * 1000 + 64) * 1000 + 128:    // s_conv4
      // batched: four waves x two n-tiles, all 5 m-tiles of an 80-pixel tile each (396 = 5 x 80 - 4: 25 m-tiles, none padded)
      rc = (nb <= SMALL_NB) ? launch_strip<64, 2, 2, 9, 32, 8, 8, 4>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took)
                            : launch_strip2<64, 2, 2, 9, 80, 8, 2, 1, 2>(L, in, nb, h, w, out, in_max, out_max, stream, &took);
      break;
    case ((1 * 100 + 9) * 1000 + 128) * 1000 + 128:   // s_conv5-7
      rc = (pad_rows(w - 9 + 1, 80) <= pad_rows(w - 9 + 1, 96)) ? launch_strip<128, 1, 1, 9, 80, 8>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took)
                                                                 : launch_strip<128, 1, 1, 9, 96, 8>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took);
      break;
    case ((1 * 100 + 7) * 1000 + 128) * 1000 + 128:   // s_conv8
      rc = (pad_rows(w - 7 + 1, 80) <= pad_rows(w - 7 + 1, 96)) ? launch_strip<128, 1, 1, 7, 80, 8>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took)
                                                                 : launch_strip<128, 1, 1, 7, 96, 8>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took);
      break;
    case ((1 * 100 + 5) * 1000 + 128) * 1000 + 128:   // s_conv9
      rc = (pad_rows(w - 5 + 1, 80) <= pad_rows(w - 5 + 1, 96)) ? launch_strip<128, 1, 1, 5, 80, 8>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took)
                                                                 : launch_strip<128, 1, 1, 5, 96, 8>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took);
      break;
    case ((1 * 100 + 3) * 1000 + 128) * 1000 + 128:   // s_conv10
      rc = (pad_rows(w - 3 + 1, 80) <= pad_rows(w - 3 + 1, 96)) ? launch_strip<128, 1, 1, 3, 80, 8>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took)
                                                                 : launch_strip<128, 1, 1, 3, 96, 8>(L, in, nb, call_nb, h, w, out, in_max, out_max, stream, &took);
      break;
    default: return 0;
  }
  if (rc) return -rc;
  return took ? 1 : 0;
}
