// The last six leg layers (s_conv5 .. s_conv10: 1 x {9,9,9,7,5,3} convolutions 128 -> 128 + bias + ReLU over the single remaining
// image row, generateNet.py:189-214) fused into ONE kernel for batched calls, f16x3 arithmetic (scaled fp16 hi/lo split on the
// fp16 matrix cores, see conv_f16x3.hip / delta_head_f16x3.hip), for gfx950.
//
// As separate strip kernels (conv_strip.hip) these layers are 30 % of the batched leg: each one stages its input strip from HBM,
// runs a 12..36-step K loop in which every wave owns one n-tile and re-reads ALL A fragments from LDS (8 x redundant: LDS-bound,
// matrix pipe 32 % busy), and writes fp32 activations back.  Here a workgroup owns 90 final output pixels of one scan
// (360 = 4 x 90) and carries its 126-pixel input strip through all six layers in LDS (one strip buffer, rewritten in place), one
// [pixel][8 fp16] plane per group of 8 channels for hi and for lo (2304 B = 9 x 256 B per plane, see conv_strip.hip for why),
// taps are address offsets, the K walk is fully unrolled.  A wave owns TWO n-tiles and every other m-tile (4 waves along N x 2
// along M): half the LDS fragment reads per MFMA.  Between layers the accumulators get bias + ReLU, the workgroup's largest
// value gives the next power-of-two scale (a tile's result never depends on the other scans of the call), and the hi/lo
// halves go straight into the other strip buffer; only the last layer writes fp32 to HBM.  Halo recompute: 608 instead of 557
// output pixels per scan (+9 %).  Weight fragments [tap * 4 + chunk][n-tile][hi,lo][lane][8] straight from L2, one K step
// ahead (the next layer's first step under the epilogue of the current one); same per-accumulator summation order as the per-layer kernels (tap-major, chunk, term).
#include <utility>

#include "ovn_internal.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NLAY = 6;
constexpr int CH = 128;
constexpr int TOUT = 90;                 // final output pixels per workgroup
constexpr int XT = OVN_FEAT_W / TOUT;    // 4 workgroups per scan
constexpr int PIXP = 144;                // pixels per plane: 126 input pixels + what the padded rows of the last m-tile read
constexpr int PLANE = PIXP * 8;          // fp16 elements per 8-channel plane: 2304 B = 9 x 256 B
constexpr int NPL = CH / 8;              // 16 planes
constexpr int HALF = NPL * PLANE;        // elements of the hi (or lo) image of one strip buffer
constexpr size_t TAIL_LDS = 2 * (size_t)HALF * sizeof(_Float16) + 64;   // one strip buffer (hi + lo), rewritten in place, + reduction scratch: 73,792 B

struct TailArgs {
  const float* in;    // (nb, 1, win, 128) fp32: output of the layer before the tail
  float* out;         // (nb, 1, 360, 128) fp32
  const _Float16* wp[NLAY];
  const float* bias[NLAY];
  float sw[NLAY];
  float one;          // 1.0f (keeps v_fma_mix selectable, see conv_f16x3.hip)
  int win;            // input width (396)
};

__device__ __forceinline__ void split2(float x0, float x1, float one, _Float16& h0, _Float16& h1, _Float16& l0, _Float16& l1) {
  const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  h0 = hp[0];
  h1 = hp[1];
  l0 = (_Float16)__builtin_fmaf(x0, one, -(float)hp[0]);
  l1 = (_Float16)__builtin_fmaf(x1, one, -(float)hp[1]);
}

// K steps 0 .. RING - 2 of a layer's weights -> the ring slots of the same numbers (n-tiles 2 wn, 2 wn + 1 of this wave)
template <int RING>
__device__ __forceinline__ void tail_preload(f16x8 (&bq)[RING][2][2], const _Float16* __restrict__ wp) {
  const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 3;
  const _Float16* wbase = wp + (size_t)(2 * wn) * (2 * 512) + lane * 8;
#pragma unroll
  for (int ks = 0; ks < RING - 1; ++ks) {
    const _Float16* q = wbase + (size_t)ks * (8 * 2 * 512);
    bq[ks][0][0] = *reinterpret_cast<const f16x8*>(q);
    bq[ks][0][1] = *reinterpret_cast<const f16x8*>(q + 512);
    bq[ks][1][0] = *reinterpret_cast<const f16x8*>(q + 1024);
    bq[ks][1][1] = *reinterpret_cast<const f16x8*>(q + 1536);
  }
}

// One layer: in strip (ih, il; scaled by s_in) -> out strip (oh, ol; scaled by the returned scale) or, LAST, fp32 rows in HBM.
template <int KW, int WOUT, bool LAST, int RING, int ABUF>
__device__ __forceinline__ float tail_layer(const _Float16* ih, const _Float16* il, _Float16* oh,   // oh / ol may be ih / il (in place)
                                            _Float16* ol, float* __restrict__ red, const _Float16* __restrict__ wp,
                                            const float* __restrict__ bias, float s_in, float sw, float one, float* __restrict__ gout,
                                            const _Float16* __restrict__ wp_next, f16x8 (&bq)[RING][2][2]) {
  constexpr int MT = (WOUT + 15) / 16;   // m-tiles of this layer's output
  constexpr int MTW = (MT + 1) / 2;      // per wave: m-tiles wm, wm + 2, ...
  constexpr int NK = KW * 4;             // K steps: tap-major, 4 chunks of 32 channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, g = lane >> 4;
  const int wn = wave & 3, wm = wave >> 2;
  const _Float16* ah_base = ih + g * PLANE + (16 * wm + lrow) * 8;
  const _Float16* al_base = il + g * PLANE + (16 * wm + lrow) * 8;
  f32x4 acc[MTW][2];
#pragma unroll
  for (int i = 0; i < MTW; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const _Float16* wbase = wp + (size_t)(2 * wn) * (2 * 512) + lane * 8;   // n-tiles 2 wn, 2 wn + 1
  // bq: [ring slot][n-tile][hi, lo]; slots 0 .. RING - 2 already hold this layer's first K steps (loaded by the previous layer / the
  // kernel prologue).  ABUF 2: A fragments of step s + 1 are read while step s feeds the matrix pipe; 1: read at their own step (the
  // two-workgroups-per-CU build, 128 registers per lane: the other waves of the SIMD cover the LDS round trip)
  f16x8 fh[ABUF][MTW], fl[ABUF][MTW];
#define TAIL_LOAD_B(SLOT, KS)                                                         \
  {                                                                                   \
    const _Float16* q = wbase + (size_t)(KS) * (8 * 2 * 512);                         \
    bq[SLOT][0][0] = *reinterpret_cast<const f16x8*>(q);                              \
    bq[SLOT][0][1] = *reinterpret_cast<const f16x8*>(q + 512);                        \
    bq[SLOT][1][0] = *reinterpret_cast<const f16x8*>(q + 1024);                       \
    bq[SLOT][1][1] = *reinterpret_cast<const f16x8*>(q + 1536);                       \
  }
#define TAIL_READ_A(BUF, KS)                                                          \
  {                                                                                   \
    constexpr int toff_ = ((KS) >> 2) * 8 + 4 * PLANE * ((KS) & 3);                   \
    _Pragma("unroll") for (int i = 0; i < MTW; ++i) {                                 \
      fh[BUF][i] = *reinterpret_cast<const f16x8*>(ah_base + toff_ + i * 256);        \
      fl[BUF][i] = *reinterpret_cast<const f16x8*>(al_base + toff_ + i * 256);        \
    }                                                                                 \
  }
  if (ABUF == 2) TAIL_READ_A(0, 0)
  [&]<int... K>(std::integer_sequence<int, K...>) {
    (([&] {
       if constexpr (K + RING - 1 < NK) TAIL_LOAD_B((K + RING - 1) % RING, K + RING - 1)
       if constexpr (ABUF == 2) {
         if constexpr (K + 1 < NK) TAIL_READ_A((K + 1) & 1, K + 1)
       } else {
         TAIL_READ_A(0, K)
       }
       __builtin_amdgcn_sched_barrier(0);
#pragma unroll
       for (int j = 0; j < 2; ++j) {
#pragma unroll
         for (int i = 0; i < MTW; ++i)
           if (wm + 2 * i < MT)
             acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[K & (ABUF - 1)][i], bq[K % RING][j][0], acc[i][j], 0, 0, 0);
#pragma unroll
         for (int i = 0; i < MTW; ++i)
           if (wm + 2 * i < MT)
             acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[K & (ABUF - 1)][i], bq[K % RING][j][0], acc[i][j], 0, 0, 0);
#pragma unroll
         for (int i = 0; i < MTW; ++i)
           if (wm + 2 * i < MT)
             acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[K & (ABUF - 1)][i], bq[K % RING][j][1], acc[i][j], 0, 0, 0);
       }
       __builtin_amdgcn_sched_barrier(0);
     }()),
     ...);
  }(std::make_integer_sequence<int, NK>{});
#undef TAIL_LOAD_B
#undef TAIL_READ_A
  // the next layer's first four weight steps travel while this layer's epilogue runs (its barriers would otherwise be followed by an
  // exposed L2 round trip: six of them per workgroup)
  if (!LAST) tail_preload<RING>(bq, wp_next);

  // C/D layout: lane holds channel n = 16 (2 wn + j) + lrow, pixels 16 (wm + 2 i) + 4 g + r
  const float inv = 1.0f / (s_in * sw);
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bv[j] = bias[16 * (2 * wn + j) + lrow];
  float vmax = 0.f;
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 16 * (wm + 2 * i) + 4 * g + r;
        const float v = fmaxf(fmaf(acc[i][j][r], inv, bv[j]), 0.0f);
        acc[i][j][r] = v;
        if (p < WOUT) vmax = fmaxf(vmax, v);
      }
  if (LAST) {
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 16 * (wm + 2 * i) + 4 * g + r;
        if (p < WOUT) {
#pragma unroll
          for (int j = 0; j < 2; ++j) gout[(size_t)p * CH + 16 * (2 * wn + j) + lrow] = acc[i][j][r];
        }
      }
    return 1.0f;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
  if (lane == 0) red[wave] = vmax;
  __syncthreads();   // every wave is done reading the input strip and has published its maximum
  float m = red[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
  const float s_out = ovn_pow2_scale_for(m);
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = 16 * (2 * wn + j) + lrow;
      _Float16* dh = oh + (n >> 3) * PLANE + (n & 7);
      _Float16* dl = ol + (n >> 3) * PLANE + (n & 7);
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const int p = 16 * (wm + 2 * i) + 4 * g + r;
        if (p < PIXP - 1) {   // rows past WOUT are never read by a valid output of the next layer; keep the stores inside the plane
          _Float16 h0, h1, l0, l1;
          split2(acc[i][j][r] * s_out, acc[i][j][r + 1] * s_out, one, h0, h1, l0, l1);
          dh[p * 8] = h0;
          dh[(p + 1) * 8] = h1;
          dl[p * 8] = l0;
          dl[(p + 1) * 8] = l1;
        }
      }
    }
  __syncthreads();   // output strip complete (and `red` free again)
  return s_out;
}

// Two workgroups per CU: every layer has a barrier between its last fragment read and its first store, so the output strip
// overwrites the input strip -- ONE buffer (73.8 KB) instead of a ping-pong pair -- and with 128 registers per lane (weight
// fragments one step ahead, A fragments read at their own step) two workgroups share a CU: one computes while the other stages its
// strip, splits its activations or waits at a barrier (the one-per-CU build with 238 registers, a 5-deep ring and two buffers:
// + 1 % leg time; identical bits).
template <int K0, int K1, int K2, int K3, int K4, int K5>
__global__ __launch_bounds__(512, 4) void leg_tail_kernel(TailArgs a) {
  constexpr int RING = 2, ABUF = 1;
  constexpr int HALO = (K0 - 1) + (K1 - 1) + (K2 - 1) + (K3 - 1) + (K4 - 1) + (K5 - 1);
  constexpr int W0 = TOUT + HALO;              // input pixels per workgroup (126)
  constexpr int W1 = W0 - (K0 - 1), W2 = W1 - (K1 - 1), W3 = W2 - (K2 - 1), W4 = W3 - (K3 - 1), W5 = W4 - (K4 - 1);
  static_assert(W5 - (K5 - 1) == TOUT && W0 <= PIXP - 16 && ((W1 + 15) / 16) * 16 + K0 - 1 <= PIXP, "strip does not fit the planes");
  extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
  _Float16* b0h = reinterpret_cast<_Float16*>(tail_smem);
  _Float16* b0l = b0h + HALF;
  _Float16* b1h = b0h;   // in place
  _Float16* b1l = b0l;
  float* red = reinterpret_cast<float*>(b0l + HALF);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / XT, xt = blockIdx.x - b * XT;
  const int x0 = xt * TOUT;

  // ---- input strip -> registers (all loads in flight) -> maximum -> scaled hi/lo planes of buffer 0 ----
  constexpr int Q = CH / 4;                    // float4 groups per pixel
  constexpr int TOTAL = W0 * Q;                // 4032
  constexpr int PER = (TOTAL + 511) / 512;     // 8
  const float* src = a.in + ((size_t)b * a.win + x0) * CH;
  f16x8 bq[RING][2][2];
  tail_preload<RING>(bq, a.wp[0]);   // ahead of the strip loads: both are in flight together
  f32x4 v[PER];
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + 512 * k;
    v[k] = (i < TOTAL) ? *reinterpret_cast<const f32x4*>(src + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[k][0]), fabsf(v[k][1])), fmaxf(fabsf(v[k][2]), fabsf(v[k][3]))));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
  const float s0 = ovn_pow2_scale_for(m);
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid + 512 * k;
    if (i < TOTAL) {
      const int pix = i / Q;
      const int c = 4 * (i - pix * Q);
      _Float16 h[4], l[4];
      split2(v[k][0] * s0, v[k][1] * s0, a.one, h[0], h[1], l[0], l[1]);
      split2(v[k][2] * s0, v[k][3] * s0, a.one, h[2], h[3], l[2], l[3]);
      const int o = (c >> 3) * PLANE + pix * 8 + (c & 7);
      typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
      *reinterpret_cast<f16x4v*>(b0h + o) = (f16x4v){h[0], h[1], h[2], h[3]};
      *reinterpret_cast<f16x4v*>(b0l + o) = (f16x4v){l[0], l[1], l[2], l[3]};
    }
  }
  __syncthreads();

  float s = s0;
  s = tail_layer<K0, W1, false, RING, ABUF>(b0h, b0l, b1h, b1l, red, a.wp[0], a.bias[0], s, a.sw[0], a.one, nullptr, a.wp[1], bq);
  s = tail_layer<K1, W2, false, RING, ABUF>(b1h, b1l, b0h, b0l, red, a.wp[1], a.bias[1], s, a.sw[1], a.one, nullptr, a.wp[2], bq);
  s = tail_layer<K2, W3, false, RING, ABUF>(b0h, b0l, b1h, b1l, red, a.wp[2], a.bias[2], s, a.sw[2], a.one, nullptr, a.wp[3], bq);
  s = tail_layer<K3, W4, false, RING, ABUF>(b1h, b1l, b0h, b0l, red, a.wp[3], a.bias[3], s, a.sw[3], a.one, nullptr, a.wp[4], bq);
  s = tail_layer<K4, W5, false, RING, ABUF>(b0h, b0l, b1h, b1l, red, a.wp[4], a.bias[4], s, a.sw[4], a.one, nullptr, a.wp[5], bq);
  (void)tail_layer<K5, TOUT, true, RING, ABUF>(b1h, b1l, nullptr, nullptr, red, a.wp[5], a.bias[5], s, a.sw[5], a.one,
                                   a.out + ((size_t)b * OVN_FEAT_W + x0) * CH, nullptr, bq);
}

}  // namespace

// True if layers first .. first + 5 are the reference's 1 x {9,9,9,7,5,3} 128 -> 128 tail on a single row of `w` pixels ending at
// the 360-wide feature volume (any other configuration keeps the per-layer kernels).
bool ovn_leg_tail_matches(const ovn_ctx* ctx, size_t first, int h, int w) {
  static const int kws[NLAY] = {9, 9, 9, 7, 5, 3};
  if (h != 1 || first + NLAY != ctx->leg.size()) return false;
  int ww = w;
  for (int l = 0; l < NLAY; ++l) {
    const OvnConvLayer& L = ctx->leg[first + l];
    if (L.kh != 1 || L.kw != kws[l] || L.cin != CH || L.cout != CH || L.sh != 1 || L.sw != 1 || !L.relu || L.out_cols != 0 ||
        L.wp_h == nullptr)
      return false;
    ww -= kws[l] - 1;
  }
  return ww == OVN_FEAT_W;
}

int ovn_leg_tail_forward(const ovn_ctx* ctx, size_t first, const float* in, int nb, int w, float* out, hipStream_t stream) {
  TailArgs a;
  a.in = in;
  a.out = out;
  for (int l = 0; l < NLAY; ++l) {
    const OvnConvLayer& L = ctx->leg[first + l];
    a.wp[l] = reinterpret_cast<const _Float16*>(L.wp_h);
    a.bias[l] = L.bias;
    a.sw[l] = L.sw_h;
  }
  a.one = 1.0f;
  a.win = w;
  int rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(leg_tail_kernel<9, 9, 9, 7, 5, 3>), TAIL_LDS);
  if (rc) return rc;
  hipLaunchKernelGGL((leg_tail_kernel<9, 9, 9, 7, 5, 3>), dim3((unsigned)nb * XT), dim3(512), TAIL_LDS, stream, a);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
