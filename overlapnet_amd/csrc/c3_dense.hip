// c_conv3 (3x3, 128 -> 256, ReLU) + Flatten + Dense(1) fused, input patch resident in LDS, for gfx950.
//
// Reference: generateNet.py:108-114 (Conv2D(256,(3,3),relu) -> Flatten -> Dense(1, sigmoid)).
// The generic implicit-GEMM kernel (conv_f16x3.hip) gathers every o2 element 18 times (9 taps x 2 column blocks), splits it
// into hi/lo halves each time and pushes it through the slow LDS store path behind a barrier per 32-deep K chunk.  Here a
// workgroup owns a band of output rows of ONE pair (22 rows = bands of 8, 7, 7): its input patch ((rows+2) x 24 pixels x 128
// channels) is loaded and split ONCE into an LDS-resident hi/lo image (one [pixel][8 fp16] plane per group of 8 channels:
// conflict-free ds_read_b128 per 16-lane group, see conv_strip.hip), and the 3x3 taps are just address offsets into it -- no re-staging and no barrier
// in the 36-step K loop (9 taps x 4 channel chunks of 32).  The 8 waves split the 256 output channels (2 n-tiles each), every
// wave walks all m-tiles (8 x 22 = 176 pixels = 11 exact tiles), 66 MFMAs per K step against 4 weight-fragment loads straight
// from L2 (prefetched one step ahead).  The Dense dot product is taken in the epilogue on the accumulators (o3 never goes to
// HBM unless asked for): partial sums per (band, channel half, m-tile half), combined in one fixed order by the pair's last
// workgroup to arrive (deterministic whichever that is).
#include "ovn_internal.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

// scaled fp16 hi/lo arithmetic ("f16x3", see delta_head_f16x3.hip)
struct ArithF16 {
  typedef _Float16 elem;
  typedef f16x8 v8;
  typedef f16x4 v4;
  static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int G = OVN_G;                 // 24 input rows / cols
constexpr int OW = OVN_O3_HW;            // 22 output rows / cols
constexpr int CI = OVN_C2_OUT;           // 128 input channels
constexpr int CO = OVN_C3_OUT;           // 256 output channels
constexpr int NBAND = 3;                 // output-row bands per pair: [0,8) [8,15) [15,22)
constexpr int MAX_ROWS = 8;
constexpr int MAX_MT = (MAX_ROWS * OW + 15) / 16;          // 11 m-tiles
constexpr int IN_PIX_MAX = (MAX_ROWS + 2) * G;             // 240 input pixels
constexpr int PLANE = IN_PIX_MAX * 8 + 8;                  // fp16 elements per 8-channel plane: [pixel][8] + one 16-B slot, so that the 16
                                                           // planes a wave writes at once start in different banks (3840 B apart they all hit
                                                           // bank 0; the fragment reads are unaffected: 0.637 -> 0.625 ms in one run)
constexpr int NPL = CI / 8;                                // 16 planes
constexpr size_t LDS_BYTES = 2 * (size_t)NPL * PLANE * sizeof(_Float16) + 64;   // hi + lo images + reduction scratch (16 floats)
constexpr int NW = 8;                    // waves that split the 256 output channels (2 n-tiles each)

__device__ __forceinline__ int band_start(int b) { return b == 0 ? 0 : (b == 1 ? 8 : 15); }
__device__ __forceinline__ int band_rows(int b) { return b == 0 ? 8 : 7; }

// `o2max` (n) holds the float bits of each pair's max o2 value (written by the Delta kernel's epilogue); the patch is scaled by
// s2 = 2^14 / 2^ceil(log2 max) before the split, the weights were scaled by `sw3` when they were registered, and the
// accumulators are divided by s2 * sw3 (powers of two: exact).
// NWV waves per workgroup and MSPLIT workgroups along the m-tiles: <8, 1> (one workgroup per band: sweeps), or <4, 2> (FOUR workgroups
// per band -- half of the output channels x tiles 0 .. 5 / 6 .. 10, each with its own copy of the patch, one wave per SIMD: a handful
// of pairs are a few workgroups per pair deep in their own MFMA time; one workgroup per band 49 us for a single pair, two 37 us, four
// 23 us).  The Dense partial sums leave the kernel per (band, channel half, m-tile half) in both builds -- the sweep build keeps two
// sums per thread, split at tile MT_SPLIT -- and are combined in one fixed order by the pair's last workgroup: same bits.
constexpr int MT_SPLIT = 6;
template <class A, int NWV, int MSPLIT>
__global__ __launch_bounds__(64 * NWV) void c3_dense_kernel(const float* __restrict__ o2, const typename A::elem* __restrict__ wp,
                                                           const float* __restrict__ b3, const float* __restrict__ wd,
                                                           const unsigned* __restrict__ o2max, float sw3,
                                                           float* __restrict__ partial, float* __restrict__ o3,
                                                           unsigned* __restrict__ arrived, const float* __restrict__ bd,
                                                           float* __restrict__ overlap, float* __restrict__ logit) {
  typedef typename A::elem elem_t;
  typedef typename A::v8 v8_t;
  typedef typename A::v4 v4_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  elem_t* ih = reinterpret_cast<elem_t*>(smem);
  elem_t* il = ih + NPL * PLANE;
  float* red = reinterpret_cast<float*>(il + NPL * PLANE);

  constexpr int HALVES = NW / NWV;
  constexpr int MTW = MSPLIT == 1 ? MAX_MT : MT_SPLIT;        // m-tile slots of this workgroup
  const int sub = blockIdx.x % (HALVES * MSPLIT);
  const int unit = blockIdx.x / (HALVES * MSPLIT), half = sub % HALVES, mh = sub / HALVES;
  const int mt0 = MT_SPLIT * mh;                              // first m-tile of this workgroup
  const int pair = unit / NBAND;
  const int band = unit - pair * NBAND;
  const int r0 = band_start(band);
  const int nrows = band_rows(band);
  const int npix = nrows * OW;                 // output pixels of this band
  const int nmt = (npix + 15) >> 4;            // 11 or 10
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_l = tid >> 6;                 // wave within the workgroup
  const int wave = wave_l + NWV * half;        // wave of the band: output channels 32 wave .. 32 wave + 31
  const int lrow = lane & 15;
  const int g = lane >> 4;

  // weight fragments: wp[kc][nt(16)][hi,lo][lane][8], kc = tap * 4 + channel chunk; this wave's n-tiles 2w, 2w+1
  const elem_t* wsrc = wp + ((size_t)(2 * wave) * 2) * 512 + lane * 8;
  v8_t bq[3][4];   // weight fragments of three K steps in flight (an L2 round trip is longer than one step)
#define C3_LOAD_B(DST, KC)                                                                  \
  {                                                                                         \
    const elem_t* q = wsrc + (size_t)(KC) * (16 * 2 * 512);                                 \
    DST[0] = *reinterpret_cast<const v8_t*>(q);                                           \
    DST[1] = *reinterpret_cast<const v8_t*>(q + 512);                                     \
    DST[2] = *reinterpret_cast<const v8_t*>(q + 1024);                                    \
    DST[3] = *reinterpret_cast<const v8_t*>(q + 1536);                                    \
  }
  C3_LOAD_B(bq[0], 0)   // requested ahead of the patch: both round trips overlap
  C3_LOAD_B(bq[1], 1)
  const float s2 = ovn_pow2_scale_for(__uint_as_float(o2max[pair]));
  const float inv = 1.0f / (s2 * sw3);
  // ---- input patch -> LDS, split once (x * s2 = hi + lo, both 16-bit, round to nearest) ----
  {
    const float* src = o2 + ((long long)pair * G + r0) * G * CI;     // rows r0 .. r0 + nrows + 1, contiguous in NHWC
    const int n4 = (nrows + 2) * G * CI / 4;
    // all of a thread's loads are issued before the first is used: one memory round trip per workgroup instead of fifteen
    constexpr int PER_THREAD = (MAX_ROWS + 2) * G * CI / 4 / (64 * NWV);   // 15 (30 with four waves)
    static_assert(PER_THREAD * 64 * NWV == (MAX_ROWS + 2) * G * CI / 4, "patch does not divide over the threads");
    f32x4 v[PER_THREAD];
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
      const int i = tid + k * (64 * NWV);
      v[k] = (i < n4) ? *reinterpret_cast<const f32x4*>(src + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
      const int i = tid + k * (64 * NWV);
      if (i < n4) {
        const int pix = i / (CI / 4);
        const int c = 4 * (i - pix * (CI / 4));
        v4_t h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = v[k][e] * s2;
          h[e] = (elem_t)x;
          l[e] = (elem_t)(x - (float)h[e]);
        }
        const int o = (c >> 3) * PLANE + pix * 8 + (c & 7);
        *reinterpret_cast<v4_t*>(ih + o) = h;
        *reinterpret_cast<v4_t*>(il + o) = l;
      }
    }
  }

  // per m-tile: LDS offset of this lane's output pixel at tap (0,0), channel group 8g
  int abase[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    int p = 16 * (mt0 + mt) + lrow;
    if (p >= npix) p = npix - 1;               // padded rows of the last tile recompute the last pixel (never stored)
    const int oy = p / OW;
    const int ox = p - oy * OW;
    abase[mt] = g * PLANE + (oy * G + ox) * 8;
  }

  f32x4 acc[MTW][2];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

// m-tiles go two at a time and term-major, so that consecutive MFMAs never chain on one accumulator (4 apart)
#define C3_MFMA2(M0, M1, A0H, A0L, A1H, A1L, SRC)                                                  \
  acc[M0][0] = A::mfma(A0H, SRC[0], acc[M0][0]);          \
  acc[M0][1] = A::mfma(A0H, SRC[2], acc[M0][1]);          \
  acc[M1][0] = A::mfma(A1H, SRC[0], acc[M1][0]);          \
  acc[M1][1] = A::mfma(A1H, SRC[2], acc[M1][1]);          \
  acc[M0][0] = A::mfma(A0L, SRC[0], acc[M0][0]);          \
  acc[M0][1] = A::mfma(A0L, SRC[2], acc[M0][1]);          \
  acc[M1][0] = A::mfma(A1L, SRC[0], acc[M1][0]);          \
  acc[M1][1] = A::mfma(A1L, SRC[2], acc[M1][1]);          \
  acc[M0][0] = A::mfma(A0H, SRC[1], acc[M0][0]);          \
  acc[M0][1] = A::mfma(A0H, SRC[3], acc[M0][1]);          \
  acc[M1][0] = A::mfma(A1H, SRC[1], acc[M1][0]);          \
  acc[M1][1] = A::mfma(A1H, SRC[3], acc[M1][1]);
// A fragments are read one tile pair ahead of the MFMAs that consume them (the LDS round trip hides behind 12 MFMAs)
#define C3_READ_A(BUF, SLOT, MT)                                                            \
  fh[BUF][SLOT] = *reinterpret_cast<const v8_t*>(ih + abase[MT] + toff);                  \
  fl[BUF][SLOT] = *reinterpret_cast<const v8_t*>(il + abase[MT] + toff);
#define C3_STEP(SRC, KC)                                                                    \
  {                                                                                         \
    const int tap = (KC) >> 2;                                                              \
    const int ky = tap / 3;                                                                 \
    const int toff = (ky * G + (tap - 3 * ky)) * 8 + 4 * PLANE * ((KC) & 3);                \
    v8_t fh[2][2], fl[2][2];                                                              \
    C3_READ_A(0, 0, 0)                                                                      \
    C3_READ_A(0, 1, 1)                                                                      \
    _Pragma("unroll") for (int q = 0; q < MTW / 2; ++q) {                                \
      const int cb = q & 1, nb = cb ^ 1;                                                    \
      if (q + 1 < MTW / 2) {                                                             \
        C3_READ_A(nb, 0, 2 * q + 2)                                                         \
        C3_READ_A(nb, 1, 2 * q + 3)                                                         \
      } else if (MTW % 2 == 1 && nmt == MAX_MT) {                                                           \
        C3_READ_A(nb, 0, MTW - 1)                                                        \
      }                                                                                     \
      __builtin_amdgcn_sched_barrier(0);  /* keep the reads ahead of the MFMAs they do not feed */ \
      C3_MFMA2(2 * q, 2 * q + 1, fh[cb][0], fl[cb][0], fh[cb][1], fl[cb][1], SRC)           \
      __builtin_amdgcn_sched_barrier(0);                                                    \
    }                                                                                       \
    if (MTW % 2 == 1 && nmt == MAX_MT) {  /* the 8-row band has an 11th tile */                             \
      const v8_t ah = fh[(MTW / 2) & 1][0], al = fl[(MTW / 2) & 1][0];              \
      acc[MTW - 1][0] = A::mfma(ah, SRC[0], acc[MTW - 1][0]); \
      acc[MTW - 1][1] = A::mfma(ah, SRC[2], acc[MTW - 1][1]); \
      acc[MTW - 1][0] = A::mfma(al, SRC[0], acc[MTW - 1][0]); \
      acc[MTW - 1][1] = A::mfma(al, SRC[2], acc[MTW - 1][1]); \
      acc[MTW - 1][0] = A::mfma(ah, SRC[1], acc[MTW - 1][0]); \
      acc[MTW - 1][1] = A::mfma(ah, SRC[3], acc[MTW - 1][1]); \
    }                                                                                       \
  }
  __syncthreads();  // patch complete
#pragma unroll 1
  for (int kc = 0; kc < 36; kc += 3) {
    C3_LOAD_B(bq[2], kc + 2)
    C3_STEP(bq[0], kc)
    if (kc + 3 < 36) C3_LOAD_B(bq[0], kc + 3)
    C3_STEP(bq[1], kc + 1)
    if (kc + 4 < 36) C3_LOAD_B(bq[1], kc + 4)
    C3_STEP(bq[2], kc + 2)
  }
#undef C3_LOAD_B
#undef C3_STEP
#undef C3_MFMA2
#undef C3_READ_A

  // ---- epilogue: bias + ReLU, optional o3 store, Dense partials (tiles below / from MT_SPLIT).  C/D: lane holds channel lrow of each
  //      n-tile, rows 4g..4g+3
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = 32 * wave + 16 * nt + lrow;
    const float bv = b3[n];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 16 * (mt0 + mt) + 4 * g + r;
        if (p < npix) {
          const float v = fmaxf(fmaf(acc[mt][nt][r], inv, bv), 0.0f);
          const long long fi = (long long)(r0 * OW + p) * CO + n;     // Flatten index (H, W, C) of this value
          s[(MSPLIT == 1 && mt >= MT_SPLIT) ? 1 : 0] += v * wd[fi];
          if (o3) o3[(long long)pair * OVN_DENSE_IN + fi] = v;
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s[0] += __shfl_down(s[0], off, 64);
    if (MSPLIT == 1) s[1] += __shfl_down(s[1], off, 64);
  }
  if (lane == 0) {
    red[wave_l] = s[0];
    if (MSPLIT == 1) red[NW + wave_l] = s[1];
  }
  __syncthreads();
  // partial[pair][band][half of the output channels][m-tile half]: (w0 + w1) + (w2 + w3) of each channel half
  if (tid == 0) {
    float* dst = partial + ((size_t)pair * NBAND + band) * 4;
    if (NWV == NW) {
      dst[0] = (red[0] + red[1]) + (red[2] + red[3]);
      dst[1] = (red[NW + 0] + red[NW + 1]) + (red[NW + 2] + red[NW + 3]);
      dst[2] = (red[4] + red[5]) + (red[6] + red[7]);
      dst[3] = (red[NW + 4] + red[NW + 5]) + (red[NW + 6] + red[NW + 7]);
    } else {
      dst[2 * half + mh] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    // Dense bias + sigmoid by the LAST workgroup of the pair to arrive (overlap != NULL): its 12 partial sums are combined in ONE fixed
    // order whichever workgroup that is -- no separate finishing launch (5 us of a 285 us single-pair query).  `arrived[pair]` counts
    // the pair's workgroups and is left at zero again; the partials of the other workgroups are read past this CU's caches.
    if (overlap != nullptr) {
      constexpr unsigned PARTS = NBAND * HALVES * MSPLIT;
      __threadfence();
      const unsigned prev = atomicAdd(arrived + pair, 1u);
      if (prev == PARTS - 1) {
        __threadfence();
        const float* q = partial + (size_t)pair * OVN_DENSE_PARTIALS;
        float z = 0.f;
#pragma unroll
        for (int b = 0; b < NBAND; ++b) {
          const float q0 = __hip_atomic_load(q + 4 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float q1 = __hip_atomic_load(q + 4 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float q2 = __hip_atomic_load(q + 4 * b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float q3 = __hip_atomic_load(q + 4 * b + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float zb = (q0 + q1) + (q2 + q3);     // a band = (its two m-tile halves of channel half 0) + (those of channel half 1)
          z = b == 0 ? zb : z + zb;                   // bands in order
        }
        z += bd[0];
        if (logit) logit[pair] = z;
        overlap[pair] = 1.0f / (1.0f + expf(-z));
        __hip_atomic_store(arrived + pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// The same finish as its own launch, for sweeps: there the device-scope release in front of the arrival counter (an L2 write-back per
// workgroup on this multi-XCD part) costs more than a launch -- c3_dense 0.634 + 0.008 ms separate against 0.690 ms fused per 1024 pairs.
__global__ __launch_bounds__(256) void dense_finish_kernel(const float* __restrict__ partial, const float* __restrict__ bd, int n,
                                                           float* __restrict__ overlap, float* __restrict__ logit) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const float* q = partial + (size_t)p * OVN_DENSE_PARTIALS;
  float z = 0.f;
#pragma unroll
  for (int b = 0; b < NBAND; ++b) {
    const float zb = (q[4 * b] + q[4 * b + 1]) + (q[4 * b + 2] + q[4 * b + 3]);   // the order of the fused finish
    z = b == 0 ? zb : z + zb;
  }
  z += bd[0];
  if (logit) logit[p] = z;
  overlap[p] = 1.0f / (1.0f + expf(-z));
}

}  // namespace

// o2 (n,24,24,128) fp32 -> Dense partial sums per (output-row band, half of the output channels, m-tile half) in `partial` (12 n) and,
// with `overlap` given, logit / overlap of every pair -- for a handful of pairs finished by the pair's last workgroup inside the
// kernel (`arrived`: n zeroed words that the kernel leaves zeroed), for sweeps by a small launch of its own; the same sums in the same
// order either way [+ o3 (n,22,22,256) when not NULL].
int ovn_c3_dense_forward(const ovn_ctx* ctx, const float* o2, const unsigned* o2max, int n, float* partial, float* o3,
                         unsigned* arrived, float* overlap, float* logit, hipStream_t stream) {
  OVN_REQUIRE(o2max != nullptr, OVN_ERR_ARG, "ovn_c3_dense_forward: the per-pair maxima of o2 are required");
  OVN_REQUIRE(overlap == nullptr || arrived != nullptr, OVN_ERR_ARG, "ovn_c3_dense_forward: arrival counters missing");
  int rc;
  if (n <= 21) {   // a handful of pairs: four 4-wave workgroups per band (12 n <= 252 workgroups: still one round, one per CU)
    rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(c3_dense_kernel<ArithF16, 4, 2>), LDS_BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL((c3_dense_kernel<ArithF16, 4, 2>), dim3(4 * NBAND * n), dim3(64 * 4), LDS_BYTES, stream, o2,
                       reinterpret_cast<const _Float16*>(ctx->c3.wp_h), ctx->c3.bias, ctx->wd, o2max, ctx->c3.sw_h, partial, o3, arrived,
                       ctx->bd, overlap, logit);
  } else {         // sweeps: the finish as its own small launch (see dense_finish_kernel)
    rc = ovn_allow_dynamic_lds(reinterpret_cast<const void*>(c3_dense_kernel<ArithF16, NW, 1>), LDS_BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL((c3_dense_kernel<ArithF16, NW, 1>), dim3(NBAND * n), dim3(64 * NW), LDS_BYTES, stream, o2,
                       reinterpret_cast<const _Float16*>(ctx->c3.wp_h), ctx->c3.bias, ctx->wd, o2max, ctx->c3.sw_h, partial, o3,
                       (unsigned*)nullptr, ctx->bd, (float*)nullptr, (float*)nullptr);
    if (overlap != nullptr)
      hipLaunchKernelGGL(dense_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, partial, ctx->bd, n, overlap, logit);
  }
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
