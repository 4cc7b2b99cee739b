// Internal declarations shared by the translation units of libovn_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/ovn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing -------------------------------------------------------------------------------
void ovn_set_error(const char* fmt, ...);

#define OVN_HIP_CHECK(expr)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ovn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return OVN_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

#define OVN_REQUIRE(cond, code, ...)   \
  do {                                 \
    if (!(cond)) {                     \
      ovn_set_error(__VA_ARGS__);      \
      return (code);                   \
    }                                  \
  } while (0)

// Selects a context's device for the duration of a C-ABI call and restores the caller's current device afterwards: a
// process may hold contexts on several GPUs, and a library call must not change which GPU the caller's next allocation
// or kernel lands on.
struct OvnDeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit OvnDeviceGuard(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return;
    if (cur == dev) {
      ok = true;
      return;
    }
    ok = (hipSetDevice(dev) == hipSuccess);
    if (ok) prev = cur;
  }
  ~OvnDeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  OvnDeviceGuard(const OvnDeviceGuard&) = delete;
  OvnDeviceGuard& operator=(const OvnDeviceGuard&) = delete;
};
#define OVN_ON_DEVICE(dev)                                                             \
  OvnDeviceGuard ovn_device_guard_(dev);                                               \
  OVN_REQUIRE(ovn_device_guard_.ok, OVN_ERR_HIP, "cannot select HIP device %d", (int)(dev))

// ---- feature geometry fixed by the reference network ----------------------------------------------
constexpr int OVN_FEAT_W = 360;   // leg_output_width, config/network.yml:77
constexpr int OVN_FEAT_C = 128;   // s_conv10 filters, generateNet.py:214
constexpr int OVN_FEAT_ELEMS = OVN_FEAT_W * OVN_FEAT_C;
constexpr int OVN_S = 15;         // conv1NetworkHead_conv1size default, generateNet.py:88-89
constexpr int OVN_G = OVN_FEAT_W / OVN_S;            // 24
constexpr int OVN_C1_OUT = 64;    // c_conv1 filters
constexpr int OVN_C2_OUT = 128;   // c_conv2 filters
constexpr int OVN_C3_OUT = 256;   // c_conv3 filters
constexpr int OVN_O3_HW = OVN_G - 2;                 // 22
constexpr int OVN_DENSE_IN = OVN_O3_HW * OVN_O3_HW * OVN_C3_OUT;  // 123904
constexpr int OVN_A2_IN_YAW_MAX_PAIRS = 64;                  // sweeps up to this many pairs carry the query's A2 tasks in their yaw launch
constexpr int OVN_DENSE_PARTIALS = 12;                        // Dense partial sums per pair left by c3_dense_kernel: 3 row bands x 2 channel halves x 2 m-tile halves
constexpr int OVN_ACTMAX_SLOTS = 32;                           // layers with per-scan activation maxima (f16x3 scales)
constexpr int OVN_LEG_SLICE = 1024;                             // scans per pass of ovn_leg over its ping-pong scratch
constexpr int OVN_ACTMAX_STRIDE = 32;                          // words between the maxima of two scans: every scan's word has its
                                                               // own 128-byte line (atomics on one line serialise at the memory side)
constexpr int OVN_SPEC_W = 368;                                 // floats per spectrum row: Re[0..180] | pad | Im at 184.. | pad
constexpr int OVN_SPEC_ELEMS = OVN_FEAT_C * OVN_SPEC_W;        // 47104 floats = 188,416 B per scan
static_assert(OVN_DELTA_CACHE_ELEMS >= OVN_FEAT_ELEMS + 24 * 128 + 4, "include/ovn_hip.h: OVN_DELTA_CACHE_ELEMS");

// ---- scaled fp16 hi/lo arithmetic ("f16x3") ----------------------------------------------------------
// Power-of-two scale that brings a tensor whose largest magnitude is m into [2^13, 2^14): 2^(14 - e) with 2^(e-1) <= m < 2^e.
// fp16 holds 2^14 with 2x headroom below 65504; m == 0, Inf or NaN -> 1 (nothing to protect).
__host__ __device__ inline float ovn_pow2_scale_for(float m) {
  if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
  int e = 0;
  (void)frexpf(m, &e);
  int k = 14 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return ldexpf(1.0f, k);
}

#ifdef __HIPCC__
// max |value| of a WORKGROUP folded into one device word: wave shuffle reduction, the waves' maxima through `red` (>= 16 floats of
// LDS that nothing else uses at this point), then at most ONE atomicMax per workgroup (|v| orders like its float bits; skipped when
// it would not raise the word).  Per-wave atomics on per-scan words cost s_conv1 / s_conv2 +75 % (164 k device-scope atomics per
// 1025 scans, all waves of a scan arriving at a zeroed word together).  Every thread of the workgroup must call it.
__device__ __forceinline__ void ovn_fold_absmax_wg(float vmax, unsigned* word, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
  const int nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int w = 1; w < nw; ++w) m = fmaxf(m, red[w]);
    const unsigned bits = __float_as_uint(m);
    if (m > 0.f && bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
  }
}
#endif

// static scales / norms of the Delta-head weights (delta_head_f16x3.hip)
struct OvnHeadScales {
  float sw1 = 1.f, sw2 = 1.f;     // power-of-two scales of the c_conv1 / c_conv2 kernels
  float sws = 1.f;                // ... of the tap-summed c_conv1 kernel (left-volume linear term)
  float w1_colsum = 0.f;          // max over output channels o of sum |W1[., ., o]|: bound of c_conv1's output per unit input
  float b1_absmax = 0.f;
};

// ---- a convolution layer in MFMA fragment order ----------------------------------------------------
// Raise a kernel's dynamic-LDS limit once per (kernel, device): the attribute is per device, and one process may hold
// contexts on several GPUs.  Thread-safe.
int ovn_allow_dynamic_lds(const void* kernel, size_t bytes);

struct OvnConvLayer {
  std::string name;
  int kh = 0, kw = 0, cin = 0, cout = 0, sh = 1, sw = 1;
  int relu = 1;
  int out_cols = 0;  // 0 = cout; else only the first out_cols output channels are stored, at row stride out_cols
                     // (a layer padded with zero filters so that a wide tile divides cout: the 368-column DFT layers)
  int K = 0;      // kh*kw*cin
  int nkc = 0;    // ceil(K/16)
  float* wp = nullptr;    // [nkc][cout/16][64][4] fragment-ordered copy (device)
  float* bias = nullptr;  // [cout] (device)
  void* wp_h = nullptr;   // optional hi/lo fp16 fragments of sw_h * W, [ceil(K/32)][cout/16][2][64][8] (conv_f16x3.hip)
  float sw_h = 1.f;       // power-of-two weight scale of wp_h
  void* wp_h16 = nullptr; // layers with cin 4 / 16 and kw <= 16: the same fragments in the K order (ky, kx padded to 16, c) of the
                          // pixel-major strip kernel (conv_strip.hip), [kh * 512 / cin... steps][cout/16][2][64][8]
};

struct ovn_ctx {
  int device = 0;
  int in_h = 0, in_w = 0, in_c = 0;
  std::vector<OvnConvLayer> leg;
  bool finalized = false;
  int feat_w = 0;
  // head
  bool head_set = false;
  int head_s = OVN_S;      // conv1NetworkHead_conv1size (ovn_set_head_geometry); the fast Delta paths serve 15, anything else the
  int head_g = OVN_G;      // general fp32 path of delta_head_generic.hip; head_g = 360 // head_s
  float* w1p = nullptr;  // c_conv1 in the K-permuted fragment order of the fused kernel
  float* b1 = nullptr;
  OvnConvLayer c2;       // c_conv2 as a [960][128] GEMM operand in fragment order
  OvnConvLayer c3;       // c_conv3 as a regular conv layer
  void* w1p_h = nullptr;   // c_conv1 / c_conv2 scaled hi/lo fp16 fragments (delta_head_f16x3.hip)
  void* w2p_h = nullptr;
  float* w1raw = nullptr;  // c_conv1 kernel as registered, [1920][64]: B operand of the right-volume linear term
  float* w1sum = nullptr;  // c_conv1 kernel summed over its 15 taps, [128][64]: B operand of the left-volume linear term
  float* w1col = nullptr;  // [64] column sums of the c_conv1 kernel (shift term)
  void* wsp_h = nullptr;   // w1sum as scaled hi/lo fp16 fragments
  float* w2sum = nullptr;  // c_conv2 kernel summed over its 15 taps, [64][128]: the right-volume linear term pushed through c_conv2
  OvnHeadScales hs;
  int leg_mode = 1;        // 0 = fp32 MFMA (conv_f32.hip), 1 = scaled 3-term fp16 split on the fp16 MFMA (conv_f16x3.hip)
  int head_compact = 1;    // ovn_set_head_compaction: 1 = 1-vs-N sweeps drop the query's dead channels from the Delta contraction (exact)
  int proj_trig = 0;       // ovn_set_projection_trig: 0 = NumPy-on-AVX512 (SVML) float32 angles, 1 = correctly rounded float32 angles
  unsigned* actmax = nullptr;   // [layer][scan of the slice][OVN_ACTMAX_STRIDE] float bits of max |layer input| of that scan (f16x3 scales)
  int head_mode = 1;       // 0 = fp32 MFMA (exact fp32), 1 = scaled 3-term fp16 split on the fp16 MFMA (default)
  float* wd = nullptr;   // dense kernel [123904]
  float* bd = nullptr;   // dense bias [1]
  // spectral correlation head: constant twiddle layers (corr_spectral.hip)
  OvnConvLayer dft;        // forward transform as a conv layer (fp32 mode) + its fp16 hi/lo fragments (dft_f16x3_kernel)
  double* tw64 = nullptr;  // [2][360] cos / sin (2 pi m / 360) in fp64: start values of the inverse transform (spectral_corr_kernel)
  // optional RCCL communicator of the sharded sweep (comm.hip)
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  // launch structure of a head call (ovn_set_head_pipeline): pairs per pass over the scratch, pairs per sub-chunk (0 = the whole
  // chunk), streams the sub-chunks alternate between (1 or 2), spectral yaw head on its own side stream
  int64_t head_chunk = 1024;
  int64_t head_sub = 0;
  int head_streams = 1;
  int head_yaw_side = 0;   // measured (profiles/r3a_pipeline_matrix.md): no gain from any of the forked forms, the serial order is the default
  bool aux_ready = false;
  hipStream_t aux[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  // scratch
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // where the last ovn_heads call left its c_conv2 / c_conv3 activations (first chunk), for tests
  // optional per-kernel timing with HIP events on the launch stream (ovn_profile_begin/end)
  bool prof = false;
  struct ProfRec {
    hipEvent_t a, b;
    int kind;
  };
  std::vector<ProfRec> prof_recs;
  const float* dbg_o2 = nullptr;
  const float* dbg_o3 = nullptr;   // fp32 head mode only; in f16x3 mode o3 is recomputed on request (dbg_partial = scratch)
  float* dbg_partial = nullptr;
  const unsigned* dbg_o2max = nullptr;
  int64_t dbg_n = 0;
  unsigned* c3_arrived = nullptr;       // arrival counters of c3_dense_kernel, one per pair of a chunk, zero between launches
  int64_t c3_arrived_n = 0;
  const unsigned* dbg_live = nullptr;   // live-channel list of the most recent f16x3 Delta sweep (NULL: it walked all 128 channels)
};

// kernel classes reported by ovn_profile_end
enum { OVN_K_LEG = 0, OVN_K_CORR = 1, OVN_K_DELTA = 2, OVN_K_C3 = 3, OVN_K_DENSE = 4, OVN_K_PROJ = 5, OVN_K_SPECTRUM = 6,
       OVN_K_CORR_SPECTRAL = 7, OVN_K_DELTA_PREP = 8, OVN_K_DELTA_C2 = 9, OVN_K_COUNT = 10 };

struct OvnProfScope {
  ovn_ctx* ctx;
  hipStream_t stream;
  hipEvent_t a = nullptr, b = nullptr;
  int kind;
  OvnProfScope(ovn_ctx* c, int k, hipStream_t s) : ctx(c), stream(s), kind(k) {
    if (ctx->prof && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, stream);
  }
  ~OvnProfScope() {
    if (ctx->prof && a && b) {
      (void)hipEventRecord(b, stream);
      ctx->prof_recs.push_back({a, b, kind});
    }
  }
};

int ovn_ws_reserve(ovn_ctx* ctx, size_t bytes, hipStream_t stream);

// ---- kernels' host launchers (each returns an OVN_* code) ------------------------------------------
// conv_f32.hip
int ovn_conv_prepare(OvnConvLayer* L, const float* kernel_dev, const float* bias_dev, hipStream_t stream);
void ovn_conv_release(OvnConvLayer* L);
int ovn_conv_forward(const OvnConvLayer& L, const float* in, int nb, int h, int w, float* out, int* oh,
                     int* ow, hipStream_t stream);

// conv_f16x3.hip
// scaled fp16 hi/lo fragments of the layer (wp_h, sw_h); synchronises `stream` (the weight maximum is read back)
int ovn_conv_prepare_f16x3(OvnConvLayer* L, const float* kernel_dev, hipStream_t stream);
// in_max[scan]: float bits of max |in| of every scan of the call (device words, left there by the producer of `in` or by
// ovn_absmax_forward); out_max: NULL, or zeroed device words [scan] into which max |out| of every scan is folded for the next
// layer.  Scales are per SCAN and every call size takes the same kernels: a scan's result does not depend on its batch.
int ovn_conv_forward_f16x3(const OvnConvLayer& L, const float* in, int nb, int h, int w, float* out, int* oh, int* ow,
                           const unsigned* in_max, unsigned* out_max, hipStream_t stream);
int ovn_absmax_forward(const float* x, int n_scans, long long per_scan, unsigned* out_max, hipStream_t stream);

// delta_head.hip
int ovn_delta_prepare_w1(const float* c1_kernel_dev, float** w1p_out, hipStream_t stream);
int ovn_delta_c12_forward(const ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r,
                          const int32_t* ridx, int n, float* o2, hipStream_t stream);
int ovn_dense_sigmoid_forward(const ovn_ctx* ctx, const float* o3, int n, float* overlap, float* logit,
                              hipStream_t stream);

// delta_head_f16x3.hip.  `scratch` (ovn_delta_f16x3_scratch_bytes(n, ridx != NULL) bytes, caller-owned: 2.9 MB per pair) holds the
// per-pair scales, both packed volumes, the linear terms and the c_conv1 min-term rows between the two kernels; *o2max_out points
// at the per-pair maxima of the c_conv2 output inside it (input of ovn_c3_dense_forward).
int ovn_delta_prepare_f16x3(ovn_ctx* ctx, const float* c1_kernel_dev, const float* c1_bias_dev, const float* c2_kernel_dev,
                            hipStream_t stream);
size_t ovn_delta_f16x3_scratch_bytes(int n, bool per_pair_right);
int ovn_delta_c12_f16x3_forward(ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r,
                                const int32_t* ridx, int n, void* scratch, unsigned** o2max_out, float* o2, hipStream_t stream,
                                int pair0 = 0,    // pair0: index of the call's first pair in the sweep (rotation of the K walks)
                                const float* dcache_l = nullptr,   // Delta cache rows of the left pool (ovn_delta_cache), 1-vs-N only
                                bool a2_done = false);   // A2raw of the (single) right volume is already in the scratch (ovn_delta_f16x3_a2raw)
float* ovn_delta_f16x3_a2raw(void* scratch, int n);
int ovn_delta_cache_forward(ovn_ctx* ctx, const float* feats, int n, float* cache, hipStream_t stream);
int ovn_delta_walk_stats(ovn_ctx* ctx, int32_t* out16, hipStream_t stream);

// delta_head_generic.hip: the Delta head for any conv1size (fp32, generality path)
size_t ovn_delta_generic_pair_bytes(int G);
int ovn_delta_generic_forward(const ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r,
                              const int32_t* ridx, int n, void* scratch, float* overlap, float* logit, hipStream_t stream);

// corr_head.hip
int ovn_corr_forward(const float* feats_l, const int32_t* lidx, const float* feats_r, const int32_t* ridx,
                     int n, int32_t* yaw, float* corr, hipStream_t stream);

// corr_spectral.hip
int ovn_spectral_prepare(ovn_ctx* ctx, hipStream_t stream);
int ovn_spectrum_forward(ovn_ctx* ctx, const float* feats, int n, float* spectra, hipStream_t stream);
// conv_strip.hip: LDS-resident strip kernels for the 3 x KW / stride (2,1) leg layers with 64 outputs (f16x3 mode)
// call_nb: scans of the whole call this slice belongs to (kernel choice is per call, not per slice)
bool ovn_conv_strip_own_scale(const OvnConvLayer& L, long long call_nb, int h, int w);
int ovn_conv_strip_try(const OvnConvLayer& L, const float* in, int nb, long long call_nb, int h, int w, float* out,
                       const unsigned* in_max, unsigned* out_max, hipStream_t stream);

// leg_front.hip: s_conv1 + s_conv2 of the C = 4 network fused (f16x3): the 850 KB activation between them stays in LDS
bool ovn_leg_front_matches(const ovn_ctx* ctx, size_t first, int h, int w);
int ovn_leg_front_forward(const ovn_ctx* ctx, size_t first, const float* in, int nb, int h, int w, float* out, int* oh_out, int* ow_out,
                          unsigned* out_max, hipStream_t stream);

// leg_tail.hip: the last six leg layers (1 x {9,9,9,7,5,3}, 128 -> 128) fused, activations carried through LDS (f16x3, batched calls)
bool ovn_leg_tail_matches(const ovn_ctx* ctx, size_t first, int h, int w);
int ovn_leg_tail_forward(const ovn_ctx* ctx, size_t first, const float* in, int nb, int w, float* out, hipStream_t stream);

// c3_dense.hip: c_conv3 + Flatten + Dense fused (f16x3 mode), input patch resident in LDS
// o2max: the per-pair maxima of o2 left by the f16x3 Delta kernel (scale of the fp16 split)
int ovn_c3_dense_forward(const ovn_ctx* ctx, const float* o2, const unsigned* o2max, int n, float* partial, float* o3,
                         unsigned* arrived, float* overlap, float* logit, hipStream_t stream);

// overlap_gt.hip
int ovn_gt_range_forward(const float* points, const int64_t* offsets, int n_scans, long long max_points, const double* ref_poses,
                         const double* inv_cur_pose, int H, int W, double fov_up_deg, double fov_down_deg, double max_range,
                         float* range_out, hipStream_t stream);
int ovn_gt_count_forward(const float* ref_ranges, const float* cur_range, int n, int npix, int32_t* counts, hipStream_t stream);
int ovn_best_match_forward(const float* overlap, const int32_t* yaw, const int32_t* ids, int n, float threshold,
                           int index_offset, int32_t* out, hipStream_t stream);
// a2_feats_r / a2raw non-NULL (small 1-vs-N sweeps): the launch also computes A2raw of that right volume (delta_a2.h) in extra workgroups
int ovn_corr_spectral_forward(ovn_ctx* ctx, const float* spec_l, const int32_t* lidx, const float* spec_r,
                              const int32_t* ridx, int n, int32_t* yaw, float* corr, hipStream_t stream,
                              const float* a2_feats_r = nullptr, float* a2raw = nullptr);

// projection.hip
int ovn_project_forward(ovn_ctx* ctx, const float* points, const int64_t* offsets, int n_scans,
                        int64_t max_points, int H, int W, double fov_up_deg, double fov_down_deg,
                        double max_range, float* range, float* vertex, float* intensity, int32_t* idx,
                        float* normal, float* stacked, int use_depth, int use_normals, int use_intensity,
                        hipStream_t stream);

int ovn_projection_angles_forward(const float* points, int64_t n, int H, int W, double fov_up_deg, double fov_down_deg,
                                  double max_range, float* yaw, float* pitch, int32_t* pixel, hipStream_t stream, int trig = 0);
int ovn_normals_forward(const float* range, const float* vertex, int n_scans, int H, int W, float* normal,
                        hipStream_t stream);

// selftest.hip
int ovn_mfma_selftest(hipStream_t stream);
