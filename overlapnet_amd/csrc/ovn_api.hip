// extern "C" surface of libovn_hip.so (declared in include/ovn_hip.h) -- argument checking, weight
// re-tiling, scratch management and the launch sequences.  No torch types anywhere: plain pointers.
#include <stdarg.h>
#include <mutex>
#include <set>
#include <utility>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ovn_internal.h"

int ovn_allow_dynamic_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  OVN_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({kernel, dev})) return OVN_OK;
  OVN_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  done.insert({kernel, dev});
  return OVN_OK;
}

static thread_local char g_err[1024] = "";

void ovn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// One scratch block per context, grown on demand and never shrunk.  Growing it synchronises `stream` and frees the old block
// (hipFree waits for the device), so a context must be driven from ONE stream at a time (include/ovn_hip.h says so).
int ovn_ws_reserve(ovn_ctx* ctx, size_t bytes, hipStream_t stream) {
  if (bytes <= ctx->ws_bytes) return OVN_OK;
  if (ctx->ws) {
    OVN_HIP_CHECK(hipStreamSynchronize(stream));  // earlier launches may still use the old block
    OVN_HIP_CHECK(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    ctx->dbg_o2 = ctx->dbg_o3 = nullptr;          // they pointed into the old block
    ctx->dbg_partial = nullptr;
    ctx->dbg_o2max = nullptr;
    ctx->dbg_n = 0;
  }
  const size_t want = bytes + bytes / 8;  // a little headroom so near-equal requests do not thrash
  OVN_HIP_CHECK(hipMalloc(&ctx->ws, want));
  ctx->ws_bytes = want;
  return OVN_OK;
}

extern "C" {

int ovn_abi_version(void) { return OVN_ABI_VERSION; }

const char* ovn_last_error(void) { return g_err; }

int ovn_create(int device_id, int in_h, int in_w, int in_c, ovn_ctx** out) {
  OVN_REQUIRE(out != nullptr, OVN_ERR_ARG, "ovn_create: out is NULL");
  OVN_REQUIRE(in_h > 0 && in_w > 0 && in_c > 0, OVN_ERR_ARG, "ovn_create: bad input shape %dx%dx%d", in_h, in_w, in_c);
  int ndev = 0;
  OVN_HIP_CHECK(hipGetDeviceCount(&ndev));
  OVN_REQUIRE(device_id >= 0 && device_id < ndev, OVN_ERR_ARG, "ovn_create: device %d not present (%d devices)", device_id, ndev);
  OVN_ON_DEVICE(device_id);
  hipDeviceProp_t prop;
  OVN_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
  OVN_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, OVN_ERR_STATE,
              "ovn_create: this library is built for gfx950 only, device %d is %s", device_id, prop.gcnArchName);
  ovn_ctx* c = new ovn_ctx();
  c->device = device_id;
  c->in_h = in_h;
  c->in_w = in_w;
  c->in_c = in_c;
  // experiment knobs (tools/experiments): the defaults are the measured best, ovn_set_head_pipeline is the API
  if (const char* e = getenv("OVN_HEAD_CHUNK")) c->head_chunk = atoll(e) > 0 ? atoll(e) : c->head_chunk;
  if (const char* e = getenv("OVN_HEAD_SUBCHUNK")) c->head_sub = atoll(e) >= 0 ? atoll(e) : c->head_sub;
  if (const char* e = getenv("OVN_HEAD_STREAMS")) c->head_streams = atoi(e) == 2 ? 2 : 1;
  if (const char* e = getenv("OVN_YAW_SIDE")) c->head_yaw_side = atoi(e) ? 1 : 0;
  int rc = ovn_spectral_prepare(c, nullptr);
  if (rc) {
    ovn_conv_release(&c->dft);
    if (c->tw64) (void)hipFree(c->tw64);
    delete c;
    return rc;
  }
  *out = c;
  return OVN_OK;
}

int ovn_destroy(ovn_ctx* ctx) {
  if (!ctx) return OVN_OK;
  if (ctx->comm) (void)ovn_comm_destroy(ctx);
  OVN_ON_DEVICE(ctx->device);
  (void)hipDeviceSynchronize();
  for (auto& l : ctx->leg) ovn_conv_release(&l);
  ovn_conv_release(&ctx->c2);
  ovn_conv_release(&ctx->c3);
  ovn_conv_release(&ctx->dft);
  if (ctx->tw64) (void)hipFree(ctx->tw64);
  if (ctx->w1p) (void)hipFree(ctx->w1p);
  if (ctx->b1) (void)hipFree(ctx->b1);
  if (ctx->wd) (void)hipFree(ctx->wd);
  if (ctx->bd) (void)hipFree(ctx->bd);
  if (ctx->w1p_h) (void)hipFree(ctx->w1p_h);
  if (ctx->w2p_h) (void)hipFree(ctx->w2p_h);
  if (ctx->w1raw) (void)hipFree(ctx->w1raw);
  if (ctx->w1sum) (void)hipFree(ctx->w1sum);
  if (ctx->w1col) (void)hipFree(ctx->w1col);
  if (ctx->wsp_h) (void)hipFree(ctx->wsp_h);
  if (ctx->w2sum) (void)hipFree(ctx->w2sum);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->actmax) (void)hipFree(ctx->actmax);
  if (ctx->c3_arrived) (void)hipFree(ctx->c3_arrived);
  if (ctx->aux_ready) {
    for (int i = 0; i < 2; ++i) {
      (void)hipStreamDestroy(ctx->aux[i]);
      (void)hipEventDestroy(ctx->ev_join[i]);
    }
    (void)hipEventDestroy(ctx->ev_fork);
  }
  delete ctx;
  return OVN_OK;
}

int ovn_add_leg_layer(ovn_ctx* ctx, const char* name, const float* kernel_dev, const float* bias_dev, int kh, int kw,
                      int cin, int cout, int stride_h, int stride_w, void* stream) {
  OVN_REQUIRE(ctx && name && kernel_dev && bias_dev, OVN_ERR_ARG, "ovn_add_leg_layer: NULL argument");
  OVN_REQUIRE(kh > 0 && kw > 0 && cin > 0 && cout > 0 && stride_h > 0 && stride_w > 0, OVN_ERR_ARG,
              "ovn_add_leg_layer(%s): bad geometry", name);
  OVN_REQUIRE(!ctx->finalized, OVN_ERR_STATE, "ovn_add_leg_layer(%s): context already finalized", name);
  const int expect_cin = ctx->leg.empty() ? ctx->in_c : ctx->leg.back().cout;
  OVN_REQUIRE(cin == expect_cin, OVN_ERR_ARG, "ovn_add_leg_layer(%s): cin=%d but previous layer produces %d", name, cin, expect_cin);
  OVN_ON_DEVICE(ctx->device);
  OvnConvLayer L;
  L.name = name;
  L.kh = kh;
  L.kw = kw;
  L.cin = cin;
  L.cout = cout;
  L.sh = stride_h;
  L.sw = stride_w;
  L.relu = 1;  // every leg layer is Conv2D(..., activation='relu'), generateNet.py:161-214
  int rc = ovn_conv_prepare(&L, kernel_dev, bias_dev, (hipStream_t)stream);
  if (rc) return rc;
  rc = ovn_conv_prepare_f16x3(&L, kernel_dev, (hipStream_t)stream);
  if (rc) {
    ovn_conv_release(&L);
    return rc;
  }
  ctx->leg.push_back(L);
  return OVN_OK;
}

int ovn_set_head_weights(ovn_ctx* ctx, const float* c1k, const float* c1b, const float* c2k, const float* c2b,
                         const float* c3k, const float* c3b, const float* dk, const float* db, void* stream_) {
  OVN_REQUIRE(ctx && c1k && c1b && c2k && c2b && c3k && c3b && dk && db, OVN_ERR_ARG, "ovn_set_head_weights: NULL argument");
  OVN_ON_DEVICE(ctx->device);
  hipStream_t stream = (hipStream_t)stream_;
  {  // drop whatever an earlier (possibly half-failed) call left behind
    ovn_conv_release(&ctx->c2);
    ovn_conv_release(&ctx->c3);
    if (ctx->w1p) (void)hipFree(ctx->w1p);
    if (ctx->b1) (void)hipFree(ctx->b1);
    if (ctx->wd) (void)hipFree(ctx->wd);
    if (ctx->bd) (void)hipFree(ctx->bd);
    if (ctx->w1p_h) (void)hipFree(ctx->w1p_h);
    if (ctx->w2p_h) (void)hipFree(ctx->w2p_h);
    if (ctx->w1raw) (void)hipFree(ctx->w1raw);
    if (ctx->w1sum) (void)hipFree(ctx->w1sum);
    if (ctx->w1col) (void)hipFree(ctx->w1col);
    if (ctx->wsp_h) (void)hipFree(ctx->wsp_h);
      if (ctx->w2sum) (void)hipFree(ctx->w2sum);
    ctx->w1p_h = ctx->w2p_h = ctx->wsp_h = nullptr;
    ctx->w1raw = ctx->w1sum = ctx->w1col = ctx->w2sum = nullptr;
    ctx->w1p = ctx->b1 = ctx->wd = ctx->bd = nullptr;
    ctx->head_set = false;
  }
  const int hs = ctx->head_s, hg = ctx->head_g;
  const bool general = (hs != OVN_S);   // any other conv1size: general fp32 path (delta_head_generic.hip), no fast-path operands
  int rc = OVN_OK;
  if (!general) {
    rc = ovn_delta_prepare_w1(c1k, &ctx->w1p, stream);
    if (rc) return rc;
  } else {
    const size_t w1_bytes = (size_t)hs * OVN_FEAT_C * OVN_C1_OUT * sizeof(float);
    OVN_HIP_CHECK(hipMalloc((void**)&ctx->w1raw, w1_bytes));
    OVN_HIP_CHECK(hipMemcpyAsync(ctx->w1raw, c1k, w1_bytes, hipMemcpyDeviceToDevice, stream));
  }
  OVN_HIP_CHECK(hipMalloc((void**)&ctx->b1, OVN_C1_OUT * sizeof(float)));
  OVN_HIP_CHECK(hipMemcpyAsync(ctx->b1, c1b, OVN_C1_OUT * sizeof(float), hipMemcpyDeviceToDevice, stream));
  // c_conv2 (15,1,64,128): as a GEMM operand it is the [960][128] matrix, k = di*64 + o
  ctx->c2 = OvnConvLayer();
  ctx->c2.name = "c_conv2";
  ctx->c2.kh = hs;
  ctx->c2.kw = 1;
  ctx->c2.cin = OVN_C1_OUT;
  ctx->c2.cout = OVN_C2_OUT;
  ctx->c2.sh = hs;
  ctx->c2.sw = 1;
  ctx->c2.relu = 1;
  rc = ovn_conv_prepare(&ctx->c2, c2k, c2b, stream);
  if (rc) return rc;
  if (!general) {
    rc = ovn_delta_prepare_f16x3(ctx, c1k, c1b, c2k, stream);
    if (rc) return rc;
  }
  ctx->c3 = OvnConvLayer();
  ctx->c3.name = "c_conv3";
  ctx->c3.kh = 3;
  ctx->c3.kw = 3;
  ctx->c3.cin = OVN_C2_OUT;
  ctx->c3.cout = OVN_C3_OUT;
  ctx->c3.sh = 1;
  ctx->c3.sw = 1;
  ctx->c3.relu = 1;
  rc = ovn_conv_prepare(&ctx->c3, c3k, c3b, stream);
  if (rc) return rc;
  if (!general) {
    rc = ovn_conv_prepare_f16x3(&ctx->c3, c3k, stream);
    if (rc) return rc;
  }
  const size_t dense_in = (size_t)(hg - 2) * (hg - 2) * OVN_C3_OUT;   // 123904 at conv1size 15
  OVN_HIP_CHECK(hipMalloc((void**)&ctx->wd, dense_in * sizeof(float)));
  OVN_HIP_CHECK(hipMalloc((void**)&ctx->bd, sizeof(float)));
  OVN_HIP_CHECK(hipMemcpyAsync(ctx->wd, dk, dense_in * sizeof(float), hipMemcpyDeviceToDevice, stream));
  OVN_HIP_CHECK(hipMemcpyAsync(ctx->bd, db, sizeof(float), hipMemcpyDeviceToDevice, stream));
  OVN_HIP_CHECK(hipStreamSynchronize(stream));
  ctx->head_set = true;
  return OVN_OK;
}

int ovn_finalize(ovn_ctx* ctx, int* feat_w) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_finalize: ctx is NULL");
  OVN_REQUIRE(!ctx->leg.empty(), OVN_ERR_STATE, "ovn_finalize: no leg layers registered");
  int h = ctx->in_h, w = ctx->in_w, c = ctx->in_c;
  for (const auto& l : ctx->leg) {
    OVN_REQUIRE(h >= l.kh && w >= l.kw, OVN_ERR_ARG, "ovn_finalize: layer %s does not fit its %dx%d input", l.name.c_str(), h, w);
    h = (h - l.kh) / l.sh + 1;
    w = (w - l.kw) / l.sw + 1;
    c = l.cout;
  }
  OVN_REQUIRE(h == 1 && w == OVN_FEAT_W && c == OVN_FEAT_C, OVN_ERR_ARG,
              "ovn_finalize: leg produces %dx%dx%d, the heads need 1x%dx%d", h, w, c, OVN_FEAT_W, OVN_FEAT_C);
  ctx->feat_w = w;
  ctx->finalized = true;
  if (feat_w) *feat_w = w;
  return OVN_OK;
}

int ovn_leg(ovn_ctx* ctx, const float* images_dev, int64_t n, float* features_dev, void* stream_) {
  OVN_REQUIRE(ctx && ctx->finalized, OVN_ERR_STATE, "ovn_leg: context not finalized");
  OVN_REQUIRE(n >= 0, OVN_ERR_ARG, "ovn_leg: n < 0");
  if (n == 0) return OVN_OK;
  OVN_REQUIRE(images_dev && features_dev, OVN_ERR_ARG, "ovn_leg: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  hipStream_t stream = (hipStream_t)stream_;
  // largest intermediate activation per scan decides the ping-pong buffer size
  size_t max_act = 0;
  {
    int h = ctx->in_h, w = ctx->in_w;
    for (const auto& l : ctx->leg) {
      h = (h - l.kh) / l.sh + 1;
      w = (w - l.kw) / l.sw + 1;
      const size_t e = (size_t)h * w * l.cout;
      if (e > max_act) max_act = e;
    }
  }
  // process the batch in slices so the scratch stays bounded (2 x slice x 770 KB at C=4: at most 1.6 GB); the slices are BALANCED
  // (1025 scans = 513 + 512, not 1024 + 1: a one-scan slice costs a fifth of a 256-scan one, its kernels being a handful of
  // workgroups deep in their own latency) -- a scan's result does not depend on the slice it falls into.  Slices of 256 / 512 / 1024
  // scans: 5.18 / 5.00 / 4.93 ms per 1025 scans (fewer launch ramps and drains between the five kernels of a slice)
  const int64_t nslices = (n + OVN_LEG_SLICE - 1) / OVN_LEG_SLICE;
  const int64_t slice = (n + nslices - 1) / nslices;
  const size_t buf_bytes = ((size_t)slice * max_act * sizeof(float) + 255) & ~(size_t)255;
  int rc = ovn_ws_reserve(ctx, 2 * buf_bytes, stream);
  if (rc) return rc;
  float* buf[2] = {reinterpret_cast<float*>(ctx->ws), reinterpret_cast<float*>(static_cast<char*>(ctx->ws) + buf_bytes)};
  const size_t in_elems = (size_t)ctx->in_h * ctx->in_w * ctx->in_c;
  OVN_REQUIRE(ctx->leg.size() + 1 <= OVN_ACTMAX_SLOTS, OVN_ERR_STATE, "ovn_leg: too many leg layers");
  const size_t actmax_bytes = (size_t)OVN_ACTMAX_SLOTS * OVN_LEG_SLICE * OVN_ACTMAX_STRIDE * sizeof(unsigned);
  if (ctx->leg_mode != 0 && !ctx->actmax) OVN_HIP_CHECK(hipMalloc((void**)&ctx->actmax, actmax_bytes));
  for (int64_t s0 = 0; s0 < n; s0 += slice) {
    const int nb = (int)((n - s0 < slice) ? (n - s0) : slice);
    const float* cur = images_dev + (size_t)s0 * in_elems;
    int h = ctx->in_h, w = ctx->in_w;
    if (ctx->leg_mode != 0) {
      // f16x3: word [li][scan of the slice] (rows of `slice` words: a one-scan call clears 12 words, not 12 x 1024) = max |input of
      // layer li| of that scan, folded by the kernel that produces it.  Scales are per scan
      // and every call size runs the same kernels, so a scan's feature volume does not depend on the batch it is computed in
      // (the first layer's kernel at C = 4 and the fused tail take the maximum of their own strip / tile instead)
      OVN_HIP_CHECK(hipMemsetAsync(ctx->actmax, 0, (ctx->leg.size() + 1) * (size_t)slice * OVN_ACTMAX_STRIDE * sizeof(unsigned), stream));
      const bool own = (reinterpret_cast<uintptr_t>(cur) & 15) == 0 && ovn_conv_strip_own_scale(ctx->leg[0], n, h, w);
      if (!own) {
        OvnProfScope ps(ctx, OVN_K_LEG, stream);
        rc = ovn_absmax_forward(cur, nb, (long long)in_elems, ctx->actmax, stream);
        if (rc) return rc;
      }
    }
    for (size_t li = 0; li < ctx->leg.size(); ++li) {
      const bool last = (li + 1 == ctx->leg.size());
      float* dst = last ? features_dev + (size_t)s0 * OVN_FEAT_ELEMS : buf[li & 1];
      int oh = 0, ow = 0;
      // f16x3, C = 4: s_conv1 + s_conv2 as one kernel (the activation between them never leaves the CU)
      if (ctx->leg_mode != 0 && li == 0 && (reinterpret_cast<uintptr_t>(cur) & 15) == 0 && ovn_leg_front_matches(ctx, li, h, w)) {
        OvnProfScope ps(ctx, OVN_K_LEG, stream);
        float* dst2 = buf[(li + 1) & 1];
        rc = ovn_leg_front_forward(ctx, li, cur, nb, h, w, dst2, &oh, &ow, ctx->actmax + (li + 2) * (size_t)slice * OVN_ACTMAX_STRIDE, stream);
        if (rc) return rc;
        cur = dst2;
        h = oh;
        w = ow;
        ++li;          // two layers done
        continue;
      }
      // f16x3: the six 1 x KW layers at the end run as one kernel with the activations kept in LDS
      if (ctx->leg_mode != 0 && ovn_leg_tail_matches(ctx, li, h, w)) {
        OvnProfScope ps(ctx, OVN_K_LEG, stream);
        rc = ovn_leg_tail_forward(ctx, li, cur, nb, w, features_dev + (size_t)s0 * OVN_FEAT_ELEMS, stream);
        if (rc) return rc;
        break;
      }
      {
        OvnProfScope ps(ctx, OVN_K_LEG, stream);
        rc = (ctx->leg_mode == 0) ? ovn_conv_forward(ctx->leg[li], cur, nb, h, w, dst, &oh, &ow, stream)
                                  : ovn_conv_forward_f16x3(ctx->leg[li], cur, nb, h, w, dst, &oh, &ow, ctx->actmax + li * (size_t)slice * OVN_ACTMAX_STRIDE,
                                                           last ? nullptr : ctx->actmax + (li + 1) * (size_t)slice * OVN_ACTMAX_STRIDE, stream);
      }
      if (rc) return rc;
      cur = dst;
      h = oh;
      w = ow;
    }
  }
  return OVN_OK;
}

int ovn_corr_head(ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r, const int32_t* ridx,
                  int64_t n, int32_t* yaw, float* corr, void* stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_corr_head: ctx is NULL");
  OVN_REQUIRE(n >= 0 && n < (1ll << 31), OVN_ERR_ARG, "ovn_corr_head: bad n");
  if (n == 0) return OVN_OK;
  OVN_REQUIRE(feats_l && feats_r && yaw, OVN_ERR_ARG, "ovn_corr_head: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  OvnProfScope ps(ctx, OVN_K_CORR, (hipStream_t)stream);
  return ovn_corr_forward(feats_l, lidx, feats_r, ridx, (int)n, yaw, corr, (hipStream_t)stream);
}

int ovn_spectrum(ovn_ctx* ctx, const float* feats_dev, int64_t n, float* spectra_dev, void* stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_spectrum: ctx is NULL");
  OVN_REQUIRE(n >= 0 && n < (1ll << 24), OVN_ERR_ARG, "ovn_spectrum: bad n");
  if (n == 0) return OVN_OK;
  OVN_REQUIRE(feats_dev && spectra_dev, OVN_ERR_ARG, "ovn_spectrum: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  OvnProfScope ps(ctx, OVN_K_SPECTRUM, (hipStream_t)stream);
  return ovn_spectrum_forward(ctx, feats_dev, (int)n, spectra_dev, (hipStream_t)stream);
}

int ovn_corr_head_spectral(ovn_ctx* ctx, const float* spec_l, const int32_t* lidx, const float* spec_r,
                           const int32_t* ridx, int64_t n, int32_t* yaw, float* corr, void* stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_corr_head_spectral: ctx is NULL");
  OVN_REQUIRE(n >= 0 && n < (1ll << 31), OVN_ERR_ARG, "ovn_corr_head_spectral: bad n");
  if (n == 0) return OVN_OK;
  OVN_REQUIRE(spec_l && spec_r && yaw, OVN_ERR_ARG, "ovn_corr_head_spectral: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  OvnProfScope ps(ctx, OVN_K_CORR_SPECTRAL, (hipStream_t)stream);
  return ovn_corr_spectral_forward(ctx, spec_l, lidx, spec_r, ridx, (int)n, yaw, corr, (hipStream_t)stream);
}

// ---- side streams of a head call -----------------------------------------------------------------------------------------------
// A head call may spread its launches over the caller's stream and two context-owned side streams: the HBM-bound yaw head next to
// the matrix-core-bound Delta kernels, and the sub-chunks of a sweep alternating between two streams so that the prepare / c_conv2 /
// c_conv3 kernels of one sub-chunk run beside the contraction kernel of the next.  Fork and join are events on the caller's stream:
// to the caller the call still behaves as if everything had been enqueued on `stream`.
static int head_streams_ready(ovn_ctx* ctx) {
  if (ctx->aux_ready) return OVN_OK;
  for (int i = 0; i < 2; ++i) {
    OVN_HIP_CHECK(hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking));
    OVN_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
  }
  OVN_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  ctx->aux_ready = true;
  return OVN_OK;
}

struct OvnFork {   // fork on construction-time request, join (on every exit path) in the destructor
  ovn_ctx* ctx;
  hipStream_t stream;
  bool used[2] = {false, false};
  bool forked = false;
  OvnFork(ovn_ctx* c, hipStream_t s) : ctx(c), stream(s) {}
  int fork() {
    if (forked) return OVN_OK;
    int rc = head_streams_ready(ctx);
    if (rc) return rc;
    OVN_HIP_CHECK(hipEventRecord(ctx->ev_fork, stream));
    forked = true;
    return OVN_OK;
  }
  // side stream i, ordered behind everything the caller had enqueued on `stream` when fork() ran
  int side(int i, hipStream_t* out) {
    int rc = fork();
    if (rc) return rc;
    if (!used[i]) {
      OVN_HIP_CHECK(hipStreamWaitEvent(ctx->aux[i], ctx->ev_fork, 0));
      used[i] = true;
    }
    *out = ctx->aux[i];
    return OVN_OK;
  }
  int join() {
    int rc = OVN_OK;
    for (int i = 0; i < 2; ++i)
      if (used[i]) {
        used[i] = false;
        if (hipEventRecord(ctx->ev_join[i], ctx->aux[i]) != hipSuccess || hipStreamWaitEvent(stream, ctx->ev_join[i], 0) != hipSuccess) {
          ovn_set_error("joining the head's side stream failed");
          rc = OVN_ERR_HIP;
        }
      }
    return rc;
  }
  ~OvnFork() { (void)join(); }
};

// Delta (overlap) head on n pairs [+ one of the correlation heads]; shared by ovn_heads, ovn_delta_head and ovn_heads_spectral.
//   corr_mode 0: none; 1: direct form on the feature volumes (per chunk, on the caller's stream, as ovn_heads always did);
//             2: spectral form on (spec_l, spec_r): ONE launch for all n pairs on a side stream, beside the Delta kernels.
// The sweep is cut into chunks of <= ctx->head_chunk pairs (the scratch is sized for one chunk) and every chunk into sub-chunks of
// ctx->head_sub pairs that alternate between ctx->head_streams streams (sub-chunk j of every chunk uses scratch region j and
// stream j % streams, so a region is only ever reused in stream order).
static int delta_head_run(ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r,
                          const int32_t* ridx, int64_t n, float* overlap, float* logit, int32_t* yaw, float* corr,
                          int corr_mode, const float* spec_l, const float* spec_r, const float* dcache_l, hipStream_t stream) {
  if (ctx->head_s != OVN_S) {   // general conv1size: fp32 generality path, chunked so that the scratch stays near 2 GB
    const size_t pb = ovn_delta_generic_pair_bytes(ctx->head_g);
    int64_t chunk = (int64_t)((2ull << 30) / pb);
    chunk = chunk < 1 ? 1 : (chunk > 1024 ? 1024 : chunk);
    const int64_t cmax = n < chunk ? n : chunk;
    int rc = ovn_ws_reserve(ctx, (size_t)cmax * pb + 1024, stream);
    if (rc) return rc;
    ctx->dbg_o2 = ctx->dbg_o3 = nullptr;
    ctx->dbg_partial = nullptr;
    ctx->dbg_o2max = nullptr;
    ctx->dbg_n = 0;
    if (corr_mode == 2) {
      OvnProfScope ps(ctx, OVN_K_CORR_SPECTRAL, stream);
      rc = ovn_corr_spectral_forward(ctx, spec_l, lidx, spec_r, ridx, (int)n, yaw, corr, stream);
      if (rc) return rc;
    }
    for (int64_t p0 = 0; p0 < n; p0 += chunk) {
      const int np = (int)((n - p0 < chunk) ? (n - p0) : chunk);
      const float* fl = lidx ? feats_l : feats_l + (size_t)p0 * OVN_FEAT_ELEMS;
      const int32_t* li = lidx ? lidx + p0 : nullptr;
      const int32_t* ri = ridx ? ridx + p0 : nullptr;
      if (corr_mode == 1) {
        OvnProfScope ps(ctx, OVN_K_CORR, stream);
        rc = ovn_corr_forward(fl, li, feats_r, ri, np, yaw + p0, corr ? corr + (size_t)p0 * OVN_FEAT_W : nullptr, stream);
        if (rc) return rc;
      }
      OvnProfScope ps(ctx, OVN_K_DELTA, stream);
      rc = ovn_delta_generic_forward(ctx, fl, li, feats_r, ri, np, ctx->ws, overlap + p0, logit ? logit + p0 : nullptr, stream);
      if (rc) return rc;
    }
    return OVN_OK;
  }
  const size_t o2_elems = (size_t)OVN_G * OVN_G * OVN_C2_OUT;   // 24*24*128 per pair
  const size_t o3_elems = (size_t)OVN_DENSE_IN;                 // 22*22*256 per pair
  const bool fused = (ctx->head_mode != 0);
  const int64_t chunk = ctx->head_chunk;                        // pairs per pass over the scratch (f16x3: 3.2 MB per pair)
  const int64_t cmax = n < chunk ? n : chunk;
  // sub-chunks: only the f16x3 kernels are split (the fp32 mode is one long kernel per chunk and keeps the round-2 structure)
  int64_t sub = (fused && ctx->head_sub > 0 && ctx->head_sub < cmax) ? ctx->head_sub : cmax;
  const int nsub_max = (int)((cmax + sub - 1) / sub);
  const int nstreams = (fused && nsub_max > 1 && ctx->head_streams > 1) ? 2 : 1;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t o2_bytes = al((size_t)cmax * o2_elems * sizeof(float));
  // second scratch region: o3 (n,22,22,256) in fp32 mode; in f16x3 mode c_conv3 and the Dense layer are one kernel and only
  // OVN_DENSE_PARTIALS partial sums per pair (band x half of the output channels x half of the m-tiles) leave it
  const size_t o3_bytes = fused ? al((size_t)cmax * OVN_DENSE_PARTIALS * sizeof(float)) : al((size_t)cmax * o3_elems * sizeof(float));
  // f16x3 mode: per-pair scales, packed volumes, linear terms and the c_conv1 rows between the two Delta kernels (2.9 MB per pair),
  // one self-contained block per sub-chunk
  const size_t sc_sub = fused ? al(ovn_delta_f16x3_scratch_bytes((int)sub, ridx != nullptr)) : 0;
  int rc = ovn_ws_reserve(ctx, o2_bytes + o3_bytes + sc_sub * nsub_max, stream);
  if (rc) return rc;
  if (fused && ctx->c3_arrived_n < chunk) {   // arrival counters of the fused c_conv3 + Dense kernel, one per pair of a chunk (sized
    OVN_HIP_CHECK(hipStreamSynchronize(stream));   // by the chunk, not by this call: a growing sweep must not re-allocate): zeroed
    if (ctx->c3_arrived) (void)hipFree(ctx->c3_arrived);   // once, left zeroed by every launch
    ctx->c3_arrived = nullptr;
    ctx->c3_arrived_n = 0;
    OVN_HIP_CHECK(hipMalloc((void**)&ctx->c3_arrived, (size_t)chunk * sizeof(unsigned)));
    OVN_HIP_CHECK(hipMemsetAsync(ctx->c3_arrived, 0, (size_t)chunk * sizeof(unsigned), stream));
    ctx->c3_arrived_n = chunk;
  }
  float* o2 = reinterpret_cast<float*>(ctx->ws);
  float* o3 = reinterpret_cast<float*>(static_cast<char*>(ctx->ws) + o2_bytes);
  char* dscratch = static_cast<char*>(ctx->ws) + o2_bytes + o3_bytes;
  ctx->dbg_o2max = nullptr;
  ctx->dbg_o2 = o2;
  ctx->dbg_o3 = fused ? nullptr : o3;
  ctx->dbg_partial = fused ? o3 : nullptr;
  ctx->dbg_n = sub < cmax ? sub : cmax;      // the activations of the first sub-chunk stay addressable for the tests

  OvnFork fk(ctx, stream);
  bool a2_in_yaw = false;
  if (corr_mode == 2) {
    hipStream_t ys = stream;
    if (ctx->head_yaw_side) {
      rc = fk.side(1, &ys);
      if (rc) return rc;
    }
    OvnProfScope ps(ctx, OVN_K_CORR_SPECTRAL, ys);
    // a small 1-vs-N sweep in one sub-chunk: the query's right-volume term of the Delta head rides in the yaw launch (csrc/delta_a2.h)
    a2_in_yaw = fused && ridx == nullptr && n <= OVN_A2_IN_YAW_MAX_PAIRS && n <= chunk && ys == stream && nsub_max == 1 && ctx->head_s == OVN_S;
    rc = ovn_corr_spectral_forward(ctx, spec_l, lidx, spec_r, ridx, (int)n, yaw, corr, ys, a2_in_yaw ? feats_r : nullptr,
                                   a2_in_yaw ? ovn_delta_f16x3_a2raw(dscratch, (int)n) : nullptr);
    if (rc) return rc;
  }
  for (int64_t c0 = 0; c0 < n; c0 += chunk) {
    const int64_t cn = (n - c0 < chunk) ? (n - c0) : chunk;
    if (corr_mode == 1) {
      OvnProfScope ps(ctx, OVN_K_CORR, stream);
      rc = ovn_corr_forward(lidx ? feats_l : feats_l + (size_t)c0 * OVN_FEAT_ELEMS, lidx ? lidx + c0 : nullptr, feats_r,
                            ridx ? ridx + c0 : nullptr, (int)cn, yaw + c0, corr ? corr + (size_t)c0 * OVN_FEAT_W : nullptr, stream);
      if (rc) return rc;
    }
    int j = 0;
    for (int64_t q0 = 0; q0 < cn; q0 += sub, ++j) {
      const int64_t p0 = c0 + q0;
      const int np = (int)((cn - q0 < sub) ? (cn - q0) : sub);
      hipStream_t st = stream;
      if (nstreams == 2 && (j & 1)) {
        rc = fk.side(0, &st);
        if (rc) return rc;
      }
      const float* fl = lidx ? feats_l : feats_l + (size_t)p0 * OVN_FEAT_ELEMS;
      const int32_t* li = lidx ? lidx + p0 : nullptr;
      const int32_t* ri = ridx ? ridx + p0 : nullptr;
      float* o2s = o2 + (size_t)q0 * o2_elems;
      if (fused) {   // times its prepare kernels, the contraction kernel and c_conv2 separately
        unsigned* o2max = nullptr;
        float* part = o3 + (size_t)q0 * OVN_DENSE_PARTIALS;
        rc = ovn_delta_c12_f16x3_forward(ctx, fl, li, feats_r, ri, np, dscratch + (size_t)j * sc_sub, &o2max, o2s, st, (int)(p0 & 0x3fffffff),
                                          // a cache row belongs to a CANDIDATE: without an index list it moves with the feature pointer
                                          dcache_l ? (lidx ? dcache_l : dcache_l + (size_t)p0 * OVN_DELTA_CACHE_ELEMS) : nullptr, a2_in_yaw);
        if (rc) return rc;
        if (p0 == 0) ctx->dbg_o2max = o2max;
        {   // c_conv3 + Flatten + Dense + sigmoid: one launch (the pair's last workgroup finishes it)
          OvnProfScope ps(ctx, OVN_K_C3, st);
          rc = ovn_c3_dense_forward(ctx, o2s, o2max, np, part, nullptr, ctx->c3_arrived + q0, overlap + p0, logit ? logit + p0 : nullptr, st);
        }
      } else {
        float* o3s = o3 + (size_t)q0 * o3_elems;
        {
          OvnProfScope ps(ctx, OVN_K_DELTA, st);
          rc = ovn_delta_c12_forward(ctx, fl, li, feats_r, ri, np, o2s, st);
        }
        if (rc) return rc;
        int oh = 0, ow = 0;
        {
          OvnProfScope ps(ctx, OVN_K_C3, st);
          rc = ovn_conv_forward(ctx->c3, o2s, np, OVN_G, OVN_G, o3s, &oh, &ow, st);
        }
        if (rc) return rc;
        OvnProfScope ps(ctx, OVN_K_DENSE, st);
        rc = ovn_dense_sigmoid_forward(ctx, o3s, np, overlap + p0, logit ? logit + p0 : nullptr, st);
      }
      if (rc) return rc;
    }
  }
  return fk.join();
}

int ovn_heads(ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r, const int32_t* ridx,
              int64_t n, float* overlap, int32_t* yaw, float* logit, float* corr, void* stream_) {
  OVN_REQUIRE(ctx && ctx->head_set, OVN_ERR_STATE, "ovn_heads: head weights not set");
  OVN_REQUIRE(n >= 0 && n < (1ll << 31), OVN_ERR_ARG, "ovn_heads: bad n");
  if (n == 0) return OVN_OK;
  OVN_REQUIRE(feats_l && feats_r && overlap && yaw, OVN_ERR_ARG, "ovn_heads: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  return delta_head_run(ctx, feats_l, lidx, feats_r, ridx, n, overlap, logit, yaw, corr, 1, nullptr, nullptr, nullptr, (hipStream_t)stream_);
}

int ovn_heads_spectral(ovn_ctx* ctx, const float* feats_l, const float* spec_l, const float* dcache_l, const int32_t* lidx,
                       const float* feats_r, const float* spec_r, const int32_t* ridx, int64_t n, float* overlap, int32_t* yaw,
                       float* logit, float* corr, void* stream_) {
  OVN_REQUIRE(ctx && ctx->head_set, OVN_ERR_STATE, "ovn_heads_spectral: head weights not set");
  OVN_REQUIRE(n >= 0 && n < (1ll << 31), OVN_ERR_ARG, "ovn_heads_spectral: bad n");
  if (n == 0) return OVN_OK;
  OVN_REQUIRE(feats_l && feats_r && spec_l && spec_r && overlap && yaw, OVN_ERR_ARG, "ovn_heads_spectral: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  return delta_head_run(ctx, feats_l, lidx, feats_r, ridx, n, overlap, logit, yaw, corr, 2, spec_l, spec_r,
                        ctx->head_mode != 0 ? dcache_l : nullptr, (hipStream_t)stream_);
}

int ovn_delta_head(ovn_ctx* ctx, const float* feats_l, const int32_t* lidx, const float* feats_r, const int32_t* ridx,
                   int64_t n, float* overlap, float* logit, void* stream_) {
  OVN_REQUIRE(ctx && ctx->head_set, OVN_ERR_STATE, "ovn_delta_head: head weights not set");
  OVN_REQUIRE(n >= 0 && n < (1ll << 31), OVN_ERR_ARG, "ovn_delta_head: bad n");
  if (n == 0) return OVN_OK;
  OVN_REQUIRE(feats_l && feats_r && overlap, OVN_ERR_ARG, "ovn_delta_head: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  return delta_head_run(ctx, feats_l, lidx, feats_r, ridx, n, overlap, logit, nullptr, nullptr, 0, nullptr, nullptr, nullptr, (hipStream_t)stream_);
}

int ovn_delta_cache(ovn_ctx* ctx, const float* feats_dev, int64_t n, float* cache_dev, void* stream) {
  OVN_REQUIRE(ctx && ctx->head_set, OVN_ERR_STATE, "ovn_delta_cache: head weights not set");
  OVN_REQUIRE(n >= 0 && n < (1ll << 24), OVN_ERR_ARG, "ovn_delta_cache: bad n");
  if (n == 0) return OVN_OK;
  OVN_REQUIRE(feats_dev && cache_dev, OVN_ERR_ARG, "ovn_delta_cache: NULL buffer");
  OVN_REQUIRE((reinterpret_cast<uintptr_t>(cache_dev) & 15) == 0, OVN_ERR_ARG, "ovn_delta_cache: cache_dev must be 16-byte aligned");
  OVN_REQUIRE(ctx->head_s == OVN_S, OVN_ERR_STATE, "ovn_delta_cache: only the default head geometry (conv1size 15) has a Delta cache");
  OVN_ON_DEVICE(ctx->device);
  OvnProfScope ps(ctx, OVN_K_DELTA_PREP, (hipStream_t)stream);
  return ovn_delta_cache_forward(ctx, feats_dev, (int)n, cache_dev, (hipStream_t)stream);
}

int ovn_set_head_geometry(ovn_ctx* ctx, int conv1size) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_set_head_geometry: ctx is NULL");
  OVN_REQUIRE(!ctx->head_set, OVN_ERR_STATE, "ovn_set_head_geometry: call it before ovn_set_head_weights");
  OVN_REQUIRE(conv1size >= 1 && OVN_FEAT_W / conv1size >= 3, OVN_ERR_ARG,
              "ovn_set_head_geometry: conv1size %d leaves fewer than 3 x 3 groups of the 360 columns for c_conv3", conv1size);
  ctx->head_s = conv1size;
  ctx->head_g = OVN_FEAT_W / conv1size;
  return OVN_OK;
}

int ovn_set_head_pipeline(ovn_ctx* ctx, int64_t chunk_pairs, int64_t sub_chunk_pairs, int streams, int yaw_on_side_stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_set_head_pipeline: ctx is NULL");
  OVN_REQUIRE(chunk_pairs >= 1 && chunk_pairs <= (1 << 20), OVN_ERR_ARG, "ovn_set_head_pipeline: chunk_pairs %lld", (long long)chunk_pairs);
  OVN_REQUIRE(sub_chunk_pairs >= 0, OVN_ERR_ARG, "ovn_set_head_pipeline: sub_chunk_pairs %lld", (long long)sub_chunk_pairs);
  OVN_REQUIRE(streams == 1 || streams == 2, OVN_ERR_ARG, "ovn_set_head_pipeline: streams %d (1 or 2)", streams);
  ctx->head_chunk = chunk_pairs;
  ctx->head_sub = sub_chunk_pairs;
  ctx->head_streams = streams;
  ctx->head_yaw_side = yaw_on_side_stream ? 1 : 0;
  return OVN_OK;
}

int ovn_get_head_pipeline(ovn_ctx* ctx, int64_t* chunk_pairs, int64_t* sub_chunk_pairs, int* streams, int* yaw_on_side_stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_get_head_pipeline: ctx is NULL");
  if (chunk_pairs) *chunk_pairs = ctx->head_chunk;
  if (sub_chunk_pairs) *sub_chunk_pairs = ctx->head_sub;
  if (streams) *streams = ctx->head_streams;
  if (yaw_on_side_stream) *yaw_on_side_stream = ctx->head_yaw_side;
  return OVN_OK;
}

int ovn_best_match(ovn_ctx* ctx, const float* overlap, const int32_t* yaw, const int32_t* ids, int64_t n, float threshold,
                   int64_t index_offset, int32_t* out, void* stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_best_match: ctx is NULL");
  OVN_REQUIRE(n >= 0 && n < (1ll << 31), OVN_ERR_ARG, "ovn_best_match: bad n");
  OVN_REQUIRE(index_offset >= 0 && index_offset + n < (1ll << 31), OVN_ERR_ARG, "ovn_best_match: bad index_offset");
  OVN_REQUIRE(out != nullptr && (n == 0 || overlap != nullptr), OVN_ERR_ARG, "ovn_best_match: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  return ovn_best_match_forward(overlap, yaw, ids, (int)n, threshold, (int)index_offset, out, (hipStream_t)stream);
}

int ovn_project(ovn_ctx* ctx, const float* points_dev, const int64_t* offsets_dev, int n_scans,
                int64_t max_points_per_scan, int proj_h, int proj_w, double fov_up_deg, double fov_down_deg,
                double max_range, float* range_dev, float* vertex_dev, float* intensity_dev, int32_t* idx_dev,
                float* normal_dev, float* stacked_dev, int use_depth, int use_normals, int use_intensity, void* stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_project: ctx is NULL");
  OVN_REQUIRE(n_scans == 0 || (points_dev || max_points_per_scan == 0), OVN_ERR_ARG, "ovn_project: points is NULL");
  OVN_REQUIRE(n_scans == 0 || offsets_dev, OVN_ERR_ARG, "ovn_project: offsets is NULL");
  OVN_ON_DEVICE(ctx->device);
  OvnProfScope ps(ctx, OVN_K_PROJ, (hipStream_t)stream);
  return ovn_project_forward(ctx, points_dev, offsets_dev, n_scans, max_points_per_scan, proj_h, proj_w, fov_up_deg,
                             fov_down_deg, max_range, range_dev, vertex_dev, intensity_dev, idx_dev, normal_dev,
                             stacked_dev, use_depth, use_normals, use_intensity, (hipStream_t)stream);
}

int ovn_projection_angles(ovn_ctx* ctx, const float* points_dev, int64_t n_points, int proj_h, int proj_w, double fov_up_deg,
                          double fov_down_deg, double max_range, float* yaw_dev, float* pitch_dev, int32_t* pixel_dev,
                          void* stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_projection_angles: ctx is NULL");
  OVN_REQUIRE(n_points >= 0 && proj_h > 0 && proj_w > 0, OVN_ERR_ARG, "ovn_projection_angles: bad sizes");
  if (n_points == 0) return OVN_OK;
  OVN_REQUIRE(points_dev != nullptr, OVN_ERR_ARG, "ovn_projection_angles: points is NULL");
  OVN_ON_DEVICE(ctx->device);
  return ovn_projection_angles_forward(points_dev, n_points, proj_h, proj_w, fov_up_deg, fov_down_deg, max_range, yaw_dev,
                                       pitch_dev, pixel_dev, (hipStream_t)stream, ctx->proj_trig);
}

int ovn_normals(ovn_ctx* ctx, const float* range_dev, const float* vertex_dev, int n_scans, int proj_h, int proj_w,
                float* normal_dev, void* stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_normals: ctx is NULL");
  OVN_REQUIRE(n_scans >= 0 && proj_h > 0 && proj_w > 0, OVN_ERR_ARG, "ovn_normals: bad sizes");
  if (n_scans == 0) return OVN_OK;
  OVN_REQUIRE(range_dev && vertex_dev && normal_dev, OVN_ERR_ARG, "ovn_normals: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  return ovn_normals_forward(range_dev, vertex_dev, n_scans, proj_h, proj_w, normal_dev, (hipStream_t)stream);
}

int ovn_gt_range_images(ovn_ctx* ctx, const float* points_dev, const int64_t* offsets_dev, int n_scans,
                        int64_t max_points_per_scan, const double* ref_poses_dev, const double* inv_cur_pose_dev, int proj_h,
                        int proj_w, double fov_up_deg, double fov_down_deg, double max_range, float* range_dev, void* stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_gt_range_images: ctx is NULL");
  OVN_REQUIRE(n_scans >= 0 && proj_h > 0 && proj_w > 0 && max_points_per_scan >= 0, OVN_ERR_ARG, "ovn_gt_range_images: bad sizes");
  if (n_scans == 0) return OVN_OK;
  OVN_REQUIRE(offsets_dev && range_dev && (points_dev || max_points_per_scan == 0), OVN_ERR_ARG, "ovn_gt_range_images: NULL buffer");
  OVN_REQUIRE(n_scans <= 65535, OVN_ERR_ARG, "ovn_gt_range_images: at most 65535 scans per call");
  OVN_ON_DEVICE(ctx->device);
  return ovn_gt_range_forward(points_dev, offsets_dev, n_scans, max_points_per_scan, ref_poses_dev, inv_cur_pose_dev, proj_h,
                              proj_w, fov_up_deg, fov_down_deg, max_range, range_dev, (hipStream_t)stream);
}

int ovn_gt_overlap_counts(ovn_ctx* ctx, const float* ref_ranges_dev, const float* cur_range_dev, int n_scans, int proj_h,
                          int proj_w, int32_t* counts_dev, void* stream) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_gt_overlap_counts: ctx is NULL");
  OVN_REQUIRE(n_scans >= 0 && proj_h > 0 && proj_w > 0, OVN_ERR_ARG, "ovn_gt_overlap_counts: bad sizes");
  OVN_REQUIRE(cur_range_dev && counts_dev && (ref_ranges_dev || n_scans == 0), OVN_ERR_ARG, "ovn_gt_overlap_counts: NULL buffer");
  OVN_ON_DEVICE(ctx->device);
  return ovn_gt_count_forward(ref_ranges_dev, cur_range_dev, n_scans, proj_h * proj_w, counts_dev, (hipStream_t)stream);
}

int ovn_set_head_precision(ovn_ctx* ctx, int mode) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_set_head_precision: ctx is NULL");
  OVN_REQUIRE(mode == 0 || mode == 1, OVN_ERR_ARG, "ovn_set_head_precision: mode %d (0 = fp32 MFMA, 1 = f16x3 MFMA)", mode);
  ctx->head_mode = mode;
  return OVN_OK;
}

int ovn_set_head_compaction(ovn_ctx* ctx, int on) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_set_head_compaction: ctx is NULL");
  OVN_REQUIRE(on == 0 || on == 1, OVN_ERR_ARG, "ovn_set_head_compaction: %d (0 = walk all 128 channels, 1 = drop the query's dead channels)", on);
  ctx->head_compact = on;
  return OVN_OK;
}

int ovn_set_projection_trig(ovn_ctx* ctx, int mode) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_set_projection_trig: ctx is NULL");
  OVN_REQUIRE(mode == 0 || mode == 1, OVN_ERR_ARG, "ovn_set_projection_trig: mode %d (0 = NumPy / SVML float32, 1 = correctly rounded)", mode);
  ctx->proj_trig = mode;
  return OVN_OK;
}

int ovn_set_leg_precision(ovn_ctx* ctx, int mode) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_set_leg_precision: ctx is NULL");
  OVN_REQUIRE(mode == 0 || mode == 1, OVN_ERR_ARG, "ovn_set_leg_precision: mode %d (0 = fp32 MFMA, 1 = f16x3 MFMA)", mode);
  ctx->leg_mode = mode;
  return OVN_OK;
}

int ovn_profile_begin(ovn_ctx* ctx) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_profile_begin: ctx is NULL");
  for (auto& r : ctx->prof_recs) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  ctx->prof_recs.clear();
  ctx->prof = true;
  return OVN_OK;
}

int ovn_profile_end(ovn_ctx* ctx, double* ms_by_kind, int64_t* launches_by_kind) {
  OVN_REQUIRE(ctx && ms_by_kind && launches_by_kind, OVN_ERR_ARG, "ovn_profile_end: NULL argument");
  OVN_ON_DEVICE(ctx->device);
  ctx->prof = false;
  for (int k = 0; k < OVN_K_COUNT; ++k) {
    ms_by_kind[k] = 0.0;
    launches_by_kind[k] = 0;
  }
  int rc = OVN_OK;
  for (auto& r : ctx->prof_recs) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(r.b);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) {
      ovn_set_error("ovn_profile_end: %s", hipGetErrorString(e));
      rc = OVN_ERR_HIP;
    } else {
      ms_by_kind[r.kind] += ms;
      launches_by_kind[r.kind] += 1;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  ctx->prof_recs.clear();
  return rc;
}

int ovn_debug_conv(ovn_ctx* ctx, int layer, const float* in_dev, int nb, int h, int w, float* out_dev, void* stream) {
  OVN_REQUIRE(ctx && layer >= 0 && layer < (int)ctx->leg.size(), OVN_ERR_ARG, "ovn_debug_conv: no such leg layer");
  OVN_REQUIRE(in_dev && out_dev && nb >= 0, OVN_ERR_ARG, "ovn_debug_conv: bad buffers");
  OVN_ON_DEVICE(ctx->device);
  int oh = 0, ow = 0;
  if (ctx->leg_mode == 0) return ovn_conv_forward(ctx->leg[layer], in_dev, nb, h, w, out_dev, &oh, &ow, (hipStream_t)stream);
  OVN_REQUIRE(nb <= OVN_LEG_SLICE, OVN_ERR_ARG, "ovn_debug_conv: at most %d images per call", OVN_LEG_SLICE);
  if (!ctx->actmax) OVN_HIP_CHECK(hipMalloc((void**)&ctx->actmax, (size_t)OVN_ACTMAX_SLOTS * OVN_LEG_SLICE * OVN_ACTMAX_STRIDE * sizeof(unsigned)));
  OVN_HIP_CHECK(hipMemsetAsync(ctx->actmax, 0, 2 * (size_t)OVN_LEG_SLICE * OVN_ACTMAX_STRIDE * sizeof(unsigned), (hipStream_t)stream));
  int rc = ovn_absmax_forward(in_dev, nb, (long long)h * w * ctx->leg[layer].cin, ctx->actmax, (hipStream_t)stream);
  if (rc) return rc;
  return ovn_conv_forward_f16x3(ctx->leg[layer], in_dev, nb, h, w, out_dev, &oh, &ow, ctx->actmax, ctx->actmax + (size_t)OVN_LEG_SLICE * OVN_ACTMAX_STRIDE,
                                (hipStream_t)stream);
}

int ovn_debug_head_activations(ovn_ctx* ctx, int64_t n, float* o2_dev, float* o3_dev, void* stream) {
  OVN_REQUIRE(ctx && ctx->dbg_o2 && n >= 0 && n <= ctx->dbg_n, OVN_ERR_STATE,
              "ovn_debug_head_activations: call right after ovn_heads with n <= its (first-chunk) pair count");
  OVN_ON_DEVICE(ctx->device);
  if (o2_dev)
    OVN_HIP_CHECK(hipMemcpyAsync(o2_dev, ctx->dbg_o2, (size_t)n * OVN_G * OVN_G * OVN_C2_OUT * sizeof(float),
                                 hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (o3_dev && ctx->dbg_o3)
    OVN_HIP_CHECK(hipMemcpyAsync(o3_dev, ctx->dbg_o3, (size_t)n * OVN_DENSE_IN * sizeof(float), hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
  else if (o3_dev)  // f16x3 mode: o3 never left the fused kernel -- run it again on the o2 still in scratch, with o3 output
    return ovn_c3_dense_forward(ctx, ctx->dbg_o2, ctx->dbg_o2max, (int)n, ctx->dbg_partial, o3_dev, nullptr, nullptr, nullptr, (hipStream_t)stream);
  return OVN_OK;
}

int ovn_head_walk_stats(ovn_ctx* ctx, int32_t* out16_host, void* stream) {
  OVN_REQUIRE(ctx && out16_host, OVN_ERR_ARG, "ovn_head_walk_stats: NULL argument");
  OVN_ON_DEVICE(ctx->device);
  return ovn_delta_walk_stats(ctx, out16_host, (hipStream_t)stream);
}

int64_t ovn_workspace_bytes(ovn_ctx* ctx) { return ctx ? (int64_t)ctx->ws_bytes : 0; }

int ovn_selftest(ovn_ctx* ctx) {
  OVN_REQUIRE(ctx != nullptr, OVN_ERR_ARG, "ovn_selftest: ctx is NULL");
  OVN_ON_DEVICE(ctx->device);
  return ovn_mfma_selftest(nullptr);
}

}  // extern "C"
