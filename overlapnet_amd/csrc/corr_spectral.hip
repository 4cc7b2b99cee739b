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
//   * ovn_spectrum:  the DFT is a dense contraction with a constant 360 x 368 twiddle matrix -> run on the fp32
//     matrix cores through the generic conv kernel (a 360x1 'valid' convolution over the (360,128) feature image).
//   * spectral_product_kernel: one workgroup per pair, thread = (4 consecutive frequencies, 32 channels); partial
//     sums of the 4 channel groups are combined in LDS in a fixed order (deterministic).
//   * the inverse transform of the 368-vector C^ is again a constant contraction (1x1 conv, 368 -> 368) and the
//     argmax (first maximum wins) is a fixed-order wave reduction.
#include <math.h>

#include <vector>

#include "ovn_internal.h"

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

// C^ for one pair: out[pair][f] = Re, out[pair][184 + f] = Im, padding zero.
__global__ __launch_bounds__(PROD_THREADS) void spectral_product_kernel(const float* __restrict__ spec_l,
                                                                       const int32_t* __restrict__ lidx,
                                                                       const float* __restrict__ spec_r,
                                                                       const int32_t* __restrict__ ridx,
                                                                       float* __restrict__ chat) {
  __shared__ float part[CG][2][IM_OFF];
  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const float* L = spec_l + (long long)(lidx ? lidx[pair] : pair) * OVN_SPEC_ELEMS;
  const float* R = spec_r + (long long)(ridx ? ridx[pair] : 0) * OVN_SPEC_ELEMS;
  if (tid < FQ * CG) {
    const int fq = tid % FQ;
    const int cg = tid / FQ;
    f32x4 sre = {0.f, 0.f, 0.f, 0.f}, sim = {0.f, 0.f, 0.f, 0.f};
    const float* lrow = L + (cg * 32) * SW + 4 * fq;
    const float* rrow = R + (cg * 32) * SW + 4 * fq;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(lrow + c * SW);            // Re L^
      const f32x4 b = *reinterpret_cast<const f32x4*>(lrow + c * SW + IM_OFF);   // Im L^
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
  if (tid < IM_OFF) {
    float re = 0.f, im = 0.f;
    if (tid < NF) {
      re = (part[0][0][tid] + part[1][0][tid]) + (part[2][0][tid] + part[3][0][tid]);
      im = (part[0][1][tid] + part[1][1][tid]) + (part[2][1][tid] + part[3][1][tid]);
    }
    chat[(long long)pair * SW + tid] = re;
    chat[(long long)pair * SW + IM_OFF + tid] = im;
  }
}

// yaw = 180 - argmax (first maximum wins) over corr[pair][0..359]; optionally copies the 360 values out.
__global__ __launch_bounds__(256) void corr_argmax_kernel(const float* __restrict__ corr368, int32_t* __restrict__ yaw,
                                                          float* __restrict__ corr_out) {
  __shared__ float rv[4];
  __shared__ int ri[4];
  const int pair = blockIdx.x;
  const int tid = threadIdx.x;
  const float* c = corr368 + (long long)pair * SW;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int k = tid; k < FW; k += 256) {
    const float v = c[k];
    if (corr_out) corr_out[(long long)pair * FW + k] = v;
    if (v > bv || (v == bv && k < bi)) {
      bv = v;
      bi = k;
    }
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
    for (int w = 1; w < 4; ++w)
      if (rv[w] > v || (rv[w] == v && ri[w] < i)) {
        v = rv[w];
        i = ri[w];
      }
    yaw[pair] = FW / 2 - i;
  }
}

int upload_layer(OvnConvLayer* L, const std::vector<float>& w, hipStream_t stream) {
  float* dw = nullptr;
  float* db = nullptr;
  OVN_HIP_CHECK(hipMalloc((void**)&dw, w.size() * sizeof(float)));
  OVN_HIP_CHECK(hipMalloc((void**)&db, (size_t)L->cout * sizeof(float)));
  OVN_HIP_CHECK(hipMemcpyAsync(dw, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  OVN_HIP_CHECK(hipMemsetAsync(db, 0, (size_t)L->cout * sizeof(float), stream));
  int rc = ovn_conv_prepare(L, dw, db, stream);  // synchronises the stream
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
    int rc = upload_layer(&L, w, stream);
    if (rc) return rc;
  }
  {  // inverse transform of the Hermitian half, shifted by W/2 (RangePadding2D), as a (1,1,368,368) convolution
    OvnConvLayer& L = ctx->idft;
    L = OvnConvLayer();
    L.name = "idft360";
    L.kh = 1;
    L.kw = 1;
    L.cin = SW;
    L.cout = SWP;
    L.out_cols = SW;
    L.sh = 1;
    L.sw = 1;
    L.relu = 0;
    std::vector<float> w((size_t)SW * SWP, 0.f);
    for (int f = 0; f < NF; ++f) {
      const double wf = ((f == 0 || f == FW / 2) ? 1.0 : 2.0) / FW;
      for (int k = 0; k < FW; ++k) {
        const double ang = w0 * (double)((long long)f * (k + FW / 2) % FW);
        w[(size_t)f * SWP + k] = (float)(wf * cos(ang));
        w[(size_t)(IM_OFF + f) * SWP + k] = (float)(-wf * sin(ang));
      }
    }
    int rc = upload_layer(&L, w, stream);
    if (rc) return rc;
  }
  return OVN_OK;
}

int ovn_spectrum_forward(ovn_ctx* ctx, const float* feats, int n, float* spectra, hipStream_t stream) {
  int oh = 0, ow = 0;
  // input viewed as (n, H=360, W=128, C=1): out (n, 1, 128, 368) = spectra (n, 128, 368)
  return ovn_conv_forward(ctx->dft, feats, n, FW, FC, spectra, &oh, &ow, stream);
}

int ovn_corr_spectral_forward(ovn_ctx* ctx, const float* spec_l, const int32_t* lidx, const float* spec_r,
                              const int32_t* ridx, int n, int32_t* yaw, float* corr, hipStream_t stream) {
  const size_t vec_bytes = ((size_t)n * SW * sizeof(float) + 255) & ~(size_t)255;
  int rc = ovn_ws_reserve(ctx, 2 * vec_bytes, stream);
  if (rc) return rc;
  float* chat = reinterpret_cast<float*>(ctx->ws);
  float* c368 = reinterpret_cast<float*>(static_cast<char*>(ctx->ws) + vec_bytes);
  hipLaunchKernelGGL(spectral_product_kernel, dim3(n), dim3(PROD_THREADS), 0, stream, spec_l, lidx, spec_r, ridx, chat);
  OVN_HIP_CHECK(hipGetLastError());
  int oh = 0, ow = 0;
  rc = ovn_conv_forward(ctx->idft, chat, n, 1, 1, c368, &oh, &ow, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(corr_argmax_kernel, dim3(n), dim3(256), 0, stream, c368, yaw, corr);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
