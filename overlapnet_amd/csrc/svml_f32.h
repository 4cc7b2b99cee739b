// float32 atan2 / asin with the bits NumPy produces for the reference's range_projection (src/utils/utils.py:86-87).
//
// NumPy's float32 arctan2 / arcsin loops run Intel SVML's __svml_atan2f16 / __svml_asinf16 on AVX512_SKX x86-64 CPUs (the machine
// the golden vectors are generated on).  Those kernels are 1-4 ulp approximations built on VRCP14PS / VRSQRT14PS + FMA
// refinement; against a correctly rounded atan2 / asin (rounds 1-3 of this kernel: float64 functions rounded to float32) a point
// lands in a neighbouring pixel about once per 200 k points (measured: 12 range pixels over 24 transformed KITTI clouds).  This
// header evaluates the SAME operation sequence on the GPU: every vfmadd is one fmaf (v_fma_f32), every vmulps / vaddps one
// rounded float32 operation (compile with -ffp-contract=off), the constants are the kernels' data tables and the two 14-bit
// hardware approximations come from approx14_tables.h (integer tables that reproduce the instructions for every input).
// The lanes SVML hands to its scalar call-out (an argument that is 0 / NaN / outside [2^-125, 2^123); |x| > 1) take exact
// special values or the float64 function.  Checked against NumPy itself through the CPU twin of this file
// (oracle/svml_f32.c, tests/test_oracle_svml.py) and on the GPU against the reference's own outputs (tests/test_gpu_parity.py).
#pragma once
#define OVN_APPROX14_QUAL __device__
#include "approx14_tables.h"

namespace ovn_svml {

__device__ __forceinline__ float rcp14(float x) {          // VRCP14PS, normal x with a normal reciprocal
  const unsigned u = __float_as_uint(x), sign = u & 0x80000000u, m = u & 0x7fffffu;
  const int e = (int)((u >> 23) & 0xff);
  if (m == 0) return __uint_as_float(sign | (unsigned)(254 - e) << 23);
  const int i = (int)(m >> 17), lo = (int)((m >> 7) & 1023);
  const unsigned v = (unsigned)((OVN_RCP14_A[i] - OVN_RCP14_B[i] * lo) >> 9);
  return __uint_as_float(sign | (unsigned)(253 - e) << 23 | (v & 0xffffu) << 7);
}

__device__ __forceinline__ float rsqrt14(float x) {        // VRSQRT14PS, normal x > 0
  const unsigned u = __float_as_uint(x), m = u & 0x7fffffu;
  const int e = (int)((u >> 23) & 0xff) - 127;
  const int par = e & 1;
  const int h = (e - par) / 2;
  if (m == 0 && par == 0) return __uint_as_float((unsigned)(127 - h) << 23);
  const int i = par << 5 | (int)(m >> 18), lo = (int)((m >> 8) & 1023);
  const unsigned v = (unsigned)((OVN_RSQRT14_A[i] - OVN_RSQRT14_B[i] * lo) >> 9);
  return __uint_as_float((unsigned)(126 - h) << 23 | (v & 0xffffu) << 7);
}

__device__ __forceinline__ float atan2f_np(float y, float x) {
  const unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
  const unsigned ax = ux & 0x7fffffffu, ay = uy & 0x7fffffffu;
  if (ax - 0x01000000u >= 0x7c000000u || ay - 0x01000000u >= 0x7c000000u) {
    // SVML's scalar call-out.  A coordinate that is exactly 0 has an exact answer; the rest (denormals, > 2^123, NaN) cannot
    // come out of a LiDAR and take the float64 function
    const float PI = __uint_as_float(0x40490fdbu), PI_2 = __uint_as_float(0x3fc90fdbu);
    if (ax <= 0x7f800000u && ay <= 0x7f800000u && (ax == 0u || ay == 0u)) {
      float r;
      if (ay == 0u) r = (ux & 0x80000000u) ? PI : 0.0f;      // atan2(+-0, x): 0 for x >= +0, pi for x <= -0
      else r = PI_2;                                          // atan2(y != 0, +-0)
      return __uint_as_float(__float_as_uint(r) | (uy & 0x80000000u));
    }
    return (float)atan2((double)y, (double)x);
  }
  const float fx = __uint_as_float(ax), fy = __uint_as_float(ay);
  const bool small = fy < fx;
  const float num = small ? fy : -fx;
  const float den = small ? fx : fy;
  const float off = small ? 0.0f : __uint_as_float(0x3fc90fdbu);
  float r = rcp14(den);
  const float e = fmaf(-den, r, 1.0f);
  r = fmaf(e, r, r);
  const float q0 = num * r;
  const float rem = fmaf(-den, q0, num);
  const float q = fmaf(rem, r, q0);
  const float s = q * q;
  const float s2 = s * s;
  float ev = fmaf(s2, __uint_as_float(0x3b322cc0u), __uint_as_float(0x3d2bc384u));
  ev = fmaf(s2, ev, __uint_as_float(0x3dd96474u));
  float od = fmaf(s2, __uint_as_float(0xbc7f2631u), __uint_as_float(0xbd987629u));
  od = fmaf(s2, od, __uint_as_float(0xbe1161f8u));
  ev = fmaf(s2, ev, __uint_as_float(0x3e4cb79fu));
  od = fmaf(s2, od, __uint_as_float(0xbeaaaa49u));
  ev = fmaf(s2, ev, 1.0f);
  od = fmaf(s, od, ev);
  float res = fmaf(q, od, off);
  res = __uint_as_float(__float_as_uint(res) | (ux & 0x80000000u));
  if (x <= 0.0f) res = res + __uint_as_float(0x40490fdbu);
  return __uint_as_float(__float_as_uint(res) | (uy & 0x80000000u));
}

__device__ __forceinline__ float asinf_np(float x) {
  const unsigned ux = __float_as_uint(x);
  const float a = __uint_as_float(ux & 0x7fffffffu);
  if (!(a <= 1.0f)) return (float)asin((double)x);
  const bool big = !(a < 0.5f);
  const float t = fmaf(-a, 0.5f, 0.5f);
  const float z = big ? t : a * a;
  float base = a;
  if (big) {
    const float rs = (t < __uint_as_float(0x2f800000u)) ? 0.0f : rsqrt14(t);
    const float t2 = t + t;
    const float rr = rs * rs;
    const float sq = t2 * rs;
    const float d = fmaf(rr, t2, -2.0f);
    const float sd = sq * d;
    const float c = fmaf(d, __uint_as_float(0xbdc00004u), __uint_as_float(0x3e800001u));
    base = fmaf(sd, c, -sq);
  }
  float p1 = fmaf(z, __uint_as_float(0x3d3a9ab4u), __uint_as_float(0x3d997c12u));
  float p = fmaf(z, __uint_as_float(0x3d2edc07u), __uint_as_float(0x3cc32a6bu));
  const float zz = z * z;
  p = fmaf(p, zz, p1);
  p = fmaf(p, z, __uint_as_float(0x3e2aaaffu));
  p = z * p;
  float res = fmaf(p, base, base);
  if (big) res = res + __uint_as_float(0x3fc90fdbu);
  return __uint_as_float(__float_as_uint(res) ^ (ux & 0x80000000u));
}

}  // namespace ovn_svml
