/* TEST INFRASTRUCTURE (oracle) -- never linked into or called by the product.
 *
 * Restatement of NumPy's float32 `arctan2` / `arcsin` as the reference's range_projection reaches them
 * (src/utils/utils.py:86-87: `yaw = -np.arctan2(scan_y, scan_x)`, `pitch = np.arcsin(scan_z / depth)` on float32 arrays).
 * Third-party dependency: NumPy (the reference pins none; 2.2.6 here), whose x86-64 wheels dispatch these two ufunc loops on
 * AVX512_SKX CPUs to Intel SVML (numpy/SVML, BSD-3): `__svml_atan2f16` / `__svml_asinf16` ("la" kernels,
 * linux/avx512/svml_z0_atan2_s_la.s / svml_z0_asin_s_la.s).  The operation sequence below follows those kernels' main paths
 * instruction by instruction (every vfmadd is one fmaf, every vmulps/vaddps one rounded operation, rn-sae), the constants are
 * the kernels' data tables, and VRCP14PS / VRSQRT14PS are reproduced exactly from approx14_tables.h (see
 * make_approx14_tables.py).  PINNED: tests/test_oracle_svml.py compares these functions with np.arctan2 / np.arcsin themselves
 * on >= 1e8 inputs whenever NumPy reports AVX512_SKX (0 differing results), and with committed vectors otherwise.
 *
 * Outside the kernels' main paths (atan2: an argument that is 0, NaN, or outside [2^-125, 2^123); asin: |x| > 1) SVML calls a
 * scalar routine; here those lanes take the correctly rounded value (float64 libm, rounded once) -- for the values a point
 * cloud can produce there (a coordinate that is exactly 0: results 0, pi/2, pi) the two agree exactly.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/build_oracle.py); fmaf must be a correctly rounded fused operation
 * (libm's is, -mfma's is).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "approx14_tables.h"

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* VRCP14PS for a normal x whose reciprocal is normal */
static float rcp14(float x) {
  const uint32_t u = f2u(x), sign = u & 0x80000000u, m = u & 0x7fffffu;
  const int e = (int)((u >> 23) & 0xff);
  if (m == 0) return u2f(sign | (uint32_t)(254 - e) << 23);                      /* 2^k -> 2^-k */
  const int i = (int)(m >> 17), lo = (int)((m >> 7) & 1023);
  const uint32_t v = (uint32_t)((OVN_RCP14_A[i] - OVN_RCP14_B[i] * lo) >> 9);    /* in [2^16, 2^17): value v / 2^17 of 1 / 1.m */
  return u2f(sign | (uint32_t)(253 - e) << 23 | (v & 0xffffu) << 7);
}

/* VRSQRT14PS for a normal x > 0 */
static float rsqrt14(float x) {
  const uint32_t u = f2u(x), m = u & 0x7fffffu;
  const int e = (int)((u >> 23) & 0xff) - 127;            /* x = 1.m * 2^e */
  const int par = e & 1;                                  /* x = (1.m * 2^par) * 4^h, h = (e - par) / 2 */
  const int h = (e - par) / 2;
  if (m == 0 && par == 0) return u2f((uint32_t)(127 - h) << 23);
  const int i = par << 5 | (int)(m >> 18), lo = (int)((m >> 8) & 1023);
  const uint32_t v = (uint32_t)((OVN_RSQRT14_A[i] - OVN_RSQRT14_B[i] * lo) >> 9);
  return u2f((uint32_t)(126 - h) << 23 | (v & 0xffffu) << 7);
}

float ovn_svml_atan2f(float y, float x) {
  const uint32_t ux = f2u(x), uy = f2u(y);
  const uint32_t ax = ux & 0x7fffffffu, ay = uy & 0x7fffffffu;
  /* main path: both magnitudes in [2^-125, 2^123) (satan2 data +0x400 / +0x440) */
  if (ax - 0x01000000u >= 0x7c000000u || ay - 0x01000000u >= 0x7c000000u) return (float)atan2((double)y, (double)x);
  const float fx = u2f(ax), fy = u2f(ay);
  const int small = fy < fx;                      /* |y| < |x| */
  const float num = small ? fy : -fx;
  const float den = small ? fx : fy;
  const float off = small ? 0.0f : u2f(0x3fc90fdbu);   /* pi/2 */
  float r = rcp14(den);
  const float e = fmaf(-den, r, 1.0f);
  r = fmaf(e, r, r);
  const float q0 = num * r;
  const float rem = fmaf(-den, q0, num);
  const float q = fmaf(rem, r, q0);
  const float s = q * q;
  const float s2 = s * s;
  float ev = fmaf(s2, u2f(0x3b322cc0u), u2f(0x3d2bc384u));
  ev = fmaf(s2, ev, u2f(0x3dd96474u));
  float od = fmaf(s2, u2f(0xbc7f2631u), u2f(0xbd987629u));
  od = fmaf(s2, od, u2f(0xbe1161f8u));
  ev = fmaf(s2, ev, u2f(0x3e4cb79fu));
  od = fmaf(s2, od, u2f(0xbeaaaa49u));
  ev = fmaf(s2, ev, 1.0f);
  od = fmaf(s, od, ev);
  float res = fmaf(q, od, off);
  res = u2f(f2u(res) | (ux & 0x80000000u));       /* x < 0: -res */
  if (x <= 0.0f) res = res + u2f(0x40490fdbu);    /* + pi */
  return u2f(f2u(res) | (uy & 0x80000000u));
}

float ovn_svml_asinf(float x) {
  const uint32_t ux = f2u(x);
  const float a = u2f(ux & 0x7fffffffu);
  if (!(a <= 1.0f)) return (float)asin((double)x);             /* |x| > 1 or NaN: SVML's scalar call-out */
  const int big = !(a < 0.5f);
  const float t = fmaf(-a, 0.5f, 0.5f);                        /* (1 - |x|) / 2 */
  const float z = big ? t : a * a;                             /* vminps(x^2, t) */
  float base = a;
  if (big) {
    const float rs = (t < u2f(0x2f800000u)) ? 0.0f : rsqrt14(t);
    const float t2 = t + t;
    const float rr = rs * rs;
    const float sq = t2 * rs;                                  /* ~ 2 sqrt(t) */
    const float d = fmaf(rr, t2, -2.0f);
    const float sd = sq * d;
    float c = fmaf(d, u2f(0xbdc00004u), u2f(0x3e800001u));
    base = fmaf(sd, c, -sq);                                   /* -2 sqrt(t), refined */
  }
  float p1 = fmaf(z, u2f(0x3d3a9ab4u), u2f(0x3d997c12u));
  float p = fmaf(z, u2f(0x3d2edc07u), u2f(0x3cc32a6bu));
  const float zz = z * z;
  p = fmaf(p, zz, p1);
  p = fmaf(p, z, u2f(0x3e2aaaffu));
  p = z * p;
  float res = fmaf(p, base, base);
  if (big) res = res + u2f(0x3fc90fdbu);
  return u2f(f2u(res) ^ (ux & 0x80000000u));
}

void ovn_svml_atan2f_array(const float* y, const float* x, float* out, long n) {
  for (long i = 0; i < n; ++i) out[i] = ovn_svml_atan2f(y[i], x[i]);
}

void ovn_svml_asinf_array(const float* x, float* out, long n) {
  for (long i = 0; i < n; ++i) out[i] = ovn_svml_asinf(x[i]);
}

float ovn_rcp14(float x) { return rcp14(x); }
float ovn_rsqrt14(float x) { return rsqrt14(x); }
