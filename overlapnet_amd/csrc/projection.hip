// Spherical projection (range / vertex / intensity / index images) and normal map for gfx950.
// Compile this file with -ffp-contract=off: the float32 arithmetic below is written to round exactly
// like the reference's NumPy code does, one IEEE operation at a time.
//
// Reference: src/utils/utils.py:59-134 (range_projection) and :137-186 (gen_normal_map, wrap).
//   range_projection sorts the points by decreasing depth and scatters them so that the nearest point
//   wins each pixel (:107-132).  Here the same winner is found without sorting: every kept point does a
//   64-bit atomicMin of the key (float bits of depth << 32 | point index) on its pixel -- depth > 0 so
//   the float bits order like the floats, and equal depths fall back to the lower point index.
//   atan2 / asin reproduce NumPy's float32 results bit for bit (svml_f32.h: the SVML kernels NumPy runs on the machine the
//   golden vectors come from, VRCP14PS / VRSQRT14PS included) -- rounds 1-3 used the float64 functions rounded to float32, which
//   put a point into the neighbouring pixel a few times per million points (12 range pixels over the 24 transformed clouds of
//   the preprocessing parity set; now 0).  What remains undefined in the reference itself: two points of one pixel
//   with bit-identical minimal depth (its unstable argsort picks either; the lower index wins here; 20 such pixels in those 24
//   clouds, same range value either way).
//   The scatter is bound by its atomics (measured, tools/experiments/README.md round 3: 0.9 ms per 1025 clouds, 0.35 ms of it without
//   them; replacing the float64 atan2 / asin by exact precomputed pixel thresholds changed nothing).
//   gen_normal_map's per-pixel Python loop (:149-173) becomes one thread per pixel; the vector norm
//   reproduces np.linalg.norm on a float32 3-vector (float32 products summed in a double -- OpenBLAS
//   sdot's scalar tail -- then rounded to float32 and square-rooted).
#include "ovn_internal.h"
#include "svml_f32.h"

namespace {

constexpr unsigned long long EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;
constexpr int PB = 256;  // points per block in the scatter kernel

struct ProjGeom {
  int H, W;
  float fov_down_abs;  // float32(|fov_down| in radians)
  float fov;           // float32(|fov_down| + |fov_up|)
  float max_range;
  int trig;            // 0: NumPy's float32 arctan2 / arcsin on AVX512_SKX x86-64 (SVML, svml_f32.h); 1: the correctly rounded float32
                       // results (float64 function rounded once) -- what a host whose NumPy calls a correctly rounded libm produces
};

// utils.py:75-104 for one point: depth, the range filter, both angles and the pixel.  false = dropped by the filter.
__device__ __forceinline__ bool point_to_pixel(float x, float y, float z, const ProjGeom& gm, float& depth, float& yaw, float& pitch,
                                               int& pix) {
  depth = sqrtf((x * x + y * y) + z * z);                      // utils.py:75
  yaw = pitch = 0.f;
  pix = 0;
  if (!((depth > 0.0f) && (depth < gm.max_range))) return false;   // utils.py:76-77
  if (gm.trig == 0) {
    yaw = -ovn_svml::atan2f_np(y, x);                          // utils.py:86  (np.arctan2 on float32: svml_f32.h)
    pitch = ovn_svml::asinf_np(z / depth);                     // utils.py:87  (np.arcsin on float32)
  } else {                                                     // ovn_set_projection_trig(ctx, 1)
    yaw = -(float)atan2((double)y, (double)x);
    pitch = (float)asin((double)(z / depth));
  }
  float px = 0.5f * (yaw / 3.14159274101257324f + 1.0f);       // utils.py:90
  float py = 1.0f - (pitch + gm.fov_down_abs) / gm.fov;        // utils.py:91
  px = px * (float)gm.W;                                       // utils.py:94
  py = py * (float)gm.H;                                       // utils.py:95
  px = fmaxf(0.0f, fminf((float)(gm.W - 1), floorf(px)));      // utils.py:98-100
  py = fmaxf(0.0f, fminf((float)(gm.H - 1), floorf(py)));      // utils.py:102-104
  pix = (int)py * gm.W + (int)px;
  return true;
}

__global__ void proj_angles_kernel(const float* __restrict__ points, long long n, ProjGeom gm, float* __restrict__ yaw_out,
                                   float* __restrict__ pitch_out, int* __restrict__ pixel_out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 pt = *reinterpret_cast<const f32x4*>(points + i * 4);
    float depth, yaw, pitch;
    int pix;
    const bool keep = point_to_pixel(pt[0], pt[1], pt[2], gm, depth, yaw, pitch, pix);
    if (yaw_out) yaw_out[i] = yaw;
    if (pitch_out) pitch_out[i] = pitch;
    if (pixel_out) pixel_out[i] = keep ? pix : -1;
  }
}

__global__ void proj_clear_kernel(unsigned long long* __restrict__ keys, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    keys[i] = EMPTY_KEY;
}

// grid = (ceil(max_points/PB), n_scans)
__global__ __launch_bounds__(PB) void proj_scatter_kernel(const float* __restrict__ points,
                                                          const long long* __restrict__ offsets, ProjGeom gm,
                                                          unsigned long long* __restrict__ keys,
                                                          int* __restrict__ local_idx, int* __restrict__ block_cnt,
                                                          int blocks_per_scan, long long max_points) {
  __shared__ int wave_cnt[PB / 64];
  const int scan = blockIdx.y;
  const long long beg = offsets[scan];
  const long long npts = offsets[scan + 1] - beg;
  const long long p = (long long)blockIdx.x * PB + threadIdx.x;
  bool keep = false;
  float depth = 0.f;
  int pix = 0;
  if (p < npts) {
    const f32x4 pt = *reinterpret_cast<const f32x4*>(points + (beg + p) * 4);
    float yaw, pitch;
    keep = point_to_pixel(pt[0], pt[1], pt[2], gm, depth, yaw, pitch, pix);
  }
  {
    // The scatter is bound by its 64-bit atomics, and consecutive points of a scan mostly fall into the same pixel (56 % of
    // the points of the KITTI fixture share the pixel of their predecessor: 63 kept lanes, 28 distinct pixels per wave).
    // Runs of adjacent lanes with the same pixel are combined in registers first: a lane at position 0, 4, 8, .. of its
    // run issues one atomicMin with the minimum key of (up to) the four lanes it covers.  Same result: min is associative.
    unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned long long)(unsigned)p;
    const int lane = threadIdx.x & 63;
    const int ppix = __shfl_up(pix, 1, 64);
    const int pkeep = __shfl_up((int)keep, 1, 64);
    const bool head = keep && (lane == 0 || !pkeep || ppix != pix);
    const unsigned long long hm = __ballot(head);
    const unsigned long long le = hm & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));   // heads at or below this lane
    const int hpos = le ? 63 - __clzll((long long)le) : -1;                               // this lane's run head
    const int run = keep ? hpos : -2 - lane;                                              // unique id for dropped lanes
#pragma unroll
    for (int d = 1; d <= 2; d <<= 1) {
      const unsigned long long kd = __shfl_down(key, d, 64);
      const int rd = __shfl_down(run, d, 64);
      if (lane + d < 64 && rd == run && kd < key) key = kd;
    }
    if (keep && ((lane - hpos) & 3) == 0) atomicMin(&keys[(long long)scan * gm.H * gm.W + pix], key);
  }
  if (local_idx) {
    // index of this point among the KEPT points of its block (utils.py:117-118 numbers points after the filter)
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63;
    const int below = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_cnt[w];
    if (p < npts) local_idx[(long long)scan * max_points + p] = base + below;
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < PB / 64; ++w) tot += wave_cnt[w];
      block_cnt[(long long)scan * blocks_per_scan + blockIdx.x] = tot;
    }
  }
}

// exclusive scan of the per-block kept counts of every scan (a few hundred entries per scan)
__global__ void proj_block_scan_kernel(int* __restrict__ block_cnt, int blocks_per_scan) {
  if (threadIdx.x != 0) return;
  int* c = block_cnt + (long long)blockIdx.x * blocks_per_scan;
  int run = 0;
  for (int b = 0; b < blocks_per_scan; ++b) {
    const int v = c[b];
    c[b] = run;
    run += v;
  }
}

__device__ __forceinline__ float norm3_like_numpy(float x, float y, float z) {
  // np.linalg.norm(float32[3]) == sqrt(float32(double(x*x) + double(y*y) + double(z*z)))
  const double s = ((double)(x * x) + (double)(y * y)) + (double)(z * z);
  return sqrtf((float)s);
}

// normal of pixel p from its right (u, wrapped) and lower (v) neighbours, utils.py:149-173; (-1,-1,-1) when undefined
__device__ __forceinline__ void normal_of(const f32x4& p, const f32x4& u, const f32x4& v, float& nx, float& ny, float& nz) {
  nx = ny = nz = -1.f;
  const float ux = u[0] - p[0], uy = u[1] - p[1], uz = u[2] - p[2];
  const float vx = v[0] - p[0], vy = v[1] - p[1], vz = v[2] - p[2];
  const float un = norm3_like_numpy(ux, uy, uz);
  const float vn = norm3_like_numpy(vx, vy, vz);
  const float ax = vx / vn, ay = vy / vn, az = vz / vn;  // v_norm
  const float bx = ux / un, by = uy / un, bz = uz / un;  // u_norm
  const float wx = ay * bz - az * by;                    // np.cross(v_norm, u_norm), utils.py:168
  const float wy = az * bx - ax * bz;
  const float wz = ax * by - ay * bx;
  const float wn = norm3_like_numpy(wx, wy, wz);
  if (wn > 0.0f) {  // NaN fails the test and the pixel stays -1 (utils.py:170)
    nx = wx / wn;
    ny = wy / wn;
    nz = wz / wn;
  }
}

// One thread per pixel: resolve the winner of the pixel and, when normals are wanted, of its right and lower neighbours straight
// from the key image, and write every requested output ONCE -- the range / vertex images are not materialised unless the caller
// asked for them (the stacked leg input alone: 8 B of key + three 16-byte point reads in, 4 C bytes out per pixel; the former
// gather -> range / vertex images -> normal kernel pair moved 2.4 GB more per 1025 scans: 1.17 -> 0.77 ms).
__global__ __launch_bounds__(256) void proj_resolve_kernel(const float* __restrict__ points, const long long* __restrict__ offsets,
                                                           const unsigned long long* __restrict__ keys, const int* __restrict__ local_idx,
                                                           const int* __restrict__ block_pref, int blocks_per_scan, long long max_points,
                                                           int H, int W, int n_scans, float* __restrict__ range,
                                                           float* __restrict__ vertex, float* __restrict__ intensity,
                                                           int32_t* __restrict__ idx, float* __restrict__ normal,
                                                           float* __restrict__ stacked, int use_depth, int use_normals,
                                                           int use_intensity, int C) {
  const int HW = H * W;
  const bool want_n = (normal != nullptr) || (stacked != nullptr && use_normals);
  // A workgroup resolves a TILE of 8 rows x 32 columns, not 256 pixels of one row: the lower neighbour of 7 of its 8 rows is a pixel of
  // the same workgroup, so the point lines gathered for row y + 1 serve as row y + 1's own winners too (points of one image row are
  // consecutive in a ring-ordered scan; the row below lives ~30 KB further on).  And all tiles of a scan are resolved on ONE XCD
  // (workgroup b runs on XCD b mod 8 -- tools/experiments/xcd_atomics.hip; the grid size is a multiple of 8): the scan's points and
  // keys pass through one L2 instead of eight.  Together 1.63 -> 1.56 ms per 1025 clouds for the three kernels (same box); handing
  // the neighbours' winners over through LDS instead of gathering them again (296 gathers per tile instead of 768) made it SLOWER
  // (1.65): the gathers hit L1 / L2, what the kernel moves through HBM is keys + points + output either way.
  constexpr int TR = 8, TC = 32;
  const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR;
  const int tps = tiles_y * tiles_x;
  const long long n_tiles = (long long)((n_scans + 7) / 8) * 8 * tps;
  const int ty = threadIdx.x / TC, tx = threadIdx.x - ty * TC;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long k = tile >> 3;
    const int scan = (int)(tile & 7) + 8 * (int)(k / tps);
    if (scan >= n_scans) continue;
    const int tin = (int)(k % tps);
    const int tyi = tin / tiles_x;
    const int py = tyi * TR + ty, px = (tin - tyi * tiles_x) * TC + tx;
    if (py >= H || px >= W) continue;
    const int pix = py * W + px;
    const long long sbase = (long long)scan * HW;
    const long long q = sbase + pix;
    const float* pts = points + offsets[scan] * 4;
    const unsigned long long key = keys[q];
    f32x4 v = {-1.f, -1.f, -1.f, -1.f};
    float d = -1.f, it = -1.f, nx = -1.f, ny = -1.f, nz = -1.f;
    int id = -1;
    if (key != EMPTY_KEY) {
      const long long p = (long long)(key & 0xFFFFFFFFull);
      const f32x4 pt = *reinterpret_cast<const f32x4*>(pts + p * 4);
      d = __uint_as_float((unsigned)(key >> 32));
      it = pt[3];
      v = (f32x4){pt[0], pt[1], pt[2], 1.0f};
      if (idx) id = block_pref[(long long)scan * blocks_per_scan + (int)(p / PB)] + local_idx[(long long)scan * max_points + p];
      if (want_n) {
        const int y = pix / W;
        const int x = pix - y * W;
        if (y < H - 1) {
          const int xr = (x + 1 >= W) ? (x + 1 - W) : (x + 1);  // wrap(), utils.py:178-186
          const unsigned long long ku = keys[sbase + (long long)y * W + xr];
          const unsigned long long kv = keys[sbase + (long long)(y + 1) * W + x];
          if (ku != EMPTY_KEY && kv != EMPTY_KEY) {           // range > 0 at both neighbours (a kept point has depth > 0)
            const f32x4 pu = *reinterpret_cast<const f32x4*>(pts + (long long)(ku & 0xFFFFFFFFull) * 4);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(pts + (long long)(kv & 0xFFFFFFFFull) * 4);
            normal_of(pt, pu, pv, nx, ny, nz);
          }
        }
      }
    }
    if (range) range[q] = d;
    if (vertex) *reinterpret_cast<f32x4*>(vertex + q * 4) = v;
    if (intensity) intensity[q] = it;
    if (idx) idx[q] = id;
    if (normal) {
      normal[q * 3 + 0] = nx;
      normal[q * 3 + 1] = ny;
      normal[q * 3 + 2] = nz;
    }
    if (stacked) {
      float* o = stacked + q * C;
      if (C == 4 && use_depth && use_normals) {
        *reinterpret_cast<f32x4*>(o) = (f32x4){d, nx, ny, nz};
      } else {
        int c = 0;
        if (use_depth) o[c++] = d;
        if (use_normals) {
          o[c++] = nx;
          o[c++] = ny;
          o[c++] = nz;
        }
        if (use_intensity) o[c++] = it;
      }
    }
  }
}

// one thread per pixel: normal (utils.py:149-173) + the stacked leg input
__global__ void proj_normal_kernel(const float* __restrict__ range, const float* __restrict__ vertex,
                                   const float* __restrict__ intensity, int H, int W, int n_scans,
                                   float* __restrict__ normal, float* __restrict__ stacked, int use_depth,
                                   int use_normals, int use_intensity, int C) {
  const int HW = H * W;
  const long long total = (long long)n_scans * HW;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (long long)gridDim.x * blockDim.x) {
    const int pix = (int)(q % HW);
    const long long sbase = q - pix;
    const int y = pix / W;
    const int x = pix - y * W;
    float nx = -1.f, ny = -1.f, nz = -1.f;
    const float d = range[q];
    if ((normal || (stacked && use_normals)) && y < H - 1 && d > 0.0f) {
      const int xr = (x + 1 >= W) ? (x + 1 - W) : (x + 1);  // wrap(), utils.py:178-186
      const long long qu = sbase + (long long)y * W + xr;
      const long long qv = sbase + (long long)(y + 1) * W + x;
      if (range[qu] > 0.0f && range[qv] > 0.0f) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(vertex + q * 4);
        const f32x4 u = *reinterpret_cast<const f32x4*>(vertex + qu * 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(vertex + qv * 4);
        const float ux = u[0] - p[0], uy = u[1] - p[1], uz = u[2] - p[2];
        const float vx = v[0] - p[0], vy = v[1] - p[1], vz = v[2] - p[2];
        const float un = norm3_like_numpy(ux, uy, uz);
        const float vn = norm3_like_numpy(vx, vy, vz);
        const float ax = vx / vn, ay = vy / vn, az = vz / vn;  // v_norm
        const float bx = ux / un, by = uy / un, bz = uz / un;  // u_norm
        const float wx = ay * bz - az * by;                    // np.cross(v_norm, u_norm), utils.py:168
        const float wy = az * bx - ax * bz;
        const float wz = ax * by - ay * bx;
        const float wn = norm3_like_numpy(wx, wy, wz);
        if (wn > 0.0f) {  // NaN fails the test and the pixel stays -1 (utils.py:170)
          nx = wx / wn;
          ny = wy / wn;
          nz = wz / wn;
        }
      }
    }
    if (normal) {
      normal[q * 3 + 0] = nx;
      normal[q * 3 + 1] = ny;
      normal[q * 3 + 2] = nz;
    }
    if (stacked) {
      float* o = stacked + q * C;
      int c = 0;
      if (use_depth) o[c++] = d;
      if (use_normals) {
        o[c++] = nx;
        o[c++] = ny;
        o[c++] = nz;
      }
      if (use_intensity) o[c++] = intensity[q];
    }
  }
}

// same double arithmetic as utils.py:70-72, then the float32 casts NumPy applies to the scalars
ProjGeom make_geom(int H, int W, double fov_up_deg, double fov_down_deg, double max_range, int trig) {
  const double up = fov_up_deg / 180.0 * 3.14159265358979323846;
  const double down = fov_down_deg / 180.0 * 3.14159265358979323846;
  ProjGeom gm;
  gm.H = H;
  gm.W = W;
  gm.fov_down_abs = (float)fabs(down);
  gm.fov = (float)(fabs(down) + fabs(up));
  gm.max_range = (float)max_range;
  gm.trig = trig;
  return gm;
}

}  // namespace

int ovn_projection_angles_forward(const float* points, int64_t n, int H, int W, double fov_up_deg, double fov_down_deg,
                                  double max_range, float* yaw, float* pitch, int32_t* pixel, hipStream_t stream, int trig) {
  const ProjGeom gm = make_geom(H, W, fov_up_deg, fov_down_deg, max_range, trig);
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(proj_angles_kernel, dim3(blocks), dim3(256), 0, stream, points, (long long)n, gm, yaw, pitch, pixel);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

int ovn_project_forward(ovn_ctx* ctx, const float* points, const int64_t* offsets, int n_scans, int64_t max_points,
                        int H, int W, double fov_up_deg, double fov_down_deg, double max_range, float* range,
                        float* vertex, float* intensity, int32_t* idx, float* normal, float* stacked, int use_depth,
                        int use_normals, int use_intensity, hipStream_t stream) {
  OVN_REQUIRE(n_scans >= 0 && H > 0 && W > 0 && max_points >= 0, OVN_ERR_ARG, "ovn_project: bad sizes");
  OVN_REQUIRE(max_points < (1ll << 32), OVN_ERR_ARG, "ovn_project: more than 2^32 points per scan");
  if (n_scans == 0) return OVN_OK;
  const int C = (use_depth ? 1 : 0) + (use_normals ? 3 : 0) + (use_intensity ? 1 : 0);
  OVN_REQUIRE(!stacked || C > 0, OVN_ERR_ARG, "ovn_project: stacked output requested with no channel enabled");

  const ProjGeom gm = make_geom(H, W, fov_up_deg, fov_down_deg, max_range, ctx->proj_trig);

  const long long HW = (long long)H * W;
  const long long npix = HW * n_scans;
  const int blocks_per_scan = (int)((max_points + PB - 1) / PB) > 0 ? (int)((max_points + PB - 1) / PB) : 1;

  // scratch carve-up (all 16-byte aligned): the key image, and for the index image the kept-point numbering
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_keys = carve((size_t)npix * 8);
  const size_t o_lidx = idx ? carve((size_t)n_scans * (size_t)max_points * 4 + 16) : 0;
  const size_t o_bcnt = idx ? carve((size_t)n_scans * blocks_per_scan * 4) : 0;
  int rc = ovn_ws_reserve(ctx, off, stream);
  if (rc) return rc;
  char* ws = static_cast<char*>(ctx->ws);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + o_keys);
  int* block_cnt = idx ? reinterpret_cast<int*>(ws + o_bcnt) : nullptr;
  // local_idx is indexed [scan][point]: n_scans * max_points ints
  int* local_idx = idx ? reinterpret_cast<int*>(ws + o_lidx) : nullptr;

  hipLaunchKernelGGL(proj_clear_kernel, dim3(1024), dim3(256), 0, stream, keys, npix);
  if (max_points > 0) {
    hipLaunchKernelGGL(proj_scatter_kernel, dim3(blocks_per_scan, n_scans), dim3(PB), 0, stream, points,
                       reinterpret_cast<const long long*>(offsets), gm, keys, local_idx, block_cnt, blocks_per_scan,
                       (long long)max_points);
    if (idx) hipLaunchKernelGGL(proj_block_scan_kernel, dim3(n_scans), dim3(64), 0, stream, block_cnt, blocks_per_scan);
  }
  const long long n_tiles = (long long)((n_scans + 7) / 8) * 8 * ((H + 7) / 8) * ((W + 31) / 32);
  const int gblocks = (int)(n_tiles < 16384 ? n_tiles : 16384);   // a multiple of 8 either way
  hipLaunchKernelGGL(proj_resolve_kernel, dim3(gblocks), dim3(256), 0, stream, points, reinterpret_cast<const long long*>(offsets),
                     keys, local_idx, block_cnt, blocks_per_scan, (long long)max_points, H, W, n_scans, range, vertex, intensity, idx,
                     normal, stacked, use_depth, use_normals, use_intensity, C);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}

int ovn_normals_forward(const float* range, const float* vertex, int n_scans, int H, int W, float* normal,
                        hipStream_t stream) {
  const long long npix = (long long)H * W * n_scans;
  if (npix == 0) return OVN_OK;
  const int gblocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
  hipLaunchKernelGGL(proj_normal_kernel, dim3(gblocks), dim3(256), 0, stream, range, vertex, (const float*)nullptr, H, W,
                     n_scans, normal, (float*)nullptr, 0, 0, 0, 0);
  OVN_HIP_CHECK(hipGetLastError());
  return OVN_OK;
}
