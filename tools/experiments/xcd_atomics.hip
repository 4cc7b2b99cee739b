// Experiment (round 4): are 64-bit atomicMin's cheaper when every workgroup that touches an image runs on ONE XCD and the atomic is
// issued at workgroup scope (no sc1: performed in that XCD's L2) instead of agent scope (sc1: forwarded to the memory side)?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/xcd_atomics.hip -o tools/bin/xcd_atomics && tools/bin/xcd_atomics
// Prints: XCC_ID histogram by (blockIdx % 8) -> is dispatch round-robin; time of N atomics in the four combinations
// {image shared by all XCDs, image private to one XCD} x {agent scope, workgroup scope}; result check of the private/workgroup form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }   // HW_REG_XCC_ID[3:0]

__global__ void k_hist(int* hist) { if (threadIdx.x == 0) atomicAdd(&hist[(blockIdx.x & 7) * 16 + xcc_id()], 1); }

// images: n_img images of IMG keys.  mode 0: image = blockIdx / blocks_per_img (blocks of an image on all XCDs), agent scope
//         mode 1: same mapping, workgroup scope (WRONG across XCDs -- timing only)
//         mode 2: image chosen so that all its blocks share blockIdx % 8 (XCD-private), agent scope
//         mode 3: XCD-private, workgroup scope
template <int MODE, int SWZ = 0>
__global__ __launch_bounds__(256) void k_atom(unsigned long long* keys, int img_keys, int blocks_per_img, int n_img, unsigned seed) {
  int b = blockIdx.x, img, blk;
  if (MODE < 2) { img = b / blocks_per_img; blk = b - img * blocks_per_img; }
  else { const int x = b & 7, k = b >> 3; img = x + 8 * (k / blocks_per_img); blk = k % blocks_per_img; }
  if (img >= n_img) return;
  unsigned long long* im = keys + (size_t)img * img_keys;
  // 256 points per block, LiDAR-like locality: point p of the image lands near pixel p * img_keys / (blocks*256) (+ small jitter)
  const int p = blk * 256 + threadIdx.x;
  unsigned h = (unsigned)p * 2654435761u ^ seed ^ (unsigned)img * 40503u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  int pix = (int)(((long long)p * img_keys) / ((long long)blocks_per_img * 256) + (h & 3)) % img_keys;
  // SWZ: neighbouring pixels SWZ-way interleaved (pixel i -> slot (i % SWZ) * (img_keys / SWZ) + i / SWZ): the ~28 pixels a wave touches
  // then lie in ~28 different cache lines (L2 channels) instead of ~5
  if (SWZ) pix = (pix % SWZ) * (img_keys / SWZ) + pix / SWZ;
  const unsigned long long key = ((unsigned long long)(h >> 8) << 32) | (unsigned)p;
  if (MODE & 1) __hip_atomic_fetch_min(&im[pix], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_min(&im[pix], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  int* hist; CK(hipMalloc(&hist, 128 * 4)); CK(hipMemset(hist, 0, 128 * 4));
  hipLaunchKernelGGL(k_hist, dim3(8192), dim3(64), 0, 0, hist);
  std::vector<int> h(128); CK(hipMemcpy(h.data(), hist, 512, hipMemcpyDeviceToHost));
  printf("XCC_ID histogram, row = blockIdx %% 8:\n");
  for (int r = 0; r < 8; ++r) { for (int c = 0; c < 16; ++c) printf("%5d", h[r * 16 + c]); printf("\n"); }
  const int IMG = 57600, BPI = 487, NIMG = 1024;
  unsigned long long *keys, *ref;
  CK(hipMalloc(&keys, (size_t)NIMG * IMG * 8)); CK(hipMalloc(&ref, (size_t)NIMG * IMG * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> a((size_t)NIMG * IMG), b((size_t)NIMG * IMG);
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(keys, 0xff, (size_t)NIMG * IMG * 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      const int grid = (mode < 2) ? NIMG * BPI : 8 * ((NIMG + 7) / 8) * BPI;
      if (mode == 0) hipLaunchKernelGGL(k_atom<0>, dim3(grid), dim3(256), 0, 0, keys, IMG, BPI, NIMG, 12345u);
      if (mode == 1) hipLaunchKernelGGL(k_atom<1>, dim3(grid), dim3(256), 0, 0, keys, IMG, BPI, NIMG, 12345u);
      if (mode == 2) hipLaunchKernelGGL(k_atom<2>, dim3(grid), dim3(256), 0, 0, keys, IMG, BPI, NIMG, 12345u);
      if (mode == 3) hipLaunchKernelGGL(k_atom<3>, dim3(grid), dim3(256), 0, 0, keys, IMG, BPI, NIMG, 12345u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipDeviceSynchronize());
    if (mode == 0) CK(hipMemcpy(a.data(), keys, a.size() * 8, hipMemcpyDeviceToHost));
    if (mode == 3) {
      CK(hipMemcpy(b.data(), keys, b.size() * 8, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
      printf("mode 3 vs mode 0: %zu differing keys of %zu\n", bad, a.size());
    }
    printf("mode %d (%s, %s scope): %.3f ms for %.1f M atomics\n", mode, mode < 2 ? "image on all XCDs" : "XCD-private image",
           (mode & 1) ? "workgroup" : "agent", best, NIMG * (double)BPI * 256 / 1e6);
  }
  for (int swz = 0; swz < 3; ++swz) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(keys, 0xff, (size_t)NIMG * IMG * 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (swz == 0) hipLaunchKernelGGL((k_atom<0, 0>), dim3(NIMG * BPI), dim3(256), 0, 0, keys, IMG, BPI, NIMG, 12345u);
      if (swz == 1) hipLaunchKernelGGL((k_atom<0, 32>), dim3(NIMG * BPI), dim3(256), 0, 0, keys, IMG, BPI, NIMG, 12345u);
      if (swz == 2) hipLaunchKernelGGL((k_atom<0, 8>), dim3(NIMG * BPI), dim3(256), 0, 0, keys, IMG, BPI, NIMG, 12345u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("layout %s: %.3f ms for %.1f M atomics (agent scope, image on all XCDs)\n", swz == 0 ? "linear" : swz == 1 ? "32-way interleaved" : "8-way interleaved", best, NIMG * (double)BPI * 256 / 1e6);
  }
  return 0;
}
