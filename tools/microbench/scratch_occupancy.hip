// How many waves does a SIMD hold when every lane owns a private (scratch) array of NW doubles?
// Each wave runs a register-only dependent FMA chain (the private array is touched through a runtime index at both ends only, so it
// exists) and stamps wall_clock64 at start and end; from the stamps: the largest number of waves alive at one time, per SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/microbench/scratch_occupancy.hip -o tools/microbench/scratch_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NW> __global__ void __launch_bounds__(256) chain(double* out, long long* stamps, int iters, int idx) {
  double a[NW > 0 ? NW : 1];
  const long long t0 = wall_clock64();
  double s = 1.0 + 1e-9 * threadIdx.x;
  if (NW > 0) { a[(idx + threadIdx.x) % NW] = s; a[(idx + 3 * threadIdx.x + 1) % NW] = 2.0 * s; s += a[(2 * idx + threadIdx.x) % NW]; }
  for (int it = 0; it < iters; it++) {
    s = __builtin_fma(s, 0.999999, 1e-12);
    s = __builtin_fma(s, 1.000001, 1e-12);
    s = __builtin_fma(s, 0.999999, 1e-12);
    s = __builtin_fma(s, 1.000001, 1e-12);
  }
  if (NW > 0) { a[(idx + 5 * threadIdx.x + 2) % NW] = s; s += a[(3 * idx + threadIdx.x) % NW]; }
  const long long t1 = wall_clock64();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  out[g] = s;
  if ((threadIdx.x & 63) == 0) { stamps[2 * (g >> 6)] = t0; stamps[2 * (g >> 6) + 1] = t1; }
}

template <int NW> int run(int waves_per_simd, int iters) {
  const int wgs = 256 * waves_per_simd;            // 256 CUs x (4 waves per WG = 1 per SIMD) x waves_per_simd
  const size_t n = (size_t)wgs * 256;
  double* out; long long* st;
  CHECK(hipMalloc(&out, n * 8)); CHECK(hipMalloc(&st, n / 64 * 16));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    chain<NW><<<wgs, 256>>>(out, st, iters, 7 + rep);
    hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
  }
  std::vector<long long> h(n / 64 * 2);
  CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<std::pair<long long, int>> ev;
  double alive = 0; long long lo = h[0], hi = h[1];
  for (size_t w = 0; w < n / 64; w++) { ev.push_back({h[2 * w], +1}); ev.push_back({h[2 * w + 1], -1}); alive += (double)(h[2 * w + 1] - h[2 * w]); lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
  std::sort(ev.begin(), ev.end());
  int cur = 0, peak = 0;
  for (auto& e : ev) { cur += e.second; peak = std::max(peak, cur); }
  printf("private %6d B/lane  grid %2d waves/SIMD  %8.3f ms   waves alive per SIMD: peak %.2f, average %.2f\n", NW * 8, waves_per_simd, best,
         peak / 1024.0, alive / (double)(hi - lo) / 1024.0);
  hipFree(out); hipFree(st);
  return 0;
}

int main(int argc, char** argv) {
  // default: long-lived waves (the occupancy itself); "short": waves that live ~100 us, 128 per SIMD -- the rate at which workgroups
  // are launched then shows (a launch with scratch has its wave slots set up)
  const bool brief = argc > 1;
  const int iters = brief ? 2000 : 100000;
  for (int wps : {brief ? 128 : 8, brief ? 256 : 32}) {
    if (run<0>(wps, iters) || run<16>(wps, iters) || run<128>(wps, iters) || run<402>(wps, iters) || run<602>(wps, iters) || run<2048>(wps, iters) || run<6002>(wps, iters)) return 1;
  }
  return 0;
}
