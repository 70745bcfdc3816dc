// Microbenchmark (GPU box): throughput of scattered fp64 atomic adds as a
// function of the atomic's memory scope and of who touches which memory.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o atomic_scope atomic_scope.hip
// Each lane adds 1.0 to pseudo-random elements of a region of R doubles.
//   mode 0: all blocks -> one shared region, agent scope (what the engine does)
//   mode 1: block b -> region of XCD (b % 8), agent scope
//   mode 2: block b -> region of XCD (b % 8), workgroup scope (atomic may be
//           executed in that XCD's L2; only valid because no other XCD
//           touches the region)
//   mode 3: as 0, but the 8 lanes of a group add to the 8 doubles of ONE
//           random 64-B line (one memory request carries 8 adds)
//   mode 4: as 0, 16 lanes -> 16 consecutive doubles (128 B: the multi-ion
//           kernels' accumulator rows)
// Verifies the sums (mode 2 would lose updates if the XCD mapping assumption
// or the L2 execution were wrong).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void scatter(double *base, size_t region, int per_lane) {
  const unsigned xcd = blockIdx.x & 7u;
  double *mem = base + (MODE == 0 ? 0 : (size_t)xcd * region);
  unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  for (int k = 0; k < per_lane; ++k) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    size_t i = (size_t)((s >> 20) % region);
    if (MODE == 3 || MODE == 4) {
      const int g = MODE == 3 ? 8 : 16;
      /* the group's first lane picks the line for all of them */
      const unsigned long long lead =
          __shfl((unsigned long long)i, (threadIdx.x & 63) & ~(g - 1), 64);
      i = ((size_t)lead & ~(size_t)(g - 1)) + (threadIdx.x & (g - 1));
      if (i >= region)
        i -= g;
    }
    if (MODE == 2)
      __hip_atomic_fetch_add(mem + i, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      __hip_atomic_fetch_add(mem + i, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main(int argc, char **argv) {
  const size_t region = argc > 1 ? (size_t)atoll(argv[1]) : (2u << 20); // doubles per region
  const int per_lane = 2048;
  const int blocks = 2048;
  double *mem;
  CHECK(hipMalloc(&mem, 8 * region * sizeof(double)));
  hipEvent_t t0, t1;
  CHECK(hipEventCreate(&t0));
  CHECK(hipEventCreate(&t1));
  for (int mode = 0; mode < 5; ++mode) {
    CHECK(hipMemset(mem, 0, 8 * region * sizeof(double)));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(t0));
    if (mode == 0) scatter<0><<<blocks, 256>>>(mem, region, per_lane);
    if (mode == 1) scatter<1><<<blocks, 256>>>(mem, region, per_lane);
    if (mode == 2) scatter<2><<<blocks, 256>>>(mem, region, per_lane);
    if (mode == 3) scatter<3><<<blocks, 256>>>(mem, region, per_lane);
    if (mode == 4) scatter<4><<<blocks, 256>>>(mem, region, per_lane);
    CHECK(hipEventRecord(t1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, t0, t1));
    std::vector<double> host(8 * region);
    CHECK(hipMemcpy(host.data(), mem, host.size() * sizeof(double), hipMemcpyDeviceToHost));
    double sum = 0.;
    for (double v : host) sum += v;
    const double n = (double)blocks * 256 * per_lane;
    printf("region %8.2f MB  mode %d  %8.3f ms  %7.2f G atomics/s  sum %s (%.0f of %.0f)\n",
           region * 8. / 1e6, mode, ms, n / ms / 1e6, sum == n ? "ok" : "LOST UPDATES", sum, n);
  }
  return 0;
}
