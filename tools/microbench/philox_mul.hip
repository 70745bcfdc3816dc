// Microbenchmark (GPU box): what a Philox4x32-10 block costs a SIMD, by the
// way its two 32 x 32 -> 64 bit products per round are written - the packet
// RNG of the engine (device_physics.h, PacketRng::next) regenerates one to
// three blocks per packet in the interaction kernels.
//   hipcc -O3 --offload-arch=gfx950 -o philox_mul philox_mul.hip
//   0  __umulhi(a, b) and a * b            (two multiplies per product)
//   1  (uint64_t)a * b                     (whatever the compiler makes of it)
//   2  v_mad_u64_u32 by inline asm         (one instruction per product)
// Every wave generates BLOCKS dependent blocks; waves per SIMD: 1, 2, 4.
// Prints cycles per block per wave at the measured clock and the blocks per
// microsecond of the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int BLOCKS = 1 << 12;

template <int MODE>
__device__ __forceinline__ void product(uint32_t a, uint32_t b, uint32_t &hi,
                                        uint32_t &lo) {
  if (MODE == 0) {
    hi = __umulhi(a, b);
    lo = a * b;
  } else if (MODE == 1) {
    const uint64_t p = (uint64_t)a * b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  } else {
    uint64_t p;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a), "v"(b) : "vcc");
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) philox(uint32_t *out, uint32_t seed,
                                              long long *cycles) {
  uint32_t c0 = threadIdx.x, c1 = blockIdx.x, c2 = 0, c3 = 0;
  const long long t0 = clock64();
  for (int b = 0; b < BLOCKS; ++b) {
    uint32_t k0 = seed, k1 = 7u;
    c2 += b;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0, lo0, hi1, lo1;
      product<MODE>(0xD2511F53u, c0, hi0, lo0);
      product<MODE>(0xCD9E8D57u, c2, hi1, lo1);
      c0 = hi1 ^ c1 ^ k0;
      c1 = lo1;
      c2 = hi0 ^ c3 ^ k1;
      c3 = lo0;
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *cycles = t1 - t0;
}

template <int MODE> void run(uint32_t *out, long long *cycles, int waves) {
  // 256 CUs x 4 SIMDs x waves
  const int groups = 256 * waves;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  philox<MODE><<<groups, 256>>>(out, 42u, cycles);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  philox<MODE><<<groups, 256>>>(out, 43u, cycles);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long c;
  CHECK(hipMemcpy(&c, cycles, sizeof c, hipMemcpyDeviceToHost));
  const double blocks = (double)groups * 4 * BLOCKS; /* wave-blocks */
  printf("mode %d, %d waves/SIMD: %8.3f ms, %7.1f ns per block per SIMD, "
         "%6.1f clock64 ticks per block per wave, %8.1f packet-blocks/us "
         "chip-wide\n",
         MODE, waves, ms, ms * 1e6 / (BLOCKS * waves), (double)c / BLOCKS,
         blocks * 64 / (ms * 1e3));
}

int main() {
  uint32_t *out;
  long long *cycles;
  CHECK(hipMalloc(&out, sizeof(uint32_t) * 256 * 256 * 8));
  CHECK(hipMalloc(&cycles, sizeof(long long)));
  for (int waves : {1, 2, 4}) {
    run<0>(out, cycles, waves);
    run<1>(out, cycles, waves);
    run<2>(out, cycles, waves);
  }
  return 0;
}
