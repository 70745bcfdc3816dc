// Microbenchmark (GPU box): what one ds_add_f64 of a wave costs the LDS unit
// of a CU, by address pattern - the question behind the multi-ion transport
// kernel's combining table (kernels.h, table_row): 16 such instructions per
// march iteration, the four quarters of a wave adding to the 16 doubles of a
// table row each.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o lds_atomic lds_atomic.hip
// Every wave issues PER_WAVE ds_add_f64 to a table in LDS; 3 workgroups of 4
// waves per CU (the kernel's occupancy). Patterns (lane l, quarter q = l / 16,
// i = l % 16):
//   0  64 distinct doubles, contiguous                    (no sharing at all)
//   1  row q of 4 rows of 16 doubles, rows 128 B apart    (4 cells, no sharing)
//   2  all quarters in ONE row: 4 lanes per address       (the 4 packets share a cell)
//   3  quarters 0,1 in one row, 2,3 in another            (2 lanes per address)
//   4  all 64 lanes one address
//   5  pattern 2 with only quarter 0 active (exec mask)   (after a cross-quarter sum)
//   6  pattern 1 with rows 136 B apart                    (bank skew between quarters)
//   7  pattern 2, ds_add_rtn_f64 (returning)              (for reference)
//   8  16 lanes per address (groups of 4 lanes x 16: the "soft photon" layout)
//   9  pattern 0 as ds_add_f32                            (for reference)
// Prints ns per instruction per CU and cycles at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int PER_WAVE = 1 << 14;

template <int MODE>
__global__ void __launch_bounds__(256, 3) lds_adds(double *out, double seed) {
  __shared__ double table[4096];
  for (int k = threadIdx.x; k < 4096; k += 256)
    table[k] = 0.;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, i = lane & 15;
  int index;
  switch (MODE) {
  case 0: index = lane; break;
  case 1: index = 16 * q + i; break;
  case 2: case 5: case 7: index = i; break;
  case 3: index = 16 * (q >> 1) + i; break;
  case 4: index = 0; break;
  case 6: index = 17 * q + i; break;
  case 8: index = lane & 3; break;
  default: index = lane; break;
  }
  // each wave its own region, and a different row every instruction (the
  // kernel's slots move too); 8 rows of 64 doubles per wave
  double *base = table + 1024 * wave;
  double v = seed;
  if (MODE == 9) {
    float *fbase = reinterpret_cast<float *>(base);
    for (int k = 0; k < PER_WAVE; ++k)
      atomicAdd(fbase + ((k & 7) << 7) + index, (float)v);
  } else if (MODE == 5) {
    if (q == 0)
      for (int k = 0; k < PER_WAVE; ++k)
        atomicAdd(base + ((k & 7) << 6) + index, v);
  } else if (MODE == 7) {
    for (int k = 0; k < PER_WAVE; ++k)
      v += 1e-300 * atomicAdd(base + ((k & 7) << 6) + index, v);
  } else {
    for (int k = 0; k < PER_WAVE; ++k)
      atomicAdd(base + ((k & 7) << 6) + index, v);
  }
  __syncthreads();
  if (threadIdx.x < 64)
    out[blockIdx.x * 64 + threadIdx.x] = table[threadIdx.x] + v;
}

template <int MODE> void run(double *out, int blocks, int cus) {
  hipEvent_t t0, t1;
  CHECK(hipEventCreate(&t0));
  CHECK(hipEventCreate(&t1));
  lds_adds<MODE><<<blocks, 256>>>(out, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(t0));
  lds_adds<MODE><<<blocks, 256>>>(out, 1.0);
  CHECK(hipEventRecord(t1));
  CHECK(hipEventSynchronize(t1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, t0, t1));
  // instructions per CU: 12 waves x PER_WAVE
  const double per_cu = (double)blocks / cus * 4. * PER_WAVE;
  const double ns = ms * 1e6 / per_cu;
  printf("pattern %d: %8.3f ms  %6.2f ns per ds_add per CU  = %5.1f cycles at 2.4 GHz\n",
         MODE, ms, ns, ns * 2.4);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 3;
  double *out;
  CHECK(hipMalloc(&out, (size_t)blocks * 64 * sizeof(double)));
  printf("%s: %d CUs, %d workgroups of 4 waves, %d ds_add per wave\n",
         prop.name, cus, blocks, PER_WAVE);
  run<0>(out, blocks, cus);
  run<1>(out, blocks, cus);
  run<2>(out, blocks, cus);
  run<3>(out, blocks, cus);
  run<4>(out, blocks, cus);
  run<5>(out, blocks, cus);
  run<6>(out, blocks, cus);
  run<7>(out, blocks, cus);
  run<8>(out, blocks, cus);
  run<9>(out, blocks, cus);
  return 0;
}
