// Microbenchmark (GPU box): what a CU mask on a stream does on this part -
// where the workgroups of a launch land (XCC, shader engine, CU) and how long
// a kernel bound by vector issue takes, per mask word (the word is repeated
// over the 256 bits, as CMI_EXP_CU_MASK does in engine.hip).
//   hipcc -O3 --offload-arch=gfx950 -o cu_mask cu_mask.hip
//   ./cu_mask ffffffff 55555555 0000ffff 77777777 11111111
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <set>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20: id | offset << 6 | (size - 1) << 11
__global__ void __launch_bounds__(256) where(uint32_t *out) {
  if (threadIdx.x == 0) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // stay a while, so that the launch spreads over every CU it may use
  double x = threadIdx.x;
  for (int i = 0; i < 20000; ++i)
    x = __fma_rn(x, 1.0000001, 1e-9);
  if (x == 42.)
    out[0] = 0;
}

// vector issue: eight independent chains of fp64 fma per lane
__global__ void __launch_bounds__(256) busy(double *out, int trips) {
  double a[8];
  for (int k = 0; k < 8; ++k)
    a[k] = threadIdx.x + k;
  for (int i = 0; i < trips; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      a[k] = __fma_rn(a[k], 1.0000001, 1e-9);
  double s = 0.;
  for (int k = 0; k < 8; ++k)
    s += a[k];
  if (s == 42.)
    out[0] = s;
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  const int nblocks = 4096;
  uint32_t *d_where;
  double *d_out;
  CHECK(hipMalloc(&d_where, sizeof(uint32_t) * 2 * nblocks));
  CHECK(hipMalloc(&d_out, sizeof(double)));
  std::vector<uint32_t> h(2 * nblocks);
  for (int m = 1; m < argc; ++m) {
    const uint32_t w = (uint32_t)strtoul(argv[m], nullptr, 16);
    uint32_t mask[8];
    for (int k = 0; k < 8; ++k)
      mask[k] = w;
    // a single word may also be given per position: w0,w1,...,w7
    if (strchr(argv[m], ',')) {
      char *copy = strdup(argv[m]);
      int k = 0;
      for (char *t = strtok(copy, ","); t && k < 8; t = strtok(nullptr, ","))
        mask[k++] = (uint32_t)strtoul(t, nullptr, 16);
      for (; k < 8; ++k)
        mask[k] = 0;
      free(copy);
    }
    hipStream_t s;
    CHECK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    where<<<nblocks, 256, 0, s>>>(d_where);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(h.data(), d_where, sizeof(uint32_t) * 2 * nblocks,
                    hipMemcpyDeviceToHost));
    std::map<uint32_t, std::set<uint32_t>> cus; // xcc -> {se, sh, cu}
    for (int b = 0; b < nblocks; ++b) {
      const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
      // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
      cus[xcc].insert((hw >> 8) & 0xff);
    }
    int total = 0;
    printf("mask %s:", argv[m]);
    for (auto &x : cus) {
      printf(" xcc%u:%zu", x.first, x.second.size());
      total += (int)x.second.size();
    }
    printf("  = %d CUs\n", total);
    if (m == 1 || getenv("CU_MASK_VERBOSE")) {
      for (auto &x : cus) {
        printf("   xcc%u se/sh/cu ids:", x.first);
        for (uint32_t id : x.second)
          printf(" %u.%u.%u", (id >> 5) & 7, (id >> 4) & 1, id & 15);
        printf("\n");
      }
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    busy<<<256 * 8, 256, 0, s>>>(d_out, 1000);
    CHECK(hipEventRecord(e0, s));
    busy<<<256 * 8, 256, 0, s>>>(d_out, 100000);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 2. * 8 * 100000. * 256 * 8 * 256;
    printf("   busy kernel %.2f ms, %.1f TFLOP/s fp64\n", ms,
           flops / ms * 1e-9);
    CHECK(hipStreamDestroy(s));
  }
  return 0;
}
