// Microbenchmark (GPU box): what scattered rows of a flight-slot array cost to
// read and to update in place, by row size - the question behind the tile
// rounds' slot layout (tile_kernels.h): a visit reads one 128-B slot (+ 128 B
// of weights) at a random place and rewrites its second 64 B.
//   hipcc -O3 --offload-arch=gfx950 -o row_gather row_gather.hip
// N rows are visited in a random order (a fixed permutation, as the tile
// order of the slots is), 8 or 4 lanes per row with 16 B per lane:
//   mode 0  read 128 B of a 128-B row                       (the slot today)
//   mode 1  read the first 64 B of a 128-B row              (half a line)
//   mode 2  read a 64-B row of a dense array of 64-B rows   (a "hot" array)
//   mode 3  mode 0 + write the second 64 B back             (a visit today)
//   mode 4  mode 2 + write the 64-B row back                (hot array visit)
//   mode 5  read 256 B: a 128-B row of each of two arrays   (slot + weights)
//   mode 6  mode 5 + write 64 B back                        (multi-ion visit)
//   mode 7  mode 0 in the order of the rows                 (a sorted array)
//   mode 8  read 128 B + write the whole 128 B back in place
//   mode 9  mode 3 + a 4-B key written to keys[row]         (H-only visit today:
//           two scattered partial-line writes)
//   mode 10 read 128 B at random, write 128 B + key at position i of a second
//           array                                           (append, dense)
//   mode 11 mode 6 + the scattered 4-B key                  (multi-ion visit today)
//   mode 12 read 128 + 128 B at random, write 128 B + key at position i
//   mode 13 write 64 B of random 128-B rows (no read)
//   mode 14 write random 128-B rows (no read)
// Prints GB/s of USEFUL bytes and ns per row. Footprint: N x 128 B (x 2 for
// modes 5, 6) - far beyond the 256 MiB Infinity Cache at the default N.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256)
    visit(double2 *rows, double2 *weights, const uint32_t *order, uint32_t n,
          double *sink, double2 *rows_out, uint32_t *keys) {
  constexpr int LANES = (MODE == 1 || MODE == 2 || MODE == 4) ? 4 : 8;
  constexpr int STRIDE = (MODE == 2 || MODE == 4) ? 4 : 8; // double2 per row
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
  const int part = (int)(tid % LANES);
  double acc = 0.;
  for (uint64_t i = tid / LANES; i < n; i += nthreads / LANES) {
    const uint32_t r = (MODE == 7) ? (uint32_t)i : order[i];
    double2 *row = rows + (size_t)STRIDE * r;
    if (MODE == 13 || MODE == 14) {
      if (MODE == 14 || part >= 4)
        row[part] = make_double2((double)i, 1.);
      continue;
    }
    const double2 v = row[part];
    acc += v.x + v.y;
    if (MODE == 8)
      row[part] = make_double2(v.x + 1., v.y);
    if (MODE == 9 || MODE == 11) {
      if (part >= 4)
        row[part] = make_double2(v.x + 1., v.y);
      if (part == 0)
        keys[r] = (uint32_t)i;
    }
    if (MODE == 10 || MODE == 12) {
      rows_out[(size_t)8 * i + part] = make_double2(v.x + 1., v.y);
      if (part == 0)
        keys[i] = r;
    }
    if (MODE == 5 || MODE == 6 || MODE == 11 || MODE == 12) {
      const double2 w = weights[(size_t)8 * r + part];
      acc += w.x * w.y;
    }
    if (MODE == 3 || MODE == 6) {
      if (part >= 4)
        row[part] = make_double2(v.x + 1., v.y);
    }
    if (MODE == 4)
      row[part] = make_double2(v.x + 1., v.y);
  }
  if (acc == 12345.678)
    *sink = acc;
}

template <int MODE>
void run(double2 *rows, double2 *weights, const uint32_t *order, uint32_t n,
         double *sink, const char *what, double useful_bytes_per_row,
         double2 *rows_out = nullptr, uint32_t *keys = nullptr) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const int blocks = 256 * 8;
  visit<MODE><<<blocks, 256>>>(rows, weights, order, n, sink, rows_out, keys);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int k = 0; k < 3; ++k)
    visit<MODE><<<blocks, 256>>>(rows, weights, order, n, sink, rows_out, keys);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= 3.f;
  printf("mode %d  %-52s %8.3f ms  %7.1f ps/row  %7.0f GB/s useful\n", MODE,
         what, ms, ms * 1e9 / n, useful_bytes_per_row * n / (ms * 1e6));
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atof(argv[1]) : 40000000u;
  double2 *rows, *weights;
  uint32_t *order;
  double *sink;
  CHECK(hipMalloc(&rows, (size_t)n * 128));
  CHECK(hipMalloc(&weights, (size_t)n * 128));
  CHECK(hipMalloc(&order, (size_t)n * 4));
  CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(rows, 0, (size_t)n * 128));
  CHECK(hipMemset(weights, 0, (size_t)n * 128));
  std::vector<uint32_t> h(n);
  for (uint32_t i = 0; i < n; ++i)
    h[i] = i;
  std::mt19937_64 gen(42);
  std::shuffle(h.begin(), h.end(), gen);
  CHECK(hipMemcpy(order, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  printf("%u rows, footprint %.1f GB per array\n", n, n * 128. / 1e9);
  run<0>(rows, weights, order, n, sink, "read 128-B rows, random", 128.);
  run<1>(rows, weights, order, n, sink, "read first 64 B of 128-B rows, random", 64.);
  run<2>(rows, weights, order, n, sink, "read 64-B rows of a dense 64-B array, random", 64.);
  run<3>(rows, weights, order, n, sink, "read 128 B + write back 64 B, random", 192.);
  run<4>(rows, weights, order, n, sink, "read 64-B row + write it back, random", 128.);
  run<5>(rows, weights, order, n, sink, "read 128 B + 128 B of two arrays, random", 256.);
  run<6>(rows, weights, order, n, sink, "read 128 + 128 B, write back 64 B, random", 320.);
  run<7>(rows, weights, order, n, sink, "read 128-B rows in order", 128.);
  double2 *rows_out;
  uint32_t *keys;
  CHECK(hipMalloc(&rows_out, (size_t)n * 128));
  CHECK(hipMalloc(&keys, (size_t)n * 4));
  run<8>(rows, weights, order, n, sink, "read 128 B + write 128 B back, random", 256.);
  run<9>(rows, weights, order, n, sink, "read 128, write back 64 B + scattered 4-B key", 196., rows_out, keys);
  run<10>(rows, weights, order, n, sink, "read 128 random, write 128 B + key DENSE", 260., rows_out, keys);
  run<11>(rows, weights, order, n, sink, "read 128+128, write back 64 B + scattered key", 324., rows_out, keys);
  run<12>(rows, weights, order, n, sink, "read 128+128 random, write 128 B + key DENSE", 388., rows_out, keys);
  run<13>(rows, weights, order, n, sink, "write 64 B of random 128-B rows", 64.);
  run<14>(rows, weights, order, n, sink, "write random 128-B rows", 128.);
  return 0;
}
