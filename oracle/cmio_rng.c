/*
 * cmio_rng.c - ORACLE (test infrastructure): the packet random number stream.
 *
 * Replaces src/RandomGenerator.hpp:39-272 (sequential ranlxd2, one stream per
 * thread) by a counter-based generator so that a packet's random numbers do
 * not depend on which thread/lane processes it. The reference's contract that
 * is kept: get_uniform_random_double() returns an independent uniform double
 * (src/RandomGenerator.hpp:215-225); we return it in the open interval (0,1)
 * so that -log(xi) stays finite (the reference can return exactly 0).
 */
#include "cmio.h"

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void cmio_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                        uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += PHILOX_W0;
    k1 += PHILOX_W1;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

double cmio_rng_uniform(uint32_t seed, uint32_t iteration, uint64_t packet,
                        uint32_t draw) {
  const uint32_t ctr[4] = {(uint32_t)packet, (uint32_t)(packet >> 32),
                           draw >> 1, 0u};
  const uint32_t key[2] = {seed, iteration};
  uint32_t r[4];
  cmio_philox4x32_10(ctr, key, r);
  const uint32_t lo = r[2 * (draw & 1u)];
  const uint32_t hi = r[2 * (draw & 1u) + 1];
  const uint64_t bits = (((uint64_t)hi << 32) | lo) >> 12;
  /* 52 random bits + 0.5 is exactly representable: u in (0,1) strictly */
  return ((double)bits + 0.5) * 0x1.0p-52;
}

#ifdef _OPENMP
#include <omp.h>
#endif
void cmio_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0)
    omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* number of OpenMP threads cmio_shoot / cmio_update_cells will use */
int cmio_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* index of the calling thread inside a parallel region (0 outside one) */
int cmio_thread_index(void) {
#ifdef _OPENMP
  return omp_get_thread_num();
#else
  return 0;
#endif
}
