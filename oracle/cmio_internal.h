/*
 * cmio_internal.h - ORACLE (test infrastructure): internal declarations.
 */
#ifndef CMIO_INTERNAL_H
#define CMIO_INTERNAL_H

#include "cmio.h"

/* per-packet random stream cursor (see cmio_rng.c) */
typedef struct {
  uint32_t seed;
  uint32_t iteration;
  uint64_t packet;
  uint32_t draw;
  /* tests only (cmio_reemit_scripted): when set, draw d returns script[d]
   * instead of the Philox stream, so that a test can steer a function
   * through a chosen branch */
  const double *script;
} cmio_rng;

static inline double cmio_rng_next(cmio_rng *rng) {
  if (rng->script)
    return rng->script[rng->draw++];
  return cmio_rng_uniform(rng->seed, rng->iteration, rng->packet,
                          rng->draw++);
}

/* src/PhysicalConstants.hpp:61-131 */
#define CMIO_PLANCK 6.626070040e-34
#define CMIO_BOLTZMANN 1.38064852e-23
#define CMIO_LIGHTSPEED 299792458.
#define CMIO_ELECTRONVOLT 1.6021766208e-19
#define CMIO_ELECTRON_MASS 9.10938356e-31
#define CMIO_RYDBERG 2.179872325e-18

/* UnitConverter::to_SI<QUANTITY_FREQUENCY>(x, "eV"):
 * src/UnitConverter.hpp:156-159,266-300 */
static inline double cmio_eV_to_Hz(double eV) {
  return eV * CMIO_ELECTRONVOLT * (1. / CMIO_PLANCK) / 1.;
}

/* src/Utilities.hpp:726-742 */
static inline uint_fast32_t cmio_locate(double x, const double *xarr,
                                        uint_fast32_t length) {
  /* bisection on the half-open bracket [lo, hi): invariant xarr[lo] < x (or
   * lo == 0) and x <= xarr[hi] (or hi == length); result clamped so that
   * result + 1 is a valid index */
  uint_fast32_t lo = 0, hi = length;
  while (hi - lo > 1) {
    const uint_fast32_t mid = (lo + hi) >> 1;
    if (x > xarr[mid])
      lo = mid;
    else
      hi = mid;
  }
  return (lo == length - 1) ? lo - 1 : lo;
}

/* physics hooks implemented in cmio_physics.c */
double cmio_cross_section(const cmio_model *model, int ion, double frequency);
double cmio_recombination_rate(const cmio_model *model, int ion, double T);
double cmio_spectrum_sample(const cmio_model *model, cmio_rng *rng);
double cmio_continuous_spectrum_sample(const cmio_model *model, cmio_rng *rng);
/* returns new frequency (0 = absorbed for good) and sets *type */
double cmio_reemit_frequency(const cmio_model *model, const cmio_photon *photon,
                             double AHe, double T, double xH, double xHe,
                             cmio_rng *rng, int32_t *type);

/* the per-packet pieces of cmio_shoot, for cmio_transport_fast.c: emission
 * + first optical depth, and PhotonSource::reemit */
void cmio_emit_stream(const cmio_model *model, cmio_rng *rng,
                      cmio_photon *photon, double *tau);
int cmio_reemit_stream(const cmio_model *model, const cmio_cells *cells,
                       int64_t cell, cmio_photon *photon, cmio_rng *rng);

#endif
