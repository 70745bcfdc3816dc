/*
 * cmio_stubs.c - ORACLE (test infrastructure): parts not restated yet.
 */
#include "cmio_internal.h"

#include <stdio.h>
#include <stdlib.h>

double cmio_spectrum_sample(const cmio_model *model, cmio_rng *rng) {
  (void)rng;
  if (model->spectrum_type == CMIO_SPECTRUM_MONOCHROMATIC) {
    /* src/MonochromaticPhotonSourceSpectrum.hpp:97-100: no random number */
    return model->mono_frequency;
  }
  fprintf(stderr, "cmio: spectrum type %d not implemented\n",
          model->spectrum_type);
  abort();
}

double cmio_reemit_frequency(const cmio_model *model, const cmio_photon *photon,
                             double AHe, double T, double xH, double xHe,
                             cmio_rng *rng, int32_t *type) {
  (void)photon; (void)AHe; (void)T; (void)xH; (void)xHe; (void)rng; (void)type;
  fprintf(stderr, "cmio: reemission type %d not implemented\n",
          model->reemit_type);
  abort();
}

void cmio_update_cells(const cmio_grid *grid, const cmio_model *model,
                       cmio_cells *cells, uint32_t loop, double totweight) {
  if (model->do_temperature && loop > (uint32_t)model->t_min_iteration) {
    fprintf(stderr, "cmio: temperature calculation not implemented\n");
    abort();
  }
  cmio_calculate_ionization_state(grid, model, cells, totweight);
}
