/*
 * cmio_update.c - ORACLE (test infrastructure): per-iteration cell update
 * dispatch.
 */
#include "cmio_internal.h"

#include <stdio.h>
#include <stdlib.h>

/* src/TemperatureCalculator.cpp:944-970 */
void cmio_update_cells(const cmio_grid *grid, const cmio_model *model,
                       cmio_cells *cells, uint32_t loop, double totweight) {
  cmio_update_cells_range(
      grid, model, cells, loop, totweight, 0,
      (int64_t)grid->ncell[0] * grid->ncell[1] * grid->ncell[2]);
}

/* ... for a block of cells (the reference's MPI path,
 * src/IonizationSimulation.cpp:532-537) */
void cmio_update_cells_range(const cmio_grid *grid, const cmio_model *model,
                             cmio_cells *cells, uint32_t loop,
                             double totweight, int64_t first, int64_t count) {
  if (model->do_temperature && loop > (uint32_t)model->t_min_iteration) {
    cmio_calculate_temperature_range(grid, model, cells, totweight, first,
                                     count);
  } else {
    cmio_calculate_ionization_state_range(grid, model, cells, totweight, first,
                                          count);
  }
}
