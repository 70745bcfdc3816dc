/*
 * cmio_stubs.c - ORACLE (test infrastructure): per-iteration cell update
 * dispatch.
 */
#include "cmio_internal.h"

#include <stdio.h>
#include <stdlib.h>

/* src/TemperatureCalculator.cpp:944-970 */
void cmio_update_cells(const cmio_grid *grid, const cmio_model *model,
                       cmio_cells *cells, uint32_t loop, double totweight) {
  if (model->do_temperature && loop > (uint32_t)model->t_min_iteration) {
    cmio_calculate_temperature(grid, model, cells, totweight);
  } else {
    cmio_calculate_ionization_state(grid, model, cells, totweight);
  }
}
