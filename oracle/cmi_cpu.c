/*
 * cmi_cpu.c - the CPU twin of the engine's C ABI: the core
 * entry points of include/cmi_gpu.h under the names cmi_cpu_*, with the same
 * argument lists, error codes and call-sequence rules, on top of the oracle
 * (cmio_*.c).
 *
 * THIS IS TEST INFRASTRUCTURE like the rest of oracle/: only tests/ load
 * libcmi_cpu.so - so that a parity test can drive BOTH libraries through one
 * and the same sequence of ABI calls (tests/test_cpu_twin.py,
 * tests/abi_driver.py). The product (cmacionize_amd/) never loads it and has,
 * by design, no CPU path. The precedent in the reference is its C interface
 * over the CPU simulation, src/CMILibrary.hpp:46-72.
 *
 * The prototypes are checked against include/cmi_gpu.h at compile time (the
 * table at the end of this file): a cmi_cpu_ function whose argument list
 * drifts from its cmi_gpu_ twin does not compile.
 *
 * Not in the twin (cmi_cpu_has() says so): continuous sources, trackers,
 * emissivities, exports / flights, the group API, tunings, probes.
 */
#include "../include/cmi_gpu.h"
#include "cmio.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct cmi_gpu_engine { /* (the ABI's opaque handle: here the twin's own) */
  cmio_grid grid;
  cmio_model model;
  cmio_cells cells;
  int64_t ncell;
  double *state;  /* [16][ncell]: n, T, x[14] */
  double *acc;    /* [16][ncell]: J[14], heating[2] */
  double *source_position, *source_cumulative;
  int have_sources, have_spectrum, have_xsec, have_recomb, have_cells;
  int tables_stale;
  double totweight, typecount[CMIO_NTYPE];
};

static __thread char last_error[512];

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error, sizeof last_error, fmt, ap);
  va_end(ap);
  return code;
}

const char *cmi_cpu_last_error(void) { return last_error; }

int cmi_cpu_create(const cmi_gpu_config *config, cmi_gpu_engine **out) {
  if (!config || !out)
    return fail(CMI_GPU_EINVAL, "null argument");
  for (int a = 0; a < 3; ++a)
    if (config->ncell[a] <= 0 || !(config->sides[a] > 0.))
      return fail(CMI_GPU_EINVAL, "ncell and sides must be positive");
  if (config->sub_ncell[0] | config->sub_ncell[1] | config->sub_ncell[2])
    return fail(CMI_GPU_EINVAL, "the CPU twin holds whole grids only");
  cmi_gpu_engine *e = (cmi_gpu_engine *)calloc(1, sizeof *e);
  if (!e)
    return fail(CMI_GPU_ENOMEM, "out of memory");
  e->ncell = 1;
  for (int a = 0; a < 3; ++a) {
    e->grid.anchor[a] = config->anchor[a];
    e->grid.sides[a] = config->sides[a];
    e->grid.ncell[a] = config->ncell[a];
    e->grid.periodic[a] = config->periodic[a];
    e->ncell *= config->ncell[a];
  }
  e->state = (double *)calloc((size_t)16 * e->ncell, sizeof(double));
  e->acc = (double *)calloc((size_t)16 * e->ncell, sizeof(double));
  if (!e->state || !e->acc) {
    free(e->state);
    free(e->acc);
    free(e);
    return fail(CMI_GPU_ENOMEM, "out of memory");
  }
  e->cells.number_density = e->state;
  e->cells.temperature = e->state + e->ncell;
  for (int i = 0; i < CMIO_NION; ++i) {
    e->cells.ionic_fraction[i] = e->state + (size_t)(2 + i) * e->ncell;
    e->cells.mean_intensity[i] = e->acc + (size_t)i * e->ncell;
  }
  e->cells.heating[0] = e->acc + (size_t)14 * e->ncell;
  e->cells.heating[1] = e->acc + (size_t)15 * e->ncell;
  /* the defaults of the engine: no re-emission, no temperature solve, the
   * TemperatureCalculator's defaults (include/cmi_gpu.h) */
  e->model.reemit_type = CMIO_REEMIT_NONE;
  e->model.t_min_iteration = 3;
  e->model.t_epsilon = 1.e-3;
  e->model.t_max_iterations = 100;
  e->model.crlim = 0.75;
  e->model.crscale = 1.33333 * 3.086e19;
  e->model.t_min_ionized = 4000.;
  *out = e;
  return CMI_GPU_OK;
}

int cmi_cpu_destroy(cmi_gpu_engine *e) {
  if (!e)
    return CMI_GPU_OK;
  if (e->model.tables)
    cmio_tables_free((cmio_tables *)e->model.tables);
  free(e->source_position);
  free(e->source_cumulative);
  free(e->state);
  free(e->acc);
  free(e);
  return CMI_GPU_OK;
}

int cmi_cpu_synchronize(cmi_gpu_engine *e) {
  return e ? CMI_GPU_OK : fail(CMI_GPU_EINVAL, "null engine");
}

int64_t cmi_cpu_number_of_cells(const cmi_gpu_engine *e) {
  return e ? e->ncell : 0;
}

int cmi_cpu_set_sources(cmi_gpu_engine *e, int32_t n, const double *positions,
                        const double *weights, double total_luminosity) {
  if (!e || n < 0 || (n > 0 && (!positions || !weights)))
    return fail(CMI_GPU_EINVAL, "set_sources: bad argument");
  free(e->source_position);
  free(e->source_cumulative);
  e->source_position = (double *)malloc(sizeof(double) * 3 * (n ? n : 1));
  e->source_cumulative = (double *)malloc(sizeof(double) * (n ? n : 1));
  double sum = 0.;
  for (int32_t i = 0; i < n; ++i) {
    for (int a = 0; a < 3; ++a)
      e->source_position[3 * i + a] = positions[3 * i + a];
    sum += weights[i];
    e->source_cumulative[i] = sum;
  }
  if (n > 0)
    e->source_cumulative[n - 1] = 1.;
  e->model.nsource = n;
  e->model.source_position = e->source_position;
  e->model.source_cumulative = e->source_cumulative;
  e->model.total_luminosity = total_luminosity;
  e->model.discrete_luminosity = total_luminosity;
  e->have_sources = 1;
  return CMI_GPU_OK;
}

int cmi_cpu_set_spectrum_monochromatic(cmi_gpu_engine *e, double frequency) {
  if (!e || !(frequency > 0.))
    return fail(CMI_GPU_EINVAL, "set_spectrum_monochromatic: bad argument");
  e->model.spectrum_type = CMIO_SPECTRUM_MONOCHROMATIC;
  e->model.mono_frequency = frequency;
  e->have_spectrum = 1;
  e->tables_stale = 1;
  return CMI_GPU_OK;
}

int cmi_cpu_set_spectrum_planck(cmi_gpu_engine *e, double temperature) {
  if (!e || !(temperature > 0.))
    return fail(CMI_GPU_EINVAL, "set_spectrum_planck: bad argument");
  e->model.spectrum_type = CMIO_SPECTRUM_PLANCK;
  e->model.planck_temperature = temperature;
  e->have_spectrum = 1;
  e->tables_stale = 1;
  return CMI_GPU_OK;
}

int cmi_cpu_set_cross_sections_fixed(cmi_gpu_engine *e, const double *sigma) {
  if (!e || !sigma)
    return fail(CMI_GPU_EINVAL, "set_cross_sections_fixed: bad argument");
  e->model.xsec_type = CMIO_XSEC_FIXED;
  memcpy(e->model.xsec_fixed, sigma, sizeof e->model.xsec_fixed);
  e->have_xsec = 1;
  e->tables_stale = 1;
  return CMI_GPU_OK;
}

int cmi_cpu_set_cross_sections_verner(cmi_gpu_engine *e) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  e->model.xsec_type = CMIO_XSEC_VERNER;
  e->have_xsec = 1;
  e->tables_stale = 1;
  return CMI_GPU_OK;
}

int cmi_cpu_set_recombination_rates_fixed(cmi_gpu_engine *e,
                                          const double *alpha) {
  if (!e || !alpha)
    return fail(CMI_GPU_EINVAL, "set_recombination_rates_fixed: bad argument");
  e->model.recomb_type = CMIO_RECOMB_FIXED;
  memcpy(e->model.recomb_fixed, alpha, sizeof e->model.recomb_fixed);
  e->have_recomb = 1;
  return CMI_GPU_OK;
}

int cmi_cpu_set_recombination_rates_verner(cmi_gpu_engine *e) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  e->model.recomb_type = CMIO_RECOMB_VERNER;
  e->have_recomb = 1;
  return CMI_GPU_OK;
}

int cmi_cpu_set_abundances(cmi_gpu_engine *e, const double *abundances) {
  if (!e || !abundances)
    return fail(CMI_GPU_EINVAL, "set_abundances: bad argument");
  /* (He, C, N, O, Ne, S: include/cmi_gpu.h) */
  for (int i = 0; i < 6; ++i)
    e->model.abundance[CMIO_EL_He + i] = abundances[i];
  e->tables_stale = 1;
  return CMI_GPU_OK;
}

int cmi_cpu_set_reemission(cmi_gpu_engine *e, int32_t type,
                           double fixed_probability, double fixed_frequency) {
  if (!e || type < CMI_GPU_REEMIT_NONE || type > CMI_GPU_REEMIT_FIXED)
    return fail(CMI_GPU_EINVAL, "set_reemission: bad argument");
  e->model.reemit_type = type; /* (the same numbering) */
  e->model.reemit_fixed_probability = fixed_probability;
  e->model.reemit_fixed_frequency = fixed_frequency;
  e->tables_stale = 1;
  return CMI_GPU_OK;
}

int cmi_cpu_set_temperature_params(cmi_gpu_engine *e,
                                   const cmi_gpu_temperature_params *p) {
  if (!e || !p)
    return fail(CMI_GPU_EINVAL, "set_temperature_params: bad argument");
  e->model.do_temperature = p->do_temperature_calculation;
  e->model.t_min_iteration = p->minimum_number_of_iterations;
  e->model.t_epsilon = p->epsilon_convergence;
  e->model.t_max_iterations = p->maximum_number_of_iterations;
  e->model.pahfac = p->pah_heating_factor;
  e->model.crfac = p->cosmic_ray_heating_factor;
  e->model.crlim = p->cosmic_ray_heating_limit;
  e->model.crscale = p->cosmic_ray_heating_scale_length;
  e->model.t_min_ionized = p->minimum_ionized_temperature;
  return CMI_GPU_OK;
}

int cmi_cpu_upload_cells(cmi_gpu_engine *e, const double *number_density,
                         const double *temperature,
                         const double *ionic_fractions) {
  if (!e || !number_density || !temperature || !ionic_fractions)
    return fail(CMI_GPU_EINVAL, "upload_cells: bad argument");
  const size_t bytes = sizeof(double) * (size_t)e->ncell;
  memcpy(e->cells.number_density, number_density, bytes);
  memcpy(e->cells.temperature, temperature, bytes);
  memcpy(e->cells.ionic_fraction[0], ionic_fractions, CMIO_NION * bytes);
  e->have_cells = 1;
  return CMI_GPU_OK;
}

static double *field_pointer(cmi_gpu_engine *e, int32_t field) {
  if (field < 0 || field >= CMI_GPU_NFIELD)
    return NULL;
  return field < CMI_GPU_FIELD_MEAN_INTENSITY
             ? e->state + (size_t)field * e->ncell
             : e->acc + (size_t)(field - CMI_GPU_FIELD_MEAN_INTENSITY) *
                            e->ncell;
}

int cmi_cpu_upload_field(cmi_gpu_engine *e, int32_t field,
                         const double *values) {
  double *dst = (e && values) ? field_pointer(e, field) : NULL;
  if (!dst)
    return fail(CMI_GPU_EINVAL, "upload_field: unknown field %d", field);
  memcpy(dst, values, sizeof(double) * (size_t)e->ncell);
  return CMI_GPU_OK;
}

int cmi_cpu_download_field(cmi_gpu_engine *e, int32_t field, double *values) {
  const double *src = (e && values) ? field_pointer(e, field) : NULL;
  if (!src)
    return fail(CMI_GPU_EINVAL, "download_field: unknown field %d", field);
  memcpy(values, src, sizeof(double) * (size_t)e->ncell);
  return CMI_GPU_OK;
}

int cmi_cpu_reset_grid(cmi_gpu_engine *e) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  cmio_reset_grid(&e->grid, &e->cells);
  e->totweight = 0.;
  memset(e->typecount, 0, sizeof e->typecount);
  return CMI_GPU_OK;
}

int cmi_cpu_shoot(cmi_gpu_engine *e, uint32_t seed, uint32_t iteration,
                  uint64_t first_packet, uint64_t n_packets) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (!e->have_sources || !e->have_spectrum || !e->have_xsec ||
      !e->have_cells)
    return fail(CMI_GPU_ESTATE,
                "cmi_cpu_shoot: sources, their spectra, cross sections and "
                "cell data must be set first");
  if (n_packets == 0)
    return CMI_GPU_OK;
  if (n_packets >= (1ull << 32))
    return fail(CMI_GPU_EINVAL,
                "cmi_cpu_shoot: at most 2^32 - 1 packets per call");
  if (e->tables_stale || !e->model.tables) {
    if (e->model.tables)
      cmio_tables_free((cmio_tables *)e->model.tables);
    e->model.tables = cmio_tables_create(&e->model);
    e->tables_stale = 0;
  }
  cmio_clear_error();
  cmio_shoot(&e->grid, &e->model, &e->cells, seed, iteration, first_packet,
             n_packets, &e->totweight, e->typecount);
  if (cmio_last_error())
    return fail(CMI_GPU_ESTATE, "%s", cmio_last_error());
  return CMI_GPU_OK;
}

int cmi_cpu_get_counters(cmi_gpu_engine *e, double *totweight,
                         double *typecount, uint64_t *nsteps) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (totweight)
    *totweight = e->totweight;
  if (typecount)
    memcpy(typecount, e->typecount, sizeof e->typecount);
  if (nsteps)
    *nsteps = 0; /* (the oracle's classic loop does not count cell crossings) */
  return CMI_GPU_OK;
}

int cmi_cpu_update_cells_range(cmi_gpu_engine *e, uint32_t loop,
                               double totweight, int64_t first_cell,
                               int64_t ncell) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (first_cell < 0 || ncell < 0 || first_cell + ncell > e->ncell)
    return fail(CMI_GPU_EINVAL, "update_cells_range: cells [%lld, %lld) are "
                "not inside the engine's %lld cells", (long long)first_cell,
                (long long)(first_cell + ncell), (long long)e->ncell);
  if (ncell == 0)
    return CMI_GPU_OK;
  if (!e->have_sources || !e->have_recomb || !e->have_cells)
    return fail(CMI_GPU_ESTATE,
                "cmi_cpu_update_cells: sources, recombination rates and cell "
                "data must be set first");
  if (!(totweight > 0.))
    return fail(CMI_GPU_EINVAL, "update_cells: totweight must be positive");
  cmio_clear_error();
  cmio_update_cells_range(&e->grid, &e->model, &e->cells, loop, totweight,
                          first_cell, ncell);
  if (cmio_last_error())
    return fail(CMI_GPU_ESTATE, "%s", cmio_last_error());
  return CMI_GPU_OK;
}

int cmi_cpu_update_cells(cmi_gpu_engine *e, uint32_t loop, double totweight) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  return cmi_cpu_update_cells_range(e, loop, totweight, 0, e->ncell);
}

/* the entry points of include/cmi_gpu.h the twin has, by their ABI name */
static const char *const TWINS[] = {
    "cmi_gpu_create", "cmi_gpu_destroy", "cmi_gpu_last_error",
    "cmi_gpu_synchronize", "cmi_gpu_number_of_cells", "cmi_gpu_set_sources",
    "cmi_gpu_set_spectrum_monochromatic", "cmi_gpu_set_spectrum_planck",
    "cmi_gpu_set_cross_sections_fixed", "cmi_gpu_set_cross_sections_verner",
    "cmi_gpu_set_recombination_rates_fixed",
    "cmi_gpu_set_recombination_rates_verner", "cmi_gpu_set_abundances",
    "cmi_gpu_set_reemission", "cmi_gpu_set_temperature_params",
    "cmi_gpu_upload_cells", "cmi_gpu_upload_field", "cmi_gpu_download_field",
    "cmi_gpu_reset_grid", "cmi_gpu_shoot", "cmi_gpu_get_counters",
    "cmi_gpu_update_cells", "cmi_gpu_update_cells_range"};

int cmi_cpu_has(const char *abi_name) {
  for (size_t k = 0; k < sizeof TWINS / sizeof TWINS[0]; ++k)
    if (abi_name && strcmp(abi_name, TWINS[k]) == 0)
      return 1;
  return 0;
}

/* compile-time check: every twin has the argument list of the entry point it
 * mirrors (an assignment between incompatible function pointer types is an
 * error with -Werror=incompatible-pointer-types, which the Makefile sets) */
#define TWIN(name)                                                             \
  static __typeof__(cmi_gpu_##name) *const check_##name                       \
      __attribute__((unused)) = cmi_cpu_##name
TWIN(create);
TWIN(destroy);
TWIN(last_error);
TWIN(synchronize);
TWIN(number_of_cells);
TWIN(set_sources);
TWIN(set_spectrum_monochromatic);
TWIN(set_spectrum_planck);
TWIN(set_cross_sections_fixed);
TWIN(set_cross_sections_verner);
TWIN(set_recombination_rates_fixed);
TWIN(set_recombination_rates_verner);
TWIN(set_abundances);
TWIN(set_reemission);
TWIN(set_temperature_params);
TWIN(upload_cells);
TWIN(upload_field);
TWIN(download_field);
TWIN(reset_grid);
TWIN(shoot);
TWIN(get_counters);
TWIN(update_cells);
TWIN(update_cells_range);
