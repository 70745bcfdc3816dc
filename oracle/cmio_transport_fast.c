/*
 * cmio_transport_fast.c - ORACLE (test infrastructure): the CPU BASELINE form
 * of cmio_shoot. Same packets, same random numbers, same arithmetic per step
 * as cmio_transport.c, organised the way the reference organises its classic
 * path so that its speed is a fair stand-in for the reference's on the same
 * cores:
 *
 *  - cells are an array of structures (the reference's IonizationVariables,
 *    src/IonizationVariables.hpp:81-118): the three numbers a step reads and
 *    the accumulators it adds to sit in neighbouring cache lines, instead of
 *    19 separate arrays;
 *  - the accumulators of a cell are updated under ONE per-cell lock, as
 *    DensityGrid::update_integrals does (src/DensityGrid.hpp:150-197 with
 *    src/Lock.hpp), instead of 16 atomic read-modify-writes;
 *  - a hydrogen-only run (every other cross section identically zero) adds
 *    only what can be non-zero: J_H and the hydrogen heating term, lock-free
 *    (the reference's LOCKFREE / HYDROGEN_ONLY build options together,
 *    CMakeLists.txt:155-165, src/LockFree.hpp:46-52). That is more than the
 *    default reference build does for the Stromgren benchmarks (it adds all
 *    16 terms under the lock whatever the cross sections are), so the
 *    baseline is generous to the CPU.
 *
 * Results equal cmio_shoot's up to the order of the additions
 * (tests/test_oracle_pinning.py::test_fast_shoot_equals_shoot).
 */
#include "cmio_internal.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  double n, xH, xHe;
  double acc[16]; /* 14 mean intensities, 2 heating terms */
  int lock;
  int pad;
} fast_cell; /* 160 bytes */

static inline void cell_lock(fast_cell *c) {
  while (__atomic_exchange_n(&c->lock, 1, __ATOMIC_ACQUIRE)) {
    while (__atomic_load_n(&c->lock, __ATOMIC_RELAXED)) {
    }
  }
}
static inline void cell_unlock(fast_cell *c) {
  __atomic_store_n(&c->lock, 0, __ATOMIC_RELEASE);
}

static inline void atomic_add(double *p, double v) {
#pragma omp atomic
  *p += v;
}

/* cmio_interact (src/CartesianDensityGrid.cpp:375-452) on the AoS cells */
static int64_t interact_fast(const cmio_grid *grid, fast_cell *cells,
                             cmio_photon *photon, double optical_depth,
                             int hydrogen_only, double nuH, double nuHe,
                             const double cellside[3],
                             const double inverse_cellside[3]) {
  double origin[3] = {photon->position[0], photon->position[1],
                      photon->position[2]};
  int32_t index[3];
  for (int a = 0; a < 3; ++a)
    index[a] = (int32_t)((origin[a] - grid->anchor[a]) * inverse_cellside[a]);
  const double sigma_H = photon->cross_section[CMIO_ION_H_n];
  const double sigma_He = photon->cross_section[CMIO_ION_He_n];
  const double sigma_He_corr = photon->cross_section_He_corr;
  int64_t last_cell = -1;
  int inside = 1;
  for (;;) {
    /* is_inside, :187-227 */
    inside = 1;
    for (int a = 0; a < 3; ++a) {
      if (!grid->periodic[a]) {
        inside &= (index[a] >= 0 && index[a] < grid->ncell[a]);
      } else {
        if (index[a] < 0) {
          index[a] = grid->ncell[a] - 1;
          origin[a] += grid->sides[a];
        }
        if (index[a] >= grid->ncell[a]) {
          index[a] = 0;
          origin[a] -= grid->sides[a];
        }
      }
    }
    if (!inside || !(optical_depth > 0.))
      break;
    /* get_wall_intersection, :280-318 */
    double d[3];
    for (int a = 0; a < 3; ++a) {
      const double lo = grid->anchor[a] + cellside[a] * index[a];
      const double hi = lo + cellside[a];
      if (photon->direction[a] > 0.)
        d[a] = (hi - origin[a]) * photon->inverse_direction[a];
      else if (photon->direction[a] < 0.)
        d[a] = (lo - origin[a]) * photon->inverse_direction[a];
      else
        d[a] = DBL_MAX;
    }
    double ds = fmin(d[0], fmin(d[1], d[2]));
    int32_t next_index[3];
    double wall[3];
    for (int a = 0; a < 3; ++a) {
      next_index[a] =
          (d[a] == ds) ? ((photon->direction[a] > 0.) ? 1 : -1) : 0;
      wall[a] = origin[a] + ds * photon->direction[a];
    }
    const int64_t cell = ((int64_t)index[0] * grid->ncell[1] + index[1]) *
                             grid->ncell[2] +
                         index[2];
    last_cell = cell;
    fast_cell *c = &cells[cell];
    const double tau =
        ds * c->n * (sigma_H * c->xH + sigma_He_corr * c->xHe);
    optical_depth -= tau;
    if (optical_depth < 0.) {
      const double Scorr = ds * optical_depth / tau;
      for (int a = 0; a < 3; ++a)
        origin[a] += (wall[a] - origin[a]) * (ds + Scorr) / ds;
      ds += Scorr;
    } else {
      for (int a = 0; a < 3; ++a) {
        origin[a] = wall[a];
        index[a] += next_index[a];
      }
    }
    /* update_integrals, src/DensityGrid.hpp:150-197 */
    if (c->n > 0.) {
      const double dsw = ds * photon->weight;
      if (hydrogen_only) {
        atomic_add(&c->acc[0], dsw * sigma_H);
        atomic_add(&c->acc[14], dsw * sigma_H * (photon->energy - nuH));
      } else {
        cell_lock(c);
        for (int ion = 0; ion < CMIO_NION; ++ion)
          c->acc[ion] += dsw * photon->cross_section[ion];
        c->acc[14] += dsw * sigma_H * (photon->energy - nuH);
        c->acc[15] += dsw * sigma_He * (photon->energy - nuHe);
        cell_unlock(c);
      }
    }
  }
  for (int a = 0; a < 3; ++a)
    photon->position[a] = origin[a];
  return inside ? last_cell : -1;
}

void cmio_shoot_fast(const cmio_grid *grid, const cmio_model *model,
                     cmio_cells *cells, uint32_t seed, uint32_t iteration,
                     uint64_t first_packet, uint64_t n_packets,
                     double *totweight, double typecount[CMIO_NTYPE]) {
  const int64_t ncell =
      (int64_t)grid->ncell[0] * grid->ncell[1] * grid->ncell[2];
  fast_cell *aos = (fast_cell *)malloc(sizeof(fast_cell) * (size_t)ncell);
  if (!aos) {
    cmio_set_error("cmio_shoot_fast: out of memory");
    return;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < ncell; ++i) {
    aos[i].n = cells->number_density[i];
    aos[i].xH = cells->ionic_fraction[CMIO_ION_H_n][i];
    aos[i].xHe = cells->ionic_fraction[CMIO_ION_He_n][i];
    memset(aos[i].acc, 0, sizeof aos[i].acc);
    aos[i].lock = 0;
  }
  int hydrogen_only = model->xsec_type == CMIO_XSEC_FIXED;
  for (int ion = 1; ion < CMIO_NION && hydrogen_only; ++ion)
    hydrogen_only = model->xsec_fixed[ion] == 0.;
  double cellside[3], inverse_cellside[3];
  for (int a = 0; a < 3; ++a) {
    cellside[a] = grid->sides[a] / grid->ncell[a];
    inverse_cellside[a] = 1. / cellside[a];
  }
  const double nuH = cmio_eV_to_Hz(13.6), nuHe = cmio_eV_to_Hz(24.6);

  double tw = 0., tc0 = 0., tc1 = 0., tc2 = 0., tc3 = 0.;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : tw, tc0, tc1, tc2, tc3)
  for (uint64_t i = 0; i < n_packets; ++i) {
    cmio_rng rng = {seed, iteration, first_packet + i, 0, NULL};
    cmio_photon photon;
    double tau;
    cmio_emit_stream(model, &rng, &photon, &tau);
    int64_t cell = interact_fast(grid, aos, &photon, tau, hydrogen_only, nuH,
                                 nuHe, cellside, inverse_cellside);
    while (cell >= 0) {
      /* PhotonSource::reemit, src/PhotonSource.cpp:272-308: the cell's
       * temperature and fractions do not change during an iteration */
      cmio_cells view = *cells;
      if (!cmio_reemit_stream(model, &view, cell, &photon, &rng))
        break;
      tau = -log(cmio_rng_next(&rng));
      cell = interact_fast(grid, aos, &photon, tau, hydrogen_only, nuH, nuHe,
                           cellside, inverse_cellside);
    }
    tw += photon.weight;
    switch (photon.type) {
    case CMIO_TYPE_PRIMARY:
      tc0 += photon.weight;
      break;
    case CMIO_TYPE_DIFFUSE_HI:
      tc1 += photon.weight;
      break;
    case CMIO_TYPE_DIFFUSE_HeI:
      tc2 += photon.weight;
      break;
    default:
      tc3 += photon.weight;
      break;
    }
  }
  *totweight += tw;
  typecount[0] += tc0;
  typecount[1] += tc1;
  typecount[2] += tc2;
  typecount[3] += tc3;
  /* the iteration's tallies, added to the caller's arrays */
  for (int k = 0; k < 16; ++k) {
    if (hydrogen_only && k != 0 && k != 14)
      continue;
    double *dst = k < CMIO_NION ? cells->mean_intensity[k]
                                : cells->heating[k - CMIO_NION];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ncell; ++i)
      dst[i] += aos[i].acc[k];
  }
  free(aos);
}
