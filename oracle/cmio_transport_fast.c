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
 *  - the cells within a few cells of a discrete source - every packet of
 *    the source starts there, so with many threads their cache lines are
 *    fought over by all of them - get one PRIVATE set of accumulators per
 *    thread, summed into the cells when the packets are done. That is the
 *    remedy of the reference's task-based path: copies of the subgrid that
 *    holds the source (src/DensitySubGridCreator.hpp:437-531), added up by
 *    update_original (:556-574). CMIO_FAST_HOT_RADIUS (environment, cells,
 *    default 8; 0 switches it off) sets the half-width of the cube.
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

/* the cubes of cells around the discrete sources whose accumulators every
 * thread keeps privately */
#define CMIO_FAST_MAX_HOT 8
typedef struct {
  int n;                             /* cubes */
  int32_t lo[CMIO_FAST_MAX_HOT][3];  /* first cell */
  int32_t size[CMIO_FAST_MAX_HOT][3];
  int64_t first[CMIO_FAST_MAX_HOT + 1]; /* first private slot of a cube */
  int nvalue;                        /* accumulators per slot: 2 or 16 */
} hot_region;

/* private slot of cell `index`, or -1 */
static inline int64_t hot_slot(const hot_region *hot, const int32_t index[3]) {
  for (int b = 0; b < hot->n; ++b) {
    const uint32_t dx = (uint32_t)(index[0] - hot->lo[b][0]);
    const uint32_t dy = (uint32_t)(index[1] - hot->lo[b][1]);
    const uint32_t dz = (uint32_t)(index[2] - hot->lo[b][2]);
    if (dx < (uint32_t)hot->size[b][0] && dy < (uint32_t)hot->size[b][1] &&
        dz < (uint32_t)hot->size[b][2])
      return hot->first[b] +
             ((int64_t)dx * hot->size[b][1] + dy) * hot->size[b][2] + dz;
  }
  return -1;
}

/* cmio_interact (src/CartesianDensityGrid.cpp:375-452) on the AoS cells */
static int64_t interact_fast(const cmio_grid *grid, fast_cell *cells,
                             cmio_photon *photon, double optical_depth,
                             int hydrogen_only, double nuH, double nuHe,
                             const double cellside[3],
                             const double inverse_cellside[3],
                             const hot_region *hot, double *mine) {
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
    const int64_t slot = hot->n != 0 ? hot_slot(hot, index) : -1;
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
      if (slot >= 0) {
        /* this thread's own copy: no lock, no atomic */
        double *acc = mine + slot * hot->nvalue;
        if (hydrogen_only) {
          acc[0] += dsw * sigma_H;
          acc[1] += dsw * sigma_H * (photon->energy - nuH);
        } else {
          for (int ion = 0; ion < CMIO_NION; ++ion)
            acc[ion] += dsw * photon->cross_section[ion];
          acc[14] += dsw * sigma_H * (photon->energy - nuH);
          acc[15] += dsw * sigma_He * (photon->energy - nuHe);
        }
      } else if (hydrogen_only) {
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

  /* the cubes around the discrete sources (their union, cube by cube: a
   * cell inside two cubes belongs to the first) */
  hot_region hot;
  memset(&hot, 0, sizeof hot);
  hot.nvalue = hydrogen_only ? 2 : 16;
  {
    int radius = 8;
    const char *env = getenv("CMIO_FAST_HOT_RADIUS");
    if (env)
      radius = atoi(env);
    if (cmio_num_threads() < 2)
      radius = 0; /* nobody to fight with */
    for (int s = 0; s < model->nsource && radius > 0 &&
                    hot.n < CMIO_FAST_MAX_HOT; ++s) {
      int b = hot.n;
      int64_t cube = 1;
      for (int a = 0; a < 3; ++a) {
        const double x = model->source_position[3 * s + a];
        int32_t at = (int32_t)floor((x - grid->anchor[a]) *
                                    inverse_cellside[a]);
        int32_t lo = at - radius, hi = at + radius; /* [lo, hi) */
        if (lo < 0)
          lo = 0;
        if (hi > grid->ncell[a])
          hi = grid->ncell[a];
        if (hi < lo)
          hi = lo;
        hot.lo[b][a] = lo;
        hot.size[b][a] = hi - lo;
        cube *= hi - lo;
      }
      if (cube == 0)
        continue; /* a source outside the box */
      hot.first[b + 1] = hot.first[b] + cube;
      ++hot.n;
    }
  }
  const int64_t nslot = hot.first[hot.n];
  const int nthread = cmio_num_threads();
  double *private_acc = NULL;
  if (nslot != 0) {
    private_acc = (double *)calloc((size_t)nthread * (size_t)nslot *
                                       (size_t)hot.nvalue,
                                   sizeof(double));
    if (!private_acc) {
      cmio_set_error("cmio_shoot_fast: out of memory");
      free(aos);
      return;
    }
  }

  double tw = 0., tc0 = 0., tc1 = 0., tc2 = 0., tc3 = 0.;
#pragma omp parallel reduction(+ : tw, tc0, tc1, tc2, tc3)
  {
    double *mine = private_acc
                       ? private_acc + (size_t)cmio_thread_index() *
                                           (size_t)nslot * (size_t)hot.nvalue
                       : NULL;
#pragma omp for schedule(dynamic, 1024)
    for (uint64_t i = 0; i < n_packets; ++i) {
      cmio_rng rng = {seed, iteration, first_packet + i, 0, NULL};
      cmio_photon photon;
      double tau;
      cmio_emit_stream(model, &rng, &photon, &tau);
      int64_t cell = interact_fast(grid, aos, &photon, tau, hydrogen_only,
                                   nuH, nuHe, cellside, inverse_cellside,
                                   &hot, mine);
      while (cell >= 0) {
        /* PhotonSource::reemit, src/PhotonSource.cpp:272-308: the cell's
         * temperature and fractions do not change during an iteration */
        cmio_cells view = *cells;
        if (!cmio_reemit_stream(model, &view, cell, &photon, &rng))
          break;
        tau = -log(cmio_rng_next(&rng));
        cell = interact_fast(grid, aos, &photon, tau, hydrogen_only, nuH,
                             nuHe, cellside, inverse_cellside, &hot, mine);
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
  }
  /* DensitySubGridCreator::update_original: the copies into the cells, in
   * thread order */
  for (int b = 0; b < hot.n; ++b) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < hot.first[b + 1] - hot.first[b]; ++q) {
      const int32_t dz = (int32_t)(q % hot.size[b][2]);
      const int32_t dy = (int32_t)((q / hot.size[b][2]) % hot.size[b][1]);
      const int32_t dx =
          (int32_t)(q / ((int64_t)hot.size[b][2] * hot.size[b][1]));
      const int64_t cell =
          ((int64_t)(hot.lo[b][0] + dx) * grid->ncell[1] +
           (hot.lo[b][1] + dy)) * grid->ncell[2] + (hot.lo[b][2] + dz);
      for (int t = 0; t < nthread; ++t) {
        const double *acc = private_acc +
                            ((size_t)t * (size_t)nslot +
                             (size_t)(hot.first[b] + q)) * (size_t)hot.nvalue;
        if (hydrogen_only) {
          aos[cell].acc[0] += acc[0];
          aos[cell].acc[14] += acc[1];
        } else {
          for (int k = 0; k < 16; ++k)
            aos[cell].acc[k] += acc[k];
        }
      }
    }
  }
  free(private_acc);
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
