/*
 * cmio_transport.c - ORACLE (test infrastructure): photon packet emission and
 * DDA transport through the regular Cartesian grid.
 *
 * Restates, in plain C over SoA arrays:
 *   src/IonizationPhotonShootJob.hpp:117-146   (cmio_shoot)
 *   src/PhotonSource.cpp:189-199,208-249,272-308 (emit / reemit)
 *   src/PhotonSource.hpp:140-148                (random direction)
 *   src/CartesianDensityGrid.cpp:152-161,170-176,187-227,280-318,375-452
 *   src/DensityGrid.hpp:117-140,150-197         (optical depth, integrals)
 */
#include "cmio_internal.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* src/DensityGrid.hpp:219-222: ionization thresholds in Hz */
static double ionization_energy_H(void) { return cmio_eV_to_Hz(13.6); }
static double ionization_energy_He(void) { return cmio_eV_to_Hz(24.6); }

/* src/PhotonSource.cpp:189-199 */
static void set_cross_sections(const cmio_model *model, cmio_photon *photon,
                               double energy) {
  for (int ion = 0; ion < CMIO_NION; ++ion) {
    photon->cross_section[ion] = cmio_cross_section(model, ion, energy);
  }
  photon->cross_section_He_corr =
      model->abundance[CMIO_EL_He] * photon->cross_section[CMIO_ION_He_n];
}

/* src/PhotonSource.hpp:140-148 + src/Photon.hpp:165-170 */
static void set_random_direction(cmio_photon *photon, cmio_rng *rng) {
  const double cost = 2. * cmio_rng_next(rng) - 1.;
  const double sint = sqrt(fmax(1. - cost * cost, 0.));
  const double phi = 2. * M_PI * cmio_rng_next(rng);
  const double cosp = cos(phi);
  const double sinp = sin(phi);
  photon->direction[0] = sint * cosp;
  photon->direction[1] = sint * sinp;
  photon->direction[2] = cost;
  for (int a = 0; a < 3; ++a) {
    photon->inverse_direction[a] = 1. / photon->direction[a];
  }
}

/* PhotonSource ctor, src/PhotonSource.cpp:104-130 */
void cmio_mix_sources(cmio_model *model) {
  const double discrete_luminosity =
      model->nsource > 0 ? model->discrete_luminosity : 0.;
  const double continuous_luminosity =
      model->continuous_type != 0 ? model->continuous_luminosity : 0.;
  model->total_luminosity = discrete_luminosity + continuous_luminosity;
  model->continuous_probability = 0.;
  model->discrete_photon_weight = 1.;
  model->continuous_photon_weight = 1.;
  if (model->total_luminosity > 0.) {
    if (discrete_luminosity > 0.) {
      if (continuous_luminosity > 0.) {
        model->continuous_probability = 0.5;
      } else {
        model->continuous_probability = 0.;
      }
      model->discrete_photon_weight = 1.;
      if (continuous_luminosity > 0.) {
        model->continuous_photon_weight =
            (1. - model->continuous_probability) * continuous_luminosity /
            model->continuous_probability / discrete_luminosity;
      }
    } else {
      model->continuous_probability = 1.;
      model->discrete_photon_weight = 0.;
      model->continuous_photon_weight = 1.;
    }
  }
}

/* IsotropicContinuousPhotonSource::get_random_incoming_direction,
 * src/IsotropicContinuousPhotonSource.hpp:95-191 */
static void isotropic_incoming(const cmio_model *model, cmio_rng *rng,
                               cmio_photon *photon) {
  const double *anchor = model->continuous_box_anchor;
  const double *sides = model->continuous_box_sides;
  double focus[3];
  for (int a = 0; a < 3; ++a) {
    focus[a] = anchor[a] + sides[a] * cmio_rng_next(rng);
  }
  set_random_direction(photon, rng);
  const double *direction = photon->direction;
  double l[3];
  for (int a = 0; a < 3; ++a) {
    const double top = anchor[a] + sides[a];
    if (direction[a] < 0.) {
      l[a] = (top - focus[a]) / direction[a];
    } else if (direction[a] > 0.) {
      l[a] = (anchor[a] - focus[a]) / direction[a];
    } else {
      l[a] = -DBL_MAX;
    }
  }
  const double maxl = fmax(fmax(l[0], l[1]), l[2]);
  for (int a = 0; a < 3; ++a) {
    const double top = anchor[a] + sides[a];
    double x = focus[a] + maxl * direction[a];
    x = fmin(x, top - DBL_EPSILON * sides[a]);
    photon->position[a] = fmax(x, anchor[a]);
  }
}

/* PlanarContinuousPhotonSource::get_random_incoming_direction,
 * src/PlanarContinuousPhotonSource.hpp:165-188 */
static void planar_incoming(const cmio_model *model, cmio_rng *rng,
                            cmio_photon *photon) {
  const int fixed = model->continuous_axis;
  /* get_non_fixed_index, :69-76: the two other axes in their natural order */
  const int i0 = (fixed + 1) % 3, i1 = (fixed + 2) % 3;
  const int n0 = i0 < i1 ? i0 : i1, n1 = i0 < i1 ? i1 : i0;
  photon->position[n0] = model->continuous_anchor[0] +
                         cmio_rng_next(rng) * model->continuous_side[0];
  photon->position[n1] = model->continuous_anchor[1] +
                         cmio_rng_next(rng) * model->continuous_side[1];
  photon->position[fixed] = model->continuous_intercept;
  set_random_direction(photon, rng);
}

/* src/PhotonSource.cpp:208-249. The first uniform is drawn also when there is
 * no continuous source (continuous_probability = 0). */
static void random_photon(const cmio_model *model, cmio_rng *rng,
                          cmio_photon *photon) {
  double x = cmio_rng_next(rng);
  double energy;
  if (x >= model->continuous_probability) {
    x = cmio_rng_next(rng);
    int i = 0;
    while (x > model->source_cumulative[i]) {
      ++i;
    }
    for (int a = 0; a < 3; ++a) {
      photon->position[a] = model->source_position[3 * i + a];
    }
    set_random_direction(photon, rng);
    energy = cmio_spectrum_sample(model, rng);
    /* (models filled in by hand leave the mix at zero: weight 1) */
    photon->weight = model->continuous_type != 0
                         ? model->discrete_photon_weight
                         : 1.;
  } else {
    if (model->continuous_type == 2) {
      planar_incoming(model, rng, photon);
    } else {
      isotropic_incoming(model, rng, photon);
    }
    energy = cmio_continuous_spectrum_sample(model, rng);
    photon->weight = model->continuous_photon_weight;
  }
  photon->energy = energy;
  photon->type = CMIO_TYPE_PRIMARY;
  set_cross_sections(model, photon, energy);
}

void cmio_emit_stream(const cmio_model *model, cmio_rng *rng,
                      cmio_photon *photon, double *tau) {
  random_photon(model, rng, photon);
  *tau = -log(cmio_rng_next(rng));
}

void cmio_emit(const cmio_model *model, uint32_t seed, uint32_t iteration,
               uint64_t packet, cmio_photon *photon, double *tau,
               uint32_t *draws) {
  cmio_rng rng = {seed, iteration, packet, 0, NULL};
  random_photon(model, &rng, photon);
  *tau = -log(cmio_rng_next(&rng));
  if (draws)
    *draws = rng.draw;
}

void cmio_wall_intersection(const double origin[3], const double direction[3],
                            const double inverse_direction[3],
                            const double cell_anchor[3],
                            const double cell_sides[3], int32_t next_index[3],
                            double *ds, double intersection[3]) {
  double d[3];
  for (int a = 0; a < 3; ++a) {
    const double top = cell_anchor[a] + cell_sides[a];
    if (direction[a] > 0.) {
      d[a] = (top - origin[a]) * inverse_direction[a];
    } else if (direction[a] < 0.) {
      d[a] = (cell_anchor[a] - origin[a]) * inverse_direction[a];
    } else {
      d[a] = DBL_MAX;
    }
  }
  const double dmin = fmin(d[0], fmin(d[1], d[2]));
  for (int a = 0; a < 3; ++a) {
    /* every axis that ties the minimum advances (edge / corner crossings) */
    next_index[a] = (d[a] == dmin) ? ((direction[a] > 0.) ? 1 : -1) : 0;
    intersection[a] = origin[a] + dmin * direction[a];
  }
  *ds = dmin;
}

/* src/CartesianDensityGrid.cpp:187-227 */
static int is_inside(const cmio_grid *grid, int32_t index[3],
                     double position[3]) {
  int inside = 1;
  for (int a = 0; a < 3; ++a) {
    if (!grid->periodic[a]) {
      inside &= (index[a] >= 0 && index[a] < grid->ncell[a]);
    } else {
      if (index[a] < 0) {
        index[a] = grid->ncell[a] - 1;
        position[a] += grid->sides[a];
      }
      if (index[a] >= grid->ncell[a]) {
        index[a] = 0;
        position[a] -= grid->sides[a];
      }
    }
  }
  return inside;
}

/* SpectrumTrackers (src/SpectrumTracker.hpp:41-262) on cells of the grid:
 * DensityGrid::update_integrals counts every packet that crosses a cell with
 * a tracker (src/DensityGrid.hpp:188-191). One set per process, installed by
 * the test (cmio_set_trackers); counts[(k * 3 + type) * nbins + bin]. */
static struct {
  int32_t n, nbins;
  const int64_t *cell;
  const double *cos_opening_angle; /* [n] */
  const double *direction;         /* [n][3], normalised or null vector */
  uint64_t *counts;
  /* AbsorptionTrackers among them (src/AbsorptionTracker.hpp:49-235): kind[k]
   * != 0, sums in absorption[(k * 4 + type) * CMIO_NION + ion] */
  const int32_t *kind;
  double *absorption;
  /* a number of bins per tracker (NULL: nbins for all): tracker k's counts
   * then start at 3 x (the bins of the trackers before it) */
  const int32_t *bins;
  /* WeightedSpectrumTrackers (src/WeightedSpectrumTracker.hpp:44-446): kind[k]
   * == 2, sums in flux[4 x (bins of the trackers before k) + type bins[k] +
   * bin]; bins_type[k] 0 = LinearFrequencyBins between bins_min[k] and
   * bins_max[k], 1 = LevelFrequencyBins */
  double *flux;
  const int32_t *bins_type;
  const double *bins_min, *bins_max;
} trackers = {0,    0,    NULL, NULL, NULL, NULL, NULL,
              NULL, NULL, NULL, NULL, NULL, NULL};

void cmio_set_trackers(int32_t n, int32_t nbins, const int64_t *cell,
                       const double *cos_opening_angle,
                       const double *direction, uint64_t *counts) {
  trackers.n = n;
  trackers.nbins = nbins;
  trackers.cell = cell;
  trackers.cos_opening_angle = cos_opening_angle;
  trackers.direction = direction;
  trackers.counts = counts;
  trackers.kind = NULL;
  trackers.absorption = NULL;
  trackers.bins = NULL;
  trackers.flux = NULL;
}

void cmio_set_tracker_bins(const int32_t *bins) { trackers.bins = bins; }

void cmio_set_tracker_kinds(const int32_t *kind, double *absorption) {
  trackers.kind = kind;
  trackers.absorption = absorption;
}

void cmio_set_tracker_weighted(double *flux, const int32_t *bins_type,
                               const double *bins_min, const double *bins_max) {
  trackers.flux = flux;
  trackers.bins_type = bins_type;
  trackers.bins_min = bins_min;
  trackers.bins_max = bins_max;
}

/* sqrt(|u|^2 |w|^2 - (u . w)^2) for u = b - a, w = c - a: twice the area of
 * the triangle a b c, the way src/WeightedSpectrumTracker.hpp:240-278 writes
 * it (norm2 and dot_product of src/CoordinateVector.hpp: x, y, z in order) */
static double twice_triangle(const double *a, const double *b, const double *c,
                             int guarded) {
  const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  const double w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
  const double uw = u[0] * w[0] + u[1] * w[1] + u[2] * w[2];
  const double u2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
  const double w2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double a2 = u2 * w2 - uw * uw;
  /* "make sure we don't get NaN": only the y and z pairs are guarded */
  if (guarded)
    return a2 > 0. ? sqrt(a2) : 0.;
  return sqrt(a2);
}

/* WeightedSpectrumTracker::get_projected_area,
 * src/WeightedSpectrumTracker.hpp:212-290: the corners of the unit cube
 * around m = (1/2, 1/2, 1/2) projected along the direction, p = v - ((v - m)
 * . d) d, and the hexagon they span as three pairs of triangles */
double cmio_projected_area(const double *direction) {
  double p[2][2][2][3];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 2; ++k) {
        const double v[3] = {(double)i, (double)j, (double)k};
        const double along = (v[0] - 0.5) * direction[0] +
                             (v[1] - 0.5) * direction[1] +
                             (v[2] - 0.5) * direction[2];
        for (int a = 0; a < 3; ++a)
          p[i][j][k][a] = v[a] - along * direction[a];
      }
  const double ax1 = twice_triangle(p[1][0][0], p[1][0][1], p[1][1][1], 0);
  const double ax2 = twice_triangle(p[1][0][0], p[1][1][0], p[1][1][1], 0);
  const double ay1 = twice_triangle(p[1][1][0], p[1][1][1], p[0][1][1], 1);
  const double ay2 = twice_triangle(p[1][1][0], p[0][1][1], p[0][1][0], 1);
  const double az1 = twice_triangle(p[1][0][1], p[0][0][1], p[0][1][1], 1);
  const double az2 = twice_triangle(p[1][0][1], p[0][1][1], p[1][1][1], 1);
  return 0.5 * (ax1 + ax2 + ay1 + ay2 + az1 + az2);
}

/* LevelFrequencyBins (src/LevelFrequencyBins.hpp:52-86): get_bin_number is
 * Utilities::locate (src/Utilities.hpp:726-742) over the ionization energies
 * of src/ElementData.hpp:39-105 in ascending order and 4 x hydrogen's */
static int32_t level_bin(double frequency) {
  static const double edges[CMIO_NION + 1] = {
      3.28810279e+15, 3.29284691e+15, 3.51435505e+15, 5.21432028e+15,
      5.64310422e+15, 5.89588678e+15, 5.94523574e+15, 7.15759434e+15,
      8.41222200e+15, 8.49136314e+15, 9.90492110e+15, 1.14182796e+16,
      1.14732262e+16, 1.15792700e+16, 4 * 3.28810279e+15};
  uint32_t jl = 0, ju = CMIO_NION + 1;
  while (ju - jl > 1) {
    const uint32_t jm = (ju + jl) >> 1;
    if (frequency > edges[jm])
      jl = jm;
    else
      ju = jm;
  }
  if (jl == CMIO_NION)
    --jl;
  return (int32_t)jl;
}

int32_t cmio_frequency_bin(int32_t type, int32_t nbins, double minimum,
                           double maximum, double frequency) {
  if (type == 1)
    return level_bin(frequency);
  /* LinearFrequencyBins::get_bin_number, src/LinearFrequencyBins.hpp:115-125,
   * with the constructor's inverse width, :66-67 */
  if (frequency < minimum)
    return 0;
  if (frequency >= maximum)
    return nbins - 1;
  return (int32_t)((frequency - minimum) * (nbins / (maximum - minimum)));
}

/* SpectrumTracker::count_photon, src/SpectrumTracker.hpp:176-212, and
 * AbsorptionTracker::count_photon, src/AbsorptionTracker.hpp:133-138, with the
 * dmean_intensity of DensitySubGrid::update_intensity_counters
 * (src/DensitySubGrid.hpp:596-603): distance x cross section x weight */
static void count_photon(int64_t cell, const cmio_photon *photon, double ds) {
  const double minimum_frequency = 3.289e15;
  size_t first = 0; /* bins of the trackers before k */
  for (int32_t k = 0; k < trackers.n; ++k) {
    const int32_t nbins = trackers.bins ? trackers.bins[k] : trackers.nbins;
    const size_t base = 3 * first;
    first += (size_t)nbins;
    const double inverse_frequency_width = 1. / (3. * 3.289e15 / nbins);
    if (trackers.cell[k] != cell)
      continue;
    if (trackers.kind && trackers.kind[k] == 2) {
      /* WeightedSpectrumTracker::count_photon, :300-319 */
      if (photon->type >= 0 && photon->type < 4) {
        const int32_t index = cmio_frequency_bin(
            trackers.bins_type[k], nbins, trackers.bins_min[k],
            trackers.bins_max[k], photon->energy);
        const double inverse_weight =
            1. / cmio_projected_area(photon->direction);
        double *bin = trackers.flux + 4 * (base / 3) +
                      (size_t)photon->type * (size_t)nbins + index;
#pragma omp atomic
        *bin += inverse_weight;
      }
      continue;
    }
    if (trackers.kind && trackers.kind[k] != 0) {
      if (photon->type >= 0 && photon->type < 4) {
        double *bins = trackers.absorption +
                       ((size_t)k * 4 + (size_t)photon->type) * CMIO_NION;
        for (int ion = 0; ion < CMIO_NION; ++ion) {
          const double d = ds * photon->weight * photon->cross_section[ion];
#pragma omp atomic
          bins[ion] += d;
        }
      }
      continue;
    }
    const double *d = trackers.direction + 3 * k;
    if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] > 0.) {
      const double dot_product = photon->direction[0] * d[0] +
                                 photon->direction[1] * d[1] +
                                 photon->direction[2] * d[2];
      if (dot_product < trackers.cos_opening_angle[k])
        continue;
    }
    const uint32_t index = (uint32_t)((photon->energy - minimum_frequency) *
                                      inverse_frequency_width);
    if (index < (uint32_t)nbins && photon->type >= 0 && photon->type < 3) {
#pragma omp atomic
      trackers.counts[base + (size_t)photon->type * (size_t)nbins + index] += 1;
    }
  }
}

int64_t cmio_interact(const cmio_grid *grid, const cmio_model *model,
                      cmio_cells *cells, cmio_photon *photon,
                      double optical_depth, int64_t *trace_cell,
                      double *trace_ds, int64_t trace_cap, int64_t *trace_n) {
  (void)model;
  double cellside[3], inverse_cellside[3];
  for (int a = 0; a < 3; ++a) {
    cellside[a] = grid->sides[a] / grid->ncell[a];
    inverse_cellside[a] = 1. / cellside[a];
  }
  const double nuH = ionization_energy_H();
  const double nuHe = ionization_energy_He();

  double origin[3] = {photon->position[0], photon->position[1],
                      photon->position[2]};
  int32_t index[3];
  for (int a = 0; a < 3; ++a) {
    /* get_cell_indices: truncating conversion, :152-161 */
    index[a] = (int32_t)((origin[a] - grid->anchor[a]) * inverse_cellside[a]);
  }

  int64_t nstep = 0;
  int64_t last_cell = -1;
  while (is_inside(grid, index, origin) && optical_depth > 0.) {
    double cell_anchor[3];
    for (int a = 0; a < 3; ++a) {
      cell_anchor[a] = grid->anchor[a] + cellside[a] * index[a];
    }
    double ds;
    int32_t next_index[3];
    double next_wall[3];
    cmio_wall_intersection(origin, photon->direction,
                           photon->inverse_direction, cell_anchor, cellside,
                           next_index, &ds, next_wall);

    const int64_t cell = ((int64_t)index[0] * grid->ncell[1] + index[1]) *
                             grid->ncell[2] +
                         index[2];
    last_cell = cell;

    /* get_optical_depth, src/DensityGrid.hpp:117-140 (fixed abundances) */
    const double tau =
        ds * cells->number_density[cell] *
        (photon->cross_section[CMIO_ION_H_n] *
             cells->ionic_fraction[CMIO_ION_H_n][cell] +
         photon->cross_section_He_corr *
             cells->ionic_fraction[CMIO_ION_He_n][cell]);
    optical_depth -= tau;

    if (optical_depth < 0.) {
      const double Scorr = ds * optical_depth / tau;
      for (int a = 0; a < 3; ++a) {
        origin[a] += (next_wall[a] - origin[a]) * (ds + Scorr) / ds;
      }
      ds += Scorr;
    } else {
      for (int a = 0; a < 3; ++a) {
        origin[a] = next_wall[a];
        index[a] += next_index[a];
      }
    }

    /* update_integrals, src/DensityGrid.hpp:150-197 */
    if (cells->number_density[cell] > 0.) {
      const double dsw = ds * photon->weight;
      for (int ion = 0; ion < CMIO_NION; ++ion) {
        const double dj = dsw * photon->cross_section[ion];
#pragma omp atomic
        cells->mean_intensity[ion][cell] += dj;
      }
      const double dhH =
          dsw * photon->cross_section[CMIO_ION_H_n] * (photon->energy - nuH);
      const double dhHe =
          dsw * photon->cross_section[CMIO_ION_He_n] * (photon->energy - nuHe);
#pragma omp atomic
      cells->heating[0][cell] += dhH;
#pragma omp atomic
      cells->heating[1][cell] += dhHe;
      if (trackers.n != 0)
        count_photon(cell, photon, ds);
    }

    if (trace_cell && nstep < trace_cap) {
      trace_cell[nstep] = cell;
      trace_ds[nstep] = ds;
    }
    ++nstep;
  }
  if (trace_n)
    *trace_n = nstep;

  if (nstep == 0 && optical_depth > 0.) {
    /* cmac_error in the reference, :436-442 */
    cmio_set_error("cmio_interact: photon leaves the system immediately "
                   "(position: %g %g %g, direction: %g %g %g)!",
                   origin[0], origin[1], origin[2], photon->direction[0],
                   photon->direction[1], photon->direction[2]);
  }

  for (int a = 0; a < 3; ++a) {
    photon->position[a] = origin[a];
  }
  if (!is_inside(grid, index, origin)) {
    last_cell = -1;
  }
  return last_cell;
}

/* src/PhotonSource.cpp:272-308 */
static int reemit(const cmio_model *model, const cmio_cells *cells,
                  int64_t cell, cmio_photon *photon, cmio_rng *rng) {
  if (model->reemit_type == CMIO_REEMIT_NONE) {
    photon->type = CMIO_TYPE_ABSORBED;
    return 0;
  }
  int32_t type;
  const double new_frequency = cmio_reemit_frequency(
      model, photon, model->abundance[CMIO_EL_He], cells->temperature[cell],
      cells->ionic_fraction[CMIO_ION_H_n][cell],
      cells->ionic_fraction[CMIO_ION_He_n][cell], rng, &type);
  photon->type = type;
  if (new_frequency == 0.) {
    return 0;
  }
  photon->energy = new_frequency;
  set_random_direction(photon, rng);
  set_cross_sections(model, photon, new_frequency);
  return 1;
}

int cmio_reemit_stream(const cmio_model *model, const cmio_cells *cells,
                       int64_t cell, cmio_photon *photon, cmio_rng *rng) {
  return reemit(model, cells, cell, photon, rng);
}

void cmio_shoot(const cmio_grid *grid, const cmio_model *model,
                cmio_cells *cells, uint32_t seed, uint32_t iteration,
                uint64_t first_packet, uint64_t n_packets, double *totweight,
                double typecount[CMIO_NTYPE]) {
  double tw = 0.;
  double tc0 = 0., tc1 = 0., tc2 = 0., tc3 = 0.;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : tw, tc0, tc1, tc2, tc3)
  for (uint64_t i = 0; i < n_packets; ++i) {
    cmio_rng rng = {seed, iteration, first_packet + i, 0, NULL};
    cmio_photon photon;
    random_photon(model, &rng, &photon);
    double tau = -log(cmio_rng_next(&rng));
    int64_t cell =
        cmio_interact(grid, model, cells, &photon, tau, NULL, NULL, 0, NULL);
    while (cell >= 0 && reemit(model, cells, cell, &photon, &rng)) {
      tau = -log(cmio_rng_next(&rng));
      cell =
          cmio_interact(grid, model, cells, &photon, tau, NULL, NULL, 0, NULL);
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
}

void cmio_reset_grid(const cmio_grid *grid, cmio_cells *cells) {
  const int64_t ncell =
      (int64_t)grid->ncell[0] * grid->ncell[1] * grid->ncell[2];
  for (int ion = 0; ion < CMIO_NION; ++ion) {
    for (int64_t i = 0; i < ncell; ++i)
      cells->mean_intensity[ion][i] = 0.;
  }
  for (int h = 0; h < 2; ++h) {
    for (int64_t i = 0; i < ncell; ++i)
      cells->heating[h][i] = 0.;
  }
}
