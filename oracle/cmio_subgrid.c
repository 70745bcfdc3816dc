/*
 * cmio_subgrid.c - ORACLE (test infrastructure): the reference's TASK-BASED
 * transport semantics, restated for a serial walk of one packet at a time.
 *
 * The reference's task-based path (src/TaskBasedIonizationSimulation.cpp)
 * cuts the grid into subgrids (DensitySubGridCreator,
 * src/DensitySubGridCreator.hpp:94-98,314-396: the numbers of subgrids must
 * divide the numbers of cells) and moves packets between them in buffers.
 * What a packet deposits does not depend on the scheduling, only on
 * DensitySubGrid::interact (src/DensitySubGrid.hpp:1137-1274) and on how a
 * packet is handed from one subgrid to the next (TravelDirections,
 * src/TravelDirections.hpp); this file follows exactly those:
 *
 *  - positions are RELATIVE to the subgrid's anchor inside interact() (:1154),
 *    and snapped onto the face / edge / corner the packet enters through
 *    (update_photon_position, :248-352);
 *  - the first cell comes from the entry direction (get_start_index,
 *    :635-685): index 0 or n - 1 along the axes of the entry face, computed
 *    from the position along the others;
 *  - cell walls are index * cell_size and (index + 1.) * cell_size (:1181-1186),
 *    the optical depth is summed UP to the target (:1225-1231) and the last
 *    path length shortened by lmin *= 1 - (tau_done - tau_target) / tau;
 *  - after a step the position is set ON the wall that was hit (:1245-1253);
 *  - the packet's cross sections are pre-multiplied by the abundance of their
 *    element, hydrogen excepted (SourceDiscretePhotonTaskContext,
 *    src/SourceDiscretePhotonTaskContext.hpp:172-180), so the mean intensity
 *    of an ion of element E is A_E times the classic path's
 *    (update_intensity_counters, src/DensitySubGrid.hpp:589-617);
 *  - the heating terms use 3.288e15 Hz and 5.948e15 Hz as thresholds (:607,
 *    :611) where the classic path uses 13.6 eV and 24.6 eV;
 *  - path lengths are tallied in cells without gas too (the classic path
 *    skips them, src/DensityGrid.hpp:159).
 *
 * Emission and re-emission draw from the packet's Philox stream in the order
 * of the CLASSIC path (cmio_emit_stream, cmio_reemit_stream), as the engine
 * does in every mode, so that the packets here are the packets of cmio_shoot
 * and of the engine's decomposed mode: the three can be compared packet sum by
 * packet sum. (The reference's own task-based emission draws direction, tau,
 * frequency in another order, from per-thread generators - not reproducible
 * by construction.)
 *
 * Parity: UNPINNED against the reference's own numbers - its
 * test/testDensitySubGrid.cpp holds no known answers (it runs interact() and
 * checks a restart round trip). Pinned instead against cmio_shoot, which is
 * (tests/test_oracle_subgrid.py): both paths of the reference compute the
 * same physics, so their tallies agree up to the rounding of the different
 * arithmetic, and exactly through the two relations above.
 */
#include "cmio_internal.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* element of each ion, src/ElementNames.hpp get_element() */
static const int ion_element[CMIO_NION] = {
    CMIO_EL_H, CMIO_EL_He, CMIO_EL_C,  CMIO_EL_C,  CMIO_EL_N,
    CMIO_EL_N, CMIO_EL_N,  CMIO_EL_O,  CMIO_EL_O,  CMIO_EL_Ne,
    CMIO_EL_Ne, CMIO_EL_S, CMIO_EL_S,  CMIO_EL_S};

typedef struct {
  const cmio_grid *grid;
  int32_t nsub[3];      /* subgrids per axis */
  int32_t sub_ncell[3]; /* cells per subgrid per axis */
  double cell_size[3];
} subgrid_layout;

/*
 * DensitySubGrid::interact, src/DensitySubGrid.hpp:1137-1274, for subgrid
 * `sub` of the layout. `entry[a]` is the packet's entry along axis a: -1
 * through the lower face, +1 through the upper face, 0 not through a face of
 * that axis (all zero: TRAVELDIRECTION_INSIDE). `sigma` are the packet's
 * pre-multiplied cross sections. On return `out[a]` is the exit along each
 * axis (all zero = target optical depth reached inside, *cell = the cell
 * where) and *tau_target what is left of the optical depth.
 */
static void subgrid_interact(const subgrid_layout *L, const int32_t sub[3],
                             cmio_cells *cells, cmio_photon *photon,
                             const double sigma[CMIO_NION],
                             const int entry[3], double *tau_target_io,
                             int out[3], int64_t *cell, uint64_t *nsteps) {
  const cmio_grid *g = L->grid;
  double anchor[3], position[3];
  int32_t three_index[3];
  const double *direction = photon->direction;
  const double *inverse_direction = photon->inverse_direction;
  for (int a = 0; a < 3; ++a) {
    /* DensitySubGridCreator::create_subgrid: anchor of the subgrid
     * (src/DensitySubGridCreator.hpp:330-340) */
    const double subgrid_side = g->sides[a] / L->nsub[a];
    anchor[a] = g->anchor[a] + sub[a] * subgrid_side;
    position[a] = photon->position[a] - anchor[a];
    /* update_photon_position (:248-352) and get_start_index (:635-685) */
    if (entry[a] < 0) {
      position[a] = 0.;
      three_index[a] = 0;
    } else if (entry[a] > 0) {
      position[a] = L->sub_ncell[a] * L->cell_size[a];
      three_index[a] = L->sub_ncell[a] - 1;
    } else {
      three_index[a] = (int32_t)(position[a] * (1. / L->cell_size[a]));
    }
  }
  double tau_done = 0.;
  const double tau_target = *tau_target_io;
  int64_t active = -1;
  for (;;) {
    int inside = 1;
    for (int a = 0; a < 3; ++a)
      inside &= three_index[a] >= 0 && three_index[a] < L->sub_ncell[a];
    if (!(tau_done < tau_target) || !inside)
      break;
    const int64_t gx = (int64_t)sub[0] * L->sub_ncell[0] + three_index[0];
    const int64_t gy = (int64_t)sub[1] * L->sub_ncell[1] + three_index[1];
    const int64_t gz = (int64_t)sub[2] * L->sub_ncell[2] + three_index[2];
    active = (gx * g->ncell[1] + gy) * g->ncell[2] + gz;
    double cell_low[3], cell_high[3], l[3];
    for (int a = 0; a < 3; ++a) {
      cell_low[a] = three_index[a] * L->cell_size[a];
      cell_high[a] = (three_index[a] + 1.) * L->cell_size[a];
      if (direction[a] > 0.) {
        l[a] = (cell_high[a] - position[a]) * inverse_direction[a];
      } else if (direction[a] < 0.) {
        l[a] = (cell_low[a] - position[a]) * inverse_direction[a];
      } else {
        l[a] = DBL_MAX;
      }
    }
    double lmin = fmin(l[0], fmin(l[1], l[2]));
    /* get_optical_depth, :556-579 (HAS_HELIUM, fixed abundances) */
    const double tau =
        lmin * cells->number_density[active] *
        (sigma[CMIO_ION_H_n] * cells->ionic_fraction[CMIO_ION_H_n][active] +
         sigma[CMIO_ION_He_n] * cells->ionic_fraction[CMIO_ION_He_n][active]);
    tau_done += tau;
    if (tau_done >= tau_target) {
      const double correction = (tau_done - tau_target) / tau;
      lmin *= (1. - correction);
    } else {
      for (int a = 0; a < 3; ++a) {
        if (l[a] == lmin) {
          three_index[a] += (direction[a] > 0.) ? 1 : -1;
        }
      }
    }
    /* update_intensity_counters, :589-617 */
    {
      double dmean_intensity[CMIO_NION];
      for (int ion = 0; ion < CMIO_NION; ++ion) {
        dmean_intensity[ion] = lmin * sigma[ion] * photon->weight;
#pragma omp atomic
        cells->mean_intensity[ion][active] += dmean_intensity[ion];
      }
      const double dhH =
          dmean_intensity[CMIO_ION_H_n] * (photon->energy - 3.288e15);
      const double dhHe =
          dmean_intensity[CMIO_ION_He_n] * (photon->energy - 5.948e15);
#pragma omp atomic
      cells->heating[0][active] += dhH;
#pragma omp atomic
      cells->heating[1][active] += dhHe;
    }
    /* :1245-1253 (l == lmin is tested AFTER lmin was shortened, as there:
     * when the target is reached no wall is hit unless the correction was
     * exactly zero) */
    for (int a = 0; a < 3; ++a) {
      position[a] = (l[a] == lmin)
                        ? ((direction[a] > 0.) ? cell_high[a] : cell_low[a])
                        : position[a] + lmin * direction[a];
    }
    ++*nsteps;
  }
  *tau_target_io = tau_target - tau_done;
  for (int a = 0; a < 3; ++a)
    photon->position[a] = position[a] + anchor[a];
  if (tau_done >= tau_target) {
    out[0] = out[1] = out[2] = 0; /* TRAVELDIRECTION_INSIDE */
    *cell = active;
  } else {
    /* get_output_direction, :699-790 */
    for (int a = 0; a < 3; ++a)
      out[a] = three_index[a] < 0 ? -1
                                  : (three_index[a] >= L->sub_ncell[a] ? 1 : 0);
    *cell = -1;
  }
}

/* One flight through the subgrids: interact() in the subgrid the packet is in,
 * then into the neighbour on the other side of the face / edge / corner it
 * left through (TravelDirections::output_to_input_direction,
 * src/TravelDirections.hpp), until the optical depth is used up (returns the
 * cell) or the packet leaves the box (returns -1; periodic axes wrap,
 * src/DensitySubGridCreator.hpp:373-394). */
static int64_t subgrid_flight(const subgrid_layout *L, cmio_cells *cells,
                              const cmio_model *model, cmio_photon *photon,
                              double tau, uint64_t *nsteps,
                              uint64_t *handovers) {
  const cmio_grid *g = L->grid;
  double sigma[CMIO_NION];
  for (int ion = 0; ion < CMIO_NION; ++ion) {
    sigma[ion] = photon->cross_section[ion];
    if (ion != CMIO_ION_H_n)
      sigma[ion] *= model->abundance[ion_element[ion]];
  }
  /* DensitySubGridCreator::get_subgrid(position), :265-290 */
  int32_t sub[3];
  int entry[3] = {0, 0, 0};
  for (int a = 0; a < 3; ++a) {
    sub[a] = (int32_t)floor((photon->position[a] - g->anchor[a]) /
                            g->sides[a] * L->nsub[a]);
    if (sub[a] < 0 || sub[a] >= L->nsub[a]) {
      if (!g->periodic[a])
        return -1; /* starts outside the box */
      sub[a] = ((sub[a] % L->nsub[a]) + L->nsub[a]) % L->nsub[a];
    }
  }
  for (;;) {
    int out[3];
    int64_t cell;
    subgrid_interact(L, sub, cells, photon, sigma, entry, &tau, out, &cell,
                     nsteps);
    if ((out[0] | out[1] | out[2]) == 0)
      return cell;
    for (int a = 0; a < 3; ++a) {
      sub[a] += out[a];
      entry[a] = -out[a];
      if (sub[a] < 0 || sub[a] >= L->nsub[a]) {
        if (!g->periodic[a])
          return -1;
        /* the neighbour across the periodic face; the position moves with it
         * (the subgrid's interact() snaps it onto the entry face anyway) */
        photon->position[a] += (sub[a] < 0 ? 1. : -1.) * g->sides[a];
        sub[a] = sub[a] < 0 ? L->nsub[a] - 1 : 0;
      }
    }
    ++*handovers;
  }
}

void cmio_subgrid_shoot(const cmio_grid *grid, const int32_t nsub[3],
                        const cmio_model *model, cmio_cells *cells,
                        uint32_t seed, uint32_t iteration,
                        uint64_t first_packet, uint64_t n_packets,
                        double *totweight, double typecount[CMIO_NTYPE],
                        uint64_t *nsteps_out, uint64_t *handovers_out) {
  subgrid_layout L;
  L.grid = grid;
  for (int a = 0; a < 3; ++a) {
    if (nsub[a] < 1 || grid->ncell[a] % nsub[a] != 0) {
      /* src/DensitySubGridCreator.hpp:94-98 */
      cmio_set_error("Number of subgrids not compatible with number of "
                     "cells!");
      return;
    }
    L.nsub[a] = nsub[a];
    L.sub_ncell[a] = grid->ncell[a] / nsub[a];
    /* DensitySubGrid ctor: box side of the subgrid / its number of cells */
    L.cell_size[a] = (grid->sides[a] / nsub[a]) / L.sub_ncell[a];
  }
  double tw = 0., tc0 = 0., tc1 = 0., tc2 = 0., tc3 = 0.;
  uint64_t nsteps = 0, handovers = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : tw, tc0, tc1, tc2, tc3, nsteps, handovers)
  for (uint64_t i = 0; i < n_packets; ++i) {
    cmio_rng rng = {seed, iteration, first_packet + i, 0, NULL};
    cmio_photon photon;
    double tau;
    cmio_emit_stream(model, &rng, &photon, &tau);
    int64_t cell =
        subgrid_flight(&L, cells, model, &photon, tau, &nsteps, &handovers);
    /* PhotonReemitTaskContext, src/PhotonReemitTaskContext.hpp:100-212: the
     * packet goes on from where it was absorbed, in the same subgrid */
    while (cell >= 0 && cmio_reemit_stream(model, cells, cell, &photon, &rng)) {
      tau = -log(cmio_rng_next(&rng));
      cell =
          subgrid_flight(&L, cells, model, &photon, tau, &nsteps, &handovers);
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
  if (nsteps_out)
    *nsteps_out += nsteps;
  if (handovers_out)
    *handovers_out += handovers;
}
