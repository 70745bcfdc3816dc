/*
 * kernels.h - the engine's HIP kernels (gfx950, wave64).
 */
#ifndef CMI_KERNELS_H
#define CMI_KERNELS_H

#include "device_transport.h"
#include "device_reemit.h"

#define CMI_BLOCK 256
/* idle lanes of a wave are refilled with new packets once this many of them
 * are waiting (or when the whole wave is idle) */
#define CMI_REFILL_THRESHOLD 16

/* hardware fp64 atomic add (global_atomic_add_f64), no CAS loop */
__device__ __forceinline__ void atomic_add_f64(double *address, double value) {
  unsafeAtomicAdd(address, value);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  return v;
}

struct ShootArgs {
  GridDev grid;
  ModelDev model;
  CellsDev cells;
  CountersDev *counters;
  uint64_t first_packet;
  uint64_t n_packets;
  uint32_t seed;
  uint32_t iteration;
};

/* update_integrals, src/DensityGrid.hpp:150-197: every crossed non-vacuum
 * cell receives ds * w * sigma_ion for each ion, and the two heating terms. */
template <bool FULL, bool HEAT>
__device__ __forceinline__ void update_integrals(const ShootArgs &a,
                                                 const Packet<FULL> &p,
                                                 int64_t cell, double ds) {
  const double dsw = ds * p.weight;
  if (FULL) {
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      atomic_add_f64(a.cells.acc[i] + cell, dsw * p.sigma[FULL ? i : 0]);
    if (HEAT) {
      atomic_add_f64(a.cells.acc[CMI_NION] + cell,
                     dsw * p.sigma[ION_H_n] * (p.nu - a.model.nu_H));
      atomic_add_f64(a.cells.acc[CMI_NION + 1] + cell,
                     dsw * p.sigma[FULL ? ION_He_n : 0] *
                         (p.nu - a.model.nu_He));
    }
  } else {
    atomic_add_f64(a.cells.acc[ION_H_n] + cell, dsw * p.sigma_H);
    if (HEAT)
      atomic_add_f64(a.cells.acc[CMI_NION] + cell,
                     dsw * p.sigma_H * (p.nu - a.model.nu_H));
  }
}

/*
 * Transport kernel: IonizationPhotonShootJob::execute
 * (src/IonizationPhotonShootJob.hpp:117-146) for a range of packets.
 *
 * One lane carries one packet at a time. Lanes are persistent: lane g takes
 * packets first + g, first + g + S, ... (S = lanes in the grid), so the
 * packet -> lane map is fixed and the result does not depend on scheduling
 * (up to the summation order of the atomics). A wave refills its idle lanes
 * only when enough of them wait, which keeps the (long, divergent) emission
 * code from running for a lane or two at a time.
 */
template <bool FULL, bool HEAT, bool REEMIT>
__global__ void __launch_bounds__(CMI_BLOCK)
    shoot_kernel(const ShootArgs a) {
  const uint64_t lanes = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t end = a.first_packet + a.n_packets;
  uint64_t next_packet = a.first_packet + gid;

  Packet<FULL> p;
  PacketRng rng;
  bool active = false;
  int64_t last_cell = -1;

  double tw = 0., tc0 = 0., tc1 = 0., tc2 = 0., tc3 = 0.;
  unsigned long long nsteps = 0;

  for (;;) {
    const bool waiting = !active && next_packet < end;
    const unsigned long long waiting_mask = __ballot(waiting);
    const unsigned long long active_mask = __ballot(active);
    if (waiting_mask == 0ull && active_mask == 0ull)
      break;
    if (waiting_mask != 0ull &&
        (active_mask == 0ull ||
         __popcll(waiting_mask) >= CMI_REFILL_THRESHOLD)) {
      if (waiting) {
        rng.init(a.seed, a.iteration, next_packet);
        emit_packet(a.grid, a.model, rng, p);
        next_packet += lanes;
        active = true;
        last_cell = -1;
      }
    }
    if (active) {
      bool absorbed = false, done = false;
      if (is_inside(a.grid, p)) {
        if (p.tau > 0.) {
          double2 kappa;
          const double ds =
              dda_step(a.grid, a.cells.opacity, p, last_cell, kappa);
          ++nsteps;
          if (kappa.x >= 0.) /* number density > 0 */
            update_integrals<FULL, HEAT>(a, p, last_cell, ds);
          /* tau < 0: absorbed inside last_cell (the index was not advanced,
           * so the packet is still inside the box) */
          absorbed = (p.tau < 0.);
        } else {
          /* tau hit 0 exactly on a wall, packet still inside:
           * interact() returns the last traversed cell */
          absorbed = (last_cell >= 0);
          done = !absorbed;
        }
      } else {
        done = true; /* left the box: DensityGrid::end() */
      }
      if (absorbed) {
        /* PhotonSource::reemit, src/PhotonSource.cpp:272-308 */
        bool again = false;
        if (REEMIT) {
          again = reemit_packet(a.grid, a.model, a.cells, last_cell, rng, p);
        } else {
          p.type = TYPE_ABSORBED;
        }
        last_cell = -1;
        done = !again;
      }
      if (done) {
        tw += p.weight;
        tc0 += (p.type == TYPE_PRIMARY) ? p.weight : 0.;
        tc1 += (p.type == TYPE_DIFFUSE_HI) ? p.weight : 0.;
        tc2 += (p.type == TYPE_DIFFUSE_HeI) ? p.weight : 0.;
        tc3 += (p.type == TYPE_ABSORBED) ? p.weight : 0.;
        active = false;
      }
    }
  }

  /* IonizationPhotonShootJobMarket::update_counters */
  tw = wave_sum(tw);
  tc0 = wave_sum(tc0);
  tc1 = wave_sum(tc1);
  tc2 = wave_sum(tc2);
  tc3 = wave_sum(tc3);
  double ns = wave_sum((double)nsteps);
  if ((threadIdx.x & 63) == 0) {
    atomic_add_f64(&a.counters->totweight, tw);
    atomic_add_f64(&a.counters->typecount[0], tc0);
    atomic_add_f64(&a.counters->typecount[1], tc1);
    atomic_add_f64(&a.counters->typecount[2], tc2);
    atomic_add_f64(&a.counters->typecount[3], tc3);
    atomicAdd(&a.counters->nsteps, (unsigned long long)ns);
  }
}

/* ------------------------------------------------------- cell update -- */

struct UpdateArgs {
  GridDev grid;
  ModelDev model;
  CellsDev cells;
  double jfac; /* L / totweight / V_cell */
  double hfac;
};

/* IonizationStateCalculator::calculate_ionization_state over the grid
 * (src/IonizationStateCalculator.cpp:511-530 -> :70-272), one cell per lane,
 * grid-stride; also rebuilds the transport record of each cell. */
template <bool FULL>
__global__ void __launch_bounds__(CMI_BLOCK)
    ionization_kernel(const UpdateArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       c < a.grid.ncell_total; c += stride) {
    const double ntot = a.cells.number_density[c];
    const double T = a.cells.temperature[c];
    double J[CMI_NION], heating[2], x[CMI_NION];
    if (FULL) {
#pragma unroll
      for (int i = 0; i < CMI_NION; ++i)
        J[i] = a.cells.acc[i][c];
    } else {
      J[0] = a.cells.acc[0][c];
#pragma unroll
      for (int i = 1; i < CMI_NION; ++i)
        J[i] = 0.;
    }
    heating[0] = a.cells.acc[CMI_NION][c];
    heating[1] = a.cells.acc[CMI_NION + 1][c];
    cmi_ionization_state_cell(a.model, a.jfac, a.hfac, ntot, T, J, heating, x);
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      a.cells.x[i][c] = x[i];
    a.cells.acc[CMI_NION][c] = heating[0];
    a.cells.acc[CMI_NION + 1][c] = heating[1];
    a.cells.opacity[c] = (ntot > 0.)
                             ? make_double2(ntot * x[ION_H_n], ntot * x[ION_He_n])
                             : make_double2(-1., 0.);
  }
}

/* build the transport records from n, x_H, x_He (after an upload) */
__global__ void __launch_bounds__(CMI_BLOCK)
    opacity_kernel(const CellsDev cells, int64_t ncell) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncell;
       c += stride) {
    const double ntot = cells.number_density[c];
    cells.opacity[c] =
        (ntot > 0.) ? make_double2(ntot * cells.x[ION_H_n][c],
                                   ntot * cells.x[ION_He_n][c])
                    : make_double2(-1., 0.);
  }
}

/* ----------------------------------------------------- parity probes -- */

__global__ void emit_probe_kernel(const GridDev grid, const ModelDev model,
                                  uint32_t seed, uint32_t iteration,
                                  uint64_t first, uint64_t n, double *position,
                                  double *direction, double *frequency,
                                  double *sigma, double *tau) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  PacketRng rng;
  rng.init(seed, iteration, first + i);
  Packet<true> p;
  emit_packet(grid, model, rng, p);
  for (int a = 0; a < 3; ++a) {
    position[3 * i + a] = p.pos[a];
    direction[3 * i + a] = p.dir[a];
  }
  frequency[i] = p.nu;
  for (int k = 0; k < CMI_NION; ++k)
    sigma[CMI_NION * i + k] = p.sigma[k];
  tau[i] = p.tau;
}

__global__ void trace_probe_kernel(const GridDev grid, const double2 *opacity,
                                   uint64_t n, const double *position,
                                   const double *direction, const double *tau,
                                   const double *sigma_H,
                                   const double *sigma_He_corr,
                                   int32_t max_steps, int64_t *out_cell,
                                   double *out_ds, int32_t *out_nsteps,
                                   int64_t *out_last_cell,
                                   double *out_position) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  Packet<false> p;
  for (int a = 0; a < 3; ++a) {
    p.pos[a] = position[3 * i + a];
    p.dir[a] = direction[3 * i + a];
    p.inv_dir[a] = 1. / p.dir[a];
  }
  p.tau = tau[i];
  p.sigma_H = sigma_H[i];
  p.sigma_He_corr = sigma_He_corr[i];
  p.weight = 1.;
  locate_cell(grid, p);
  int32_t steps = 0;
  int64_t last = -1;
  while (is_inside(grid, p) && p.tau > 0.) {
    int64_t cell;
    double2 kappa;
    const double ds = dda_step(grid, opacity, p, cell, kappa);
    last = cell;
    if (steps < max_steps) {
      out_cell[(uint64_t)max_steps * i + steps] = cell;
      out_ds[(uint64_t)max_steps * i + steps] = ds;
    }
    ++steps;
  }
  if (!is_inside(grid, p))
    last = -1;
  out_nsteps[i] = steps;
  out_last_cell[i] = last;
  for (int a = 0; a < 3; ++a)
    out_position[3 * i + a] = p.pos[a];
}

#endif
