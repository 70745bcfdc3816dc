/*
 * tile_kernels.h - tile rounds: transport of INCOHERENT flights (re-emitted
 * packets start anywhere and fly in any direction) with the accumulators of
 * one tile of the grid in LDS.
 *
 * Why: a flight of a later generation shares no cells with its wave
 * neighbours, so every DDA step of shoot_kernel costs one memory-side atomic
 * request (two for the 128-B rows of multi-ion transport), and the chip
 * executes ~23 G such requests per second whatever the schedule
 * (profiles/r01/atomic_rates.txt) - 8 adds fit one request, these carry one.
 *
 * How: the engine's grid is cut into tiles of T^3 cells. Flights wait in a
 * queue of rows (the marcher's own state, exactly as in a hand-over between
 * blocks of a decomposed grid), sorted by the tile of the cell they are
 * about to enter. A workgroup takes (a chunk of) the flights of ONE tile,
 * keeps the tile's accumulators in LDS (16^3 x 8 B = 32 KB hydrogen-only,
 * 8^3 x 16 x 8 B = 64 KB for 14 ions + 2 heating terms), marches every flight
 * until it leaves the tile, is absorbed or leaves the grid, adding with
 * ds_add_f64, and finally writes the tile back with full-line global atomics
 * (one request per 8 adds). Nothing is appended anywhere: a flight that leaves
 * the tile writes its new state and the key of the tile it enters into ITS
 * OWN slot, an absorbed packet its absorption record (the interaction kernel
 * of the same round turns the slot into the re-emitted flight, or frees it),
 * a finished packet frees its slot. (A first version appended to output
 * queues: one returning atomic per wave on one counter word - the chip does
 * ~90 of those per microsecond, 2e6 of them made a round of 3.6e7 flights
 * take 24 ms.) Rounds of {sort the slots by key, plan, tile kernel,
 * interaction kernel} follow until few flights are left; free slots sort
 * behind the flights and are squeezed out when they outnumber them.
 * Re-emission generations and tile crossings mix freely.
 *
 * The estimator is untouched: the same packets cross the same cells with the
 * same path lengths (the marcher state travels bit for bit, as between blocks
 * of a decomposed grid - src/DensitySubGrid.hpp:1137-1274 is the reference's
 * form of the same idea: a packet is marched subgrid by subgrid, and
 * re-emitted packets are re-queued on the subgrid where they were absorbed,
 * src/PhotonReemitTaskContext.hpp:107-209). Only the order of the additions
 * changes.
 */
#ifndef CMI_TILE_KERNELS_H
#define CMI_TILE_KERNELS_H

#include "device_reemit.h"
#include "device_transport.h"

/* tile side (log2) and workgroup size per transport flavour */
#define CMI_TILE_LOG2_H 4    /* hydrogen-only: 16^3 cells */
#define CMI_TILE_LOG2_FULL 3 /* 14 ions + heating: 8^3 cells */
#define CMI_TILE_THREADS_H 256
#define CMI_TILE_THREADS_FULL 512
/* flights per unit of work: a tile with more is shared by several workgroups
 * (each with its own LDS copy, all written back with atomics) */
#ifndef CMI_TILE_ITEM_FLIGHTS
#define CMI_TILE_ITEM_FLIGHTS 2048
#endif
#define CMI_TILE_PLAN_THREADS 1024

/* keys of slots that hold no flight (tiles have keys < ntiles): a slot whose
 * packet is gone sorts behind every tile; an absorbed packet waits for the
 * interaction kernel of the same round (never seen by a sort) */
#define CMI_TILE_KEY_DEAD(tiles) ((uint32_t)(tiles).ntiles)
#define CMI_TILE_KEY_ABSORBED(tiles) ((uint32_t)(tiles).ntiles + 1u)

struct TileArgs {
  GridDev grid;
  ModelDev model;
  CellsDev cells;
  CountersDev *counters;
  TileGridDev tiles;
  int32_t refill_threshold;
  /* the flights: every slot is updated IN PLACE (a flight that goes on into
   * another tile gets its new marcher state and key, an absorbed packet its
   * absorption record and CMI_TILE_KEY_ABSORBED, a finished one
   * CMI_TILE_KEY_DEAD) - no output queue, no shared counter */
  FlightRowsDev rows;
  const uint32_t *order; /* slots sorted by tile; NULL = identity */
  const TileItemDev *items;
  const unsigned int *nitems;
  unsigned int *next_item;
  ExchangeDev xout; /* decomposed grids: flights that leave the block */
};

/* tile index of tile coordinates */
__device__ __forceinline__ uint32_t tile_index(const TileGridDev &t, int32_t tx,
                                               int32_t ty, int32_t tz) {
  return (uint32_t)((tx * t.ntile[1] + ty) * t.ntile[2] + tz);
}

/* Plan of a round: the sorted keys are cut into units of work of at most
 * CMI_TILE_ITEM_FLIGHTS flights of one tile. One workgroup. */
struct TilePlanArgs {
  TileGridDev tiles;
  const uint32_t *sorted_keys;
  unsigned int nslots; /* slots sorted (flights + dead slots behind them) */
  TileItemDev *items;
  unsigned int *nitems;
  unsigned int *next_item;
  unsigned int *nlive; /* out: flights among the slots */
};

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *a,
                                                    uint32_t n, uint32_t v) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < v)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(CMI_TILE_PLAN_THREADS)
    tile_plan_kernel(const TilePlanArgs a) {
  __shared__ uint32_t partial[CMI_TILE_PLAN_THREADS];
  const uint32_t n = a.nslots;
  const uint32_t ntiles = (uint32_t)a.tiles.ntiles;
  /* thread k owns a contiguous range of tiles */
  const uint32_t per = (ntiles + CMI_TILE_PLAN_THREADS - 1) /
                       CMI_TILE_PLAN_THREADS;
  const uint32_t t0 = threadIdx.x * per;
  const uint32_t t1 = t0 + per < ntiles ? t0 + per : ntiles;
  uint32_t mine = 0;
  if (t0 < ntiles) {
    uint32_t begin = lower_bound_u32(a.sorted_keys, n, t0);
    for (uint32_t t = t0; t < t1; ++t) {
      const uint32_t end = lower_bound_u32(a.sorted_keys, n, t + 1);
      mine += (end - begin + CMI_TILE_ITEM_FLIGHTS - 1) / CMI_TILE_ITEM_FLIGHTS;
      begin = end;
    }
  }
  partial[threadIdx.x] = mine;
  __syncthreads();
  /* inclusive scan of the 1024 partial counts (Hillis-Steele in LDS) */
  for (int off = 1; off < CMI_TILE_PLAN_THREADS; off <<= 1) {
    const uint32_t v =
        threadIdx.x >= (unsigned)off ? partial[threadIdx.x - off] : 0u;
    __syncthreads();
    partial[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t at = partial[threadIdx.x] - mine;
  if (t0 < ntiles) {
    uint32_t begin = lower_bound_u32(a.sorted_keys, n, t0);
    for (uint32_t t = t0; t < t1; ++t) {
      const uint32_t end = lower_bound_u32(a.sorted_keys, n, t + 1);
      for (uint32_t b = begin; b < end; b += CMI_TILE_ITEM_FLIGHTS) {
        TileItemDev it;
        it.tile = t;
        it.begin = b;
        it.end = b + CMI_TILE_ITEM_FLIGHTS < end ? b + CMI_TILE_ITEM_FLIGHTS
                                                 : end;
        it.pad = 0;
        a.items[at++] = it;
      }
      begin = end;
    }
  }
  if (threadIdx.x == CMI_TILE_PLAN_THREADS - 1) {
    *a.nitems = partial[threadIdx.x];
    *a.next_item = 0;
    *a.nlive = lower_bound_u32(a.sorted_keys, n, ntiles);
  }
}

/* squeeze the dead slots out: slot j of `to` = slot order[j] of `from`
 * (afterwards the slots are in tile order and the order is the identity) */
struct TileCompactArgs {
  FlightRowsDev from, to;
  const uint32_t *order;
  const uint32_t *sorted_keys;
  const unsigned int *nlive;
  int32_t with_weights;
};

__global__ void __launch_bounds__(CMI_BLOCK)
    tile_compact_kernel(const TileCompactArgs a) {
  /* 8 lanes per row: 16 doubles = 8 x 16 B */
  const uint64_t n = *a.nlive;
  const uint64_t stride = ((uint64_t)gridDim.x * blockDim.x) >> 3;
  const int part = threadIdx.x & 7;
  for (uint64_t j = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
       j < n; j += stride) {
    const uint32_t src = a.order[j];
    const double2 *r = reinterpret_cast<const double2 *>(
        a.from.rows + (size_t)CMI_FLIGHT_DOUBLES * src);
    reinterpret_cast<double2 *>(a.to.rows +
                                (size_t)CMI_FLIGHT_DOUBLES * j)[part] = r[part];
    if (a.with_weights) {
      const double2 *w = reinterpret_cast<const double2 *>(
          a.from.weights + (size_t)CMI_NACC * src);
      reinterpret_cast<double2 *>(a.to.weights + (size_t)CMI_NACC * j)[part] =
          w[part];
    }
    if (part == 0)
      a.to.keys[j] = a.sorted_keys[j];
  }
}

/* a new flight into slot q (interaction kernel) */
template <bool FULL>
__device__ __forceinline__ void
write_flight_row(const FlightRowsDev &out, unsigned int q,
                 const Packet<FULL> &p, uint32_t packed_lc, uint32_t key,
                 uint32_t packet_id, uint32_t meta,
                 const double (&weights)[CMI_NACC]) {
  double *r = out.rows + (size_t)CMI_FLIGHT_DOUBLES * q;
  double4 *r4 = reinterpret_cast<double4 *>(r);
  r4[0] = make_double4(p.pos[0], p.pos[1], p.pos[2], p.dir[0]);
  r4[1] = make_double4(p.dir[1], p.dir[2], p.t, p.tmax[0]);
  r4[2] = make_double4(p.tmax[1], p.tmax[2], p.tau, p.nu);
  r4[3] = make_double4(
      __longlong_as_double((long long)p.cell),
      __longlong_as_double(
          (long long)(((unsigned long long)meta << 32) | packet_id)),
      __longlong_as_double((long long)packed_lc), 0.);
  out.keys[q] = key;
  if (FULL) {
    double4 *w = reinterpret_cast<double4 *>(out.weights + (size_t)CMI_NACC * q);
#pragma unroll
    for (int i = 0; i < CMI_NACC; i += 4)
      w[i >> 2] = make_double4(weights[i], weights[i + 1], weights[i + 2],
                               weights[i + 3]);
  }
}

/*
 * The tile kernel: PhotonTraversalTaskContext::execute
 * (src/PhotonTraversalTaskContext.hpp:100-278) for the flights of one tile,
 * with DensitySubGrid::interact (src/DensitySubGrid.hpp:1137-1274) as the
 * incremental marcher of device_transport.h and the tile's
 * update_integrals (src/DensityGrid.hpp:150-197) in LDS.
 */
template <bool FULL, bool HEAT>
__global__ void __launch_bounds__(FULL ? CMI_TILE_THREADS_FULL
                                       : CMI_TILE_THREADS_H,
                                  FULL ? 4 : (HEAT ? 2 : 4))
    tile_kernel(const TileArgs a) {
  constexpr int L = FULL ? CMI_TILE_LOG2_FULL : CMI_TILE_LOG2_H;
  constexpr int T = 1 << L;
  constexpr int TC = T * T * T;
  constexpr int NV = FULL ? CMI_NACC : (HEAT ? 2 : 1);
  constexpr int NT = FULL ? CMI_TILE_THREADS_FULL : CMI_TILE_THREADS_H;
  /* value v of tile cell k lives at acc[v * TC + k]: neighbouring cells in
   * neighbouring banks whatever the value */
  __shared__ double acc[NV * TC];
  __shared__ unsigned int s_item, s_next;

  const int lane = threadIdx.x & 63;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  const bool any_periodic =
      (a.grid.periodic[0] | a.grid.periodic[1] | a.grid.periodic[2]) != 0;
  const uint32_t key_dead = CMI_TILE_KEY_DEAD(a.tiles);
  const uint32_t key_absorbed = CMI_TILE_KEY_ABSORBED(a.tiles);

  Packet<FULL> p;
  p.sigma_H = 0.;
  p.sigma_He_corr = 0.;
  p.nu = 0.;
  p.weight = 1.;
  double weights[CMI_NACC];
  uint32_t packet_id = 0, lane_meta = 0, slot = 0;
  unsigned int tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0;
  unsigned int nsteps = 0, natomics = 0, nwavesteps = 0;

  for (;;) {
    if (threadIdx.x == 0)
      s_item = atomicAdd(a.next_item, 1u);
    for (int k = threadIdx.x; k < NV * TC; k += NT)
      acc[k] = 0.;
    __syncthreads();
    const unsigned int item = s_item;
    if (item >= *a.nitems)
      break;
    const TileItemDev it = a.items[item];
    if (threadIdx.x == 0)
      s_next = it.begin;
    __syncthreads();
    /* the tile: coordinates of its first cell */
    const int32_t tz = (int32_t)(it.tile % (uint32_t)a.tiles.ntile[2]);
    const int32_t ty = (int32_t)((it.tile / (uint32_t)a.tiles.ntile[2]) %
                                 (uint32_t)a.tiles.ntile[1]);
    const int32_t tx = (int32_t)(it.tile / ((uint32_t)a.tiles.ntile[2] *
                                            (uint32_t)a.tiles.ntile[1]));
    const int32_t o[3] = {tx << L, ty << L, tz << L};

    bool active = false;
    int32_t last_cell = -1, last_lidx = 0;
    for (;;) {
      const unsigned long long active_mask = __ballot(active);
      const unsigned long long idle_mask = ~active_mask;
      /* (wave-uniform enough: a stale "true" costs one empty refill) */
      const bool avail =
          __ballot(*(volatile unsigned int *)&s_next < it.end) != 0ull;
      if (active_mask == 0ull && !avail)
        break;
      if (avail && idle_mask != 0ull &&
          (active_mask == 0ull ||
           (int)__popcll(idle_mask) >= a.refill_threshold)) {
        unsigned int base = 0;
        if (lane == 0)
          base = atomicAdd(&s_next, (unsigned int)__popcll(idle_mask));
        base = __shfl(base, 0, 64);
        const unsigned int i = base + __popcll(idle_mask & lane_lt);
        if (!active && i < it.end) {
          slot = a.order ? a.order[i] : i;
          const double4 *r = reinterpret_cast<const double4 *>(
              a.rows.rows + (size_t)CMI_FLIGHT_DOUBLES * slot);
          const double4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
          p.pos[0] = r0.x;
          p.pos[1] = r0.y;
          p.pos[2] = r0.z;
          p.dir[0] = r0.w;
          p.dir[1] = r1.x;
          p.dir[2] = r1.y;
          p.t = r1.z;
          p.tmax[0] = r1.w;
          p.tmax[1] = r2.x;
          p.tmax[2] = r2.y;
          p.tau = r2.z;
          p.nu = r2.w;
          p.cell = (int32_t)__double_as_longlong(r3.x);
          const unsigned long long idmeta =
              (unsigned long long)__double_as_longlong(r3.y);
          packet_id = (uint32_t)idmeta;
          lane_meta = (uint32_t)(idmeta >> 32);
          const uint32_t plc = (uint32_t)__double_as_longlong(r3.z);
          p.type = (int32_t)(lane_meta >> 28);
          p.weight = 1.;
          if (FULL) {
            const double4 *w = reinterpret_cast<const double4 *>(
                a.rows.weights + (size_t)CMI_NACC * slot);
#pragma unroll
            for (int k = 0; k < CMI_NACC; k += 4) {
              const double4 w4 = w[k >> 2];
              weights[k] = w4.x;
              weights[k + 1] = w4.y;
              weights[k + 2] = w4.z;
              weights[k + 3] = w4.w;
            }
            p.sigma_H = weights[ION_H_n];
            p.sigma_He = weights[ION_He_n];
            p.sigma_He_corr = a.model.abundance[0] * p.sigma_He;
          } else {
            p.sigma_H = a.model.xsec_fixed[ION_H_n];
            p.sigma_He = a.model.xsec_fixed[ION_He_n];
            p.sigma_He_corr = a.model.abundance[0] * p.sigma_He;
            weights[ION_H_n] = p.sigma_H;
            weights[CMI_NION] = p.sigma_H * (p.nu - a.model.nu_H);
          }
          const int32_t stride[3] = {a.grid.ncell[1] * a.grid.ncell[2],
                                     a.grid.ncell[2], 1};
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            p.lc[ax] = (int32_t)((plc >> (8 * ax)) & 0xffu);
            const bool fwd = p.dir[ax] > 0.;
            /* exactly start_flight()'s expression: the increments must be
             * the same numbers in every tile the flight crosses */
            p.inv_dir[ax] = 1. / p.dir[ax];
            p.tdelta[ax] = (p.dir[ax] != 0.)
                               ? a.grid.cellside[ax] * fabs(p.inv_dir[ax])
                               : 0.;
            p.cstep[ax] = fwd ? stride[ax] : -stride[ax];
            p.lsgn[ax] = fwd ? 1 : -1;
            const int32_t g = o[ax] + p.lc[ax];
            p.rem[ax] = fwd ? a.grid.ncell[ax] - 1 - g : g;
          }
          active = true;
          last_cell = -1;
        }
      }
      const bool avail_after =
          __ballot(*(volatile unsigned int *)&s_next < it.end) != 0ull;

      /* ---- hot loop: march + LDS accumulation ---- */
      auto in_tile = [&]() {
        return (((uint32_t)(p.lc[0] | p.lc[1] | p.lc[2])) >> L) == 0u;
      };
      double2 kappa_next = make_double2(0., 0.);
      if (active && p.tau > 0. && !fast_outside(p) && in_tile())
        kappa_next = fast_load_record(a.cells.opacity, p);
      for (;;) {
        const bool stepping =
            active && p.tau > 0. && !fast_outside(p) && in_tile();
        const unsigned long long flying = __ballot(stepping);
        if (flying == 0ull ||
            (avail_after && (int)__popcll(~flying) >= a.refill_threshold))
          break;
        ++nwavesteps;
        if (stepping) {
          const double2 kappa = kappa_next;
          last_lidx = (p.lc[0] << (2 * L)) | (p.lc[1] << L) | p.lc[2];
          const double ds = fast_step<FULL, true>(p, last_cell, kappa);
          ++nsteps;
          if (any_periodic && p.tau >= 0.) {
            /* fast_wrap(), and the flight has left this tile: the last tile
             * of an axis may be clipped (fewer than T cells), so the tile
             * coordinate alone would not say so */
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
              if (a.grid.periodic[ax] && p.rem[ax] < 0) {
                if (p.cstep[ax] > 0)
                  p.lc[ax] = T;
                p.cell -= p.cstep[ax] * a.grid.ncell[ax];
                p.rem[ax] = a.grid.ncell[ax] - 1;
                p.pos[ax] -=
                    (p.cstep[ax] > 0 ? 1. : -1.) * a.grid.box_sides[ax];
              }
            }
          }
          if (p.tau > 0. && !fast_outside(p) && in_tile())
            kappa_next = fast_load_record(a.cells.opacity, p);
          if (kappa.x >= 0.) { /* number density > 0 */
            const double dsw = ds * p.weight;
            if (FULL) {
#pragma unroll
              for (int i = 0; i < CMI_NACC; ++i)
                if (HEAT || i < CMI_NION)
                  atomicAdd(&acc[i * TC + last_lidx], dsw * weights[i]);
            } else {
              atomicAdd(&acc[last_lidx], dsw * weights[ION_H_n]);
              if (HEAT)
                atomicAdd(&acc[TC + last_lidx], dsw * weights[CMI_NION]);
            }
          }
        }
      }

      /* ---- end of the tile visit for every lane that cannot step ---- */
      if (active) {
        const bool outside_grid = fast_outside(p);
        const bool stay = p.tau > 0. && !outside_grid && in_tile();
        if (!stay) {
          bool absorbed = false, done = false, moved = false;
          if (p.tau < 0.) {
            absorbed = true;
          } else if (!outside_grid && !(p.tau > 0.)) {
            /* tau hit 0 exactly on a wall, packet still inside the grid:
             * interact() returns the last traversed cell */
            absorbed = last_cell >= 0;
            done = !absorbed;
          } else if (outside_grid) {
            done = true; /* left the grid: DensityGrid::end() ... */
            if (a.grid.decomposed && last_cell >= 0) {
              /* ... or only this block of it: hand the flight over */
              const int64_t cell_global =
                  exit_cell_global(a.grid, p, last_cell);
              if (cell_global >= 0) {
                const unsigned long long leaving = __ballot(true);
                unsigned int base = 0;
                const int first = __ffsll((long long)leaving) - 1;
                if (lane == first)
                  base = atomicAdd(a.xout.count,
                                   (unsigned int)__popcll(leaving));
                base = __shfl(base, first, 64);
                const unsigned int q = base + __popcll(leaving & lane_lt);
                if (q < a.xout.capacity) {
                  double *r = a.xout.rows + (size_t)CMI_FLIGHT_DOUBLES * q;
#pragma unroll
                  for (int ax = 0; ax < 3; ++ax) {
                    r[ax] = p.pos[ax];
                    r[3 + ax] = p.dir[ax];
                    r[7 + ax] = p.tmax[ax];
                  }
                  r[6] = p.t;
                  r[10] = p.tau;
                  r[11] = p.nu;
                  r[12] = __longlong_as_double(cell_global);
                  r[13] = __longlong_as_double((long long)(
                      ((unsigned long long)lane_meta << 32) | packet_id));
                  r[14] = 0.;
                  r[15] = 0.;
                }
                done = false; /* goes on elsewhere: this slot is free */
              }
            }
          } else {
            moved = true; /* into another tile of this grid */
          }
          double *r = a.rows.rows + (size_t)CMI_FLIGHT_DOUBLES * slot;
          uint32_t key = key_dead;
          if (moved) {
            /* the marcher's state at the wall, into the same slot; the
             * coordinates of the cell being entered (a periodic axis has
             * already wrapped the long index; the coordinate follows) */
            int32_t g[3];
            uint32_t plc = 0;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
              g[ax] = o[ax] + p.lc[ax];
              if (g[ax] < 0)
                g[ax] = a.grid.ncell[ax] - 1;
              else if (g[ax] >= a.grid.ncell[ax])
                g[ax] = 0;
              plc |= (uint32_t)(g[ax] & (T - 1)) << (8 * ax);
            }
            key = tile_index(a.tiles, g[0] >> L, g[1] >> L, g[2] >> L);
            if (any_periodic) { /* a wrap shifts the origin of the flight */
              r[0] = p.pos[0];
              r[1] = p.pos[1];
              r[2] = p.pos[2];
            }
            r[6] = p.t;
            r[7] = p.tmax[0];
            r[8] = p.tmax[1];
            r[9] = p.tmax[2];
            r[10] = p.tau;
            r[12] = __longlong_as_double((long long)p.cell);
            r[14] = __longlong_as_double((long long)plc);
          } else if (absorbed) {
            /* the absorption record, for the interaction kernel of this
             * round: where, in which cell (frequency, id and random stream
             * position are in the slot already) */
            end_flight(p);
            r[0] = p.pos[0];
            r[1] = p.pos[1];
            r[2] = p.pos[2];
            r[12] = __longlong_as_double((long long)last_cell);
            key = key_absorbed;
          }
          a.rows.keys[slot] = key;
          if (done) {
            tc0 += (p.type == TYPE_PRIMARY) ? 1u : 0u;
            tc1 += (p.type == TYPE_DIFFUSE_HI) ? 1u : 0u;
            tc2 += (p.type == TYPE_DIFFUSE_HeI) ? 1u : 0u;
            tc3 += (p.type == TYPE_ABSORBED) ? 1u : 0u;
          }
          active = false;
        }
      }
    }

    /* ---- write the tile back: one full-line atomic per 8 values ---- */
    __syncthreads();
    if (FULL) {
      const int i = threadIdx.x & 15;
      if (HEAT || i < CMI_NION) {
        for (int k = threadIdx.x >> 4; k < TC; k += NT / 16) {
          const int32_t lx = k >> (2 * L), ly = (k >> L) & (T - 1),
                        lz = k & (T - 1);
          const int32_t gx = o[0] + lx, gy = o[1] + ly, gz = o[2] + lz;
          if (gx < a.grid.ncell[0] && gy < a.grid.ncell[1] &&
              gz < a.grid.ncell[2]) {
            const double v = acc[i * TC + k];
            if (v != 0.) {
              const int64_t cell =
                  ((int64_t)gx * a.grid.ncell[1] + gy) * a.grid.ncell[2] + gz;
              atomic_add_f64(acc_at(a.cells, i, cell), v);
              ++natomics;
            }
          }
        }
      }
    } else {
      for (int k = threadIdx.x; k < TC; k += NT) {
        const int32_t lx = k >> (2 * L), ly = (k >> L) & (T - 1),
                      lz = k & (T - 1);
        const int32_t gx = o[0] + lx, gy = o[1] + ly, gz = o[2] + lz;
        if (gx < a.grid.ncell[0] && gy < a.grid.ncell[1] &&
            gz < a.grid.ncell[2]) {
          const int64_t cell =
              ((int64_t)gx * a.grid.ncell[1] + gy) * a.grid.ncell[2] + gz;
          const double v = acc[k];
          if (v != 0.) {
            atomic_add_f64(acc_at(a.cells, ION_H_n, cell), v);
            ++natomics;
          }
          if (HEAT) {
            const double h = acc[TC + k];
            if (h != 0.) {
              atomic_add_f64(acc_at(a.cells, CMI_NION, cell), h);
              ++natomics;
            }
          }
        }
      }
    }
    __syncthreads();
  }

  const double s0 = wave_sum((double)tc0);
  const double s1 = wave_sum((double)tc1);
  const double s2 = wave_sum((double)tc2);
  const double s3 = wave_sum((double)tc3);
  const double ns = wave_sum((double)nsteps);
  const double na = wave_sum((double)natomics);
  if (lane == 0) {
    if ((s0 + s1) + (s2 + s3) != 0.) {
      atomic_add_f64(&a.counters->totweight, (s0 + s1) + (s2 + s3));
      atomic_add_f64(&a.counters->typecount[0], s0);
      atomic_add_f64(&a.counters->typecount[1], s1);
      atomic_add_f64(&a.counters->typecount[2], s2);
      atomic_add_f64(&a.counters->typecount[3], s3);
    }
    atomicAdd(&a.counters->nsteps, (unsigned long long)ns);
    atomicAdd(&a.counters->natomics, (unsigned long long)na);
    atomicAdd(&a.counters->nwavesteps, (unsigned long long)nwavesteps);
  }
}

#endif
