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

/* waves per SIMD the tile kernels are built for (tile sides and workgroup
 * sizes: TileShape, device_common.h) */
#ifndef CMI_TILE_WAVES
#define CMI_TILE_WAVES 4
#endif
/* flights per unit of work: a tile with more is shared by several workgroups
 * (each with its own LDS copy, all written back with atomics); measured on
 * 256^3 with the round-5 tiles (32 x 16 x 16 / 8 x 8 x 16), ms of transport
 * per diffuse / lexington iteration: 4096 / 2048 flights 74.4 / 179.3,
 * 8192 / 4096 72.8 / 174.7, 16384 / 8192 71.9 / 172.9, 32768 / 16384 72.1 /
 * 173.2 */
#ifndef CMI_TILE_ITEM_FLIGHTS_H
#define CMI_TILE_ITEM_FLIGHTS_H 16384
#endif
#ifndef CMI_TILE_ITEM_FLIGHTS_FULL
#define CMI_TILE_ITEM_FLIGHTS_FULL 8192
#endif
#define CMI_TILE_PLAN_THREADS 1024

/* key of a slot that holds no flight (tiles have keys < ntiles): sorts behind
 * every tile */
#define CMI_TILE_KEY_DEAD(tiles) ((uint32_t)(tiles).ntiles)

struct TileArgs {
  GridDev grid;
  ModelDev model;
  CellsDev cells;
  CountersDev *counters;
  TileGridDev tiles;
  int32_t refill_threshold;
  /* the flights: every slot is updated IN PLACE (a flight that goes on into
   * another tile gets its new marcher state and key, a finished one
   * CMI_TILE_KEY_DEAD; the slot of an absorbed packet is rewritten by the
   * interaction kernel of the round) - no output queue, no shared counter */
  FlightRowsDev rows;
  /* The flights of a round are known by POSITION: position i < n has the key
   * keys[i] and sits in slot slot_in[i] (NULL: slot i) of the rows. `order`
   * lists the positions by tile; the flight at place j of that list gets
   * position j of the NEXT round: its new key goes to keys_out[j], its slot
   * to slot_out[j] - dense, coalesced writes (a 4-B key written into the
   * slot's place of a key array is one more scattered partial-line write per
   * visit: 8 ps of 60, tools/microbench/row_gather.hip), the flights that
   * ended drop out of the next round's arrays by themselves (their keys sort
   * last), and the rows never move. */
  const uint32_t *order;
  const uint32_t *slot_in;
  uint32_t *keys_out;
  uint32_t *slot_out;
  const TileItemDev *items;
  const unsigned int *nitems;
  unsigned int *next_item;
  /* the packets absorbed in this round, for the interaction kernel: unit of
   * work k leaves its absorption records (position, cell, frequency, id,
   * meta - and the slot they came from) at positions [begin, begin + n_k) of
   * these arrays and n_k in absorbed_count[k] */
  QueueDev ended;
  uint32_t *ended_slot;
  uint32_t *ended_pos; /* its position in the next round (keys_out) */
  unsigned int *absorbed_count;
  ExchangeDev xout; /* decomposed grids: flights that leave the block */
};

/* tile index of tile coordinates */
__device__ __forceinline__ uint32_t tile_index(const TileGridDev &t, int32_t tx,
                                               int32_t ty, int32_t tz) {
  return (uint32_t)((tx * t.ntile[1] + ty) * t.ntile[2] + tz);
}

/* Plan of a round, from the sorted keys: first the position where every
 * tile's flights begin ... */
struct TilePlanArgs {
  TileGridDev tiles;
  const uint32_t *sorted_keys;
  unsigned int nslots; /* slots sorted (flights + dead slots behind them) */
  uint32_t *tile_begin; /* [ntiles + 2] */
  uint32_t item_flights;
  TileItemDev *items;
  unsigned int *nitems;
  unsigned int *next_item;
  unsigned int *nlive; /* out: flights among the slots */
};

__global__ void __launch_bounds__(CMI_BLOCK)
    tile_begin_kernel(const TilePlanArgs a) {
  const uint32_t n = a.nslots;
  const uint32_t last = (uint32_t)a.tiles.ntiles + 1u; /* keys 0 .. ntiles */
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n;
       i += stride) {
    /* keys in (key[i-1], key[i]] begin at i; beyond the last key, at n */
    const uint32_t lo = (i == 0) ? 0u : a.sorted_keys[i - 1] + 1u;
    const uint32_t hi = (i == n) ? last : a.sorted_keys[i];
    for (uint32_t k = lo; k <= hi && k <= last; ++k)
      a.tile_begin[k] = (uint32_t)i;
  }
}

/* The slots in tile order by counting, when a workgroup's LDS holds one
 * counter per tile (a radix sort needs two passes over 13..15 key bits and
 * carries the dead slots along; counting reads the keys twice and writes only
 * the order of the live ones): CMI_TILE_SORT_BLOCKS workgroups own one
 * contiguous chunk of the slots each;
 *   tile_count_kernel   the chunk's flights per tile -> blockhist[b][tile]
 *   tile_column_kernel  per tile, over the chunks: blockhist[b][tile] becomes
 *                       the number of that tile's flights in earlier chunks,
 *                       total[tile] their number in all
 *   tile_offsets_kernel exclusive scan of total -> tile_begin (one workgroup)
 *   tile_scatter_kernel order[tile_begin[tile] + blockhist[b][tile] + rank in
 *                       the chunk] = slot
 * The order of a tile's flights within a chunk is whatever the LDS atomics
 * make it: it only decides which unit of work a flight lands in. */
#ifndef CMI_TILE_SORT_BLOCKS
#define CMI_TILE_SORT_BLOCKS 256
#endif
#define CMI_TILE_SORT_THREADS 1024
#define CMI_TILE_SORT_MAX_TILES 32768

struct TileSortArgs {
  const uint32_t *keys;
  unsigned int nslots;
  uint32_t ntiles;
  /* workgroups that share the slots: a multiple of 8, at most
   * CMI_TILE_SORT_BLOCKS - fewer when there are few slots (every workgroup
   * costs a pass over one counter per tile in each of the three kernels) */
  uint32_t nblocks;
  uint32_t *blockhist; /* [CMI_TILE_SORT_BLOCKS][ntiles] */
  uint32_t *total;     /* [ntiles] */
  uint32_t *tile_begin; /* [ntiles + 2] */
  uint32_t *order;
};

__device__ __forceinline__ void tile_sort_chunk(const TileSortArgs &a,
                                                uint64_t &first,
                                                uint64_t &last) {
  const uint64_t chunk = ((uint64_t)a.nslots + a.nblocks - 1) / a.nblocks;
  first = (uint64_t)blockIdx.x * chunk;
  last = first + chunk < a.nslots ? first + chunk : a.nslots;
}

__global__ void __launch_bounds__(CMI_TILE_SORT_THREADS)
    tile_count_kernel(const TileSortArgs a) {
  __shared__ uint32_t count[CMI_TILE_SORT_MAX_TILES];
  for (uint32_t t = threadIdx.x; t < a.ntiles; t += CMI_TILE_SORT_THREADS)
    count[t] = 0;
  __syncthreads();
  uint64_t first, last;
  tile_sort_chunk(a, first, last);
  for (uint64_t i = first + threadIdx.x; i < last; i += CMI_TILE_SORT_THREADS) {
    const uint32_t key = a.keys[i];
    if (key < a.ntiles)
      atomicAdd(&count[key], 1u);
  }
  __syncthreads();
  uint32_t *mine = a.blockhist + (size_t)blockIdx.x * a.ntiles;
  for (uint32_t t = threadIdx.x; t < a.ntiles; t += CMI_TILE_SORT_THREADS)
    mine[t] = count[t];
}

__global__ void __launch_bounds__(CMI_BLOCK)
    tile_column_kernel(const TileSortArgs a) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.ntiles)
    return;
  uint32_t run = 0;
  uint32_t *column = a.blockhist + t;
  for (uint32_t b = 0; b < a.nblocks; b += 8) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      v[k] = column[(size_t)(b + k) * a.ntiles];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      column[(size_t)(b + k) * a.ntiles] = run;
      run += v[k];
    }
  }
  a.total[t] = run;
}

__global__ void __launch_bounds__(CMI_TILE_SORT_THREADS)
    tile_offsets_kernel(const TileSortArgs a) {
  __shared__ uint32_t partial[CMI_TILE_SORT_THREADS];
  /* the totals through LDS: coalesced loads that do not wait for one another
   * (a thread summing its 32 consecutive totals from global memory waited for
   * every one of them: 48 us per round with 32768 tiles), and the begins back
   * the same way */
  /* (one word of padding per 32: a thread's consecutive entries and its
   * neighbours' then lie in different banks - without it all 64 lanes of a
   * wave hit one bank in the loops below) */
  __shared__ uint32_t staged[CMI_TILE_SORT_MAX_TILES +
                             CMI_TILE_SORT_MAX_TILES / 32];
  auto at_lds = [](uint32_t t) { return t + (t >> 5); };
  for (uint32_t t = threadIdx.x; t < a.ntiles; t += CMI_TILE_SORT_THREADS)
    staged[at_lds(t)] = a.total[t];
  __syncthreads();
  const uint32_t per =
      (a.ntiles + CMI_TILE_SORT_THREADS - 1) / CMI_TILE_SORT_THREADS;
  const uint32_t t0 = threadIdx.x * per;
  const uint32_t t1 = t0 + per < a.ntiles ? t0 + per : a.ntiles;
  uint32_t mine = 0;
  for (uint32_t t = t0; t < t1; ++t)
    mine += staged[at_lds(t)];
  partial[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < CMI_TILE_SORT_THREADS; off <<= 1) {
    const uint32_t v =
        threadIdx.x >= (unsigned)off ? partial[threadIdx.x - off] : 0u;
    __syncthreads();
    partial[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t at = partial[threadIdx.x] - mine;
  for (uint32_t t = t0; t < t1; ++t) {
    const uint32_t here = staged[at_lds(t)];
    staged[at_lds(t)] = at;
    at += here;
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < a.ntiles; t += CMI_TILE_SORT_THREADS)
    a.tile_begin[t] = staged[at_lds(t)];
  if (threadIdx.x == CMI_TILE_SORT_THREADS - 1) {
    /* the dead slots follow the flights (tile_begin_kernel's convention) */
    a.tile_begin[a.ntiles] = partial[threadIdx.x];
    a.tile_begin[a.ntiles + 1] = a.nslots;
  }
}

__global__ void __launch_bounds__(CMI_TILE_SORT_THREADS)
    tile_scatter_kernel(const TileSortArgs a) {
  __shared__ uint32_t cursor[CMI_TILE_SORT_MAX_TILES];
  const uint32_t *mine = a.blockhist + (size_t)blockIdx.x * a.ntiles;
  for (uint32_t t = threadIdx.x; t < a.ntiles; t += CMI_TILE_SORT_THREADS)
    cursor[t] = a.tile_begin[t] + mine[t];
  __syncthreads();
  uint64_t first, last;
  tile_sort_chunk(a, first, last);
  for (uint64_t i = first + threadIdx.x; i < last; i += CMI_TILE_SORT_THREADS) {
    const uint32_t key = a.keys[i];
    if (key < a.ntiles)
      a.order[atomicAdd(&cursor[key], 1u)] = (uint32_t)i;
  }
}

/* ... then the units of work: at most item_flights flights of one tile each.
 * One workgroup. */
__global__ void __launch_bounds__(CMI_TILE_PLAN_THREADS)
    tile_plan_kernel(const TilePlanArgs a) {
  __shared__ uint32_t partial[CMI_TILE_PLAN_THREADS];
  const uint32_t ntiles = (uint32_t)a.tiles.ntiles;
  const uint32_t M = a.item_flights;
  /* (the begins through LDS where they fit: see tile_offsets_kernel) */
  __shared__ uint32_t staged[CMI_TILE_SORT_MAX_TILES +
                             CMI_TILE_SORT_MAX_TILES / 32 + 2];
  const bool in_lds = ntiles <= CMI_TILE_SORT_MAX_TILES;
  if (in_lds) {
    for (uint32_t t = threadIdx.x; t <= ntiles; t += CMI_TILE_PLAN_THREADS)
      staged[t + (t >> 5)] = a.tile_begin[t];
    __syncthreads();
  }
  auto tile_begin = [&](uint32_t t) {
    return in_lds ? staged[t + (t >> 5)] : a.tile_begin[t];
  };
  /* thread k owns a contiguous range of tiles */
  const uint32_t per = (ntiles + CMI_TILE_PLAN_THREADS - 1) /
                       CMI_TILE_PLAN_THREADS;
  const uint32_t t0 = threadIdx.x * per;
  const uint32_t t1 = t0 + per < ntiles ? t0 + per : ntiles;
  uint32_t mine = 0;
  for (uint32_t t = t0; t < t1; ++t)
    mine += (tile_begin(t + 1) - tile_begin(t) + M - 1) / M;
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
  for (uint32_t t = t0; t < t1; ++t) {
    const uint32_t begin = tile_begin(t), end = tile_begin(t + 1);
    for (uint32_t b = begin; b < end; b += M) {
      TileItemDev it;
      it.tile = t;
      it.begin = b;
      it.end = b + M < end ? b + M : end;
      it.pad = 0;
      a.items[at++] = it;
    }
  }
  if (threadIdx.x == CMI_TILE_PLAN_THREADS - 1) {
    *a.nitems = partial[threadIdx.x];
    *a.next_item = 0;
    *a.nlive = tile_begin(ntiles);
  }
}

/* The rows of the live flights into fresh rows in tile order: row j of `to` =
 * the flight at place j of the tile order (afterwards position j sits in slot
 * j). The rows of a round's flights lie scattered among the rows of all that
 * have ended since; multi-ion transport reads two of them per visit (slot and
 * weights, 25 GB at 1e8 packets), and once most slots are dead the visits pay
 * for the sparse footprint (measured without any compaction: the rounds of
 * 1e7 ... 2e7 flights 2.0 ... 2.7 ms instead of 1.5 ... 2.2). */
struct TileCompactArgs {
  FlightRowsDev from, to;
  const uint32_t *order;   /* positions by tile */
  const uint32_t *slot_in; /* position -> slot of `from` (NULL: identity) */
  const uint32_t *keys_in; /* by position */
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
    const uint32_t i = a.order[j];
    const uint32_t src = a.slot_in ? a.slot_in[i] : i;
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
      a.to.keys[j] = a.keys_in[i];
  }
}

/* a new flight into slot q (interaction kernel) */
template <bool FULL, bool DEFER = false>
__device__ __forceinline__ void
write_flight_row(const FlightRowsDev &out, unsigned int q, uint32_t *key_out,
                 const Packet<FULL> &p, uint32_t packed_lc, uint32_t key,
                 uint32_t packet_id, uint32_t meta,
                 const double (&weights)[CMI_NACC]) {
  double4 *r4 = reinterpret_cast<double4 *>(out.rows +
                                            (size_t)CMI_FLIGHT_DOUBLES * q);
  r4[0] = make_double4(p.pos[0], p.pos[1], p.pos[2], p.dir[0]);
  r4[1] = make_double4(
      p.dir[1], p.dir[2], p.nu,
      __longlong_as_double(
          (long long)(((unsigned long long)meta << 32) | packet_id)));
  r4[2] = make_double4(p.t, p.tmax[0], p.tmax[1], p.tmax[2]);
  r4[3] = make_double4(
      p.tau,
      __longlong_as_double((long long)(((unsigned long long)packed_lc << 32) |
                                       (uint32_t)p.cell)),
      p.tdelta[0], p.tdelta[1]);
  *key_out = key;
  if (FULL && !DEFER) {
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
 * incremental marcher of device_transport.h - here with tile-local
 * bookkeeping only - and both the tile's transport records
 * (get_optical_depth, src/DensityGrid.hpp:117-140) and its accumulators
 * (update_integrals, :150-197) in LDS: the march loop touches no global
 * memory.
 */
template <bool FULL, bool HEAT>
__global__ void __launch_bounds__((TileShape<FULL, HEAT>::THREADS),
                                  (!FULL && HEAT) ? 2 : CMI_TILE_WAVES)
    tile_kernel(const TileArgs a) {
  using Shape = TileShape<FULL, HEAT>;
  constexpr int LX = Shape::LX, LY = Shape::LY, LZ = Shape::LZ;
  constexpr int TX = Shape::TX, TY = Shape::TY, TZ = Shape::TZ;
  constexpr int TC = Shape::CELLS;
  constexpr int NV = FULL ? CMI_NACC : (HEAT ? 2 : 1);
  constexpr int NT = Shape::THREADS;
  /* value v of tile cell k lives at acc[v * TC + k]: neighbouring cells in
   * neighbouring banks whatever the value */
  __shared__ double acc[NV * TC];
  /* transport records of the tile's cells: n x_H (< 0: vacuum), and n x_He
   * for multi-ion transport */
  __shared__ double opac[(FULL ? 2 : 1) * TC];
  /* (three claims in flight: the slot thread 0 fills in trip k - the unit after
   * next - is one no wave can still be reading: its last reader was the end
   * of trip k - 3, barriers ago; with two slots the first trip's claim
   * overwrote the slot the other waves read after the prologue's barrier) */
  __shared__ unsigned int s_ring[3], s_next, s_nabs;

  const int lane = threadIdx.x & 63;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  const uint32_t key_dead = CMI_TILE_KEY_DEAD(a.tiles);

  /* marcher state of the lane's flight, tile-local */
  double pos[3], dir[3], tmax[3], tdelta[3];
  double t = 0., tau = 0., nu = 0., sigma_H = 0., sigma_He_corr = 0.;
  double weights[CMI_NACC];
  int32_t lc[3] = {0, 0, 0}, lsgn[3] = {0, 0, 0};
  int32_t type = 0;
  uint32_t packet_id = 0, lane_meta = 0, slot = 0, place = 0;
  double tc0 = 0., tc1 = 0., tc2 = 0., tc3 = 0.; /* sums of packet weights */
  unsigned int nsteps = 0, natomics = 0, nwavesteps = 0;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
    pos[ax] = dir[ax] = tmax[ax] = tdelta[ax] = 0.;

  /* The units of work of this workgroup, claimed one after the other from
   * the round's counter - two ahead: while the flights of unit k are in the
   * march, the descriptor and the transport records of unit k + 1 are on
   * their way into registers (multi-ion tiles: one record per thread) and
   * the claim of unit k + 2 is in flight. Late rounds consist of thousands of
   * units with a hundred flights each; claimed, described and loaded one
   * after the other, each unit cost ~18 us of dependent latencies (returning
   * atomic -> descriptor -> records -> order -> rows) whatever its flights -
   * 1.2 ms per round of a 256^3 grid's 32768 multi-ion tiles, half the tile
   * kernel's time in lexingtonHII40 (49.9 -> 43.3 ms per iteration with the
   * units fetched ahead).
   * (Hydrogen-only tiles - eight records per thread, a tenth as many units -
   * claim, describe and load a unit when they get to it: measured with the
   * descriptor alone fetched ahead, 76.4 -> 78.2 ms per diffuse iteration.) */
  constexpr bool PREFETCH = (TC <= NT); /* one record per thread */
  const unsigned int nitems = *a.nitems;
  unsigned int item = 0, item_next = 0;
  TileItemDev it;
  it.tile = it.begin = it.end = it.pad = 0;
  double2 rec_ahead = make_double2(-1., 0.);
  /* tile coordinates of a unit's tile and its extent (the last tile of an
   * axis may be clipped) */
  auto tile_origin = [&](const TileItemDev &d, int32_t (&o)[3],
                         int32_t (&td)[3]) {
    const int32_t tz = (int32_t)(d.tile % (uint32_t)a.tiles.ntile[2]);
    const int32_t ty = (int32_t)((d.tile / (uint32_t)a.tiles.ntile[2]) %
                                 (uint32_t)a.tiles.ntile[1]);
    const int32_t tx = (int32_t)(d.tile / ((uint32_t)a.tiles.ntile[2] *
                                           (uint32_t)a.tiles.ntile[1]));
    o[0] = tx << LX;
    o[1] = ty << LY;
    o[2] = tz << LZ;
    td[0] = a.grid.ncell[0] - o[0] < TX ? a.grid.ncell[0] - o[0] : TX;
    td[1] = a.grid.ncell[1] - o[1] < TY ? a.grid.ncell[1] - o[1] : TY;
    td[2] = a.grid.ncell[2] - o[2] < TZ ? a.grid.ncell[2] - o[2] : TZ;
  };
  auto load_record = [&](const int32_t (&o)[3], const int32_t (&td)[3], int k) {
    const int32_t lx = k >> (LY + LZ), ly = (k >> LZ) & (TY - 1),
                  lz = k & (TZ - 1);
    double2 rec = make_double2(-1., 0.);
    if (lx < td[0] && ly < td[1] && lz < td[2])
      rec = a.cells.opacity[((int64_t)(o[0] + lx) * a.grid.ncell[1] + o[1] +
                             ly) * a.grid.ncell[2] + o[2] + lz];
    return rec;
  };
  if (PREFETCH) {
    if (threadIdx.x == 0) {
      s_ring[0] = atomicAdd(a.next_item, 1u);
      s_ring[1] = atomicAdd(a.next_item, 1u);
    }
    __syncthreads();
    item = s_ring[0];
    item_next = s_ring[1];
    if (item < nitems) {
      it = a.items[item];
      int32_t o[3], td[3];
      tile_origin(it, o, td);
      if ((int)threadIdx.x < TC)
        rec_ahead = load_record(o, td, threadIdx.x);
    }
  }

  for (int trip = 0; item < nitems; ++trip) {
    if (!PREFETCH) {
      if (threadIdx.x == 0)
        s_ring[0] = atomicAdd(a.next_item, 1u);
      for (int k = threadIdx.x; k < NV * TC; k += NT)
        acc[k] = 0.;
      __syncthreads();
      item = s_ring[0];
      if (item >= nitems)
        break;
      it = a.items[item];
    }
    if (threadIdx.x == 0) {
      if (PREFETCH)
        s_ring[(trip + 2) % 3] =
            atomicAdd(a.next_item, 1u); /* the one after next */
      s_nabs = 0;
      s_next = it.begin;
    }
    int32_t o[3], td[3];
    tile_origin(it, o, td);
    const bool clipped = td[0] != TX || td[1] != TY || td[2] != TZ;
    if (PREFETCH) {
      for (int k = threadIdx.x; k < NV * TC; k += NT)
        acc[k] = 0.;
      if ((int)threadIdx.x < TC) {
        opac[threadIdx.x] = rec_ahead.x;
        if (FULL)
          opac[TC + threadIdx.x] = rec_ahead.y;
      }
    } else {
      for (int k = threadIdx.x; k < TC; k += NT) {
        const double2 rec = load_record(o, td, k);
        opac[k] = rec.x;
        if (FULL)
          opac[TC + k] = rec.y;
      }
    }
    __syncthreads();
    /* the next unit's descriptor and records: needed after this unit's march */
    TileItemDev it_ahead = it;
    if (PREFETCH && item_next < nitems) {
      it_ahead = a.items[item_next];
      int32_t on[3], tdn[3];
      tile_origin(it_ahead, on, tdn);
      if ((int)threadIdx.x < TC)
        rec_ahead = load_record(on, tdn, threadIdx.x);
    }

    bool active = false;
    int32_t last_lidx = 0;
    bool stepped = false; /* the flight has crossed a cell in this tile */
    /* not zero once a coordinate has left [0, side) (negative ones included) */
    auto beyond = [&]() -> int32_t {
      if constexpr (LX == LY && LY == LZ)
        return (int32_t)(((uint32_t)(lc[0] | lc[1] | lc[2])) >> LX);
      else
        return (lc[0] >> LX) | (lc[1] >> LY) | (lc[2] >> LZ);
    };
    auto in_tile = [&]() {
      bool in = beyond() == 0;
      if (clipped)
        in = in && lc[0] < td[0] && lc[1] < td[1] && lc[2] < td[2];
      return in;
    };
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
          place = i;
          const uint32_t from = a.order[i];
          slot = a.slot_in ? a.slot_in[from] : from;
          a.slot_out[i] = slot;
          const double4 *r = reinterpret_cast<const double4 *>(
              a.rows.rows + (size_t)CMI_FLIGHT_DOUBLES * slot);
          const double4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
          pos[0] = r0.x;
          pos[1] = r0.y;
          pos[2] = r0.z;
          dir[0] = r0.w;
          dir[1] = r1.x;
          dir[2] = r1.y;
          nu = r1.z;
          const unsigned long long idmeta =
              (unsigned long long)__double_as_longlong(r1.w);
          packet_id = (uint32_t)idmeta;
          lane_meta = (uint32_t)(idmeta >> 32);
          t = r2.x;
          tmax[0] = r2.y;
          tmax[1] = r2.z;
          tmax[2] = r2.w;
          tau = r3.x;
          const uint32_t plc =
              (uint32_t)((unsigned long long)__double_as_longlong(r3.y) >> 32);
          type = (int32_t)(lane_meta >> 28);
          /* the packet's weight (Photon::get_weight; 1 unless both kinds of
           * sources are present) goes into the accumulation weights once */
          const double pw = a.model.photon_weight[cmi_meta_origin(lane_meta)];
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
            sigma_H = weights[ION_H_n];
            sigma_He_corr = a.model.abundance[0] * weights[ION_He_n];
#pragma unroll
            for (int k = 0; k < CMI_NACC; ++k)
              weights[k] *= pw;
          } else {
            sigma_H = a.model.xsec_fixed[ION_H_n];
            weights[ION_H_n] = sigma_H * pw;
            weights[CMI_NION] = sigma_H * (nu - a.model.nu_H) * pw;
          }
          /* the wall spacings start_flight() computed: two travel in the
           * slot, the third is its expression again (the increments must be
           * the same numbers in every tile the flight crosses) - two
           * divisions less per visit */
          tdelta[0] = r3.z;
          tdelta[1] = r3.w;
          {
            const double inv_dir = 1. / dir[2];
            tdelta[2] =
                (dir[2] != 0.) ? a.grid.cellside[2] * fabs(inv_dir) : 0.;
          }
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            lc[ax] = (int32_t)((plc >> (8 * ax)) & 0xffu);
            lsgn[ax] = (dir[ax] > 0.) ? 1 : -1;
          }
          active = true;
          stepped = false;
        }
      }
      const bool avail_after =
          __ballot(*(volatile unsigned int *)&s_next < it.end) != 0ull;

      /* ---- hot loop: fast_step() of device_transport.h (the same
       * floating-point operations in the same order) on tile-local
       * coordinates, records and accumulators in LDS ---- */
      int32_t lidx = Shape::index(lc[0], lc[1], lc[2]);
      double kx = 0., ky = 0.;
      if (active && tau > 0. && in_tile()) {
        kx = opac[lidx];
        if (FULL)
          ky = opac[TC + lidx];
      }
      /* (round 4: the lanes in flight as a scalar mask straight from the
       * compares, the refill test on scalars, the axis advances under the
       * execution mask - see the first generation's loop in kernels.h) */
      const unsigned long long active_lanes = wave_ballot(active);
      const int idle_limit = __builtin_amdgcn_readfirstlane(
          avail_after ? a.refill_threshold : 65);
      for (;;) {
        unsigned long long flying =
            active_lanes & mask_gt(tau, 0.) &
            mask_eq(beyond(), 0);
        if (clipped)
          flying &= wave_ballot(lc[0] < td[0] && lc[1] < td[1] &&
                                lc[2] < td[2]);
        if (flying == 0ull || (int)__popcll(~flying) >= idle_limit)
          break;
        const bool stepping = lanes_of(flying);
        ++nwavesteps;
        if (stepping) {
          last_lidx = lidx;
          stepped = true;
          const double tmin = min_f64(tmax[0], min_f64(tmax[1], tmax[2]));
          double ds = tmin - t;
          const double kH = max_f64(kx, 0.);
          const double tau_cell = FULL
                                      ? ds * (sigma_H * kH + sigma_He_corr * ky)
                                      : ds * (sigma_H * kH);
          const bool matter = kx >= 0.; /* number density > 0 */
          tau -= tau_cell;
          const double t_old = t;
          t = tmin;
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            /* every tied axis advances (tmax + tdelta: the exactly rounded
             * sum, as fast_step()'s) */
            unsigned long long saved;
            asm volatile("s_and_saveexec_b64 %2, %3\n\t"
                         "v_add_f64 %0, %0, %4\n\t"
                         "v_add_u32 %1, %1, %5\n\t"
                         "s_mov_b64 exec, %2"
                         : "+v"(tmax[ax]), "+v"(lc[ax]), "=&s"(saved)
                         : "s"(mask_eq(tmax[ax], tmin)), "v"(tdelta[ax]),
                           "v"(lsgn[ax])
                         : "scc");
          }
          if (tau < 0.) {
            ds += ds * tau / tau_cell; /* Scorr */
            t = t_old + ds;
          }
          ++nsteps;
          lidx = Shape::index(lc[0], lc[1], lc[2]);
          if (tau > 0. && in_tile()) {
            kx = opac[lidx];
            if (FULL)
              ky = opac[TC + lidx];
          }
          if (matter) {
            if (FULL) {
#pragma unroll
              for (int i = 0; i < CMI_NACC; ++i)
                /* (most cross sections of a re-emitted photon are zero - it
                 * sits below those ions' thresholds: no add, same sum) */
                if ((HEAT || i < CMI_NION) && weights[i] != 0.)
                  atomicAdd(&acc[i * TC + last_lidx], ds * weights[i]);
            } else {
              atomicAdd(&acc[last_lidx], ds * weights[ION_H_n]);
              if (HEAT)
                atomicAdd(&acc[TC + last_lidx], ds * weights[CMI_NION]);
            }
          }
        }
      }

      /* ---- end of the tile visit for every lane that cannot step ---- */
      if (active && !(tau > 0. && in_tile())) {
        /* where the flight is now: coordinates in the engine's grid (one past
         * a face if it has left it); a periodic axis wraps, and the origin of
         * the flight shifts by a box side (fast_wrap()) */
        int32_t g[3];
        bool outside_grid = false, wrapped = false;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          g[ax] = o[ax] + lc[ax];
          if (tau >= 0. && (g[ax] < 0 || g[ax] >= a.grid.ncell[ax])) {
            if (a.grid.periodic[ax]) {
              pos[ax] -= (g[ax] < 0 ? -1. : 1.) * a.grid.box_sides[ax];
              g[ax] = g[ax] < 0 ? a.grid.ncell[ax] - 1 : 0;
              wrapped = true;
            } else {
              outside_grid = true;
            }
          }
        }
        bool absorbed = false, done = false, moved = false;
        if (tau < 0.) {
          absorbed = true;
        } else if (!outside_grid && !(tau > 0.)) {
          /* tau hit 0 exactly on a wall, packet still inside the grid:
           * interact() returns the last traversed cell */
          absorbed = stepped;
          done = !absorbed;
        } else if (outside_grid) {
          done = true; /* left the grid: DensityGrid::end() ... */
          if (a.grid.decomposed && stepped) {
            /* ... or only this block of it: hand the flight over if the cell
             * it enters exists in the whole grid */
            bool in_whole = true;
            int64_t gg[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
              gg[ax] = (int64_t)g[ax] + a.grid.offset[ax];
              if (gg[ax] < 0 || gg[ax] >= a.grid.global_ncell[ax]) {
                if (a.grid.global_periodic[ax]) {
                  /* across a periodic face of the whole box */
                  pos[ax] += (gg[ax] < 0 ? 1. : -1.) * a.grid.box_sides[ax];
                  gg[ax] = gg[ax] < 0 ? a.grid.global_ncell[ax] - 1 : 0;
                } else {
                  in_whole = false;
                }
              }
            }
            if (in_whole) {
              const int64_t cell_global =
                  (gg[0] * a.grid.global_ncell[1] + gg[1]) *
                      a.grid.global_ncell[2] +
                  gg[2];
              const unsigned long long leaving = __ballot(true);
              unsigned int base = 0;
              const int first = __ffsll((long long)leaving) - 1;
              if (lane == first)
                base = atomicAdd(a.xout.count, (unsigned int)__popcll(leaving));
              base = __shfl(base, first, 64);
              const unsigned int q = base + __popcll(leaving & lane_lt);
              if (q < a.xout.capacity) {
                double *r = a.xout.rows + (size_t)CMI_FLIGHT_DOUBLES * q;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                  r[ax] = pos[ax];
                  r[3 + ax] = dir[ax];
                  r[7 + ax] = tmax[ax];
                }
                r[6] = t;
                r[10] = tau;
                r[11] = nu;
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
          /* the marcher's state at the wall, into the same slot */
          const uint32_t plc = (uint32_t)(g[0] & (TX - 1)) |
                               ((uint32_t)(g[1] & (TY - 1)) << 8) |
                               ((uint32_t)(g[2] & (TZ - 1)) << 16);
          key = tile_index(a.tiles, g[0] >> LX, g[1] >> LY, g[2] >> LZ);
          const uint32_t cell =
              (uint32_t)((g[0] * a.grid.ncell[1] + g[1]) * a.grid.ncell[2] +
                         g[2]);
          if (wrapped) {
            r[0] = pos[0];
            r[1] = pos[1];
            r[2] = pos[2];
          }
          double4 *r4 = reinterpret_cast<double4 *>(r);
          r4[2] = make_double4(t, tmax[0], tmax[1], tmax[2]);
          r4[3] = make_double4(
              tau,
              __longlong_as_double(
                  (long long)(((unsigned long long)plc << 32) | cell)),
              tdelta[0], tdelta[1]);
        } else if (absorbed) {
          /* the absorption record, for the interaction kernel of this round:
           * where (end_flight()), in which cell, which packet - densely, in
           * this unit's stretch of the record arrays; the slot keeps its key
           * until that kernel has decided */
          const int32_t lx = last_lidx >> (LY + LZ),
                        ly = (last_lidx >> LZ) & (TY - 1),
                        lz = last_lidx & (TZ - 1);
          const int32_t cell =
              ((o[0] + lx) * a.grid.ncell[1] + o[1] + ly) * a.grid.ncell[2] +
              o[2] + lz;
          const unsigned int q = it.begin + atomicAdd(&s_nabs, 1u);
#pragma unroll
          for (int ax = 0; ax < 3; ++ax)
            a.ended.pos[ax][q] = pos[ax] + t * dir[ax];
          a.ended.nu[q] = nu;
          a.ended.cell[q] = cell;
          a.ended.id[q] = packet_id;
          a.ended.meta[q] = lane_meta;
          a.ended_slot[q] = slot;
          a.ended_pos[q] = place;
        }
        if (!absorbed)
          a.keys_out[place] = key;
        if (done) {
          const double w = a.model.photon_weight[cmi_meta_origin(lane_meta)];
          tc0 += (type == TYPE_PRIMARY) ? w : 0.;
          tc1 += (type == TYPE_DIFFUSE_HI) ? w : 0.;
          tc2 += (type == TYPE_DIFFUSE_HeI) ? w : 0.;
          tc3 += (type == TYPE_ABSORBED) ? w : 0.;
        }
        active = false;
      }
    }

    /* ---- write the tile back: one full-line atomic per 8 values; hand the
     * absorbed slots to the interaction kernel ---- */
    __syncthreads();
    if (threadIdx.x == 0)
      a.absorbed_count[item] = s_nabs;
    if (FULL) {
      const int i = threadIdx.x & 15;
      if (HEAT || i < CMI_NION) {
        for (int k = threadIdx.x >> 4; k < TC; k += NT / 16) {
          const double v = acc[i * TC + k];
          if (v != 0.) {
            const int32_t lx = k >> (LY + LZ), ly = (k >> LZ) & (TY - 1),
                          lz = k & (TZ - 1);
            const int64_t cell =
                ((int64_t)(o[0] + lx) * a.grid.ncell[1] + o[1] + ly) *
                    a.grid.ncell[2] +
                o[2] + lz;
            atomic_add_f64(acc_at(a.cells, i, cell), v);
            ++natomics;
          }
        }
      }
    } else {
      for (int k = threadIdx.x; k < TC; k += NT) {
        const double v = acc[k];
        const double h = HEAT ? acc[TC + k] : 0.;
        if (v != 0. || h != 0.) {
          const int32_t lx = k >> (LY + LZ), ly = (k >> LZ) & (TY - 1),
                        lz = k & (TZ - 1);
          const int64_t cell =
              ((int64_t)(o[0] + lx) * a.grid.ncell[1] + o[1] + ly) *
                  a.grid.ncell[2] +
              o[2] + lz;
          if (v != 0.) {
            atomic_add_f64(acc_at(a.cells, ION_H_n, cell), v);
            ++natomics;
          }
          if (HEAT && h != 0.) {
            atomic_add_f64(acc_at(a.cells, CMI_NION, cell), h);
            ++natomics;
          }
        }
      }
    }
    __syncthreads();
    if (PREFETCH) {
      item = item_next;
      item_next = s_ring[(trip + 2) % 3];
      it = it_ahead;
    }
  }

  const double s0 = wave_sum(tc0);
  const double s1 = wave_sum(tc1);
  const double s2 = wave_sum(tc2);
  const double s3 = wave_sum(tc3);
  const double ns = wave_sum((double)nsteps);
  const double na = wave_sum((double)natomics);
  if (lane == 0) {
    if ((s0 + s1) + (s2 + s3) != 0.) {
      atomic_add_f64(&counter_shard(a.counters)->totweight, (s0 + s1) + (s2 + s3));
      atomic_add_f64(&counter_shard(a.counters)->typecount[0], s0);
      atomic_add_f64(&counter_shard(a.counters)->typecount[1], s1);
      atomic_add_f64(&counter_shard(a.counters)->typecount[2], s2);
      atomic_add_f64(&counter_shard(a.counters)->typecount[3], s3);
    }
    atomicAdd(&counter_shard(a.counters)->nsteps, (unsigned long long)ns);
    atomicAdd(&counter_shard(a.counters)->natomics, (unsigned long long)na);
    atomicAdd(&counter_shard(a.counters)->nwavesteps, (unsigned long long)nwavesteps);
  }
}

#endif
