/*
 * kernels.h - the engine's HIP kernels (gfx950, wave64).
 */
#ifndef CMI_KERNELS_H
#define CMI_KERNELS_H

#include "device_transport.h"
#include "device_reemit.h"
#include "device_thermal.h"
#include "device_emissivity.h"

#ifndef CMI_BLOCK
#define CMI_BLOCK 256
#endif
/* the "exp_no_atomics" experiments (results are wrong by design) exist only in
 * builds with -DCMI_EXPERIMENTS (make variant ...); the product kernels carry
 * none of their branches */
#ifdef CMI_EXPERIMENTS
#define CMI_EXP(a) ((a).exp_no_atomics)
#else
#define CMI_EXP(a) 0
#endif
/* slots of a block's combining table (a.aggregate == CMI_AGG_BLOCK) */
#ifndef CMI_TABLE_BITS
#define CMI_TABLE_BITS 10
#endif
#define CMI_TABLE_SLOTS (1 << CMI_TABLE_BITS)
/* threads per block and table slots of the hydrogen-only kernels that are
 * BUILT for the table (TABLE: the first generation). The waves of a block share
 * the table and meet at a barrier between two bundles: 8 waves with 2048 slots
 * combine more than 4 with 1024 and wait no longer for each other (measured,
 * key + sort + kernel of config 2 in ms: 128 threads / 512 slots 47.3, 256 /
 * 1024 34.4, 512 / 2048 33.4, 1024 / 4096 34.6; 256 / 512 37.1, 256 / 2048
 * 56.1, 512 / 1024 34.8, 512 / 4096 54.5 - LDS then limits the waves per CU).
 * Every other kernel is slower with 512 threads (CMI_BLOCK). */
#ifndef CMI_TABLE_BLOCK
#define CMI_TABLE_BLOCK 512
#endif
#ifndef CMI_TABLE_BLOCK_BITS
#define CMI_TABLE_BLOCK_BITS 11
#endif
/* ... and on grids of more than 2^25 cells (BIG: engine.hip picks the build),
 * where a bundle crosses more cells than the smaller table holds: 1024
 * threads with 4096 slots (measured, key + sort + kernel in ms, 512 / 1024
 * threads: 256^3 33.4 / 34.6, 384^3 49.3 / 48.4, 512^3 73.0 / 66.5) */
#define CMI_TABLE_BLOCK_BIG 1024
#define CMI_TABLE_BLOCK_BIG_BITS 12
#define CMI_TABLE_BIG_CELLS (1ll << 25)
template <bool FULL, bool TABLE, bool BIG = false>
constexpr int shoot_block_threads() {
  return (TABLE && !FULL) ? (BIG ? CMI_TABLE_BLOCK_BIG : CMI_TABLE_BLOCK)
                          : CMI_BLOCK;
}
/* (measured in round 5, ms per iteration of config 2: 2 probes 35.1, 4: 34.5,
 * 8: 34.2) */
#ifndef CMI_TABLE_PROBES
#define CMI_TABLE_PROBES 8
#endif
/* experiment (round 6): this many low bits of a PAD table slot come straight
 * from the padded cell index (0: all hashed) */
#ifndef CMI_PAD_HASH_LOW_BITS
#define CMI_PAD_HASH_LOW_BITS 0
#endif
/* experiment: look at a slot's tag with a plain LDS read before trying the
 * compare-and-swap (a hit then costs no LDS atomic; measured: 39.0 instead of
 * 34.5 ms - the second round trip of every first visit costs more than the
 * atomics of the hits) */
#ifndef CMI_TABLE_READ_FIRST
#define CMI_TABLE_READ_FIRST 0
#endif
/* run sums in front of the table cover groups of 2^this lanes */
#ifndef CMI_TABLE_SCAN_ROUNDS
#define CMI_TABLE_SCAN_ROUNDS 2
#endif
/* multi-ion kernels: slots (of 16 doubles) of the block's combining table and
 * march-loop iterations between two write-backs of it (measured on
 * lexingtonHII40 256^3, first generation of 1e8 packets: 4 -> 206 ms, 6 -> 199,
 * 8 -> 196, 10 -> 196, 16 -> 201, 24 -> 215, 32 -> 240: a longer window
 * overflows the table into per-lane atomics, a shorter one meets at the
 * barrier more often) */
#define CMI_FTABLE_BITS 7
#define CMI_FTABLE_SLOTS (1 << CMI_FTABLE_BITS)
#ifndef CMI_FTABLE_WINDOW
#define CMI_FTABLE_WINDOW 8
#endif
/* a.aggregate: what happens to a step's contributions before HBM sees them */
#define CMI_AGG_NONE 0  /* one atomic per lane and step */
#define CMI_AGG_RUNS 1  /* cross-lane run sums, one atomic per run */
#define CMI_AGG_BLOCK 2 /* + per-block combining table in LDS */
/* idle lanes of a wave are refilled with new packets once this many of them
 * are waiting (or when the whole wave is idle). 64 = a wave always carries one
 * group of 64 direction-sorted packets: its lanes stay in the same cells, so
 * the cross-lane sums collapse most atomics (measured on MI355X, 256^3
 * Stromgren: threshold 16/32/48/64 -> 0.80/0.59/0.39/0.20 atomics per step and
 * 223/152/88/29 ms per 2e7 packets; the idle lanes cost far less than the
 * atomics they save). */
#define CMI_REFILL_THRESHOLD 64

/* hardware fp64 atomic add (global_atomic_add_f64), no CAS loop */
__device__ __forceinline__ void atomic_add_f64(double *address, double value) {
  unsafeAtomicAdd(address, value);
}

/* the wave's mask of a condition straight from the compare (HIP's __ballot
 * goes through a 0/1 value and a second compare: two vector instructions) */
__device__ __forceinline__ unsigned long long wave_ballot(bool condition) {
  return __builtin_amdgcn_ballot_w64(condition);
}

/* Conditions as wave masks in scalar registers, straight from the compare
 * (v_cmp writes an SGPR pair; no 0/1 value in between), and back: the mask as
 * the lanes' predicate (no instruction at all). Loop conditions, run
 * boundaries and the table's "still looking for a slot" are kept that way:
 * their bookkeeping is scalar arithmetic on 64-bit masks instead of vector
 * instructions on flags. (LLVM's comparison predicates.) */
#define CMI_FCMP_OEQ 1
#define CMI_FCMP_OGT 2
#define CMI_ICMP_EQ 32
#define CMI_ICMP_NE 33
#define CMI_ICMP_SGE 39
__device__ __forceinline__ unsigned long long mask_gt(double x, double y) {
  return __builtin_amdgcn_fcmp(x, y, CMI_FCMP_OGT);
}
__device__ __forceinline__ unsigned long long mask_eq(double x, double y) {
  return __builtin_amdgcn_fcmp(x, y, CMI_FCMP_OEQ);
}
__device__ __forceinline__ unsigned long long mask_eq(int32_t x, int32_t y) {
  return __builtin_amdgcn_sicmp(x, y, CMI_ICMP_EQ);
}
__device__ __forceinline__ unsigned long long mask_ne(int32_t x, int32_t y) {
  return __builtin_amdgcn_sicmp(x, y, CMI_ICMP_NE);
}
__device__ __forceinline__ unsigned long long mask_ge(int32_t x, int32_t y) {
  return __builtin_amdgcn_sicmp(x, y, CMI_ICMP_SGE);
}
__device__ __forceinline__ bool lanes_of(unsigned long long mask) {
  return __builtin_amdgcn_inverse_ballot_w64(mask);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  return v;
}

/* id of a place of the ended queue that holds no flight
 * (ShootArgs::park_in_place; packet ids of a call are < 2^32 - 1) */
#define CMI_QUEUE_HOLE 0xffffffffu

struct ShootArgs {
  GridDev grid;
  ModelDev model;
  CellsDev cells;
  CountersDev *counters;
  /* the launch handles positions [0, n_packets); position i is packet
   * first_packet + batch_offset + order[i] of the iteration (emission), or
   * entry i of qin / xin. Packet ids in queues and hand-over records are
   * relative to first_packet. */
  uint64_t first_packet;
  uint64_t batch_offset;
  uint64_t n_packets;
  /* processing order: position i of the launch handles packet
   * first_packet + order[i] (direction-sorted); NULL = identity */
  const uint32_t *order;
  /* PRE: row (packet id - batch_offset) holds what emit_physics would
   * compute for that packet - 14 cross sections, the frequency, the optical
   * depth - written by direction_key_kernel */
  const double *pre_rows;
  uint32_t chunk; /* consecutive positions a wave consumes before it jumps */
  uint32_t seed;
  uint32_t iteration;
  int32_t refill_threshold;
  int32_t exp_no_atomics; /* experiment: skip the accumulation */
  TrackersDev trackers;   /* EXACT kernels only */
  int32_t aggregate;      /* CMI_AGG_* */
  /* qin.id != NULL: this launch flies the ready flights of qin instead of
   * emitting new packets. qout.id != NULL: a packet that is absorbed is parked
   * in qout (an "ended flight") for the interaction kernel to decide about
   * its re-emission; otherwise it is re-emitted in place (REEMIT variants) or
   * counted as absorbed. */
  QueueDev qin, qout;
  /* qout of a launch of NEW packets: an absorbed packet is parked at the
   * place of its position in the launch's order (qout.id is pre-set to
   * CMI_QUEUE_HOLE; the interaction kernel runs over all n_packets places and
   * skips the holes) instead of at a place claimed from the queue's counter -
   * the bundles of a launch end one after the other, 1.6e6 returning atomics
   * on one word per 1e8 packets, ~3 ms of the first generation of configs 3
   * and 4 */
  int32_t park_in_place;
  /* decomposed grids: xin != NULL: this launch continues the flights handed
   * over by other blocks (CMI_FLIGHT_DOUBLES doubles each); packets that
   * leave this block into another one are appended to xout */
  const double *xin;
  /* xin rows hold the long index of the entered cell in THIS engine's grid
   * (flights left over by the tile rounds) instead of the whole grid's */
  int32_t xin_local;
#ifdef CMI_DBG_XIN_SLOTS
  /* (the episode of DESIGN_LOG.md "A result that changed with one more kernel
   * argument": the member and the indirection of commit 3e4ff5c, for
   * tools/debug/episode.sh) */
  const uint32_t *xin_slots = nullptr;
#endif
  ExchangeDev xout;
  /* PAD kernels: n x_H of every cell of the grid with CMI_PAD_LAYERS layers
   * of ghost cells around it (pad_record_kernel): -0. marks a vacuum cell
   * (no opacity, and a sign bit that says "no gas"), -2 a ghost cell - the
   * march learns from the record it loads anyway that the packet has left
   * the box */
  const double *pad_H;
  int32_t xcd_remap;
  /* padded extents ny + 2 L, nz + 2 L (L = CMI_PAD_LAYERS) and the
   * reciprocals of (ny + 2 L)(nz + 2 L) and nz + 2 L */
  int32_t pad_ny, pad_nz;
  double pad_inv_yz, pad_inv_z;
};

/* waves per SIMD the multi-ion first generation is built for (128 VGPRs and
 * 33 KB of LDS per block at 4; the other multi-ion variants keep their table
 * and weights stage - 49 KB - and 3 waves) */
#ifndef CMI_FULL_WAVES
#define CMI_FULL_WAVES 4
#endif
/* waves per SIMD the PAD kernel is built for (64 VGPRs; measured 46.5 -> 44.0
 * ms per iteration of 1e8 packets against 6; the variant with the heating
 * term does not fit 64 registers and stays at 6) */
#ifndef CMI_PAD_WAVES
#define CMI_PAD_WAVES 8
#endif
/* rounds of the run-sum scan in front of the PAD kernel's table (0: groups of
 * four lanes by quad_perm reads) */
#ifndef CMI_PAD_SCAN_ROUNDS
#define CMI_PAD_SCAN_ROUNDS 3
#endif
/* ghost layers around the grid */
#define CMI_PAD_LAYERS 1
#define CMI_PAD_VACUUM (-0.)
#define CMI_PAD_GHOST (-2.)
/* long index in the grid of the padded long index c of a cell inside it
 * (c < 2^29; (c + 0.5) / d is never within 0.5 / d of an integer, far more
 * than the rounding of the product) */
__device__ __forceinline__ int32_t cmi_unpad_cell(const ShootArgs &a,
                                                  const GridDev &g,
                                                  int32_t c) {
  const double cc = (double)c + 0.5;
  const int32_t ix = (int32_t)(cc * a.pad_inv_yz);
  const int32_t r = c - ix * (a.pad_ny * a.pad_nz);
  const int32_t iy = (int32_t)(((double)r + 0.5) * a.pad_inv_z);
  const int32_t iz = r - iy * a.pad_nz;
  return ((ix - CMI_PAD_LAYERS) * g.ncell[1] + (iy - CMI_PAD_LAYERS)) *
             g.ncell[2] +
         (iz - CMI_PAD_LAYERS);
}

/* PAD kernels on a block of a decomposed grid: the packet has stepped out of
 * the block into the ghost cell p.cell (a padded long index). exit_cell_global
 * (device_transport.h) for that cell: its long index in the WHOLE grid, or -1
 * if it lies outside the whole grid; across a periodic face of the whole box
 * the flight's origin is shifted by a box side. (One ghost layer is enough:
 * every tied axis advances by one cell per step.) */
template <bool FULL>
__device__ __forceinline__ int64_t
exit_cell_global_padded(const ShootArgs &a, const GridDev &g, Packet<FULL> &p) {
  const int32_t c = p.cell;
  const int32_t ix = (int32_t)(((double)c + 0.5) * a.pad_inv_yz);
  const int32_t r = c - ix * (a.pad_ny * a.pad_nz);
  const int32_t iy = (int32_t)(((double)r + 0.5) * a.pad_inv_z);
  const int32_t iz = r - iy * a.pad_nz;
  int64_t gc[3] = {(int64_t)ix - CMI_PAD_LAYERS + g.offset[0],
                   (int64_t)iy - CMI_PAD_LAYERS + g.offset[1],
                   (int64_t)iz - CMI_PAD_LAYERS + g.offset[2]};
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    if (gc[ax] < 0 || gc[ax] >= g.global_ncell[ax]) {
      if (!g.global_periodic[ax])
        return -1;
      /* across a periodic face of the whole box: is_inside()'s wrap */
      p.pos[ax] += (gc[ax] < 0 ? 1. : -1.) * g.box_sides[ax];
      gc[ax] = gc[ax] < 0 ? g.global_ncell[ax] - 1 : 0;
    }
  }
  return (gc[0] * g.global_ncell[1] + gc[1]) * g.global_ncell[2] + gc[2];
}

/* quad_perm DPP controls: lane j of every group of 4 reads lane j - 1 / j - 2
 * / j - 3 / j + 1 of its group (lanes without such a neighbour read some lane
 * of the group; the caller masks them) */
#define CMI_DPP_QUAD_SHR1 0x90 /* [0,0,1,2] */
#define CMI_DPP_QUAD_SHR2 0x40 /* [0,0,0,1] */
#define CMI_DPP_QUAD_SHR3 0x00 /* [0,0,0,0] */
#define CMI_DPP_QUAD_SHL1 0xF9 /* [1,2,3,3] */

/* update_integrals, src/DensityGrid.hpp:150-197, hydrogen-only form without
 * cross-lane aggregation: the crossed non-vacuum cell receives ds * w *
 * sigma_H, and the hydrogen heating term. */
template <bool HEAT>
__device__ __forceinline__ void update_integrals_H(const ShootArgs &a,
                                                   double sigma_H, double nu,
                                                   double weight, int64_t cell,
                                                   double ds) {
  const double dsw = ds * weight;
  atomic_add_f64(acc_at(a.cells, ION_H_n, cell), dsw * sigma_H);
  if (HEAT)
    atomic_add_f64(acc_at(a.cells, CMI_NION, cell),
                   dsw * sigma_H * (nu - a.model.nu_H));
}

/* DPP lane moves (gfx9 family): no LDS round trip, VALU latency only.
 * dpp_keep: lanes without a source lane keep `old`; dpp_zero: they read 0. */
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_keep(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_zero(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_zero_f64(double v) {
  const int lo = dpp_zero<CTRL, ROW_MASK>(__double2loint(v));
  const int hi = dpp_zero<CTRL, ROW_MASK>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
#define CMI_DPP_ROW_SHR(n) (0x110 + (n))
#define CMI_DPP_WAVE_SHL1 0x130
#define CMI_DPP_WAVE_SHR1 0x138
#define CMI_DPP_ROW_BCAST15 0x142
#define CMI_DPP_ROW_BCAST31 0x143

/* Sum `v` over runs of consecutive lanes that hold the same `key`: segmented
 * inclusive scan in the 6 DPP rounds of a wave64 scan (row_shr 1, 2, 4, 8
 * inside each row of 16 lanes, then row_bcast15 into rows 1 and 3 and
 * row_bcast31 into rows 2 and 3). `stop` of a lane says "a run starts inside
 * the window I have summed so far"; a lane adds the incoming partial sum only
 * while it is clear - as v = fma(stop ? 0 : 1, incoming, v), which is the
 * exactly rounded sum or v itself. Lanes a round has no source for read
 * zeros, which change neither v nor stop. On return the LAST lane of every run
 * holds the run's total and is flagged in `tail`. Must be called by all 64
 * lanes. */
template <int N, int ROUNDS>
__device__ __forceinline__ void run_sums(int32_t key, double (&v)[N],
                                         bool &tail) {
  constexpr int rounds = ROUNDS;
  /* lane 0 / lane 63 have no neighbour: they keep ~key, which differs */
  const int32_t prev = dpp_keep<CMI_DPP_WAVE_SHR1, 0xf>(~key, key);
  const int32_t next = dpp_keep<CMI_DPP_WAVE_SHL1, 0xf>(~key, key);
  const int lane = threadIdx.x & 63;
  /* with fewer than 6 rounds the sums stop at groups of 2^rounds lanes: a run
   * that crosses a group boundary ends in one tail per group it touches */
  const int group_mask = (1 << rounds) - 1;
  int stop = ((key != prev) || (lane & group_mask) == 0) ? 1 : 0;
  tail = (key != next) || (lane & group_mask) == group_mask;
#define CMI_SCAN_ROUND(CTRL, ROW_MASK)                                         \
  {                                                                            \
    const double take = stop ? 0. : 1.;                                        \
    _Pragma("unroll") for (int k = 0; k < N; ++k) v[k] =                       \
        __fma_rn(take, dpp_zero_f64<CTRL, ROW_MASK>(v[k]), v[k]);              \
    stop |= dpp_zero<CTRL, ROW_MASK>(stop);                                    \
  }
  CMI_SCAN_ROUND(CMI_DPP_ROW_SHR(1), 0xf)
  if (rounds > 1)
    CMI_SCAN_ROUND(CMI_DPP_ROW_SHR(2), 0xf)
  if (rounds > 2)
    CMI_SCAN_ROUND(CMI_DPP_ROW_SHR(4), 0xf)
  if (rounds > 3)
    CMI_SCAN_ROUND(CMI_DPP_ROW_SHR(8), 0xf)
  if (rounds > 4)
    CMI_SCAN_ROUND(CMI_DPP_ROW_BCAST15, 0xa)
  if (rounds > 5)
    CMI_SCAN_ROUND(CMI_DPP_ROW_BCAST31, 0xc)
#undef CMI_SCAN_ROUND
}

/* The same sums over runs of equal keys inside every group of 4 lanes, without
 * the rounds of a scan: a lane reads the keys and values of the (up to) three
 * lanes below it in its group with quad_perm DPP moves and adds those that
 * belong to its run.
 * The LAST lane of every run (inside its group) holds the run's total and is
 * flagged in `tail`. Must be called by all 64 lanes. */
template <int CTRL> __device__ __forceinline__ int quad_read_i32(int v) {
  /* (every lane of a group has a source lane: no "old" value to keep, so no
   * copy of one in front of the move) */
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ double quad_read_f64(double v) {
  const int lo = quad_read_i32<CTRL>(__double2loint(v));
  const int hi = quad_read_i32<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ void quad_run_sums(int32_t key, double (&v)[N],
                                              bool &tail) {
  const int q = threadIdx.x & 3;
  const int32_t k1 = quad_read_i32<CMI_DPP_QUAD_SHR1>(key);
  const int32_t k2 = quad_read_i32<CMI_DPP_QUAD_SHR2>(key);
  const int32_t k3 = quad_read_i32<CMI_DPP_QUAD_SHR3>(key);
  const int32_t kn = quad_read_i32<CMI_DPP_QUAD_SHL1>(key);
  double v1[N], v2[N], v3[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    v1[k] = quad_read_f64<CMI_DPP_QUAD_SHR1>(v[k]);
    v2[k] = quad_read_f64<CMI_DPP_QUAD_SHR2>(v[k]);
    v3[k] = quad_read_f64<CMI_DPP_QUAD_SHR3>(v[k]);
  }
  const bool e1 = (q >= 1) && key == k1;
  const bool e2 = e1 && (q >= 2) && key == k2;
  const bool e3 = e2 && (q == 3) && key == k3;
  tail = (q == 3) || key != kn;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    /* (selects: a value that is not part of the run adds as +0.) */
    v[k] += e1 ? v1[k] : 0.;
    v[k] += e2 ? v2[k] : 0.;
    v[k] += e3 ? v3[k] : 0.;
  }
}

/* run_sums() for groups of at most 16 lanes with the bookkeeping in scalar
 * registers: which lanes start a run (or a group) is a 64-bit mask, "a run
 * starts inside the window summed so far" after a round of distance d is
 * stops | stops << d - two scalar instructions instead of a DPP move, an OR
 * and a compare per lane - and a lane takes the partial sum d lanes below it
 * through a select between zero and the DPP-moved value (v_cndmask_b32_dpp
 * under the mask in VCC) and a plain add: three vector instructions per value
 * and round. (A group's first 2^r - 1 lanes have their stop bit set by round
 * r whatever the neighbouring group's bits shifted in say.) The LAST lane of
 * every run holds its total; `tails` is the mask of those lanes. */
#define CMI_DPP_ROW_SHL(n) (0x100 + (n))
/* stops ? 0 : (v of the lane D below, 0 where the row has none): one
 * v_cndmask_b32_dpp per half under the mask in VCC. (Inline assembly: the
 * compiler turns the select into moves under the execution mask. Two wait
 * states between the vector instruction that wrote v and a DPP read of it:
 * the s_mov and the s_nop.) */
#define CMI_TAKE_UNLESS(D)                                                     \
  template <> __device__ __forceinline__ double take_unless<D>(               \
      unsigned long long stops, double v, int zero) {                          \
    int lo, hi;                                                                \
    asm("s_mov_b64 vcc, %4\n\ts_nop 0\n\t"                                    \
        "v_cndmask_b32_dpp %0, %2, %5, vcc row_shr:" #D                        \
        " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                         \
        "v_cndmask_b32_dpp %1, %3, %5, vcc row_shr:" #D                        \
        " row_mask:0xf bank_mask:0xf bound_ctrl:1"                             \
        : "=&v"(lo), "=&v"(hi)                                                 \
        : "v"(__double2loint(v)), "v"(__double2hiint(v)), "s"(stops),          \
          "v"(zero)                                                            \
        : "vcc");                                                              \
    return __hiloint2double(hi, lo);                                           \
  }
template <int D>
__device__ __forceinline__ double take_unless(unsigned long long stops,
                                              double v, int zero);
CMI_TAKE_UNLESS(1)
CMI_TAKE_UNLESS(2)
CMI_TAKE_UNLESS(4)
CMI_TAKE_UNLESS(8)
#undef CMI_TAKE_UNLESS
/* (`zero`: a vector register that holds 0, kept by the caller across its
 * loop) */
template <int N, int ROUNDS>
__device__ __forceinline__ void run_sums_masked(int32_t key, double (&v)[N],
                                                unsigned long long &tails,
                                                int zero) {
  static_assert(ROUNDS >= 1 && ROUNDS <= 4, "groups of 2 to 16 lanes");
  /* first / last lanes of the groups of 2^ROUNDS lanes */
  constexpr unsigned long long first =
      ROUNDS == 1 ? 0x5555555555555555ull
                  : ROUNDS == 2 ? 0x1111111111111111ull
                                : ROUNDS == 3 ? 0x0101010101010101ull
                                              : 0x0001000100010001ull;
  constexpr unsigned long long last = first << ((1 << ROUNDS) - 1);
  /* (lanes at the ends of a row of 16 read 0: they start / end a group) */
  const int32_t prev = dpp_zero<CMI_DPP_ROW_SHR(1), 0xf>(key);
  const int32_t next = dpp_zero<CMI_DPP_ROW_SHL(1), 0xf>(key);
  unsigned long long stops = mask_ne(key, prev) | first;
  tails = mask_ne(key, next) | last;
#define CMI_MASKED_ROUND(D)                                                    \
  {                                                                            \
    _Pragma("unroll") for (int k = 0; k < N; ++k) v[k] +=                      \
        take_unless<D>(stops, v[k], zero);                                     \
    stops |= stops << (D);                                                     \
  }
  CMI_MASKED_ROUND(1)
  if (ROUNDS > 1)
    CMI_MASKED_ROUND(2)
  if (ROUNDS > 2)
    CMI_MASKED_ROUND(4)
  if (ROUNDS > 3)
    CMI_MASKED_ROUND(8)
#undef CMI_MASKED_ROUND
}

/* FULL mode (all 14 ions + 2 heating terms per step): update_integrals as a
 * cooperative, transposed accumulation. A lane's 16 accumulation weights
 * (sigma_ion, sigma (nu - nu_0)) are constants of its packet; they are written
 * to LDS once per flight, [lane][16], and every lane then keeps the TRANSPOSED
 * weights of its quarter of the wave in registers: lane (q, i) holds
 * weight[16 q + r][i], r = 0..15 - accumulator i of the 16 packets of quarter
 * q. Per step only ds * w and the destination change, and those travel by DPP
 * row broadcasts (row_newbcast: lane r of each row of 16 to the whole row):
 * each quarter - 16 lanes, one per accumulator - walks its own 16 packets in a
 * fixed, fully unrolled loop, keeps running sums along runs of equal
 * destinations and issues one contiguous 128-B group of adds where a run ends:
 * ds_add_f64 into the block's combining table or global atomics into the AoS
 * accumulator row of the cell (two memory-side 64-B requests - float atomics
 * execute at the memory side in 64-B requests and their count, not the bytes,
 * bounds the kernel without the table). No LDS traffic in the walk at all. */
#define CMI_DEST_SAME INT32_MIN
/* packets per batch of the walk (divides 16) */
#ifndef CMI_WALK_PART
#define CMI_WALK_PART 4
#endif
struct FullStage {
  double weight[64][CMI_NACC];
};

#define CMI_DPP_ROW_NEWBCAST(n) (0x150 + (n))
template <int R> __device__ __forceinline__ int row_bcast_i32(int v) {
  /* every lane has a source lane: bound_ctrl lets the compiler drop the
   * initialisation of the destination */
  return __builtin_amdgcn_update_dpp(0, v, CMI_DPP_ROW_NEWBCAST(R), 0xf, 0xf,
                                     true);
}
template <int R> __device__ __forceinline__ double row_bcast_f64(double v) {
  /* ONE v_mov_b64_dpp: row_newbcast is the DPP control gfx90a and later
   * accept on 64-bit operands */
  return __builtin_amdgcn_update_dpp(0., v, CMI_DPP_ROW_NEWBCAST(R), 0xf, 0xf,
                                     true);
}

/* the transposed weights of the lane's quarter, after (re)launches */
__device__ __forceinline__ void load_quarter_weights(const FullStage &st,
                                                     double (&wq)[CMI_NACC]) {
  const int lane = threadIdx.x & 63;
  asm volatile("" ::: "memory"); /* written by other lanes: re-read */
#pragma unroll
  for (int r = 0; r < 16; ++r)
    wq[r] = st.weight[(lane & 48) + r][lane & 15];
}

template <bool HEAT, int PART>
__device__ __forceinline__ void
walk_part(const ShootArgs &a, const double (&wq)[CMI_NACC], int32_t dest,
          double dsw, int32_t &carry, bool mine, double *acc_i, double *table_i,
          unsigned int &natomics) {
  constexpr int N = CMI_WALK_PART;
  int32_t d[N + 1];
  double sum[N];
  /* destination and ds * w of the part's packets, from their lanes */
#define CMI_WALK_FETCH(r)                                                      \
  d[r] = row_bcast_i32<N * PART + (r)>(dest);                                  \
  sum[r] = row_bcast_f64<N * PART + (r)>(dsw) * wq[N * PART + (r)];
  CMI_WALK_FETCH(0)
  CMI_WALK_FETCH(1)
  CMI_WALK_FETCH(2)
  CMI_WALK_FETCH(3)
#if CMI_WALK_PART > 4
  CMI_WALK_FETCH(4)
  CMI_WALK_FETCH(5)
  CMI_WALK_FETCH(6)
  CMI_WALK_FETCH(7)
#endif
#if CMI_WALK_PART > 8
  CMI_WALK_FETCH(8)
  CMI_WALK_FETCH(9)
  CMI_WALK_FETCH(10)
  CMI_WALK_FETCH(11)
  CMI_WALK_FETCH(12)
  CMI_WALK_FETCH(13)
  CMI_WALK_FETCH(14)
  CMI_WALK_FETCH(15)
#endif
#undef CMI_WALK_FETCH
  if (d[0] == CMI_DEST_SAME)
    d[0] = carry;
#pragma unroll
  for (int r = 1; r < N; ++r)
    d[r] = (d[r] == CMI_DEST_SAME) ? d[r - 1] : d[r];
  carry = d[N - 1];
  d[N] = -1; /* a run that goes on in the next part is added in two pieces */
  /* running sums along each run, in registers ... */
#pragma unroll
  for (int r = 1; r < N; ++r)
    sum[r] = __fma_rn((d[r] == d[r - 1]) ? 1. : 0., sum[r - 1], sum[r]);
  /* ... and one add where a run ends */
#pragma unroll
  for (int r = 0; r < N; ++r) {
    if (d[r] != d[r + 1] && d[r] != -1 && mine && sum[r] != 0. &&
        CMI_EXP(a) != 3) { /* 3 = experiment: walk without the adds */
      if (d[r] >= 0) {
        atomicAdd(table_i + d[r] * CMI_NACC, sum[r]); /* ds_add_f64 */
      } else {
        atomic_add_f64(acc_i + ((int64_t)(-(d[r] + 2)) << 4), sum[r]);
        ++natomics;
      }
    }
  }
}

/* acc += (t of lane R of the row) x w: ONE v_fmac_f64_dpp - row_newbcast is
 * the DPP control the double-precision ALU accepts (the compiler emits a
 * v_mov_b64_dpp and a multiply instead) */
template <int R>
__device__ __forceinline__ void fmac_row_bcast(double &acc, double t,
                                               double w);
#define CMI_FMAC_ROW_BCAST(R)                                                  \
  template <>                                                                  \
  __device__ __forceinline__ void fmac_row_bcast<R>(double &acc, double t,     \
                                                    double w) {                \
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #R                           \
        " row_mask:0xf bank_mask:0xf"                                          \
        : "+v"(acc)                                                            \
        : "v"(t), "v"(w));                                                     \
  }
CMI_FMAC_ROW_BCAST(0)
CMI_FMAC_ROW_BCAST(1)
CMI_FMAC_ROW_BCAST(2)
CMI_FMAC_ROW_BCAST(3)
CMI_FMAC_ROW_BCAST(4)
CMI_FMAC_ROW_BCAST(5)
CMI_FMAC_ROW_BCAST(6)
CMI_FMAC_ROW_BCAST(7)
CMI_FMAC_ROW_BCAST(8)
CMI_FMAC_ROW_BCAST(9)
CMI_FMAC_ROW_BCAST(10)
CMI_FMAC_ROW_BCAST(11)
CMI_FMAC_ROW_BCAST(12)
CMI_FMAC_ROW_BCAST(13)
CMI_FMAC_ROW_BCAST(14)
CMI_FMAC_ROW_BCAST(15)
#undef CMI_FMAC_ROW_BCAST

/* Cells of a wave's step that are summed in registers before the table sees
 * them (accumulate_full, table mode): the lanes of a wave follow neighbouring
 * rays of one bundle and sit in one or two cells at most steps */
#ifndef CMI_GROUP_PASSES
#define CMI_GROUP_PASSES 3
#endif

/* Table mode: every row adds its term to its slot straight away - 16
 * ds_add_f64 per lane and step, no running sums, no branches: the LDS unit
 * merges what the running sums would have merged. Lanes without a slot add
 * (zero, or a term that is also added elsewhere - see below) to a dummy row
 * after the table that is never written back. */
/* `row_offset` = the BYTE offset of the lane's slot row in the table (the
 * dummy row for lanes without a slot), so that a row costs three vector
 * instructions and the add: v_add_u32_dpp (address), v_mov_b64_dpp (path
 * length), v_mul_f64, ds_add_f64 */
template <int R>
__device__ __forceinline__ void table_row(const double (&wq)[CMI_NACC],
                                          int32_t row_offset, double dsw,
                                          double *table_i) {
  const int32_t offset = row_bcast_i32<R>(row_offset);
  const double term = row_bcast_f64<R>(dsw) * wq[R];
  /* (skipping the adds of zero - most cross sections of a photon are zero -
   * was measured, twice: the exec masking costs the first generation more
   * than the LDS unit gains, +18 ms before and +3 ms after the emission
   * physics left this kernel) */
  atomicAdd(reinterpret_cast<double *>(reinterpret_cast<char *>(table_i) +
                                       offset),
            term); /* ds_add_f64 */
}

/* Sum of x over the four quarters of the wave (lanes i, 16 + i, 32 + i, 48 + i),
 * in every lane: gfx950's lane swaps - v_permlane32_swap exchanges the upper
 * half of one register with the lower half of another, v_permlane16_swap the
 * odd rows of one with the even rows of another; with both registers holding
 * x, their sum afterwards is x + (x of the other half / other row). */
__device__ __forceinline__ double fold_quarters(double x) {
  int alo = __double2loint(x), ahi = __double2hiint(x);
  int blo = alo, bhi = ahi;
  asm volatile("s_nop 1\n\t"
               "v_permlane32_swap_b32 %0, %2\n\t"
               "v_permlane32_swap_b32 %1, %3"
               : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi));
  x = __hiloint2double(ahi, alo) + __hiloint2double(bhi, blo);
  alo = __double2loint(x);
  ahi = __double2hiint(x);
  blo = alo;
  bhi = ahi;
  asm volatile("s_nop 1\n\t"
               "v_permlane16_swap_b32 %0, %2\n\t"
               "v_permlane16_swap_b32 %1, %3"
               : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi));
  return __hiloint2double(ahi, alo) + __hiloint2double(bhi, blo);
}

/* FULL mode, first generation (round 4): update_integrals cell by cell, with
 * no table in between. The lanes of a wave follow neighbouring rays of one
 * bundle and sit in one or two cells at most steps. Take the cell of the
 * first lane that still has something to add; sum the contributions of all
 * lanes in that cell in registers - lane (q, i) adds up accumulator i of the
 * 16 packets of its quarter: 16 v_fmac_f64_dpp with the path length as a row
 * broadcast, in four chains -, fold the four quarters, and let 16 lanes add
 * the cell's row with ONE contiguous 128-B group of global atomics (two
 * memory-side requests; values that are zero - most cross sections of a soft
 * photon - are not sent); then the next cell, until no lane is left.
 *
 * What this replaced: a combining table in LDS into which every lane put its
 * 16 terms with 16 ds_add_f64 per step. The LDS unit charges for the
 * INSTRUCTION - 8 cycles for a ds_add_f64 of 64 lanes to 64 addresses, 11
 * with four lanes per address, 6 with 16 lanes active
 * (tools/microbench/lds_atomic.hip) - and 16 of them per step were 170 of
 * the ~300 cycles a CU spent per wave step; the table's slots cost every
 * step an LDS round trip (compare-and-swap) in a kernel that is bound by the
 * latency of its dependent chain at 3 waves per SIMD, and its write-backs a
 * barrier of the block every 8 steps - for about as many memory-side atomic
 * requests as the cell sums of the waves need by themselves. Without the
 * table the block needs 17 KB less LDS and no barriers. */
template <bool HEAT>
__device__ __forceinline__ void
accumulate_full_grouped(const ShootArgs &a, const double (&wq)[CMI_NACC],
                        unsigned long long remaining, int32_t cell,
                        double dsw, unsigned int &natomics) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15;
  /* (lane i: COLUMN i of the row - threshold order, cmi_acc_column - the
   * transposed weights are staged in that order) */
  const bool writer =
      lane < 16 && (HEAT || ((CMI_ACC_COLUMNS_OF_IONS >> i) & 1u) != 0u);
  /* column i of cell c: AoS rows of CMI_NACC doubles */
  double *const acc_i = a.cells.acc_base + i;
  while (remaining != 0ull) {
    const int lead = __ffsll((long long)remaining) - 1;
    const int32_t c = __builtin_amdgcn_readlane(cell, lead);
    const unsigned long long in_cell = remaining & mask_eq(cell, c);
    remaining &= ~in_cell;
    if (CMI_EXP(a) == 2 || CMI_EXP(a) >= 4) /* exp.: no sums */
      continue;
    const double t = lanes_of(in_cell) ? dsw : 0.;
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0.; /* four chains of four */
    /* (two wait states between the select that wrote t and a DPP read) */
    asm volatile("s_nop 1" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(t));
    fmac_row_bcast<0>(s0, t, wq[0]);
    fmac_row_bcast<1>(s1, t, wq[1]);
    fmac_row_bcast<2>(s2, t, wq[2]);
    fmac_row_bcast<3>(s3, t, wq[3]);
    fmac_row_bcast<4>(s0, t, wq[4]);
    fmac_row_bcast<5>(s1, t, wq[5]);
    fmac_row_bcast<6>(s2, t, wq[6]);
    fmac_row_bcast<7>(s3, t, wq[7]);
    fmac_row_bcast<8>(s0, t, wq[8]);
    fmac_row_bcast<9>(s1, t, wq[9]);
    fmac_row_bcast<10>(s2, t, wq[10]);
    fmac_row_bcast<11>(s3, t, wq[11]);
    fmac_row_bcast<12>(s0, t, wq[12]);
    fmac_row_bcast<13>(s1, t, wq[13]);
    fmac_row_bcast<14>(s2, t, wq[14]);
    fmac_row_bcast<15>(s3, t, wq[15]);
    const double sum = fold_quarters((s0 + s1) + (s2 + s3));
    if (writer && sum != 0. && CMI_EXP(a) != 3) { /* 3: no adds */
      atomic_add_f64(acc_i + ((int64_t)c << 4), sum);
      ++natomics;
    }
  }
}

template <bool HEAT>
__device__ __forceinline__ void
accumulate_full(const ShootArgs &a, const double (&wq)[CMI_NACC],
                bool accumulate, int32_t cell, double dsw,
                unsigned int &natomics, int32_t *table_tag,
                double *table_val) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15;
  /* lane i works on COLUMN i of the cells' rows (threshold order,
   * cmi_acc_column): the transposed weights are staged in that order */
  const bool mine = HEAT || ((CMI_ACC_COLUMNS_OF_IONS >> i) & 1u) != 0u;
  /* column i of cell c: AoS rows of CMI_NACC doubles */
  double *const acc_i = a.cells.acc_base + i;
  double *const table_i = table_val + i;
  double term = accumulate ? dsw : 0.;

  if (table_tag != nullptr) {
    /* Round 4: cell by cell instead of packet by packet. What the LDS unit
     * charges for is the INSTRUCTION - 8 cycles for a ds_add_f64 of 64 lanes
     * to 64 addresses, 11 with four lanes per address, 6 with 16 lanes
     * active (tools/microbench/lds_atomic.hip) - and 16 of them per step
     * were 170 of the ~300 cycles a CU spent per wave step. The lanes of a
     * wave sit in one or two cells at most steps: take the cell of the first
     * lane that still has something to add (its slot claimed by the heads of
     * the step's runs - a few lanes' compare-and-swap instead of 64 on one
     * address), sum the contributions of all lanes in that cell in registers - lane
     * (q, i) adds up accumulator i of the 16 packets of its quarter, 16
     * v_fmac_f64_dpp with the path length as a row broadcast - and add the
     * four quarters' sums with ONE ds_add_f64. Lanes in further cells
     * (more than CMI_GROUP_PASSES distinct ones) go the old way below. */
    unsigned long long remaining =
        CMI_EXP(a) == 4 ? 0ull : wave_ballot(accumulate);
    /* The slots first, for all cells of the step at once: the first lane of
     * every run of lanes in one cell - a handful of lanes - looks its cell's
     * slot up (compare-and-swap with linear probing, as in the hydrogen-only
     * table); every cell's lowest lane is such a head. One LDS round trip
     * per step, whatever the number of cells (claiming inside the loop
     * below, cell after cell, cost 36 ms of a 114 ms launch in latency). */
    const int32_t run_key = accumulate ? cell : ~lane;
    const int32_t run_prev =
        dpp_keep<CMI_DPP_WAVE_SHR1, 0xf>(~run_key, run_key);
    const unsigned long long heads = remaining & mask_ne(run_key, run_prev);
    int32_t head_slot = -1;
    if (lanes_of(heads)) {
      uint32_t s = ((uint32_t)cell * 0x9E3779B1u) >> (32 - CMI_FTABLE_BITS);
      for (int probe = 0; probe < CMI_TABLE_PROBES; ++probe) {
        const int32_t was = atomicCAS(&table_tag[s], -1, cell);
        if (was == -1 || was == cell) {
          head_slot = (int32_t)s;
          break;
        }
        s = (s + 1) & (CMI_FTABLE_SLOTS - 1);
      }
    }
    unsigned long long unplaced = 0ull; /* lanes whose cell found no slot */
    for (int pass = 0; pass < CMI_GROUP_PASSES && remaining != 0ull; ++pass) {
      const int lead = __ffsll((long long)remaining) - 1;
      const int32_t c = __builtin_amdgcn_readlane(cell, lead);
      const int32_t slot = __builtin_amdgcn_readlane(head_slot, lead);
      const unsigned long long in_cell = remaining & mask_eq(cell, c);
      remaining &= ~in_cell;
      if (slot < 0) {
        unplaced |= in_cell;
        continue;
      }
      if (CMI_EXP(a) == 2 || CMI_EXP(a) >= 4) /* exp.: no sums */
        continue;
      const double t = lanes_of(in_cell) ? dsw : 0.;
      double s0 = 0., s1 = 0., s2 = 0., s3 = 0.; /* four chains of four */
      /* (two wait states between the select that wrote t and a DPP read) */
      asm volatile("s_nop 1" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(t));
      fmac_row_bcast<0>(s0, t, wq[0]);
      fmac_row_bcast<1>(s1, t, wq[1]);
      fmac_row_bcast<2>(s2, t, wq[2]);
      fmac_row_bcast<3>(s3, t, wq[3]);
      fmac_row_bcast<4>(s0, t, wq[4]);
      fmac_row_bcast<5>(s1, t, wq[5]);
      fmac_row_bcast<6>(s2, t, wq[6]);
      fmac_row_bcast<7>(s3, t, wq[7]);
      fmac_row_bcast<8>(s0, t, wq[8]);
      fmac_row_bcast<9>(s1, t, wq[9]);
      fmac_row_bcast<10>(s2, t, wq[10]);
      fmac_row_bcast<11>(s3, t, wq[11]);
      fmac_row_bcast<12>(s0, t, wq[12]);
      fmac_row_bcast<13>(s1, t, wq[13]);
      fmac_row_bcast<14>(s2, t, wq[14]);
      fmac_row_bcast<15>(s3, t, wq[15]);
      if (CMI_EXP(a) != 3) /* 3 = experiment: sums without the adds */
        atomicAdd(table_i + slot * CMI_NACC,
                  (s0 + s1) + (s2 + s3)); /* ds_add_f64 */
    }
    /* what is left: lanes in a fourth, fifth ... cell of this step, and
     * lanes whose cell found no free slot */
    remaining |= unplaced;
    if (remaining == 0ull)
      return;
    accumulate = lanes_of(remaining);
    /* every such lane claims (or finds) the slot of its cell, as in the
     * hydrogen-only table. dest >= 0: slot; -1: nothing to add;
     * other negatives: cell -(dest + 2), no slot was free */
    term = accumulate ? dsw : 0.;
    int32_t dest = -1;
    if (accumulate) {
      dest = -(cell + 2);
      if (CMI_EXP(a) != 4) {
        uint32_t s = ((uint32_t)cell * 0x9E3779B1u) >> (32 - CMI_FTABLE_BITS);
        for (int probe = 0; probe < CMI_TABLE_PROBES; ++probe) {
          const int32_t was = atomicCAS(&table_tag[s], -1, cell);
          if (was == -1 || was == cell) {
            dest = (int32_t)s;
            break;
          }
          s = (s + 1) & (CMI_FTABLE_SLOTS - 1);
        }
      }
    }
    if (CMI_EXP(a) == 2 || CMI_EXP(a) >= 4) /* exp.: no walk */
      return;
    const int32_t row_offset =
        (dest >= 0 ? dest : CMI_FTABLE_SLOTS) *
        (int32_t)(CMI_NACC * sizeof(double));
    table_row<0>(wq, row_offset, term, table_i);
    table_row<1>(wq, row_offset, term, table_i);
    table_row<2>(wq, row_offset, term, table_i);
    table_row<3>(wq, row_offset, term, table_i);
    if (CMI_EXP(a) == 6) /* experiment: a quarter of the adds */
      return;
    table_row<4>(wq, row_offset, term, table_i);
    table_row<5>(wq, row_offset, term, table_i);
    table_row<6>(wq, row_offset, term, table_i);
    table_row<7>(wq, row_offset, term, table_i);
    table_row<8>(wq, row_offset, term, table_i);
    table_row<9>(wq, row_offset, term, table_i);
    table_row<10>(wq, row_offset, term, table_i);
    table_row<11>(wq, row_offset, term, table_i);
    table_row<12>(wq, row_offset, term, table_i);
    table_row<13>(wq, row_offset, term, table_i);
    table_row<14>(wq, row_offset, term, table_i);
    table_row<15>(wq, row_offset, term, table_i);
    /* rare: packets that found no slot go the general way below, alone */
    const bool direct = dest < -1;
    if (__ballot(direct) == 0ull)
      return;
    int32_t carry = -1;
    const int32_t only_direct = direct ? dest : -1;
    const double only_term = direct ? term : 0.;
    walk_part<HEAT, 0>(a, wq, only_direct, only_term, carry, mine, acc_i,
                       table_i, natomics);
#if CMI_WALK_PART < 16
    walk_part<HEAT, 1>(a, wq, only_direct, only_term, carry, mine, acc_i,
                       table_i, natomics);
#endif
#if CMI_WALK_PART < 8
    walk_part<HEAT, 2>(a, wq, only_direct, only_term, carry, mine, acc_i,
                       table_i, natomics);
    walk_part<HEAT, 3>(a, wq, only_direct, only_term, carry, mine, acc_i,
                       table_i, natomics);
#endif
    return;
  }

  /* no table: only the first lane of a run of equal cells (and of each
   * quarter) names the destination, the others post "same as the packet
   * before me", and the walk merges each run into one group of atomics */
  const int32_t key = accumulate ? cell : ~lane;
  const int32_t prev = dpp_keep<CMI_DPP_WAVE_SHR1, 0xf>(~key, key);
  const bool head = (key != prev) || (lane & 15) == 0;
  int32_t dest = -1;
  if (accumulate)
    dest = head ? -(cell + 2) : CMI_DEST_SAME;
  if (CMI_EXP(a) == 2 || CMI_EXP(a) >= 4) /* experiment: no walk */
    return;
  int32_t carry = -1; /* destination of the last packet of the part before */
  walk_part<HEAT, 0>(a, wq, dest, term, carry, mine, acc_i, table_i, natomics);
#if CMI_WALK_PART < 16
  walk_part<HEAT, 1>(a, wq, dest, term, carry, mine, acc_i, table_i, natomics);
#endif
#if CMI_WALK_PART < 8
  walk_part<HEAT, 2>(a, wq, dest, term, carry, mine, acc_i, table_i, natomics);
  walk_part<HEAT, 3>(a, wq, dest, term, carry, mine, acc_i, table_i, natomics);
#endif
}

/* Decomposed grids: does the block `g` fly the packet that emit_geometry has
 * just started (p.index = its cell in the block's coordinates)?
 *
 * Every block can decide for every packet of the iteration; a source outside
 * the whole grid is the business of the block at the grid's origin. A source
 * ON a cell wall - the benchmarks' star sits on the corner shared by the 8
 * octants - sends most of its packets through a first step of length zero
 * into a neighbouring cell: such steps are pure geometry (no opacity, no
 * optical depth), so every block takes them here, identically, before asking
 * whose packet it is (`skipped` = 1: p.index and p.tmax have advanced).
 * Otherwise the block with the source cell would emit everything and hand
 * 7/8 of it over. Several engines may hold the block (copies): each takes its
 * share of the ids. */
template <bool FULL, bool EXACT>
__device__ __forceinline__ bool
block_owns_start(const GridDev &g, Packet<FULL> &p, uint32_t packet_id,
                 int &skipped, bool &in_block) {
  skipped = 0;
  if (!EXACT) {
    /* (one such step at most: afterwards every tmax is positive) */
    const double tmin = min_f64(p.tmax[0], min_f64(p.tmax[1], p.tmax[2]));
    bool start_in_grid = true;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const int32_t gi = p.index[ax] + g.offset[ax];
      start_in_grid &= (gi >= 0 && gi < g.global_ncell[ax]);
    }
    if (start_in_grid && tmin == p.t) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        if (p.tmax[ax] == tmin) { /* exactly fast_step's advance */
          p.tmax[ax] = __fma_rn(1., p.tdelta[ax], p.tmax[ax]);
          p.index[ax] += (p.dir[ax] > 0.) ? 1 : -1;
        }
      }
      skipped = 1;
    }
  }
  bool in_grid = true;
  in_block = true;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const int32_t gi = p.index[ax] + g.offset[ax];
    in_block &= (p.index[ax] >= 0 && p.index[ax] < g.ncell[ax]);
    in_grid &= (gi >= 0 && gi < g.global_ncell[ax]);
  }
  const bool at_origin = (g.offset[0] | g.offset[1] | g.offset[2]) == 0;
  bool mine = in_block || (!in_grid && at_origin);
  /* several engines hold this block: each emits its share */
  if (g.copy_count > 1)
    mine &= (int32_t)(packet_id % (uint32_t)g.copy_count) == g.copy_rank;
  return mine;
}

/*
 * Transport kernel: IonizationPhotonShootJob::execute
 * (src/IonizationPhotonShootJob.hpp:117-146) for a range of packets.
 *
 * One lane carries one packet at a time; lanes are persistent. A wave consumes
 * the launch's packet positions in chunks (chunk c = w, w + W, ... for wave w
 * of W), handing the next positions to its idle lanes once enough of them
 * wait - the emission code is long and divergent, so it should not run for a
 * lane or two at a time. The positions are mapped to packet ids through
 * `order`, which the host has sorted by emission direction: the lanes of a
 * wave then travel through the same cells, their loads coalesce and (with
 * a.aggregate) their contributions to the same cell are summed across the
 * wave before one lane issues the atomic (update_integrals,
 * src/DensityGrid.hpp:150-197: the reference does one locked
 * read-modify-write per packet and cell; sums are associative up to
 * rounding). EXACT selects the marcher (device_transport.h).
 */
/* TABLE: the launch is known to use the block combining table on a
 * non-periodic grid (the first generation of every benchmark config): those
 * choices are compile-time constants and the code of the other aggregation
 * modes and of the periodic wrap is not in the march loop at all. */
/* waves per SIMD of the multi-ion kernels without the cell-by-cell sums (the
 * pass kernels of the tail) */
#ifndef CMI_FULL_PASS_WAVES
#define CMI_FULL_PASS_WAVES 3
#endif
template <bool FULL, bool HEAT, bool REEMIT, bool EXACT, bool TABLE = false,
          bool PRE = false, bool PAD = false, bool TRACK = false,
          bool BIG = false>
__global__ void
    __launch_bounds__((shoot_block_threads<FULL, TABLE, BIG>()),
                      REEMIT ? 1
                             : (FULL ? (TABLE ? CMI_FULL_WAVES
                                              : CMI_FULL_PASS_WAVES)
                                    : ((PAD && !HEAT) ? CMI_PAD_WAVES
                                       : BIG          ? 4
                                                      : 6)))
        shoot_kernel(const ShootArgs a) {
  static_assert(!BIG || PAD, "BIG: the padded march on large grids");
  constexpr int BLOCK = shoot_block_threads<FULL, TABLE, BIG>();
  /* (hydrogen-only table: slots = 2^TBITS) */
  constexpr int TBITS =
      (TABLE && !FULL)
          ? (BIG ? CMI_TABLE_BLOCK_BIG_BITS : CMI_TABLE_BLOCK_BITS)
          : CMI_TABLE_BITS;
  constexpr int TSLOTS = 1 << TBITS;
  /* PAD: the hydrogen-only first generation on a whole, non-periodic grid,
   * marching through the padded records (ShootArgs::pad_H) */
  static_assert(!PAD || (TABLE && !FULL && !REEMIT && !EXACT && !PRE),
                "PAD is a specialisation of the hydrogen-only TABLE kernel");
  static_assert(!TRACK || (!TABLE && !EXACT && !PRE && !PAD),
                "TRACK is the tracker hook in the plain incremental marcher");
  const int lane = threadIdx.x & 63;
  const uint64_t lane_lt = (1ull << lane) - 1ull;
  /* Blocks are dealt round-robin over the 8 XCDs (blocks b and b + 8 share
   * one - speed only, nothing depends on it): with xcd_remap the blocks of an
   * XCD take neighbouring positions of the sorted packet order, so that the
   * ray bundles flying through the same cells share one L2 */
  uint32_t block = blockIdx.x;
  if (a.xcd_remap) {
    const uint32_t xcd = block & 7u, q = gridDim.x >> 3, r = gridDim.x & 7u;
    block = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) +
            (block >> 3);
  }
  const uint64_t wave = (uint64_t)block * (BLOCK / 64) +
                        (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * (BLOCK / 64);
  const uint64_t chunk = a.chunk;

  /* wave-uniform cursor into the launch's position range */
  uint64_t chunk_begin = wave * chunk;
  uint64_t pos = chunk_begin;
  uint64_t pos_end = chunk_begin + chunk < a.n_packets ? chunk_begin + chunk
                                                       : a.n_packets;
  if (pos > pos_end)
    pos = pos_end;

  Packet<FULL> p;
  /* Lanes without a packet take part in the cross-lane sums with a path
   * length of zero times their cross section / weights: those must be finite
   * (0 x NaN would poison the neighbours' sums), so nothing a sum multiplies
   * may start out as register or LDS garbage. */
  p.sigma_H = 0.;
  p.sigma_He_corr = 0.;
  p.nu = 0.;
  p.weight = 1.;
  PacketRng rng;
  /* (park_in_place launches: the packet's POSITION in the launch's order
   * instead - the id follows from it, id_of() below - so that parking at the
   * position costs no register across the march: a second value did cost the
   * hydrogen-only first generation, whose 64 are all taken, 0.3 of 34.6 ms,
   * and a build without the parking code, of all things, spilled the cross
   * section inside the march loop: 38.5 ms) */
  uint32_t packet_id = 0;
  auto id_of = [&](uint32_t held) -> uint32_t {
    return a.park_in_place
               ? (uint32_t)a.batch_offset + (a.order ? a.order[held] : held)
               : held;
  };
  uint32_t lane_meta = 0; /* rng position of the lane's packet (cmi_pack_meta) */
  bool active = false;
  int32_t last_cell = -1; /* EXACT marcher on grids >= 2^31 cells: see below */
  int64_t last_cell_wide = -1;

  /* packets finished per type (their weights are summed at the end:
   * src/IonizationPhotonShootJob.hpp:143-144); those that came from the
   * continuous source are counted a second time, per wave, in LDS */
  unsigned int tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0;
  __shared__ unsigned int s_continuous[BLOCK / 64][4];
  if (lane < 4)
    s_continuous[threadIdx.x >> 6][lane] = 0;
  unsigned int nsteps = 0, natomics = 0; /* per lane and launch: < 2^32 */
  unsigned int nwavesteps = 0;
  /* PAD: the steps of the whole wave (wave-uniform: scalar arithmetic), and
   * the record of the cell the lane's packet is about to cross - it lives
   * across refills of other lanes, and says "outside" by itself */
  unsigned long long nsteps_wave = 0;
  double pad_next = CMI_PAD_GHOST;
  const bool any_periodic =
      !TABLE &&
      (a.grid.periodic[0] | a.grid.periodic[1] | a.grid.periodic[2]) != 0;
  const int aggregate = TABLE ? CMI_AGG_BLOCK : a.aggregate;

  /* H-only: per-wave write-combining cache in LDS. Scattered fp64 atomics
   * execute at the memory side at a chip-wide rate of a few 1e10 per second
   * whatever the schedule (measured: the kernel's time follows the number of
   * atomics, not the number of steps), so contributions are first combined on
   * chip: a run total goes into a direct-mapped table of (cell, partial sum)
   * with LDS atomics; only an entry that is evicted by another cell - or
   * flushed when the wave ends - costs a global atomic. A wave follows one
   * bundle of neighbouring rays, so consecutive steps (and the next bundle of
   * the wave's chunk) keep hitting the same few cells. The table is private to
   * the wave: no barriers, LDS operations of a wave execute in order. */
  /* CMI_AGG_BLOCK: per-block combining table. The 4 waves of a block follow
   * neighbouring ray bundles (consecutive positions of the sorted order), which
   * cross the same cells within a few steps of each other - too far apart in
   * time for the small wave cache, and a cache shared between waves cannot
   * evict safely without locks. So the table is insert-only: a run total
   * claims a slot for its cell with an LDS compare-and-swap (linear probing,
   * CMI_TABLE_PROBES tries, then it falls back to a global atomic) and adds
   * with ds_add_f64; between two bundles the block meets at a barrier and
   * flushes every used slot with ONE global atomic per cell. */
  /* (the multi-ion first generation sums cell by cell in registers and has
   * no table: accumulate_full_grouped) */
  constexpr bool GROUPED = FULL && TABLE;
  constexpr int lds_slots =
      GROUPED ? 1 : (FULL ? CMI_FTABLE_SLOTS : TSLOTS);
  constexpr int lds_values = FULL ? CMI_NACC : (HEAT ? 2 : 1);
  __shared__ int32_t lds_tag[lds_slots];
  /* FULL: one more row, the sink of table_row() for lanes without a slot */
  __shared__ double lds_val[lds_values * (lds_slots + (FULL ? 1 : 0))];
  __shared__ int32_t block_has_work[BLOCK / 64];
  const int wib = threadIdx.x >> 6;
  /* FULL: per-wave accumulation weights and per-step scratch in LDS */
  __shared__ FullStage full_stage[FULL ? BLOCK / 64 : 1];
  FullStage &stage = full_stage[FULL ? wib : 0];
  double weights[CMI_NACC];
  double wq[CMI_NACC]; /* FULL: transposed weights of the lane's quarter */
  if (FULL) {
#pragma unroll
    for (int i = 0; i < CMI_NACC; ++i) {
      weights[i] = 0.;
      wq[i] = 0.;
      stage.weight[lane][i] = 0.;
    }
  }
  /* Multi-ion kernels use the same kind of table with 16 values per slot
   * (one accumulator row). 128 B per slot leave room for 256 slots only, so
   * the block writes it back every CMI_FTABLE_WINDOW iterations of the march
   * loop as well - the waves of a block advance together, one Manhattan shell
   * per iteration, so a cell's contributions arrive within a few iterations
   * of each other. */
  const bool use_table =
      !GROUPED && (TABLE || a.aggregate == CMI_AGG_BLOCK);
  if (use_table) {
    for (int k = threadIdx.x; k < lds_slots; k += BLOCK)
      lds_tag[k] = -1;
    for (int k = threadIdx.x; k < lds_values * lds_slots; k += BLOCK)
      lds_val[k] = 0.;
    __syncthreads();
  }
  /* A flush point: every wave of the block arrives, the table is written
   * back with one global atomic per value, and all learn whether any wave has
   * work left. Waves may arrive from different places in the code - the
   * points are interchangeable - but every wave must keep arriving until the
   * block is done: a wave never exits while others may wait at a barrier. */
  auto flush_point = [&](bool has_work) -> bool {
    if (lane == 0)
      block_has_work[wib] = has_work ? 1 : 0;
    __syncthreads();
    if (FULL) {
      const int i = threadIdx.x & 15;
      for (int k = threadIdx.x >> 4; k < lds_slots; k += BLOCK / 16) {
        const int32_t t = lds_tag[k];
        if (t >= 0) {
          /* (table rows are in column order, like the cells' rows) */
          if (HEAT || ((CMI_ACC_COLUMNS_OF_IONS >> i) & 1u) != 0u) {
            atomic_add_f64(a.cells.acc_base + (int64_t)t * CMI_NACC + i,
                           lds_val[k * CMI_NACC + i]);
            lds_val[k * CMI_NACC + i] = 0.;
            ++natomics;
          }
          if (i == 0)
            lds_tag[k] = -1; /* after the 16 lanes of the wave have read it */
        }
      }
    } else {
      for (int k = threadIdx.x; k < lds_slots; k += BLOCK) {
        const int32_t t = lds_tag[k];
        if (t >= 0) {
          const int32_t c = PAD ? cmi_unpad_cell(a, a.grid, t) : t;
          atomic_add_f64(acc_at(a.cells, ION_H_n, c), lds_val[k]);
          lds_val[k] = 0.;
          if (HEAT) {
            atomic_add_f64(acc_at(a.cells, CMI_NION, c),
                           lds_val[lds_slots + k]);
            lds_val[lds_slots + k] = 0.;
          }
          lds_tag[k] = -1;
          natomics += HEAT ? 2 : 1;
        }
      }
    }
    int any_work = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w)
      any_work |= block_has_work[w];
    __syncthreads();
    return any_work != 0;
  };
  /* add (v0[, v1]) to `cell` through the block table, for the lanes given as
   * a mask (called by all 64 lanes): the search for a slot is kept as that
   * mask - `looking` loses the lanes that found theirs - and the probe loop's
   * exit is a scalar test */
  auto table_add_masked = [&](unsigned long long looking, int32_t cell,
                              double v0, double v1) {
    uint32_t slot;
    if (PAD) {
      /* (a full-rate 24-bit multiply - the cells of a bundle differ in their
       * low bits; asm: the compiler widens __umul24 to the quarter-rate
       * v_mul_lo_u32) */
      uint32_t product;
#if CMI_PAD_HASH_LOW_BITS > 0
      /* the low bits of the slot are the low bits of the padded index - z + 2 y
       * + 4 x modulo 8 with an even padded extent: the cells of a 2 x 2 x 2
       * neighbourhood, where the lanes of a bundle sit at any one step, fall
       * into different LDS banks -, the rest is hashed */
      asm("v_mul_u32_u24 %0, 0x9e3779, %1"
          : "=v"(product)
          : "v"(cell >> CMI_PAD_HASH_LOW_BITS));
      slot = (((product >> (24 - TBITS + CMI_PAD_HASH_LOW_BITS))
               << CMI_PAD_HASH_LOW_BITS) |
              ((uint32_t)cell & ((1u << CMI_PAD_HASH_LOW_BITS) - 1u))) &
             (TSLOTS - 1);
#else
      asm("v_mul_u32_u24 %0, 0x9e3779, %1" : "=v"(product) : "v"(cell));
      slot = (product >> (24 - TBITS)) & (TSLOTS - 1);
#endif
    } else {
      slot = ((uint32_t)cell * 0x9E3779B1u) >> (32 - TBITS);
    }
    for (int probe = 0; probe < CMI_TABLE_PROBES; ++probe) {
      /* (lanes that are not looking: whatever the register holds, they are
       * masked out below) */
      int32_t was;
      asm volatile("" : "=v"(was));
      unsigned long long found;
      if (CMI_TABLE_READ_FIRST) {
        if (lanes_of(looking))
          was = *(volatile int32_t *)&lds_tag[slot];
        found = looking & mask_eq(was, cell);
        const unsigned long long vacant = looking & mask_eq(was, -1);
        if (vacant != 0ull) {
          int32_t got;
          asm volatile("" : "=v"(got));
          if (lanes_of(vacant))
            got = atomicCAS(&lds_tag[slot], -1, cell);
          found |= vacant & (mask_eq(got, -1) | mask_eq(got, cell));
        }
      } else {
        if (lanes_of(looking))
          was = atomicCAS(&lds_tag[slot], -1, cell);
        found = looking & (mask_eq(was, -1) | mask_eq(was, cell));
      }
      if (lanes_of(found)) {
        atomicAdd(&lds_val[slot], v0); /* ds_add_f64 */
        if (HEAT)
          atomicAdd(&lds_val[TSLOTS + slot], v1);
      }
      looking &= ~found;
      if (lanes_of(looking))
        slot = (slot + 1) & (TSLOTS - 1);
      if (looking == 0ull)
        return;
    }
    if (lanes_of(looking)) {
      const int32_t c = PAD ? cmi_unpad_cell(a, a.grid, cell) : cell;
      atomic_add_f64(acc_at(a.cells, ION_H_n, c), v0);
      if (HEAT)
        atomic_add_f64(acc_at(a.cells, CMI_NION, c), v1);
      natomics += HEAT ? 2 : 1;
    }
  };
  for (;;) {
    const unsigned long long active_mask = __ballot(active);
    const unsigned long long idle_mask = ~active_mask;
    if (pos == pos_end && chunk_begin + nwaves * chunk < a.n_packets) {
      /* current chunk used up: jump to this wave's next one */
      chunk_begin += nwaves * chunk;
      pos = chunk_begin;
      pos_end = chunk_begin + chunk < a.n_packets ? chunk_begin + chunk
                                                  : a.n_packets;
    }
    const uint64_t avail = pos_end - pos;
    if (use_table) {
      /* between two bundles */
      if (!flush_point(active_mask != 0ull || avail != 0))
        break;
    } else if (active_mask == 0ull && avail == 0) {
      break;
    }
    if (avail != 0 && idle_mask != 0ull &&
        (active_mask == 0ull || (int)__popcll(idle_mask) >= a.refill_threshold)) {
      const uint64_t rank = __popcll(idle_mask & lane_lt);
      if (!active && rank < avail) {
        const uint64_t i = pos + rank;
        bool mine = true;
        if (!PRE && a.xin != nullptr) {
          /* a flight handed over by another block of the grid */
#ifdef CMI_DBG_XIN_SLOTS
          const uint64_t row =
              (a.xin_local && a.xin_slots) ? (uint64_t)a.xin_slots[i] : i;
          const double *r = a.xin + (size_t)CMI_FLIGHT_DOUBLES * row;
#else
          const double *r = a.xin + (size_t)CMI_FLIGHT_DOUBLES * i;
#endif
          int64_t cell_global;
          unsigned long long idmeta;
          if (a.xin_local) {
            /* a slot of the tile rounds (tile_kernels.h) */
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
              p.pos[ax] = r[ax];
              p.dir[ax] = r[3 + ax];
              p.inv_dir[ax] = 1. / p.dir[ax];
              p.tmax[ax] = r[CMI_SLOT_TMAX + ax];
            }
            p.t = r[CMI_SLOT_T];
            p.tau = r[CMI_SLOT_TAU];
            p.nu = r[CMI_SLOT_NU];
            cell_global = (int64_t)(uint32_t)__double_as_longlong(
                r[CMI_SLOT_CELL]);
            idmeta = (unsigned long long)__double_as_longlong(
                r[CMI_SLOT_IDMETA]);
          } else {
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
              p.pos[ax] = r[ax];
              p.dir[ax] = r[3 + ax];
              p.inv_dir[ax] = 1. / p.dir[ax];
              p.tmax[ax] = r[7 + ax];
            }
            p.t = r[6];
            p.tau = r[10];
            p.nu = r[11];
            cell_global = __double_as_longlong(r[12]);
            idmeta = (unsigned long long)__double_as_longlong(r[13]);
          }
          packet_id = (uint32_t)idmeta;
          lane_meta = (uint32_t)(idmeta >> 32);
          if (REEMIT)
            rng.resume(a.seed, a.iteration, a.first_packet + packet_id,
                       lane_meta & 0xffffffu, (lane_meta >> 24) & 1u);
          p.type = (int32_t)(lane_meta >> 28);
          p.weight = a.model.photon_weight[cmi_meta_origin(lane_meta)];
          set_cross_sections(a.model, p, weights);
          if (!EXACT) {
            if (a.xin_local)
              resume_flight_local(a.grid, p, (int32_t)cell_global);
            else
              resume_flight(a.grid, p, cell_global);
          }
        } else if (!PRE && a.qin.id != nullptr) {
          /* a ready flight: re-emitted by the interaction kernel */
          packet_id = a.qin.id[i];
          lane_meta = a.qin.meta[i];
          if (REEMIT)
            rng.resume(a.seed, a.iteration, a.first_packet + packet_id,
                       lane_meta & 0xffffffu, (lane_meta >> 24) & 1u);
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            p.pos[ax] = a.qin.pos[ax][i];
            p.dir[ax] = a.qin.dir[ax][i];
            p.inv_dir[ax] = 1. / p.dir[ax];
          }
          p.type = (int32_t)(lane_meta >> 28);
          p.weight = a.model.photon_weight[cmi_meta_origin(lane_meta)];
          p.nu = a.qin.nu[i];
          p.tau = a.qin.tau[i];
          set_cross_sections(a.model, p, weights);
          start_flight<FULL, EXACT>(a.grid, p);
        } else {
          packet_id =
              (uint32_t)a.batch_offset + (a.order ? a.order[i] : (uint32_t)i);
          rng.init(a.seed, a.iteration, a.first_packet + packet_id);
          const uint32_t origin =
              emit_geometry<FULL, EXACT>(a.grid, a.model, rng, p);
          if (a.grid.decomposed) {
            /* a block flies the packets that start in it (block_owns_start;
             * the launch's positions may already be the block's own packets
             * only - block_select_kernel - and then all of them are) */
            int skipped = 0;
            bool in_block = true;
            mine = block_owns_start<FULL, EXACT>(a.grid, p, packet_id, skipped,
                                                 in_block);
            if (skipped != 0 && mine) {
              nsteps += skipped; /* they are DDA steps of the undivided run */
#pragma unroll
              for (int ax = 0; ax < 3; ++ax)
                p.rem[ax] = (p.dir[ax] > 0.)
                                ? a.grid.ncell[ax] - 1 - p.index[ax]
                                : p.index[ax];
              if (!in_block)
                p.rem[0] = -1;
              p.cell = (p.index[0] * a.grid.ncell[1] + p.index[1]) *
                           a.grid.ncell[2] +
                       p.index[2];
            }
          }
          if (PAD) {
            /* the padded long index and its strides (start_flight has set
             * index[], the signs and rem[0] < 0 for a start outside the box) */
            const int32_t sy = a.pad_nz, sx = a.pad_ny * a.pad_nz;
            p.cstep[0] = (p.dir[0] > 0.) ? sx : -sx;
            p.cstep[1] = (p.dir[1] > 0.) ? sy : -sy;
            p.cstep[2] = (p.dir[2] > 0.) ? 1 : -1;
            const bool outside = fast_outside(p);
            p.cell = outside ? 0
                             : ((p.index[0] + CMI_PAD_LAYERS) * a.pad_ny +
                                (p.index[1] + CMI_PAD_LAYERS)) *
                                       a.pad_nz +
                                   (p.index[2] + CMI_PAD_LAYERS);
            pad_next = outside ? CMI_PAD_GHOST : a.pad_H[p.cell];
          }
          if (mine) {
            if (PRE)
              emit_physics_from_row<FULL>(
                  a.model, rng, p, weights, origin,
                  a.pre_rows + (size_t)CMI_NACC *
                                   (packet_id - (uint32_t)a.batch_offset));
            else
              emit_physics<FULL>(a.model, rng, p, weights, origin);
            lane_meta = cmi_pack_meta(rng.block, rng.have, 0, origin);
          }
          if (a.park_in_place)
            packet_id = (uint32_t)i; /* the position; the id: id_of() */
        }
        if (FULL) {
#pragma unroll
          for (int i = 0; i < CMI_NACC; ++i)
            stage.weight[lane][i] = weights[cmi_acc_of_column(i)];
        }
        active = mine;
        last_cell = -1;
        last_cell_wide = -1;
      }
      const uint64_t taken = __popcll(idle_mask);
      pos += taken < avail ? taken : avail;
    }
    /* more positions left for this wave (in this or a later chunk)? */
    const uint64_t avail_after =
        (pos_end - pos) + (chunk_begin + nwaves * chunk < a.n_packets ? 1 : 0);

    /* ---- hot loop: only the march and the accumulation. It runs until so
     * few lanes are still in flight that the wave is due for a refill (or
     * none is); what happens to a packet at the end of its flight is decided
     * after the loop, for all finished lanes together. ---- */
    /* FAST: the record of the cell a lane is about to cross is loaded one
     * iteration ahead, so that the load overlaps the accumulation */
    int window = 0;
    if (FULL)
      load_quarter_weights(stage, wq);
    if constexpr (PAD) {
      /* The same loop on the padded records. What bounds it is instruction
       * issue - a SIMD issues one vector AND one scalar instruction per four
       * cycles, so the loop is as long as the larger of its two counts
       * (measured: trading vector selects for execution-mask arithmetic, 2
       * scalar instructions per vector one saved, gained nothing; neither did
       * a software pipeline with two record loads in flight, nor 8 waves per
       * SIMD - the loads are not what it waits for). So: no cell counters per
       * axis (the ghost record says "outside"), every tied axis advances by
       * selects, run sums over groups of 8 lanes, a 24-bit multiply for the
       * hash. */
      const double sigma = p.sigma_H;
      double wsig = p.weight * p.sigma_H;
      double hw = HEAT ? wsig * (p.nu - a.model.nu_H) : 0.;
      /* (kept in registers: recomputing them costs a multiplication a step) */
      asm volatile("" : "+v"(wsig), "+v"(hw));
      /* Round 4: every condition of the loop is a wave mask in scalar
       * registers (mask_gt, mask_eq, lanes_of): which lanes step, which axes
       * advance, where runs end, who still looks for a slot. The loop's exit
       * is then a scalar branch (the compiler no longer treats the loop as
       * divergent and keeps its counters in scalar registers), the run sums
       * need three vector instructions per round (run_sums_masked), and
       * vacuum is the record -0.: sigma x -0. leaves the optical depth alone
       * without a max(), its sign bit says "do not accumulate". */
      const unsigned long long active_lanes = wave_ballot(active);
      const int32_t not_lane = ~lane;
      /* (in vector registers: the scalar copy did not survive the register
       * pressure - it was spilled to lanes of a VGPR and read back, two
       * v_readlane and a wait, in every iteration) */
      const char *pad_base = reinterpret_cast<const char *>(a.pad_H);
      asm volatile("" : "+v"(pad_base));
      /* a refill is due once this many lanes are idle (never, if the wave
       * has no positions left) */
      const int idle_limit = __builtin_amdgcn_readfirstlane(
          avail_after != 0 ? a.refill_threshold : 65);
      int zero = 0;
      asm volatile("" : "+v"(zero)); /* one register for the whole loop */
      unsigned int bundle_steps = 0; /* wave-uniform: a scalar register */
      for (;;) {
        const unsigned long long flying =
            active_lanes & mask_gt(p.tau, 0.) & mask_gt(pad_next, -1.5);
        if (flying == 0ull || (int)__popcll(~flying) >= idle_limit)
          break;
        ++bundle_steps;
        nsteps_wave += (unsigned long long)__popcll(flying);
        const bool stepping = lanes_of(flying);
        /* number density > 0: the record's sign bit is clear */
        const unsigned long long accumulating =
            flying & mask_ge(__double2hiint(pad_next), 0);
        /* the run key - the cell about to be crossed, or something no other
         * lane has - before the march moves on (this IS the copy of the old
         * cell index the accumulation needs) */
        const int32_t key = lanes_of(accumulating) ? p.cell : not_lane;
        /* (lanes that do not step take part in the run sums with a key of
         * their own and are never a tail that adds: their path length may be
         * anything, run_sums_masked selects, it does not multiply) */
        double ds;
        asm volatile("" : "=v"(ds));
        if (stepping) {
          const double k = pad_next;
          const double tmin =
              min_f64(p.tmax[0], min_f64(p.tmax[1], p.tmax[2]));
          const double t_old = p.t;
          ds = tmin - t_old;
          const double sk = sigma * k;
          p.tau -= ds * sk;
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            /* every tied axis advances, under the execution mask: one add
             * each for the wall parameter and the cell (vector instructions
             * are what the loop is short of; the mask costs two scalar ones,
             * no branch round two instructions) */
            unsigned long long saved;
            asm volatile("s_and_saveexec_b64 %2, %3\n\t"
                         "v_add_f64 %0, %0, %4\n\t"
                         "v_add_u32 %1, %1, %5\n\t"
                         "s_mov_b64 exec, %2"
                         : "+v"(p.tmax[ax]), "+v"(p.cell), "=&s"(saved)
                         : "s"(mask_eq(p.tmax[ax], tmin)), "v"(p.tdelta[ax]),
                           "v"(p.cstep[ax])
                         : "scc");
          }
          p.t = tmin;
          if (!(p.tau > 0.)) {
            /* the flight ends in this cell (a cell with gas - vacuum leaves
             * the optical depth alone -, so the key is the cell) */
            last_cell = key;
            if (p.tau < 0.) {
              /* Scorr = ds tau / tau_cell = tau / (sigma n x_H): reciprocal,
               * one Newton step, the quotient and its correction (~1e-16) */
              double r = __builtin_amdgcn_rcp(sk);
              r = __fma_rn(__fma_rn(-sk, r, 1.), r, r);
              double corr = p.tau * r;
              corr = __fma_rn(__fma_rn(-corr, sk, p.tau), r, corr);
              ds += corr;
              p.t = t_old + ds;
            }
          }
        }
        /* (a GLOBAL load: through the laundered pointer the compiler knows
         * no address space and issues flat_load, which also counts as an LDS
         * operation - the wait for the table's compare-and-swap below would
         * then wait for this record as well, every iteration) */
        if (stepping && p.tau >= 0.)
          pad_next = *reinterpret_cast<
              const __attribute__((address_space(1))) double *>(
              (const __attribute__((address_space(1))) char *)pad_base +
              ((uint32_t)p.cell << 3));
        if (CMI_EXP(a) == 12) {
          /* experiment: the march alone (results are wrong) */
          asm volatile("" ::"v"(ds), "s"(accumulating));
          continue;
        }
        double v[2] = {ds * wsig, HEAT ? ds * hw : 0.};
        /* run sums over groups of 2^rounds lanes (measured at 8 waves/SIMD
         * in round 3, ms per iteration of 1e8 packets: groups of 4 44.2,
         * groups of 8 43.0, groups of 16 44.6) */
        unsigned long long tails;
        if (HEAT)
          run_sums_masked<2, CMI_PAD_SCAN_ROUNDS>(key, v, tails, zero);
        else
          run_sums_masked<1, CMI_PAD_SCAN_ROUNDS>(
              key, reinterpret_cast<double(&)[1]>(v), tails, zero);
        if (CMI_EXP(a) == 11) {
          /* experiment: no table (results are wrong) */
          asm volatile("" ::"v"(v[0]), "s"(tails));
          continue;
        }
        table_add_masked(tails & accumulating, key, v[0], v[1]);
      }
      nwavesteps += bundle_steps;
      /* the rest of the kernel reads the flight's end the usual way */
      if (active && !(p.tau > 0. && pad_next > -1.5)) {
        p.rem[0] = (p.tau >= 0. && !(pad_next > -1.5)) ? -1 : 0;
        p.rem[1] = 0;
        p.rem[2] = 0;
        if (last_cell >= 0)
          last_cell = cmi_unpad_cell(a, a.grid, last_cell);
      }
    }
    double2 kappa_next = make_double2(0., 0.);
    if (!PAD && !EXACT && active && p.tau > 0. && !fast_outside(p))
      kappa_next = fast_load_record(a.cells.opacity, p);
    /* (round 4: the lanes in flight as a scalar mask straight from the
     * compares, the refill test on scalars - see the PAD loop) */
    const unsigned long long active_lanes = wave_ballot(active);
    const int idle_limit = __builtin_amdgcn_readfirstlane(
        avail_after != 0 ? a.refill_threshold : 65);
    for (; !PAD;) {
      unsigned long long flying;
      if (EXACT)
        flying = wave_ballot(active && p.tau > 0. && is_inside(a.grid, p));
      else
        flying = active_lanes & mask_gt(p.tau, 0.) &
                 mask_ge(p.rem[0] | p.rem[1] | p.rem[2], 0);
      if (flying == 0ull || (int)__popcll(~flying) >= idle_limit)
        break;
      const bool stepping = lanes_of(flying);
      ++nwavesteps;
      double ds = 0.;
      bool accumulate = false;
      if (stepping) {
        double2 kappa;
        if (EXACT) {
          ds = dda_step(a.grid, a.cells.opacity, p, last_cell_wide, kappa);
          last_cell = (int32_t)last_cell_wide;
        } else {
          kappa = kappa_next;
          ds = fast_step(p, last_cell, kappa);
        }
        ++nsteps;
        accumulate = (kappa.x >= 0.); /* number density > 0 */
        /* DensityGrid::update_integrals' tracker hook,
         * src/DensityGrid.hpp:188-191 (SpectrumTracker::count_photon), and
         * DensitySubGrid::update_intensity_counters',
         * src/DensitySubGrid.hpp:614-617 (AbsorptionTracker::count_photon).
         * In the kernels of the exact marcher and in the TRACK builds of the
         * incremental one (blocks of a decomposed grid while trackers count:
         * in the other builds the hook's registers would spill the march
         * loop). */
        if ((EXACT || TRACK) && a.trackers.n != 0 && accumulate) {
          const int64_t tracked_cell =
              EXACT ? last_cell_wide : (int64_t)last_cell;
          for (int k = 0; k < a.trackers.n; ++k) {
            if (tracked_cell != a.trackers.cell[k])
              continue;
            if (a.trackers.kind[k] == CMI_TRACKER_ABSORPTION) {
              /* src/AbsorptionTracker.hpp:133-138: the step's contribution
               * to every mean intensity, by photon type */
              double *bins = a.trackers.absorption +
                             ((size_t)k * 4 + (size_t)p.type) * CMI_NION;
              const double dsw = ds * p.weight;
              if (FULL) {
                for (int ion = 0; ion < CMI_NION; ++ion)
                  atomic_add_f64(bins + ion,
                                 dsw * stage.weight[lane][cmi_acc_column(ion)]);
              } else {
                atomic_add_f64(bins + ION_H_n, dsw * p.sigma_H);
              }
              continue;
            }
            if (a.trackers.kind[k] == CMI_TRACKER_WEIGHTED) {
              /* src/WeightedSpectrumTracker.hpp:300-319: one over the area
               * the cell shows the packet, by photon type and frequency bin */
              const double direction[3] = {p.dir[0], p.dir[1], p.dir[2]};
              atomic_add_f64(
                  a.trackers.flux + 4 * (size_t)a.trackers.first_bin[k] +
                      (size_t)p.type * (size_t)a.trackers.nbins[k] +
                      (size_t)cmi_frequency_bin(a.trackers, k, p.nu),
                  1. / cmi_projected_area(direction));
              continue;
            }
            const double *d = a.trackers.direction[k];
            if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] > 0. &&
                p.dir[0] * d[0] + p.dir[1] * d[1] + p.dir[2] * d[2] <
                    a.trackers.cos_opening_angle[k])
              continue;
            const uint32_t bin =
                (uint32_t)((p.nu - a.trackers.minimum_frequency) *
                           a.trackers.inverse_frequency_width[k]);
            if (bin < (uint32_t)a.trackers.nbins[k] && p.type < TYPE_ABSORBED)
              atomicAdd(a.trackers.counts +
                            3 * (size_t)a.trackers.first_bin[k] +
                            (size_t)p.type * (size_t)a.trackers.nbins[k] + bin,
                        1ull);
          }
        }
      }
      if (!EXACT && any_periodic && stepping && p.tau >= 0.)
        fast_wrap(a.grid, p);
      if (!EXACT && stepping && p.tau > 0. && !fast_outside(p))
        kappa_next = fast_load_record(a.cells.opacity, p);
      if (CMI_EXP(a) == 1)
        continue;
      if (GROUPED) {
        accumulate_full_grouped<HEAT>(
            a, wq, CMI_EXP(a) == 4 ? 0ull : wave_ballot(accumulate), last_cell,
            ds * p.weight, natomics);
      } else if (FULL) {
        accumulate_full<HEAT>(a, wq, accumulate, last_cell, ds * p.weight,
                              natomics, use_table ? lds_tag : nullptr,
                              lds_val);
        if (use_table && CMI_EXP(a) != 5 &&
            ++window == CMI_FTABLE_WINDOW) {
          window = 0;
          (void)flush_point(true);
        }
      } else if (aggregate != CMI_AGG_NONE) {
        /* lanes in the same cell: one add for the whole run */
        const int32_t key = accumulate ? last_cell : ~lane;
        const double dsw = accumulate ? ds * p.weight : 0.;
        bool tail;
        double v[2] = {dsw * p.sigma_H,
                       HEAT ? dsw * p.sigma_H * (p.nu - a.model.nu_H) : 0.};
        if (use_table) {
          /* the table merges equal cells anyway: sums over groups of 2^ROUNDS
           * lanes are enough to keep the LDS atomics few */
          unsigned long long tails;
          int zero = 0;
          asm volatile("" : "+v"(zero));
          if (HEAT)
            run_sums_masked<2, CMI_TABLE_SCAN_ROUNDS>(key, v, tails, zero);
          else
            run_sums_masked<1, CMI_TABLE_SCAN_ROUNDS>(
                key, reinterpret_cast<double(&)[1]>(v), tails, zero);
          table_add_masked(tails & wave_ballot(accumulate), last_cell, v[0],
                           v[1]);
          continue;
        }
        if (HEAT)
          run_sums<2, 6>(key, v, tail);
        else
          run_sums<1, 6>(key, reinterpret_cast<double(&)[1]>(v), tail);
        const bool add = tail && accumulate;
        if (add) {
          atomic_add_f64(acc_at(a.cells, ION_H_n, last_cell), v[0]);
          if (HEAT)
            atomic_add_f64(acc_at(a.cells, CMI_NION, last_cell), v[1]);
          natomics += HEAT ? 2 : 1;
        }
      } else if (accumulate) {
        const int64_t c = EXACT ? last_cell_wide : (int64_t)last_cell;
        update_integrals_H<HEAT>(a, p.sigma_H, p.nu, p.weight, c, ds);
        natomics += HEAT ? 2 : 1;
      }
    }

    /* ---- end of flight for every lane that cannot step any more ---- */
    if (active) {
      bool inside_now;
      if (EXACT)
        inside_now = is_inside(a.grid, p);
      else
        inside_now = (p.tau < 0.) || !fast_outside(p);
      const int64_t cell_now = EXACT ? last_cell_wide : (int64_t)last_cell;
      if (!(inside_now && p.tau > 0.)) {
        bool absorbed = false, done = false;
        if (p.tau < 0.) {
          /* absorbed inside last_cell (the packet is still inside the box) */
          absorbed = true;
        } else if (inside_now) {
          /* tau hit 0 exactly on a wall, packet still inside: interact()
           * returns the last traversed cell */
          absorbed = (cell_now >= 0);
          done = !absorbed;
        } else {
          done = true; /* left the box: DensityGrid::end() */
          if (!EXACT && a.grid.decomposed && (PAD || last_cell >= 0)) {
            /* ... or only this block of it: hand the flight over. (PAD: the
             * ghost cell the packet stands in says where it went; a packet
             * that never entered the block stands in the corner ghost cell
             * of padded index 0, which belongs to no block.) */
            int64_t cell_global;
            if constexpr (PAD)
              cell_global = pad_next > -1.5
                                ? -1
                                : exit_cell_global_padded(a, a.grid, p);
            else
              cell_global = exit_cell_global(a.grid, p, last_cell);
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
                const uint32_t meta =
                    REEMIT
                        ? cmi_pack_meta(rng.block, rng.have, (uint32_t)p.type,
                                        cmi_meta_origin(lane_meta))
                        : ((lane_meta & CMI_META_KEEP_MASK) |
                           ((uint32_t)p.type << 28));
                r[13] = __longlong_as_double((long long)(
                    ((unsigned long long)meta << 32) | id_of(packet_id)));
                r[14] = 0.;
                r[15] = 0.;
              }
              done = false;
              active = false; /* continues in another block */
            }
          }
        }
        if (absorbed && a.qout.id != nullptr) {
          /* park the packet for the interaction kernel (PhotonSource::reemit
           * happens there): wave-level compaction into the queue */
          if (!EXACT)
            end_flight(p); /* position of the absorption */
          unsigned int q = packet_id; /* park_in_place: the position */
          if (!a.park_in_place) {
            const unsigned long long parked = __ballot(true);
            unsigned int base = 0;
            const int first = __ffsll((long long)parked) - 1;
            if (lane == first)
              base = atomicAdd(a.qout.count, (unsigned int)__popcll(parked));
            base = __shfl(base, first, 64);
            q = base + __popcll(parked & lane_lt);
          }
#pragma unroll
          for (int ax = 0; ax < 3; ++ax)
            a.qout.pos[ax][q] = p.pos[ax];
          a.qout.nu[q] = p.nu;
          a.qout.cell[q] = (int32_t)cell_now;
          a.qout.id[q] = id_of(packet_id);
          a.qout.meta[q] =
              REEMIT ? cmi_pack_meta(rng.block, rng.have, (uint32_t)p.type,
                                     cmi_meta_origin(lane_meta))
                     : ((lane_meta & CMI_META_KEEP_MASK) |
                        ((uint32_t)p.type << 28));
          active = false; /* leaves this launch, not finished */
          last_cell = -1;
          last_cell_wide = -1;
        } else if (absorbed) {
          /* PhotonSource::reemit, src/PhotonSource.cpp:272-308 */
          double new_frequency = 0.;
          if (REEMIT)
            new_frequency =
                reemit_decide<FULL, EXACT>(a.model, a.cells, cell_now, rng, p);
          else
            p.type = TYPE_ABSORBED;
          last_cell = -1;
          last_cell_wide = -1;
          done = (new_frequency == 0.);
          if (REEMIT && !done) {
            reemit_launch<FULL, EXACT>(a.grid, a.model, new_frequency, rng, p,
                                       weights);
            if (FULL) {
#pragma unroll
              for (int i = 0; i < CMI_NACC; ++i)
                stage.weight[lane][i] = weights[cmi_acc_of_column(i)];
            }
          }
        }
        if (done) {
          tc0 += (p.type == TYPE_PRIMARY) ? 1u : 0u;
          tc1 += (p.type == TYPE_DIFFUSE_HI) ? 1u : 0u;
          tc2 += (p.type == TYPE_DIFFUSE_HeI) ? 1u : 0u;
          tc3 += (p.type == TYPE_ABSORBED) ? 1u : 0u;
          if (cmi_meta_origin(lane_meta) != 0)
            atomicAdd(&s_continuous[threadIdx.x >> 6][p.type], 1u);
          active = false;
        }
      }
    }
  }
  /* IonizationPhotonShootJobMarket::update_counters */
  double s0 = wave_sum((double)tc0);
  double s1 = wave_sum((double)tc1);
  double s2 = wave_sum((double)tc2);
  double s3 = wave_sum((double)tc3);
  {
    /* n packets of a type, c of them from the continuous source:
     * (n - c) w_discrete + c w_continuous */
    const unsigned int *c = s_continuous[threadIdx.x >> 6];
    const double w0 = a.model.photon_weight[0], w1 = a.model.photon_weight[1];
    s0 = (s0 - c[0]) * w0 + c[0] * w1;
    s1 = (s1 - c[1]) * w0 + c[1] * w1;
    s2 = (s2 - c[2]) * w0 + c[2] * w1;
    s3 = (s3 - c[3]) * w0 + c[3] * w1;
  }
  double ns = wave_sum((double)nsteps);
  double na = wave_sum((double)natomics);
  if (lane == 0) {
    atomic_add_f64(&counter_shard(a.counters)->totweight, (s0 + s1) + (s2 + s3));
    atomic_add_f64(&counter_shard(a.counters)->typecount[0], s0);
    atomic_add_f64(&counter_shard(a.counters)->typecount[1], s1);
    atomic_add_f64(&counter_shard(a.counters)->typecount[2], s2);
    atomic_add_f64(&counter_shard(a.counters)->typecount[3], s3);
    atomicAdd(&counter_shard(a.counters)->nsteps,
              (unsigned long long)ns + nsteps_wave);
    atomicAdd(&counter_shard(a.counters)->natomics, (unsigned long long)na);
    atomicAdd(&counter_shard(a.counters)->nwavesteps, (unsigned long long)nwavesteps);
  }
}

#include "tile_kernels.h"

/*
 * Interaction kernel: PhotonSource::reemit (src/PhotonSource.cpp:272-308) for
 * every ended flight of `qin` - the diffuse re-emission decision of the
 * handler, and for the packets that go on: new isotropic direction and new
 * optical depth (src/IonizationPhotonShootJob.hpp:139-141). Survivors are
 * compacted into `qout` as ready flights for the next transport launch; the
 * others are counted as absorbed. The packet's random stream continues where
 * the transport kernel left it, so the draws are those of the reference's
 * loop, in its order. One lane per packet; keeping this long, divergent code
 * (spectrum sampling, pow, sincos, log) out of the transport kernel is what
 * lets that one run at 8 waves per SIMD.
 */
struct InteractArgs {
  ModelDev model;
  CellsDev cells;
  CountersDev *counters;
  uint64_t first_packet;
  uint32_t seed;
  uint32_t iteration;
  QueueDev qin;  /* ended flights */
  /* not 0: the places [0, n_in) of qin, holes (id CMI_QUEUE_HOLE) among them
   * (ShootArgs::park_in_place), instead of *qin.count dense entries */
  uint32_t n_in;
  QueueDev qout; /* ready flights */
  /* ROWS: the survivors become flight rows of the tile rounds instead - the
   * new flight is set up here (cell, wall parameters, cross sections) and
   * waits with the key of the tile it starts in */
  GridDev grid;
  TileGridDev tiles;
  FlightRowsDev rows;
  /* interaction_slots_kernel: the absorption records of a tile round */
  const TileItemDev *items;
  const unsigned int *nitems;
  const unsigned int *absorbed_before; /* running totals of the counts */
  const uint32_t *ended_slot;
  /* ... whose new keys (or "free") go to the flights' positions in the next
   * round: key_out[ended_pos[record]] (TileArgs) */
  const uint32_t *ended_pos;
  uint32_t *key_out;
  /* DEFER: the accumulation weights (14 Verner cross sections) of the new
   * flights are left to flight_weights_kernel; the slots kernel lists the
   * slots it filled */
  uint32_t *new_slots;
  unsigned int *new_count;
};

/* The handler's decision for an absorbed packet, in two halves so that the
 * interaction kernels can take the first for a batch of ended flights and the
 * second - the sampled spectrum - only for the survivors, on full waves.
 *
 * What the first half reads from the cell of the absorption: */
struct InteractCell {
  double T, xH, xHe;
};
/* helium takes nothing where A_He sigma_He == 0 (the H-only models: no
 * helium, or FixedValue cross sections with sigma_He = 0): the reference's
 * pHabs = x_H sigma_H / (x_H sigma_H + 0) is exactly 1 whatever x_H > 0 the
 * cell of the absorption has (its opacity was positive), and x_H, x_He are
 * used nowhere else on that side of the decision - their two gathers (of
 * three) are left out */
template <bool FULL>
__device__ __forceinline__ bool interaction_needs_helium(const InteractArgs &a) {
  return FULL ||
         (a.model.abundance[0] * a.model.xsec_fixed[ION_He_n] != 0.);
}
template <bool FULL>
__device__ __forceinline__ InteractCell
interaction_gather(const InteractArgs &a, int32_t cell, bool valid) {
  InteractCell c = {1.e4, 1., 0.};
  if (valid && a.model.reemit_type != 2) {
    c.T = a.cells.temperature[cell];
    if (interaction_needs_helium<FULL>(a)) {
      c.xH = a.cells.x[ION_H_n][cell];
      c.xHe = a.cells.x[ION_He_n][cell];
    }
  }
  return c;
}
/* CMI_REEMIT_*: what becomes of a packet of frequency nu absorbed in that
 * cell */
template <bool FULL>
__device__ __forceinline__ int32_t
interaction_decide(const InteractArgs &a, double nu, const InteractCell &c,
                   PacketRng &rng) {
  if (a.model.reemit_type == 2) {
    /* FixedValueDiffuseReemissionHandler::reemit,
     * src/FixedValueDiffuseReemissionHandler.hpp:73-86 */
    return (rng.next() < a.model.reemit_fixed_probability)
               ? CMI_REEMIT_FIXED
               : CMI_REEMIT_ABSORBED;
  }
  double sigma_H, sigma_He;
  if (FULL) {
    cmi_cross_sections_H_He(a.model, nu, sigma_H, sigma_He);
  } else {
    sigma_H = a.model.xsec_fixed[ION_H_n];
    sigma_He = a.model.xsec_fixed[ION_He_n];
  }
  return physical_reemit_kind(a.model, sigma_H, sigma_He, c.T, c.xH, c.xHe,
                              rng);
}

/* a workgroup's survivors of one batch, staged in LDS: where the ended flight
 * is, the state of its random stream with the kind of its re-emission (packed
 * like a flight's meta word, the kind in the place of the photon type), and
 * the temperature its spectrum is sampled at */
template <unsigned int BATCH> struct InteractStage {
  uint32_t item[BATCH], state[BATCH];
  double T[BATCH], cached[BATCH];
  unsigned int n, base;
};
/* called by all lanes of a wave: a place in the stage for the lanes that
 * want one */
template <unsigned int BATCH>
__device__ __forceinline__ unsigned int
stage_reserve(InteractStage<BATCH> &stage, bool mine) {
  const int lane = threadIdx.x & 63;
  const unsigned long long want = __ballot(mine);
  unsigned int at = 0;
  if (lane == 0 && want)
    at = atomicAdd(&stage.n, (unsigned int)__popcll(want));
  return __builtin_amdgcn_readfirstlane(at) +
         (unsigned int)__popcll(want & ((1ull << lane) - 1ull));
}

/* PhotonSource::reemit + IonizationPhotonShootJob::execute
 * (src/PhotonSource.cpp:296-303, src/IonizationPhotonShootJob.hpp:139-141),
 * in the order of reemit_launch(): the re-emitted packet as a flight of the
 * tile rounds, starting at p.pos. Returns false if the start lies outside the
 * box: the flight ends at once (CartesianDensityGrid::is_inside, :187-227). */
template <bool FULL, bool DEFER = false>
__device__ __forceinline__ bool
interaction_new_flight(const InteractArgs &a, double new_frequency,
                       int32_t type, PacketRng &rng, Packet<FULL> &p,
                       double (&weights)[CMI_NACC], uint32_t &plc,
                       uint32_t &key) {
  p.nu = new_frequency;
  random_direction(p, rng);
  if (!DEFER)
    set_cross_sections(a.model, p, weights);
  p.tau = -log(rng.next());
  p.type = type;
  start_flight<FULL, false>(a.grid, p);
  if (fast_outside(p))
    return false;
  /* (the tile sides of this run's tile kernel: TileGridDev) */
  const int32_t tx = p.index[0] >> a.tiles.log2[0],
                ty = p.index[1] >> a.tiles.log2[1],
                tz = p.index[2] >> a.tiles.log2[2];
  plc = (uint32_t)(p.index[0] - (tx << a.tiles.log2[0])) |
        ((uint32_t)(p.index[1] - (ty << a.tiles.log2[1])) << 8) |
        ((uint32_t)(p.index[2] - (tz << a.tiles.log2[2])) << 16);
  key = tile_index(a.tiles, tx, ty, tz);
  return true;
}

/* `mine` lanes of the workgroup each want one slot of a queue: ONE atomic on
 * the queue's counter for the whole workgroup (a counter word takes ~90
 * returning atomics per microsecond chip-wide; one per wave of 64 entries was
 * the bound of this kernel at 1e8 entries). Must be called by every thread. */
__device__ __forceinline__ unsigned int
block_reserve(bool mine, unsigned int *counter, unsigned int *s_count,
              unsigned int *s_base) {
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const unsigned long long want = __ballot(mine);
  if (lane == 0)
    s_count[wib] = (unsigned int)__popcll(want);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int total = 0;
    for (int w = 0; w < CMI_BLOCK / 64; ++w)
      total += s_count[w];
    *s_base = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  unsigned int q = *s_base;
  for (int w = 0; w < wib; ++w)
    q += s_count[w];
  q += (unsigned int)__popcll(want & ((1ull << lane) - 1ull));
  __syncthreads(); /* s_count / s_base are reused by the next call */
  return q;
}

#ifndef CMI_INTERACT_WAVES_H
#define CMI_INTERACT_WAVES_H 4
#endif
/* multi-ion transport with deferred weights: the two Verner cross sections of
 * the decision and the new flight want ~150 registers; at 4 waves per SIMD
 * (128) they spill 70-110 bytes per lane - and are the faster all the same
 * (round 5, ms per lexington iteration, first-generation kernel / all rounds'
 * slots kernels: 2 waves 9.14 / 7.1, 3: 9.33 / 7.2, 4: 8.31 / 6.8, 5: 9.63 /
 * 7.1, 6: 10.46 / 7.1; the hydrogen-only kernels: 3: 4.03, 4: 3.64, 6: 5.88,
 * 8: 4.03) */
#ifndef CMI_INTERACT_WAVES_FULL
#define CMI_INTERACT_WAVES_FULL 4
#endif
/* trips of a batch whose loads are issued together (multi-ion transport) */
#ifndef CMI_INTERACT_GROUP_FULL
#define CMI_INTERACT_GROUP_FULL 2
#endif
/* Ended flights a workgroup decides on before it asks for room in the output
 * (x CMI_BLOCK). Two things are bought with the batch: (1) the returning
 * atomic on the output counter - ~90 per microsecond chip-wide on one word, so
 * 1e8 ended flights at one atomic per 256 were 4.3 ms on their own - is paid
 * once per batch; (2) the survivors of the batch (a third of the packets of
 * stromgren_diffuse) are staged in LDS and the long second half - direction,
 * optical depth, wall parameters, the row - runs on full waves instead of on
 * the surviving lanes of every wave. */
#ifndef CMI_INTERACT_BATCH
#define CMI_INTERACT_BATCH 4
#endif
template <bool FULL, bool ROWS, bool DEFER = false>
__global__ void __launch_bounds__(CMI_BLOCK,
                                  (FULL && !DEFER) ? 1
                                  : FULL           ? CMI_INTERACT_WAVES_FULL
                                                   : CMI_INTERACT_WAVES_H)
    interaction_kernel(const InteractArgs a) {
  constexpr int TRIPS = CMI_INTERACT_BATCH;
  /* (multi-ion transport: the decision holds two Verner cross sections; the
   * records of four trips beside them spill) */
  constexpr int GROUP = FULL ? CMI_INTERACT_GROUP_FULL : TRIPS;
  constexpr unsigned int BATCH = TRIPS * CMI_BLOCK;
  __shared__ InteractStage<BATCH> stage;
  const int lane = threadIdx.x & 63;
  const uint64_t count = a.n_in ? (uint64_t)a.n_in : (uint64_t)*a.qin.count;
  const uint64_t stride = (uint64_t)gridDim.x * BATCH;
  double tw = 0., tc3 = 0.;
  double tc1 = 0., tc2 = 0.; /* ROWS: re-emitted outside the box (rounding) */
  /* (the trip count is the same for every thread of a workgroup) */
  for (uint64_t base = (uint64_t)blockIdx.x * BATCH; base < count;
       base += stride) {
    if (threadIdx.x == 0)
      stage.n = 0;
    __syncthreads();
    /* the handler's decision for every ended flight of the batch: the loads
     * of GROUP trips first (records, then the cells they point to), so that
     * their latencies overlap */
    for (int k0 = 0; k0 < TRIPS; k0 += GROUP) {
      uint32_t meta[GROUP], id[GROUP];
      int32_t cell[GROUP];
      double nu[GROUP];
      bool there[GROUP];
      InteractCell at_cell[GROUP];
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const uint64_t i =
            base + (unsigned int)(k0 + g) * CMI_BLOCK + threadIdx.x;
        meta[g] = id[g] = 0;
        cell[g] = 0;
        nu[g] = 0.;
        there[g] = false;
        if (i < count) {
          id[g] = a.qin.id[i];
          there[g] = id[g] != CMI_QUEUE_HOLE;
        }
        if (there[g]) {
          meta[g] = a.qin.meta[i];
          cell[g] = a.qin.cell[i];
          if (FULL)
            nu[g] = a.qin.nu[i];
        }
      }
#pragma unroll
      for (int g = 0; g < GROUP; ++g)
        at_cell[g] = interaction_gather<FULL>(a, cell[g], there[g]);
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const unsigned int local =
            (unsigned int)(k0 + g) * CMI_BLOCK + threadIdx.x;
        const bool valid = there[g];
        const uint32_t origin = cmi_meta_origin(meta[g]);
        int32_t kind = CMI_REEMIT_ABSORBED;
        PacketRng rng;
        if (valid) {
          rng.resume(a.seed, a.iteration, a.first_packet + id[g],
                     meta[g] & 0xffffffu, (meta[g] >> 24) & 1u);
          kind = interaction_decide<FULL>(a, nu[g], at_cell[g], rng);
        }
        const bool again = kind != CMI_REEMIT_ABSORBED;
        const unsigned int at = stage_reserve(stage, again);
        if (again) {
          stage.item[at] = local;
          stage.state[at] =
              cmi_pack_meta(rng.block, rng.have, (uint32_t)kind, origin);
          stage.T[at] = at_cell[g].T;
          stage.cached[at] = rng.cached;
        } else if (valid) {
          const double w = a.model.photon_weight[origin];
          tw += w;
          tc3 += w;
        }
      }
    }
    __syncthreads();
    const unsigned int n = stage.n;
    if (threadIdx.x == 0)
      stage.base = n ? atomicAdd(ROWS ? a.rows.count : a.qout.count, n) : 0u;
    __syncthreads();
    const unsigned int q0 = stage.base;
    /* the survivors, one per lane: the new frequency from its spectrum,
     * PhotonSource::reemit's new direction and the new optical depth */
    for (unsigned int j = threadIdx.x; j < n; j += CMI_BLOCK) {
      const uint64_t i = base + stage.item[j];
      const uint32_t state = stage.state[j];
      const uint32_t packet = a.qin.id[i];
      const uint32_t origin = cmi_meta_origin(state);
      PacketRng rng;
      rng.init(a.seed, a.iteration, a.first_packet + packet);
      rng.block = state & 0xffffffu;
      rng.have = (state >> 24) & 1u;
      rng.cached = stage.cached[j];
      int32_t type;
      const double new_frequency = sample_reemission(
          a.model, (int32_t)(state >> 28), stage.T[j], rng, type);
      const unsigned int q = q0 + j;
      if (ROWS) {
        Packet<FULL> p;
        double weights[CMI_NACC];
        uint32_t plc = 0, key = 0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax)
          p.pos[ax] = a.qin.pos[ax][i];
        const bool fly = interaction_new_flight<FULL, DEFER>(
            a, new_frequency, type, rng, p, weights, plc, key);
        if (!fly) {
          const double w = a.model.photon_weight[origin];
          tw += w;
          tc1 += (type == TYPE_DIFFUSE_HI) ? w : 0.;
          tc2 += (type == TYPE_DIFFUSE_HeI) ? w : 0.;
        }
        if (q < a.rows.capacity) {
          if (fly)
            write_flight_row<FULL, DEFER>(
                a.rows, q, a.rows.keys + q, p, plc, key, packet,
                cmi_pack_meta(rng.block, rng.have, (uint32_t)type, origin),
                weights);
          else /* its row stays a free slot */
            a.rows.keys[q] = CMI_TILE_KEY_DEAD(a.tiles);
        }
      } else {
        Packet<false> p;
        random_direction(p, rng);
        const double tau = -log(rng.next());
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          a.qout.pos[ax][q] = a.qin.pos[ax][i];
          a.qout.dir[ax][q] = p.dir[ax];
        }
        a.qout.tau[q] = tau;
        a.qout.nu[q] = new_frequency;
        a.qout.id[q] = packet;
        a.qout.meta[q] =
            cmi_pack_meta(rng.block, rng.have, (uint32_t)type, origin);
      }
    }
    /* (the stage is rewritten after the barrier at the head of the next
     * batch) */
  }
  tw = wave_sum(tw);
  tc3 = wave_sum(tc3);
  if (ROWS) {
    tc1 = wave_sum(tc1);
    tc2 = wave_sum(tc2);
  }
  if (lane == 0 && tw != 0.) {
    atomic_add_f64(&counter_shard(a.counters)->totweight, tw);
    atomic_add_f64(&counter_shard(a.counters)->typecount[3], tc3);
    if (ROWS && tc1 + tc2 != 0.) {
      atomic_add_f64(&counter_shard(a.counters)->typecount[1], tc1);
      atomic_add_f64(&counter_shard(a.counters)->typecount[2], tc2);
    }
  }
}

/* The same for the tile rounds, in place: unit of work k of the tile kernel
 * has left the absorption records of its packets at positions
 * [items[k].begin, + absorbed_count[k]) of `qin`, with the slot each came
 * from; the slot becomes the re-emitted flight, or a free slot. First the
 * running totals of the counts (one workgroup) ... */
__global__ void __launch_bounds__(CMI_TILE_PLAN_THREADS)
    absorbed_scan_kernel(const unsigned int *nitems,
                         const unsigned int *absorbed_count,
                         unsigned int *absorbed_before) {
  __shared__ uint32_t partial[CMI_TILE_PLAN_THREADS];
  const uint32_t n = *nitems;
  const uint32_t per = (n + CMI_TILE_PLAN_THREADS - 1) / CMI_TILE_PLAN_THREADS;
  const uint32_t k0 = threadIdx.x * per;
  const uint32_t k1 = k0 + per < n ? k0 + per : n;
  uint32_t mine = 0;
  for (uint32_t k = k0; k < k1; ++k)
    mine += absorbed_count[k];
  partial[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < CMI_TILE_PLAN_THREADS; off <<= 1) {
    const uint32_t v =
        threadIdx.x >= (unsigned)off ? partial[threadIdx.x - off] : 0u;
    __syncthreads();
    partial[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t at = partial[threadIdx.x] - mine;
  for (uint32_t k = k0; k < k1; ++k) {
    absorbed_before[k] = at;
    at += absorbed_count[k];
  }
  if (threadIdx.x == CMI_TILE_PLAN_THREADS - 1)
    absorbed_before[n] = partial[threadIdx.x];
}

/* ... then one lane per record, whatever unit it belongs to, in the batches
 * of interaction_kernel: the decisions of a batch with their loads overlapped,
 * then the survivors' new flights on full waves (for multi-ion transport that
 * half ends with 14 Verner cross sections unless they are deferred). The
 * running totals of the units around the batch are searched in LDS. */
#define CMI_SLOTS_WINDOW 128
/* (hydrogen-only rounds, x 256 ended flights per batch; ms per launch of
 * stromgren_diffuse's rounds: 1: 0.153, 2: 0.131, 4: 0.152, 8: 0.167) */
#ifndef CMI_SLOTS_BATCH_H
#define CMI_SLOTS_BATCH_H 2
#endif
template <bool FULL, bool DEFER = false>
__global__ void __launch_bounds__(CMI_BLOCK,
                                  (FULL && !DEFER) ? 1
                                  : FULL           ? CMI_INTERACT_WAVES_FULL
                                                   : CMI_INTERACT_WAVES_H)
    interaction_slots_kernel(const InteractArgs a) {
  constexpr int TRIPS = FULL ? CMI_INTERACT_BATCH : CMI_SLOTS_BATCH_H;
  constexpr int GROUP = FULL ? CMI_INTERACT_GROUP_FULL : TRIPS;
  constexpr unsigned int BATCH = TRIPS * CMI_BLOCK;
  __shared__ InteractStage<BATCH> stage;
  __shared__ uint32_t s_slot[BATCH];
  __shared__ uint32_t s_place[BATCH];
  __shared__ uint32_t s_before[CMI_SLOTS_WINDOW + 1];
  const int lane = threadIdx.x & 63;
  const uint32_t key_dead = CMI_TILE_KEY_DEAD(a.tiles);
  const uint32_t nitems = *a.nitems;
  const uint64_t total = a.absorbed_before[nitems];
  const uint64_t stride = (uint64_t)gridDim.x * BATCH;
  double tw = 0., tc1 = 0., tc2 = 0., tc3 = 0.;
  for (uint64_t base = (uint64_t)blockIdx.x * BATCH; base < total;
       base += stride) {
    /* the unit of work the batch's first record belongs to: last k with
     * before[k] <= base (a uniform search: scalar loads), and the totals of
     * the units that follow it */
    uint32_t first = 0, hi = nitems;
    while (hi - first > 1) {
      const uint32_t mid = (first + hi) >> 1;
      if (a.absorbed_before[mid] <= (uint32_t)base)
        first = mid;
      else
        hi = mid;
    }
    if (threadIdx.x == 0)
      stage.n = 0;
    if (threadIdx.x <= CMI_SLOTS_WINDOW)
      s_before[threadIdx.x] = first + threadIdx.x <= nitems
                                  ? a.absorbed_before[first + threadIdx.x]
                                  : 0xffffffffu;
    __syncthreads();
    for (int k0 = 0; k0 < TRIPS; k0 += GROUP) {
      uint32_t record[GROUP], slot[GROUP], meta[GROUP], id[GROUP];
      uint32_t place[GROUP];
      int32_t cell[GROUP];
      double nu[GROUP];
      InteractCell at_cell[GROUP];
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const uint64_t j =
            base + (unsigned int)(k0 + g) * CMI_BLOCK + threadIdx.x;
        record[g] = 0;
        if (j < total) {
          /* this lane's unit: the last l with before[first + l] <= j */
          uint32_t l = 0, h = CMI_SLOTS_WINDOW + 1;
          while (h - l > 1) {
            const uint32_t mid = (l + h) >> 1;
            if (s_before[mid] <= (uint32_t)j)
              l = mid;
            else
              h = mid;
          }
          uint32_t unit = first + l;
          uint32_t before = s_before[l];
          if (l == CMI_SLOTS_WINDOW) /* further on than the window */
            while (a.absorbed_before[unit + 1] <= (uint32_t)j)
              before = a.absorbed_before[++unit];
          record[g] = a.items[unit].begin + ((uint32_t)j - before);
        }
      }
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const uint64_t j =
            base + (unsigned int)(k0 + g) * CMI_BLOCK + threadIdx.x;
        slot[g] = meta[g] = id[g] = place[g] = 0;
        cell[g] = 0;
        nu[g] = 0.;
        if (j < total) {
          const uint32_t i = record[g];
          slot[g] = a.ended_slot[i];
          place[g] = a.ended_pos[i];
          meta[g] = a.qin.meta[i];
          id[g] = a.qin.id[i];
          cell[g] = a.qin.cell[i];
          if (FULL)
            nu[g] = a.qin.nu[i];
        }
      }
#pragma unroll
      for (int g = 0; g < GROUP; ++g)
        at_cell[g] = interaction_gather<FULL>(
            a, cell[g],
            base + (unsigned int)(k0 + g) * CMI_BLOCK + threadIdx.x < total);
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const bool valid =
            base + (unsigned int)(k0 + g) * CMI_BLOCK + threadIdx.x < total;
        const uint32_t origin = cmi_meta_origin(meta[g]);
        int32_t kind = CMI_REEMIT_ABSORBED;
        PacketRng rng;
        if (valid) {
          rng.resume(a.seed, a.iteration, a.first_packet + id[g],
                     meta[g] & 0xffffffu, (meta[g] >> 24) & 1u);
          kind = interaction_decide<FULL>(a, nu[g], at_cell[g], rng);
        }
        const bool again = kind != CMI_REEMIT_ABSORBED;
        const unsigned int at = stage_reserve(stage, again);
        if (again) {
          stage.item[at] = record[g];
          s_slot[at] = slot[g];
          s_place[at] = place[g];
          stage.state[at] =
              cmi_pack_meta(rng.block, rng.have, (uint32_t)kind, origin);
          stage.T[at] = at_cell[g].T;
          stage.cached[at] = rng.cached;
        } else if (valid) {
          const double w = a.model.photon_weight[origin];
          tw += w;
          tc3 += w;
          a.key_out[place[g]] = key_dead;
        }
      }
    }
    __syncthreads();
    const unsigned int n = stage.n;
    if (DEFER) {
      /* the slots that are filled, for flight_weights_kernel (a flight that
       * starts outside the box is listed too: weights for a free slot) */
      if (threadIdx.x == 0)
        stage.base = n ? atomicAdd(a.new_count, n) : 0u;
      __syncthreads();
    }
    for (unsigned int j = threadIdx.x; j < n; j += CMI_BLOCK) {
      const uint32_t i = stage.item[j];
      const uint32_t state = stage.state[j];
      const uint32_t to = s_slot[j];
      const uint32_t packet = a.qin.id[i];
      const uint32_t origin = cmi_meta_origin(state);
      PacketRng rng;
      rng.init(a.seed, a.iteration, a.first_packet + packet);
      rng.block = state & 0xffffffu;
      rng.have = (state >> 24) & 1u;
      rng.cached = stage.cached[j];
      int32_t type;
      const double new_frequency = sample_reemission(
          a.model, (int32_t)(state >> 28), stage.T[j], rng, type);
      Packet<FULL> p;
      double weights[CMI_NACC];
      uint32_t plc = 0, key = key_dead;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax)
        p.pos[ax] = a.qin.pos[ax][i];
      if (interaction_new_flight<FULL, DEFER>(a, new_frequency, type, rng, p,
                                              weights, plc, key)) {
        write_flight_row<FULL, DEFER>(
            a.rows, to, a.key_out + s_place[j], p, plc, key, packet,
            cmi_pack_meta(rng.block, rng.have, (uint32_t)type, origin),
            weights);
      } else {
        const double w = a.model.photon_weight[origin];
        a.key_out[s_place[j]] = key_dead;
        tw += w;
        tc1 += (type == TYPE_DIFFUSE_HI) ? w : 0.;
        tc2 += (type == TYPE_DIFFUSE_HeI) ? w : 0.;
      }
      if (DEFER)
        a.new_slots[stage.base + j] = to;
    }
    __syncthreads(); /* stage.n is reset at the head of the next batch */
  }
  tw = wave_sum(tw);
  tc1 = wave_sum(tc1);
  tc2 = wave_sum(tc2);
  tc3 = wave_sum(tc3);
  if (lane == 0 && tw != 0.) {
    atomic_add_f64(&counter_shard(a.counters)->totweight, tw);
    atomic_add_f64(&counter_shard(a.counters)->typecount[1], tc1);
    atomic_add_f64(&counter_shard(a.counters)->typecount[2], tc2);
    atomic_add_f64(&counter_shard(a.counters)->typecount[3], tc3);
  }
}

/* The accumulation weights of new flights (PhotonSource::set_cross_sections,
 * src/PhotonSource.cpp:189-199: 14 Verner cross sections - ~30 pow() - and the
 * two heating weights) from the frequency in their rows. On its own instead of
 * inside the interaction kernels: there half the lanes idle through it (their
 * packets were absorbed for good) at 2 waves/SIMD under the decision's
 * registers; here every lane has a flight. slots == nullptr: the flights are
 * the slots [0, n). */
struct FlightWeightsArgs {
  ModelDev model;
  FlightRowsDev rows;
  const uint32_t *slots;
  const unsigned int *count; /* or */
  uint64_t n;
};

/* (159 VGPRs, 3 waves/SIMD; bounded to 128 registers it spills and is 15 %
 * slower.) A workgroup takes the flights of a batch in the order of their
 * verner_class (order_by_verner_class, below): re-emitted photons are mostly
 * hydrogen's Lyman continuum just above 13.6 eV, and a wave of those jumps
 * over 19 of the 22 fits. */
template <int TRIPS>
__device__ __forceinline__ void
order_by_verner_class(const uint32_t (&cls)[TRIPS], unsigned int n,
                      uint32_t *s_count, uint16_t *order);
#ifndef CMI_WEIGHTS_BATCH
#define CMI_WEIGHTS_BATCH 4
#endif
#ifndef CMI_WEIGHTS_WAVES
#define CMI_WEIGHTS_WAVES 1
#endif
__global__ void __launch_bounds__(CMI_BLOCK, CMI_WEIGHTS_WAVES)
    flight_weights_kernel(const FlightWeightsArgs a) {
  constexpr int TRIPS = CMI_WEIGHTS_BATCH;
  constexpr unsigned int BATCH = TRIPS * CMI_BLOCK;
  __shared__ double s_nu[BATCH];
  __shared__ uint32_t s_slot[BATCH];
  __shared__ uint16_t s_order[BATCH];
  __shared__ uint32_t s_count[CMI_VERNER_NCLASS];
  uint64_t total = a.count ? (uint64_t)*a.count : a.n;
  if (total > a.rows.capacity)
    total = a.rows.capacity;
  const uint64_t stride = (uint64_t)gridDim.x * BATCH;
  for (uint64_t base = (uint64_t)blockIdx.x * BATCH; base < total;
       base += stride) {
    const unsigned int n =
        total - base < BATCH ? (unsigned int)(total - base) : BATCH;
    uint32_t cls[TRIPS];
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const unsigned int local = (unsigned int)k * CMI_BLOCK + threadIdx.x;
      cls[k] = 0;
      if (local < n) {
        const uint64_t i = base + local;
        const uint32_t q = a.slots ? a.slots[i] : (uint32_t)i;
        const double nu = a.rows.rows[(size_t)CMI_FLIGHT_DOUBLES * q + 6];
        s_slot[local] = q;
        s_nu[local] = nu;
        cls[k] = verner_class(a.model, nu);
      }
    }
    order_by_verner_class<TRIPS>(cls, n, s_count, s_order);
    for (unsigned int j = threadIdx.x; j < n; j += CMI_BLOCK) {
      const unsigned int local = s_order[j];
      const uint32_t q = s_slot[local];
      Packet<true> p;
      p.nu = s_nu[local];
      double weights[CMI_NACC];
      set_cross_sections<true>(a.model, p, weights);
      double4 *w = reinterpret_cast<double4 *>(a.rows.weights +
                                               (size_t)CMI_NACC * q);
#pragma unroll
      for (int k = 0; k < CMI_NACC; k += 4)
        w[k >> 2] = make_double4(weights[k], weights[k + 1], weights[k + 2],
                                 weights[k + 3]);
    }
    __syncthreads(); /* the batch's arrays are rewritten by the next one */
  }
}

/* Sort key of a packet. Reproduces the first draws of emit_packet: the source,
 * the emission direction - binned on an equal-area 2048 x 2048 (cos theta,
 * phi) lattice and Morton-interleaved, 22 bits - and the first optical depth.
 *
 *   key = [source | direction, top dir_hi_bits | tau class | direction, rest]
 *
 * Packets of one coarse direction bin follow the same ray, so how far they get
 * is a monotonic function of their optical depth tau = -ln u: the tau class
 * (the top tau_bits of u) splits the bin's packets into groups that end their
 * flights at about the same step. With bins of ~64 x 2^tau_bits packets each
 * wave gets one group: its lanes stay busy until the bundle ends together,
 * instead of idling behind the longest flight, while the waves of a block
 * still share the bin's cells for the combining table. Inside a group the
 * packets keep their fine direction order, so neighbouring lanes are
 * neighbouring rays. tau_bits = 0 gives the plain direction order. */
struct KeyArgs {
  ModelDev model;
  uint64_t first_packet;
  uint64_t n_packets;
  uint32_t seed;
  uint32_t iteration;
  uint32_t dir_hi_bits; /* 0..dir_bits */
  uint32_t dir_bits;    /* direction bits kept in the key: <= 22 */
  uint32_t tau_bits;    /* 0..3 */
  /* multi-ion transport: a packet's range also depends on its frequency; the
   * class is then the octave of tau sigma_ref / (sigma_H + A_He sigma_He) */
  int32_t full_ions;
  double sigma_ref;
  uint32_t source_mask;
  uint32_t *keys;
  uint32_t *ids;
  /* multi-ion transport: row i receives the emission physics of packet
   * first_packet + i (emit_physics_from_row reads it back); NULL = none */
  double *pre_rows;
  /* not NULL: entry j of the launch is packet first_packet + select[j] (the
   * packets of the launch that start in this block of a decomposed grid,
   * block_select_kernel); keys[j] / ids[j] = select[j], row select[j] */
  const uint32_t *select;
};

/* Decomposed grids: the packets of a launch that THIS block flies - the
 * reference hands a subgrid's source task its own share of the packets
 * (src/DistributedPhotonSource.hpp:140-200,
 * src/SourceDiscretePhotonTaskContext.hpp:112-200); here every block decides
 * for every packet id from the packet's own random numbers (emit_geometry +
 * block_owns_start: ~8 draws and a sincos per packet, no physics), and only
 * the block's own packets get keys, are sorted and are flown. select[0 ..
 * *count) = their ids relative to first_packet, in no particular order (the
 * sort orders them). */
struct SelectArgs {
  GridDev grid;
  ModelDev model;
  uint64_t first_packet; /* of the launch: packet i is first_packet + i ... */
  uint64_t batch_offset; /* ... and has the id batch_offset + i in its call */
  uint64_t n_packets;
  uint32_t seed;
  uint32_t iteration;
  uint32_t *select;
  unsigned int *count;
};
template <bool EXACT>
__global__ void __launch_bounds__(CMI_BLOCK)
    block_select_kernel(const SelectArgs a) {
  const int lane = threadIdx.x & 63;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t rounded = (a.n_packets + 63) & ~(uint64_t)63;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
       i < rounded; i += stride) {
    bool mine = false;
    if (i < a.n_packets) {
      PacketRng rng;
      rng.init(a.seed, a.iteration, a.first_packet + i);
      Packet<false> p;
      (void)emit_geometry<false, EXACT>(a.grid, a.model, rng, p);
      int skipped;
      bool in_block;
      mine = block_owns_start<false, EXACT>(
          a.grid, p, (uint32_t)(a.batch_offset + i), skipped, in_block);
    }
    const unsigned long long owners = __ballot(mine);
    if (owners == 0ull)
      continue;
    unsigned int base = 0;
    const int first = __ffsll((long long)owners) - 1;
    if (lane == first)
      base = atomicAdd(a.count, (unsigned int)__popcll(owners));
    base = __shfl(base, first, 64);
    if (mine)
      a.select[base + __popcll(owners & ((1ull << lane) - 1ull))] =
          (uint32_t)i;
  }
}

__device__ __forceinline__ uint32_t spread_bits_11(uint32_t x) {
  /* 11 bits -> every other bit of 22 */
  x &= 0x7ffu;
  x = (x | (x << 8)) & 0x00ff00ffu;
  x = (x | (x << 4)) & 0x0f0f0f0fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

/* sigma_H + A_He sigma_He of a photon in single precision, for the range
 * class of the sort key (verner_term_sigma's terms of H0 and He0 with float
 * powers: a seventh of the instructions; 1e-6 is plenty for an octave) */
__device__ inline float approximate_opacity_cross_section(const ModelDev &m,
                                                          double nu) {
  if (!m.xsec_verner)
    return (float)(m.xsec_fixed[ION_H_n] +
                   m.abundance[0] * m.xsec_fixed[ION_He_n]);
  if (CMI_UNLIKELY(m.xsec_verner == 2))
    return (float)(cmi_table_value(m.xsec_table, ION_H_n, nu) +
                   m.abundance[0] *
                       cmi_table_value(m.xsec_table, ION_He_n, nu));
  const VernerTermDev *terms = m.tables->verner;
  float sum = 0.f;
  for (int k = 0; k < CMI_VERNER_NTERM_DEV; ++k) {
    const VernerTermDev &t = terms[k];
    if (t.ion != ION_H_n && t.ion != ION_He_n) /* wave-uniform */
      continue;
    if (nu < t.E_th || t.shell > t.ntot)
      continue;
    if (t.shell < t.ntot && t.shell > t.ninn && nu < t.einn)
      continue;
    float s;
    if (t.shell <= t.ninn || nu >= t.einn) {
      const float y = (float)(nu * t.A_E_0_inv);
      s = (float)t.A_sigma_0 * ((y - 1.f) * (y - 1.f) + (float)t.A_y_w_sq) *
          __powf(y, (float)t.A_Plconst) *
          __powf(1.f + sqrtf(y * (float)t.A_y_a_inv), -(float)t.A_P);
    } else {
      const float x = (float)(nu * t.B_E_0_inv - t.B_y_0);
      const float y = sqrtf(x * x + (float)t.B_y_1_sq);
      s = (float)t.B_sigma_0 * ((x - 1.f) * (x - 1.f) + (float)t.B_y_w_sq) *
          __powf(y, 0.5f * (float)t.B_P - 5.5f) *
          __powf(1.f + sqrtf(y * (float)t.B_y_a_inv), -(float)t.B_P);
    }
    sum += (t.ion == ION_H_n) ? s : (float)m.abundance[0] * s;
  }
  return sum;
}

#ifndef CMI_KEY_WAVES
#define CMI_KEY_WAVES 3
#endif
/* A workgroup's photons in the order of their verner_class: in `order` the
 * indices [0, n) of the batch, class after class (any order inside a class).
 * cls = the class of the photon local = trip * CMI_BLOCK + threadIdx.x of each
 * of the thread's TRIPS photons (ignored beyond n). Called by all threads;
 * ends with a barrier. */
template <int TRIPS>
__device__ __forceinline__ void
order_by_verner_class(const uint32_t (&cls)[TRIPS], unsigned int n,
                      uint32_t *s_count /* [CMI_VERNER_NCLASS] */,
                      uint16_t *order) {
  if (threadIdx.x < CMI_VERNER_NCLASS)
    s_count[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TRIPS; ++k)
    if ((unsigned int)k * CMI_BLOCK + threadIdx.x < n)
      atomicAdd(&s_count[cls[k]], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t at = 0;
    for (int c = 0; c < CMI_VERNER_NCLASS; ++c) {
      const uint32_t here = s_count[c];
      s_count[c] = at;
      at += here;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TRIPS; ++k) {
    const unsigned int local = (unsigned int)k * CMI_BLOCK + threadIdx.x;
    if (local < n)
      order[atomicAdd(&s_count[cls[k]], 1u)] = (uint16_t)local;
  }
  __syncthreads();
}

/* direction_key_kernel with the emission physics (pre_rows): the 22 Verner
 * fits of a packet's 14 cross sections are two logarithms, an exponential and
 * a square root each - 8 of this kernel's 16 ms per 1e8 packets of a 40 000 K
 * star when every wave evaluates every term because one of its 64 photons
 * lies above the term's threshold. Four in five of that star's photons lie
 * below 21.6 eV and above three thresholds only. So: a workgroup draws the
 * packets of a batch (direction, frequency, optical depth) into LDS, orders
 * them by the number of thresholds they lie above, and evaluates the fits in
 * that order - most waves then jump over most terms (cmi_cross_sections). */
/* (batches of 2 / 4 / 8 x 256 packets: 13.3 / 11.9 / 11.2 ms per 1e8 packets of
 * lexingtonHII40 - longer runs of one class; 3 / 2 / 4 waves per SIMD: 11.9 /
 * 12.9 / 17.0; round 6) */
#ifndef CMI_KEY_BATCH
#define CMI_KEY_BATCH 8
#endif
__device__ __forceinline__ void
direction_key_batches(const KeyArgs &a) {
  constexpr int TRIPS = CMI_KEY_BATCH;
  constexpr unsigned int BATCH = TRIPS * CMI_BLOCK;
  __shared__ double s_nu[BATCH], s_tau[BATCH];
  __shared__ uint32_t s_src[BATCH], s_morton[BATCH];
  __shared__ uint16_t s_order[BATCH];
  __shared__ uint32_t s_count[CMI_VERNER_NCLASS];
  const uint32_t lo_bits = a.dir_bits - a.dir_hi_bits;
  const uint64_t stride = (uint64_t)gridDim.x * BATCH;
  for (uint64_t base = (uint64_t)blockIdx.x * BATCH; base < a.n_packets;
       base += stride) {
    const unsigned int n = a.n_packets - base < BATCH
                               ? (unsigned int)(a.n_packets - base)
                               : BATCH;
    uint32_t cls[TRIPS];
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const unsigned int local = (unsigned int)k * CMI_BLOCK + threadIdx.x;
      cls[k] = 0;
      if (local >= n)
        continue;
      /* the draws of emit_packet, in its order */
      PacketRng rng;
      const uint64_t id =
          a.select ? (uint64_t)a.select[base + local] : base + local;
      rng.init(a.seed, a.iteration, a.first_packet + id);
      const uint32_t origin =
          rng.next() >= a.model.continuous_probability ? 0u : 1u;
      uint32_t src = 0;
      if (origin == 0) {
        const double xs = rng.next(); /* source pick */
        while (xs > a.model.source_cumulative[src])
          ++src;
      } else {
        src = (uint32_t)a.model.nsource;
        (void)rng.next();
        (void)rng.next();
        if (a.model.continuous_type == 1)
          (void)rng.next();
      }
      const double u_cost = rng.next(); /* cos(theta) = 2 u - 1 */
      const double u_phi = rng.next();  /* phi = 2 pi u */
      const uint32_t ic = (uint32_t)(u_cost * 2048.);
      const uint32_t ip = (uint32_t)(u_phi * 2048.);
      const double nu = sample_source_spectrum(a.model, rng, origin);
      s_nu[local] = nu;
      s_tau[local] = -log(rng.next());
      s_src[local] = src;
      s_morton[local] = (spread_bits_11(ic) | (spread_bits_11(ip) << 1)) >>
                        (22u - a.dir_bits);
      cls[k] = verner_class(a.model, nu);
    }
    order_by_verner_class<TRIPS>(cls, n, s_count, s_order);
    for (unsigned int j = threadIdx.x; j < n; j += CMI_BLOCK) {
      const unsigned int local = s_order[j];
      const uint64_t i = base + local;
      Packet<true> q;
      double weights[CMI_NACC];
      q.nu = s_nu[local];
      set_cross_sections<true>(a.model, q, weights);
      const double tau = s_tau[local];
      weights[CMI_NION] = q.nu;
      weights[CMI_NION + 1] = tau;
      /* (the packet's id in the launch: its row, and what the sort carries) */
      const uint32_t id = a.select ? a.select[i] : (uint32_t)i;
      double4 *row =
          reinterpret_cast<double4 *>(a.pre_rows + (size_t)CMI_NACC * id);
#pragma unroll
      for (int k = 0; k < CMI_NACC; k += 4)
        row[k >> 2] = make_double4(weights[k], weights[k + 1], weights[k + 2],
                                   weights[k + 3]);
      uint32_t tau_class = 0;
      if (a.tau_bits != 0) {
        const double range = tau * a.sigma_ref / (q.sigma_H + q.sigma_He_corr);
        const int octave = (int)floor(log2(range)) + (1 << (a.tau_bits - 1));
        const int top = (1 << a.tau_bits) - 1;
        tau_class = (uint32_t)(octave < 0 ? 0 : (octave > top ? top : octave));
      }
      const uint32_t morton = s_morton[local];
      const uint32_t hi = morton >> lo_bits;
      const uint32_t lo = morton & ((1u << lo_bits) - 1u);
      a.keys[i] = ((s_src[local] & a.source_mask)
                   << (a.dir_bits + a.tau_bits)) |
                  (hi << (a.tau_bits + lo_bits)) | (tau_class << lo_bits) | lo;
      a.ids[i] = id;
    }
    __syncthreads(); /* the batch's arrays are rewritten by the next one */
  }
}

__global__ void __launch_bounds__(CMI_BLOCK, CMI_KEY_WAVES)
    emission_key_kernel(const KeyArgs a) {
  direction_key_batches(a);
}

/* ... and without (pre_rows == nullptr): a kernel of its own, so that its few
 * registers are not the other's 168 */
/* (5 waves per SIMD, as before the table modes and the selection list joined
 * it: their registers are spilled on their own paths) */
__global__ void __launch_bounds__(CMI_BLOCK, 5)
    direction_key_kernel(const KeyArgs a) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t lo_bits = a.dir_bits - a.dir_hi_bits;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
       i < a.n_packets; i += stride) {
    PacketRng rng;
    const uint64_t id = a.select ? (uint64_t)a.select[i] : i;
    rng.init(a.seed, a.iteration, a.first_packet + id);
    const uint32_t origin =
        rng.next() >= a.model.continuous_probability ? 0u : 1u;
    uint32_t src = 0;
    if (origin == 0) {
      const double xs = rng.next(); /* source pick */
      while (xs > a.model.source_cumulative[src])
        ++src;
    } else {
      /* the continuous source: one more "source", its packets ordered by
       * direction like the others (they enter all over the box) */
      src = (uint32_t)a.model.nsource;
      (void)rng.next(); /* the focus point, or the point in the plane */
      (void)rng.next();
      if (a.model.continuous_type == 1)
        (void)rng.next();
    }
    const double u_cost = rng.next(); /* cos(theta) = 2 u - 1 */
    const double u_phi = rng.next();  /* phi = 2 pi u */
    const uint32_t ic = (uint32_t)(u_cost * 2048.);
    const uint32_t ip = (uint32_t)(u_phi * 2048.);
    const uint32_t morton =
        (spread_bits_11(ic) | (spread_bits_11(ip) << 1)) >> (22u - a.dir_bits);
    uint32_t tau_class = 0;
    if (a.tau_bits != 0) {
      if (!a.full_ions) {
        (void)sample_source_spectrum(a.model, rng, origin); /* its draws */
        const double u_tau = rng.next();            /* tau = -ln u */
        tau_class = (uint32_t)(u_tau * (double)(1u << a.tau_bits));
      } else {
        const double nu = sample_source_spectrum(a.model, rng, origin);
        const double tau = -log(rng.next());
        /* (single precision: the class only orders the packets) */
        const float sigma = approximate_opacity_cross_section(a.model, nu);
        const float range = (float)(tau * a.sigma_ref) / sigma;
        const int octave = (int)floorf(log2f(range)) + (1 << (a.tau_bits - 1));
        const int top = (1 << a.tau_bits) - 1;
        tau_class = (uint32_t)(octave < 0 ? 0 : (octave > top ? top : octave));
      }
    }
    const uint32_t hi = morton >> lo_bits;
    const uint32_t lo = morton & ((1u << lo_bits) - 1u);
    a.keys[i] = ((src & a.source_mask) << (a.dir_bits + a.tau_bits)) |
                (hi << (a.tau_bits + lo_bits)) | (tau_class << lo_bits) | lo;
    a.ids[i] = (uint32_t)id;
  }
}

/* ------------------------------------------------------- cell update -- */

struct UpdateArgs {
  GridDev grid;
  ModelDev model;
  CellsDev cells;
  double jfac; /* L / totweight / V_cell */
  double hfac;
  /* the cells [first, first + count) of the engine's grid (the reference's
   * MPI path gives every rank a block of cells,
   * src/IonizationSimulation.cpp:540-618) */
  int64_t first, count;
};

/* IonizationStateCalculator::calculate_ionization_state over the grid
 * (src/IonizationStateCalculator.cpp:511-530 -> :70-272), one cell per lane,
 * grid-stride; also rebuilds the transport record of each cell. */
/* HEATING = false: a hydrogen-only run that does not track the heating terms
 * (no transport kernel adds to them, no temperature solve reads them): their
 * two fields are neither read nor normalised. */
/* (hydrogen-only update of 256^3 cells, ms: the compiler's choice - 3 waves
 * per SIMD - 1.92, built for 4: 1.80, for 6: 3.42) */
#ifndef CMI_IONIZATION_WAVES
#define CMI_IONIZATION_WAVES 4
#endif
template <bool FULL, bool HEATING = true>
__global__ void __launch_bounds__(CMI_BLOCK, FULL ? 2 : CMI_IONIZATION_WAVES)
    ionization_kernel(const UpdateArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t c = a.first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       c < a.first + a.count; c += stride) {
    const double ntot = a.cells.number_density[c];
    const double T = a.cells.temperature[c];
    double J[CMI_NION], heating[2], x[CMI_NION];
    if (FULL) {
#pragma unroll
      for (int i = 0; i < CMI_NION; ++i)
        J[i] = (*acc_at(a.cells, i, c));
    } else {
      J[0] = (*acc_at(a.cells, 0, c));
#pragma unroll
      for (int i = 1; i < CMI_NION; ++i)
        J[i] = 0.;
    }
    heating[0] = HEATING ? (*acc_at(a.cells, CMI_NION, c)) : 0.;
    heating[1] = HEATING ? (*acc_at(a.cells, CMI_NION + 1, c)) : 0.;
    cmi_ionization_state_cell(a.model, a.jfac, a.hfac, ntot, T, J, heating, x);
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      a.cells.x[i][c] = x[i];
    if (HEATING) {
      (*acc_at(a.cells, CMI_NION, c)) = heating[0];
      (*acc_at(a.cells, CMI_NION + 1, c)) = heating[1];
    }
    a.cells.opacity[c] = (ntot > 0.)
                             ? make_double2(ntot * x[ION_H_n], ntot * x[ION_He_n])
                             : make_double2(-1., 0.);
  }
}

/* EmissivityCalculator::calculate_emissivities over the grid
 * (src/EmissivityCalculator.cpp:439-470): one cell per lane, the selected
 * lines written as out[k][cell - first]. Post-processing of the final grid. */
struct EmissivityArgs {
  ModelDev model;
  CellsDev cells;
  int64_t first, count;
  int32_t nlines;
  int32_t lines[CMI_NEMISSIONLINE];
  double *out;
};

__global__ void __launch_bounds__(CMI_BLOCK)
    emissivity_kernel(const EmissivityArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < a.count;
       k += stride) {
    const int64_t c = a.first + k;
    double x[CMI_NION], values[CMI_NEMISSIONLINE];
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      x[i] = a.cells.x[i][c];
    cell_emissivities(a.model, a.cells.number_density[c],
                      a.cells.temperature[c], x, values);
    for (int l = 0; l < a.nlines; ++l)
      a.out[(int64_t)l * a.count + k] = values[a.lines[l]];
  }
}

/* TemperatureCalculator::calculate_temperature over the grid, temperature
 * branch (src/TemperatureCalculator.cpp:944-964 -> :567-931): one cell per
 * lane. fp64-ALU / transcendental bound (up to 100 x 3 balance evaluations,
 * each with ten 5x5 level-population solves). */
/* register budget of the solve (left alone the compiler takes 256 VGPRs
 * without a spill and the kernel is a third slower; with 128 it spills ~140
 * values to scratch, also inside the solve loop, still runs at 2 waves per
 * SIMD - and is the fastest of 96 / 128 / 192 / 256 and of
 * amdgpu_waves_per_eu 3 / 4: DESIGN.md 4.4) and whether the coefficient tables
 * are staged in LDS */
#ifndef CMI_TEMPERATURE_VGPRS
#define CMI_TEMPERATURE_VGPRS 128
#endif
#ifndef CMI_TEMPERATURE_LDS_TABLES
#define CMI_TEMPERATURE_LDS_TABLES 1
#endif
__global__ void __launch_bounds__(CMI_BLOCK)
    __attribute__((amdgpu_num_vgpr(CMI_TEMPERATURE_VGPRS)))
    temperature_kernel(const UpdateArgs a_in) {
  UpdateArgs a = a_in;
#if CMI_TEMPERATURE_LDS_TABLES
  /* the coefficient tables (recombination, charge transfer, line cooling:
   * 19 KB) are read ~1000 times per balance evaluation: staged in LDS */
  __shared__ TablesDev lds_tables;
  {
    const uint64_t *src = reinterpret_cast<const uint64_t *>(a_in.model.tables);
    uint64_t *dst = reinterpret_cast<uint64_t *>(&lds_tables);
    for (unsigned k = threadIdx.x; k < sizeof(TablesDev) / 8; k += CMI_BLOCK)
      dst[k] = src[k];
    __syncthreads();
  }
  a.model.tables = &lds_tables;
#endif
  /* Every lane owns the cells first + lane, + stride, ... and works through
   * them at its own pace: the secant solve of a cell takes 1 to ~10 steps (a
   * step = three balance evaluations, ~60 k instructions), and a wave that
   * gave each lane ONE cell would run as long as its slowest cell (measured:
   * 5.5 steps per wave for ~3 per cell). A lane whose cell has converged - or
   * needs no solve: 57 % of the benchmark's cells are neutral - stores it and
   * takes its next cell while its neighbours go on iterating theirs; all
   * lanes of a wave execute the same step code on whatever cell they hold. */
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t c = a.first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x - stride;
  const int64_t end = a.first + a.count;
  bool active = false;
  double ntot = 0., zmid = 0., T = 0.;
  TemperatureSolve s;
  /* 13 doubles of work space per thread (the line-cooling abundances of a
   * balance evaluation) */
  __shared__ double abund_s[13 * CMI_BLOCK];
  double *const abund = abund_s + threadIdx.x;
  auto store = [&](int64_t cell, const double (&x)[CMI_NION],
                   const double (&heating)[2]) {
    a.cells.temperature[cell] = T;
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      a.cells.x[i][cell] = x[i];
    (*acc_at(a.cells, CMI_NION, cell)) = heating[0];
    (*acc_at(a.cells, CMI_NION + 1, cell)) = heating[1];
    a.cells.opacity[cell] =
        (ntot > 0.) ? make_double2(ntot * x[ION_H_n], ntot * x[ION_He_n])
                    : make_double2(-1., 0.);
  };
  /* the cell's integrals are read where they are (one 128-B row of the
   * accumulator block) whenever a balance evaluation needs them */
  CellIntegrals J;
  J.J = a.cells.acc_base;
  J.stride = a.cells.acc_field_stride;
  J.row = a.cells.acc_cell_stride != 1;
  J.jfac = a.jfac;
  for (;;) {
    /* take the next cell(s): up to four that need no solve per trip */
#pragma unroll 1
    for (int tries = 0; tries < 4 && !active && c + stride < end; ++tries) {
      c += stride;
      ntot = a.cells.number_density[c];
      T = a.cells.temperature[c];
      J.J = acc_at(a.cells, 0, c);
      double x[CMI_NION], heating[2];
      heating[0] = (*acc_at(a.cells, CMI_NION, c));
      heating[1] = (*acc_at(a.cells, CMI_NION + 1, c));
      /* z of the cell midpoint, src/CartesianDensityGrid.hpp:85-89 */
      const int64_t iz = c % a.grid.ncell[2];
      zmid = (a.grid.anchor[2] +
              a.grid.cellside[2] * (iz + a.grid.offset[2])) +
             0.5 * a.grid.cellside[2];
      active = temperature_begin(a.model, J, a.hfac, ntot, T, heating, x, s);
      if (active && !temperature_goes_on(a.model, s)) {
        /* (no iterations allowed: the cell keeps its fractions) */
#pragma unroll
        for (int i = 0; i < CMI_NION; ++i)
          x[i] = a.cells.x[i][c];
        temperature_end(a.model, ntot, J, s, T, heating, x);
        active = false;
      }
      if (!active)
        store(c, x, heating);
    }
    if (__ballot(active) == 0ull) {
      if (__ballot(c + stride < end) == 0ull)
        break;
      continue;
    }
    if (active) {
      temperature_step(a.model, ntot, zmid, J, s, abund, CMI_BLOCK);
      if (!temperature_goes_on(a.model, s)) {
        double x[CMI_NION], heating[2];
        temperature_end(a.model, ntot, J, s, T, heating, x);
        store(c, x, heating);
        active = false;
      }
    }
  }
}

#include "temperature_pipeline.h"

/* probe: one balance evaluation / one temperature solve per input row */
__global__ void thermal_probe_kernel(const ModelDev model, int64_t n,
                                     int32_t solve, const double *J,
                                     const double *heating, const double *T_in,
                                     const double *ntot, double *x_out,
                                     double *T_out, double *gain_loss) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  double j[CMI_NION], h[2], x[CMI_NION];
  for (int k = 0; k < CMI_NION; ++k) {
    j[k] = J[CMI_NION * i + k];
    x[k] = 0.;
  }
  h[0] = heating[2 * i];
  h[1] = heating[2 * i + 1];
  double T = T_in[i];
  CellIntegrals Jc;
  Jc.J = J + CMI_NION * i;
  Jc.stride = 1;
  Jc.jfac = 1.;
  if (solve) {
    temperature_cell(model, Jc, 1., ntot[i], 0.5, T, h, x);
    gain_loss[2 * i] = h[0];
    gain_loss[2 * i + 1] = h[1];
  } else {
    double h0, he0, gain, loss;
    double abund[13];
    cooling_and_heating_balance(model, h0, he0, gain, loss, T, ntot[i], 0.5,
                                Jc, h, x, abund, 1);
    x[ION_H_n] = h0;
    x[ION_He_n] = he0;
    gain_loss[2 * i] = gain;
    gain_loss[2 * i + 1] = loss;
  }
  T_out[i] = T;
  for (int k = 0; k < CMI_NION; ++k)
    x_out[CMI_NION * i + k] = x[k];
}

/* probe of the atomic-data functions, one input row per lane (parity tests of
 * the reference's fixtures on the device):
 *  kind 0: in {nu}                 -> 14 cross sections (cmi_cross_sections)
 *  kind 1: in {T}                  -> 14 recombination rates
 *  kind 2: in {T, n_e, 13 abund.}  -> line cooling (line_cooling)
 *  kind 3: in {T}                  -> 5 re-emission probabilities
 *  kind 4: in {T4}                 -> 14 x {CT recombination with H,
 *                                     CT ionization by H+, CT rec. with He} */
__global__ void physics_probe_kernel(const ModelDev model, int32_t kind,
                                     int64_t n, const double *in,
                                     int32_t in_width, double *out,
                                     int32_t out_width) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const double *r = in + (int64_t)in_width * i;
  double *o = out + (int64_t)out_width * i;
  if (kind == 0) {
    double sigma[CMI_NION];
    cmi_cross_sections(model, r[0], sigma);
    for (int k = 0; k < CMI_NION; ++k)
      o[k] = sigma[k];
  } else if (kind == 1) {
    for (int k = 0; k < CMI_NION; ++k)
      o[k] = cmi_recombination_rate(model, k, r[0]);
  } else if (kind == 2) {
    double abund[13];
    for (int k = 0; k < 13; ++k)
      abund[k] = r[2 + k];
    o[0] = line_cooling(model.tables->lc, r[0], r[1], abund, 1);
  } else if (kind == 3) {
    double pH, pHe[4];
    reemission_probabilities(r[0], pH, pHe);
    o[0] = pH;
    for (int k = 0; k < 4; ++k)
      o[1 + k] = pHe[k];
  } else {
    const TablesDev *tb = model.tables;
    for (int k = 0; k < CMI_NION; ++k) {
      o[3 * k] = ct_eval(tb->ct_recomb_H[k], r[0]);
      o[3 * k + 1] = ct_eval(tb->ct_ion_H[k], r[0]);
      o[3 * k + 2] = ct_eval(tb->ct_recomb_He[k], r[0]);
    }
  }
}

/* probe: n samples of one of the sampled spectra */
__global__ void spectrum_probe_kernel(const ModelDev model, int32_t kind,
                                      double temperature, uint32_t seed,
                                      uint64_t n, double *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  PacketRng rng;
  rng.init(seed, 0u, i);
  double nu;
  if (kind == 0)
    nu = sample_planck(model.spectra, rng);
  else if (kind == 1)
    nu = sample_lyman_continuum(model.spectra, 0, temperature, rng);
  else if (kind == 2)
    nu = sample_lyman_continuum(model.spectra, 1, temperature, rng);
  else
    nu = sample_he_two_photon(model.spectra, rng);
  out[i] = nu;
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

/* PAD kernels: n x_H of every cell inside one layer of ghost cells
 * (ShootArgs::pad_H), from the transport records */
__global__ void __launch_bounds__(CMI_BLOCK)
    pad_record_kernel(const double2 *__restrict__ opacity, double *pad,
                      int32_t nx, int32_t ny, int32_t nz) {
  const int64_t total = (int64_t)(nx + 2 * CMI_PAD_LAYERS) *
                        (ny + 2 * CMI_PAD_LAYERS) * (nz + 2 * CMI_PAD_LAYERS);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total;
       c += stride) {
    constexpr int L = CMI_PAD_LAYERS;
    const int32_t pz = (int32_t)(c % (nz + 2 * L)) - L;
    const int32_t py = (int32_t)((c / (nz + 2 * L)) % (ny + 2 * L)) - L;
    const int32_t px = (int32_t)(c / ((int64_t)(nz + 2 * L) * (ny + 2 * L))) - L;
    double v = CMI_PAD_GHOST;
    if (px >= 0 && px < nx && py >= 0 && py < ny && pz >= 0 && pz < nz) {
      const double k = opacity[((int64_t)px * ny + py) * nz + pz].x;
      v = (k >= 0.) ? k : CMI_PAD_VACUUM;
    }
    pad[c] = v;
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
  double weights[CMI_NACC];
  emit_packet<true, true>(grid, model, rng, p, weights);
  for (int a = 0; a < 3; ++a) {
    position[3 * i + a] = p.pos[a];
    direction[3 * i + a] = p.dir[a];
  }
  frequency[i] = p.nu;
  for (int k = 0; k < CMI_NION; ++k)
    sigma[CMI_NION * i + k] = weights[k];
  tau[i] = p.tau;
}

template <bool EXACT>
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
  Packet<true> p; /* carries the helium term of the optical depth */
  for (int a = 0; a < 3; ++a) {
    p.pos[a] = position[3 * i + a];
    p.dir[a] = direction[3 * i + a];
    p.inv_dir[a] = 1. / p.dir[a];
  }
  p.tau = tau[i];
  p.sigma_H = sigma_H[i];
  p.sigma_He_corr = sigma_He_corr[i];
  p.weight = 1.;
  start_flight<true, EXACT>(grid, p);
  int32_t steps = 0;
  int64_t last = -1;
  while ((EXACT ? is_inside(grid, p) : !fast_outside(p)) && p.tau > 0.) {
    int64_t cell;
    double2 kappa;
    double ds;
    if (EXACT) {
      ds = dda_step(grid, opacity, p, cell, kappa);
    } else {
      int32_t c;
      kappa = fast_load_record(opacity, p);
      ds = fast_step(p, c, kappa);
      cell = c;
      if (p.tau >= 0.)
        fast_wrap(grid, p);
    }
    last = cell;
    if (steps < max_steps) {
      out_cell[(uint64_t)max_steps * i + steps] = cell;
      out_ds[(uint64_t)max_steps * i + steps] = ds;
    }
    ++steps;
  }
  if (!(EXACT ? is_inside(grid, p) : (p.tau < 0. || !fast_outside(p))))
    last = -1;
  if (!EXACT)
    end_flight(p);
  out_nsteps[i] = steps;
  out_last_cell[i] = last;
  for (int a = 0; a < 3; ++a)
    out_position[3 * i + a] = p.pos[a];
}

#endif
