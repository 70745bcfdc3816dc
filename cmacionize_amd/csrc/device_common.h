/*
 * device_common.h - structures shared by the host side of the engine and its
 * kernels (gfx950).
 */
#ifndef CMI_DEVICE_COMMON_H
#define CMI_DEVICE_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CMI_NION 14
#define CMI_NACC 16

enum {
  ION_H_n = 0,
  ION_He_n,
  ION_C_p1,
  ION_C_p2,
  ION_N_n,
  ION_N_p1,
  ION_N_p2,
  ION_O_n,
  ION_O_p1,
  ION_Ne_n,
  ION_Ne_p1,
  ION_S_p1,
  ION_S_p2,
  ION_S_p3
};

enum { TYPE_PRIMARY = 0, TYPE_DIFFUSE_HI, TYPE_DIFFUSE_HeI, TYPE_ABSORBED };

/* reference constants, src/PhysicalConstants.hpp:61-131 */
#define CMI_PLANCK 6.626070040e-34
#define CMI_BOLTZMANN 1.38064852e-23
#define CMI_LIGHTSPEED 299792458.
#define CMI_ELECTRONVOLT 1.6021766208e-19
#define CMI_ELECTRON_MASS 9.10938356e-31

#define CMI_MAX_SOURCES_INLINE 8

/* Geometry of the regular grid (src/CartesianDensityGrid.cpp:40-95) */
struct GridDev {
  double anchor[3];
  double box_sides[3];
  double cellside[3];
  double inv_cellside[3];
  int32_t ncell[3];    /* cells of THIS engine's (sub)grid */
  int32_t periodic[3];
  int64_t ncell_total;
  /* domain decomposition (DensitySubGridCreator,
   * src/DensitySubGridCreator.hpp:314-396): the engine holds the block of
   * cells [offset, offset + ncell) of a grid of global_ncell cells. anchor,
   * box_sides and cellside always describe the WHOLE grid, so that wall
   * coordinates are the same numbers in every block. */
  int32_t offset[3];
  int32_t global_ncell[3];
  int32_t decomposed;
  /* periodicity of the WHOLE grid. A block of a decomposed grid is never
   * periodic itself (periodic[] = 0): a flight that leaves it across a
   * periodic face of the whole box is handed to the block on the other side
   * with its origin shifted by a box side
   * (src/CartesianDensityGrid.cpp:187-227). */
  int32_t global_periodic[3];
  /* copies of a block (DensitySubGridCreator::create_copies,
   * src/DensitySubGridCreator.hpp:437-531): copy `copy_rank` of `copy_count`
   * engines holding this block emits the packets whose id is congruent to it */
  int32_t copy_rank, copy_count;
};

/* A flight handed from one block of a decomposed grid to another: the FAST
 * marcher's state at the wall, CMI_FLIGHT_DOUBLES doubles per flight:
 *  [0-2] origin of the flight  [3-5] direction  [6] path parameter t
 *  [7-9] next wall parameter per axis  [10] optical depth left  [11] frequency
 *  [12] (int64) long index, in the WHOLE grid, of the cell being entered
 *  [13] (2 x uint32) packet id, meta  [14-15] unused */
#define CMI_FLIGHT_DOUBLES 16
struct ExchangeDev {
  double *rows;
  unsigned int *count;
  unsigned int capacity;
};

/* Tile rounds (incoherent flights: re-emitted packets, flights handed over):
 * the engine's grid is cut into tiles of 2^log2[0] x 2^log2[1] x 2^log2[2]
 * cells (TileShape), a
 * flight is marched one tile at a time by a workgroup that keeps the tile's
 * accumulators in LDS; between rounds the flights wait as rows of
 * CMI_FLIGHT_DOUBLES doubles (the marcher's state, as in a hand-over between
 * blocks) in slots of their own layout (below) plus, for multi-ion
 * transport, the packet's 16 accumulation weights in a second array
 * (weights[slot][16]) so that no cross section is evaluated twice.
 * keys[slot] is the tile of the entered cell (sorted between rounds). */
/* Slot layout of the tile rounds (CMI_FLIGHT_DOUBLES doubles, two 64-B
 * halves): what a tile visit never changes,
 *   [0-2] origin of the flight  [3-5] direction  [6] frequency
 *   [7] (2 x uint32) packet id, meta
 * and what it does (ONE full line written per visit),
 *   [8] path parameter t  [9-11] next wall parameter per axis
 *   [12] optical depth left
 *   [13] (2 x uint32) long index of the cell being entered, in this engine's
 *        grid | its coordinates inside the tile, x | y<<8 | z<<16
 *   [14-15] the spacing of the x and y walls along the flight (tdelta of
 *        start_flight(); constant, carried so that a visit divides once)
 * An absorbed packet leaves its absorption record in the same slot: position
 * in [0-2], cell in [13]. */
#define CMI_SLOT_NU 6
#define CMI_SLOT_IDMETA 7
#define CMI_SLOT_T 8
#define CMI_SLOT_TMAX 9
#define CMI_SLOT_TAU 12
#define CMI_SLOT_CELL 13

struct TileGridDev {
  int32_t log2[3];  /* tile sides = 1 << log2[axis] cells */
  int32_t ntile[3]; /* tiles per axis (the last one may be clipped) */
  int32_t ntiles;
};
/* Tile sides (log2) and workgroup size per transport flavour. What the LDS of
 * a CU (160 KB) holds decides: hydrogen-only J_H + records of 32 x 16 x 16
 * cells (64 + 64 KB, one workgroup of 1024 threads per CU); with the heating
 * term 16^3 cells (3 x 32 KB, 512 threads); 14 ions + heating 8 x 8 x 16
 * cells (128 KB of accumulators + 16 KB of records, 1024 threads). Measured
 * on 256^3, ms of transport per iteration (profiles/r05/tile_shapes.txt):
 * hydrogen-only 16^3 / 512 threads 76.3, 32 x 16 x 16 / 1024 74.3; multi-ion
 * 8^3 / 512 180.2, 8 x 8 x 16 / 1024 174.9 (16 x 8 x 8: 177.0) - larger
 * tiles are fewer visits, fewer rounds and above all a smaller tail for the
 * per-step atomics of the pass kernels. The experiment builds
 * (`make variant DEFS=-DCMI_TILE_LZ_FULL=3 ...`) override them. */
#ifndef CMI_TILE_LX_H
#define CMI_TILE_LX_H 5
#define CMI_TILE_LY_H 4
#define CMI_TILE_LZ_H 4
#define CMI_TILE_THREADS_H 1024
#endif
#ifndef CMI_TILE_LX_HHEAT
#define CMI_TILE_LX_HHEAT 4
#define CMI_TILE_LY_HHEAT 4
#define CMI_TILE_LZ_HHEAT 4
#define CMI_TILE_THREADS_HHEAT 512
#endif
#ifndef CMI_TILE_LX_FULL
#define CMI_TILE_LX_FULL 3
#define CMI_TILE_LY_FULL 3
#define CMI_TILE_LZ_FULL 4
#define CMI_TILE_THREADS_FULL 1024
#endif
template <bool FULL, bool HEAT> struct TileShape {
  static constexpr int LX =
      FULL ? CMI_TILE_LX_FULL : (HEAT ? CMI_TILE_LX_HHEAT : CMI_TILE_LX_H);
  static constexpr int LY =
      FULL ? CMI_TILE_LY_FULL : (HEAT ? CMI_TILE_LY_HHEAT : CMI_TILE_LY_H);
  static constexpr int LZ =
      FULL ? CMI_TILE_LZ_FULL : (HEAT ? CMI_TILE_LZ_HHEAT : CMI_TILE_LZ_H);
  static constexpr int THREADS =
      FULL ? CMI_TILE_THREADS_FULL
           : (HEAT ? CMI_TILE_THREADS_HHEAT : CMI_TILE_THREADS_H);
  static constexpr int TX = 1 << LX, TY = 1 << LY, TZ = 1 << LZ;
  static constexpr int CELLS = TX * TY * TZ;
  /* index of tile-local coordinates in the tile's LDS arrays */
  __host__ __device__ static constexpr int index(int x, int y, int z) {
    return (x << (LY + LZ)) | (y << LZ) | z;
  }
};
struct FlightRowsDev {
  double *rows;     /* [capacity][CMI_FLIGHT_DOUBLES] */
  double *weights;  /* [capacity][CMI_NACC], multi-ion transport only */
  uint32_t *keys;   /* [capacity] tile of the cell being entered */
  unsigned int *count;
  unsigned int capacity;
};
/* one unit of work of the tile kernel: flights order[begin, end) of a tile */
struct TileItemDev {
  uint32_t tile, begin, end, pad;
};

/* one (ion, shell) term of the Verner cross section, converted as in
 * src/VernerCrossSections.cpp:36-154 */
struct VernerTermDev {
  double E_th, einn;
  double A_Plconst, A_E_0_inv, A_sigma_0, A_y_a_inv, A_P, A_y_w_sq;
  double B_E_0_inv, B_sigma_0, B_y_a_inv, B_P, B_y_w_sq, B_y_0, B_y_1_sq;
  int32_t ion, shell, ninn, ntot;
};
#define CMI_VERNER_NTERM_DEV 22

/* Recombination rate of one ion as coefficient rows
 * (src/VernerRecombinationRates.cpp:140-333): the radiative fit
 *   kind 0: p0 / (tt (tt + 1)^(1 - p1) (1 + sqrt(T p3))^(1 + p1)),
 *           tt = sqrt(T p2)   (Verner & Ferland 1996; p2, p3 pre-inverted;
 *           H and He use the same form with their own constants, :165-190)
 *   kind 1: p0 (T / 1e4 K)^-p1
 * plus a dielectronic term
 *   dkind 0: none
 *   dkind 1: 1e-12 (d0 / t + d1 + d2 t + d3 t^2) t^-1.5 exp(-d4 / t),
 *            t = T / 1e4 K   (Nussbaumer & Storey 1983, :197-285)
 *   dkind 2: t^-1.5 sum_k c_k exp(-E_k / t), t = T dunit, dn terms
 *            (Mazzotta et al. 1998 in eV, Abdel-Naby et al. 2012 in K,
 *            :288-330)
 * in cm^3 s^-1 (converted to m^3 s^-1 by the evaluator). */
struct VernerRecDev {
  double p[4];
  double d[5];    /* dkind 1 */
  double dc[6];   /* dkind 2: c_k */
  double dE[6];   /* dkind 2: E_k */
  double dunit;   /* dkind 2: t = T * dunit */
  int32_t kind;
  int32_t dkind;
  int32_t dn;
  int32_t pad;
};

/* charge transfer fit a * t^b * (1 + c exp(d t)) [* exp(e / t)]
 * (src/ChargeTransferRates.cpp) */
struct CTFitDev {
  double a, b, c, d, e, lo, hi;
  int32_t kind;
  int32_t pad;
};

/* line cooling data, converted as in src/LineCoolingData.cpp:42-1399:
 * 10 five-level ions (NI NII OI OII OIII NeIII SII SIII CII CIII) and 3
 * two-level ions (NIII NeII SIV); transitions 0-1 0-2 0-3 0-4 1-2 1-3 1-4
 * 2-3 2-4 3-4 */
#define CMI_LC_NFIVE_DEV 10
#define CMI_LC_NTWO_DEV 3
#define CMI_LC_NTRANS_DEV 10
struct LineCoolingDev {
  double energy[CMI_LC_NFIVE_DEV][CMI_LC_NTRANS_DEV]; /* K */
  double A[CMI_LC_NFIVE_DEV][CMI_LC_NTRANS_DEV];      /* s^-1 */
  double cs[CMI_LC_NFIVE_DEV][CMI_LC_NTRANS_DEV][7];  /* Omega(T) fit */
  double inv_weight[CMI_LC_NFIVE_DEV][5];
  double two_energy[CMI_LC_NTWO_DEV];
  double two_A[CMI_LC_NTWO_DEV];
  double two_cs[CMI_LC_NTWO_DEV][7];
  double two_inv_weight[CMI_LC_NTWO_DEV][2];
  double prefactor; /* h^2 / (sqrt(k) (2 pi m_e)^1.5) */
};

/* All read-only physics tables, one instance in device memory */
struct TablesDev {
  VernerTermDev verner[CMI_VERNER_NTERM_DEV];
  VernerRecDev verner_rec[CMI_NION];
  CTFitDev ct_recomb_H[CMI_NION];
  CTFitDev ct_ion_H[CMI_NION];
  CTFitDev ct_recomb_He[CMI_NION];
  /* the charge transfer processes that enter the balance of each metal ion
   * (src/IonizationStateCalculator.cpp:323-501): [ion][0] recombination with
   * H, [1] ionization by H+, [2] recombination with He; kind 0 where the
   * reference's balance has no such term */
  CTFitDev metal_ct[CMI_NION][3];
  LineCoolingDev lc;
};

/* CDF tables of the sampled spectra: the members of
 * PlanckPhotonSourceSpectrum, Hydrogen/HeliumLymanContinuumSpectrum and
 * HeliumTwoPhotonContinuumSpectrum. 1.65 MB, lives in HBM (L2 resident). */
#define CMI_NFREQ 1000
#define CMI_NTEMP 100
/* Guide tables of the cumulative distributions: guide[k] = the last entry of
 * the distribution below k / CMI_NGUIDE (0 if there is none). A uniform x
 * then lies between entries guide[floor(x CMI_NGUIDE)] and guide[.. + 1] + 1:
 * Utilities::locate's bisection (src/Utilities.hpp:726-742) started from that
 * bracket ends at the same index as from [0, length) - in 0-2 steps instead
 * of 10, each a dependent load that a lane of the interaction kernels waits
 * for. */
#define CMI_NGUIDE 1024
struct SpectraDev {
  double planck_logfreq[CMI_NFREQ];
  double planck_cdf[CMI_NFREQ];
  double planck_logcdf[CMI_NFREQ];
  double lyc_T[CMI_NTEMP];
  double lyc_freq[2][CMI_NFREQ];           /* [H, He] */
  double lyc_cdf[2][CMI_NTEMP][CMI_NFREQ]; /* [H, He][T][nu] */
  double he2pc_freq[CMI_NFREQ];
  double he2pc_cdf[CMI_NFREQ];
  /* the Planck spectrum of the continuous source, if it has one */
  double planck2_logfreq[CMI_NFREQ];
  double planck2_cdf[CMI_NFREQ];
  double planck2_logcdf[CMI_NFREQ];
  uint16_t planck_guide[CMI_NGUIDE + 2];
  uint16_t planck2_guide[CMI_NGUIDE + 2];
  uint16_t he2pc_guide[CMI_NGUIDE + 2];
  uint16_t lyc_guide[2][CMI_NTEMP][CMI_NGUIDE + 2];
};

/* A caller-supplied table (cmi_gpu_set_spectrum_table,
 * cmi_gpu_set_cross_sections_table, cmi_gpu_set_recombination_rates_table):
 * the generic lowering of a plugin that is known only through the reference's
 * per-packet / per-cell virtual ("sample the virtual on the host
 * into a table"). n abscissae x[] in ascending order and the values y[] - one
 * row of n for a spectrum, CMI_NION rows of n for cross sections and
 * recombination rates. Device pointers in the kernels' copy of the model,
 * host pointers in the host's. */
struct TableDev {
  const double *x;
  const double *y;
  int32_t n;
  int32_t interpolation; /* CMI_GPU_TABLE_LINEAR / CMI_GPU_TABLE_LOGLOG */
};
#define CMI_TABLE_LINEAR 0
#define CMI_TABLE_LOGLOG 1

/* Physics set-up passed by value to the kernels */
struct ModelDev {
  /* sources */
  int32_t nsource;
  int32_t spectrum_type;
  const double *source_position;   /* device [nsource][3] */
  const double *source_cumulative; /* device [nsource] */
  double total_luminosity;
  double mono_frequency;
  double planck_temperature;
  /* cross sections / recombination */
  int32_t xsec_verner;
  int32_t recomb_verner;
  double xsec_fixed[CMI_NION];
  double recomb_fixed[CMI_NION];
  /* abundances: He C N O Ne S */
  double abundance[6];
  /* reemission */
  int32_t reemit_type;
  int32_t pad0;
  double reemit_fixed_probability;
  double reemit_fixed_frequency;
  /* thresholds in Hz, src/DensityGrid.hpp:219-222 */
  double nu_H, nu_He;
  const TablesDev *tables;
  const SpectraDev *spectra; /* NULL until a sampled spectrum is needed */
  /* TemperatureCalculator parameters, src/TemperatureCalculator.cpp:133-160 */
  double t_epsilon;
  double pahfac, crfac, crlim, crscale;
  double t_min_ionized;
  int32_t t_max_iterations;
  /* ContinuousPhotonSource + its spectrum, and the mix of the two kinds of
   * sources (PhotonSource ctor, src/PhotonSource.cpp:104-130): a packet comes
   * from the continuous source with probability continuous_probability and
   * carries photon_weight[1], from a discrete one otherwise, with
   * photon_weight[0] */
  int32_t continuous_type; /* 0 none, 1 isotropic on the box, 2 planar */
  int32_t continuous_spectrum_type; /* as spectrum_type */
  int32_t pad1;
  double continuous_probability;
  double photon_weight[2];
  double continuous_mono_frequency;
  double continuous_planck_temperature;
  /* PlanarContinuousPhotonSource: the plane x[axis] = intercept, the
   * rectangle [anchor, anchor + side] along the two other axes (in their
   * natural order) */
  int32_t continuous_axis;
  int32_t pad2;
  double continuous_intercept;
  double continuous_anchor[2], continuous_side[2];
  /* generic lowering: spectrum_type / continuous_spectrum_type == 2 sample
   * spectrum_table[0] / [1] (x = cumulative distribution, y = frequency in
   * Hz); xsec_verner == 2 interpolates xsec_table (x = frequency in Hz, y =
   * sigma[14][n] in m^2); recomb_verner == 2 interpolates recomb_table (x =
   * temperature in K, y = alpha[14][n] in m^3 s^-1) */
  TableDev spectrum_table[2];
  TableDev xsec_table;
  TableDev recomb_table;
};

/* SoA cell state, all device pointers to [ncell] doubles */
struct CellsDev {
  double *number_density;
  double *temperature;
  double *x[CMI_NION];
  /* accumulators: 14 mean intensities + 2 heating terms; element (field f,
   * cell c) lives at acc_base[f * acc_field_stride + c * acc_cell_stride].
   * SoA ([16][ncell]: strides ncell, 1) for H-only runs, AoS ([ncell][16]:
   * strides 1, 16) when all 16 are updated per step */
  double *acc_base;
  int64_t acc_field_stride;
  int64_t acc_cell_stride;
  /* transport record: {n * x_H, n * x_He}; .x < 0 marks a vacuum cell */
  double2 *opacity;
};

/* Where accumulator f (0-13: J of the ion, 14 / 15: the hydrogen / helium
 * heating term) sits in a cell's 128-B row of the AoS layout: by ionization
 * threshold, so that everything a photon below 24.59 eV (He0) can add to -
 * H0 13.60, O0 13.62, N0 14.53, Ne0 21.56, S+ 23.33, C+ 24.38 eV, the
 * hydrogen heating term; N+ 29.60 fills the half - lies in the FIRST 64-B
 * line and the rest - He0, the helium heating term, S++ 34.83, O+ 35.12, Ne+
 * 40.96, S+++ 47.30, N++ 47.45, C++ 47.89 - in the second. Atomic adds of
 * zero are not sent, and nine in ten photons of a 40 000 K star (and most
 * re-emitted ones) are that soft: their steps cost ONE memory-side 64-B
 * request instead of two - the request rate is what bounds the kernels that
 * add per step. (SoA blocks keep the accumulators' own order.)
 * column -> accumulator: 0 7 4 14 9 11 2 5 | 1 15 12 8 10 13 6 3 */
__host__ __device__ __forceinline__ constexpr int cmi_acc_column(int f) {
  /* nibble f = column of accumulator f */
  return (int)((0x93DA5C4B1E72F680ull >> (4 * f)) & 15ull);
}
__host__ __device__ __forceinline__ constexpr int cmi_acc_of_column(int col) {
  /* nibble col = accumulator in column col */
  return (int)((0x36DA8CF152B9E470ull >> (4 * col)) & 15ull);
}
static_assert(cmi_acc_column(0) == 0 && cmi_acc_of_column(3) == 14 &&
                  cmi_acc_column(15) == 9 && cmi_acc_of_column(15) == 3,
              "accumulator 0 (J_H) is column 0: a row's base is its address");
/* columns that hold a mean intensity (not a heating term): bit per column */
#define CMI_ACC_COLUMNS_OF_IONS 0xFDF7u

__host__ __device__ __forceinline__ double *acc_at(const CellsDev &cells,
                                                   int field, int64_t cell) {
  /* (the row layout is the one whose CELLS are 16 values apart: on a grid of
   * one cell the field stride of the SoA layout is 1 as well) */
  const int at =
      cells.acc_cell_stride != 1 ? cmi_acc_column(field) : field;
  return cells.acc_base + at * cells.acc_field_stride +
         cell * cells.acc_cell_stride;
}

/* Packet queues between kernels (SoA). Two kinds of entry share the struct:
 *
 *  ended flights  (transport -> interaction kernel): a packet that was absorbed;
 *                 pos = where, cell = in which cell, nu = its frequency;
 *  ready flights  (interaction -> transport kernel, or imported from another
 *                 process): a packet about to fly; pos, dir, tau = the
 *                 optical depth left, nu.
 *
 * id is the packet's index relative to the launch's first packet; meta says
 * how far its random stream has been consumed and carries its type. */
struct QueueDev {
  double *pos[3];
  double *dir[3]; /* ready flights only */
  double *tau;    /* ready flights only */
  double *nu;
  int32_t *cell;  /* ended flights only */
  uint32_t *id;
  uint32_t *meta; /* bits 0-23 rng blocks consumed, bit 24 rng cache valid,
                     bits 28-31 photon type */
  unsigned int *count;
};

/* the word that travels with a packet id: position of the packet's random
 * stream (24 bits of block counter, 1 bit "second half of the block unread"),
 * bit 25 = the packet came from the continuous source (its weight is
 * ModelDev::photon_weight[1]), bits 28-31 the photon type */
#define CMI_META_ORIGIN_SHIFT 25
#define CMI_META_KEEP_MASK 0x03ffffffu /* everything but the type */
__host__ __device__ inline uint32_t cmi_pack_meta(uint32_t rng_block,
                                                  uint32_t rng_have,
                                                  uint32_t type,
                                                  uint32_t origin) {
  return (rng_block & 0xffffffu) | ((rng_have & 1u) << 24) |
         ((origin & 1u) << CMI_META_ORIGIN_SHIFT) | (type << 28);
}
__host__ __device__ inline uint32_t cmi_meta_origin(uint32_t meta) {
  return (meta >> CMI_META_ORIGIN_SHIFT) & 1u;
}

/* packet counters accumulated by the transport kernel */
/* SpectrumTracker (src/SpectrumTracker.hpp:41-262) of up to
 * CMI_MAX_TRACKERS cells: packets that cross a tracked cell are counted by
 * frequency bin and photon type (primary, diffuse H, diffuse He), optionally
 * only those flying within a cone around a reference direction. Counted by
 * the kernels with the exact marcher (a run with trackers uses those). */
#define CMI_MAX_TRACKERS 16
struct TrackersDev {
  int32_t n; /* 0: none */
  /* bins of tracker k over [1, 4) x 3.289e15 Hz: nbins[k], its counts at
   * counts[3 first_bin[k] + type nbins[k] + bin] */
  int32_t nbins[CMI_MAX_TRACKERS];
  int32_t first_bin[CMI_MAX_TRACKERS + 1];
  double inverse_frequency_width[CMI_MAX_TRACKERS];
  double minimum_frequency;
  int64_t cell[CMI_MAX_TRACKERS]; /* index in this engine's grid, -1: not here */
  double cos_opening_angle[CMI_MAX_TRACKERS];
  double direction[CMI_MAX_TRACKERS][3]; /* normalised; all zero: any */
  unsigned long long *counts;            /* [n][3][nbins[k]] */
  /* AbsorptionTrackers (src/AbsorptionTracker.hpp:49-235) among them: sums
   * of path length x cross section x weight per photon type and ion */
  int32_t kind[CMI_MAX_TRACKERS]; /* CMI_TRACKER_* */
  double *absorption;             /* [n][4][CMI_NION] */
  /* WeightedSpectrumTrackers (src/WeightedSpectrumTracker.hpp:44-420): every
   * crossing adds 1 / (the unit cube's area as the photon sees it) to
   * flux[4 first_bin[k] + type nbins[k] + bin], the bin from the tracker's
   * FrequencyBins: type 0 = LinearFrequencyBins (src/LinearFrequencyBins.hpp:
   * 115-125: nbins[k] bins between bins_min and bins_max, frequencies outside
   * in the first / last bin), 1 = LevelFrequencyBins
   * (src/LevelFrequencyBins.hpp:45-70: the 14 ionization energies in
   * ascending order, up to 4 x hydrogen's) */
  int32_t bins_type[CMI_MAX_TRACKERS];
  /* (a weighted tracker's inverse_frequency_width[k] is that of its bins) */
  double bins_min[CMI_MAX_TRACKERS], bins_max[CMI_MAX_TRACKERS];
  double level_edges[CMI_NION + 1];
  double *flux; /* [n][4][nbins[k]] */
};
#define CMI_TRACKER_SPECTRUM 0
#define CMI_TRACKER_ABSORPTION 1
#define CMI_TRACKER_WEIGHTED 2

#define CMI_BINS_LINEAR 0
#define CMI_BINS_LEVEL 1

/* FrequencyBins::get_bin_number of tracker k: LinearFrequencyBins.hpp:115-125,
 * LevelFrequencyBins.hpp:84-86 (Utilities::locate, src/Utilities.hpp:726-742:
 * bisection for the last edge below the frequency, the last bin for anything
 * above the upper edge) */
__host__ __device__ inline int32_t cmi_frequency_bin(const TrackersDev &t,
                                                     const int k,
                                                     const double frequency) {
  if (t.bins_type[k] == CMI_BINS_LEVEL) {
    uint32_t jl = 0, ju = CMI_NION + 1;
    while (ju - jl > 1) {
      const uint32_t jm = (ju + jl) >> 1;
      if (frequency > t.level_edges[jm])
        jl = jm;
      else
        ju = jm;
    }
    return jl == CMI_NION ? CMI_NION - 1 : (int32_t)jl;
  }
  if (frequency < t.bins_min[k])
    return 0;
  if (frequency >= t.bins_max[k])
    return t.nbins[k] - 1;
  return (int32_t)((frequency - t.bins_min[k]) * t.inverse_frequency_width[k]);
}

/* WeightedSpectrumTracker::get_projected_area,
 * src/WeightedSpectrumTracker.hpp:200-276: the area of the unit cube's
 * shadow on a plane perpendicular to the direction. The reference projects
 * seven corners onto that plane and adds up the areas of six triangles (half
 * the norms of cross products, taken as sqrt(|u|^2 |w|^2 - (u . w)^2)); its
 * own test asks for exactly 1 along the axes and exactly sqrt(2) along a face
 * diagonal, so the same corners, pairs and order of operations are kept. */
__host__ __device__ inline double cmi_projected_area(const double d[3]) {
  /* corners 100 010 001 110 101 011 111, projected: v - ((v - m) . d) d */
  const double corner[7][3] = {{1., 0., 0.}, {0., 1., 0.}, {0., 0., 1.},
                               {1., 1., 0.}, {1., 0., 1.}, {0., 1., 1.},
                               {1., 1., 1.}};
  enum { P100 = 0, P010, P001, P110, P101, P011, P111 };
  double p[7][3];
  for (int v = 0; v < 7; ++v) {
    const double along = (corner[v][0] - 0.5) * d[0] +
                         (corner[v][1] - 0.5) * d[1] +
                         (corner[v][2] - 0.5) * d[2];
    for (int a = 0; a < 3; ++a)
      p[v][a] = corner[v][a] - d[a] * along;
  }
  /* {from, to of the first edge, to of the second edge} of the six triangles */
  const int triangle[6][3] = {{P100, P101, P111}, {P100, P110, P111},
                              {P110, P111, P011}, {P110, P011, P010},
                              {P101, P001, P011}, {P101, P011, P111}};
  double twice_area = 0.;
  for (int t = 0; t < 6; ++t) {
    double u[3], w[3];
    for (int a = 0; a < 3; ++a) {
      u[a] = p[triangle[t][1]][a] - p[triangle[t][0]][a];
      w[a] = p[triangle[t][2]][a] - p[triangle[t][0]][a];
    }
    const double uw = u[0] * w[0] + u[1] * w[1] + u[2] * w[2];
    const double cross2 = (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) *
                              (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) -
                          uw * uw;
    /* ("make sure we don't get NaN" - the reference guards four of the six;
     * a difference of rounding errors below zero is no area either way) */
    twice_area += cross2 > 0. ? sqrt(cross2) : 0.;
  }
  return 0.5 * twice_area;
}

struct CountersDev {
  double totweight;
  double typecount[4];
  unsigned long long nsteps;
  unsigned long long natomics; /* atomic adds issued to the accumulators */
  unsigned long long nwavesteps; /* hot-loop iterations summed over waves */
};

/* The counters exist CMI_COUNTER_SHARDS times, one 64-B line each; a wave adds
 * its sums to the shard its index picks and the host adds the shards up.
 * (Atomics on ONE line are served one after the other, ~100 per us on MI355X:
 * the 8 adds of each of the ~8000 waves of a launch kept the last wave of
 * every launch waiting for ~0.4 ms - more than the work of a late tile round.) */
#define CMI_COUNTER_SHARDS 1024
static_assert(sizeof(CountersDev) == 64, "one counter shard per 64-B line");

#ifdef __HIPCC__
__device__ __forceinline__ CountersDev *counter_shard(CountersDev *counters) {
  return counters + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) &
                     (CMI_COUNTER_SHARDS - 1));
}
#endif

#endif
