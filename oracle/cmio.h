/*
 * cmio.h - CPU ORACLE for the photon-packet transport + ionization-balance
 * hot path of CMacIonize.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, the smoke test
 * in __graft_entry__.py and the cpu_baseline leg of bench.py may load this
 * library. The shipped engine (cmacionize_amd/csrc, include/cmi_gpu.h) never
 * links, loads or calls it.
 *
 * It is a plain-C restatement of the reference's algorithm, function by
 * function; every function cites the reference file:line it follows (paths
 * relative to the reference's root). It is pinned against the reference's own
 * known-answer fixtures (tests/golden/, copied from the reference's test/
 * data files) - see tests/test_oracle_*.py. The reference itself cannot be
 * built in this image without running its cmake build (it needs the generated
 * Configuration.hpp and *DataLocation.hpp headers), so there is no
 * oracle/_ref binary; parity is pinned through the fixtures - and, end to
 * end, through a result of the reference itself: tests/golden/taskbased.hdf5
 * (the reference's test/taskbased.hdf5) is the final snapshot of its own
 * task-based run of the Stromgren benchmark at 16^3; the same run with this
 * oracle, in both transport semantics, gives its neutral fractions to Monte
 * Carlo noise (tests/test_reference_taskbased_snapshot.py).
 *
 * One deliberate difference from the reference: the random number generator.
 * The reference uses a sequential ranlxd2 stream per thread
 * (src/RandomGenerator.hpp:39-272), which cannot be reproduced by a
 * packet-parallel device kernel. Oracle and engine instead share a
 * counter-based Philox4x32-10 stream per packet (see cmio_rng_* below), so
 * that both generate bit-identical uniform numbers for packet p. Everything
 * downstream of the uniforms follows the reference.
 */
#ifndef CMIO_H
#define CMIO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ion order: src/ElementNames.hpp:101-154 */
enum {
  CMIO_ION_H_n = 0,
  CMIO_ION_He_n,
  CMIO_ION_C_p1,
  CMIO_ION_C_p2,
  CMIO_ION_N_n,
  CMIO_ION_N_p1,
  CMIO_ION_N_p2,
  CMIO_ION_O_n,
  CMIO_ION_O_p1,
  CMIO_ION_Ne_n,
  CMIO_ION_Ne_p1,
  CMIO_ION_S_p1,
  CMIO_ION_S_p2,
  CMIO_ION_S_p3,
  CMIO_NION = 14
};

/* element order: src/ElementNames.hpp (ELEMENT_H ... ELEMENT_S) */
enum {
  CMIO_EL_H = 0,
  CMIO_EL_He,
  CMIO_EL_C,
  CMIO_EL_N,
  CMIO_EL_O,
  CMIO_EL_Ne,
  CMIO_EL_S,
  CMIO_NELEMENT = 7
};

/* photon types: src/PhotonType.hpp:36-50 */
enum {
  CMIO_TYPE_PRIMARY = 0,
  CMIO_TYPE_DIFFUSE_HI,
  CMIO_TYPE_DIFFUSE_HeI,
  CMIO_TYPE_ABSORBED,
  CMIO_NTYPE = 4
};

enum {
  CMIO_SPECTRUM_MONOCHROMATIC = 0,
  CMIO_SPECTRUM_PLANCK = 1,
  CMIO_SPECTRUM_TABLE = 2
};
enum { CMIO_XSEC_FIXED = 0, CMIO_XSEC_VERNER = 1, CMIO_XSEC_TABLE = 2 };
enum { CMIO_RECOMB_FIXED = 0, CMIO_RECOMB_VERNER = 1, CMIO_RECOMB_TABLE = 2 };
enum { CMIO_TABLE_LINEAR = 0, CMIO_TABLE_LOGLOG = 1 };

/* A plugin that is known only through the reference's virtual
 * (PhotonSourceSpectrum::get_random_frequency, src/PhotonSourceSpectrum.hpp:
 * 48-50; CrossSections::get_cross_section, src/CrossSections.hpp:49-50;
 * RecombinationRates::get_recombination_rate, src/RecombinationRates.hpp:49),
 * sampled on a grid of its argument: n abscissae x[] in ascending order and
 * rows of n values y[] (1 row for a spectrum - x = cumulative distribution, y
 * = frequency -, 14 for cross sections / rates). cmio_table_value reads it
 * with Utilities::locate (src/Utilities.hpp:726-742) and linear or log-log
 * interpolation, end values outside. The arrays stay the caller's. */
typedef struct {
  const double *x;
  const double *y;
  int32_t n;
  int32_t interpolation;
} cmio_table;
double cmio_table_value(const cmio_table *table, int row, double x);
enum { CMIO_REEMIT_NONE = 0, CMIO_REEMIT_PHYSICAL = 1, CMIO_REEMIT_FIXED = 2 };

/* ---------------------------------------------------------------- RNG -- */

/* Philox4x32-10 (Salmon et al. 2011), the shared packet RNG. */
void cmio_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                        uint32_t out[4]);

/* draw number `draw` (0,1,2,...) of packet `packet` in iteration `iteration`
 * for seed `seed`: uniform double in the open interval (0,1).
 * Block b = draw/2 is Philox(ctr = {packet_lo, packet_hi, b, 0},
 * key = {seed, iteration}); draw&1 selects words {0,1} or {2,3};
 * u = ((hi:lo >> 12) + 0.5) * 2^-52 (exact in fp64). */
double cmio_rng_uniform(uint32_t seed, uint32_t iteration, uint64_t packet,
                        uint32_t draw);

int cmio_num_threads(void);
int cmio_thread_index(void);
void cmio_set_num_threads(int n);

/* Errors (cmio_error.c): the oracle never abort()s. Where the reference
 * raises cmac_error the first message is recorded, the function returns a
 * harmless value (NaN for a rate, nothing done for a loop) and the caller -
 * tests/oracle_lib.py after every call - asks for it here. */
void cmio_set_error(const char *fmt, ...)
    __attribute__((format(printf, 1, 2)));
const char *cmio_last_error(void); /* NULL: none since the last clear */
void cmio_clear_error(void);
/* test processes: the native backtrace of a thread that abort()s (a GPU
 * fault inside the HSA runtime, a heap check of glibc) goes to fd */
void cmio_install_abort_backtrace(int fd);

/* -------------------------------------------------------------- model -- */

/* Regular Cartesian grid: src/CartesianDensityGrid.cpp:40-95 */
typedef struct {
  double anchor[3];
  double sides[3];
  int32_t ncell[3];
  int32_t periodic[3];
} cmio_grid;

/* SoA cell state; mirrors src/IonizationVariables.hpp:81-118. All arrays have
 * ncell[0]*ncell[1]*ncell[2] entries, row-major (ix*ny*nz + iy*nz + iz),
 * src/CartesianDensityGrid.hpp:137-144. Owned by the caller. */
typedef struct {
  double *number_density;       /* m^-3 */
  double *temperature;          /* K */
  double *ionic_fraction[CMIO_NION];
  double *mean_intensity[CMIO_NION]; /* un-normalised sum ds*w*sigma (m^3) */
  double *heating[2];                /* H, He */
} cmio_cells;

/* Physics set-up: everything the reference's plugin objects hold. */
typedef struct {
  /* PhotonSource (src/PhotonSource.cpp:60-146): discrete sources only */
  int32_t nsource;
  const double *source_position;   /* [nsource][3] m */
  const double *source_cumulative; /* [nsource] cumulative weights, last = 1 */
  double total_luminosity;         /* s^-1 */

  /* PhotonSourceSpectrum */
  int32_t spectrum_type;
  double mono_frequency;      /* Hz */
  double planck_temperature;  /* K */

  /* CrossSections / RecombinationRates */
  int32_t xsec_type;
  double xsec_fixed[CMIO_NION];   /* m^2 */
  int32_t recomb_type;
  double recomb_fixed[CMIO_NION]; /* m^3 s^-1 */

  /* Abundances (src/Abundances.hpp), index = element, [H] unused */
  double abundance[CMIO_NELEMENT];

  /* DiffuseReemissionHandler */
  int32_t reemit_type;
  double reemit_fixed_probability; /* FixedValueDiffuseReemissionHandler */
  double reemit_fixed_frequency;   /* Hz */

  /* TemperatureCalculator parameters
   * (src/TemperatureCalculator.cpp:133-160) */
  int32_t do_temperature;
  int32_t t_min_iteration;  /* "minimum number of iterations" (3) */
  double t_epsilon;         /* 1e-3 */
  int32_t t_max_iterations; /* 100 */
  double pahfac, crfac, crlim, crscale;
  double t_min_ionized;     /* 4000 K */

  /* sampling tables built by cmio_tables_create (needed for the Planck
   * spectrum and for physical re-emission) */
  const struct cmio_tables *tables;

  /* ContinuousPhotonSource + ContinuousPhotonSourceSpectrum and the mix of
   * the two kinds of sources (src/PhotonSource.cpp:104-130), set by
   * cmio_mix_sources: continuous_type 0 = none, 1 =
   * IsotropicContinuousPhotonSource on `continuous_box`, 2 =
   * PlanarContinuousPhotonSource (the fields at the end of the struct) */
  int32_t continuous_type;
  int32_t continuous_spectrum_type;
  double continuous_mono_frequency;
  double continuous_planck_temperature;
  double continuous_box_anchor[3], continuous_box_sides[3];
  double discrete_luminosity, continuous_luminosity; /* inputs of the mix */
  double continuous_probability;
  double discrete_photon_weight, continuous_photon_weight;
  /* PlanarContinuousPhotonSource (src/PlanarContinuousPhotonSource.hpp): the
   * plane x[axis] = intercept, the rectangle [anchor, anchor + side] along the
   * two other axes in their natural order */
  int32_t continuous_axis;
  double continuous_intercept;
  double continuous_anchor[2], continuous_side[2];
  /* the *_TABLE types: [0] the discrete sources' spectrum, [1] the continuous
   * source's */
  cmio_table spectrum_table[2];
  cmio_table xsec_table;
  cmio_table recomb_table;
} cmio_model;

/* PhotonSource ctor, src/PhotonSource.cpp:104-130: total_luminosity,
 * continuous_probability and the two photon weights from
 * discrete_luminosity / continuous_luminosity */
void cmio_mix_sources(cmio_model *model);

#define CMIO_NFREQ 1000 /* frequency bins of every sampled spectrum */
#define CMIO_NTEMP 100  /* temperature bins of the Lyman continua */

/* src/PlanckPhotonSourceSpectrum.hpp, src/HydrogenLymanContinuumSpectrum.hpp,
 * src/HeliumLymanContinuumSpectrum.hpp,
 * src/HeliumTwoPhotonContinuumSpectrum.hpp: the members of these classes */
typedef struct cmio_tables {
  double planck_logfreq[CMIO_NFREQ];
  double planck_cdf[CMIO_NFREQ];
  double planck_logcdf[CMIO_NFREQ];
  double lyc_T[CMIO_NTEMP];
  double lyc_freq[2][CMIO_NFREQ];            /* [H, He] */
  double lyc_cdf[2][CMIO_NTEMP][CMIO_NFREQ]; /* [H, He][T][nu] */
  double he2pc_freq[CMIO_NFREQ];
  double he2pc_cdf[CMIO_NFREQ];
  /* the continuous source's Planck spectrum */
  double planck2_logfreq[CMIO_NFREQ];
  double planck2_cdf[CMIO_NFREQ];
  double planck2_logcdf[CMIO_NFREQ];
} cmio_tables;

/* builds every table from the model's spectrum and cross sections */
cmio_tables *cmio_tables_create(const cmio_model *model);
void cmio_tables_free(cmio_tables *tables);

/* n samples of spectrum `kind` (0 Planck, 1 H Lyc, 2 He Lyc, 3 He 2-photon)
 * at `temperature`; sample i uses packet stream i of `seed` */
void cmio_sample_spectrum(const cmio_model *model, int kind,
                          double temperature, uint32_t seed, uint64_t n,
                          double *out);
/* src/HeliumTwoPhotonContinuumSpectrum.cpp:144-160 */
double cmio_he2pc_integral(void);
/* src/PhysicalDiffuseReemissionHandler.hpp:66-105: p[0] = P(H Lyc),
 * p[1..4] = cumulative He channel probabilities */
void cmio_reemission_probabilities(double temperature, double p[5]);

/* PhysicalDiffuseReemissionHandler::reemit
 * (src/PhysicalDiffuseReemissionHandler.cpp:219-370) with the uniform random
 * numbers GIVEN by the caller instead of drawn from the packet stream: draw d
 * of the call is uniforms[d]. sigma_H / sigma_He are the packet's cross
 * sections. Returns the new frequency (0 = absorbed for good); *type receives
 * the photon type, *draws the number of uniforms consumed. For tests that
 * walk every branch of the handler. */
double cmio_reemit_scripted(const cmio_model *model, double sigma_H,
                            double sigma_He, double AHe, double T, double xH,
                            double xHe, const double *uniforms,
                            int32_t *type, uint32_t *draws);

/* SpectrumTrackers on cells (long indices) of the grid: while n != 0,
 * cmio_interact counts the packets that cross them - by frequency bin (nbins
 * over [1, 4) x 3.289e15 Hz) and photon type - into
 * counts[(k * 3 + type) * nbins + bin] (src/SpectrumTracker.hpp:176-212,
 * src/DensityGrid.hpp:188-191). The arrays stay the caller's. */
void cmio_set_trackers(int32_t n, int32_t nbins, const int64_t *cell,
                       const double *cos_opening_angle,
                       const double *direction, uint64_t *counts);
/* ... some of which (kind[k] != 0) are AbsorptionTrackers
 * (src/AbsorptionTracker.hpp:49-235): path length x cross section x weight of
 * every crossing packet into absorption[(k * 4 + type) * 14 + ion]. Call
 * after cmio_set_trackers (which resets the kinds). */
void cmio_set_tracker_kinds(const int32_t *kind, double *absorption);
/* ... each with its own number of bins (bins[n]; counts then tracker after
 * tracker, [3][bins[k]] each). Call after cmio_set_trackers. */
void cmio_set_tracker_bins(const int32_t *bins);
/* ... and some (kind[k] == 2) WeightedSpectrumTrackers
 * (src/WeightedSpectrumTracker.hpp:44-446): 1 / projected area per crossing in
 * flux[4 x (bins of the trackers before k) + type bins[k] + bin], the bin from
 * LinearFrequencyBins between bins_min[k] and bins_max[k] (bins_type[k] == 0)
 * or LevelFrequencyBins (1, 14 bins). Call after cmio_set_tracker_kinds. */
void cmio_set_tracker_weighted(double *flux, const int32_t *bins_type,
                               const double *bins_min, const double *bins_max);
/* WeightedSpectrumTracker::get_projected_area, :212-290 */
double cmio_projected_area(const double *direction);
/* FrequencyBins::get_bin_number (src/LinearFrequencyBins.hpp:115-125 for type
 * 0, src/LevelFrequencyBins.hpp:84-86 for type 1) */
int32_t cmio_frequency_bin(int32_t type, int32_t nbins, double minimum,
                           double maximum, double frequency);

/* A photon packet: src/Photon.hpp:36-69 */
typedef struct {
  double position[3];
  double direction[3];
  double inverse_direction[3];
  double energy; /* frequency, Hz */
  double cross_section[CMIO_NION];
  double cross_section_He_corr;
  double weight;
  int32_t type;
} cmio_photon;

/* ---------------------------------------------------------- transport -- */

/* src/CartesianDensityGrid.cpp:280-318 */
void cmio_wall_intersection(const double origin[3], const double direction[3],
                            const double inverse_direction[3],
                            const double cell_anchor[3],
                            const double cell_sides[3], int32_t next_index[3],
                            double *ds, double intersection[3]);

/* src/CartesianDensityGrid.cpp:375-452. Returns the long index of the cell
 * the photon was last in, or -1 if it left the box (DensityGrid::end()).
 * If trace_cell/trace_ds are non-NULL, up to trace_cap (cell, ds) pairs are
 * recorded and *trace_n receives the number of steps taken. */
int64_t cmio_interact(const cmio_grid *grid, const cmio_model *model,
                      cmio_cells *cells, cmio_photon *photon,
                      double optical_depth, int64_t *trace_cell,
                      double *trace_ds, int64_t trace_cap, int64_t *trace_n);

/* src/IonizationPhotonShootJob.hpp:117-146 for packets
 * [first_packet, first_packet + n_packets). Adds to totweight/typecount. */
void cmio_shoot(const cmio_grid *grid, const cmio_model *model,
                cmio_cells *cells, uint32_t seed, uint32_t iteration,
                uint64_t first_packet, uint64_t n_packets, double *totweight,
                double typecount[CMIO_NTYPE]);

/* The reference's TASK-BASED transport semantics (DensitySubGrid::interact,
 * src/DensitySubGrid.hpp:1137-1274, on nsub[0] x nsub[1] x nsub[2] subgrids)
 * for the same packets: see cmio_subgrid.c. The mean intensity of an ion of
 * element E comes out A_E times cmio_shoot's, the heating terms with the
 * thresholds 3.288e15 / 5.948e15 Hz. nsteps / handovers (may be NULL) count
 * cell crossings and subgrid changes. */
void cmio_subgrid_shoot(const cmio_grid *grid, const int32_t nsub[3],
                        const cmio_model *model, cmio_cells *cells,
                        uint32_t seed, uint32_t iteration,
                        uint64_t first_packet, uint64_t n_packets,
                        double *totweight, double typecount[CMIO_NTYPE],
                        uint64_t *nsteps, uint64_t *handovers);

/* The same packets with the same arithmetic, organised like the reference's
 * classic path (cells as an array of structures, one lock per cell; lock-free
 * single adds for hydrogen-only runs): the CPU BASELINE of bench.py. Equal to
 * cmio_shoot up to the order of the additions (cmio_transport_fast.c). */
void cmio_shoot_fast(const cmio_grid *grid, const cmio_model *model,
                     cmio_cells *cells, uint32_t seed, uint32_t iteration,
                     uint64_t first_packet, uint64_t n_packets,
                     double *totweight, double typecount[CMIO_NTYPE]);

/* Generate packet `packet` (src/PhotonSource.cpp:208-249) - also returns the
 * first optical depth tau = -ln(xi) and the number of draws consumed. */
void cmio_emit(const cmio_model *model, uint32_t seed, uint32_t iteration,
               uint64_t packet, cmio_photon *photon, double *tau,
               uint32_t *draws);

/* src/DensityGrid.hpp:803-807 */
void cmio_reset_grid(const cmio_grid *grid, cmio_cells *cells);

/* ----------------------------------------------------------- atomic data -- */

/* src/VernerCrossSections.cpp:259-322 (m^2; frequency in Hz) */
double cmio_verner_cross_section(int ion, double frequency);
/* src/VernerRecombinationRates.cpp:140-333 (m^3 s^-1) */
double cmio_verner_recombination_rate(int ion, double temperature);
/* src/ChargeTransferRates.cpp:44-157, :169-250, :262-395; T4 = T / 1e4 K */
double cmio_ct_recombination_rate_H(int ion, double T4);
double cmio_ct_ionization_rate_H(int ion, double T4);
double cmio_ct_recombination_rate_He(int ion, double T4);

/* --------------------------------------------------------- cell update -- */

/* src/IonizationStateCalculator.cpp:802-820 */
double cmio_ionization_state_hydrogen(double alphaH, double jH, double nH);

/* src/IonizationStateCalculator.cpp:649-753; returns 0, or 1 if the
 * reference would have hit cmac_error (more than 20 iterations). */
int cmio_ionization_states_hydrogen_helium(double alphaH, double alphaHe,
                                           double jH, double jHe, double nH,
                                           double AHe, double T, double *h0,
                                           double *he0);

/* src/IonizationStateCalculator.cpp:323-501; writes x[C_p1..S_p3] */
void cmio_ionization_states_metals(const cmio_model *model,
                                   const double j_metals[12], double ne,
                                   double T, double T4, double nh0, double nhe0,
                                   double nhp, double x[CMIO_NION]);

/* src/IonizationStateCalculator.cpp:70-272 for one cell; heating[2] is
 * normalised in place, x[14] receives the new ionic fractions. */
void cmio_ionization_state_cell(const cmio_model *model, double jfac,
                                double hfac, double ntot, double T,
                                const double J[CMIO_NION], double heating[2],
                                double x[CMIO_NION]);

/* src/IonizationStateCalculator.cpp:511-530 over the whole grid */
void cmio_calculate_ionization_state(const cmio_grid *grid,
                                     const cmio_model *model,
                                     cmio_cells *cells, double totweight);

/* ---------------------------------------------------------- line cooling -- */

/* src/LineCoolingData.cpp:1492-1555; A and B are overwritten, B = solution */
int cmio_solve_5x5(double A[5][5], double B[5]);
/* src/LineCoolingData.cpp:1767-1847: cooling rate per hydrogen atom (J s^-1
 * = kg m^2 s^-3); abundances[13] in the element order NI NII OI OII OIII NeIII
 * SII SIII CII CIII NIII NeII SIV (src/LineCoolingData.hpp:38-80) */
double cmio_line_cooling(double temperature, double electron_density,
                         const double abundances[13]);
/* accessors used to pin the data table (src/LineCoolingData.cpp:1410-1450) */
/* LineCoolingData::get_line_strengths (src/LineCoolingData.cpp:1859-1952):
 * out[10 e + t] = luminosity per hydrogen atom (J s^-1) of transition t
 * (0-1, 0-2, 0-3, 0-4, 1-2, 1-3, 1-4, 2-3, 2-4, 3-4) of five-level ion e,
 * out[100 + i] of the line of two-level ion i; 103 values */
void cmio_line_strengths(double temperature, double electron_density,
                         const double abundances[13], double *out);
/* EmissivityCalculator (src/EmissivityCalculator.cpp): see cmio_emissivity.c */
#define CMIO_NEMISSIONLINE 42
void cmio_balmer_jump(double T, double out[4]);
void cmio_emissivities(const cmio_model *model, double ntot, double T,
                       const double *x, double *out);
double cmio_lc_energy_difference(int element, int transition);
double cmio_lc_transition_probability(int element, int transition);
double cmio_lc_statistical_weight(int element, int level);

/* -------------------------------------------------------- thermal balance -- */

/* src/TemperatureCalculator.cpp:207-501; j[14], h[2] normalised integrals;
 * x[2..13] receive the metal fractions at temperature T */
void cmio_cooling_and_heating_balance(const cmio_model *model, double *h0,
                                      double *he0, double *gain, double *loss,
                                      double T, double n, double midpoint_z,
                                      const double j[CMIO_NION],
                                      const double h[2], double pahfac,
                                      double crfac, double crscale,
                                      double x[CMIO_NION]);
/* src/TemperatureCalculator.cpp:567-931 for one cell: J[14], heating[2]
 * un-normalised; *temperature in/out; x[14] in/out */
void cmio_temperature_cell(const cmio_model *model, double jfac, double hfac,
                           double ntot, double midpoint_z, double *temperature,
                           const double J[CMIO_NION], double heating[2],
                           double x[CMIO_NION]);
/* src/TemperatureCalculator.cpp:944-970, temperature branch */
void cmio_calculate_temperature(const cmio_grid *grid, const cmio_model *model,
                                cmio_cells *cells, double totweight);

/* src/TemperatureCalculator.cpp:944-970 (dispatch) over the whole grid:
 * ionization balance (src/IonizationStateCalculator.cpp:511-530,70-272) or
 * temperature solve (src/TemperatureCalculator.cpp:567-931). */
void cmio_update_cells(const cmio_grid *grid, const cmio_model *model,
                       cmio_cells *cells, uint32_t loop, double totweight);

/* the same three for cells [first, first + count) only: the `block` the
 * reference's MPI path gives a rank (src/IonizationSimulation.cpp:532-537) */
void cmio_calculate_ionization_state_range(const cmio_grid *grid,
                                           const cmio_model *model,
                                           cmio_cells *cells, double totweight,
                                           int64_t first, int64_t count);
void cmio_calculate_temperature_range(const cmio_grid *grid,
                                      const cmio_model *model,
                                      cmio_cells *cells, double totweight,
                                      int64_t first, int64_t count);
void cmio_update_cells_range(const cmio_grid *grid, const cmio_model *model,
                             cmio_cells *cells, uint32_t loop,
                             double totweight, int64_t first, int64_t count);

#ifdef __cplusplus
}
#endif

#endif /* CMIO_H */
