/*
 * cmi_gpu.h - C ABI of the MI355X photoionization engine (libcmi_gpu.so).
 *
 * This is the drop-in boundary for ONE path of CMacIonize: photon-packet
 * transport through a regular Cartesian grid plus the per-cell ionization /
 * temperature balance, i.e. the body of the iteration loop of
 * IonizationSimulation::run (reference src/IonizationSimulation.cpp:359-643):
 *
 *     reset_grid -> shoot N packets -> [reduce] -> calculate_temperature
 *
 * Everything above that loop (parameter file, plugin objects, writers) stays
 * on the host; the plugins are lowered once, at initialisation, into the flat
 * descriptors passed through the cmi_gpu_set_* calls. Every entry point cites
 * the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - plain C types only: opaque handle, pointers and sizes;
 *  - every function returns 0 on success and a non-zero CMI_GPU_E* code on
 *    failure; cmi_gpu_last_error() gives the message (thread local). Nothing
 *    aborts (the reference's cmac_error aborts, src/Error.hpp:101-110);
 *  - "host" pointers are caller-owned host memory, copied during the call;
 *    "device" pointers are HIP device memory valid on the engine's device;
 *  - all quantities in SI units, fp64, ion order of src/ElementNames.hpp:101-154
 *    (H0 He0 C+ C2+ N0 N+ N2+ O0 O+ Ne0 Ne+ S+ S2+ S3+), cells row-major
 *    ix*ny*nz + iy*nz + iz (src/CartesianDensityGrid.hpp:137-144);
 *  - one host thread per handle; work is enqueued on the handle's HIP stream
 *    and is asynchronous unless stated otherwise.
 *  - there is no CPU fallback: without a HIP device cmi_gpu_create fails.
 */
#ifndef CMI_GPU_H
#define CMI_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMI_GPU_NION 14
#define CMI_GPU_NHEATING 2
#define CMI_GPU_NTYPE 4 /* src/PhotonType.hpp:36-50 */
/* mean-intensity + heating accumulators per cell (src/DensityGrid.hpp:150-197) */
#define CMI_GPU_NACC (CMI_GPU_NION + CMI_GPU_NHEATING)

enum {
  CMI_GPU_OK = 0,
  CMI_GPU_EINVAL = 1,   /* bad argument */
  CMI_GPU_EDEVICE = 2,  /* HIP runtime error / no device */
  CMI_GPU_ESTATE = 3,   /* call sequence error (e.g. shoot before sources) */
  CMI_GPU_ENOMEM = 4
};

/* field ids for upload / download / device pointers */
enum {
  CMI_GPU_FIELD_NUMBER_DENSITY = 0, /* IonizationVariables::_number_density */
  CMI_GPU_FIELD_TEMPERATURE = 1,    /* ::_temperature */
  CMI_GPU_FIELD_IONIC_FRACTION = 2, /* + ion (14 fields) ::_ionic_fractions */
  CMI_GPU_FIELD_MEAN_INTENSITY = 16, /* + ion (14 fields) ::_mean_intensity */
  CMI_GPU_FIELD_HEATING = 30,        /* + 0 (H), 1 (He)  ::_heating */
  CMI_GPU_NFIELD = 32
};

enum {
  CMI_GPU_SPECTRUM_MONOCHROMATIC = 0,
  CMI_GPU_SPECTRUM_PLANCK = 1,
  CMI_GPU_SPECTRUM_TABLE = 2 /* cmi_gpu_set_spectrum_table */
};
/* which PhotonSourceSpectrum of the run a table stands for
 * (src/IonizationSimulation.cpp:155-168: role "PhotonSourceSpectrum" of the
 * discrete sources, role "ContinuousPhotonSourceSpectrum") */
enum { CMI_GPU_ROLE_SOURCE = 0, CMI_GPU_ROLE_CONTINUOUS = 1 };
/* how a table is read between two of its samples: linearly, or linearly in
 * the logarithms (a power law through the two samples) */
enum { CMI_GPU_TABLE_LINEAR = 0, CMI_GPU_TABLE_LOGLOG = 1 };
enum { CMI_GPU_REEMIT_NONE = 0, CMI_GPU_REEMIT_PHYSICAL = 1,
       CMI_GPU_REEMIT_FIXED = 2 };

typedef struct cmi_gpu_engine cmi_gpu_engine;

/* Geometry of the CartesianDensityGrid (src/CartesianDensityGrid.cpp:40-95:
 * box anchor/sides, number of cells, periodicity flags) + engine options. */
typedef struct {
  double anchor[3];     /* SimulationBox:anchor (m) */
  double sides[3];      /* SimulationBox:sides (m) */
  int32_t ncell[3];     /* DensityGrid:number of cells */
  int32_t periodic[3];  /* SimulationBox:periodicity */
  int32_t device;       /* HIP device ordinal */
  /* 1: also accumulate the two heating integrals during transport. The
   * reference always does (src/DensityGrid.hpp:170-186); they are only used
   * by the temperature solve, so a run with "do temperature calculation:
   * false" may pass 0 and save two atomics per step. */
  int32_t track_heating;
  /* optional: HIP stream (hipStream_t) to enqueue on; NULL = own stream */
  void *stream;
  /* optional: caller-allocated device buffer for the CMI_GPU_NACC accumulator
   * fields, contiguous [CMI_GPU_NACC][ncell] doubles (so that the caller can
   * hand it to a collective, e.g. torch.distributed over RCCL); NULL = the
   * engine allocates it. The engine zeroes it at creation; in hydrogen-only
   * runs cmi_gpu_reset_grid clears only the fields such a run adds to (J_H,
   * the heating terms if tracked), so a caller that writes into the block
   * itself - other than sums of what engines wrote there - declares that
   * with cmi_gpu_set_tuning("accumulators_dirty", 1): the next reset then
   * clears the whole block (cmi_gpu_upload_field of an accumulator field
   * does the same by itself). */
  void *external_accumulators;
  /* optional domain decomposition (replaces DensitySubGridCreator's block
   * decomposition, src/DensitySubGridCreator.hpp:314-396): the engine holds
   * only the block of sub_ncell cells starting at cell sub_offset of the grid
   * described above; all zero = the whole grid. Every field, upload and
   * download of the engine then refers to the block's cells (row-major inside
   * the block). Packets that leave the block into another one are handed over
   * through cmi_gpu_set_export_buffer / cmi_gpu_shoot_flights. */
  int32_t sub_offset[3];
  int32_t sub_ncell[3];
} cmi_gpu_config;

/* ------------------------------------------------------------ lifetime -- */

/* replaces: DensityGridFactory::generate + CartesianDensityGrid ctor
 * (src/DensityGridFactory.hpp:73-77, src/CartesianDensityGrid.cpp:40-95) */
int cmi_gpu_create(const cmi_gpu_config *config, cmi_gpu_engine **engine);
int cmi_gpu_destroy(cmi_gpu_engine *engine);
const char *cmi_gpu_last_error(void);
/* blocks until all enqueued work is done */
int cmi_gpu_synchronize(cmi_gpu_engine *engine);
int64_t cmi_gpu_number_of_cells(const cmi_gpu_engine *engine);

/* ------------------------------------------------- plugin descriptors -- */

/* replaces: PhotonSourceDistribution::{get_number_of_sources, get_position,
 * get_weight, get_total_luminosity} as consumed by the PhotonSource ctor
 * (src/PhotonSourceDistribution.hpp:54-80, src/PhotonSource.cpp:60-146).
 * positions: host [n][3] (m); weights: host [n], must sum to 1 within 1e-9
 * (same check as the reference); total_luminosity in s^-1. n = 0 removes the
 * discrete sources (a run with a continuous source only). */
int cmi_gpu_set_sources(cmi_gpu_engine *engine, int32_t n,
                        const double *positions, const double *weights,
                        double total_luminosity);

/* replaces: ContinuousPhotonSource as consumed by the PhotonSource ctor and
 * get_random_photon (src/ContinuousPhotonSource.hpp,
 * src/PhotonSource.cpp:104-130,230-238). ISOTROPIC is
 * IsotropicContinuousPhotonSource on the simulation box
 * (src/IsotropicContinuousPhotonSource.hpp:95-191). luminosity (s^-1) is what
 * the PhotonSource ctor computes: get_total_luminosity(), or
 * get_total_surface_area() x the spectrum's get_total_flux(). With both kinds
 * of sources half of the packets come from each and the continuous ones carry
 * the weight L_continuous / L_discrete; all tallies (mean intensities, heating,
 * totweight, the per-type counts) are sums of weights, as in the reference. */
enum {
  CMI_GPU_CONTINUOUS_NONE = 0,
  CMI_GPU_CONTINUOUS_ISOTROPIC = 1,
  CMI_GPU_CONTINUOUS_PLANAR = 2
};
int cmi_gpu_set_continuous_source(cmi_gpu_engine *engine, int32_t type,
                                  double luminosity);
/* ... PlanarContinuousPhotonSource (src/PlanarContinuousPhotonSource.hpp:
 * 96-196): packets start on the rectangle [anchor, anchor + sides] (host [2],
 * along the two axes other than `axis`, in their natural order) of the plane
 * x[axis] = intercept, in an isotropic direction; it has its own luminosity
 * (has_total_luminosity()). */
int cmi_gpu_set_continuous_source_planar(cmi_gpu_engine *engine, int32_t axis,
                                         double intercept,
                                         const double *anchor,
                                         const double *sides,
                                         double luminosity);
/* the continuous source's PhotonSourceSpectrum (role
 * "ContinuousPhotonSourceSpectrum", src/IonizationSimulation.cpp:164-168) */
int cmi_gpu_set_continuous_spectrum_monochromatic(cmi_gpu_engine *engine,
                                                  double frequency);
int cmi_gpu_set_continuous_spectrum_planck(cmi_gpu_engine *engine,
                                           double temperature);

/* replaces: PhotonSourceSpectrum::get_random_frequency for
 * MonochromaticPhotonSourceSpectrum (src/MonochromaticPhotonSourceSpectrum.hpp:97-100) */
int cmi_gpu_set_spectrum_monochromatic(cmi_gpu_engine *engine,
                                       double frequency);
/* ... for PlanckPhotonSourceSpectrum (src/PlanckPhotonSourceSpectrum.cpp:53-113,149-165) */
int cmi_gpu_set_spectrum_planck(cmi_gpu_engine *engine, double temperature);

/* GENERIC LOWERING ("unknown plugin => sample the virtual on the
 * host into a table"). A PhotonSourceSpectrum, CrossSections or
 * RecombinationRates implementation that is known only through the
 * reference's virtual - a third-party plugin, or one of the reference's
 * data-file spectra - is evaluated by the host ONCE, at initialisation, on a
 * grid of its argument, and the device reads the table. The three entry
 * points below take such tables (host arrays, copied); host/Plugins.hpp's base
 * classes build them by default from the virtuals alone.
 *
 * replaces: PhotonSourceSpectrum::get_random_frequency
 * (src/PhotonSourceSpectrum.hpp:48-50) of ANY spectrum, given as its quantile
 * function: cumulative[n] strictly ascending from 0 to 1, frequency[n] (Hz)
 * the frequency below which that fraction of the photons lies. A packet draws
 * one uniform x, Utilities::locate finds its interval of cumulative[]
 * (src/Utilities.hpp:726-742) and the frequency is interpolated between the
 * interval's ends - CMI_GPU_TABLE_LINEAR as
 * src/HeliumTwoPhotonContinuumSpectrum.cpp:167-180 does, CMI_GPU_TABLE_LOGLOG
 * as src/PlanckPhotonSourceSpectrum.cpp:149-165 does. role: CMI_GPU_ROLE_*. */
int cmi_gpu_set_spectrum_table(cmi_gpu_engine *engine, int32_t role, int32_t n,
                               const double *frequency,
                               const double *cumulative,
                               int32_t interpolation);
/* replaces: CrossSections::get_cross_section (src/CrossSections.hpp:49-50) of
 * ANY implementation: frequency[n] (Hz, strictly ascending) and
 * sigma[14][n] (m^2; ion-major, the ions in the order of
 * src/ElementNames.hpp:101-154). Between two samples the cross section is
 * interpolated (CMI_GPU_TABLE_*; a log-log interval with a zero in it falls
 * back to linear), outside the table it keeps the end values: a jump at an
 * ionization threshold is two neighbouring samples. The run then carries all
 * 14 cross sections per packet, as with VernerCrossSections. */
int cmi_gpu_set_cross_sections_table(cmi_gpu_engine *engine, int32_t n,
                                     const double *frequency,
                                     const double *sigma,
                                     int32_t interpolation);
/* replaces: RecombinationRates::get_recombination_rate
 * (src/RecombinationRates.hpp:49) of ANY implementation: temperature[n] (K,
 * strictly ascending), alpha[14][n] (m^3 s^-1, indexed by the recombined
 * ion). */
int cmi_gpu_set_recombination_rates_table(cmi_gpu_engine *engine, int32_t n,
                                          const double *temperature,
                                          const double *alpha,
                                          int32_t interpolation);

/* replaces: CrossSections::get_cross_section for FixedValueCrossSections
 * (src/FixedValueCrossSections.hpp:151-154); sigma: host [14] (m^2) */
int cmi_gpu_set_cross_sections_fixed(cmi_gpu_engine *engine,
                                     const double *sigma);
/* ... for VernerCrossSections (src/VernerCrossSections.cpp:259-322) */
int cmi_gpu_set_cross_sections_verner(cmi_gpu_engine *engine);

/* replaces: RecombinationRates::get_recombination_rate for
 * FixedValueRecombinationRates (src/FixedValueRecombinationRates.hpp:157-160);
 * alpha: host [14] (m^3 s^-1), indexed by the recombined ion */
int cmi_gpu_set_recombination_rates_fixed(cmi_gpu_engine *engine,
                                          const double *alpha);
/* ... for VernerRecombinationRates (src/VernerRecombinationRates.cpp:140-333) */
int cmi_gpu_set_recombination_rates_verner(cmi_gpu_engine *engine);

/* replaces: Abundances (src/Abundances.hpp) as filled by
 * FixedValueAbundanceModel (src/FixedValueAbundanceModel.hpp:44-52);
 * abundances: host [6] = He, C, N, O, Ne, S relative to H */
int cmi_gpu_set_abundances(cmi_gpu_engine *engine, const double *abundances);

/* replaces: DiffuseReemissionHandlerFactory::generate
 * (src/DiffuseReemissionHandlerFactory.hpp:94-99): NONE, PHYSICAL
 * (src/PhysicalDiffuseReemissionHandler.cpp) or FIXED
 * (src/FixedValueDiffuseReemissionHandler.hpp, needs probability+frequency) */
int cmi_gpu_set_reemission(cmi_gpu_engine *engine, int32_t type,
                           double fixed_probability, double fixed_frequency);

/* replaces: TemperatureCalculator ctor parameters
 * (src/TemperatureCalculator.cpp:133-160) */
typedef struct {
  int32_t do_temperature_calculation; /* default 0 */
  int32_t minimum_number_of_iterations; /* 3 */
  double epsilon_convergence;           /* 1e-3 */
  int32_t maximum_number_of_iterations; /* 100 */
  double pah_heating_factor;            /* 0 */
  double cosmic_ray_heating_factor;     /* 0 */
  double cosmic_ray_heating_limit;      /* 0.75 */
  double cosmic_ray_heating_scale_length; /* 1.33333 kpc in m */
  double minimum_ionized_temperature;   /* 4000 K */
} cmi_gpu_temperature_params;
int cmi_gpu_set_temperature_params(cmi_gpu_engine *engine,
                                   const cmi_gpu_temperature_params *params);

/* ----------------------------------------------------------- cell data -- */

/* replaces: DensityGrid::set_densities / DensityGridInitializationFunction
 * (src/DensityGrid.cpp:40-62, src/DensityGrid.hpp:775-790): the host
 * evaluates DensityFunction::operator() per cell and uploads SoA arrays.
 * number_density, temperature: host [ncell]; ionic_fractions: host
 * [14][ncell] or NULL (= all zero). Synchronous. */
int cmi_gpu_upload_cells(cmi_gpu_engine *engine, const double *number_density,
                         const double *temperature,
                         const double *ionic_fractions);
/* single field, host [ncell]; synchronous */
int cmi_gpu_upload_field(cmi_gpu_engine *engine, int32_t field,
                         const double *values);
/* replaces: the DensityGrid iterator accessors a DensityGridWriter reads
 * (src/DensityGridWriter.hpp:96-124); host [ncell]; synchronous */
int cmi_gpu_download_field(cmi_gpu_engine *engine, int32_t field,
                           double *values);
/* device address of the first element of a field. State fields are [ncell]
 * contiguous doubles. The 16 accumulator fields (MEAN_INTENSITY+0..13,
 * HEATING+0..1) share ONE contiguous block of 16 * ncell doubles starting at
 * the pointer of MEAN_INTENSITY+0 - that block is what a multi-process caller
 * sum-reduces; inside it element (field f, cell c) is at
 * (this function's pointer for f) + c * cell_stride, with the strides reported
 * by cmi_gpu_accumulator_layout: [16][ncell] for hydrogen-only transport
 * (field f at f * field_stride); [ncell][16] when all ions are transported,
 * a cell's 16 values then in the order of their ionization thresholds - J of
 * H0 O0 N0, the hydrogen heating term, J of Ne0 S+ C+ N+ | J of He0, the
 * helium heating term, J of S++ O+ Ne+ S+++ N++ C++ - so that a photon below
 * 24.59 eV touches one 64-B line of the row, not two (field_stride 1 tells
 * the layout, not the field's offset: ask this function per field). */
void *cmi_gpu_field_device_pointer(cmi_gpu_engine *engine, int32_t field);
int cmi_gpu_accumulator_layout(cmi_gpu_engine *engine, int64_t *field_stride,
                               int64_t *cell_stride);

/* ------------------------------------------------- the iteration body -- */

/* replaces: DensityGrid::reset_grid (src/DensityGrid.hpp:803-807) and zeroes
 * the packet counters (totweight, typecount, step counter) */
int cmi_gpu_reset_grid(cmi_gpu_engine *engine);

/* replaces: WorkDistributor::do_in_parallel(IonizationPhotonShootJobMarket) =
 * IonizationPhotonShootJob::execute for packets [first_packet, first_packet +
 * n_packets) of iteration `iteration` (src/IonizationSimulation.cpp:402,
 * src/IonizationPhotonShootJob.hpp:117-146). Packet p draws its random
 * numbers from Philox4x32-10(counter = {p, draw block}, key = {seed,
 * iteration}), so any partition of the packet range over calls / devices
 * gives the same packets. Accumulates into the mean-intensity (+heating)
 * fields and the counters. Asynchronous on the engine's stream without
 * re-emission; with re-emission passes (tuning reemit_passes = 1, the
 * default) the host reads the size of each generation's queue of re-emitted
 * packets back, so the call returns when the last generation is queued. */
int cmi_gpu_shoot(cmi_gpu_engine *engine, uint32_t seed, uint32_t iteration,
                  uint64_t first_packet, uint64_t n_packets);

/* ---- decomposed grids: the photon-buffer exchange of the task-based path
 * (PhotonTraversalTaskContext / MemorySpace / MPI photon buffers,
 * src/PhotonTraversalTaskContext.hpp:100-278). A flight that leaves the
 * engine's block into another block is appended to the export buffer as
 * CMI_GPU_FLIGHT_DOUBLES doubles:
 *   [0-2] origin, [3-5] direction, [6] path parameter, [7-9] next wall
 *   parameters, [10] optical depth left, [11] frequency,
 *   [12] int64: long index of the cell it enters, in the WHOLE grid,
 *   [13] 2 x uint32: packet id (relative to first_packet), rng position/type.
 * The caller moves the rows to the engine that owns that cell (any transport:
 * RCCL all-to-all, peer copies) and continues them there with
 * cmi_gpu_shoot_flights, which also follows their re-emissions and may export
 * again. The marcher's own state travels, so a packet's path lengths are
 * bit-identical to a run on the undivided grid. ---- */
#define CMI_GPU_FLIGHT_DOUBLES 16
/* device buffer [capacity][CMI_GPU_FLIGHT_DOUBLES] doubles, caller-owned
 * (device_rows == NULL: the engine allocates one of that capacity, for hosts
 * that exchange through host memory); resets the export count */
int cmi_gpu_set_export_buffer(cmi_gpu_engine *engine, void *device_rows,
                              uint64_t capacity);
/* flights exported since the last reset; fails with CMI_GPU_ENOMEM if more
 * left the block than the buffer holds. Synchronous. */
int cmi_gpu_get_export_count(cmi_gpu_engine *engine, uint64_t *count);
int cmi_gpu_reset_exports(cmi_gpu_engine *engine);
/* the same through host memory, for a host without a GPU-aware transport (the
 * reference's MPI photon buffers are host memory): copy the exported rows to
 * host_rows [capacity][CMI_GPU_FLIGHT_DOUBLES] (*count = their number) and
 * continue flights given in host memory. Synchronous copies. */
int cmi_gpu_download_exports(cmi_gpu_engine *engine, double *host_rows,
                             uint64_t capacity, uint64_t *count);
int cmi_gpu_shoot_flights_host(cmi_gpu_engine *engine, uint32_t seed,
                               uint32_t iteration, uint64_t first_packet,
                               const double *host_rows, uint64_t n_flights);
/* continue n_flights handed-over flights (device rows as above); seed,
 * iteration and first_packet as in the cmi_gpu_shoot call that emitted them */
int cmi_gpu_shoot_flights(cmi_gpu_engine *engine, uint32_t seed,
                          uint32_t iteration, uint64_t first_packet,
                          const void *device_rows, uint64_t n_flights);

/* replaces: IonizationPhotonShootJobMarket::update_counters
 * (src/IonizationSimulation.cpp:406): totweight and typecount[4] summed over
 * all shoot calls since the last reset; nsteps = number of cell crossings
 * (DDA steps) executed. Synchronous. Any pointer may be NULL. */
int cmi_gpu_get_counters(cmi_gpu_engine *engine, double *totweight,
                         double *typecount, uint64_t *nsteps);

/* number of atomic adds the transport kernels issued to the accumulator
 * fields since the last reset (diagnostic of the cross-lane aggregation) */
int cmi_gpu_get_atomic_count(cmi_gpu_engine *engine, uint64_t *natomics);

/* iterations of the transport kernel's march loop summed over all wavefronts
 * since the last reset: DDA steps / (64 x this) is the fraction of lanes that
 * did a step in an average iteration (diagnostic of the packet ordering) */
int cmi_gpu_get_wave_steps(cmi_gpu_engine *engine, uint64_t *nwavesteps);

/* replaces: TemperatureCalculator::calculate_temperature(loop, totweight,
 * grid, block) (src/TemperatureCalculator.cpp:944-970), i.e. per cell either
 * IonizationStateCalculator::calculate_ionization_state
 * (src/IonizationStateCalculator.cpp:70-272) or the temperature solve
 * (src/TemperatureCalculator.cpp:567-931). Reads the (reduced) accumulators,
 * writes ionic fractions (+temperature) and the transport opacities.
 * Asynchronous for the ionization balance; the temperature solve runs as a
 * pipeline of kernels that reads one count back per secant step and returns
 * with the last kernel enqueued (tuning "temperature_pipeline"). */
int cmi_gpu_update_cells(cmi_gpu_engine *engine, uint32_t loop,
                         double totweight);
/* Rebuild the transport records {n x_H, n x_He} of all cells from the state
 * fields - after number density or the H / He neutral fractions were written
 * through cmi_gpu_field_device_pointer (e.g. by the gather that follows a
 * sharded cell update, src/IonizationSimulation.cpp:540-618).
 * Asynchronous. */
int cmi_gpu_refresh_transport_records(cmi_gpu_engine *engine);

/* the same for the cells [first_cell, first_cell + ncell) of the engine's
 * grid only - the `block` argument of
 * TemperatureCalculator::calculate_temperature: in the reference's MPI path
 * every rank solves its block of cells and the new state is gathered
 * (src/IonizationSimulation.cpp:532-618, MPICommunicator::distribute_block,
 * src/MPICommunicator.hpp:224-239). Asynchronous. */
int cmi_gpu_update_cells_range(cmi_gpu_engine *engine, uint32_t loop,
                               double totweight, int64_t first_cell,
                               int64_t ncell);

/* replaces: EmissivityCalculator::calculate_emissivities over a block of the
 * grid (src/EmissivityCalculator.cpp:126-430 per cell, :439-470 over the
 * grid; LineCoolingData::get_line_strengths, src/LineCoolingData.cpp:1859-1952,
 * and EmissivityCalculator::get_balmer_jump_emission, :42-116, under it).
 * lines[nlines] picks emission lines by their index in EmissivityValues
 * (src/EmissivityValues.hpp:36-81; CMI_GPU_NUMBER_OF_EMISSIONLINES of them),
 * emissivities[k * ncell + c] receives line lines[k] of cell first_cell + c
 * (J m^-3 s^-1; the avg_* entries are the reference's weights; all zero in a
 * cell with x_H >= 0.2 or T <= 3000 K, :134). Needs the abundances
 * (cmi_gpu_set_abundances) and a full-ion state; reads the cells as they are
 * on the device. Synchronous: returns with the host array filled. */
#define CMI_GPU_NUMBER_OF_EMISSIONLINES 42
int cmi_gpu_compute_emissivities(cmi_gpu_engine *engine, int32_t nlines,
                                 const int32_t *lines, int64_t first_cell,
                                 int64_t ncell, double *emissivities);

/* replaces: TrackerManager::add_trackers with SpectrumTrackers
 * (src/TrackerManager.hpp:178-205, src/SpectrumTracker.hpp:41-262) and the
 * hook in DensityGrid::update_integrals (src/DensityGrid.hpp:188-191): every
 * packet that crosses the cell holding positions[3 k ..] (with gas in it) is
 * counted in tracker k by frequency bin (nbins bins over [1, 4) x 3.289e15 Hz,
 * :88-90) and photon type (primary, diffuse H, diffuse He) - if
 * reference_directions[3 k ..] is not the null vector only packets within
 * opening_angles[k] (radians) of it (:178-186). At most 16 trackers; n = 0
 * removes them; opening_angles / reference_directions may be NULL (all
 * packets). Counting happens while enabled (the reference adds its trackers
 * for the last iteration, src/IonizationSimulation.cpp:367-370) and makes the
 * transport run without the combining table and without tile rounds - in the
 * exact marcher on an undivided grid, in the incremental one on a block of a
 * decomposed grid (flights handed over between blocks carry its state). */
int cmi_gpu_set_spectrum_trackers(cmi_gpu_engine *engine, int32_t n,
                                  const double *positions, int32_t nbins,
                                  const double *opening_angles,
                                  const double *reference_directions);
/* The same with a number of bins per tracker (nbins[n]: the reference's
 * trackers each have their own, `number of bins` in the block file) and a
 * kind per tracker (NULL: all spectrum trackers):
 * CMI_GPU_TRACKER_SPECTRUM as above, CMI_GPU_TRACKER_ABSORPTION an
 * AbsorptionTracker (src/AbsorptionTracker.hpp:49-235, the hook of
 * DensitySubGrid::update_intensity_counters, src/DensitySubGrid.hpp:592-617):
 * for every packet crossing the cell, path length x cross section x weight
 * per ion, summed by photon type - the cell's mean-intensity sums split by
 * type (m^3; cmi_gpu_get_tracker_absorption). The engine's cross sections
 * are the classic path's (no element abundance in them: the reference's
 * task-based packets carry A_element sigma, src/SourceDiscretePhotonTask
 * Context.hpp:172-180 - multiply the ion's column by its element's abundance
 * for that convention). An engine whose cross sections are FixedValue with
 * sigma = 0 for every ion but H0 runs the hydrogen-only kernels, which add
 * only the H0 column: the other thirteen are path length x 0 in the reference
 * as well. On a block of a decomposed grid a tracker outside
 * the block counts nothing; the caller adds the blocks' (and copies') counts
 * (TrackerManager::normalize merges copies, src/TrackerManager.hpp:307-318).
 *
 * CMI_GPU_TRACKER_WEIGHTED_SPECTRUM is a WeightedSpectrumTracker
 * (src/WeightedSpectrumTracker.hpp:44-446): every packet crossing the cell
 * adds 1 / (the area the unit cube shows along the packet's direction,
 * get_projected_area, :212-290) to the bin of its frequency, by photon type
 * (cmi_gpu_get_tracker_flux). Its nbins[k] bins are LinearFrequencyBins from
 * 13.6 eV to 54.4 eV (src/LinearFrequencyBins.hpp:80-88) until
 * cmi_gpu_set_tracker_frequency_bins says otherwise; opening angle and
 * reference direction are not used. */
#define CMI_GPU_TRACKER_SPECTRUM 0
#define CMI_GPU_TRACKER_ABSORPTION 1
#define CMI_GPU_TRACKER_WEIGHTED_SPECTRUM 2
int cmi_gpu_set_trackers(cmi_gpu_engine *engine, int32_t n,
                         const double *positions, const int32_t *kinds,
                         const int32_t *nbins, const double *opening_angles,
                         const double *reference_directions);
int cmi_gpu_enable_trackers(cmi_gpu_engine *engine, int32_t enable);
/* replaces: FrequencyBinsFactory::generate for a WeightedSpectrumTracker
 * (src/FrequencyBinsFactory.hpp:57-72, `FrequencyBins:type`).
 * CMI_GPU_FREQUENCY_BINS_LINEAR: the tracker's nbins bins between
 * minimum_frequency and maximum_frequency (Hz), lower frequencies counted in
 * the first bin and higher ones in the last (LinearFrequencyBins::
 * get_bin_number, src/LinearFrequencyBins.hpp:115-125).
 * CMI_GPU_FREQUENCY_BINS_LEVEL: one bin per ion, from its ionization energy
 * (src/ElementData.hpp:39-105) to the next higher one, the last up to four
 * times hydrogen's (src/LevelFrequencyBins.hpp:52-86; the tracker must have
 * been set with 14 bins; the two frequencies are not used). After
 * cmi_gpu_set_trackers, before packets fly. */
#define CMI_GPU_FREQUENCY_BINS_LINEAR 0
#define CMI_GPU_FREQUENCY_BINS_LEVEL 1
int cmi_gpu_set_tracker_frequency_bins(cmi_gpu_engine *engine, int32_t tracker,
                                       int32_t type, double minimum_frequency,
                                       double maximum_frequency);
/* The weighted trackers' sums since the trackers were set, tracker after
 * tracker: tracker k's at flux[4 first_k + type nbins_k + bin], first_k = the
 * bins of the trackers before it, type in the order of
 * src/PhotonType.hpp:36-50 (zero rows for the other kinds of tracker). Not
 * normalised (WeightedSpectrumTracker::normalize multiplies by luminosity /
 * total weight / the cell's side squared, :106-116). Synchronous. */
int cmi_gpu_get_tracker_flux(cmi_gpu_engine *engine, double *flux);
/* probe: WeightedSpectrumTracker::get_projected_area for n directions
 * ([n][3], unit vectors), evaluated on the host by the function the kernels
 * call (test/testWeightedSpectrumTracker.cpp's known answers) */
int cmi_gpu_projected_areas(const double *directions, int64_t n,
                            double *areas);
/* absorption[(k * 4 + type) * 14 + ion] since the trackers were set, type in
 * the order of src/PhotonType.hpp:36-50 (the row of PHOTONTYPE_ABSORBED stays
 * zero: no packet flies with that type). Not normalised
 * (AbsorptionTracker::normalize multiplies by luminosity / total weight).
 * Synchronous. */
int cmi_gpu_get_tracker_absorption(cmi_gpu_engine *engine,
                                   double *absorption);
/* The counts since the trackers were set, tracker after tracker: tracker k's
 * at counts[3 first_k + type nbins_k + bin], first_k = the bins of the
 * trackers before it (with one bin count for all: counts[(k * 3 + type) *
 * nbins + bin]; SpectrumTracker::output_tracker's three columns, :226-238).
 * Synchronous. */
int cmi_gpu_get_tracker_counts(cmi_gpu_engine *engine, uint64_t *counts);

/* Performance knobs (no effect on what is computed, only on how):
 *   "sort_packets" (1)      process the packets of a launch in emission-
 *                           direction order, so that the lanes of a wave cross
 *                           the same cells
 *   "sort_tau_bits" (-1)    hydrogen-only transport: split every coarse
 *                           direction bin into 2^bits classes of the first
 *                           optical depth, so that the packets of a wave end
 *                           their flights at about the same step (0 = plain
 *                           direction order, -1 = chosen from the number of
 *                           packets per launch)
 *   "sort_dir_bits" (-1)    direction bits of the sort key, 2 .. 22 (an equal-area
 *                           2048 x 2048 lattice, Morton order); -1 = auto: 22,
 *                           or one or two fewer where that saves the radix
 *                           sort a pass
 *   "aggregate" (2)         what happens to a step's contributions before
 *                           HBM sees an atomic: 0 = one atomic per lane and
 *                           step; 1 = lanes of a wave in the same cell are
 *                           summed first; 2 = + per-block combining table in
 *                           LDS, written back between ray bundles (hydrogen-only
 *                           transport; multi-ion transport uses its own
 *                           cooperative scheme for any value > 0)
 *   "aggregate_reemit" (0)  the same for the later re-emission passes
 *   "refill_threshold" (64) idle lanes of a wave that trigger a refill
 *   "chunk" (64)            consecutive packets a wave takes at a time
 *   "max_blocks_per_cu" (8), "max_packets_per_launch" (2^27)
 *   "reemit_passes" (1)     with diffuse re-emission: absorbed packets are
 *                           parked in a queue, a separate interaction kernel
 *                           decides about their re-emission and the survivors
 *                           fly in the next launch of the transport kernel
 *                           (keeps the primary ray bundles together and the
 *                           transport kernel small); 0 = follow re-emissions
 *                           in place
 *   "refill_threshold_reemit" (32), "reemit_inline_below" (-1 = auto: 4096, on a block of a decomposed grid 262144),
 *   "reemit_max_passes" (12)  refill threshold of the later passes; a pass
 *                           with fewer packets than this, or the last allowed
 *                           pass, follows re-emissions in place
 *   "tile_rounds" (1)       with re-emission in passes: the later generations
 *                           fly in tile rounds - flights wait in slots keyed by
 *                           the tile (16^3 cells; 8^3 for multi-ion transport)
 *                           they are about to enter; every round sorts the
 *                           slots and marches each flight through ONE tile
 *                           with the tile's transport records and accumulators
 *                           in LDS (written back with full-line atomics)
 *                           instead of one memory-side atomic per DDA step
 *   "tile_min_flights" (100000), "tile_min_per_item" (-1 = auto: 200), "tile_max_rounds" (1000)
 *                           the rounds end - and passes of the transport
 *                           kernel take over - once fewer flights than this,
 *                           or fewer than this per unit of work (<= 8192
 *                           flights of one tile), are left
 *   "tile_refill_threshold" (48)  idle lanes of a wave that trigger a refill
 *                           in the tile kernel
 *   "park_in_place" (1)     runs with re-emission in passes / tile rounds: the
 *                           first generation leaves an absorbed packet's
 *                           record at the place of its position in the
 *                           launch's order instead of claiming a place from
 *                           the queue's counter (one returning atomic per
 *                           bundle on one word)
 *   "block_select" (1)      a block of a decomposed grid picks the packets
 *                           that start in it out of a launch's ids before
 *                           keys, sort and transport (the reference gives a
 *                           subgrid's source task its own share of the
 *                           packets, src/DistributedPhotonSource.hpp:140-200);
 *                           0: every block runs all ids through them and the
 *                           transport kernel drops the others
 *   "block_first_kernels" (1)  ... and flies them with the kernels built for
 *                           a whole grid's first generation (padded march,
 *                           emission rows from the key kernel); 0: the pass
 *                           kernels, as before round 6
 *   "tile_compact_ratio" (-1)  the rows of the live flights are copied into
 *                           fresh rows, in tile order, once the flights are
 *                           spread over this many slots per flight; 0: never;
 *                           -1: 2 for multi-ion transport, never for
 *                           hydrogen-only
 *   "temperature_pipeline" (1)  the temperature solve as one kernel per stage
 *                           of a secant step (ionization balance / line
 *                           cooling / update), each dense in like work -
 *                           0: one kernel holding the whole solve of a cell
 *   "temperature_finish_slots" (1024)  ... and once so few cells are still
 *                           iterating, one launch takes them to their end
 *                           (a wave per cell)
 *   "pad_march" (1)         hydrogen-only runs on a whole, non-periodic grid:
 *                           the first generation marches through a copy of
 *                           n x_H with one layer of ghost cells, whose record
 *                           says that the packet has left the box (no cell
 *                           counters per axis in the march)
 *   "xcd_remap" (0)         sorted first generation: the blocks that share an
 *                           XCD (block index mod 8) take neighbouring
 *                           positions of the packet order, so that bundles
 *                           crossing the same cells share one L2 (measured:
 *                           no difference on the benchmark grids)
 *   "pre_emission" (1)      multi-ion runs with sorted packets: the spectrum
 *                           sample, the 14 cross sections and the optical
 *                           depth of every new packet are computed by the
 *                           sort-key kernel and read back by the transport
 *                           kernel
 *   "defer_weights" (1)     multi-ion runs: the 14 cross sections of re-emitted
 *                           flights are computed by a kernel of their own, not
 *                           inside the re-emission kernels
 *   "tile_counting_sort" (1)  the slots are put in tile order by counting
 *                           (per-tile counters in LDS; up to 32768 tiles) -
 *                           0: by rocPRIM's radix sort
 *   "timing" (0)            record HIP events around every launch for
 *                           cmi_gpu_get_timing / _kernel_timing /
 *                           _launch_times (off: a run creates no events)
 *   "exact_dda" (0)         march with the reference's per-step arithmetic
 *                           (bit-identical path lengths) instead of the
 *                           incremental marcher (equal up to rounding)
 *   "exp_no_atomics" (0)    EXPERIMENT ONLY, builds with -DCMI_EXPERIMENTS
 *                           (results are wrong): 1 = skip the
 *                           accumulation; multi-ion kernels: 2 = post
 *                           destinations but skip the walk, 3 = walk without
 *                           the adds, 4 = as 2 without the table look-up,
 *                           5 = as 2 without the periodic write-backs,
 *                           6 = a quarter of the table adds
 * A host that cannot call this (the cmi-gpu executable, a code that links
 * the library mode) sets the environment variable
 * CMI_GPU_TUNING="key=value,key=value": cmi_gpu_create applies it to every
 * engine it makes (an unknown key makes it fail). For experiments and
 * bisections, not for configuration. */
int cmi_gpu_set_tuning(cmi_gpu_engine *engine, const char *key, int64_t value);

/* ------------------------------------------------- several GPUs, one host -- */

/* A group of engines driven by one host process, one per GPU of a node:
 * replicas of one grid, or the blocks of one decomposed grid. Replaces the
 * reference's MPICommunicator for this path (src/MPICommunicator.hpp); the
 * transport is RCCL over xGMI (loaded at the first reduce) and peer-to-peer
 * device writes. The engines stay owned by the caller.
 *
 * Engines of a group that hold the same cells form a CLASS: the replicas of
 * replica mode are one class; in domain mode several engines may hold the
 * same block - the reference's copies of a busy subgrid
 * (DensitySubGridCreator::create_copies, src/DensitySubGridCreator.hpp:437-531,
 * used for the subgrids that contain a source,
 * src/TaskBasedIonizationSimulation.cpp:514-560). cmi_gpu_group_create finds
 * the classes itself; copy r of c engines of a block emits the packets of
 * that block whose id is congruent to r modulo c and receives the flights of
 * those packets. The members of a class must sit on distinct devices (or all
 * on one: tests). */
typedef struct cmi_gpu_group cmi_gpu_group;
int cmi_gpu_group_create(int32_t n, cmi_gpu_engine *const *engines,
                         cmi_gpu_group **out);
int cmi_gpu_group_destroy(cmi_gpu_group *group);

/* replaces: the MPI_Allreduce(SUM) of every accumulator field after the
 * packets of all ranks have flown (src/IonizationSimulation.cpp:459-528,
 * MPICommunicator::reduce, src/MPICommunicator.hpp:504-560): afterwards every
 * engine of the (replica) group holds the sum over the group of the
 * accumulator fields a transport step can have written - ONE grouped
 * ncclAllReduce per contiguous piece instead of 16 chunked ones. The packet
 * counters are summed by the caller (cmi_gpu_get_counters of each engine).
 * Asynchronous on the engines' streams.
 * Also replaces DensitySubGridCreator::update_original_counters
 * (src/DensitySubGridCreator.hpp:556-574): the sum runs within every class
 * of the group, so in domain mode the copies of a block end with the block's
 * summed integrals - each then runs cmi_gpu_update_cells on identical input,
 * which stands in for update_copy_properties (:580-598). */
int cmi_gpu_group_reduce_accumulators(cmi_gpu_group *group);

/* replaces: the cell update of the reference's MPI path
 * (src/IonizationSimulation.cpp:532-618): every rank solves a block of the
 * cells (MPICommunicator::distribute, src/MPICommunicator.hpp:207-222), then
 * the temperature and the ionic fractions are gathered
 * (MPICommunicator::gather). Here per class of the group: member r solves
 * slab r of the class's cells (cmi_gpu_update_cells_range) and every member
 * ends with all slabs and their transport records - grouped in-place
 * ncclAllGathers over RCCL / xGMI between distinct devices; peer reads by a
 * kernel where the slabs are unequal or the engines share a device. A class
 * of one engine (a block without copies) simply updates its cells. Call after
 * cmi_gpu_group_reduce_accumulators. */
int cmi_gpu_group_update_cells(cmi_gpu_group *group, uint32_t loop,
                               double totweight);

/* replaces: the photon-buffer traffic between subgrids
 * (src/PhotonTraversalTaskContext.hpp:100-278, src/MemorySpace.hpp:96-127;
 * message format src/PhotonBuffer.hpp:46-48): one exchange round of a
 * decomposed grid whose blocks are the group's engines. Every flight in an
 * engine's export buffer is written into the inbox of the engine that owns
 * the cell it enters - by a kernel on the source device, across xGMI where
 * the owner is another GPU - the export buffers are emptied, and every engine
 * continues the flights it received (cmi_gpu_shoot_flights), which may export
 * again. *total_flights = flights handed over in this round; call until 0.
 * Only n x n counts cross the host. */
int cmi_gpu_group_exchange_flights(cmi_gpu_group *group, uint32_t seed,
                                   uint32_t iteration, uint64_t first_packet,
                                   uint64_t *total_flights);
/* What the exchange rounds cost on the HOST (the reference's counterpart is
 * the queue handling of src/TaskBasedIonizationSimulation.cpp:643-1073): over
 * the *rounds rounds that moved flights since the last reset, microseconds[0]
 * = until the n x n counts were known on the host, [1] = starting and joining
 * the owners' host threads beyond the longest cmi_gpu_shoot_flights call, [2]
 * = the whole of cmi_gpu_group_exchange_flights (with the flights). */
int cmi_gpu_group_exchange_stats(cmi_gpu_group *group, uint64_t *rounds,
                                 double *microseconds, int32_t reset);

/* --------------------------------------------------- test / measurement -- */

/* Parity probe of PhotonSource::get_random_photon + the first optical depth
 * (src/PhotonSource.cpp:208-249, src/IonizationPhotonShootJob.hpp:119-135)
 * for packets [first_packet, first_packet + n): host outputs position [n][3],
 * direction [n][3], frequency [n], cross_sections [n][14], tau [n].
 * Synchronous. */
int cmi_gpu_emit_packets(cmi_gpu_engine *engine, uint32_t seed,
                         uint32_t iteration, uint64_t first_packet, uint64_t n,
                         double *position, double *direction,
                         double *frequency, double *cross_sections,
                         double *tau);

/* Parity probe of CartesianDensityGrid::interact
 * (src/CartesianDensityGrid.cpp:375-452) for n caller-specified packets:
 * position/direction host [n][3], tau host [n], sigma_H / sigma_He_corr host
 * [n] (the two cross sections that enter the optical depth). For packet i up
 * to max_steps (cell, ds) pairs are written to out_cell/out_ds
 * [n][max_steps]; out_nsteps [n]; out_last_cell [n] (-1 = left the box);
 * out_position [n][3] final position. Does NOT touch the accumulators.
 * Uses the marcher selected by the "exact_dda" tuning knob. Synchronous. */
int cmi_gpu_trace_packets(cmi_gpu_engine *engine, uint64_t n,
                          const double *position, const double *direction,
                          const double *tau, const double *sigma_H,
                          const double *sigma_He_corr, int32_t max_steps,
                          int64_t *out_cell, double *out_ds,
                          int32_t *out_nsteps, int64_t *out_last_cell,
                          double *out_position);

/* Parity probe of the spectrum samplers (get_random_frequency of
 * PlanckPhotonSourceSpectrum kind 0, HydrogenLymanContinuumSpectrum 1,
 * HeliumLymanContinuumSpectrum 2, HeliumTwoPhotonContinuumSpectrum 3): n
 * frequencies (Hz) at `temperature`, sample i from packet stream i of `seed`.
 * Synchronous. */
int cmi_gpu_sample_spectrum(cmi_gpu_engine *engine, int32_t kind,
                            double temperature, uint32_t seed, uint64_t n,
                            double *frequencies);

/* Parity probe of the thermal balance for n independent cells given as rows:
 * J [n][14] and heating [n][2] already normalised (jfac = hfac = 1),
 * temperature [n], number_density [n].
 * solve = 0: one TemperatureCalculator::compute_cooling_and_heating_balance
 *   (src/TemperatureCalculator.cpp:207-501) at the given temperature;
 *   out_pair [n][2] = {gain, loss}, out_fractions [n][14] = {h0, he0, metals};
 * solve = 1: TemperatureCalculator::calculate_temperature (:567-931);
 *   out_temperature [n], out_fractions [n][14], out_pair = heating terms.
 * Synchronous. */
int cmi_gpu_thermal_probe(cmi_gpu_engine *engine, int64_t n, int32_t solve,
                          const double *J, const double *heating,
                          const double *temperature,
                          const double *number_density, double *out_fractions,
                          double *out_temperature, double *out_pair);

/* Parity probe of the atomic-data functions for n input rows (host arrays),
 * so that the reference's own fixtures can be checked on the device:
 *  kind 0: in [n] frequency (Hz) -> out [n][14] CrossSections::
 *          get_cross_section (src/VernerCrossSections.cpp:259-322 or
 *          src/FixedValueCrossSections.hpp:151-154), m^2;
 *  kind 1: in [n] T (K) -> out [n][14] RecombinationRates::
 *          get_recombination_rate (src/VernerRecombinationRates.cpp:140-333),
 *          m^3 s^-1;
 *  kind 2: in [n][15] {T, n_e (m^-3), 13 abundances in the order of
 *          src/LineCoolingData.hpp:38-80} -> out [n] LineCoolingData::
 *          get_cooling (src/LineCoolingData.cpp:1767-1847);
 *  kind 3: in [n] T -> out [n][5] PhysicalDiffuseReemissionHandler::
 *          set_reemission_probabilities
 *          (src/PhysicalDiffuseReemissionHandler.hpp:66-105);
 *  kind 4: in [n] T / 1e4 K -> out [n][14][3] charge transfer: recombination
 *          with H, ionization by H+, recombination with He
 *          (src/ChargeTransferRates.cpp:44-157, :169-250, :262-395).
 * Synchronous. */
int cmi_gpu_physics_probe(cmi_gpu_engine *engine, int32_t kind, int64_t n,
                          const double *in, double *out);

/* Device time (HIP events on the engine's stream) spent in cmi_gpu_shoot
 * (packet ordering + every transport launch of a batch; shoot_launches counts
 * batches) and in the cell-update kernels since the last call with
 * reset != 0. Synchronous. */
int cmi_gpu_get_timing(cmi_gpu_engine *engine, int32_t reset,
                       double *shoot_ms, uint64_t *shoot_launches,
                       double *update_ms, uint64_t *update_launches);

/* Device time of the transport kernel alone (HIP events around each
 * shoot_kernel launch, one per re-emission generation) since the last
 * cmi_gpu_get_timing(reset != 0). Synchronous. */
int cmi_gpu_get_kernel_timing(cmi_gpu_engine *engine, double *kernel_ms,
                              uint64_t *kernel_launches);
/* the same per launch, in launch order: duration and the number of flights
 * the launch started (packets of the batch, then of each re-emission
 * generation). *count receives the number of launches; at most `capacity`
 * entries are written. Synchronous. */
int cmi_gpu_get_launch_times(cmi_gpu_engine *engine, uint64_t capacity,
                             double *ms, uint64_t *packets, uint64_t *count);
/* ... and the value of the DDA step counter (cmi_gpu_get_counters' nsteps:
 * steps since the last cmi_gpu_reset_grid) after each of those launches: the
 * steps a launch executed are the difference to the entry before. Needs
 * "timing". Synchronous. */
int cmi_gpu_get_launch_steps(cmi_gpu_engine *engine, uint64_t capacity,
                             uint64_t *steps, uint64_t *count);

#ifdef __cplusplus
}
#endif

#endif /* CMI_GPU_H */
