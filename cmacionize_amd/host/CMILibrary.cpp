/*
 * CMILibrary.cpp - the reference's library mode on top of the GPU engine:
 * the C entry points of src/CMILibrary.hpp / src/CMILibrary.cpp:48-222, with
 * the same names, arguments and meaning, built into libcmi_gpu_library.so. A
 * code that links the reference's libCMILibrary (SPH codes, the Fortran
 * bindings of the reference's fortran/ directory) links this instead.
 *
 *   cmi_init(parameter_file, num_thread, unit_length_in_SI, unit_mass_in_SI,
 *            mapping_type, talk)      (+ the periodic dp / sp variants)
 *   cmi_compute_neutral_fraction_dp (double x y z h m -> double nH)
 *   cmi_compute_neutral_fraction_mp (double x y z, float h m -> FLOAT nH)
 *   cmi_compute_neutral_fraction_sp (float x y z h m -> float nH)
 *   cmi_destroy()
 * typed exactly as src/CMILibrary.hpp:46-72 declares them
 * (tests/support/cmi_library_caller.c is a C program compiled against those
 * prototypes).
 *
 * cmi_compute_neutral_fraction_*: SPHArrayInterface::reset with the caller's
 * particle arrays, IonizationSimulation::initialize(interface) - the
 * interface is the DensityFunction -, IonizationSimulation::run(interface) -
 * the interface is the extra DensityGridWriter - and fill_array
 * (src/CMILibrary.cpp:149-158).
 *
 * Errors: the reference aborts (cmac_error); here a message goes to stderr
 * and the neutral fractions are left untouched / cmi_init leaves the library
 * uninitialised - a host code checks cmi_gpu_library_status() (0 = fine).
 * CMI_GPU_DEVICE in the environment selects the HIP device (default 0).
 */
#include "SPHArrayInterface.hpp"

#include "../../include/cmi_library.h"

#include <cstdlib>
#include <iostream>
#include <memory>

using namespace cmi;

namespace {
std::unique_ptr<GpuIonizationSimulation> global_ionization_simulation;
std::unique_ptr<SPHArrayInterface> global_interface;
int global_status = 0;

int device_from_environment() {
  const char *d = std::getenv("CMI_GPU_DEVICE");
  return d ? std::atoi(d) : 0;
}

template <typename Make>
void init(const char *parameter_file, int num_thread, bool talk, Make make) {
  global_status = 1;
  try {
    /* IonizationSimulation(write_output = true, every_iteration_output =
     * false, output_statistics = false, ...), src/CMILibrary.cpp:58-60 */
    global_ionization_simulation.reset(new GpuIonizationSimulation(
        true, false, false, num_thread, parameter_file,
        device_from_environment(), talk));
    global_interface.reset(make());
    global_status = 0;
  } catch (const std::exception &e) {
    std::cerr << "cmi_init: " << e.what() << std::endl;
    global_ionization_simulation.reset();
    global_interface.reset();
  }
}

template <typename TX, typename TH, typename TN>
void compute(const TX *x, const TX *y, const TX *z, const TH *h, const TH *m,
             TN *nH, size_t N) {
  if (!global_ionization_simulation || !global_interface) {
    std::cerr << "cmi_compute_neutral_fraction: cmi_init has not succeeded"
              << std::endl;
    global_status = 1;
    return;
  }
  try {
    global_interface->reset(x, y, z, h, m, N);
    global_ionization_simulation->initialize(global_interface.get());
    global_ionization_simulation->run(global_interface.get());
    global_interface->fill_array(nH);
    global_status = 0;
  } catch (const std::exception &e) {
    std::cerr << "cmi_compute_neutral_fraction: " << e.what() << std::endl;
    global_status = 1;
  }
}
} // namespace

extern "C" {

/* (declared in include/cmi_library.h: a definition that does not match its
 * declaration there does not compile) */
/* src/CMILibrary.cpp:48-62 */
void cmi_init(const char *parameter_file, const int num_thread,
              const double unit_length_in_SI, const double unit_mass_in_SI,
              const char *mapping_type, const int talk) {
  init(parameter_file, num_thread, talk != 0, [&]() {
    return new SPHArrayInterface(unit_length_in_SI, unit_mass_in_SI,
                                 mapping_type);
  });
}

/* :78-92 */
void cmi_init_periodic_dp(const char *parameter_file, const int num_thread,
                          const double unit_length_in_SI,
                          const double unit_mass_in_SI,
                          const double *box_anchor, const double *box_sides,
                          const char *mapping_type, const int talk) {
  init(parameter_file, num_thread, talk != 0, [&]() {
    return new SPHArrayInterface(unit_length_in_SI, unit_mass_in_SI,
                                 box_anchor, box_sides, mapping_type);
  });
}

/* :110-124 */
void cmi_init_periodic_sp(const char *parameter_file, const int num_thread,
                          const double unit_length_in_SI,
                          const double unit_mass_in_SI, const float *box_anchor,
                          const float *box_sides, const char *mapping_type,
                          const int talk) {
  init(parameter_file, num_thread, talk != 0, [&]() {
    return new SPHArrayInterface(unit_length_in_SI, unit_mass_in_SI,
                                 box_anchor, box_sides, mapping_type);
  });
}

/* :129-133 */
void cmi_destroy() {
  global_ionization_simulation.reset();
  global_interface.reset();
}

/* :149-158 */
void cmi_compute_neutral_fraction_dp(const double *x, const double *y,
                                     const double *z, const double *h,
                                     const double *m, double *nH,
                                     const size_t N) {
  compute(x, y, z, h, m, nH, N);
}

/* :174-183 */
void cmi_compute_neutral_fraction_mp(const double *x, const double *y,
                                     const double *z, const float *h,
                                     const float *m, float *nH,
                                     const size_t N) {
  compute(x, y, z, h, m, nH, N);
}

/* :199-208 */
void cmi_compute_neutral_fraction_sp(const float *x, const float *y,
                                     const float *z, const float *h,
                                     const float *m, float *nH,
                                     const size_t N) {
  compute(x, y, z, h, m, nH, N);
}

/* not in the reference (it aborts on errors): 0 = the last call succeeded */
int cmi_gpu_library_status() { return global_status; }

/* Not in the reference either: the two mappings of the coupling object on
 * their own, host side only (no engine, no GPU), for the tests that pin them
 * - the known answers of test/testSPHArrayInterface.cpp:70-155 are total
 * hydrogen numbers of exactly this call sequence (reset, initialize,
 * DensityGrid::initialize). precision: 0 = all arrays double, 1 = double
 * positions with float h and m, 2 = all float (the three reset overloads).
 * periodic_box: NULL or {anchor[3], sides[3]} in the caller's length unit.
 * Returns 0, or 1 with a message on stderr. */
namespace {
SPHArrayInterface *make_interface(const char *mapping_type, double ul,
                                  double um, const double *periodic_box) {
  if (periodic_box)
    return new SPHArrayInterface(ul, um, periodic_box, periodic_box + 3,
                                 mapping_type);
  return new SPHArrayInterface(ul, um, mapping_type);
}
void reset_interface(SPHArrayInterface &interface, int precision,
                     const void *x, const void *y, const void *z,
                     const void *h, const void *m, size_t N) {
  if (precision == 0)
    interface.reset((const double *)x, (const double *)y, (const double *)z,
                    (const double *)h, (const double *)m, N);
  else if (precision == 1)
    interface.reset((const double *)x, (const double *)y, (const double *)z,
                    (const float *)h, (const float *)m, N);
  else
    interface.reset((const float *)x, (const float *)y, (const float *)z,
                    (const float *)h, (const float *)m, N);
  interface.initialize();
}
DensityGrid *make_grid(const double *grid_anchor, const double *grid_sides,
                       const int *ncell) {
  const SimulationBox box({grid_anchor[0], grid_anchor[1], grid_anchor[2]},
                          {grid_sides[0], grid_sides[1], grid_sides[2]},
                          {false, false, false});
  return new DensityGrid(box, {ncell[0], ncell[1], ncell[2]});
}
} // namespace

int cmi_gpu_library_map_to_cells(const char *mapping_type, int precision,
                                 const void *x, const void *y, const void *z,
                                 const void *h, const void *m, size_t N,
                                 double unit_length_in_SI,
                                 double unit_mass_in_SI,
                                 const double *periodic_box,
                                 const double *grid_anchor,
                                 const double *grid_sides, const int *ncell,
                                 double *number_density) {
  try {
    std::unique_ptr<SPHArrayInterface> interface(make_interface(
        mapping_type, unit_length_in_SI, unit_mass_in_SI, periodic_box));
    reset_interface(*interface, precision, x, y, z, h, m, N);
    std::unique_ptr<DensityGrid> grid(
        make_grid(grid_anchor, grid_sides, ncell));
    const int64_t n = grid->get_number_of_cells();
    std::string error;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
      try {
        DensityGrid::iterator cell(grid.get(), i);
        number_density[i] = (*interface)(cell).get_number_density();
      } catch (const std::exception &e) {
#pragma omp critical(cmi_library_probe_error)
        error = e.what();
      }
    }
    if (!error.empty())
      throw std::runtime_error(error);
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "cmi_gpu_library_map_to_cells: " << e.what() << std::endl;
    return 1;
  }
}

int cmi_gpu_library_map_to_particles(
    const char *mapping_type, int precision, const void *x, const void *y,
    const void *z, const void *h, const void *m, size_t N,
    double unit_length_in_SI, double unit_mass_in_SI,
    const double *periodic_box, const double *grid_anchor,
    const double *grid_sides, const int *ncell,
    const double *neutral_fraction_of_cells, double *nH) {
  try {
    std::unique_ptr<SPHArrayInterface> interface(make_interface(
        mapping_type, unit_length_in_SI, unit_mass_in_SI, periodic_box));
    reset_interface(*interface, precision, x, y, z, h, m, N);
    std::unique_ptr<DensityGrid> grid(
        make_grid(grid_anchor, grid_sides, ncell));
    const int64_t n = grid->get_number_of_cells();
    grid->_ionic_fraction[ION_H_n].assign(neutral_fraction_of_cells,
                                          neutral_fraction_of_cells + n);
    ParameterFile no_parameters;
    interface->write(*grid, 0, no_parameters);
    interface->fill_array(nH);
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "cmi_gpu_library_map_to_particles: " << e.what() << std::endl;
    return 1;
  }
}

} // extern "C"
