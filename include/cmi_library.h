/*
 * cmi_library.h - the C interface of cmacionize_amd/libcmi_gpu_library.so:
 * the reference's LIBRARY MODE (an SPH code hands over its particles and gets
 * their neutral fractions back) on the MI355X engine.
 *
 * The seven entry points below are the reference's own, name by name and
 * type by type (src/CMILibrary.hpp:46-72; defined in src/CMILibrary.cpp:
 * 48-222): a program written against the reference's header links this
 * library unchanged. tests/support/cmi_library_caller.c is such a program; it
 * declares the prototypes itself, from the reference's header.
 *
 * mapping_type: "M_over_V", "centroid", "Petkova" (SPHArrayInterface::
 * get_mapping_type, src/SPHArrayInterface.hpp:105-118), and here also
 * "Petkova_oriented" (host/PetkovaMapping.hpp).
 *
 * Differences from the reference: errors are reported on stderr and through
 * cmi_gpu_library_status() instead of aborting the calling program; the
 * environment variable CMI_GPU_DEVICE selects the HIP device (default 0).
 */
#ifndef CMI_LIBRARY_H
#define CMI_LIBRARY_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/CMILibrary.hpp:46-48, src/CMILibrary.cpp:48-62: non-periodic box (the
 * particles' extent + 1 %). unit_*_in_SI: the caller's length and mass units
 * in m and kg. talk != 0: progress messages on stdout. */
void cmi_init(const char *parameter_file, const int num_thread,
              const double unit_length_in_SI, const double unit_mass_in_SI,
              const char *mapping_type, const int talk);
/* src/CMILibrary.hpp:49-53, src/CMILibrary.cpp:78-92: periodic box, anchor
 * and sides in the caller's length unit */
void cmi_init_periodic_dp(const char *parameter_file, const int num_thread,
                          const double unit_length_in_SI,
                          const double unit_mass_in_SI,
                          const double *box_anchor, const double *box_sides,
                          const char *mapping_type, const int talk);
/* src/CMILibrary.hpp:54-58, src/CMILibrary.cpp:110-124 */
void cmi_init_periodic_sp(const char *parameter_file, const int num_thread,
                          const double unit_length_in_SI,
                          const double unit_mass_in_SI, const float *box_anchor,
                          const float *box_sides, const char *mapping_type,
                          const int talk);
/* src/CMILibrary.hpp:59, src/CMILibrary.cpp:129-133 */
void cmi_destroy();

/* src/CMILibrary.hpp:61-64, src/CMILibrary.cpp:149-158: particles -> cell
 * densities, the whole ionization simulation, cells -> particles. x y z h:
 * positions and smoothing lengths (kernel support radii) in the caller's
 * length unit, m: masses in its mass unit, nH: N neutral fractions out. */
void cmi_compute_neutral_fraction_dp(const double *x, const double *y,
                                     const double *z, const double *h,
                                     const double *m, double *nH,
                                     const size_t N);
/* src/CMILibrary.hpp:65-67, src/CMILibrary.cpp:174-183: double positions,
 * FLOAT h, m and nH */
void cmi_compute_neutral_fraction_mp(const double *x, const double *y,
                                     const double *z, const float *h,
                                     const float *m, float *nH, const size_t N);
/* src/CMILibrary.hpp:68-70, src/CMILibrary.cpp:199-208 */
void cmi_compute_neutral_fraction_sp(const float *x, const float *y,
                                     const float *z, const float *h,
                                     const float *m, float *nH, const size_t N);

/* ---- not in the reference ---------------------------------------------- */

/* 0 = the last cmi_init* / cmi_compute_* call succeeded (the reference aborts
 * the program instead, cmac_error) */
int cmi_gpu_library_status();

/* The coupling object's two mappings on their own, host side only (no engine,
 * no GPU) - what the tests pin against the known answers of
 * test/testSPHArrayInterface.cpp:70-155: SPHArrayInterface::reset +
 * initialize + one DensityFunction call per cell of a Cartesian grid
 * (src/SPHArrayInterface.cpp:146-203,931-1010), resp. + write + fill_array
 * (src/SPHArrayInterface.cpp:1018-1075, src/SPHArrayInterface.hpp:123-251).
 * precision: 0 = all arrays double, 1 = double x y z with float h m, 2 = all
 * float (the three reset overloads). periodic_box: NULL or {anchor[3],
 * sides[3]} in the caller's length unit; grid_anchor, grid_sides in m;
 * ncell[3]; number_density [ncell] out in m^-3;
 * neutral_fraction_of_cells [ncell] in, nH [N] (double) out.
 * Return 0, or 1 with a message on stderr. */
int cmi_gpu_library_map_to_cells(const char *mapping_type, int precision,
                                 const void *x, const void *y, const void *z,
                                 const void *h, const void *m, size_t N,
                                 double unit_length_in_SI,
                                 double unit_mass_in_SI,
                                 const double *periodic_box,
                                 const double *grid_anchor,
                                 const double *grid_sides, const int *ncell,
                                 double *number_density);
int cmi_gpu_library_map_to_particles(
    const char *mapping_type, int precision, const void *x, const void *y,
    const void *z, const void *h, const void *m, size_t N,
    double unit_length_in_SI, double unit_mass_in_SI,
    const double *periodic_box, const double *grid_anchor,
    const double *grid_sides, const int *ncell,
    const double *neutral_fraction_of_cells, double *nH);

#ifdef __cplusplus
}
#endif

#endif
