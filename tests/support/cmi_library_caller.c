/*
 * cmi_library_caller.c - a plain C program that uses libcmi_gpu_library.so the
 * way an SPH code uses the reference's libCMILibrary: the prototypes below are
 * typed from the reference's header (src/CMILibrary.hpp:46-72), NOT from this
 * repository's sources, so a mismatch in an argument list or an element type
 * shows up here as a compile error, a crash or a damaged guard word.
 *
 *   cmi_library_caller PARAMETER_FILE MAPPING_TYPE OUTPUT_FILE
 *
 * Runs the three call sequences a caller can choose from
 *   cmi_init_periodic_dp -> cmi_compute_neutral_fraction_dp   (all double)
 *   cmi_init             -> cmi_compute_neutral_fraction_mp   (double x y z,
 *                                                  float h m, FLOAT nH)
 *   cmi_init_periodic_sp -> cmi_compute_neutral_fraction_sp   (all float)
 * on the same cloud of particles (a jittered lattice filling a 10 pc box of
 * 100 hydrogen atoms per cm^3, caller's units parsec and solar mass), every nH
 * buffer exactly N elements long between two guard words. Writes
 * "x y z nH_dp nH_mp nH_sp" per particle to OUTPUT_FILE. Exit code: 0 = all
 * guards intact and every nH was written; 2 = a guard word was overwritten;
 * 3 = an nH element was left untouched; 4 = usage / allocation.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* src/CMILibrary.hpp:46-72 */
void cmi_init(const char *parameter_file, const int num_thread,
              const double unit_length_in_SI, const double unit_mass_in_SI,
              const char *mapping_type, const int talk);
void cmi_init_periodic_dp(const char *parameter_file, const int num_thread,
                          const double unit_length_in_SI,
                          const double unit_mass_in_SI,
                          const double *box_anchor, const double *box_sides,
                          const char *mapping_type, const int talk);
void cmi_init_periodic_sp(const char *parameter_file, const int num_thread,
                          const double unit_length_in_SI,
                          const double unit_mass_in_SI, const float *box_anchor,
                          const float *box_sides, const char *mapping_type,
                          const int talk);
void cmi_destroy();
void cmi_compute_neutral_fraction_dp(const double *x, const double *y,
                                     const double *z, const double *h,
                                     const double *m, double *nH,
                                     const size_t N);
void cmi_compute_neutral_fraction_mp(const double *x, const double *y,
                                     const double *z, const float *h,
                                     const float *m, float *nH, const size_t N);
void cmi_compute_neutral_fraction_sp(const float *x, const float *y,
                                     const float *z, const float *h,
                                     const float *m, float *nH, const size_t N);

#define SIDE 6
#define NPART (SIDE * SIDE * SIDE)
#define NGUARD 8
#define UNTOUCHED (-77.)

static const double parsec = 3.086e16, solar_mass = 1.98855e30;

/* small multiplicative generator for the jitter: the same cloud on every run */
static unsigned long long jitter_state = 88172645463325252ull;
static double jitter(void) {
  jitter_state ^= jitter_state << 13;
  jitter_state ^= jitter_state >> 7;
  jitter_state ^= jitter_state << 17;
  return (double)(jitter_state >> 11) / 9007199254740992. - 0.5;
}

/* an nH buffer between two runs of guard words */
static const unsigned char guard_byte = 0xA5;
static void *guarded(size_t element, size_t n) {
  unsigned char *raw = malloc((n + 2 * NGUARD) * element);
  if (!raw)
    exit(4);
  memset(raw, guard_byte, (n + 2 * NGUARD) * element);
  return raw + NGUARD * element;
}
static int guards_intact(const void *buffer, size_t element, size_t n) {
  const unsigned char *raw = (const unsigned char *)buffer - NGUARD * element;
  size_t i;
  for (i = 0; i < NGUARD * element; ++i)
    if (raw[i] != guard_byte || raw[(NGUARD + n) * element + i] != guard_byte)
      return 0;
  return 1;
}

int main(int argc, char **argv) {
  double x[NPART], y[NPART], z[NPART], h[NPART], m[NPART];
  float xf[NPART], yf[NPART], zf[NPART], hf[NPART], mf[NPART];
  double anchor[3] = {-5., -5., -5.}, sides[3] = {10., 10., 10.};
  float anchor_f[3] = {-5.f, -5.f, -5.f}, sides_f[3] = {10.f, 10.f, 10.f};
  double *nH_dp;
  float *nH_mp, *nH_sp;
  const double spacing = 10. / SIDE;
  /* 100 cm^-3 of hydrogen in (10 pc)^3, in solar masses */
  const double total_mass =
      100.e6 * 1.6737236e-27 * 1.e3 * parsec * parsec * parsec / solar_mass;
  int i, j, k, n = 0, status = 0;
  FILE *out;

  if (argc != 4) {
    fprintf(stderr, "usage: %s PARAMETER_FILE MAPPING_TYPE OUTPUT_FILE\n",
            argv[0]);
    return 4;
  }
  for (i = 0; i < SIDE; ++i)
    for (j = 0; j < SIDE; ++j)
      for (k = 0; k < SIDE; ++k, ++n) {
        x[n] = anchor[0] + (i + 0.5 + 0.3 * jitter()) * spacing;
        y[n] = anchor[1] + (j + 0.5 + 0.3 * jitter()) * spacing;
        z[n] = anchor[2] + (k + 0.5 + 0.3 * jitter()) * spacing;
        h[n] = 2. * spacing;
        m[n] = total_mass / NPART;
        xf[n] = (float)x[n];
        yf[n] = (float)y[n];
        zf[n] = (float)z[n];
        hf[n] = (float)h[n];
        mf[n] = (float)m[n];
      }

  nH_dp = guarded(sizeof(double), NPART);
  nH_mp = guarded(sizeof(float), NPART);
  nH_sp = guarded(sizeof(float), NPART);
  for (n = 0; n < NPART; ++n) {
    nH_dp[n] = UNTOUCHED;
    nH_mp[n] = (float)UNTOUCHED;
    nH_sp[n] = (float)UNTOUCHED;
  }

  cmi_init_periodic_dp(argv[1], 1, parsec, solar_mass, anchor, sides, argv[2],
                       0);
  cmi_compute_neutral_fraction_dp(x, y, z, h, m, nH_dp, NPART);
  cmi_destroy();

  cmi_init(argv[1], 1, parsec, solar_mass, argv[2], 0);
  cmi_compute_neutral_fraction_mp(x, y, z, hf, mf, nH_mp, NPART);
  cmi_destroy();

  cmi_init_periodic_sp(argv[1], 1, parsec, solar_mass, anchor_f, sides_f,
                       argv[2], 0);
  cmi_compute_neutral_fraction_sp(xf, yf, zf, hf, mf, nH_sp, NPART);
  cmi_destroy();

  if (!guards_intact(nH_dp, sizeof(double), NPART)) {
    fprintf(stderr, "guard words around the double nH buffer damaged\n");
    status = 2;
  }
  if (!guards_intact(nH_mp, sizeof(float), NPART)) {
    fprintf(stderr, "guard words around the mixed-precision float nH buffer "
                    "damaged\n");
    status = 2;
  }
  if (!guards_intact(nH_sp, sizeof(float), NPART)) {
    fprintf(stderr, "guard words around the float nH buffer damaged\n");
    status = 2;
  }
  out = fopen(argv[3], "w");
  if (!out)
    return 4;
  for (n = 0; n < NPART; ++n) {
    if (!status && (nH_dp[n] == UNTOUCHED || nH_mp[n] == (float)UNTOUCHED ||
                    nH_sp[n] == (float)UNTOUCHED)) {
      fprintf(stderr, "nH of particle %d was not written\n", n);
      status = 3;
    }
    fprintf(out, "%.17g %.17g %.17g %.17g %.9g %.9g\n", x[n], y[n], z[n],
            nH_dp[n], (double)nH_mp[n], (double)nH_sp[n]);
  }
  fclose(out);
  return status;
}
