/*
 * cmio_linecooling.c - ORACLE (test infrastructure): collisionally excited
 * line cooling of 10 five-level and 3 two-level ions.
 *
 * Restates src/LineCoolingData.cpp:
 *   :42-1399   data set-up (numbers in cmio_linecooling_data.h)
 *   :1492-1555 solve_system_of_linear_equations
 *   :1569-1701 compute_level_populations
 *   :1714-1738 compute_level_population (two levels)
 *   :1767-1847 get_cooling
 */
#include "cmio_internal.h"
#include "cmio_linecooling_data.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* transition (lower i, upper j) -> index in the 10-entry tables */
static const int TR[5][5] = {{-1, 0, 1, 2, 3},
                             {-1, -1, 4, 5, 6},
                             {-1, -1, -1, 7, 8},
                             {-1, -1, -1, -1, 9},
                             {-1, -1, -1, -1, -1}};

typedef struct {
  double energy[CMI_LC_NFIVE][CMI_LC_NTRANS]; /* K */
  double two_energy[CMI_LC_NTWO];             /* K */
  double prefactor; /* h^2 / (sqrt(k) (2 pi m_e)^1.5) */
  int ready;
} lc_tables;

static lc_tables g_lc;

static double unit_factor(int unit) {
  /* src/LineCoolingData.cpp:46-61 */
  if (unit == 0)
    return 100. * CMIO_PLANCK * CMIO_LIGHTSPEED / CMIO_BOLTZMANN;
  if (unit == 1)
    return CMIO_ELECTRONVOLT / CMIO_BOLTZMANN;
  return CMIO_RYDBERG / CMIO_BOLTZMANN;
}

static void lc_init(void) {
  for (int e = 0; e < CMI_LC_NFIVE; ++e) {
    const double f = unit_factor(cmi_lc_five[e].unit);
    const double *lev = cmi_lc_five[e].levels;
    for (int j = 1; j < 5; ++j) {
      /* from the ground state: level * factor; between excited levels:
       * (difference of levels) * factor */
      g_lc.energy[e][TR[0][j]] = lev[j - 1] * f;
      for (int i = 1; i < j; ++i) {
        g_lc.energy[e][TR[i][j]] = (lev[j - 1] - lev[i - 1]) * f;
      }
    }
  }
  for (int e = 0; e < CMI_LC_NTWO; ++e) {
    g_lc.two_energy[e] = cmi_lc_two[e].energy * unit_factor(cmi_lc_two[e].unit);
  }
  g_lc.prefactor = CMIO_PLANCK * CMIO_PLANCK /
                   (sqrt(CMIO_BOLTZMANN) *
                    pow(2. * M_PI * CMIO_ELECTRON_MASS, 1.5));
  g_lc.ready = 1;
}

/* the tables are built when the library is loaded - before any thread of a
 * parallel region can ask for them (a lazy "if (!ready) init" under a
 * critical section is not enough: the compiler may store the flag before the
 * last table entries) */
__attribute__((constructor)) static void lc_ensure(void) {
  if (!g_lc.ready)
    lc_init();
}

double cmio_lc_energy_difference(int element, int transition) {
  lc_ensure();
  return g_lc.energy[element][transition];
}
double cmio_lc_transition_probability(int element, int transition) {
  return cmi_lc_five[element].A[transition];
}
double cmio_lc_statistical_weight(int element, int level) {
  return 1. / cmi_lc_five[element].inv_weight[level];
}

/* Gaussian elimination with partial pivoting on a 5x5 system, solution in B;
 * returns 1 for a singular matrix. */
int cmio_solve_5x5(double A[5][5], double B[5]) {
  for (int j = 0; j < 5; ++j) {
    int imax = 0;
    double Amax = 0.;
    for (int i = j; i < 5; ++i) {
      if (fabs(A[i][j]) > fabs(Amax)) {
        Amax = A[i][j];
        imax = i;
      }
    }
    if (Amax == 0.)
      return 1;
    const double Amax_inv = 1. / Amax;
    for (int k = 0; k < 5; ++k) {
      if (imax != j) {
        const double t = A[j][k];
        A[j][k] = A[imax][k];
        A[imax][k] = t;
      }
      A[j][k] *= Amax_inv;
    }
    if (imax != j) {
      const double t = B[j];
      B[j] = B[imax];
      B[imax] = t;
    }
    B[j] *= Amax_inv;
    for (int i = j + 1; i < 5; ++i) {
      for (int k = j + 1; k < 5; ++k)
        A[i][k] -= A[i][j] * A[j][k];
      B[i] -= A[i][j] * B[j];
    }
  }
  /* back substitution (the diagonal is 1 after the scaling above) */
  for (int i = 3; i >= 0; --i) {
    for (int j = 4; j > i; --j)
      B[i] -= B[j] * A[i][j];
  }
  return 0;
}

/* Omega(T) fit: T^(1+a0) (a1 + a2/T + a3 ln T + a4 T (1 + (a5-1) T^a6)) */
static double collision_strength(const double a[7], double prefactor, double T,
                                 double Tinv, double logT) {
  return prefactor * pow(T, 1. + a[0]) *
         (a[1] + a[2] * Tinv + a[3] * logT +
          a[4] * T * (1. + (a[5] - 1.) * pow(T, a[6])));
}

static int level_populations(int e, double prefactor, double T, double Tinv,
                             double logT, double pop[5]) {
  const cmi_lc_five_level *d = &cmi_lc_five[e];
  double down[CMI_LC_NTRANS], up[CMI_LC_NTRANS];
  for (int t = 0; t < CMI_LC_NTRANS; ++t) {
    const double cs = collision_strength(d->cs[t], prefactor, T, Tinv, logT);
    down[t] = cs;
    up[t] = cs * exp(-g_lc.energy[e][t] * Tinv);
  }
  double M[5][5];
  for (int k = 0; k < 5; ++k) {
    M[0][k] = 1.; /* normalisation: populations sum to 1 */
    pop[k] = 0.;
  }
  pop[0] = 1.;
  for (int i = 1; i < 5; ++i) {
    /* gains from lower levels by collisional excitation */
    for (int j = 0; j < i; ++j)
      M[i][j] = up[TR[j][i]] * d->inv_weight[j];
    /* losses: radiative decay to every lower level, collisions down and up */
    double sumA = d->A[TR[0][i]];
    for (int j = 1; j < i; ++j)
      sumA += d->A[TR[j][i]];
    double sumC = down[TR[0][i]];
    for (int j = 1; j < i; ++j)
      sumC += down[TR[j][i]];
    for (int k = i + 1; k < 5; ++k)
      sumC += up[TR[i][k]];
    M[i][i] = -(sumA + d->inv_weight[i] * sumC);
    /* gains from higher levels: radiative + collisional de-excitation */
    for (int k = i + 1; k < 5; ++k)
      M[i][k] = d->A[TR[i][k]] + d->inv_weight[k] * down[TR[i][k]];
  }
  return cmio_solve_5x5(M, pop);
}

double cmio_line_cooling(double temperature, double electron_density,
                         const double abundances[13]) {
  if (electron_density == 0.)
    return 1.e-99;
  lc_ensure();
  const double kb = CMIO_BOLTZMANN;
  const double prefactor =
      g_lc.prefactor * electron_density / sqrt(temperature);
  const double Tinv = 1. / temperature;
  const double logT = log(temperature);

  double cooling = 0.;
  for (int e = 0; e < CMI_LC_NFIVE; ++e) {
    double pop[5];
    if (level_populations(e, prefactor, temperature, Tinv, logT, pop)) {
      cmio_set_error("cmio: singular level matrix (element %d, T %g)", e,
                     temperature);
      return NAN;
    }
    const cmi_lc_five_level *d = &cmi_lc_five[e];
    double cl[5];
    /* level 1 has a single line; the reference multiplies left to right */
    cl[1] = pop[1] * d->A[TR[0][1]] * g_lc.energy[e][TR[0][1]];
    for (int i = 2; i < 5; ++i) {
      double s = d->A[TR[0][i]] * g_lc.energy[e][TR[0][i]];
      for (int j = 1; j < i; ++j)
        s += d->A[TR[j][i]] * g_lc.energy[e][TR[j][i]];
      cl[i] = pop[i] * s;
    }
    cooling += abundances[e] * kb * (cl[1] + cl[2] + cl[3] + cl[4]);
  }
  for (int i = 0; i < CMI_LC_NTWO; ++i) {
    const cmi_lc_two_level *d = &cmi_lc_two[i];
    const double ksi = g_lc.two_energy[i];
    const double cs = collision_strength(d->cs, prefactor, temperature, Tinv,
                                         logT);
    const double Texp = exp(-ksi * Tinv);
    const double pop = cs * Texp * d->inv_weight[0] /
                       (d->A + cs * (d->inv_weight[1] + Texp * d->inv_weight[0]));
    cooling += abundances[CMI_LC_NFIVE + i] * kb * ksi * d->A * pop;
  }
  return cooling;
}

/* LineCoolingData::get_line_strengths, src/LineCoolingData.cpp:1859-1952: the
 * luminosity of every line per hydrogen atom (J s^-1): out[10 e + t] for
 * transition t (order 0-1, 0-2, 0-3, 0-4, 1-2, 1-3, 1-4, 2-3, 2-4, 3-4) of the
 * five-level ion e, out[100 + i] for the line of two-level ion i */
void cmio_line_strengths(double temperature, double electron_density,
                         const double abundances[13], double *out) {
  lc_ensure();
  const double kb = CMIO_BOLTZMANN;
  const double prefactor =
      g_lc.prefactor * electron_density / sqrt(temperature);
  const double Tinv = 1. / temperature;
  const double logT = log(temperature);
  for (int e = 0; e < CMI_LC_NFIVE; ++e) {
    double pop[5];
    if (level_populations(e, prefactor, temperature, Tinv, logT, pop)) {
      cmio_set_error("cmio: singular level matrix (element %d, T %g)", e,
                     temperature);
      return;
    }
    const cmi_lc_five_level *d = &cmi_lc_five[e];
    const double pre = abundances[e] * kb;
    for (int lo = 0; lo < 4; ++lo)
      for (int hi = lo + 1; hi < 5; ++hi) {
        const int t = TR[lo][hi];
        out[CMI_LC_NTRANS * e + t] = pre * pop[hi] * d->A[t] * g_lc.energy[e][t];
      }
  }
  for (int i = 0; i < CMI_LC_NTWO; ++i) {
    const cmi_lc_two_level *d = &cmi_lc_two[i];
    const double ksi = g_lc.two_energy[i];
    const double cs = collision_strength(d->cs, prefactor, temperature, Tinv,
                                         logT);
    const double Texp = exp(-ksi * Tinv);
    const double pop = cs * Texp * d->inv_weight[0] /
                       (d->A + cs * (d->inv_weight[1] + Texp * d->inv_weight[0]));
    out[CMI_LC_NTRANS * CMI_LC_NFIVE + i] =
        abundances[CMI_LC_NFIVE + i] * kb * pop * ksi * d->A;
  }
}
