/*
 * cmio_physics.c - ORACLE (test infrastructure): atomic data (L2 layer).
 *
 * Restates
 *   src/VernerCrossSections.cpp:36-154 (load), :166-245 (fit), :259-322 (ions)
 *   src/VernerRecombinationRates.cpp:38-90 (load), :104-130 (fit),
 *                                    :140-333 (ions)
 *   src/ChargeTransferRates.cpp:44-395
 *   src/FixedValueCrossSections.hpp:151-154,
 *   src/FixedValueRecombinationRates.hpp:157-160
 */
#include "cmio_atomic_data.h"
#include "cmio_internal.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

/* ---------------------------------------------- Verner cross sections -- */

typedef struct {
  int ion;
  int shell, ninn, ntot;
  double E_th, einn;
  /* fit A */
  double A_Plconst, A_E_0_inv, A_sigma_0, A_y_a_inv, A_P, A_y_w_sq;
  /* fit B */
  double B_E_0_inv, B_sigma_0, B_y_a_inv, B_P, B_y_w_sq, B_y_0, B_y_1_sq;
} verner_term;

static verner_term g_terms[CMI_VERNER_NTERM];
static int g_terms_ready = 0;

static void verner_init(void) {
  /* same conversions, same operation order as the reference constructor */
  const double eV_to_Hz = CMIO_ELECTRONVOLT / CMIO_PLANCK;
  for (int i = 0; i < CMI_VERNER_NTERM; ++i) {
    const cmi_verner_term *raw = &cmi_verner_terms[i];
    verner_term *t = &g_terms[i];
    t->ion = raw->ion;
    t->shell = raw->shell;
    t->ninn = raw->ninn;
    t->ntot = raw->ntot;
    const double E_th = raw->A[0], E_0 = raw->A[1], sigma_0 = raw->A[2],
                 y_a = raw->A[3], P = raw->A[4], y_w = raw->A[5];
    t->E_th = E_th * eV_to_Hz;
    t->einn = (raw->N < 3) ? 1.e30 : raw->einn_eV * eV_to_Hz;
    t->A_Plconst = 0.5 * P - 5.5 - raw->l;
    t->A_E_0_inv = 1. / (E_0 * eV_to_Hz);
    t->A_sigma_0 = 1.e-22 * sigma_0;
    t->A_y_a_inv = 1. / y_a;
    t->A_P = P;
    t->A_y_w_sq = y_w * y_w;
    const double bE_0 = raw->B[2], bsigma_0 = raw->B[3], by_a = raw->B[4],
                 bP = raw->B[5], by_w = raw->B[6], by_0 = raw->B[7],
                 by_1 = raw->B[8];
    t->B_E_0_inv = 1. / (bE_0 * eV_to_Hz);
    t->B_sigma_0 = 1.e-22 * bsigma_0;
    t->B_y_a_inv = 1. / by_a;
    t->B_P = bP;
    t->B_y_w_sq = by_w * by_w;
    t->B_y_0 = by_0;
    t->B_y_1_sq = by_1 * by_1;
  }
  g_terms_ready = 1;
}

__attribute__((constructor)) static void verner_load(void) {
  if (!g_terms_ready)
    verner_init();
}

/* get_cross_section_verner for one (ion, shell) term. The reference's
 * special cases for Z = 15, 17, 19, > 20 and for neutral/singly ionized
 * Z > 18 cannot occur for the tracked ions (Z in {1,2,6,7,8,10,16}). */
static double verner_term_cross_section(const verner_term *t, double e) {
  if (e < t->E_th) {
    return 0.;
  }
  const int is = t->shell;
  const int nout = t->ntot;
  if (is > nout) {
    return 0.;
  }
  const int nint = t->ninn;
  const double einn = t->einn;
  if (is < nout && is > nint && e < einn) {
    return 0.;
  }
  if (is <= nint || e >= einn) {
    const double y = e * t->A_E_0_inv;
    const double ym1 = y - 1.;
    const double Fy = (ym1 * ym1 + t->A_y_w_sq) * pow(y, t->A_Plconst) *
                      pow(1. + sqrt(y * t->A_y_a_inv), -t->A_P);
    return t->A_sigma_0 * Fy;
  } else {
    const double x = e * t->B_E_0_inv - t->B_y_0;
    const double y = sqrt(x * x + t->B_y_1_sq);
    const double xm1 = x - 1.;
    const double Fy = (xm1 * xm1 + t->B_y_w_sq) * pow(y, 0.5 * t->B_P - 5.5) *
                      pow(1. + sqrt(y * t->B_y_a_inv), -t->B_P);
    return t->B_sigma_0 * Fy;
  }
}

double cmio_verner_cross_section(int ion, double energy) {
  /* (g_terms is built when the library is loaded, verner_load below) */
  /* terms of one ion are adjacent and in the reference's summation order */
  double sigma = 0.;
  int first = 1;
  for (int i = 0; i < CMI_VERNER_NTERM; ++i) {
    if (g_terms[i].ion == ion) {
      const double s = verner_term_cross_section(&g_terms[i], energy);
      sigma = first ? s : sigma + s;
      first = 0;
    }
  }
  return sigma;
}

/* a sampled plugin (cmio.h): Utilities::locate's interval
 * (src/Utilities.hpp:726-742), linear or log-log interpolation (the forms of
 * src/HeliumTwoPhotonContinuumSpectrum.cpp:167-180 and
 * src/PlanckPhotonSourceSpectrum.cpp:149-165), the end values outside */
double cmio_table_value(const cmio_table *t, int row, double x) {
  const double *xs = t->x;
  const double *ys = t->y + (size_t)row * (size_t)t->n;
  const uint32_t n = (uint32_t)t->n;
  if (!(x > xs[0]))
    return ys[0];
  if (!(x < xs[n - 1]))
    return ys[n - 1];
  const uint32_t lo = (uint32_t)cmio_locate(x, xs, n);
  const double x0 = xs[lo], x1 = xs[lo + 1], y0 = ys[lo], y1 = ys[lo + 1];
  if (t->interpolation == CMIO_TABLE_LOGLOG && y0 > 0. && y1 > 0. && x0 > 0.)
    return y0 * exp(log(y1 / y0) * (log(x / x0) / log(x1 / x0)));
  return y0 + (y1 - y0) * ((x - x0) / (x1 - x0));
}

double cmio_cross_section(const cmio_model *model, int ion, double frequency) {
  if (model->xsec_type == CMIO_XSEC_FIXED) {
    return model->xsec_fixed[ion];
  }
  if (model->xsec_type == CMIO_XSEC_TABLE) {
    return cmio_table_value(&model->xsec_table, ion, frequency);
  }
  return cmio_verner_cross_section(ion, frequency);
}

/* ------------------------------------------ Verner recombination rates -- */

static double verner_rec_fit(int ion, double T) {
  for (int i = 0; i < CMI_VERNER_NREC; ++i) {
    const cmi_verner_rec *r = &cmi_verner_recs[i];
    if (r->ion != ion)
      continue;
    if (r->kind == 0) {
      /* rnew[2], rnew[3] are stored inverted by the reference constructor */
      const double inv2 = (r->p[2] != 0.) ? 1. / r->p[2] : r->p[2];
      const double inv3 = (r->p[3] != 0.) ? 1. / r->p[3] : r->p[3];
      const double tt = sqrt(T * inv2);
      return r->p[0] / (tt * pow(tt + 1., 1. - r->p[1]) *
                        pow(1. + sqrt(T * inv3), 1. + r->p[1]));
    } else {
      const double tt = T * 1.e-4;
      return r->p[0] * pow(tt, -r->p[1]);
    }
  }
  cmio_set_error("cmio: no recombination fit for ion %d", ion);
  return NAN;
}

/* dielectronic terms, Nussbaumer & Storey form:
 * 1e-12 * (a/T4 + b + c*T4 + d*T4^2) * T4^-1.5 * exp(-f/T4) */
double cmio_verner_recombination_rate(int ion, double temperature) {
  double rate = 0.;
  switch (ion) {
  case CMIO_ION_H_n: {
    const double T1 = temperature / 3.148;
    const double T2 = temperature / 7.036e5;
    rate = 7.982e-11 / (sqrt(T1) * pow(1. + sqrt(T1), 0.252) *
                        pow(1. + sqrt(T2), 1.748));
    break;
  }
  case CMIO_ION_He_n: {
    const double T1 = temperature / 15.54;
    const double T2 = temperature / 3.676e7;
    rate = 3.294e-11 / (sqrt(T1) * pow(1. + sqrt(T1), 0.309) *
                        pow(1. + sqrt(T2), 1.691));
    break;
  }
  case CMIO_ION_C_p1: {
    const double T4 = temperature * 1.e-4;
    const double T4_inv = 1. / T4;
    rate = verner_rec_fit(ion, temperature) +
           1.e-12 *
               (1.8267 * T4_inv + 4.1012 + 4.8443 * T4 + 0.2261 * T4 * T4) *
               pow(T4, -1.5) * exp(-0.5960 * T4_inv);
    break;
  }
  case CMIO_ION_C_p2: {
    const double T4 = temperature * 1.e-4;
    const double T4_inv = 1. / T4;
    rate = verner_rec_fit(ion, temperature) +
           1.e-12 *
               (2.3196 * T4_inv + 10.7328 + 6.8830 * T4 - 0.1824 * T4 * T4) *
               pow(T4, -1.5) * exp(-0.4101 * T4_inv);
    break;
  }
  case CMIO_ION_N_n: {
    const double T4 = temperature * 1.e-4;
    rate = verner_rec_fit(ion, temperature) +
           1.e-12 * (0.6310 + 0.1990 * T4 - 0.0197 * T4 * T4) *
               pow(T4, -1.5) * exp(-0.4398 / T4);
    break;
  }
  case CMIO_ION_N_p1: {
    const double T4 = temperature * 1.e-4;
    const double T4_inv = 1. / T4;
    rate = verner_rec_fit(ion, temperature) +
           1.e-12 *
               (0.0320 * T4_inv - 0.6624 + 4.3191 * T4 + 0.0003 * T4 * T4) *
               pow(T4, -1.5) * exp(-0.5946 * T4_inv);
    break;
  }
  case CMIO_ION_N_p2: {
    const double T4 = temperature * 1.e-4;
    const double T4_inv = 1. / T4;
    rate = verner_rec_fit(ion, temperature) +
           1.e-12 *
               (-0.8806 * T4_inv + 11.2406 + 30.7066 * T4 - 1.1721 * T4 * T4) *
               pow(T4, -1.5) * exp(-0.6127 * T4_inv);
    break;
  }
  case CMIO_ION_O_n: {
    const double T4 = temperature * 1.e-4;
    const double T4_inv = 1. / T4;
    rate = verner_rec_fit(ion, temperature) +
           1.e-12 *
               (-0.0001 * T4_inv + 0.0001 + 0.0956 * T4 + 0.0193 * T4 * T4) *
               pow(T4, -1.5) * exp(-0.4106 * T4_inv);
    break;
  }
  case CMIO_ION_O_p1: {
    const double T4 = temperature * 1.e-4;
    const double T4_inv = 1. / T4;
    rate = verner_rec_fit(ion, temperature) +
           1.e-12 *
               (-0.0036 * T4_inv + 0.7519 + 1.5252 * T4 - 0.0838 * T4 * T4) *
               pow(T4, -1.5) * exp(-0.2769 * T4_inv);
    break;
  }
  case CMIO_ION_Ne_n:
    rate = verner_rec_fit(ion, temperature);
    break;
  case CMIO_ION_Ne_p1: {
    const double T4 = temperature * 1.e-4;
    const double T4_inv = 1. / T4;
    rate = verner_rec_fit(ion, temperature) +
           1.e-12 *
               (0.0129 * T4_inv - 0.1779 + 0.9353 * T4 - 0.0682 * T4 * T4) *
               pow(T4, -1.5) * exp(-0.4156 * T4_inv);
    break;
  }
  case CMIO_ION_S_p1: {
    const double T_in_eV = temperature / 1.16045221e4;
    rate = verner_rec_fit(ion, temperature) +
           1.37e-9 * exp(-14.95 / T_in_eV) * pow(T_in_eV, -1.5);
    break;
  }
  case CMIO_ION_S_p2: {
    const double T_in_eV = temperature / 1.16045221e4;
    const double T_in_eV_inv = 1. / T_in_eV;
    rate = verner_rec_fit(ion, temperature) +
           (8.0729e-9 * exp(-17.56 * T_in_eV_inv) +
            1.1012e-10 * exp(-7.07 * T_in_eV_inv)) *
               pow(T_in_eV, -1.5);
    break;
  }
  case CMIO_ION_S_p3: {
    const double T_inv = 1. / temperature;
    rate = verner_rec_fit(ion, temperature) +
           (5.817e-7 * exp(-362.8 * T_inv) + 1.391e-6 * exp(-1058. * T_inv) +
            1.123e-5 * exp(-7160. * T_inv) + 1.521e-4 * exp(-3.26e4 * T_inv) +
            1.875e-3 * exp(-1.235e5 * T_inv) +
            2.097e-2 * exp(-2.07e5 * T_inv)) *
               pow(temperature, -1.5);
    break;
  }
  default:
    cmio_set_error("cmio: unknown ion %d", ion);
    return NAN;
  }
  rate *= 1.e-6; /* cm^3 s^-1 -> m^3 s^-1 */
  return fmax(0., rate);
}

double cmio_recombination_rate(const cmio_model *model, int ion, double T) {
  if (model->recomb_type == CMIO_RECOMB_FIXED) {
    return model->recomb_fixed[ion];
  }
  if (model->recomb_type == CMIO_RECOMB_TABLE) {
    return cmio_table_value(&model->recomb_table, ion, T);
  }
  return cmio_verner_recombination_rate(ion, T);
}

/* ------------------------------------------------ charge transfer rates -- */

/* All Kingdon & Ferland (1996) style fits share one shape:
 *   a * t^b * (1 + c * exp(d * t)) [* exp(e / t)],  t = clamp(T4, lo, hi)
 * kind: 0 = zero, 1 = constant a, 2 = fit, 3 = fit with the exp(e/t) factor,
 *       4 = a * t * t */
typedef struct {
  int kind;
  double a, b, c, d, e, lo, hi;
} ct_fit;

/* src/ChargeTransferRates.cpp:44-157 */
static const ct_fit ct_recomb_H[CMIO_NION] = {
    [CMIO_ION_H_n] = {-1, 0, 0, 0, 0, 0, 0, 0},
    [CMIO_ION_He_n] = {2, 7.47e-21, 2.06, 9.93, -3.89, 0, 0.6, 10.},
    [CMIO_ION_C_p1] = {2, 1.67e-19, 2.79, 304.74, -4.07, 0, 0.5, 5.},
    [CMIO_ION_C_p2] = {2, 3.25e-15, 0.21, 0.19, -3.29, 0, 0.1, 10.},
    [CMIO_ION_N_n] = {2, 1.01e-18, -0.29, -0.92, -8.38, 0, 0.01, 5.},
    [CMIO_ION_N_p1] = {2, 3.05e-16, 0.6, 2.65, -0.93, 0, 0.1, 10.},
    [CMIO_ION_N_p2] = {2, 4.54e-15, 0.57, -0.65, -0.89, 0, 0.001, 10.},
    [CMIO_ION_O_n] = {2, 1.04e-15, 3.15e-2, -0.61, -9.73, 0, 0.001, 1.},
    [CMIO_ION_O_p1] = {2, 1.04e-15, 0.27, 2.02, -5.92, 0, 0.01, 10.},
    [CMIO_ION_Ne_n] = {0, 0, 0, 0, 0, 0, 0, 0},
    [CMIO_ION_Ne_p1] = {1, 1.e-20, 0, 0, 0, 0, 0, 0},
    [CMIO_ION_S_p1] = {1, 1.e-20, 0, 0, 0, 0, 0, 0},
    [CMIO_ION_S_p2] = {2, 2.29e-15, 4.02e-2, 1.59, -6.06, 0, 0.1, 3.},
    [CMIO_ION_S_p3] = {2, 6.44e-15, 0.13, 2.69, -5.69, 0, 0.1, 3.},
};

/* src/ChargeTransferRates.cpp:169-250 */
static const ct_fit ct_ion_H[CMIO_NION] = {
    [CMIO_ION_H_n] = {-1, 0, 0, 0, 0, 0, 0, 0},
    [CMIO_ION_N_n] = {3, 4.55e-18, -0.29, -0.92, -8.38, -1.086, 0.01, 5.},
    [CMIO_ION_O_n] = {3, 7.4e-17, 0.47, 24.37, -0.74, -0.023, 0.001, 1.},
    /* all others: kind 0 */
};

/* src/ChargeTransferRates.cpp:262-395 */
static const ct_fit ct_recomb_He[CMIO_NION] = {
    [CMIO_ION_He_n] = {-1, 0, 0, 0, 0, 0, 0, 0},
    [CMIO_ION_C_p2] = {4, 4.6e-17, 0, 0, 0, 0, 0.1, 3.},
    [CMIO_ION_N_p1] = {2, 3.3e-16, 0.29, 1.3, -4.5, 0, 0.1, 3.},
    [CMIO_ION_N_p2] = {1, 1.5e-16, 0, 0, 0, 0, 0, 0},
    /* 2.e-16 * t^0.95 and 1.1e-15 * t^0.56: bracket is exactly 1 */
    [CMIO_ION_O_p1] = {2, 2.e-16, 0.95, 0., 0., 0, 0.5, 5.},
    [CMIO_ION_Ne_p1] = {1, 1.e-20, 0, 0, 0, 0, 0, 0},
    [CMIO_ION_S_p2] = {2, 1.1e-15, 0.56, 0., 0., 0, 0.1, 3.},
    [CMIO_ION_S_p3] = {2, 7.6e-19, 0.32, 3.4, -5.25, 0, 0.1, 3.},
};

static double ct_eval(const ct_fit *f, double T4) {
  switch (f->kind) {
  case 0:
    return 0.;
  case 1:
    return f->a;
  case 2: {
    double t = fmax(T4, f->lo);
    t = fmin(t, f->hi);
    return f->a * pow(t, f->b) * (1. + f->c * exp(f->d * t));
  }
  case 3: {
    double t = fmax(T4, f->lo);
    t = fmin(t, f->hi);
    return f->a * pow(t, f->b) * (1. + f->c * exp(f->d * t)) * exp(f->e / t);
  }
  case 4: {
    double t = fmax(T4, f->lo);
    t = fmin(t, f->hi);
    return f->a * t * t;
  }
  default:
    cmio_set_error("cmio: charge transfer of an ion with itself");
    return NAN;
  }
}

double cmio_ct_recombination_rate_H(int ion, double T4) {
  return ct_eval(&ct_recomb_H[ion], T4);
}
double cmio_ct_ionization_rate_H(int ion, double T4) {
  return ct_eval(&ct_ion_H[ion], T4);
}
double cmio_ct_recombination_rate_He(int ion, double T4) {
  return ct_eval(&ct_recomb_He[ion], T4);
}
