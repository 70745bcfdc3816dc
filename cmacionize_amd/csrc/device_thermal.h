/*
 * device_thermal.h - line cooling and the thermal balance of one cell.
 */
#ifndef CMI_DEVICE_THERMAL_H
#define CMI_DEVICE_THERMAL_H

#include "device_physics.h"

/* transition (lower i, upper j) -> index of the 10-entry transition tables */
__device__ __forceinline__ constexpr int lc_tr(int i, int j) {
  return i == 0 ? j - 1 : (i == 1 ? j + 2 : (i == 2 ? j + 4 : 9));
}

/* LineCoolingData::solve_system_of_linear_equations,
 * src/LineCoolingData.cpp:1492-1555: Gaussian elimination with partial
 * pivoting, fully unrolled with static indices so that the 5x5 system stays
 * in registers. Returns false for a singular matrix. */
__device__ inline bool solve_5x5(double (&A)[5][5], double (&B)[5]) {
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    int imax = 0;
    double Amax = 0.;
#pragma unroll
    for (int i = j; i < 5; ++i) {
      if (fabs(A[i][j]) > fabs(Amax)) {
        Amax = A[i][j];
        imax = i;
      }
    }
    if (Amax == 0.)
      return false;
    const double Amax_inv = 1. / Amax;
    /* swap rows j and imax (select instead of dynamic indexing) */
#pragma unroll
    for (int i = j + 1; i < 5; ++i) {
      if (imax == i) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const double t = A[j][k];
          A[j][k] = A[i][k];
          A[i][k] = t;
        }
        const double t = B[j];
        B[j] = B[i];
        B[i] = t;
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
      A[j][k] *= Amax_inv;
    B[j] *= Amax_inv;
#pragma unroll
    for (int i = j + 1; i < 5; ++i) {
#pragma unroll
      for (int k = j + 1; k < 5; ++k)
        A[i][k] -= A[i][j] * A[j][k];
      B[i] -= A[i][j] * B[j];
    }
  }
#pragma unroll
  for (int i = 3; i >= 0; --i) {
#pragma unroll
    for (int j = 4; j > i; --j)
      B[i] -= B[j] * A[i][j];
  }
  return true;
}

/* exp(x) for |x| <= 0.01 to the last bit or two: the series to x^7 (the next
 * term is 2.5e-21), seven fused multiply-adds instead of the ~30 instructions
 * of exp() */
__device__ __forceinline__ double lc_exp_small(double x) {
  double q = 1. / 5040.;
  q = __fma_rn(q, x, 1. / 720.);
  q = __fma_rn(q, x, 1. / 120.);
  q = __fma_rn(q, x, 1. / 24.);
  q = __fma_rn(q, x, 1. / 6.);
  q = __fma_rn(q, x, 0.5);
  q = __fma_rn(q, x, 1.);
  return __fma_rn(q, x, 1.);
}

/* collision strength fit, src/LineCoolingData.cpp:1590-1601. The two powers
 * of T are taken as exp(y ln T) with the logarithm the fit needs anyway: a
 * balance evaluation has 206 of them, and the general pow() is 4-5x the
 * instructions of exp(). Relative difference to pow(): ~|y ln T| ulp ~ 2e-15,
 * far inside the 1e-6 of the reference's own line-cooling test.
 *
 * The fit's last term, a4 T (1 + (a5 - 1) T^a6): a4 is zero for 20 of the 103
 * transitions (the term is then skipped: x + 0 = x), and of the others all
 * but two have |a6| < 3.7e-4, i.e. |a6 ln T| < 0.0086 up to the solve's limit
 * of 1.1e10 K - the series above instead of exp(). The coefficients of a
 * transition are the same for every lane: both tests are scalar branches. */
__device__ __forceinline__ double lc_collision_strength(const double *a,
                                                        double prefactor,
                                                        double T, double Tinv,
                                                        double logT) {
  double fit = a[1] + a[2] * Tinv + a[3] * logT;
  if (a[4] != 0.) {
    const double x = a[6] * logT;
    const double power = fabs(a[6]) < 4.e-4 ? lc_exp_small(x) : exp(x);
    fit += a[4] * T * (1. + (a[5] - 1.) * power);
  }
  return prefactor * exp((1. + a[0]) * logT) * fit;
}

/* LineCoolingData::get_cooling, src/LineCoolingData.cpp:1767-1847 with
 * compute_level_populations (:1569-1701) and compute_level_population
 * (:1714-1738). abund[k * abund_stride], k < 13: NI NII OI OII OIII NeIII SII
 * SIII CII CIII NIII NeII SIV number fractions relative to H (the stride lets
 * a kernel keep one such list per thread in LDS instead of 26 registers).
 *
 * The rate matrix of a five-level ion is filled transition by transition:
 * transition (lo, hi) with collision rates down (hi -> lo) and up = down x
 * Boltzmann factor contributes to four entries; nothing but the matrix, the
 * right-hand side and five per-level sums is live across the ten transitions
 * (a first version kept all 20 rates and spilled ~700 registers per lane:
 * the temperature kernel moved 340 GB of scratch per 256^3 update). The sums
 * run in a different order than the reference's sumC (same terms). */
/* the cooling of five-level ion e per unit of abundance and kb:
 * sum_i pop_i sum_j A_ji E_ji (the body of line_cooling's first loop) */
template <bool COLD>
__device__ __forceinline__ double
lc_five_level_cooling_at(const LineCoolingDev &lc, int e, double prefactor,
                         double temperature, double Tinv, double logT) {
  const double *A = lc.A[e];
  const double *w = lc.inv_weight[e];
  double M[5][5];
  double pop[5] = {1., 0., 0., 0., 0.};
  double sumC[5] = {0., 0., 0., 0., 0.}; /* collisions out of each level */
  /* the Boltzmann factor of transition (lo, hi), exp(-(E_hi - E_lo) / T), as
   * the quotient of the levels' factors: four exponentials and three
   * reciprocals per ion instead of ten exponentials (the transition energies
   * ARE differences of level energies, src/LineCoolingData.cpp:76-97 and
   * likewise for every ion;
   * relative difference ~E / T x 1e-16) */
  double level[5], level_inv[4];
  level[0] = level_inv[0] = 1.;
  if (!COLD) {
#pragma unroll
    for (int k = 1; k < 5; ++k) {
      level[k] = exp(-lc.energy[e][lc_tr(0, k)] * Tinv);
      if (k < 4)
        level_inv[k] = 1. / level[k];
    }
  }
#pragma unroll
  for (int lo = 0; lo < 4; ++lo) {
#pragma unroll
    for (int hi = lo + 1; hi < 5; ++hi) {
      const int t = lc_tr(lo, hi);
      const double down = lc_collision_strength(lc.cs[e][t], prefactor,
                                                temperature, Tinv, logT);
      const double boltzmann =
          COLD ? exp(-lc.energy[e][t] * Tinv)
               : (lo == 0 ? level[hi] : level[hi] * level_inv[lo]);
      const double up = down * boltzmann;
      /* level hi is fed from lo, level lo from hi (row 0 is replaced by
       * the normalisation below) */
      M[hi][lo] = up * w[lo];
      if (lo > 0)
        M[lo][hi] = A[t] + w[hi] * down;
      sumC[hi] += down;
      sumC[lo] += up;
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k)
    M[0][k] = 1.; /* populations sum to 1 */
#pragma unroll
  for (int i = 1; i < 5; ++i) {
    double sumA = A[lc_tr(0, i)];
#pragma unroll
    for (int j = 1; j < i; ++j)
      sumA += A[lc_tr(j, i)];
    M[i][i] = -(sumA + w[i] * sumC[i]);
  }
  /* a singular matrix aborts the reference (cmac_error); here the ion then
   * contributes the populations of the unsolved right-hand side */
  (void)solve_5x5(M, pop);
  const double *E = lc.energy[e];
  double cl[5];
  cl[1] = pop[1] * A[lc_tr(0, 1)] * E[lc_tr(0, 1)];
#pragma unroll
  for (int i = 2; i < 5; ++i) {
    double s = A[lc_tr(0, i)] * E[lc_tr(0, i)];
#pragma unroll
    for (int j = 1; j < i; ++j)
      s += A[lc_tr(j, i)] * E[lc_tr(j, i)];
    cl[i] = pop[i] * s;
  }
  return cl[1] + cl[2] + cl[3] + cl[4];
}

/* (COLD: a temperature so low that a level's factor leaves the normal range
 * - below ~130 K, never reached by the solve, possible through the probes and
 * with a t_min_ionized set that low -: every transition's own exponential, as
 * the reference has it; a path of its own so that the other stays short) */
__device__ __attribute__((noinline)) double
lc_five_level_cooling_cold(const LineCoolingDev &lc, int e, double prefactor,
                           double temperature, double Tinv, double logT) {
  return lc_five_level_cooling_at<true>(lc, e, prefactor, temperature, Tinv,
                                        logT);
}
__device__ __forceinline__ double
lc_five_level_cooling(const LineCoolingDev &lc, int e, double prefactor,
                      double temperature, double Tinv, double logT) {
  if (__builtin_expect(lc.energy[e][lc_tr(0, 4)] * Tinv > 600., 0))
    return lc_five_level_cooling_cold(lc, e, prefactor, temperature, Tinv,
                                      logT);
  return lc_five_level_cooling_at<false>(lc, e, prefactor, temperature, Tinv,
                                         logT);
}

/* two-level ion i: the population of its upper level (the body of
 * line_cooling's second loop) */
__device__ __forceinline__ double
lc_two_level_cooling(const LineCoolingDev &lc, int i, double prefactor,
                     double temperature, double Tinv, double logT) {
  const double ksi = lc.two_energy[i];
  const double cs = lc_collision_strength(lc.two_cs[i], prefactor, temperature,
                                          Tinv, logT);
  const double Texp = exp(-ksi * Tinv);
  const double pop =
      cs * Texp * lc.two_inv_weight[i][0] /
      (lc.two_A[i] +
       cs * (lc.two_inv_weight[i][1] + Texp * lc.two_inv_weight[i][0]));
  return pop;
}

/* (the body, for a kernel that wants it in its own register budget:
 * temp_linecool_kernel; everybody else calls line_cooling below) */
__device__ __forceinline__ double
line_cooling_inlined(const LineCoolingDev &lc, double temperature,
                     double electron_density, const double *abund,
                     int abund_stride) {
  if (electron_density == 0.)
    return 1.e-99;
  const double kb = CMI_BOLTZMANN;
  const double prefactor = lc.prefactor * electron_density / sqrt(temperature);
  const double Tinv = 1. / temperature;
  const double logT = log(temperature);

  double cooling = 0.;
  /* (one loop body for the ten ions: unrolled, the ~300 exp() expansions of a
   * balance evaluation alone are 90 KB of code - with three evaluations per
   * secant step inlined the solve was 286 KB and ran out of the instruction
   * cache) */
#pragma unroll 1
  for (int e = 0; e < CMI_LC_NFIVE_DEV; ++e)
    cooling += abund[e * abund_stride] * kb *
               lc_five_level_cooling(lc, e, prefactor, temperature, Tinv, logT);
#pragma unroll 1
  for (int i = 0; i < CMI_LC_NTWO_DEV; ++i)
    cooling += abund[(CMI_LC_NFIVE_DEV + i) * abund_stride] * kb *
               lc.two_energy[i] * lc.two_A[i] *
               lc_two_level_cooling(lc, i, prefactor, temperature, Tinv, logT);
  return cooling;
}

__device__ inline double line_cooling(const LineCoolingDev &lc,
                                      double temperature,
                                      double electron_density,
                                      const double *abund, int abund_stride) {
  return line_cooling_inlined(lc, temperature, electron_density, abund,
                              abund_stride);
}

/* the integrals of a cell as the solve reads them: J[k * stride], normalised
 * on the fly (14 registers less than a normalised copy) */
struct CellIntegrals {
  const double *J;
  int64_t stride;
  double jfac;
  /* J points at a cell's row of the AoS accumulator block: its 16 values
   * stand in threshold order (cmi_acc_column), not in the ions' */
  bool row = false;
  __device__ __forceinline__ double operator()(int ion) const {
    return jfac * J[row ? cmi_acc_column(ion) : ion * stride];
  }
};

/* TemperatureCalculator::compute_cooling_and_heating_balance,
 * src/TemperatureCalculator.cpp:207-501. j: normalised mean intensities,
 * h[2]: normalised heating terms; x[2..13] receive the metal fractions at
 * temperature T; abund / abund_stride: 13 doubles of work space
 * (line_cooling). */
__device__ inline void cooling_and_heating_balance(
    const ModelDev &m, double &h0, double &he0, double &gain, double &loss,
    double T, double n, double midpoint_z, const CellIntegrals &j,
    const double h[2], double x[CMI_NION], double *abund, int abund_stride) {
  enum { NI = 0, NII, OI, OII, OIII, NeIII, SII, SIII, CII, CIII, NIII, NeII,
         SIV };
  const double alphaH = cmi_recombination_rate(m, ION_H_n, T);
  const double alphaHe = cmi_recombination_rate(m, ION_He_n, T);
  const double T4 = T * 1.e-4;
  const double sqrtT = sqrt(T);
  const double logT = log(T);
  const double AHe = m.abundance[0];

  cmi_ionization_states_hydrogen_helium(alphaH, alphaHe, j(ION_H_n),
                                        j(ION_He_n), n, AHe, T, h0, he0);
  const double ne = n * (1. - h0 + AHe * (1. - he0));
  const double nhp = n * (1. - h0);
  const double nhep = (1. - he0) * n * AHe;
  const double nenhp = ne * nhp;
  const double nenhep = ne * nhep;

  gain = n * (h[0] * h0 + h[1] * AHe * he0);
  /* T4^-0.861 and T^0.3647 from the logarithm at hand (ln 1e4 = 9.2103...) */
  const double alpha_e_2sP =
      4.17e-20 * exp(-0.861 * (logT - 9.210340371976184));
  const double pHots = 1. / (1. + 77. * he0 / (sqrtT * h0));
  gain += pHots * 1.21765423e-18 * alpha_e_2sP * nenhep;
  gain += 1.5e-37 * n * ne * m.pahfac;
  double heatcr = 0.;
  if (m.crfac > 0.) {
    heatcr = m.crfac * 1.2e-25 / sqrt(ne);
    if (m.crscale > 0.)
      heatcr *= exp(-fabs(midpoint_z) / m.crscale);
  }
  gain += heatcr;

  const double nh0 = n * h0;
  const double nhe0 = n * he0 * AHe;
  cmi_ionization_states_metals(m, j, ne, T, T4, nh0, nhe0, nhp, x);

  const double AC = m.abundance[1], AN = m.abundance[2], AO = m.abundance[3],
               ANe = m.abundance[4], AS = m.abundance[5];
#define AB(k) abund[(k)*abund_stride]
  AB(CII) = AC * (1. - x[ION_C_p1] - x[ION_C_p2]);
  AB(CIII) = AC * x[ION_C_p1];
  AB(NI) = AN * (1. - x[ION_N_n] - x[ION_N_p1] - x[ION_N_p2]);
  AB(NII) = AN * x[ION_N_n];
  AB(NIII) = AN * x[ION_N_p1];
  AB(OI) = AO * (1. - x[ION_O_n] - x[ION_O_p1]);
  AB(OII) = AO * x[ION_O_n];
  AB(OIII) = AO * x[ION_O_p1];
  AB(NeII) = ANe * x[ION_Ne_n];
  AB(NeIII) = ANe * x[ION_Ne_p1];
  AB(SII) = AS * (1. - x[ION_S_p1] - x[ION_S_p2] - x[ION_S_p3]);
  AB(SIII) = AS * x[ION_S_p1];
  AB(SIV) = AS * x[ION_S_p2];
#undef AB

  loss = line_cooling(m.tables->lc, T, ne, abund, abund_stride) * n;
  const double c = 5.5 - logT;
  const double gff = 1.1 + 0.34 * exp(-c * c / 3.);
  loss += 1.42e-40 * gff * sqrtT * (nenhp + nenhep);
  const double Lhp =
      2.85e-40 * nenhp * sqrtT * (5.914 - 0.5 * logT + 0.01184 * cbrt(T));
  const double Lhep = 1.55e-39 * nenhep * exp(0.3647 * logT);
  loss += Lhp + Lhep;
  loss = fmax(loss, 0.);
  gain = fmax(gain, 0.);
}

/* TemperatureCalculator::calculate_temperature(vars, jfac, hfac, midpoint),
 * src/TemperatureCalculator.cpp:567-931, in three pieces so that a kernel can
 * interleave the secant steps of different cells (temperature_kernel): the
 * state of one cell's solve between two steps ... */
struct TemperatureSolve {
  double T0, gain0, loss0, h0, he0;
  double Tlast; /* temperature of the last balance evaluation */
  double h[2];  /* normalised heating terms */
  int32_t niter;
};

/* ... its start (:567-640): returns false for a cell that needs no solve (no
 * radiation, vacuum, or too neutral for cosmic ray heating: 500 K, H and He
 * neutral, all metal fractions - including N0, O0, Ne0 - zero; the outputs
 * are then final) ... */
__device__ inline bool temperature_begin(const ModelDev &m,
                                         const CellIntegrals &j, double hfac,
                                         double ntot, double &temperature,
                                         double heating[2],
                                         double x[CMI_NION],
                                         TemperatureSolve &s) {
  const double jH = j(ION_H_n);
  const double jHe = j(ION_He_n);
  bool neutral = (jH == 0. && jHe == 0.) || ntot == 0.;
  if (!neutral && m.crfac > 0.) {
    double h0, he0;
    const double alphaH = cmi_recombination_rate(m, ION_H_n, 8000.);
    const double alphaHe = cmi_recombination_rate(m, ION_He_n, 8000.);
    cmi_ionization_states_hydrogen_helium(alphaH, alphaHe, jH, jHe, ntot,
                                          m.abundance[0], 8000., h0, he0);
    neutral = h0 > m.crlim;
  }
  if (neutral) {
    temperature = 500.;
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      x[i] = 0.;
    x[ION_H_n] = 1.;
    x[ION_He_n] = 1.;
    heating[0] = 0.;
    heating[1] = 0.;
    return false;
  }
  s.T0 = (temperature <= 4000.) ? 8000. : temperature;
  s.Tlast = s.T0;
  s.h[0] = hfac * heating[0];
  s.h[1] = hfac * heating[1];
  s.niter = 0;
  s.gain0 = 1.;
  s.loss0 = 0.;
  s.h0 = 0.;
  s.he0 = 0.;
  return true;
}

/* ... the loop condition (:652-653) ... */
__device__ __forceinline__ bool temperature_goes_on(const ModelDev &m,
                                                    const TemperatureSolve &s) {
  return fabs(s.gain0 - s.loss0) > m.t_epsilon * s.gain0 &&
         s.niter < m.t_max_iterations;
}

/* ... one secant step (:654-745): the three balance evaluations - at 1.1 T0,
 * 0.9 T0 and T0, in the reference's order - as three trips of one loop body
 * ... */
__device__ inline void temperature_step(const ModelDev &m, double ntot,
                                        double midpoint_z,
                                        const CellIntegrals &j,
                                        TemperatureSolve &s, double *abund,
                                        int abund_stride) {
  const double logtt = log(1.1 / 0.9);
  ++s.niter;
  const double T0 = s.T0;
  const double T1 = 1.1 * T0;
  s.Tlast = T0;
  double gain1 = 0., loss1 = 0., gain2 = 0., loss2 = 0.;
#pragma unroll 1
  for (int k = 0; k < 3; ++k) {
    const double Tk = (k == 0) ? T1 : ((k == 1) ? 0.9 * T0 : T0);
    double h0k, he0k, gaink, lossk;
    double x[CMI_NION]; /* the metal fractions at Tk: only feed the cooling */
    cooling_and_heating_balance(m, h0k, he0k, gaink, lossk, Tk, ntot,
                                midpoint_z, j, s.h, x, abund, abund_stride);
    if (k == 0) {
      gain1 = gaink;
      loss1 = lossk;
    } else if (k == 1) {
      gain2 = gaink;
      loss2 = lossk;
    } else {
      s.h0 = h0k;
      s.he0 = he0k;
      s.gain0 = gaink;
      s.loss0 = lossk;
    }
  }
  double expgain;
  if (gain2 > 0.)
    expgain = (gain1 > 0.) ? log(gain1 / gain2) : -99.;
  else
    expgain = (gain1 > 0.) ? 99. : 0.;
  double exploss;
  if (loss2 > 0.)
    exploss = (loss1 > 0.) ? log(loss1 / loss2) : -99.;
  else
    exploss = (loss1 > 0.) ? 99. : 0.;
  const double expdiff = expgain - exploss;
  if (s.gain0 > 0. && expdiff != 0.)
    s.T0 = T0 * pow(s.loss0 / s.gain0, logtt / expdiff);
  else
    s.T0 = T1;
  if (s.T0 < m.t_min_ionized) {
    s.T0 = 500.;
    s.h0 = 1.;
    s.he0 = 1.;
    s.gain0 = 1.;
    s.loss0 = 1.;
  }
  if (s.T0 > 1.e10) {
    s.T0 = 1.e10;
    s.h0 = 1.e-10;
    s.he0 = 1.e-10;
    s.gain0 = 1.;
    s.loss0 = 1.;
  }
}

/* ... and its end (:747-931). The reference leaves in the cell the metal
 * fractions of the LAST balance evaluation (at Tlast, with that evaluation's
 * h0, he0); they are evaluated here, once, from the same inputs by the same
 * code instead of being carried through every step. */
__device__ inline void temperature_end(const ModelDev &m, double ntot,
                                       const CellIntegrals &j,
                                       const TemperatureSolve &s,
                                       double &temperature, double heating[2],
                                       double x[CMI_NION]) {
  temperature = fmin(30000., s.T0);
  double h0 = s.h0, he0 = s.he0;
  const bool clamped = (h0 == 1. || h0 <= 1.e-10); /* T out of range */
  if (!clamped && s.niter > 0) {
    const double AHe = m.abundance[0];
    const double ne = ntot * (1. - h0 + AHe * (1. - he0));
    const double nhp = ntot * (1. - h0);
    const double nh0 = ntot * h0;
    const double nhe0 = ntot * he0 * AHe;
    cmi_ionization_states_metals(m, j, ne, s.Tlast, s.Tlast * 1.e-4, nh0, nhe0,
                                 nhp, x);
  }
  if (j.J[ION_H_n * j.stride] == 0.)
    h0 = 1.;
  if (j.J[ION_He_n * j.stride] == 0.)
    he0 = 1.;
  x[ION_H_n] = h0;
  x[ION_He_n] = he0;
  if (h0 == 1. || h0 <= 1.e-10) {
#pragma unroll
    for (int i = ION_C_p1; i < CMI_NION; ++i)
      x[i] = 0.;
  }
  heating[0] = s.h[0];
  heating[1] = s.h[1];
}

/* one cell from start to end. J: un-normalised integrals (J.jfac normalises
 * them), heating[2] un-normalised in, normalised out; temperature in/out;
 * x[14] in (kept if no iteration runs) / out. */
__device__ inline void temperature_cell(const ModelDev &m,
                                        const CellIntegrals &J, double hfac,
                                        double ntot, double midpoint_z,
                                        double &temperature,
                                        double heating[2],
                                        double x[CMI_NION]) {
  TemperatureSolve s;
  double abund[13];
  if (!temperature_begin(m, J, hfac, ntot, temperature, heating, x, s))
    return;
  while (temperature_goes_on(m, s))
    temperature_step(m, ntot, midpoint_z, J, s, abund, 1);
  temperature_end(m, ntot, J, s, temperature, heating, x);
}

#endif
