/*
 * device_physics.h - device functions: packet RNG, atomic data, ionization
 * balance. Each function cites the reference code it implements.
 */
#ifndef CMI_DEVICE_PHYSICS_H
#define CMI_DEVICE_PHYSICS_H

#include "device_common.h"

/* ---------------------------------------------------------------- RNG -- */

/* Philox4x32-10 counter-based generator. Replaces the per-thread sequential
 * ranlxd2 stream of src/RandomGenerator.hpp:39-272: a packet's uniforms are a
 * pure function of (seed, iteration, packet id, draw index), independent of
 * which lane runs it. */
struct PacketRng {
  uint32_t seed, iteration;
  uint32_t p_lo, p_hi;
  uint32_t block; /* next Philox block to generate (= draws consumed / 2) */
  uint32_t have;  /* the second double of the last block is still unused */
  double cached;

  __device__ __forceinline__ void init(uint32_t seed_, uint32_t iteration_,
                                       uint64_t packet) {
    seed = seed_;
    iteration = iteration_;
    p_lo = (uint32_t)packet;
    p_hi = (uint32_t)(packet >> 32);
    block = 0;
    have = 0;
    cached = 0.;
  }

  __device__ __forceinline__ static double to_unit(uint32_t lo, uint32_t hi) {
    const uint64_t bits = (((uint64_t)hi << 32) | lo) >> 12;
    /* 52 random bits + 1/2, scaled: strictly inside (0,1), exact in fp64 */
    return ((double)bits + 0.5) * 0x1.0p-52;
  }

  /* resume a stream that has consumed `block_` blocks; if `have_` the second
   * double of block block_ - 1 is still unused and is regenerated */
  __device__ __forceinline__ void resume(uint32_t seed_, uint32_t iteration_,
                                         uint64_t packet, uint32_t block_,
                                         uint32_t have_) {
    init(seed_, iteration_, packet);
    if (have_) {
      block = block_ - 1;
      (void)next(); /* regenerates the block, caches its second double */
    } else {
      block = block_;
    }
  }

  /* draw d of the packet = word pair (d & 1) of block d / 2; draws are
   * consumed strictly in order, so one cached double is enough */
  __device__ __forceinline__ double next() {
    if (have) {
      have = 0;
      return cached;
    }
    uint32_t c0 = p_lo, c1 = p_hi, c2 = block, c3 = 0u;
    uint32_t k0 = seed, k1 = iteration;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      /* (64-bit products: one v_mad_u64_u32 each instead of v_mul_hi_u32 +
       * v_mul_lo_u32 - a block costs a SIMD 97-106 ns instead of 143,
       * tools/microbench/philox_mul.hip) */
      const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
      const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
      const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
      const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
      c0 = hi1 ^ c1 ^ k0;
      c1 = lo1;
      c2 = hi0 ^ c3 ^ k1;
      c3 = lo0;
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    ++block;
    cached = to_unit(c2, c3);
    have = 1;
    return to_unit(c0, c1);
  }
};

/* src/Utilities.hpp:726-742 (bisection, result in [0, length-2]) */
__device__ __forceinline__ uint32_t cmi_locate(double x, const double *arr,
                                               uint32_t length) {
  uint32_t lo = 0, hi = length;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (x > arr[mid])
      lo = mid;
    else
      hi = mid;
  }
  return (lo == length - 1) ? lo - 1 : lo;
}

/* the same index for a uniform x in (0, 1) and a cumulative distribution
 * with its guide table (SpectraDev): the bisection from the bracket the guide
 * gives. Invariants of the bisection - arr[lo] < x or lo == 0, arr[hi] >= x or
 * hi == length - hold for the bracket: arr[guide[k]] < k / G <= x and
 * arr[guide[k + 1] + 1] >= (k + 1) / G > x. */
__host__ __device__ __forceinline__ uint32_t
cmi_locate_guided(double x, const double *arr, const uint16_t *guide,
                  uint32_t length) {
  const uint32_t k = (uint32_t)(x * CMI_NGUIDE);
  uint32_t lo = guide[k], hi = (uint32_t)guide[k + 1] + 1u;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (x > arr[mid])
      lo = mid;
    else
      hi = mid;
  }
  return (lo == length - 1) ? lo - 1 : lo;
}

/* ... and for a table that is close to linear in its index (the temperatures
 * of the Lyman continuum spectra): the index a linear table would give, moved
 * to where the bisection ends - the last entry below x (0 if none) */
__host__ __device__ __forceinline__ uint32_t
cmi_locate_linear(double x, const double *arr, uint32_t length) {
  const double guess =
      (x - arr[0]) * ((double)(length - 1) / (arr[length - 1] - arr[0]));
  uint32_t g = guess > 0. ? (guess < (double)(length - 1) ? (uint32_t)guess
                                                          : length - 1)
                          : 0u;
  while (g > 0 && !(x > arr[g]))
    --g;
  while (g < length - 1 && x > arr[g + 1])
    ++g;
  return (g == length - 1) ? g - 1 : g;
}

/* ------------------------------------------- caller-supplied tables ---- */

/* the table modes are the rare case: the register allocator should place
 * whatever it has to spill for them on their side of the branch (inlined
 * without the hint, the table path cost the key, weights and temperature
 * kernels 20-50 scalar spills each and direction_key_kernel a wave per SIMD) */
#ifdef CMI_EXP_NO_TABLE_MODES /* experiment: what the table modes cost the others */
#define CMI_UNLIKELY(c) (false && (c))
#else
#define CMI_UNLIKELY(c) __builtin_expect(!!(c), 0)
#endif

/* Row `row` of a table at abscissa x: Utilities::locate's interval
 * (src/Utilities.hpp:726-742), linear interpolation in (x, y) or in
 * (log x, log y) - a power law between two samples, the form the reference's
 * Planck table is sampled in (src/PlanckPhotonSourceSpectrum.cpp:149-165) -;
 * outside [x[0], x[n - 1]] the end values (no extrapolation). A log-log
 * interval with a sample that is not positive falls back to linear. */
/* ... in two steps, so that the 14 rows of a cross-section table share one
 * search: where x lies (the interval and the weight of its upper end), then a
 * row's value there */
struct TableAt {
  uint32_t lo;   /* interval [lo, lo + 1] */
  double linear; /* (x - x0) / (x1 - x0), 0 / 1 outside the table */
  double loglog; /* log(x / x0) / log(x1 / x0) where that exists, else < 0 */
};
__host__ __device__ inline TableAt cmi_table_locate(const TableDev &t,
                                                    double x) {
  const double *xs = t.x;
  const uint32_t n = (uint32_t)t.n;
  TableAt at;
  at.loglog = -1.;
  if (!(x > xs[0])) {
    at.lo = 0;
    at.linear = 0.;
    return at;
  }
  if (!(x < xs[n - 1])) {
    at.lo = n - 2;
    at.linear = 1.;
    return at;
  }
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (x > xs[mid])
      lo = mid;
    else
      hi = mid;
  }
  if (lo == n - 1)
    --lo;
  const double x0 = xs[lo], x1 = xs[lo + 1];
  at.lo = lo;
  at.linear = (x - x0) / (x1 - x0);
  if (t.interpolation == CMI_TABLE_LOGLOG && x0 > 0.)
    at.loglog = log(x / x0) / log(x1 / x0);
  return at;
}
__host__ __device__ inline double cmi_table_row(const TableDev &t, int row,
                                                const TableAt &at) {
  const double *ys = t.y + (size_t)row * (size_t)t.n;
  const double y0 = ys[at.lo], y1 = ys[at.lo + 1];
  if (at.linear <= 0.)
    return y0;
  if (at.linear >= 1.)
    return y1;
  if (at.loglog >= 0. && y0 > 0. && y1 > 0.)
    return y0 * exp(log(y1 / y0) * at.loglog);
  return y0 + (y1 - y0) * at.linear;
}
__host__ __device__ inline double cmi_table_value(const TableDev &t, int row,
                                                 double x) {
  return cmi_table_row(t, row, cmi_table_locate(t, x));
}

/* ------------------------------------------------ Verner cross section -- */

/* VernerCrossSections::get_cross_section_verner,
 * src/VernerCrossSections.cpp:166-245, for one (ion, shell) term */
__host__ __device__ inline double verner_term_sigma(const VernerTermDev &t,
                                                    double e) {
  if (e < t.E_th)
    return 0.;
  const int is = t.shell;
  const int nout = t.ntot;
  if (is > nout)
    return 0.;
  const int nint = t.ninn;
  const double einn = t.einn;
  if (is < nout && is > nint && e < einn)
    return 0.;
  if (is <= nint || e >= einn) {
    const double y = e * t.A_E_0_inv;
    const double ym1 = y - 1.;
    /* y^a (1 + sqrt(y / y_a))^-P as ONE exponential of two logarithms: a
     * third of the instructions of two pow() (each of which is a logarithm
     * and an exponential in extended precision), relative difference
     * ~|a ln y| x 1e-16 < 1e-14 (round 4: the key kernel of multi-ion runs
     * spends its 22 ms per launch here) */
    const double Fy = (ym1 * ym1 + t.A_y_w_sq) *
                      exp(t.A_Plconst * log(y) -
                          t.A_P * log(1. + sqrt(y * t.A_y_a_inv)));
    return t.A_sigma_0 * Fy;
  } else {
    const double x = e * t.B_E_0_inv - t.B_y_0;
    const double y = sqrt(x * x + t.B_y_1_sq);
    const double xm1 = x - 1.;
    const double Fy = (xm1 * xm1 + t.B_y_w_sq) *
                      exp((0.5 * t.B_P - 5.5) * log(y) -
                          t.B_P * log(1. + sqrt(y * t.B_y_a_inv)));
    return t.B_sigma_0 * Fy;
  }
}

/* all 14 cross sections of a packet: PhotonSource::set_cross_sections,
 * src/PhotonSource.cpp:189-199 with CrossSections::get_cross_section
 * (src/VernerCrossSections.cpp:259-322 or FixedValueCrossSections) */
/* (host + device: the host uses it to tabulate the Lyman continua) */
__host__ __device__ inline void cmi_cross_sections(const ModelDev &m, double nu,
                                                   double sigma[CMI_NION]) {
  if (!m.xsec_verner) {
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      sigma[i] = m.xsec_fixed[i];
    return;
  }
  if (CMI_UNLIKELY(m.xsec_verner == 2)) {
    /* a plugin known only through CrossSections::get_cross_section
     * (src/CrossSections.hpp:49-50), sampled into a table by the host */
    const TableAt at = cmi_table_locate(m.xsec_table, nu);
#pragma unroll 1
    for (int k = 0; k < CMI_NION; ++k) {
      const double s = cmi_table_row(m.xsec_table, k, at);
#pragma unroll
      for (int i = 0; i < CMI_NION; ++i)
        if (i == k)
          sigma[i] = s;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < CMI_NION; ++i)
    sigma[i] = 0.;
  const VernerTermDev *terms = m.tables->verner;
#ifndef CMI_EXP_VERNER_TERMS /* (timing experiments: fewer terms, wrong sums) */
#define CMI_EXP_VERNER_TERMS CMI_VERNER_NTERM_DEV
#endif
  /* (one copy of the fit in the code, not 22: the unrolled loop was 100 KB of
   * instructions per call site - more than the instruction cache holds - and
   * most of the library's compile time) */
#pragma unroll 1
  for (int k = 0; k < CMI_EXP_VERNER_TERMS; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
    /* a term none of the wave's photons reaches is a jump, not a masked-off
     * fit (the callers that evaluate all 14 cross sections order their
     * photons by verner_class so that this is the rule) */
    if (__builtin_amdgcn_ballot_w64(nu >= terms[k].E_th) == 0ull)
      continue;
#endif
    const double s = verner_term_sigma(terms[k], nu);
    const int ion = terms[k].ion;
    /* static indexing keeps sigma[] in registers */
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      if (i == ion)
        sigma[i] += s;
  }
}

/* how many of the fits' thresholds a photon lies above: photons of a class
 * skip the same terms */
__device__ inline uint32_t verner_class(const ModelDev &m, double nu) {
  if (m.xsec_verner != 1)
    return 0;
  const VernerTermDev *terms = m.tables->verner;
  uint32_t c = 0;
  for (int k = 0; k < CMI_VERNER_NTERM_DEV; ++k)
    c += (nu >= terms[k].E_th) ? 1u : 0u;
  return c;
}
#define CMI_VERNER_NCLASS (CMI_VERNER_NTERM_DEV + 1)

/* the cross sections of H0 and He0 alone (the re-emission decision needs no
 * others): same terms, same order of summation as cmi_cross_sections */
__host__ __device__ inline void cmi_cross_sections_H_He(const ModelDev &m, double nu,
                                               double &sigma_H,
                                               double &sigma_He) {
  if (!m.xsec_verner) {
    sigma_H = m.xsec_fixed[ION_H_n];
    sigma_He = m.xsec_fixed[ION_He_n];
    return;
  }
  if (CMI_UNLIKELY(m.xsec_verner == 2)) {
    const TableAt at = cmi_table_locate(m.xsec_table, nu);
    sigma_H = cmi_table_row(m.xsec_table, ION_H_n, at);
    sigma_He = cmi_table_row(m.xsec_table, ION_He_n, at);
    return;
  }
  sigma_H = 0.;
  sigma_He = 0.;
  const VernerTermDev *terms = m.tables->verner;
#pragma unroll 1
  for (int k = 0; k < CMI_VERNER_NTERM_DEV; ++k) {
    const int ion = terms[k].ion; /* wave-uniform: the table is */
    if (ion == ION_H_n)
      sigma_H += verner_term_sigma(terms[k], nu);
    else if (ion == ION_He_n)
      sigma_He += verner_term_sigma(terms[k], nu);
  }
}

/* -------------------------------------------------- recombination rates -- */

/* RecombinationRates::get_recombination_rate,
 * src/VernerRecombinationRates.cpp:140-333, from the coefficient rows of
 * VernerRecDev: one code path for all 14 ions. */
__device__ inline double cmi_recombination_rate(const ModelDev &m, int ion,
                                                double temperature) {
  if (!m.recomb_verner)
    return m.recomb_fixed[ion];
  if (CMI_UNLIKELY(m.recomb_verner == 2))
    /* RecombinationRates::get_recombination_rate
     * (src/RecombinationRates.hpp:49) sampled into a table by the host */
    return cmi_table_value(m.recomb_table, ion, temperature);
  const VernerRecDev &r = m.tables->verner_rec[ion];
  /* powers as exp(y ln x) and x^-1.5 as 1 / (x sqrt x): a few 1e-16 from
   * pow() (the reference's test of these fits passes at 1e-13 here), a
   * fraction of its instructions - the temperature solve evaluates all of
   * them three times per secant step */
  double rate;
  if (r.kind == 0) {
    const double tt = sqrt(temperature * r.p[2]);
    rate = r.p[0] /
           (tt * exp((1. - r.p[1]) * log(tt + 1.) +
                     (1. + r.p[1]) * log(1. + sqrt(temperature * r.p[3]))));
  } else {
    rate = r.p[0] * exp(-r.p[1] * log(temperature * 1.e-4));
  }
  if (r.dkind == 1) {
    const double T4 = temperature * 1.e-4;
    const double T4_inv = 1. / T4;
    rate += 1.e-12 *
            (r.d[0] * T4_inv + r.d[1] + r.d[2] * T4 + r.d[3] * T4 * T4) *
            (T4_inv / sqrt(T4)) * exp(-r.d[4] * T4_inv);
  } else if (r.dkind == 2) {
    const double t = temperature * r.dunit;
    const double t_inv = 1. / t;
    double sum = 0.;
    for (int k = 0; k < r.dn; ++k)
      sum += r.dc[k] * exp(-r.dE[k] * t_inv);
    rate += sum * (t_inv / sqrt(t));
  }
  rate *= 1.e-6;
  return fmax(0., rate);
}

/* ------------------------------------------------------ charge transfer -- */

/* src/ChargeTransferRates.cpp:44-395, table driven */
__device__ inline double ct_eval(const CTFitDev &f, double T4) {
  if (f.kind == 0)
    return 0.;
  if (f.kind == 1)
    return f.a;
  double t = fmax(T4, f.lo);
  t = fmin(t, f.hi);
  if (f.kind == 4)
    return f.a * t * t;
  const double base = f.a * exp(f.b * log(t)) * (1. + f.c * exp(f.d * t));
  if (f.kind == 3)
    return base * exp(f.e / t);
  return base;
}

/* ---------------------------------------------------- ionization balance -- */

/* src/IonizationStateCalculator.cpp:802-820 */
__device__ inline double cmi_ionization_state_hydrogen(double alphaH, double jH,
                                                       double nH) {
  if (jH > 0. && nH > 0.) {
    const double aa = 0.5 * jH / (nH * alphaH);
    const double bb = 2. / aa;
    if (bb < 1.e-10) {
      return fmax(1.e-14, 0.25 * bb);
    } else {
      const double cc = sqrt(bb + 1.);
      return fmax(1.e-14, 1. + aa * (1. - cc));
    }
  }
  return 1.;
}

/* src/IonizationStateCalculator.cpp:649-753 */
__device__ inline void cmi_ionization_states_hydrogen_helium(
    double alphaH, double alphaHe, double jH, double jHe, double nH,
    double AHe, double T, double &h0, double &he0) {
  if (jH < 1.e-20) {
    h0 = 1.;
    he0 = 1.;
    return;
  }
  const double alpha_e_2sP = 4.17e-20 * exp(-0.861 * log(T * 1.e-4));
  const double ch1 = alphaH * nH / jH;
  const double ch2 = AHe * alpha_e_2sP * nH / jH;
  double che = 0.;
  if (jHe > 0.)
    che = alphaHe * nH / jHe;
  double h0old = 0.99 * (1. - exp(-0.5 / ch1));
  h0 = 0.9 * h0old;
  double he0old = 1.;
  if (che > 0.) {
    he0old = 0.5 / che;
    he0old = fmin(he0old, 1.);
  }
  he0 = 0.;
  int niter = 0;
  /* stops as soon as EITHER fraction has converged (&&), as the reference */
  while (fabs(h0 - h0old) > 1.e-4 * h0old &&
         fabs(he0 - he0old) > 1.e-4 * he0old) {
    ++niter;
    h0old = h0;
    he0old = (he0 > 0.) ? he0 : 0.;
    const double pHots = 1. / (1. + 77. * he0old / sqrt(T) / h0old);
    const double ch = ch1 - ch2 * AHe * (1. - he0old) * pHots / (1. - h0old);
    he0 = 1.;
    if (che != 0.) {
      const double bhe = (1. + 2. * AHe - h0) * che + 1.;
      const double che_bhe = che / bhe;
      const double opAHeh0 = 1. + AHe - h0;
      const double t1he = 4. * AHe * opAHeh0 * che_bhe * che_bhe;
      if (t1he < 1.e-3) {
        he0 = opAHeh0 * che_bhe;
      } else {
        he0 = (bhe - sqrt(bhe * bhe - 4. * AHe * opAHeh0 * che * che)) /
              (2. * AHe * che);
      }
    }
    const double b = ch * (2. + AHe - he0 * AHe) + 1.;
    const double ch_b = ch / b;
    const double opAHeh0AHe = 1. + AHe - he0 * AHe;
    const double t1 = 4. * ch_b * ch_b * opAHeh0AHe;
    if (t1 < 1.e-3) {
      h0 = ch_b * opAHeh0AHe;
    } else {
      h0 = (b - sqrt(b * b - 4. * ch * ch * opAHeh0AHe)) / (2. * ch);
    }
    if (niter > 10) {
      h0 = 0.5 * (h0 + h0old);
      he0 = 0.5 * (he0 + he0old);
    }
    if (niter > 20) {
      /* the reference aborts here (cmac_error); the engine keeps the last
       * iterate */
      break;
    }
  }
}

/* src/IonizationStateCalculator.cpp:323-501; x[2..13] out. The balance of
 * ion k against the next stage is always
 *   ratio_k = (j_k + n(H+) CT_ion,k) /
 *             (n_e alpha_k + n(H0) CT_rec,H,k + n(He0) CT_rec,He,k)
 * with the charge transfer terms the reference's formula for that ion has
 * (TablesDev::metal_ct; the others are zero rows): one loop body for the 12
 * ions instead of 12 + 19 inlined fit evaluations, then the reference's
 * products and normalisations per element. */
/* ratio_k of ion ION_C_p1 + k (the loop body below; temp_finish_kernel
 * evaluates the 12 ions of a cell side by side, one lane each) */
template <class Integrals>
__device__ __forceinline__ double
cmi_metal_ratio(const ModelDev &m, const Integrals &j, int ion, double ne,
                double T, double T4, double nh0, double nhe0, double nhp) {
  const TablesDev *tb = m.tables;
  const double alpha = cmi_recombination_rate(m, ion, T);
  const double num = j(ion) + nhp * ct_eval(tb->metal_ct[ion][1], T4);
  const double den = ne * alpha + nh0 * ct_eval(tb->metal_ct[ion][0], T4) +
                     nhe0 * ct_eval(tb->metal_ct[ion][2], T4);
  return num / den;
}

__device__ __forceinline__ void cmi_metal_fractions(const double (&ratio)[12],
                                                    double x[CMI_NION]);

template <class Integrals>
__device__ inline void cmi_ionization_states_metals(
    const ModelDev &m, const Integrals &j, double ne, double T, double T4,
    double nh0, double nhe0, double nhp, double x[CMI_NION]) {
  double ratio[12];
#pragma unroll 1
  for (int k = 0; k < 12; ++k) {
    const double r =
        cmi_metal_ratio(m, j, ION_C_p1 + k, ne, T, T4, nh0, nhe0, nhp);
    /* static indexing keeps ratio[] in registers */
#pragma unroll
    for (int i = 0; i < 12; ++i)
      if (i == k)
        ratio[i] = r;
  }
  cmi_metal_fractions(ratio, x);
}

/* the reference's products and normalisations per element */
__device__ __forceinline__ void cmi_metal_fractions(const double (&ratio)[12],
                                                    double x[CMI_NION]) {
#define R(ion) ratio[(ion)-ION_C_p1]
  { /* carbon */
    const double C21 = R(ION_C_p1);
    const double C31 = R(ION_C_p2) * C21;
    const double s = 1. / (1. + C21 + C31);
    x[ION_C_p1] = C21 * s;
    x[ION_C_p2] = C31 * s;
  }
  { /* nitrogen */
    const double N21 = R(ION_N_n);
    const double N31 = R(ION_N_p1) * N21;
    const double N41 = R(ION_N_p2) * N31;
    const double s = 1. / (1. + N21 + N31 + N41);
    x[ION_N_n] = N21 * s;
    x[ION_N_p1] = N31 * s;
    x[ION_N_p2] = N41 * s;
  }
  { /* oxygen */
    const double O21 = R(ION_O_n);
    const double O31 = R(ION_O_p1) * O21;
    const double s = 1. / (1. + O21 + O31);
    x[ION_O_n] = O21 * s;
    x[ION_O_p1] = O31 * s;
  }
  { /* neon */
    const double Ne21 = R(ION_Ne_n);
    const double Ne31 = R(ION_Ne_p1) * Ne21;
    const double s = 1. / (1. + Ne21 + Ne31);
    x[ION_Ne_n] = Ne21 * s;
    x[ION_Ne_p1] = Ne31 * s;
  }
  { /* sulphur */
    const double S21 = R(ION_S_p1);
    const double S31 = R(ION_S_p2) * S21;
    const double S41 = R(ION_S_p3) * S31;
    const double s = 1. / (1. + S21 + S31 + S41);
    x[ION_S_p1] = S21 * s;
    x[ION_S_p2] = S31 * s;
    x[ION_S_p3] = S41 * s;
  }
#undef R
}

/* IonizationStateCalculator::calculate_ionization_state(jfac, hfac, vars),
 * src/IonizationStateCalculator.cpp:70-272. J[14] un-normalised; heating[2]
 * normalised in place; x[14] out. */
__device__ inline void cmi_ionization_state_cell(const ModelDev &m, double jfac,
                                                 double hfac, double ntot,
                                                 double T,
                                                 const double J[CMI_NION],
                                                 double heating[2],
                                                 double x[CMI_NION]) {
  const double jH = jfac * J[ION_H_n];
  const double jHe = jfac * J[ION_He_n];
  heating[0] = hfac * heating[0];
  heating[1] = hfac * heating[1];
  if (jH > 0. && ntot > 0.) {
    const double alphaH = cmi_recombination_rate(m, ION_H_n, T);
    const double AHe = m.abundance[0];
    double h0, he0 = 0.;
    if (AHe != 0.) {
      const double alphaHe = cmi_recombination_rate(m, ION_He_n, T);
      cmi_ionization_states_hydrogen_helium(alphaH, alphaHe, jH, jHe, ntot,
                                            AHe, T, h0, he0);
    } else {
      h0 = cmi_ionization_state_hydrogen(alphaH, jH, ntot);
    }
    x[ION_H_n] = h0;
    x[ION_He_n] = he0;
    const double nhp = ntot * (1. - h0);
    const double ne = ntot * (1. - h0 + AHe * (1. - he0));
    const double T4 = T * 1.e-4;
    const double nh0 = ntot * h0;
    const double nhe0 = ntot * he0 * AHe;
    /* normalised mean intensity of an ion, by ion index */
    struct {
      const double *J;
      double jfac;
      __device__ __forceinline__ double operator()(int ion) const {
        double v = 0.; /* static indexing: J[] lives in registers */
#pragma unroll
        for (int i = ION_C_p1; i < CMI_NION; ++i)
          v = (i == ion) ? J[i] : v;
        return jfac * v;
      }
    } jm = {J, jfac};
    cmi_ionization_states_metals(m, jm, ne, T, T4, nh0, nhe0, nhp, x);
  } else {
#pragma unroll
    for (int i = 0; i < CMI_NION; ++i)
      x[i] = 0.;
    if (ntot > 0.) {
      /* neutral gas: the tracked neutral stages are fully populated */
      x[ION_H_n] = 1.;
      x[ION_He_n] = 1.;
      x[ION_N_n] = 1.;
      x[ION_O_n] = 1.;
      x[ION_Ne_n] = 1.;
    }
  }
}

#endif
