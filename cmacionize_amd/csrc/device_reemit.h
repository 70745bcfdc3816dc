/*
 * device_reemit.h - diffuse re-emission of absorbed packets.
 */
#ifndef CMI_DEVICE_REEMIT_H
#define CMI_DEVICE_REEMIT_H

#include "device_transport.h"

/* PhysicalDiffuseReemissionHandler::set_reemission_probabilities,
 * src/PhysicalDiffuseReemissionHandler.hpp:66-105. The reference stores the
 * five numbers per cell at the start of every iteration
 * (src/IonizationSimulation.cpp:380-383); they only depend on the cell's
 * temperature, which does not change while packets fly, so they are evaluated
 * where needed instead of costing 5 fields of HBM. */
__device__ inline void reemission_probabilities(double temperature,
                                                double &pH, double pHe[4]) {
  /* (the powers T4^y as exp(y ln T4) with ONE logarithm: a few 1e-16 from
   * pow, a fifth of its instructions) */
  const double lnT4 = log(temperature * 1.e-4);
  const double alpha_1_H = 1.58e-13 * exp(-0.53 * lnT4);
  const double alpha_A_agn = 4.18e-13 * exp(-0.7 * lnT4);
  pH = alpha_1_H / alpha_A_agn;
  const double alpha_1_He = 1.54e-13 * exp(-0.486 * lnT4);
  const double alpha_e_2tS = 2.1e-13 * exp(-0.381 * lnT4);
  const double alpha_e_2sS = 2.06e-14 * exp(-0.451 * lnT4);
  const double alpha_e_2sP = 4.17e-14 * exp(-0.695 * lnT4);
  const double alphaHe = alpha_1_He + alpha_e_2tS + alpha_e_2sS + alpha_e_2sP;
  pHe[0] = alpha_1_He / alphaHe;
  pHe[1] = pHe[0] + alpha_e_2tS / alphaHe;
  pHe[2] = pHe[1] + alpha_e_2sS / alphaHe;
  pHe[3] = pHe[2] + alpha_e_2sP / alphaHe;
}

/* What becomes of an absorbed packet: the spectrum its new frequency is drawn
 * from (the last draw of every branch of the handlers' reemit) */
#define CMI_REEMIT_ABSORBED 0   /* absorbed for good */
#define CMI_REEMIT_LYC_H 1      /* hydrogen Lyman continuum at the cell's T */
#define CMI_REEMIT_LYC_HE 2     /* helium Lyman continuum at the cell's T */
#define CMI_REEMIT_HE_19_8 3    /* He 2^3S -> 1^1S, 19.8 eV: no draw */
#define CMI_REEMIT_HE_2PHOTON 4 /* helium two-photon continuum */
#define CMI_REEMIT_FIXED 5      /* FixedValue handler's frequency: no draw */

/* PhysicalDiffuseReemissionHandler::reemit,
 * src/PhysicalDiffuseReemissionHandler.cpp:219-370 (Wood, Mathis & Ercolano
 * 2004, section 3.3), up to the choice of the spectrum: CMI_REEMIT_* */
__device__ inline int32_t physical_reemit_kind(const ModelDev &m,
                                               double sigma_H, double sigma_He,
                                               double T, double xH, double xHe,
                                               PacketRng &rng) {
  const double AHe = m.abundance[0];
  const double nH0anuH0 = xH * sigma_H;
  const double nHe0anuHe0 = xHe * AHe * sigma_He;
  const double pHabs = nH0anuH0 / (nH0anuH0 + nHe0anuHe0);

  double x = rng.next();
  if (x <= pHabs) {
    /* absorbed by hydrogen: Lyman continuum photon or lost */
    const double lnT4 = log(T * 1.e-4);
    const double pH = (1.58e-13 * exp(-0.53 * lnT4)) /
                      (4.18e-13 * exp(-0.7 * lnT4));
    x = rng.next();
    return (x <= pH) ? CMI_REEMIT_LYC_H : CMI_REEMIT_ABSORBED;
  }
  /* absorbed by helium: pick the recombination channel */
  double pH, pHe[4];
  reemission_probabilities(T, pH, pHe);
  x = rng.next();
  if (x <= pHe[0])
    return CMI_REEMIT_LYC_HE;
  if (x <= pHe[1])
    return CMI_REEMIT_HE_19_8;
  if (x <= pHe[2]) {
    /* two-photon continuum: 56 % chance of an H-ionizing photon */
    x = rng.next();
    return (x < 0.56) ? CMI_REEMIT_HE_2PHOTON : CMI_REEMIT_ABSORBED;
  }
  if (x <= pHe[3]) {
    /* He Lyman alpha: absorbed on the spot by H, or two-photon decay */
    const double sqrtTnH0 = sqrt(T) * xH;
    const double pHots = sqrtTnH0 / (sqrtTnH0 + 77. * xHe);
    x = rng.next();
    if (x < pHots) {
      x = rng.next();
      return (x <= pH) ? CMI_REEMIT_LYC_H : CMI_REEMIT_ABSORBED;
    }
    x = rng.next();
    return (x < 0.56) ? CMI_REEMIT_HE_2PHOTON : CMI_REEMIT_ABSORBED;
  }
  return CMI_REEMIT_ABSORBED;
}

/* ... and the frequency from that spectrum, with the photon type of the
 * re-emitted packet; 0 = absorbed for good */
__device__ inline double sample_reemission(const ModelDev &m, int32_t kind,
                                           double T, PacketRng &rng,
                                           int32_t &type) {
  type = TYPE_DIFFUSE_HeI;
  switch (kind) {
  case CMI_REEMIT_LYC_H:
    type = TYPE_DIFFUSE_HI;
    return sample_lyman_continuum(m.spectra, 0, T, rng);
  case CMI_REEMIT_LYC_HE:
    return sample_lyman_continuum(m.spectra, 1, T, rng);
  case CMI_REEMIT_HE_19_8:
    return 4.788e15;
  case CMI_REEMIT_HE_2PHOTON:
    return sample_he_two_photon(m.spectra, rng);
  case CMI_REEMIT_FIXED:
    type = TYPE_DIFFUSE_HI;
    return m.reemit_fixed_frequency;
  default:
    type = TYPE_ABSORBED;
    return 0.;
  }
}

/* both halves: returns the new frequency, 0 = absorbed for good */
__device__ inline double physical_reemit(const ModelDev &m, double sigma_H,
                                         double sigma_He, double T, double xH,
                                         double xHe, PacketRng &rng,
                                         int32_t &type) {
  const int32_t kind =
      physical_reemit_kind(m, sigma_H, sigma_He, T, xH, xHe, rng);
  return sample_reemission(m, kind, T, rng, type);
}

/* First half of PhotonSource::reemit (src/PhotonSource.cpp:272-308): the
 * handler's decision. Returns the new frequency (0 = the packet ends here) and
 * sets p.type. On re-emission the packet's position is materialised: it is
 * where the next flight starts. */
template <bool FULL, bool EXACT>
__device__ inline double reemit_decide(const ModelDev &m, const CellsDev &cells,
                                       int64_t cell, PacketRng &rng,
                                       Packet<FULL> &p) {
  double new_frequency;
  int32_t type;
  if (m.reemit_type == 2) {
    /* FixedValueDiffuseReemissionHandler::reemit,
     * src/FixedValueDiffuseReemissionHandler.hpp:73-86 */
    const double u = rng.next();
    if (u < m.reemit_fixed_probability) {
      type = TYPE_DIFFUSE_HI;
      new_frequency = m.reemit_fixed_frequency;
    } else {
      type = TYPE_ABSORBED;
      new_frequency = 0.;
    }
  } else {
    new_frequency = physical_reemit(
        m, p.sigma_H, p.sigma_He, cells.temperature[cell],
        cells.x[ION_H_n][cell], cells.x[ION_He_n][cell], rng, type);
  }
  p.type = type;
  if (new_frequency != 0. && !EXACT)
    end_flight(p); /* position of the absorption, along the OLD direction */
  return new_frequency;
}

/* Second half of PhotonSource::reemit + IonizationPhotonShootJob::execute
 * (src/PhotonSource.cpp:296-303, src/IonizationPhotonShootJob.hpp:139-141):
 * new isotropic direction, cross sections at the new frequency, new optical
 * depth; the flight starts from the cell that contains p.pos. */
template <bool FULL, bool EXACT>
__device__ inline void reemit_launch(const GridDev &g, const ModelDev &m,
                                     double new_frequency, PacketRng &rng,
                                     Packet<FULL> &p,
                                     double (&weights)[CMI_NACC]) {
  p.nu = new_frequency;
  random_direction(p, rng);
  set_cross_sections(m, p, weights);
  p.tau = -log(rng.next());
  start_flight<FULL, EXACT>(g, p);
}

#endif
