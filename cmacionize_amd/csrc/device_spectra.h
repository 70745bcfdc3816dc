/*
 * device_spectra.h - device samplers for source spectra and diffuse
 * re-emission.
 */
#ifndef CMI_DEVICE_SPECTRA_H
#define CMI_DEVICE_SPECTRA_H

#include "device_physics.h"

/* PhotonSourceSpectrum::get_random_frequency */
__device__ inline double sample_source_spectrum(const ModelDev &m,
                                                PacketRng &rng) {
  (void)rng;
  /* MonochromaticPhotonSourceSpectrum: no random number is drawn
   * (src/MonochromaticPhotonSourceSpectrum.hpp:97-100) */
  return m.mono_frequency;
}

#endif
