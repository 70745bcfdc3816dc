/*
 * third_party_plugins.cpp - plugins as a third party would write them against
 * the REFERENCE's interfaces: each class implements only the reference's
 * virtuals (src/PhotonSourceSpectrum.hpp:48-56, src/CrossSections.hpp:49-50,
 * src/RecombinationRates.hpp:49) and knows nothing of lower(), of tables or of
 * the device. They reach the GPU engine through the generic lowering of the
 * base classes in cmacionize_amd/host/Plugins.hpp.
 *
 * Built two ways by tests/test_generic_lowering.py:
 *   -shared            a library whose extern "C" functions lower the plugins
 *                      into an engine the test created, and hand the test the
 *                      tables the lowering made (so that the oracle can be
 *                      given the same ones);
 *   -DTP_WITH_MAIN     cmi-gpu with the three plugins registered under the
 *                      type names "ThirdPartyPowerLaw" / "ThirdPartyFalling".
 */
#include "../../cmacionize_amd/host/Plugins.hpp"

#include <cmath>

namespace {

const double NU_H = 3.288465385e15; /* 13.6 eV */

/* photons per unit frequency ~ nu^-2 between 13.6 and 54.4 eV, sampled by
 * inverting the cumulative distribution - with 1 - u, so the frequency FALLS
 * with the uniform */
class FallingSpectrum : public cmi::PhotonSourceSpectrum {
public:
  double get_random_frequency(cmi::RandomGenerator &random_generator,
                              double = 0.) const override {
    const double u = 1. - random_generator.get_uniform_random_double();
    return NU_H / (1. - 0.75 * u);
  }
  double get_total_flux() const override { return 1.e12; }
};

/* a rejection sampler (two uniforms per try): a triangular spectrum */
class RejectionSpectrum : public cmi::PhotonSourceSpectrum {
public:
  double get_random_frequency(cmi::RandomGenerator &random_generator,
                              double = 0.) const override {
    for (;;) {
      const double x = random_generator.get_uniform_random_double();
      const double y = random_generator.get_uniform_random_double();
      if (y <= 1. - x)
        return NU_H * (1. + 3. * x);
    }
  }
  double get_total_flux() const override { return 1.e12; }
};

/* hydrogen-like cross sections: sigma_0 (nu / nu_th)^-3 above a threshold */
class PowerLawCrossSections : public cmi::CrossSections {
public:
  double get_cross_section(const int_fast32_t ion,
                           const double energy) const override {
    static const double threshold_eV[cmi::NUMBER_OF_IONNAMES] = {
        13.6, 24.6, 0., 0., 14.5, 0., 0., 13.62, 0., 0., 0., 0., 0., 0.};
    static const double sigma_0[cmi::NUMBER_OF_IONNAMES] = {
        6.3e-22, 7.8e-22, 0., 0., 1.1e-21, 0., 0., 3.e-22,
        0.,      0.,      0., 0., 0.,      0.};
    if (sigma_0[ion] == 0.)
      return 0.;
    const double nu_th = threshold_eV[ion] * (NU_H / 13.6);
    if (energy < nu_th)
      return 0.;
    const double x = nu_th / energy;
    return sigma_0[ion] * x * x * x;
  }
};

/* power laws in the temperature */
class PowerLawRecombinationRates : public cmi::RecombinationRates {
public:
  double get_recombination_rate(const int_fast32_t ion,
                                const double temperature) const override {
    const double alpha_4 = ion == cmi::ION_H_n
                               ? 4.e-19
                               : (ion == cmi::ION_He_n ? 4.3e-19 : 1.e-18);
    const double slope =
        ion == cmi::ION_H_n ? -0.7 : (ion == cmi::ION_He_n ? -0.67 : -0.6);
    return alpha_4 * std::pow(temperature * 1.e-4, slope);
  }
};

/* what a third party adds to its own start-up code */
struct Registration {
  Registration() {
    cmi::register_photon_source_spectrum(
        "ThirdPartyFalling",
        [](const std::string &, cmi::ParameterFile &) {
          return (cmi::PhotonSourceSpectrum *)new FallingSpectrum();
        });
    cmi::register_cross_sections("ThirdPartyPowerLaw", [](cmi::ParameterFile &) {
      return (cmi::CrossSections *)new PowerLawCrossSections();
    });
    cmi::register_recombination_rates(
        "ThirdPartyPowerLaw", [](cmi::ParameterFile &) {
          return (cmi::RecombinationRates *)new PowerLawRecombinationRates();
        });
  }
} registration;

} // namespace

#ifdef TP_WITH_MAIN
#include "../../cmacionize_amd/host/cmi_gpu_main.cpp"
#else
namespace {
cmi::PhotonSourceSpectrum *make_spectrum(int which) {
  if (which == 0)
    return new FallingSpectrum();
  if (which == 1)
    return new RejectionSpectrum();
  return new cmi::UniformPhotonSourceSpectrum();
}
} // namespace

extern "C" {

/* lower the three plugins into `engine` through their base classes' generic
 * lower(); spectrum: 0 falling power law, 1 rejection sampler, 2 the
 * reference's Uniform */
int tp_lower(cmi_gpu_engine *engine, int spectrum, int continuous) {
  std::unique_ptr<cmi::PhotonSourceSpectrum> s(make_spectrum(spectrum));
  int rc = continuous ? s->lower_continuous(engine) : s->lower(engine);
  if (rc)
    return rc;
  rc = PowerLawCrossSections().lower(engine);
  if (rc)
    return rc;
  return PowerLawRecombinationRates().lower(engine);
}

/* the tables the generic lowering makes of them (for the oracle) */
int tp_spectrum_table(int spectrum, int capacity, double *frequency,
                      double *cumulative, int *interpolation, char *method) {
  std::unique_ptr<cmi::PhotonSourceSpectrum> s(make_spectrum(spectrum));
  const cmi::SpectrumTable t = cmi::tabulate_spectrum(*s);
  const int n = (int)t.frequency.size();
  if (n > capacity)
    return -n;
  for (int k = 0; k < n; ++k) {
    frequency[k] = t.frequency[k];
    cumulative[k] = t.cumulative[k];
  }
  *interpolation = t.interpolation;
  *method = t.method[0];
  return n;
}

int tp_ion_table(int which, int capacity, double *x, double *y,
                 int *interpolation) {
  const cmi::IonTable t = which == 0 ? PowerLawCrossSections().tabulate()
                                     : PowerLawRecombinationRates().tabulate();
  const int n = (int)t.x.size();
  if (n > capacity)
    return -n;
  for (int k = 0; k < n; ++k)
    x[k] = t.x[k];
  for (size_t k = 0; k < t.y.size(); ++k)
    y[k] = t.y[k];
  *interpolation = t.interpolation;
  return n;
}

/* the virtuals themselves (the test checks the tables against them) */
double tp_cross_section(int ion, double frequency) {
  return PowerLawCrossSections().get_cross_section(ion, frequency);
}
double tp_recombination_rate(int ion, double temperature) {
  return PowerLawRecombinationRates().get_recombination_rate(ion, temperature);
}
}
#endif
