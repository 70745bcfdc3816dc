/*
 * Units.hpp - unit strings of the parameter files ("10. pc", "100. cm^-3",
 * "13.6 eV", "4.e-13 cm^3 s^-1") to SI values.
 *
 * Host-side mirror of the reference's Unit / UnitConverter
 * (src/Unit.hpp:124-150, src/UnitConverter.hpp:97-160,266-300,345-440): the
 * same unit table and the same arithmetic (integer powers by repeated
 * multiplication / division, value * unit), so that a .param file gives the
 * same doubles as in the reference.
 */
#ifndef CMI_HOST_UNITS_HPP
#define CMI_HOST_UNITS_HPP

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>

namespace cmi {

struct ParameterError : std::runtime_error {
  explicit ParameterError(const std::string &what)
      : std::runtime_error(what) {}
};

enum Quantity {
  QUANTITY_FREQUENCY,
  QUANTITY_LENGTH,
  QUANTITY_NUMBER_DENSITY,
  QUANTITY_REACTION_RATE,
  QUANTITY_SURFACE_AREA,
  QUANTITY_TEMPERATURE,
  QUANTITY_FLUX,
  QUANTITY_TIME,
  QUANTITY_DENSITY,
  QUANTITY_VELOCITY,
  QUANTITY_ANGLE,
  QUANTITY_MASS
};

/* value + exponents of (length, time, mass, temperature, current, angle) */
struct Unit {
  double value;
  int dim[6];

  Unit(double v, int l, int t, int m, int T, int c, int a)
      : value(v), dim{l, t, m, T, c, a} {}

  Unit &operator*=(const Unit &o) {
    value *= o.value;
    for (int i = 0; i < 6; ++i)
      dim[i] += o.dim[i];
    return *this;
  }
  /* src/Unit.hpp:124-150 */
  Unit &operator^=(int power) {
    if (power >= 0) {
      const double base = value;
      for (int i = 1; i < power; ++i)
        value *= base;
    } else {
      const double base = value;
      value = 1.;
      for (int i = 0; i < -power; ++i)
        value /= base;
    }
    for (int i = 0; i < 6; ++i)
      dim[i] *= power;
    return *this;
  }
  bool same_quantity(const Unit &o) const {
    for (int i = 0; i < 6; ++i)
      if (dim[i] != o.dim[i])
        return false;
    return true;
  }
};

namespace constants {
/* src/PhysicalConstants.hpp:61-131 */
constexpr double planck = 6.626070040e-34;
constexpr double boltzmann = 1.38064852e-23;
constexpr double lightspeed = 299792458.;
constexpr double electronvolt = 1.6021766208e-19;
constexpr double proton_mass = 1.672621898e-27;
} // namespace constants

/* src/UnitConverter.hpp:97-160 */
inline Unit single_unit(const std::string &name) {
  if (name == "m")
    return Unit(1., 1, 0, 0, 0, 0, 0);
  if (name == "cm")
    return Unit(0.01, 1, 0, 0, 0, 0, 0);
  if (name == "pc")
    return Unit(3.086e16, 1, 0, 0, 0, 0, 0);
  if (name == "kpc")
    return Unit(3.086e19, 1, 0, 0, 0, 0, 0);
  if (name == "angstrom")
    return Unit(1.e-10, 1, 0, 0, 0, 0, 0);
  if (name == "km")
    return Unit(1000., 1, 0, 0, 0, 0, 0);
  if (name == "au")
    return Unit(149597870700., 1, 0, 0, 0, 0, 0);
  if (name == "s")
    return Unit(1., 0, 1, 0, 0, 0, 0);
  if (name == "Gyr")
    return Unit(3.154e16, 0, 1, 0, 0, 0, 0);
  if (name == "Myr")
    return Unit(3.154e13, 0, 1, 0, 0, 0, 0);
  if (name == "yr")
    return Unit(3.154e7, 0, 1, 0, 0, 0, 0);
  if (name == "h")
    return Unit(3600., 0, 1, 0, 0, 0, 0);
  if (name == "kg")
    return Unit(1., 0, 0, 1, 0, 0, 0);
  if (name == "g")
    return Unit(0.001, 0, 0, 1, 0, 0, 0);
  if (name == "Msol")
    return Unit(1.98855e30, 0, 0, 1, 0, 0, 0);
  if (name == "K")
    return Unit(1., 0, 0, 0, 1, 0, 0);
  if (name == "radians")
    return Unit(1., 0, 0, 0, 0, 0, 1);
  if (name == "degrees")
    return Unit(M_PI / 180., 0, 0, 0, 0, 0, 1);
  if (name == "Hz")
    return Unit(1., 0, -1, 0, 0, 0, 0);
  if (name == "J")
    return Unit(1., 2, -2, 1, 0, 0, 0);
  if (name == "erg")
    return Unit(1.e-7, 2, -2, 1, 0, 0, 0);
  if (name == "eV")
    return Unit(constants::electronvolt, 2, -2, 1, 0, 0, 0);
  if (name == "Pa")
    return Unit(1., -1, -2, 1, 0, 0, 0);
  if (name == "bar")
    return Unit(1.e5, -1, -2, 1, 0, 0, 0);
  throw ParameterError("Unknown unit: \"" + name + "\"!");
}

/* "cm^3 s^-1" -> composite unit, src/UnitConverter.hpp:345-432 */
inline Unit parse_unit(const std::string &name) {
  Unit result(1., 0, 0, 0, 0, 0, 0);
  bool any = false;
  size_t pos = 0;
  while (pos < name.size()) {
    while (pos < name.size() && !isalpha((unsigned char)name[pos]))
      ++pos;
    if (pos == name.size())
      break;
    size_t end = pos + 1;
    while (end < name.size() && name[end] != ' ' && name[end] != '^')
      ++end;
    Unit u = single_unit(name.substr(pos, end - pos));
    if (end < name.size() && name[end] == '^') {
      size_t p1 = end + 1;
      size_t p2 = p1 + 1;
      while (p2 < name.size() && (isdigit((unsigned char)name[p2]) ||
                                  name[p2] == '+' || name[p2] == '-'))
        ++p2;
      u ^= std::stoi(name.substr(p1, p2 - p1));
      end = p2;
    }
    if (!any) {
      result = u; /* the first unit is taken as is, later ones multiply */
      any = true;
    } else {
      result *= u;
    }
    pos = end;
  }
  if (!any)
    throw ParameterError("Empty unit provided!");
  return result;
}

inline const char *SI_unit_name(Quantity q) {
  switch (q) {
  case QUANTITY_FREQUENCY:
    return "Hz";
  case QUANTITY_LENGTH:
    return "m";
  case QUANTITY_NUMBER_DENSITY:
    return "m^-3";
  case QUANTITY_REACTION_RATE:
    return "m^3 s^-1";
  case QUANTITY_SURFACE_AREA:
    return "m^2";
  case QUANTITY_TEMPERATURE:
    return "K";
  case QUANTITY_FLUX:
    return "m^-2 s^-1";
  case QUANTITY_TIME:
    return "s";
  case QUANTITY_DENSITY:
    return "kg m^-3";
  case QUANTITY_VELOCITY:
    return "m s^-1";
  case QUANTITY_ANGLE:
    return "radians";
  case QUANTITY_MASS:
    return "kg";
  }
  return "";
}

/* UnitConverter::to_SI, src/UnitConverter.hpp:434-443, with the two
 * cross-quantity conversions of try_conversion (:266-300): photon energy and
 * photon wavelength to frequency */
inline double to_SI(Quantity q, double value, const std::string &unit) {
  const Unit SI = parse_unit(SI_unit_name(q));
  const Unit u = parse_unit(unit);
  if (SI.same_quantity(u))
    return value * u.value;
  if (q == QUANTITY_FREQUENCY) {
    const Unit energy = parse_unit("J");
    const Unit length = parse_unit("m");
    if (u.same_quantity(energy)) {
      const double Sval = value * u.value;
      return Sval * (1. / constants::planck) / SI.value;
    }
    if (u.same_quantity(length)) {
      const double Sfac = value * u.value;
      double Sval = 1.;
      Sval /= Sfac;
      return Sval * constants::lightspeed / SI.value;
    }
  }
  throw ParameterError("No known conversion from \"" + unit + "\" to \"" +
                       SI_unit_name(q) + "\"!");
}

/* Utilities::split_value, src/Utilities.hpp:690-711: "10. pc" -> (10, "pc") */
inline std::pair<double, std::string> split_value(const std::string &s) {
  size_t idx = 0;
  double value;
  try {
    value = std::stod(s, &idx);
  } catch (std::exception &) {
    throw ParameterError("Error extracting value from \"" + s +
                         "\" unit-value pair!");
  }
  while (idx < s.size() && s[idx] == ' ')
    ++idx;
  return std::make_pair(value, s.substr(idx));
}

} // namespace cmi

#endif
