/*
 * Plugins.hpp - the host-side plugin surfaces of the path, with the
 * reference's virtual signatures, the concrete plugins the benchmark configs
 * select, and their factories keyed on "<Block>:type".
 *
 * In the reference these objects are called per packet / per cell through
 * virtual functions. Here they run on the host once, at initialisation, and
 * are LOWERED into the flat descriptors of the C ABI (include/cmi_gpu.h):
 * every concrete class below has a lower() that makes the matching
 * cmi_gpu_set_* call. A plugin that implements ONLY the reference's virtuals
 * - a third-party class recompiled against these headers - inherits the
 * generic lower() of its base class: the virtual is sampled on the host into a
 * table (tabulate_spectrum / tabulate_cross_sections /
 * tabulate_recombination_rates) that goes to the device through
 * cmi_gpu_set_spectrum_table / cmi_gpu_set_cross_sections_table /
 * cmi_gpu_set_recombination_rates_table, and is made known to the factories
 * with register_photon_source_spectrum() / register_cross_sections() /
 * register_recombination_rates().
 *
 * Reference interfaces mirrored (signatures kept):
 *   Cell                         src/Cell.hpp
 *   DensityValues                src/DensityValues.hpp:36-120
 *   DensityFunction              src/DensityFunction.hpp:35-65
 *   PhotonSourceDistribution     src/PhotonSourceDistribution.hpp:54-80
 *   PhotonSourceSpectrum         src/PhotonSourceSpectrum.hpp:48-56
 *   CrossSections                src/CrossSections.hpp:49-50
 *   RecombinationRates           src/RecombinationRates.hpp:49
 *   AbundanceModel / Abundances  src/FixedValueAbundanceModel.hpp:44-66
 *   DiffuseReemissionHandler     src/DiffuseReemissionHandlerFactory.hpp:94-99
 *   SimulationBox                src/SimulationBox.hpp
 */
#ifndef CMI_HOST_PLUGINS_HPP
#define CMI_HOST_PLUGINS_HPP

#include "../../include/cmi_gpu.h"
#include "Hdf5Reader.hpp"
#include "ParameterFile.hpp"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace cmi {

constexpr int NUMBER_OF_IONNAMES = 14;
/* src/ElementNames.hpp:101-154 */
enum IonName {
  ION_H_n = 0, ION_He_n, ION_C_p1, ION_C_p2, ION_N_n, ION_N_p1, ION_N_p2,
  ION_O_n, ION_O_p1, ION_Ne_n, ION_Ne_p1, ION_S_p1, ION_S_p2, ION_S_p3
};
/* parameter names of the ions (src/ElementNames.hpp get_ion_name) */
inline const char *ion_name(int ion) {
  static const char *names[NUMBER_OF_IONNAMES] = {
      "H", "He", "C+", "C++", "N", "N+", "N++", "O", "O+", "Ne", "Ne+", "S+",
      "S++", "S+++"};
  return names[ion];
}

struct CoordinateVector {
  double v[3];
  CoordinateVector() : v{0., 0., 0.} {}
  CoordinateVector(double x, double y, double z) : v{x, y, z} {}
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
  double operator[](int i) const { return v[i]; }
  double &operator[](int i) { return v[i]; }
};

/* src/Face.hpp:37-127: a planar face of a cell, its vertices in order round
 * the face */
struct Face {
  CoordinateVector midpoint;
  std::vector<CoordinateVector> vertices;
};

/* src/Cell.hpp:36-57 */
class Cell {
public:
  virtual ~Cell() {}
  virtual CoordinateVector get_cell_midpoint() const = 0;
  virtual double get_volume() const = 0;
  virtual std::vector<Face> get_faces() const { return std::vector<Face>(); }
};

class DensityValues {
  double _number_density = 0.;
  double _ionic_fraction[NUMBER_OF_IONNAMES] = {0.};
  double _temperature = 0.;

public:
  void set_number_density(double n) { _number_density = n; }
  void set_ionic_fraction(int ion, double x) { _ionic_fraction[ion] = x; }
  void set_temperature(double T) { _temperature = T; }
  double get_number_density() const { return _number_density; }
  double get_ionic_fraction(int ion) const { return _ionic_fraction[ion]; }
  double get_temperature() const { return _temperature; }
};

/* ------------------------------------------------------- DensityFunction */

class DensityFunction {
public:
  virtual ~DensityFunction() {}
  virtual void initialize() {}
  virtual void free() {}
  virtual DensityValues operator()(const Cell &cell) = 0;
};

/* src/HomogeneousDensityFunction.hpp:41-107 */
class HomogeneousDensityFunction : public DensityFunction {
  const double _density, _temperature, _neutral_fraction_H;

public:
  HomogeneousDensityFunction(double density = 1., double temperature = 8000.,
                             double neutral_fraction_H = 1.e-6)
      : _density(density), _temperature(temperature),
        _neutral_fraction_H(neutral_fraction_H) {}
  explicit HomogeneousDensityFunction(ParameterFile &params)
      : HomogeneousDensityFunction(
            params.get_physical_value(QUANTITY_NUMBER_DENSITY,
                                      "DensityFunction:density", "100. cm^-3"),
            params.get_physical_value(QUANTITY_TEMPERATURE,
                                      "DensityFunction:temperature",
                                      "8000. K"),
            params.get_double("DensityFunction:neutral fraction H", 1.e-6)) {}
  DensityValues operator()(const Cell &) override {
    DensityValues values;
    values.set_number_density(_density);
    values.set_temperature(_temperature);
    values.set_ionic_fraction(ION_H_n, _neutral_fraction_H);
    values.set_ionic_fraction(ION_He_n, 1.e-6);
    return values;
  }
};

/* src/BlockSyntaxDensityFunction.hpp:45-195, src/BlockSyntaxBlock.hpp:91-105 */
class BlockSyntaxDensityFunction : public DensityFunction {
  struct Block {
    std::array<double, 3> origin, sides;
    double exponent, density, temperature, neutral_fraction_H;
    bool is_inside(const CoordinateVector &p) const {
      double r = 0.;
      for (int i = 0; i < 3; ++i) {
        const double x = 2. * std::abs(p[i] - origin[i]) / sides[i];
        if (exponent < 10.)
          r += std::pow(x, exponent);
        else
          r = std::max(r, x);
      }
      if (exponent < 10.)
        r = std::pow(r, 1. / exponent);
      return r <= 1.;
    }
  };
  std::vector<Block> _blocks;

  static double get_exponent(const std::string &type) {
    if (type == "rhombus")
      return 1.;
    if (type == "sphere")
      return 2.;
    if (type == "cube")
      return 10.;
    throw ParameterError("Unknown block type: \"" + type + "\"!");
  }

public:
  explicit BlockSyntaxDensityFunction(const std::string &filename) {
    ParameterFile blockfile(filename);
    const long long numblock = blockfile.get_integer("number of blocks", -1);
    if (numblock < 0)
      throw ParameterError("Parameter \"number of blocks\" not found!");
    for (long long i = 0; i < numblock; ++i) {
      const std::string b = "block[" + std::to_string(i) + "]:";
      Block block;
      block.origin = blockfile.get_physical_vector(
          QUANTITY_LENGTH, b + "origin", "[0. m, 0. m, 0. m]");
      block.sides = blockfile.get_physical_vector(QUANTITY_LENGTH, b + "sides",
                                                  "[1. m, 1. m, 1. m]");
      block.exponent = get_exponent(blockfile.get_string(b + "type", "cube"));
      if (blockfile.has_value(b + "number density")) {
        block.density = blockfile.get_physical_value(
            QUANTITY_NUMBER_DENSITY, b + "number density", "0. m^-3");
      } else {
        block.density = blockfile.get_physical_value(
                            QUANTITY_DENSITY, b + "density", "0. kg m^-3") /
                        constants::proton_mass;
      }
      block.temperature = blockfile.get_physical_value(
          QUANTITY_TEMPERATURE, b + "initial temperature", "0. K");
      block.neutral_fraction_H =
          blockfile.get_double(b + "neutral fraction H", 1.e-6);
      if (block.density < 0.)
        throw ParameterError("Negative density given for block " +
                             std::to_string(i) + "!");
      if (block.temperature < 0.)
        throw ParameterError("Negative temperature given for block " +
                             std::to_string(i) + "!");
      _blocks.push_back(block);
    }
    std::ofstream ofile(filename + ".used-values");
    blockfile.print_contents(ofile);
  }
  explicit BlockSyntaxDensityFunction(ParameterFile &params)
      : BlockSyntaxDensityFunction(
            params.get_filename("DensityFunction:filename")) {}

  DensityValues operator()(const Cell &cell) override {
    const CoordinateVector position = cell.get_cell_midpoint();
    double density = -1., temperature = -1., neutral_fraction_H = -1.;
    for (const Block &b : _blocks) {
      if (b.is_inside(position)) {
        density = b.density;
        temperature = b.temperature;
        neutral_fraction_H = b.neutral_fraction_H;
      }
    }
    if (density < 0. || temperature < 0. || neutral_fraction_H < 0.)
      throw ParameterError("No block found containing a cell midpoint!");
    DensityValues values;
    values.set_number_density(density);
    values.set_temperature(temperature);
    values.set_ionic_fraction(ION_H_n, neutral_fraction_H);
    values.set_ionic_fraction(ION_He_n, 1.e-6);
    return values;
  }
};

/* src/DensityFunctionFactory.hpp:138-172 (types on this path) */
/* src/CMacIonizeSnapshotDensityFunction.cpp:42-549: the density field of a
 * snapshot this code (or the reference) wrote - restart from a snapshot, or
 * start a run on another resolution from it. Snapshots of Cartesian grids
 * only (the grid of this path); the cells of the snapshot are found by their
 * midpoints (:298-311), a cell of the new grid takes the values of the
 * snapshot cell its midpoint lies in (:478-496). Read with the dependency-free
 * Hdf5Reader. */
class CMacIonizeSnapshotDensityFunction : public DensityFunction {
  const std::string _filename;
  const bool _use_density, _use_pressure;
  const double _initial_neutral_fraction;
  std::array<double, 3> _anchor, _sides;
  std::array<long long, 3> _ncell;
  std::vector<double> _number_density, _temperature;
  std::array<std::vector<double>, NUMBER_OF_IONNAMES> _ionic_fraction;

public:
  CMacIonizeSnapshotDensityFunction(const std::string &filename,
                                    bool use_density, bool use_pressure,
                                    double initial_neutral_fraction)
      : _filename(filename), _use_density(use_density),
        _use_pressure(use_pressure),
        _initial_neutral_fraction(initial_neutral_fraction) {
    std::ifstream file(filename);
    if (!file.is_open())
      throw ParameterError("Could not open file \"" + filename + "\"!");
  }
  explicit CMacIonizeSnapshotDensityFunction(ParameterFile &params)
      : CMacIonizeSnapshotDensityFunction(
            params.get_filename("DensityFunction:filename"),
            params.get_bool("DensityFunction:use density", false),
            params.get_bool("DensityFunction:use pressure", false),
            params.get_double("DensityFunction:initial neutral fraction",
                              1.e-6)) {}

  /* :118-436 */
  void initialize() override {
    Hdf5Reader file(_filename);
    ParameterFile parameters;
    for (const auto &kv : file.open("/Parameters").attributes)
      parameters.add_value(kv.first, Hdf5Reader::as_string(kv.second));
    _anchor = parameters.get_physical_vector(QUANTITY_LENGTH,
                                             "SimulationBox:anchor", "");
    _sides = parameters.get_physical_vector(QUANTITY_LENGTH,
                                            "SimulationBox:sides", "");
    _ncell = parameters.get_integer_vector("DensityGrid:number of cells",
                                           {-1, -1, -1});
    const std::string type =
        parameters.get_string("DensityGrid:type", "TaskBased");
    if (type != "Cartesian")
      throw ParameterError("Reconstructing a density field from a " + type +
                           "DensityGrid snapshot is not on this path "
                           "(Cartesian only)");
    /* :155-176 */
    double unit_length_in_SI = 1., unit_density_in_SI = 1.,
           unit_temperature_in_SI = 1.;
    if (file.exists("/Units")) {
      const Hdf5Reader::Object units = file.open("/Units");
      auto number = [&](const char *name) {
        const auto it = units.attributes.find(name);
        if (it == units.attributes.end())
          throw ParameterError(std::string("snapshot without \"") + name +
                               "\"");
        return Hdf5Reader::as_doubles(it->second).at(0);
      };
      unit_length_in_SI = 0.01 * number("Unit length in cgs (U_L)");
      unit_density_in_SI =
          1. / unit_length_in_SI / unit_length_in_SI / unit_length_in_SI;
      unit_temperature_in_SI = number("Unit temperature in cgs (U_T)");
    }
    const std::vector<double> midpoints =
        file.read_doubles("/PartType0/Coordinates");
    std::vector<double> densities;
    if (file.exists("/PartType0/NumberDensity") && !_use_density) {
      densities = file.read_doubles("/PartType0/NumberDensity");
    } else {
      densities = file.read_doubles("/PartType0/Density");
      unit_density_in_SI /= constants::proton_mass;
    }
    const size_t n = densities.size();
    if (midpoints.size() != 3 * n)
      throw ParameterError("snapshot with " + std::to_string(n) +
                           " densities and " +
                           std::to_string(midpoints.size() / 3) +
                           " coordinates");
    /* :213-224 */
    std::array<std::vector<double>, NUMBER_OF_IONNAMES> fractions;
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion) {
      const std::string name =
          std::string("/PartType0/NeutralFraction") + ion_name(ion);
      if (file.exists(name))
        fractions[ion] = file.read_doubles(name);
      else
        fractions[ion].assign(n, _initial_neutral_fraction);
      if (fractions[ion].size() != n)
        throw ParameterError("snapshot dataset " + name + " has the wrong size");
    }
    /* :230-243 */
    std::vector<double> temperatures;
    if (file.exists("/PartType0/Temperature") && !_use_pressure) {
      temperatures = file.read_doubles("/PartType0/Temperature");
    } else {
      temperatures = file.read_doubles("/PartType0/Pressure");
      for (size_t i = 0; i < temperatures.size(); ++i) {
        const double mu = 0.5 * (1. + fractions[ION_H_n][i]);
        temperatures[i] *= mu / (densities[i] * unit_density_in_SI *
                                 constants::boltzmann * unit_temperature_in_SI);
      }
    }
    if (temperatures.size() != n)
      throw ParameterError("snapshot temperatures have the wrong size");
    /* :298-330: every cell of the snapshot's grid from the cell that has its
     * midpoint in it (the box anchor is the origin in the file) */
    const size_t total = (size_t)_ncell[0] * _ncell[1] * _ncell[2];
    _number_density.assign(total, -1.);
    _temperature.assign(total, 0.);
    for (auto &f : _ionic_fraction)
      f.assign(total, 0.);
    for (size_t i = 0; i < n; ++i) {
      size_t index = 0;
      for (int a = 0; a < 3; ++a) {
        const double p = midpoints[3 * i + a] * unit_length_in_SI;
        const long long k = (long long)(_ncell[a] * p / _sides[a]);
        if (k < 0 || k >= _ncell[a])
          throw ParameterError("snapshot cell outside the snapshot's box");
        index = index * _ncell[a] + (size_t)k;
      }
      _number_density[index] = densities[i] * unit_density_in_SI;
      _temperature[index] = temperatures[i] * unit_temperature_in_SI;
      for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
        _ionic_fraction[ion][index] = fractions[ion][i];
    }
    for (size_t index = 0; index < total; ++index)
      if (_number_density[index] < 0.)
        throw ParameterError("No values found for cell " +
                             std::to_string(index) + " of the snapshot!");
  }

  void free() override {
    std::vector<double>().swap(_number_density);
    std::vector<double>().swap(_temperature);
    for (auto &f : _ionic_fraction)
      std::vector<double>().swap(f);
  }

  /* :470-496 */
  DensityValues operator()(const Cell &cell) override {
    const CoordinateVector position = cell.get_cell_midpoint();
    size_t index = 0;
    for (int a = 0; a < 3; ++a) {
      const long long k = (long long)(_ncell[a] * (position[a] - _anchor[a]) /
                                      _sides[a]);
      if (k < 0 || k >= _ncell[a])
        throw ParameterError("cell midpoint outside the snapshot's box");
      index = index * _ncell[a] + (size_t)k;
    }
    DensityValues values;
    values.set_number_density(_number_density[index]);
    values.set_temperature(_temperature[index]);
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      values.set_ionic_fraction(ion, _ionic_fraction[ion][index]);
    return values;
  }
};

/* src/BufferedCMacIonizeSnapshotDensityFunction.hpp:46-716: the density field
 * of a snapshot of the reference's TASK-BASED simulation - cells stored
 * subgrid after subgrid (`DensitySubGridCreator:number of subgrids` in the
 * snapshot's parameters), the cells of a subgrid x-major, no coordinates -
 * for a new box inside the old one at the old resolution or at 1 / k of it
 * (k^3 old cells averaged per new cell, never across subgrid boundaries).
 * The reference keeps `buffer size` subgrids in memory at a time and reads the
 * others on demand; here all are read by initialize() - the values a cell is
 * given are the same, and `DensityFunction:buffer size` is accepted. Values
 * are SI (as in the reference, which applies no units here). */
class BufferedCMacIonizeSnapshotDensityFunction : public DensityFunction {
  const std::string _filename;
  std::array<double, 3> _old_anchor, _subgrid_width;
  std::array<long long, 3> _number_of_subgrids, _original_subgrid_ncell,
      _mapped_subgrid_ncell;
  long long _number_of_old_cells_per_new_cell_1D = 1;
  size_t _original_subgrid_size = 0, _mapped_subgrid_size = 0;
  /* [subgrid][cell of the mapped subgrid] */
  std::vector<double> _number_density, _temperature;
  std::array<std::vector<double>, NUMBER_OF_IONNAMES> _ionic_fraction;

public:
  /* :123-371 */
  BufferedCMacIonizeSnapshotDensityFunction(
      const std::string &filename, const std::array<double, 3> &new_anchor,
      const std::array<double, 3> &new_sides,
      const std::array<long long, 3> &new_ncell)
      : _filename(filename) {
    {
      std::ifstream file(filename);
      if (!file.is_open())
        throw ParameterError("Could not open file \"" + filename + "\"!");
    }
    Hdf5Reader file(filename);
    ParameterFile parameters;
    for (const auto &kv : file.open("/Parameters").attributes)
      parameters.add_value(kv.first, Hdf5Reader::as_string(kv.second));
    if (!parameters.has_value("DensitySubGridCreator:number of subgrids"))
      throw ParameterError(
          "A BufferedCMacIonizeSnapshotDensityFunction can only be used to "
          "read task-based CMacIonize snapshots!");
    _number_of_subgrids = parameters.get_integer_vector(
        "DensitySubGridCreator:number of subgrids", {-1, -1, -1});
    _old_anchor = parameters.get_physical_vector(QUANTITY_LENGTH,
                                                 "SimulationBox:anchor", "");
    const std::array<double, 3> old_sides = parameters.get_physical_vector(
        QUANTITY_LENGTH, "SimulationBox:sides", "");
    const std::array<long long, 3> old_ncell = parameters.get_integer_vector(
        "DensityGrid:number of cells", {-1, -1, -1});
    /* :176-196: the new box inside the old one */
    std::array<double, 3> anchor_in_old_box, available_sides;
    for (int a = 0; a < 3; ++a) {
      anchor_in_old_box[a] = new_anchor[a] - _old_anchor[a];
      available_sides[a] = (new_anchor[a] + new_sides[a]) - _old_anchor[a];
      if (anchor_in_old_box[a] < 0. || available_sides[a] < new_sides[a])
        throw ParameterError(
            "New simulation box is not inside old simulation box!");
    }
    /* :198-240: ... on cell boundaries of the old grid */
    std::array<double, 3> old_cell_size, new_cell_size;
    std::array<long long, 3> cell_sides;
    for (int a = 0; a < 3; ++a) {
      old_cell_size[a] = old_sides[a] / old_ncell[a];
      new_cell_size[a] = new_sides[a] / new_ncell[a];
      const double cell_offset_float = anchor_in_old_box[a] / old_cell_size[a];
      const double cell_sides_float = available_sides[a] / old_cell_size[a];
      const long long cell_offset = (long long)std::round(cell_offset_float);
      cell_sides[a] = (long long)std::round(cell_sides_float);
      if (std::abs(cell_offset_float) - cell_offset > 1.e-10 ||
          std::abs(cell_sides_float) - cell_sides[a] > 1.e-10)
        throw ParameterError("New box not compatible with old resolution!");
    }
    /* :242-262 */
    for (int a = 1; a < 3; ++a)
      if (std::abs(old_sides[0] - old_sides[a]) > 1.e-10 ||
          std::abs(new_sides[0] - new_sides[a]) > 1.e-10 ||
          std::abs(old_cell_size[0] - old_cell_size[a]) > 1.e-10 ||
          std::abs(new_cell_size[0] - new_cell_size[a]) > 1.e-10)
        throw ParameterError("Buffered snapshot reading currently only works "
                             "for square boxes and cells!");
    /* :263-272 */
    if (cell_sides[0] <= new_ncell[0]) {
      _number_of_old_cells_per_new_cell_1D = 1;
    } else {
      if (cell_sides[0] % new_ncell[0] != 0)
        throw ParameterError(
            "New resolution not compatible with old resolution!");
      _number_of_old_cells_per_new_cell_1D = cell_sides[0] / new_ncell[0];
    }
    /* :273-316 */
    for (int a = 0; a < 3; ++a) {
      _original_subgrid_ncell[a] = old_ncell[a] / _number_of_subgrids[a];
      _subgrid_width[a] = old_sides[a] / _number_of_subgrids[a];
      const double subgrid_offset_float =
          anchor_in_old_box[a] / _subgrid_width[a];
      const long long subgrid_offset =
          (long long)std::round(subgrid_offset_float);
      if (_number_of_old_cells_per_new_cell_1D > 1 &&
          (std::abs(subgrid_offset_float - subgrid_offset) > 1.e-10 ||
           _original_subgrid_ncell[0] % _number_of_old_cells_per_new_cell_1D !=
               0))
        throw ParameterError("Degrading resolution across subgrid boundaries "
                             "not yet supported!");
      _mapped_subgrid_ncell[a] =
          _original_subgrid_ncell[a] / _number_of_old_cells_per_new_cell_1D;
    }
    _original_subgrid_size = (size_t)(_original_subgrid_ncell[0] *
                                      _original_subgrid_ncell[1] *
                                      _original_subgrid_ncell[2]);
    _mapped_subgrid_size =
        (size_t)(_mapped_subgrid_ncell[0] * _mapped_subgrid_ncell[1] *
                 _mapped_subgrid_ncell[2]);
    if (!file.exists("/PartType0/NumberDensity") &&
        !file.exists("/PartType0/Density"))
      throw ParameterError("No density variable present in snapshot file!");
    if (!file.exists("/PartType0/Temperature") &&
        !file.exists("/PartType0/Pressure"))
      throw ParameterError("No temperature variable present in snapshot file!");
  }
  explicit BufferedCMacIonizeSnapshotDensityFunction(ParameterFile &params)
      : BufferedCMacIonizeSnapshotDensityFunction(
            (params.get_integer("DensityFunction:buffer size", 100),
             params.get_filename("DensityFunction:filename")),
            params.get_physical_vector(QUANTITY_LENGTH, "SimulationBox:anchor",
                                       ""),
            params.get_physical_vector(QUANTITY_LENGTH, "SimulationBox:sides",
                                       ""),
            params.get_integer_vector("DensityGrid:number of cells",
                                      {-1, -1, -1})) {}

  /* buffer_subgrid for every subgrid, :452-585 */
  void initialize() override {
    Hdf5Reader file(_filename);
    const bool read_number_density = file.exists("/PartType0/NumberDensity");
    const bool read_temperature = file.exists("/PartType0/Temperature");
    std::vector<double> number_density = file.read_doubles(
        read_number_density ? "/PartType0/NumberDensity" : "/PartType0/Density");
    std::vector<double> temperature = file.read_doubles(
        read_temperature ? "/PartType0/Temperature" : "/PartType0/Pressure");
    const size_t nsub = (size_t)(_number_of_subgrids[0] *
                                 _number_of_subgrids[1] * _number_of_subgrids[2]);
    const size_t n = nsub * _original_subgrid_size;
    if (number_density.size() != n || temperature.size() != n)
      throw ParameterError("snapshot with " +
                           std::to_string(number_density.size()) +
                           " cells, its parameters say " + std::to_string(n));
    std::array<std::vector<double>, NUMBER_OF_IONNAMES> fractions;
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion) {
      const std::string name =
          std::string("/PartType0/NeutralFraction") + ion_name(ion);
      if (file.exists(name))
        fractions[ion] = file.read_doubles(name);
      else
        fractions[ion].assign(n, 1.e-6);
      if (fractions[ion].size() != n)
        throw ParameterError("snapshot dataset " + name + " has the wrong size");
    }
    for (size_t i = 0; i < n; ++i) {
      if (!read_number_density)
        number_density[i] /= constants::proton_mass;
      if (!read_temperature) {
        const double mu = 0.5 * (1. + fractions[ION_H_n][i]);
        temperature[i] *= mu / (number_density[i] * constants::boltzmann);
      }
    }
    const long long k = _number_of_old_cells_per_new_cell_1D;
    const double norm = 1. / (double)(k * k * k);
    _number_density.assign(nsub * _mapped_subgrid_size, 0.);
    _temperature.assign(nsub * _mapped_subgrid_size, 0.);
    for (auto &f : _ionic_fraction)
      f.assign(nsub * _mapped_subgrid_size, 0.);
    for (size_t subgrid = 0; subgrid < nsub; ++subgrid) {
      const size_t from = subgrid * _original_subgrid_size;
      const size_t to = subgrid * _mapped_subgrid_size;
      for (long long ix = 0; ix < _mapped_subgrid_ncell[0]; ++ix)
        for (long long iy = 0; iy < _mapped_subgrid_ncell[1]; ++iy)
          for (long long iz = 0; iz < _mapped_subgrid_ncell[2]; ++iz) {
            const size_t mapped =
                to + (size_t)((ix * _mapped_subgrid_ncell[1] + iy) *
                                  _mapped_subgrid_ncell[2] +
                              iz);
            for (long long oix = 0; oix < k; ++oix)
              for (long long oiy = 0; oiy < k; ++oiy)
                for (long long oiz = 0; oiz < k; ++oiz) {
                  const size_t original =
                      from + (size_t)(((k * ix + oix) *
                                           _original_subgrid_ncell[1] +
                                       (k * iy + oiy)) *
                                          _original_subgrid_ncell[2] +
                                      (k * iz + oiz));
                  _number_density[mapped] += number_density[original];
                  _temperature[mapped] += temperature[original];
                  for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
                    _ionic_fraction[ion][mapped] += fractions[ion][original];
                }
            _number_density[mapped] *= norm;
            _temperature[mapped] *= norm;
            for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
              _ionic_fraction[ion][mapped] *= norm;
          }
    }
  }

  void free() override {
    std::vector<double>().swap(_number_density);
    std::vector<double>().swap(_temperature);
    for (auto &f : _ionic_fraction)
      std::vector<double>().swap(f);
  }

  /* :640-712 */
  DensityValues operator()(const Cell &cell) override {
    const CoordinateVector p = cell.get_cell_midpoint();
    size_t subgrid = 0, index = 0;
    for (int a = 0; a < 3; ++a) {
      const long long si =
          (long long)((p[a] - _old_anchor[a]) / _subgrid_width[a]);
      if (si < 0 || si >= _number_of_subgrids[a])
        throw ParameterError("cell midpoint outside the snapshot's box");
      const double subgrid_anchor = _old_anchor[a] + si * _subgrid_width[a];
      long long ci = (long long)((p[a] - subgrid_anchor) / _subgrid_width[a] *
                                 _mapped_subgrid_ncell[a]);
      if (ci >= _mapped_subgrid_ncell[a]) /* (rounding at a subgrid's top) */
        ci = _mapped_subgrid_ncell[a] - 1;
      subgrid = subgrid * (size_t)_number_of_subgrids[a] + (size_t)si;
      index = index * (size_t)_mapped_subgrid_ncell[a] + (size_t)ci;
    }
    const size_t at = subgrid * _mapped_subgrid_size + index;
    DensityValues values;
    values.set_number_density(_number_density[at]);
    values.set_temperature(_temperature[at]);
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      values.set_ionic_fraction(ion, _ionic_fraction[ion][at]);
    return values;
  }
};

/* src/GadgetSnapshotDensityFunction.cpp:60-372: the gas particles of a Gadget
 * / SWIFT HDF5 snapshot (PartType0: Coordinates, Masses, SmoothingLength,
 * Density, optionally Temperature and NeutralFractionH) mapped onto the
 * cells with the cubic spline kernel at the cell midpoints - density = sum of
 * m W(r / h, h) over the particles whose kernel covers the midpoint,
 * temperature and neutral fraction as kernel-weighted means. The reference
 * finds those particles with an Octree; here a uniform grid of bins (side =
 * the largest smoothing length) does: the same set, hence the same sums up to
 * their order. */
class GadgetSnapshotDensityFunction : public DensityFunction {
  std::vector<double> _positions; /* [n][3], m */
  std::vector<double> _masses, _smoothing_lengths, _densities, _temperatures,
      _neutral_fractions;
  bool _periodic = false;
  std::array<double, 3> _box_sides = {0., 0., 0.}; /* periodic boxes, m */
  /* bins over the particles */
  std::array<double, 3> _bin_anchor = {0., 0., 0.}, _bin_side = {1., 1., 1.};
  std::array<int, 3> _nbin = {1, 1, 1};
  std::vector<uint32_t> _bin_start, _bin_particles;

  /* src/CubicSplineKernel.hpp:36-59 */
  static double kernel(double u, double h) {
    const double KC1 = 2.546479089470, KC2 = 15.278874536822,
                 KC5 = 5.092958178941;
    if (u < 1.) {
      if (u < 0.5)
        return (KC1 + KC2 * (u - 1.) * u * u) / (h * h * h);
      return KC5 * (1. - u) * (1. - u) * (1. - u) / (h * h * h);
    }
    return 0.;
  }
  int bin_of(double x, int a) const {
    int i = (int)std::floor((x - _bin_anchor[a]) / _bin_side[a]);
    if (_periodic)
      return ((i % _nbin[a]) + _nbin[a]) % _nbin[a];
    return i < 0 ? 0 : (i >= _nbin[a] ? _nbin[a] - 1 : i);
  }
  void build_bins() {
    const size_t n = _masses.size();
    double hmax = 0.;
    std::array<double, 3> lo = {DBL_MAX, DBL_MAX, DBL_MAX},
                          hi = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (size_t i = 0; i < n; ++i) {
      hmax = std::max(hmax, _smoothing_lengths[i]);
      for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(lo[a], _positions[3 * i + a]);
        hi[a] = std::max(hi[a], _positions[3 * i + a]);
      }
    }
    size_t total = 1;
    for (int a = 0; a < 3; ++a) {
      const double extent = _periodic ? _box_sides[a] : hi[a] - lo[a];
      _bin_anchor[a] = _periodic ? 0. : lo[a];
      int nb = hmax > 0. ? (int)std::floor(extent / hmax) : 1;
      nb = std::max(1, std::min(nb, 256));
      _nbin[a] = nb;
      _bin_side[a] = extent > 0. ? extent / nb : 1.;
      total *= (size_t)nb;
    }
    std::vector<uint32_t> count(total + 1, 0);
    std::vector<size_t> bin(n);
    for (size_t i = 0; i < n; ++i) {
      bin[i] = ((size_t)bin_of(_positions[3 * i], 0) * _nbin[1] +
                bin_of(_positions[3 * i + 1], 1)) *
                   _nbin[2] +
               bin_of(_positions[3 * i + 2], 2);
      ++count[bin[i] + 1];
    }
    for (size_t b = 0; b < total; ++b)
      count[b + 1] += count[b];
    _bin_start = count;
    _bin_particles.resize(n);
    std::vector<uint32_t> cursor(count.begin(), count.end() - 1);
    for (size_t i = 0; i < n; ++i)
      _bin_particles[cursor[bin[i]]++] = (uint32_t)i;
  }

public:
  GadgetSnapshotDensityFunction(const std::string &name, bool fallback_periodic,
                                double fallback_unit_length_in_SI,
                                double fallback_unit_mass_in_SI,
                                double fallback_unit_temperature_in_SI,
                                bool use_neutral_fraction,
                                double fallback_temperature,
                                bool comoving_integration,
                                double hubble_parameter) {
    Hdf5Reader file(name);
    auto number = [](const Hdf5Reader::Object &o, const char *attribute) {
      const auto it = o.attributes.find(attribute);
      if (it == o.attributes.end())
        throw ParameterError(std::string("snapshot without \"") + attribute +
                             "\"");
      return Hdf5Reader::as_doubles(it->second);
    };
    /* :75-102 */
    _periodic = fallback_periodic;
    if (file.exists("/RuntimePars"))
      _periodic =
          number(file.open("/RuntimePars"), "PeriodicBoundariesOn").at(0) != 0.;
    std::array<double, 3> sides = {0., 0., 0.};
    if (_periodic) {
      const std::vector<double> box = number(file.open("/Header"), "BoxSize");
      for (int a = 0; a < 3; ++a)
        sides[a] = box.size() >= 3 ? box[a] : box.at(0);
    }
    /* :104-160 */
    double unit_length_in_SI = fallback_unit_length_in_SI,
           unit_mass_in_SI = fallback_unit_mass_in_SI,
           unit_temperature_in_SI = fallback_unit_temperature_in_SI;
    if (file.exists("/Units")) {
      const Hdf5Reader::Object units = file.open("/Units");
      unit_length_in_SI = 0.01 * number(units, "Unit length in cgs (U_L)").at(0);
      unit_mass_in_SI = 0.001 * number(units, "Unit mass in cgs (U_M)").at(0);
      unit_temperature_in_SI =
          number(units, "Unit temperature in cgs (U_T)").at(0);
    } else {
      if (unit_length_in_SI == 0.)
        unit_length_in_SI = 1.;
      if (unit_mass_in_SI == 0.)
        unit_mass_in_SI = 1.;
      if (unit_temperature_in_SI == 0.)
        unit_temperature_in_SI = 1.;
    }
    if (comoving_integration) {
      unit_length_in_SI /= hubble_parameter;
      unit_mass_in_SI /= hubble_parameter;
    }
    const double unit_density_in_SI = unit_mass_in_SI / unit_length_in_SI /
                                      (unit_length_in_SI * unit_length_in_SI);
    /* :162-198 */
    _positions = file.read_doubles("/PartType0/Coordinates");
    _masses = file.read_doubles("/PartType0/Masses");
    _smoothing_lengths = file.read_doubles("/PartType0/SmoothingLength");
    _densities = file.read_doubles("/PartType0/Density");
    const size_t n = _masses.size();
    if (_positions.size() != 3 * n || _smoothing_lengths.size() != n ||
        _densities.size() != n)
      throw ParameterError("snapshot \"" + name +
                           "\": particle datasets of different lengths");
    if (file.exists("/PartType0/Temperature")) {
      _temperatures = file.read_doubles("/PartType0/Temperature");
    } else {
      if (fallback_temperature == 0.)
        fallback_temperature = 8000.;
      /* (the fallback is a temperature in K already; the loop below scales
       * every temperature by the unit, as the reference does) */
      _temperatures.assign(n, fallback_temperature);
    }
    if (use_neutral_fraction && file.exists("/PartType0/NeutralFractionH"))
      _neutral_fractions = file.read_doubles("/PartType0/NeutralFractionH");
    /* :200-221 */
    for (size_t i = 0; i < n; ++i) {
      for (int a = 0; a < 3; ++a)
        _positions[3 * i + a] *= unit_length_in_SI;
      _masses[i] *= unit_mass_in_SI;
      _smoothing_lengths[i] *= unit_length_in_SI;
      _densities[i] *= unit_density_in_SI;
      _temperatures[i] *= unit_temperature_in_SI;
    }
    for (int a = 0; a < 3; ++a)
      _box_sides[a] = sides[a] * unit_length_in_SI;
    build_bins();
  }
  explicit GadgetSnapshotDensityFunction(ParameterFile &params)
      : GadgetSnapshotDensityFunction(
            params.get_filename("DensityFunction:filename"),
            params.get_bool("DensityFunction:fallback periodic flag", false),
            params.get_physical_value(QUANTITY_LENGTH,
                                      "DensityFunction:fallback unit length",
                                      "0. m"),
            params.get_physical_value(QUANTITY_MASS,
                                      "DensityFunction:fallback unit mass",
                                      "0. kg"),
            params.get_physical_value(
                QUANTITY_TEMPERATURE,
                "DensityFunction:fallback unit temperature", "0. K"),
            params.get_bool("DensityFunction:use neutral fraction", false),
            params.get_physical_value(
                QUANTITY_TEMPERATURE,
                "DensityFunction:fallback initial temperature", "0. K"),
            params.get_bool("DensityFunction:comoving integration flag", false),
            params.get_double("DensityFunction:hubble parameter", 0.7)) {}

  /* :313-359 */
  DensityValues operator()(const Cell &cell) override {
    const CoordinateVector position = cell.get_cell_midpoint();
    double density = 0., temperature = 0.;
    double neutral_fraction = _neutral_fractions.empty() ? -1. : 0.;
    int visited[3][3], nvisit[3];
    for (int a = 0; a < 3; ++a) {
      nvisit[a] = 0;
      const int c = bin_of(position[a], a);
      for (int o = -1; o <= 1; ++o) {
        int i = c + o;
        if (_periodic)
          i = ((i % _nbin[a]) + _nbin[a]) % _nbin[a];
        else if (i < 0 || i >= _nbin[a])
          continue;
        bool seen = false;
        for (int k = 0; k < nvisit[a]; ++k)
          seen |= visited[a][k] == i;
        if (!seen)
          visited[a][nvisit[a]++] = i;
      }
    }
    for (int ix = 0; ix < nvisit[0]; ++ix)
      for (int iy = 0; iy < nvisit[1]; ++iy)
        for (int iz = 0; iz < nvisit[2]; ++iz) {
          const size_t bin =
              ((size_t)visited[0][ix] * _nbin[1] + visited[1][iy]) * _nbin[2] +
              visited[2][iz];
          for (uint32_t k = _bin_start[bin]; k < _bin_start[bin + 1]; ++k) {
            const size_t index = _bin_particles[k];
            double r2 = 0.;
            for (int a = 0; a < 3; ++a) {
              double d = position[a] - _positions[3 * index + a];
              if (_periodic) {
                /* Box::periodic_distance, src/Box.hpp */
                if (d < -0.5 * _box_sides[a])
                  d += _box_sides[a];
                if (d >= 0.5 * _box_sides[a])
                  d -= _box_sides[a];
              }
              r2 += d * d;
            }
            const double h = _smoothing_lengths[index];
            const double u = std::sqrt(r2) / h;
            if (!(u < 1.))
              continue;
            const double splineval = _masses[index] * kernel(u, h);
            density += splineval;
            temperature +=
                splineval * _temperatures[index] / _densities[index];
            if (neutral_fraction >= 0.)
              neutral_fraction += splineval * _neutral_fractions[index];
          }
        }
    DensityValues values;
    values.set_number_density(density / 1.6737236e-27);
    values.set_temperature(temperature);
    values.set_ionic_fraction(ION_H_n, neutral_fraction >= 0.
                                           ? neutral_fraction / density
                                           : 1.e-6);
    values.set_ionic_fraction(ION_He_n, 1.e-6);
    return values;
  }

  /* :366-372 */
  double get_total_hydrogen_number() const {
    double mtot = 0.;
    for (double m : _masses)
      mtot += m;
    return mtot / 1.6737236e-27;
  }
};

/* src/FLASHSnapshotDensityFunction.cpp:46-177,204-226: the density (and
 * temperature) field of a FLASH snapshot - an adaptive mesh of blocks of n^3
 * cells, "refine level" levels deep, of which the blocks of "node type" 1 are
 * the leaves that hold data. A cell of the new grid takes the values of the
 * FLASH cell its midpoint lies in. The reference rebuilds the mesh as an
 * AMRGrid (an octree per top-level block) and walks down it for every
 * query; here the leaf blocks sit in a hash map keyed by (level, block
 * indices at that level) and a query tries the levels from the deepest
 * upwards: the same cell - FLASH's blocks ARE the octree's nodes. Read with
 * the dependency-free Hdf5Reader (the runtime-parameter dictionaries are
 * datasets of {name, value} records, HDF5Tools::read_dictionary).
 * "read cosmic ray heating" (:100-175,228-280: a per-cell heating factor from
 * the magnetic field and the cosmic-ray energy gradient) needs a per-cell
 * term the engine's temperature solve does not have: refused. */
class FLASHSnapshotDensityFunction : public DensityFunction {
  std::array<double, 3> _anchor, _sides;
  std::array<int64_t, 3> _nblock;
  std::array<uint64_t, 3> _block_cells; /* cells of a block along x, y, z */
  int _deepest_level = 0;
  /* (level, ix, iy, iz) of a leaf block -> its index in the file */
  std::unordered_map<uint64_t, uint32_t> _leaves;
  std::vector<double> _density, _temperature; /* [block][z][y][x], SI */
  std::vector<double> _block_anchor, _block_sides; /* [block][3], m */
  const double _fixed_temperature;

  static uint64_t key(int level, int64_t ix, int64_t iy, int64_t iz) {
    return ((uint64_t)level << 57) | ((uint64_t)ix << 38) |
           ((uint64_t)iy << 19) | (uint64_t)iz;
  }
  /* index along axis a of the level's block that holds x */
  int64_t block_index(int level, int a, double x) const {
    const double side = _sides[a] / (double)(_nblock[a] << (level - 1));
    return (int64_t)std::floor((x - _anchor[a]) / side);
  }

public:
  FLASHSnapshotDensityFunction(const std::string &filename, double temperature,
                               bool read_cosmic_ray_heating)
      : _fixed_temperature(temperature) {
    if (read_cosmic_ray_heating)
      throw ParameterError(
          "DensityFunction:read cosmic ray heating is not on this path (the "
          "engine's temperature solve has no per-cell cosmic ray factor)");
    Hdf5Reader file(filename);
    /* centimetres, g cm^-3, K */
    const double unit_length_in_SI = 0.01, unit_density_in_SI = 1000.;
    std::map<std::string, double> real_runtime_pars =
        file.read_dictionary("/real runtime parameters");
    std::map<std::string, double> integer_runtime_pars =
        file.read_dictionary("/integer runtime parameters");
    auto need = [&](std::map<std::string, double> &pars, const char *name) {
      const auto it = pars.find(name);
      if (it == pars.end())
        throw ParameterError("FLASH snapshot \"" + filename +
                             "\" without the runtime parameter \"" + name +
                             "\"");
      return it->second;
    };
    const char *lo[3] = {"xmin", "ymin", "zmin"};
    const char *hi[3] = {"xmax", "ymax", "zmax"};
    const char *nb[3] = {"nblockx", "nblocky", "nblockz"};
    for (int a = 0; a < 3; ++a) {
      _anchor[a] = need(real_runtime_pars, lo[a]) * unit_length_in_SI;
      _sides[a] =
          need(real_runtime_pars, hi[a]) * unit_length_in_SI - _anchor[a];
      _nblock[a] = (int64_t)need(integer_runtime_pars, nb[a]);
      if (_nblock[a] < 1 || !(_sides[a] > 0.))
        throw ParameterError("FLASH snapshot \"" + filename +
                             "\": bad box or number of blocks");
    }
    const Hdf5Reader::Object extents = file.open("/bounding box");
    const Hdf5Reader::Object dens = file.open("/dens");
    if (extents.dims.size() != 3 || extents.dims[1] != 3 ||
        extents.dims[2] != 2 || dens.dims.size() != 4 ||
        dens.dims[0] != extents.dims[0])
      throw ParameterError("FLASH snapshot \"" + filename +
                           "\": unexpected dataset shapes");
    const size_t nblocks = extents.dims[0];
    /* the file's order is [block][z][y][x] */
    _block_cells = {dens.dims[3], dens.dims[2], dens.dims[1]};
    const std::vector<double> box = file.read_doubles("/bounding box");
    _density = file.read_doubles("/dens");
    for (double &rho : _density)
      rho *= unit_density_in_SI;
    if (temperature <= 0.) {
      _temperature = file.read_doubles("/temp");
      if (_temperature.size() != _density.size())
        throw ParameterError("FLASH snapshot \"" + filename +
                             "\": \"temp\" does not match \"dens\"");
    }
    const std::vector<double> levels = file.read_doubles("/refine level");
    const std::vector<double> nodetypes = file.read_doubles("/node type");
    if (levels.size() != nblocks || nodetypes.size() != nblocks)
      throw ParameterError("FLASH snapshot \"" + filename +
                           "\": one level and node type per block expected");
    _block_anchor.resize(3 * nblocks);
    _block_sides.resize(3 * nblocks);
    for (size_t i = 0; i < nblocks; ++i) {
      for (int a = 0; a < 3; ++a) {
        _block_anchor[3 * i + a] = box[(i * 3 + a) * 2] * unit_length_in_SI;
        _block_sides[3 * i + a] =
            box[(i * 3 + a) * 2 + 1] * unit_length_in_SI -
            _block_anchor[3 * i + a];
      }
      if ((int)nodetypes[i] != 1)
        continue;
      const int level = (int)levels[i]; /* FLASH counts from 1 */
      if (level < 1 || level > 19)
        throw ParameterError("FLASH snapshot \"" + filename +
                             "\": refinement level out of range");
      _deepest_level = std::max(_deepest_level, level);
      /* the block's place among the blocks of its level, from its middle */
      int64_t index[3];
      for (int a = 0; a < 3; ++a)
        index[a] = block_index(level, a, _block_anchor[3 * i + a] +
                                             0.5 * _block_sides[3 * i + a]);
      /* (the key packs an index into 19 bits: refuse a file whose blocks do
       * not fit instead of letting two leaves share a key) */
      for (int a = 0; a < 3; ++a)
        if (index[a] < 0 || index[a] >= ((int64_t)1 << 19))
          throw ParameterError("FLASH snapshot \"" + filename +
                               "\": more than 2^19 blocks along an axis at "
                               "refinement level " + std::to_string(level));
      _leaves[key(level, index[0], index[1], index[2])] = (uint32_t)i;
    }
    if (_leaves.empty())
      throw ParameterError("FLASH snapshot \"" + filename +
                           "\" holds no leaf blocks");
  }
  explicit FLASHSnapshotDensityFunction(ParameterFile &params)
      : FLASHSnapshotDensityFunction(
            params.get_filename("DensityFunction:filename"),
            params.get_physical_value(QUANTITY_TEMPERATURE,
                                      "DensityFunction:temperature", "-1. K"),
            params.get_bool("DensityFunction:read cosmic ray heating",
                            false)) {}

  /* :204-226 */
  DensityValues operator()(const Cell &cell) override {
    const CoordinateVector position = cell.get_cell_midpoint();
    for (int level = _deepest_level; level >= 1; --level) {
      int64_t index[3];
      bool inside = true;
      for (int a = 0; a < 3; ++a) {
        index[a] = block_index(level, a, position[a]);
        inside &= index[a] >= 0 && index[a] < (_nblock[a] << (level - 1));
      }
      if (!inside)
        break;
      const auto it = _leaves.find(key(level, index[0], index[1], index[2]));
      if (it == _leaves.end())
        continue;
      const size_t block = it->second;
      uint64_t c[3];
      for (int a = 0; a < 3; ++a) {
        const double u = (position[a] - _block_anchor[3 * block + a]) /
                         _block_sides[3 * block + a] * (double)_block_cells[a];
        c[a] = u <= 0. ? 0
                       : std::min<uint64_t>((uint64_t)u, _block_cells[a] - 1);
      }
      const size_t at =
          ((block * _block_cells[2] + c[2]) * _block_cells[1] + c[1]) *
              _block_cells[0] +
          c[0];
      DensityValues values;
      values.set_number_density(_density[at] / 1.6737236e-27);
      values.set_temperature(_fixed_temperature <= 0. ? _temperature[at]
                                                      : _fixed_temperature);
      values.set_ionic_fraction(ION_H_n, 1.e-6);
      values.set_ionic_fraction(ION_He_n, 1.e-6);
      return values;
    }
    throw ParameterError("a cell midpoint lies outside the FLASH snapshot's "
                         "box");
  }
};

/* src/AmunSnapshotDensityFunction.cpp:55-276: the density and temperature
 * field of an AMUN snapshot: a uniform grid in `number of files` HDF5 files,
 * file f holding the brick (f / pdims[1] % pdims[0], f % pdims[1], f /
 * (pdims[0] pdims[1])) of dims[0] x dims[1] x dims[2] cells ("/attributes":
 * dims, pdims; "/variables": dens, pres, velx, vely, velz as [z][y][x]
 * floats), in code units: rescaled to the given average number density and
 * temperature (temperature = pressure / density, isothermal sound speed
 * `AMUN soundspeed` in code units); the box is periodic and may be shifted.
 * The velocities are read by the reference for its hydro integrator and have
 * no use on this path: not kept. */
class AmunSnapshotDensityFunction : public DensityFunction {
  const std::string _folder, _prefix;
  const uint_fast32_t _padding, _number_of_files;
  const std::array<double, 3> _box_anchor, _box_sides;
  const double _average_number_density, _sound_speed, _average_temperature,
      _initial_neutral_fraction;
  const std::array<double, 3> _shift;
  std::array<uint64_t, 3> _number_of_cells = {0, 0, 0};
  std::vector<double> _number_densities, _temperatures;

  /* Utilities::compose_filename, src/Utilities.hpp */
  std::string filename(uint_fast32_t index) const {
    std::string number = std::to_string(index);
    while (number.size() < _padding)
      number = "0" + number;
    std::string folder = _folder;
    if (!folder.empty() && folder.back() != '/')
      folder += "/";
    return folder + _prefix + number + ".h5";
  }

public:
  AmunSnapshotDensityFunction(const std::string &folder,
                              const std::string &prefix, uint_fast32_t padding,
                              uint_fast32_t number_of_files,
                              const std::array<double, 3> &box_anchor,
                              const std::array<double, 3> &box_sides,
                              double number_density, double sound_speed,
                              double temperature,
                              double initial_neutral_fraction,
                              const std::array<double, 3> &shift)
      : _folder(folder), _prefix(prefix), _padding(padding),
        _number_of_files(number_of_files), _box_anchor(box_anchor),
        _box_sides(box_sides), _average_number_density(number_density),
        _sound_speed(sound_speed), _average_temperature(temperature),
        _initial_neutral_fraction(initial_neutral_fraction), _shift(shift) {}
  explicit AmunSnapshotDensityFunction(ParameterFile &params)
      : AmunSnapshotDensityFunction(
            params.get_string("DensityFunction:folder", "."),
            params.get_string("DensityFunction:prefix", ""),
            (uint_fast32_t)params.get_integer("DensityFunction:padding", 5),
            (uint_fast32_t)params.get_integer(
                "DensityFunction:number of files", 4),
            params.get_physical_vector(QUANTITY_LENGTH,
                                       "DensityFunction:box anchor",
                                       "[0. m, 0. m, 0. m]"),
            params.get_physical_vector(QUANTITY_LENGTH,
                                       "DensityFunction:box sides",
                                       "[1. m, 1. m, 1. m]"),
            params.get_physical_value(QUANTITY_NUMBER_DENSITY,
                                      "DensityFunction:average number density",
                                      "100. cm^-3"),
            params.get_double("DensityFunction:AMUN soundspeed", 0.1),
            params.get_physical_value(QUANTITY_TEMPERATURE,
                                      "DensityFunction:average temperature",
                                      "100. K"),
            params.get_double("DensityFunction:initial neutral fraction",
                              1.e-6),
            plain_vector(params.get_string("DensityFunction:shift",
                                           "[0., 0., 0.]"))) {
    if (_prefix.empty())
      throw ParameterError("\"DensityFunction:prefix\" not found");
  }
  static std::array<double, 3> plain_vector(const std::string &text) {
    std::array<double, 3> v;
    if (std::sscanf(text.c_str(), " [ %lf , %lf , %lf ]", &v[0], &v[1],
                    &v[2]) != 3)
      throw ParameterError("bad vector \"" + text + "\"");
    return v;
  }

  /* :103-218 */
  void initialize() override {
    std::vector<double> dims, pdims;
    {
      Hdf5Reader file(filename(0));
      const Hdf5Reader::Object attributes = file.open("/attributes");
      auto vector3 = [&](const char *name) {
        const auto it = attributes.attributes.find(name);
        if (it == attributes.attributes.end())
          throw ParameterError("AMUN snapshot \"" + filename(0) +
                               "\" without \"" + name + "\"");
        const std::vector<double> v = Hdf5Reader::as_doubles(it->second);
        if (v.size() != 3 || !(v[0] >= 1. && v[1] >= 1. && v[2] >= 1.))
          throw ParameterError("AMUN snapshot: bad \"" + std::string(name) +
                               "\"");
        return v;
      };
      dims = vector3("dims");
      pdims = vector3("pdims");
    }
    const uint64_t d[3] = {(uint64_t)dims[0], (uint64_t)dims[1],
                           (uint64_t)dims[2]};
    const uint64_t pd[3] = {(uint64_t)pdims[0], (uint64_t)pdims[1],
                            (uint64_t)pdims[2]};
    for (int a = 0; a < 3; ++a)
      _number_of_cells[a] = d[a] * pd[a];
    const uint64_t totnumcell =
        _number_of_cells[0] * _number_of_cells[1] * _number_of_cells[2];
    _number_densities.assign(totnumcell, 0.);
    _temperatures.assign(totnumcell, 0.);
    double average_density = 0.;
    for (uint_fast32_t ifile = 0; ifile < _number_of_files; ++ifile) {
      /* the brick of this file */
      const uint64_t brick_z = ifile / (pd[0] * pd[1]);
      const uint64_t brick_x = (ifile - brick_z * pd[0] * pd[1]) / pd[1];
      const uint64_t brick_y = ifile - brick_z * pd[0] * pd[1] - brick_x * pd[1];
      if (brick_z >= pd[2])
        throw ParameterError("AMUN snapshot: more files than bricks");
      const uint64_t offset[3] = {brick_x * d[0], brick_y * d[1],
                                  brick_z * d[2]};
      Hdf5Reader file(filename(ifile));
      const std::vector<double> dens = file.read_doubles("/variables/dens");
      const std::vector<double> pres = file.read_doubles("/variables/pres");
      if (dens.size() != d[0] * d[1] * d[2] || pres.size() != dens.size())
        throw ParameterError("AMUN snapshot \"" + filename(ifile) +
                             "\": variables do not match \"dims\"");
      for (uint64_t iz = 0; iz < d[2]; ++iz)
        for (uint64_t iy = 0; iy < d[1]; ++iy)
          for (uint64_t ix = 0; ix < d[0]; ++ix) {
            const uint64_t in_file = (iz * d[1] + iy) * d[0] + ix;
            const uint64_t in_grid =
                ((iz + offset[2]) * _number_of_cells[1] + iy + offset[1]) *
                    _number_of_cells[0] +
                ix + offset[0];
            _number_densities[in_grid] = dens[in_file];
            _temperatures[in_grid] = pres[in_file] / dens[in_file];
            average_density += dens[in_file];
          }
    }
    average_density /= (double)totnumcell;
    const double number_density_unit =
        _average_number_density / average_density;
    const double temperature_conversion_factor =
        _average_temperature / (_sound_speed * _sound_speed);
    for (uint64_t i = 0; i < totnumcell; ++i) {
      _number_densities[i] *= number_density_unit;
      _temperatures[i] *= temperature_conversion_factor;
    }
  }
  void free() override {
    _number_densities.clear();
    _temperatures.clear();
  }

  /* :234-270 */
  DensityValues operator()(const Cell &cell) override {
    const CoordinateVector midpoint = cell.get_cell_midpoint();
    uint64_t index[3];
    for (int a = 0; a < 3; ++a) {
      double dx = midpoint[a] - _box_anchor[a];
      dx -= _shift[a] * _box_sides[a];
      while (dx >= _box_sides[a])
        dx -= _box_sides[a];
      while (dx < 0.)
        dx += _box_sides[a];
      index[a] = (uint64_t)(dx / _box_sides[a] * (double)_number_of_cells[a]);
      if (index[a] >= _number_of_cells[a]) /* dx one ulp below the side */
        index[a] = _number_of_cells[a] - 1;
    }
    const uint64_t at =
        (index[2] * _number_of_cells[1] + index[1]) * _number_of_cells[0] +
        index[0];
    DensityValues values;
    values.set_number_density(_number_densities[at]);
    values.set_temperature(_temperatures[at]);
    values.set_ionic_fraction(ION_H_n, _initial_neutral_fraction);
    return values;
  }
};

DensityFunction *generate_sph_snapshot_density_function(const std::string &type,
                                                        ParameterFile &params);
inline DensityFunction *generate_density_function(ParameterFile &params) {
  const std::string type =
      params.get_string("DensityFunction:type", "Homogeneous");
  if (type == "Homogeneous")
    return new HomogeneousDensityFunction(params);
  if (type == "BlockSyntax")
    return new BlockSyntaxDensityFunction(params);
  if (type == "CMacIonizeSnapshot")
    return new CMacIonizeSnapshotDensityFunction(params);
  if (type == "BufferedCMacIonizeSnapshot")
    return new BufferedCMacIonizeSnapshotDensityFunction(params);
  if (type == "GadgetSnapshot")
    return new GadgetSnapshotDensityFunction(params);
  if (type == "FLASHSnapshot")
    return new FLASHSnapshotDensityFunction(params);
  if (type == "AmunSnapshot")
    return new AmunSnapshotDensityFunction(params);
  /* the binary dumps of SPH codes: SphSnapshots.hpp */
  if (DensityFunction *f = generate_sph_snapshot_density_function(type, params))
    return f;
  throw ParameterError("Unknown DensityFunction type: \"" + type +
                       "\" (this engine provides Homogeneous, BlockSyntax, "
                       "CMacIonizeSnapshot, BufferedCMacIonizeSnapshot, "
                       "GadgetSnapshot, FLASHSnapshot, AmunSnapshot, "
                       "PhantomSnapshot and SPHNGSnapshot; pass your own DensityFunction to "
                       "initialize())");
}

/* ---------------------------------------------- PhotonSourceDistribution */

typedef uint_fast32_t photonsourcenumber_t;

class PhotonSourceDistribution {
public:
  virtual ~PhotonSourceDistribution() {}
  virtual photonsourcenumber_t get_number_of_sources() const = 0;
  virtual CoordinateVector get_position(photonsourcenumber_t index) = 0;
  virtual double get_weight(photonsourcenumber_t index) const = 0;
  virtual double get_total_luminosity() const = 0;

  /* generic lowering: works for any implementation of the four getters */
  int lower(cmi_gpu_engine *engine) {
    const photonsourcenumber_t n = get_number_of_sources();
    std::vector<double> pos(3 * n), w(n);
    for (photonsourcenumber_t i = 0; i < n; ++i) {
      const CoordinateVector p = get_position(i);
      for (int a = 0; a < 3; ++a)
        pos[3 * i + a] = p[a];
      w[i] = get_weight(i);
    }
    return cmi_gpu_set_sources(engine, (int32_t)n, pos.data(), w.data(),
                               get_total_luminosity());
  }
};

/* src/AsciiFilePhotonSourceDistribution.hpp:45-120: several stars, positions
 * and luminosities from a YAML file
 *   number of sources: N
 *   source[i]:
 *     position: [x, y, z]
 *     luminosity: L
 * The weights are the luminosities over their sum; the engine picks a source
 * per packet by the cumulative weights (PhotonSource.cpp:74-93,222-227). */
class AsciiFilePhotonSourceDistribution : public PhotonSourceDistribution {
  std::vector<CoordinateVector> _source_positions;
  std::vector<double> _source_luminosities;
  double _total_luminosity = 0.;

public:
  explicit AsciiFilePhotonSourceDistribution(const std::string &filename) {
    ParameterFile blocks(filename);
    const long long n = blocks.get_integer("number of sources", -1);
    if (n < 0)
      throw ParameterError("Parameter \"number of sources\" not found in \"" +
                           filename + "\"!");
    for (long long i = 0; i < n; ++i) {
      const std::string b = "source[" + std::to_string(i) + "]:";
      if (!blocks.has_value(b + "position") ||
          !blocks.has_value(b + "luminosity"))
        throw ParameterError("Source " + std::to_string(i) + " of \"" +
                             filename +
                             "\" needs a position and a luminosity!");
      const std::array<double, 3> pos = blocks.get_physical_vector(
          QUANTITY_LENGTH, b + "position", "[0. m, 0. m, 0. m]");
      _source_positions.push_back(CoordinateVector(pos[0], pos[1], pos[2]));
      _source_luminosities.push_back(blocks.get_physical_value(
          QUANTITY_FREQUENCY, b + "luminosity", "0. s^-1"));
      _total_luminosity += _source_luminosities.back();
    }
    std::ofstream ofile(filename + ".used-values");
    blocks.print_contents(ofile);
  }
  explicit AsciiFilePhotonSourceDistribution(ParameterFile &params)
      : AsciiFilePhotonSourceDistribution(params.get_string(
            "PhotonSourceDistribution:filename", "sources.yml")) {}
  photonsourcenumber_t get_number_of_sources() const override {
    return (photonsourcenumber_t)_source_positions.size();
  }
  CoordinateVector get_position(photonsourcenumber_t index) override {
    return _source_positions[index];
  }
  double get_weight(photonsourcenumber_t index) const override {
    return _source_luminosities[index] / _total_luminosity;
  }
  double get_total_luminosity() const override { return _total_luminosity; }
};

/* src/SingleStarPhotonSourceDistribution.hpp:41-100 */
class SingleStarPhotonSourceDistribution : public PhotonSourceDistribution {
  const CoordinateVector _position;
  const double _luminosity;

public:
  SingleStarPhotonSourceDistribution(CoordinateVector position,
                                     double luminosity)
      : _position(position), _luminosity(luminosity) {}
  explicit SingleStarPhotonSourceDistribution(ParameterFile &params)
      : _position([&] {
          const auto p = params.get_physical_vector(
              QUANTITY_LENGTH, "PhotonSourceDistribution:position",
              "[0. pc, 0. pc, 0. pc]");
          return CoordinateVector(p[0], p[1], p[2]);
        }()),
        _luminosity(params.get_physical_value(
            QUANTITY_FREQUENCY, "PhotonSourceDistribution:luminosity",
            "4.26e49 s^-1")) {}
  photonsourcenumber_t get_number_of_sources() const override { return 1; }
  CoordinateVector get_position(photonsourcenumber_t) override {
    return _position;
  }
  double get_weight(photonsourcenumber_t) const override { return 1.; }
  double get_total_luminosity() const override { return _luminosity; }
};

inline PhotonSourceDistribution *
generate_photon_source_distribution(ParameterFile &params) {
  const std::string type =
      params.get_string("PhotonSourceDistribution:type", "SingleStar");
  if (type == "SingleStar")
    return new SingleStarPhotonSourceDistribution(params);
  if (type == "AsciiFile")
    return new AsciiFilePhotonSourceDistribution(params);
  if (type == "None")
    return nullptr;
  throw ParameterError("PhotonSourceDistribution type \"" + type +
                       "\" is not on this path (SingleStar, AsciiFile, None)");
}

/* -------------------------------------------------- PhotonSourceSpectrum */

/* RandomGenerator with the interface of src/RandomGenerator.hpp:207-236. On
 * this path the per-packet streams live on the device; the host's generator
 * exists for plugins that are known only through
 * PhotonSourceSpectrum::get_random_frequency: the generic lowering hands them
 * a generator whose NEXT uniform it dictates (script()) and counts the draws
 * (draws()), or lets it run free (a SplitMix64 stream: the reference's ranlxd
 * is not reproduced, nothing depends on the host stream's values). */
class RandomGenerator {
  uint64_t _state;
  int_fast32_t _seed;
  bool _scripted = false;
  double _script_value = 0.5;
  uint64_t _draws = 0;

  uint64_t next_bits() {
    uint64_t z = (_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }

public:
  RandomGenerator(int_fast32_t seed = 42) { set_seed(seed); }
  void set_seed(int_fast32_t seed) {
    _seed = seed;
    _state = (uint64_t)seed * 0xD1342543DE82EF95ull + 1ull;
  }
  int_fast32_t get_seed() const { return _seed; }
  /* uniform in the open interval (0, 1) */
  double get_uniform_random_double() {
    ++_draws;
    if (_scripted)
      return _script_value;
    return ((double)(next_bits() >> 12) + 0.5) * 0x1.0p-52;
  }
  int_fast32_t get_random_integer() {
    ++_draws;
    return (int_fast32_t)(next_bits() >> 33);
  }
  /* every following draw returns `value` (until unscript()) */
  void script(double value) {
    _scripted = true;
    _script_value = value;
    _draws = 0;
  }
  void unscript() {
    _scripted = false;
    _draws = 0;
  }
  uint64_t draws() const { return _draws; }
};

class PhotonSourceSpectrum;
/* a spectrum as its quantile function: cumulative[k] -> frequency[k] */
struct SpectrumTable {
  std::vector<double> frequency, cumulative;
  int32_t interpolation = CMI_GPU_TABLE_LINEAR;
  /* how the table was obtained: "scripted" (the virtual is a monotone
   * function of ONE uniform - every sampler of the reference is - and was
   * evaluated on a grid of it) or "empirical" (quantiles of 2^20 draws) */
  const char *method = "";
};
inline SpectrumTable tabulate_spectrum(const PhotonSourceSpectrum &spectrum,
                                       size_t n = 8193);

class PhotonSourceSpectrum {
public:
  virtual ~PhotonSourceSpectrum() {}
  /* per-packet virtual of the reference (src/PhotonSourceSpectrum.hpp:48-50);
   * on this path spectra are sampled on the device from the descriptor set by
   * lower() */
  virtual double get_random_frequency(RandomGenerator &random_generator,
                                      double temperature = 0.) const = 0;
  virtual double get_total_flux() const = 0;
  /* Generic lowering: a class that implements only the two
   * virtuals above is sampled on the host into its quantile function, which
   * the device reads with one uniform per packet - the reference's own table
   * samplers do the same with their own tables. Classes the device knows in
   * closed form override this. */
  virtual int lower(cmi_gpu_engine *engine) const {
    const SpectrumTable t = tabulate_spectrum(*this);
    return cmi_gpu_set_spectrum_table(engine, CMI_GPU_ROLE_SOURCE,
                                      (int32_t)t.frequency.size(),
                                      t.frequency.data(), t.cumulative.data(),
                                      t.interpolation);
  }
  /* the same spectrum in the role of ContinuousPhotonSourceSpectrum */
  virtual int lower_continuous(cmi_gpu_engine *engine) const {
    const SpectrumTable t = tabulate_spectrum(*this);
    return cmi_gpu_set_spectrum_table(engine, CMI_GPU_ROLE_CONTINUOUS,
                                      (int32_t)t.frequency.size(),
                                      t.frequency.data(), t.cumulative.data(),
                                      t.interpolation);
  }
  /* --describe */
  virtual std::string describe() const { return "\"Table\""; }
};

/* The quantile function of a spectrum from get_random_frequency alone. First
 * with dictated uniforms u_k = k / (n - 1) (the ends moved half an ulp into
 * the open interval): if every call consumes exactly one uniform and the
 * frequencies are monotone in it, they ARE the quantile function on that grid
 * (ascending as they are, or read backwards for a sampler that uses 1 - u).
 * Otherwise (rejection samplers, several uniforms) the quantiles of 2^20 free
 * draws. */
inline SpectrumTable tabulate_spectrum(const PhotonSourceSpectrum &spectrum,
                                       size_t n) {
  SpectrumTable t;
  t.frequency.resize(n);
  t.cumulative.resize(n);
  for (size_t k = 0; k < n; ++k)
    t.cumulative[k] = (double)k / (double)(n - 1);
  RandomGenerator rg(42);
  bool one_draw = true, ascending = true, descending = true;
  for (size_t k = 0; k < n && one_draw; ++k) {
    const double u = std::min(std::max(t.cumulative[k], 0x1.0p-53),
                              1. - 0x1.0p-53);
    rg.script(u);
    t.frequency[k] = spectrum.get_random_frequency(rg, 0.);
    one_draw = rg.draws() == 1 || (rg.draws() == 0 && k > 0 &&
                                   t.frequency[k] == t.frequency[0]);
    if (rg.draws() == 0 && k == 0) {
      /* no uniform at all: a line spectrum */
      std::fill(t.frequency.begin(), t.frequency.end(), t.frequency[0]);
      t.method = "scripted";
      return t;
    }
    if (k > 0) {
      ascending &= t.frequency[k] >= t.frequency[k - 1];
      descending &= t.frequency[k] <= t.frequency[k - 1];
    }
  }
  if (one_draw && (ascending || descending)) {
    if (!ascending)
      std::reverse(t.frequency.begin(), t.frequency.end());
    t.method = "scripted";
    return t;
  }
  rg.unscript();
  const size_t ndraw = (size_t)1 << 20;
  std::vector<double> draws(ndraw);
  for (double &d : draws)
    d = spectrum.get_random_frequency(rg, 0.);
  std::sort(draws.begin(), draws.end());
  for (size_t k = 0; k < n; ++k)
    t.frequency[k] = draws[(size_t)((double)k / (double)(n - 1) *
                                    (double)(ndraw - 1))];
  t.method = "empirical";
  return t;
}

/* src/UniformPhotonSourceSpectrum.hpp:36-73: flat between 13.6 and 54.4 eV -
 * the reference's class as it stands, known to this path through its virtuals
 * alone (it goes to the device by the generic lowering above) */
class UniformPhotonSourceSpectrum : public PhotonSourceSpectrum {
public:
  double get_random_frequency(RandomGenerator &random_generator,
                              double = 0.) const override {
    return (1. + 3. * random_generator.get_uniform_random_double()) * 3.289e15;
  }
  double get_total_flux() const override {
    throw ParameterError("This function should not be used!");
  }
  std::string describe() const override { return "\"Uniform\""; }
};

/* plugins from outside this header: name -> constructor, consulted by the
 * factories for any "<Block>:type" they do not know themselves (the
 * reference's factories are if-chains one adds a branch to,
 * src/PhotonSourceSpectrumFactory.hpp:93-113) */
template <typename Plugin, typename... Args> class PluginRegistry {
public:
  typedef std::function<Plugin *(Args...)> Constructor;
  static std::map<std::string, Constructor> &table() {
    static std::map<std::string, Constructor> t;
    return t;
  }
  static Plugin *create(const std::string &type, Args... args) {
    auto it = table().find(type);
    return it == table().end() ? nullptr : it->second(args...);
  }
};
typedef PluginRegistry<PhotonSourceSpectrum, const std::string &,
                       ParameterFile &>
    PhotonSourceSpectrumRegistry;
inline void register_photon_source_spectrum(
    const std::string &type, PhotonSourceSpectrumRegistry::Constructor make) {
  PhotonSourceSpectrumRegistry::table()[type] = make;
}

/* src/MonochromaticPhotonSourceSpectrum.hpp:40-113 */
class MonochromaticPhotonSourceSpectrum : public PhotonSourceSpectrum {
  const double _frequency, _total_flux;

public:
  explicit MonochromaticPhotonSourceSpectrum(double frequency,
                                             double total_flux = -1.)
      : _frequency(frequency), _total_flux(total_flux) {}
  MonochromaticPhotonSourceSpectrum(const std::string &role,
                                    ParameterFile &params)
      : MonochromaticPhotonSourceSpectrum(
            params.get_physical_value(QUANTITY_FREQUENCY, role + ":frequency",
                                      "13.6 eV"),
            params.get_physical_value(QUANTITY_FLUX, role + ":total flux",
                                      "-1. m^-2 s^-1")) {}
  double get_random_frequency(RandomGenerator &, double = 0.) const override {
    return _frequency;
  }
  /* src/MonochromaticPhotonSourceSpectrum.hpp:107-112 */
  double get_total_flux() const override {
    if (_total_flux < 0.)
      throw ParameterError(
          "Total flux was not provided for the monochromatic spectrum!");
    return _total_flux;
  }
  double get_frequency() const { return _frequency; }
  int lower(cmi_gpu_engine *engine) const override {
    return cmi_gpu_set_spectrum_monochromatic(engine, _frequency);
  }
  int lower_continuous(cmi_gpu_engine *engine) const override {
    return cmi_gpu_set_continuous_spectrum_monochromatic(engine, _frequency);
  }
};

/* src/PlanckPhotonSourceSpectrum.cpp:53-147 */
class PlanckPhotonSourceSpectrum : public PhotonSourceSpectrum {
  const double _temperature, _ionizing_flux;

public:
  explicit PlanckPhotonSourceSpectrum(double temperature,
                                      double ionizing_flux = -1.)
      : _temperature(temperature), _ionizing_flux(ionizing_flux) {}
  PlanckPhotonSourceSpectrum(const std::string &role, ParameterFile &params)
      : PlanckPhotonSourceSpectrum(
            params.get_physical_value(QUANTITY_TEMPERATURE,
                                      role + ":temperature", "4.e4 K"),
            params.get_physical_value(QUANTITY_FLUX, role + ":ionizing flux",
                                      "-1. m^-2 s^-1")) {}
  double get_random_frequency(RandomGenerator &, double = 0.) const override {
    throw ParameterError("Planck spectrum is sampled on the device");
  }
  /* src/PlanckPhotonSourceSpectrum.cpp:172-178 */
  double get_total_flux() const override {
    if (_ionizing_flux < 0.)
      throw ParameterError(
          "Ionizing flux was not provided for the Planck spectrum!");
    return _ionizing_flux;
  }
  double get_temperature() const { return _temperature; }
  int lower(cmi_gpu_engine *engine) const override {
    return cmi_gpu_set_spectrum_planck(engine, _temperature);
  }
  int lower_continuous(cmi_gpu_engine *engine) const override {
    return cmi_gpu_set_continuous_spectrum_planck(engine, _temperature);
  }
};

/* src/PhotonSourceSpectrumFactory.hpp:93-113 (types on this path) */
inline PhotonSourceSpectrum *
generate_photon_source_spectrum(const std::string &role,
                                ParameterFile &params,
                                const std::string &default_type =
                                    "Monochromatic") {
  const std::string type = params.get_string(role + ":type", default_type);
  if (type == "Monochromatic")
    return new MonochromaticPhotonSourceSpectrum(role, params);
  if (type == "Planck")
    return new PlanckPhotonSourceSpectrum(role, params);
  if (type == "Uniform")
    return new UniformPhotonSourceSpectrum();
  if (type == "None")
    return nullptr;
  if (PhotonSourceSpectrum *plugin =
          PhotonSourceSpectrumRegistry::create(type, role, params))
    return plugin;
  /* (the reference's other five - FaucherGiguere, Pegase3, PopStar, WMBasic,
   * CastelliKurucz, Masked - read data files the reference ships under
   * data/, which are not part of this path; a class that reads them and
   * implements get_random_frequency plugs in through the registry) */
  throw ParameterError("Unknown PhotonSourceSpectrum type: \"" + type + "\"");
}

/* ------------------------------------------------- ContinuousPhotonSource */

/* src/ContinuousPhotonSource.hpp:40-100 */
class ContinuousPhotonSource {
public:
  virtual ~ContinuousPhotonSource() {}
  virtual double get_total_surface_area() const = 0;
  virtual bool has_total_luminosity() const { return false; }
  virtual double get_total_luminosity() const { return 0.; }
  /* luminosity = what the PhotonSource ctor computes
   * (src/PhotonSource.cpp:104-111) */
  virtual int lower(cmi_gpu_engine *engine, double luminosity) const = 0;
};

/* src/IsotropicContinuousPhotonSource.hpp:40-203: radiation that enters the
 * simulation box isotropically through its faces */
class IsotropicContinuousPhotonSource : public ContinuousPhotonSource {
  const double _sides[3];

public:
  IsotropicContinuousPhotonSource(const double sides[3])
      : _sides{sides[0], sides[1], sides[2]} {}
  double get_total_surface_area() const override {
    return 2. * _sides[0] * _sides[1] + 2. * _sides[0] * _sides[2] +
           2. * _sides[1] * _sides[2];
  }
  int lower(cmi_gpu_engine *engine, double luminosity) const override {
    return cmi_gpu_set_continuous_source(engine, CMI_GPU_CONTINUOUS_ISOTROPIC,
                                         luminosity);
  }
};

/* src/PlanarContinuousPhotonSource.hpp:40-196: a luminous rectangle in a
 * plane perpendicular to a coordinate axis */
class PlanarContinuousPhotonSource : public ContinuousPhotonSource {
  int _axis;
  double _intercept, _anchor[2], _sides[2], _luminosity;

public:
  explicit PlanarContinuousPhotonSource(ParameterFile &params)
      : _axis(0),
        _intercept(params.get_physical_value(
            QUANTITY_LENGTH, "ContinuousPhotonSource:intercept", "0. m")),
        _anchor{params.get_physical_value(QUANTITY_LENGTH,
                                          "ContinuousPhotonSource:anchor 0",
                                          "0. m"),
                params.get_physical_value(QUANTITY_LENGTH,
                                          "ContinuousPhotonSource:anchor 1",
                                          "0. m")},
        _sides{params.get_physical_value(QUANTITY_LENGTH,
                                         "ContinuousPhotonSource:side 0",
                                         "1. m"),
               params.get_physical_value(QUANTITY_LENGTH,
                                         "ContinuousPhotonSource:side 1",
                                         "1. m")},
        _luminosity(params.get_physical_value(
            QUANTITY_FREQUENCY, "ContinuousPhotonSource:luminosity",
            "1.e48 s^-1")) {
    /* get_coordinate_index, :52-67 */
    const std::string name =
        params.get_string("ContinuousPhotonSource:normal axis", "z");
    if (name == "x")
      _axis = 0;
    else if (name == "y")
      _axis = 1;
    else if (name == "z")
      _axis = 2;
    else
      throw ParameterError("Unknown coordinate name: " + name + "!");
  }
  double get_total_surface_area() const override {
    return _sides[0] * _sides[1];
  }
  bool has_total_luminosity() const override { return true; }
  double get_total_luminosity() const override { return _luminosity; }
  int lower(cmi_gpu_engine *engine, double luminosity) const override {
    return cmi_gpu_set_continuous_source_planar(engine, _axis, _intercept,
                                                _anchor, _sides, luminosity);
  }
};

/* src/ContinuousPhotonSourceFactory.hpp (types on this path) */
inline ContinuousPhotonSource *
generate_continuous_photon_source(const double box_sides[3],
                                  ParameterFile &params) {
  const std::string type =
      params.get_string("ContinuousPhotonSource:type", "None");
  if (type == "Isotropic")
    return new IsotropicContinuousPhotonSource(box_sides);
  if (type == "Planar")
    return new PlanarContinuousPhotonSource(params);
  if (type == "None")
    return nullptr;
  throw ParameterError("ContinuousPhotonSource type \"" + type +
                       "\" is not on this path (Isotropic, Planar, None)");
}

/* -------------------------------------- CrossSections / RecombinationRates */

/* 14 functions of one argument sampled on a common grid */
struct IonTable {
  std::vector<double> x;
  std::vector<double> y; /* [NUMBER_OF_IONNAMES][x.size()] */
  int32_t interpolation = CMI_GPU_TABLE_LOGLOG;
};
/* f(ion, x) for the 14 ions on n logarithmically spaced points of
 * [lo, hi], plus - where a function jumps between two neighbouring points
 * (from zero to a finite value, or by more than a factor of 2: an ionization
 * threshold) - the two sides of the jump, found by bisection down to one ulp:
 * the table then has the jump between two adjacent samples */
template <typename F>
inline IonTable tabulate_ions(F f, double lo, double hi, size_t n) {
  std::vector<double> x(n);
  for (size_t k = 0; k < n; ++k)
    x[k] = lo * std::exp(std::log(hi / lo) * (double)k / (double)(n - 1));
  x[n - 1] = hi;
  auto jumps = [&](double a, double b) {
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion) {
      const double fa = f(ion, a), fb = f(ion, b);
      if ((fa == 0.) != (fb == 0.) || fa > 2. * fb || fb > 2. * fa)
        return true;
    }
    return false;
  };
  std::vector<double> extra;
  for (size_t k = 0; k + 1 < n; ++k) {
    if (!jumps(x[k], x[k + 1]))
      continue;
    double a = x[k], b = x[k + 1];
    for (int step = 0; step < 80 && std::nextafter(a, b) < b; ++step) {
      const double mid = 0.5 * (a + b);
      if (jumps(a, mid))
        b = mid;
      else
        a = mid;
    }
    if (a > x[k])
      extra.push_back(a);
    if (b < x[k + 1])
      extra.push_back(b);
  }
  x.insert(x.end(), extra.begin(), extra.end());
  std::sort(x.begin(), x.end());
  x.erase(std::unique(x.begin(), x.end()), x.end());
  IonTable t;
  t.x = x;
  t.y.resize((size_t)NUMBER_OF_IONNAMES * x.size());
  for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
    for (size_t k = 0; k < x.size(); ++k)
      t.y[(size_t)ion * x.size() + k] = f(ion, x[k]);
  return t;
}

class CrossSections {
public:
  virtual ~CrossSections() {}
  /* src/CrossSections.hpp:49-50 (energy: the photon's frequency in Hz) */
  virtual double get_cross_section(const int_fast32_t ion,
                                   const double energy) const = 0;
  /* the virtual on 4096 frequencies between 1e15 Hz (4.1 eV) and 1e17 Hz
   * (414 eV) - every spectrum of this path lies in [13.6, 54.4] eV - with the
   * thresholds resolved (tabulate_ions) */
  IonTable tabulate() const {
    return tabulate_ions(
        [this](int ion, double nu) { return get_cross_section(ion, nu); },
        1.e15, 1.e17, 4096);
  }
  /* generic lowering: a class that implements only get_cross_section */
  virtual int lower(cmi_gpu_engine *engine) const {
    const IonTable t = tabulate();
    return cmi_gpu_set_cross_sections_table(engine, (int32_t)t.x.size(),
                                            t.x.data(), t.y.data(),
                                            t.interpolation);
  }
  virtual std::string describe() const { return "\"Table\""; }
};
typedef PluginRegistry<CrossSections, ParameterFile &> CrossSectionsRegistry;
inline void register_cross_sections(const std::string &type,
                                    CrossSectionsRegistry::Constructor make) {
  CrossSectionsRegistry::table()[type] = make;
}

/* parameter key of an ion's cross section / of the ion that recombines INTO
 * it (src/FixedValueCrossSections.hpp:113-146,
 * src/FixedValueRecombinationRates.hpp:113-150) */
inline const char *xsec_key(int ion) {
  static const char *k[NUMBER_OF_IONNAMES] = {
      "hydrogen_0", "helium_0", "carbon_1", "carbon_2", "nitrogen_0",
      "nitrogen_1", "nitrogen_2", "oxygen_0", "oxygen_1", "neon_0", "neon_1",
      "sulphur_1", "sulphur_2", "sulphur_3"};
  return k[ion];
}
inline const char *recomb_key(int ion) {
  static const char *k[NUMBER_OF_IONNAMES] = {
      "hydrogen_1", "helium_1", "carbon_2", "carbon_3", "nitrogen_1",
      "nitrogen_2", "nitrogen_3", "oxygen_1", "oxygen_2", "neon_1", "neon_2",
      "sulphur_2", "sulphur_3", "sulphur_4"};
  return k[ion];
}

class FixedValueCrossSections : public CrossSections {
  double _cross_sections[NUMBER_OF_IONNAMES];

public:
  explicit FixedValueCrossSections(ParameterFile &params) {
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      _cross_sections[ion] = params.get_physical_value(
          QUANTITY_SURFACE_AREA, std::string("CrossSections:") + xsec_key(ion),
          ion == ION_H_n ? "6.3e-18 cm^2" : "0. m^2");
  }
  double get_cross_section(const int_fast32_t ion,
                           const double) const override {
    return _cross_sections[ion];
  }
  int lower(cmi_gpu_engine *engine) const override {
    return cmi_gpu_set_cross_sections_fixed(engine, _cross_sections);
  }
};

class VernerCrossSections : public CrossSections {
public:
  double get_cross_section(const int_fast32_t, const double) const override {
    throw ParameterError("Verner cross sections are evaluated on the device");
  }
  int lower(cmi_gpu_engine *engine) const override {
    return cmi_gpu_set_cross_sections_verner(engine);
  }
};

/* src/CrossSectionsFactory.hpp:69-73 */
inline CrossSections *generate_cross_sections(ParameterFile &params) {
  const std::string type = params.get_string("CrossSections:type", "Verner");
  if (type == "FixedValue")
    return new FixedValueCrossSections(params);
  if (type == "Verner")
    return new VernerCrossSections();
  if (CrossSections *plugin = CrossSectionsRegistry::create(type, params))
    return plugin;
  throw ParameterError("Unknown CrossSections type: \"" + type + "\"");
}

class RecombinationRates {
public:
  virtual ~RecombinationRates() {}
  /* src/RecombinationRates.hpp:49 */
  virtual double get_recombination_rate(const int_fast32_t ion,
                                        const double temperature) const = 0;
  /* the virtual on 2048 temperatures between 10 K and 1e9 K */
  IonTable tabulate() const {
    return tabulate_ions(
        [this](int ion, double T) { return get_recombination_rate(ion, T); },
        10., 1.e9, 2048);
  }
  /* generic lowering: a class that implements only get_recombination_rate */
  virtual int lower(cmi_gpu_engine *engine) const {
    const IonTable t = tabulate();
    return cmi_gpu_set_recombination_rates_table(
        engine, (int32_t)t.x.size(), t.x.data(), t.y.data(), t.interpolation);
  }
  virtual std::string describe() const { return "\"Table\""; }
};
typedef PluginRegistry<RecombinationRates, ParameterFile &>
    RecombinationRatesRegistry;
inline void
register_recombination_rates(const std::string &type,
                             RecombinationRatesRegistry::Constructor make) {
  RecombinationRatesRegistry::table()[type] = make;
}

class FixedValueRecombinationRates : public RecombinationRates {
  double _rates[NUMBER_OF_IONNAMES];

public:
  explicit FixedValueRecombinationRates(ParameterFile &params) {
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      _rates[ion] = params.get_physical_value(
          QUANTITY_REACTION_RATE,
          std::string("RecombinationRates:") + recomb_key(ion),
          ion == ION_H_n ? "4.e-13 cm^3 s^-1" : "0. m^3 s^-1");
  }
  double get_recombination_rate(const int_fast32_t ion,
                                const double) const override {
    return _rates[ion];
  }
  int lower(cmi_gpu_engine *engine) const override {
    return cmi_gpu_set_recombination_rates_fixed(engine, _rates);
  }
};

class VernerRecombinationRates : public RecombinationRates {
public:
  double get_recombination_rate(const int_fast32_t,
                                const double) const override {
    throw ParameterError("Verner rates are evaluated on the device");
  }
  int lower(cmi_gpu_engine *engine) const override {
    return cmi_gpu_set_recombination_rates_verner(engine);
  }
};

/* src/RecombinationRatesFactory.hpp:65-67 */
inline RecombinationRates *generate_recombination_rates(ParameterFile &params) {
  const std::string type =
      params.get_string("RecombinationRates:type", "Verner");
  if (type == "FixedValue")
    return new FixedValueRecombinationRates(params);
  if (type == "Verner")
    return new VernerRecombinationRates();
  if (RecombinationRates *plugin =
          RecombinationRatesRegistry::create(type, params))
    return plugin;
  throw ParameterError("Unknown RecombinationRates type: \"" + type + "\"");
}

/* ------------------------------------------------------------ Abundances */

/* src/FixedValueAbundanceModel.hpp:44-52: "AbundanceModel:<element>",
 * default 0; elements He C N O Ne S */
struct Abundances {
  double value[6] = {0., 0., 0., 0., 0., 0.};
  explicit Abundances(ParameterFile &params) {
    const std::string type =
        params.get_string("AbundanceModel:type", "FixedValue");
    if (type != "FixedValue")
      throw ParameterError("Unknown AbundanceModel type: \"" + type + "\"");
    static const char *el[6] = {"He", "C", "N", "O", "Ne", "S"};
    for (int i = 0; i < 6; ++i)
      value[i] =
          params.get_double(std::string("AbundanceModel:") + el[i], 0.);
  }
  int lower(cmi_gpu_engine *engine) const {
    return cmi_gpu_set_abundances(engine, value);
  }
};

/* ----------------------------------------------- DiffuseReemissionHandler */

/* src/DiffuseReemissionHandlerFactory.hpp:60-110 incl. the deprecated
 * "PhotonSource:diffuse field" switch */
struct DiffuseReemission {
  int type = CMI_GPU_REEMIT_NONE;
  double probability = 0.364;
  double frequency = 0.;
  explicit DiffuseReemission(ParameterFile &params) {
    if (!params.has_value("DiffuseReemissionHandler:type") &&
        params.has_value("PhotonSource:diffuse field")) {
      const bool diffuse = params.get_bool("PhotonSource:diffuse field", false);
      params.add_value("DiffuseReemissionHandler:type",
                       diffuse ? "Physical" : "None");
    }
    const std::string t =
        params.get_string("DiffuseReemissionHandler:type", "None");
    if (t == "FixedValue") {
      type = CMI_GPU_REEMIT_FIXED;
      probability = params.get_double(
          "DiffuseReemissionHandler:reemission probability", 0.364);
      frequency = params.get_physical_value(
          QUANTITY_FREQUENCY, "DiffuseReemissionHandler:reemission frequency",
          "19.8 eV");
    } else if (t == "Physical") {
      type = CMI_GPU_REEMIT_PHYSICAL;
    } else if (t == "None") {
      type = CMI_GPU_REEMIT_NONE;
    } else {
      throw ParameterError("Unknown DiffuseReemissionHandler type: \"" + t +
                           "\"!");
    }
  }
  int lower(cmi_gpu_engine *engine) const {
    return cmi_gpu_set_reemission(engine, type, probability, frequency);
  }
};

/* ---------------------------------------------------------- SimulationBox */

struct SimulationBox {
  std::array<double, 3> anchor, sides;
  std::array<bool, 3> periodicity;
  explicit SimulationBox(ParameterFile &params)
      : anchor(params.get_physical_vector(QUANTITY_LENGTH,
                                          "SimulationBox:anchor",
                                          "[0. m, 0. m, 0. m]")),
        sides(params.get_physical_vector(QUANTITY_LENGTH, "SimulationBox:sides",
                                         "[1. m, 1. m, 1. m]")),
        periodicity(params.get_bool_vector("SimulationBox:periodicity",
                                           {false, false, false})) {}
  SimulationBox(const std::array<double, 3> &box_anchor,
                const std::array<double, 3> &box_sides,
                const std::array<bool, 3> &box_periodicity)
      : anchor(box_anchor), sides(box_sides), periodicity(box_periodicity) {}
};

} // namespace cmi

/* (after everything above: it uses the plugin surfaces and the Petkova
 * mapping, which uses them too) */
#include "SphSnapshots.hpp"

#endif
