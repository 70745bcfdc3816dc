/*
 * GpuIonizationSimulation.hpp - the driver of the path, shaped like the
 * reference's IonizationSimulation (src/IonizationSimulation.hpp public
 * section; src/IonizationSimulation.cpp:101-231 ctor, :239-325 initialize,
 * :334-680 run): same constructor / initialize(DensityFunction*) /
 * run(DensityGridWriter*) trio with the same two injection hooks, same
 * parameter keys and defaults, same log lines for the timers - but the
 * iteration body runs on the MI355X engine through the C ABI.
 */
#ifndef CMI_HOST_GPUIONIZATIONSIMULATION_HPP
#define CMI_HOST_GPUIONIZATIONSIMULATION_HPP

#include "Hdf5Writer.hpp"
#include "Plugins.hpp"

#include <array>
#include <chrono>
#include <exception>
#include <thread>
#include <cstring>
#include <cstdio>
#include <ctime>
#include <iomanip>
#include <iostream>
#include <memory>
#include <sstream>

namespace cmi {

/* Host mirror of the grid for writers: the iterator surface of DensityGrid a
 * DensityGridWriter uses (src/DensityGrid.hpp:298-418,707-714,
 * src/CartesianDensityGrid.hpp:85-100) over SoA arrays downloaded from the
 * engine. */
class DensityGrid {
public:
  struct IonizationVariables {
    const DensityGrid *grid;
    int64_t index;
    double get_number_density() const { return grid->_number_density[index]; }
    double get_temperature() const { return grid->_temperature[index]; }
    double get_ionic_fraction(int ion) const {
      return grid->_ionic_fraction[ion][index];
    }
    double get_mean_intensity(int ion) const {
      return grid->_mean_intensity[ion][index];
    }
    double get_heating(int term) const { return grid->_heating[term][index]; }
  };
  class iterator : public Cell {
    const DensityGrid *_grid;
    int64_t _index;

  public:
    iterator(const DensityGrid *grid, int64_t index)
        : _grid(grid), _index(index) {}
    CoordinateVector get_cell_midpoint() const override {
      return _grid->get_cell_midpoint(_index);
    }
    double get_volume() const override { return _grid->get_cell_volume(); }
    std::vector<Face> get_faces() const override {
      return _grid->get_faces(_index);
    }
    IonizationVariables get_ionization_variables() const {
      return IonizationVariables{_grid, _index};
    }
    int64_t get_index() const { return _index; }
    iterator &operator++() {
      ++_index;
      return *this;
    }
    bool operator!=(const iterator &o) const { return _index != o._index; }
    bool operator==(const iterator &o) const { return _index == o._index; }
  };

  DensityGrid(const SimulationBox &box, const std::array<long long, 3> &ncell)
      : _box(box), _ncell(ncell) {
    /* CartesianDensityGrid ctor, src/CartesianDensityGrid.cpp:72-79 */
    for (int a = 0; a < 3; ++a)
      _cellside[a] = box.sides[a] / ncell[a];
    const int64_t n = get_number_of_cells();
    _number_density.assign(n, 0.);
    _temperature.assign(n, 0.);
    for (auto &f : _ionic_fraction)
      f.assign(n, 0.);
    for (auto &f : _mean_intensity)
      f.assign(n, 0.);
    for (auto &f : _heating)
      f.assign(n, 0.);
  }
  int64_t get_number_of_cells() const {
    return (int64_t)_ncell[0] * _ncell[1] * _ncell[2];
  }
  CoordinateVector get_cell_midpoint(int64_t index) const {
    /* get_indices + get_cell + midpoint, src/CartesianDensityGrid.hpp:85-89 */
    const int64_t ix = index / (_ncell[1] * _ncell[2]);
    const int64_t rest = index - ix * _ncell[1] * _ncell[2];
    const int64_t iy = rest / _ncell[2];
    const int64_t iz = rest - iy * _ncell[2];
    const int64_t idx[3] = {ix, iy, iz};
    CoordinateVector m;
    for (int a = 0; a < 3; ++a)
      m[a] = (_box.anchor[a] + _cellside[a] * idx[a]) + 0.5 * _cellside[a];
    return m;
  }
  double get_cell_volume() const {
    return _cellside[0] * _cellside[1] * _cellside[2];
  }
  /* CartesianDensityGrid::get_faces, src/CartesianDensityGrid.cpp:611-733:
   * faces -x, +x, -y, +y, -z, +z; the vertices of a face go round it starting
   * at its corner with the smallest coordinates, first along the lower of its
   * two axes */
  std::vector<Face> get_faces(int64_t index) const {
    static const int ring[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    const CoordinateVector mid = get_cell_midpoint(index);
    std::vector<Face> faces(6);
    for (int axis = 0; axis < 3; ++axis) {
      const int u = axis == 0 ? 1 : 0; /* the face's own two axes */
      const int w = axis == 2 ? 1 : 2;
      for (int sign = 0; sign < 2; ++sign) {
        Face &face = faces[2 * axis + sign];
        face.midpoint = mid;
        face.midpoint[axis] += (sign ? 0.5 : -0.5) * _cellside[axis];
        face.vertices.resize(4);
        for (int v = 0; v < 4; ++v) {
          CoordinateVector &p = face.vertices[v];
          p[axis] = mid[axis] + (sign ? 0.5 : -0.5) * _cellside[axis];
          p[u] = mid[u] + 0.5 * ring[v][0] * _cellside[u];
          p[w] = mid[w] + 0.5 * ring[v][1] * _cellside[w];
        }
      }
    }
    return faces;
  }
  iterator begin() const { return iterator(this, 0); }
  iterator end() const { return iterator(this, get_number_of_cells()); }

  const SimulationBox &box() const { return _box; }
  const std::array<long long, 3> &ncell() const { return _ncell; }

  /* SoA storage, filled by the driver */
  std::vector<double> _number_density, _temperature;
  std::array<std::vector<double>, NUMBER_OF_IONNAMES> _ionic_fraction,
      _mean_intensity;
  std::array<std::vector<double>, 2> _heating;
  /* emission lines asked for with "EmissivityValues:<name>: true", computed
   * for the final state (name of the dataset, values) */
  std::vector<std::pair<std::string, std::vector<double>>> _emissivity;

private:
  SimulationBox _box;
  std::array<long long, 3> _ncell;
  double _cellside[3];
};

/* src/DensityGridWriter.hpp:96-124 (the DensityGrid overload) */
class DensityGridWriter {
protected:
  std::string _output_folder;

public:
  explicit DensityGridWriter(const std::string &output_folder)
      : _output_folder(output_folder) {}
  virtual ~DensityGridWriter() {}
  virtual void write(DensityGrid &grid, uint_fast32_t iteration,
                     ParameterFile &params, double time = 0.) = 0;
};

/* Utilities::compose_filename, src/Utilities.hpp:756-775 */
inline std::string compose_filename(const std::string &folder,
                                    const std::string &prefix,
                                    const std::string &extension,
                                    uint_fast32_t counter,
                                    uint_fast32_t padding) {
  std::stringstream name;
  if (!folder.empty())
    name << folder << "/";
  name << prefix << std::setfill('0') << std::setw((int)padding) << counter
       << "." << extension;
  return name.str();
}

/* src/AsciiFileDensityGridWriter.cpp:36-84: x y z n volume x_H per cell */
class AsciiFileDensityGridWriter : public DensityGridWriter {
  std::string _prefix;

public:
  AsciiFileDensityGridWriter(const std::string &prefix,
                             const std::string &output_folder)
      : DensityGridWriter(output_folder), _prefix(prefix) {}
  AsciiFileDensityGridWriter(const std::string &output_folder,
                             ParameterFile &params)
      : AsciiFileDensityGridWriter(
            params.get_string("DensityGridWriter:prefix", "snapshot"),
            output_folder) {}
  void write(DensityGrid &grid, uint_fast32_t iteration, ParameterFile &,
             double = 0.) override {
    const std::string filename =
        compose_filename(_output_folder, _prefix, "txt", iteration, 3);
    std::ofstream file(filename);
    file << "#x (m)\ty (m)\tz (m)\tn (m^-3)\tvolume (m^3)\tneutral H "
            "fraction\n";
    for (auto it = grid.begin(); it != grid.end(); ++it) {
      const CoordinateVector x = it.get_cell_midpoint();
      const double n = it.get_ionization_variables().get_number_density();
      const double frac =
          it.get_ionization_variables().get_ionic_fraction(ION_H_n);
      file << x.x() << "\t" << x.y() << "\t" << x.z() << "\t" << n << "\t"
           << it.get_volume() << "\t" << frac << "\n";
    }
  }
};

/* Raw SoA dump of every field (n, T, 14 fractions) as fp64, for checkers.
 * Layout: int64 ncell[3], then 16 arrays of ncell doubles. */
class BinaryDensityGridWriter : public DensityGridWriter {
  std::string _prefix;

public:
  BinaryDensityGridWriter(const std::string &output_folder,
                          ParameterFile &params)
      : DensityGridWriter(output_folder),
        _prefix(params.get_string("DensityGridWriter:prefix", "snapshot")) {}
  void write(DensityGrid &grid, uint_fast32_t iteration, ParameterFile &,
             double = 0.) override {
    const std::string filename =
        compose_filename(_output_folder, _prefix, "bin", iteration, 3);
    std::ofstream file(filename, std::ios::binary);
    const int64_t nc[3] = {grid.ncell()[0], grid.ncell()[1], grid.ncell()[2]};
    file.write((const char *)nc, sizeof nc);
    const size_t bytes = sizeof(double) * grid.get_number_of_cells();
    file.write((const char *)grid._number_density.data(), bytes);
    file.write((const char *)grid._temperature.data(), bytes);
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      file.write((const char *)grid._ionic_fraction[ion].data(), bytes);
  }
};

/* GadgetDensityGridWriter (the reference's default writer),
 * src/GadgetDensityGridWriter.cpp:47-358: an HDF5 file in the Gadget-2 type 3
 * layout - groups /Header /Code /Configuration /Parameters /RuntimePars
 * /Units with attributes, /PartType0 with one dataset per output field
 * (DensityGridWriterFields: Coordinates, NumberDensity, Temperature,
 * NeutralFraction<ion>; src/DensityGridWriterFields.hpp:137-240,797-839 for
 * the names, defaults and the "DensityGridWriterFields:<name>" switches) - so
 * that the reference's own analysis scripts (benchmarks/, *.py) read the snapshots unchanged.
 * Written with the dependency-free Hdf5Writer (the image has no HDF5). */
class GadgetDensityGridWriter : public DensityGridWriter {
  std::string _prefix;
  uint_fast32_t _padding;
  bool _coordinates, _number_density, _temperature;
  bool _neutral_fraction[NUMBER_OF_IONNAMES];

public:
  GadgetDensityGridWriter(const std::string &output_folder,
                          ParameterFile &params)
      : DensityGridWriter(output_folder),
        _prefix(params.get_string("DensityGridWriter:prefix", "snapshot")),
        _padding((uint_fast32_t)params.get_integer("DensityGridWriter:padding",
                                                   3)) {
    /* defaults: DensityGridWriterFields::default_flag without hydro */
    _coordinates =
        params.get_integer("DensityGridWriterFields:Coordinates", 1) != 0;
    _number_density =
        params.get_integer("DensityGridWriterFields:NumberDensity", 1) != 0;
    _temperature =
        params.get_integer("DensityGridWriterFields:Temperature", 0) != 0;
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      _neutral_fraction[ion] =
          params.get_integer(std::string("DensityGridWriterFields:"
                                         "NeutralFraction") +
                                 ion_name(ion),
                             ion == ION_H_n ? 1 : 0) != 0;
  }

  void write(DensityGrid &grid, uint_fast32_t iteration, ParameterFile &params,
             double time = 0.) override {
    const std::string filename = compose_filename(_output_folder, _prefix,
                                                  "hdf5", iteration, _padding);
    Hdf5Writer file;
    const SimulationBox &box = grid.box();
    const uint64_t ncell = (uint64_t)grid.get_number_of_cells();
    if (ncell >= (1ull << 32))
      throw std::runtime_error(
          "Gadget snapshots count cells in 32 bits (NumPart_ThisFile)");
    /* :117-142 */
    file.attribute("Header", "BoxSize",
                   std::vector<double>{box.sides[0], box.sides[1],
                                       box.sides[2]});
    file.attribute("Header", "Dimension", (int32_t)3);
    file.attribute("Header", "Flag_Entropy_ICs", std::vector<uint32_t>(6, 0));
    file.attribute("Header", "MassTable", std::vector<double>(6, 0.));
    file.attribute("Header", "NumFilesPerSnapshot", (int32_t)1);
    std::vector<uint32_t> numpart(6, 0);
    numpart[0] = (uint32_t)ncell;
    file.attribute("Header", "NumPart_ThisFile", numpart);
    file.attribute("Header", "NumPart_Total", numpart);
    file.attribute("Header", "NumPart_Total_HighWord",
                   std::vector<uint32_t>(6, 0));
    file.attribute("Header", "Time", time);
    /* :144-162: what the reference takes from its build system */
    file.attribute("Code", "Code", std::string("cmacionize_amd (MI355X engine "
                                               "behind the CMacIonize plugin "
                                               "interfaces)"));
    file.attribute("Configuration", "NUMBER_OF_IONNAMES",
                   std::to_string(NUMBER_OF_IONNAMES));
    /* :164-171: every parameter with the value that was used */
    file.create_group("Parameters");
    for (const auto &kv : params.used_values())
      file.attribute("Parameters", kv.first, kv.second);
    /* :173-178 */
    {
      const std::time_t now = std::time(nullptr);
      char stamp[64];
      std::strftime(stamp, sizeof stamp, "%d/%m/%Y, %H:%M:%S",
                    std::localtime(&now));
      file.attribute("RuntimePars", "Creation time", std::string(stamp));
      file.attribute("RuntimePars", "Iteration", (uint32_t)iteration);
    }
    /* :180-196: SI units expressed in CGS */
    file.attribute("Units", "Unit current in cgs (U_I)", 1.);
    file.attribute("Units", "Unit length in cgs (U_L)", 100.);
    file.attribute("Units", "Unit mass in cgs (U_M)", 1000.);
    file.attribute("Units", "Unit temperature in cgs (U_T)", 1.);
    file.attribute("Units", "Unit time in cgs (U_t)", 1.);
    /* :198-352, fields in the order of the DensityGridField enum */
    file.create_group("PartType0");
    if (_coordinates) {
      DensityGrid *g = &grid;
      file.dataset("PartType0", "Coordinates", {ncell, 3},
                   [g, ncell](std::ostream &os) {
                     /* cell midpoint - box anchor, :517 */
                     const SimulationBox &b = g->box();
                     std::vector<double> chunk;
                     const uint64_t block = 1 << 16;
                     for (uint64_t first = 0; first < ncell; first += block) {
                       const uint64_t last = std::min(ncell, first + block);
                       chunk.resize(3 * (last - first));
                       for (uint64_t i = first; i < last; ++i) {
                         DensityGrid::iterator it(g, (int64_t)i);
                         const CoordinateVector x = it.get_cell_midpoint();
                         for (int a = 0; a < 3; ++a)
                           chunk[3 * (i - first) + a] = x[a] - b.anchor[a];
                       }
                       os.write(reinterpret_cast<const char *>(chunk.data()),
                                8 * chunk.size());
                     }
                   });
    }
    if (_number_density)
      file.dataset("PartType0", "NumberDensity", grid._number_density);
    if (_temperature)
      file.dataset("PartType0", "Temperature", grid._temperature);
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      if (_neutral_fraction[ion])
        file.dataset("PartType0",
                     std::string("NeutralFraction") + ion_name(ion),
                     grid._ionic_fraction[ion]);
    /* the datasets EmissivityCalculationSimulation appends to a snapshot,
     * src/EmissivityCalculationSimulation.cpp:181-193,258-263 */
    for (const auto &line : grid._emissivity)
      file.dataset("PartType0", line.first, line.second);
    file.write(filename);
  }
};

/* src/DensityGridWriterFactory.hpp:100-103 */
inline DensityGridWriter *generate_writer(const std::string &output_folder,
                                          ParameterFile &params) {
  const std::string type = params.get_string("DensityGridWriter:type", "Gadget");
  if (type == "AsciiFile")
    return new AsciiFileDensityGridWriter(output_folder, params);
  if (type == "Binary")
    return new BinaryDensityGridWriter(output_folder, params);
  if (type == "Gadget")
    return new GadgetDensityGridWriter(output_folder, params);
  throw ParameterError("Unknown DensityGridWriter type: \"" + type + "\"");
}

/* TrackerManager (src/TrackerManager.hpp:41-380) with SpectrumTrackers
 * (src/SpectrumTracker.hpp:41-262) and AbsorptionTrackers
 * (src/AbsorptionTracker.hpp:49-235): the trackers of a YAML block file,
 * placed on the engines for the last iteration (every block of a decomposed
 * grid gets all of them and counts those in its own cells; the counts of
 * blocks and copies are merged, :307-318), normalised and written as the
 * reference's text files - or, with "HDF5 output", as one HDF5 file of tracker
 * groups (:330-367; like the reference's, for AbsorptionTrackers only: its
 * SpectrumTracker has no HDF5 form, src/Tracker.hpp:112-130). */
class TrackerManager {
  /* one engine tracker per LEAF (Spectrum / Absorption); a "Multi" tracker
   * (src/MultiTracker.hpp, src/MultiTracker.cpp:35-50) is a node whose leaves
   * sit at the same position: the engine counts any number of trackers in
   * one cell */
  std::vector<double> _positions, _opening_angles, _reference_directions;
  std::vector<int32_t> _kinds;
  std::vector<int32_t> _number_of_bins; /* per leaf */
  struct Node {
    int leaf = -1;                  /* >= 0: a leaf */
    std::vector<Node> children;     /* a Multi tracker's trackers ... */
    std::vector<std::string> names; /* ... and their "output name"s ("" = the
                                       default, <file>.<i>.txt) */
  };
  std::vector<Node> _trackers;            /* the file's tracker[i] */
  std::vector<std::string> _output_names; /* per tracker[i] */
  std::vector<double> _tracker_positions; /* per tracker[i], [3] */
  const uint_fast64_t _number_of_photons;
  const bool _hdf5_output;
  const std::string _hdf5_name;
  std::vector<uint64_t> _counts;
  std::vector<double> _absorption; /* [leaf][4 types][14 ions] */
  /* WeightedSpectrum leaves (src/WeightedSpectrumTracker.hpp): their
   * FrequencyBins (src/FrequencyBinsFactory.hpp:57-72) per leaf, the side of
   * the cell they sit in, and their sums [leaf][4 types][bins of the leaf] */
  std::vector<int32_t> _bins_type;
  std::vector<double> _bins_minimum, _bins_maximum;
  double _side_length = 0.;
  std::vector<double> _flux;
  /* HDF5 output (src/TrackerManager.hpp:141-161): trackers that are the
   * same_group() share a group of the file */
  std::vector<size_t> _tracker_groups, _group_size, _group_index;

  /* LevelFrequencyBins: the ionization energies in ascending order with their
   * ions (src/LevelFrequencyBins.hpp:52-66, src/ElementData.hpp:39-105) */
  static const std::pair<double, int> *level_bins() {
    static const std::pair<double, int> bins[NUMBER_OF_IONNAMES] = {
        {3.28810279e+15, ION_H_n},   {3.29284691e+15, ION_O_n},
        {3.51435505e+15, ION_N_n},   {5.21432028e+15, ION_Ne_n},
        {5.64310422e+15, ION_S_p1},  {5.89588678e+15, ION_C_p1},
        {5.94523574e+15, ION_He_n},  {7.15759434e+15, ION_N_p1},
        {8.41222200e+15, ION_S_p2},  {8.49136314e+15, ION_O_p1},
        {9.90492110e+15, ION_Ne_p1}, {1.14182796e+16, ION_S_p3},
        {1.14732262e+16, ION_N_p2},  {1.15792700e+16, ION_C_p2}};
    return bins;
  }
  /* FrequencyBins::get_frequency of a weighted leaf's bin
   * (src/LinearFrequencyBins.hpp:133-135: the middle of the bin;
   * src/LevelFrequencyBins.hpp:94-96: its lower edge) */
  double bin_frequency(int t, int32_t bin) const {
    if (_bins_type[t] == CMI_GPU_FREQUENCY_BINS_LEVEL)
      return level_bins()[bin].first;
    const double width =
        (_bins_maximum[t] - _bins_minimum[t]) / _number_of_bins[t];
    return _bins_minimum[t] + (0.5 + bin) * width;
  }
  /* Tracker::same_group (src/AbsorptionTracker.hpp:170-172,
   * src/WeightedSpectrumTracker.hpp:348-357 with FrequencyBins::is_same) */
  bool same_group(const Node &a, const Node &b) const {
    if (a.leaf < 0 || b.leaf < 0 || _kinds[a.leaf] != _kinds[b.leaf])
      return false;
    if (_kinds[a.leaf] == CMI_GPU_TRACKER_ABSORPTION)
      return true;
    const int s = a.leaf, t = b.leaf;
    if (_bins_type[s] != _bins_type[t])
      return false;
    return _bins_type[s] == CMI_GPU_FREQUENCY_BINS_LEVEL ||
           (_number_of_bins[s] == _number_of_bins[t] &&
            _bins_minimum[s] == _bins_minimum[t] &&
            _bins_maximum[s] == _bins_maximum[t]);
  }

  static const char *photontype_name(int type) {
    /* get_photontype_name, src/PhotonType.hpp:64-85 */
    switch (type) {
    case 0:
      return "source photon";
    case 1:
      return "diffuse H photon";
    case 2:
      return "diffuse He photon";
    default:
      return "absorbed photon";
    }
  }

  /* TrackerFactory::generate, src/TrackerFactory.hpp:60-72 */
  Node generate(ParameterFile &blocks, const std::string &name,
                const std::array<double, 3> &x) {
    Node node;
    const std::string type = blocks.get_string(name + "type", "Spectrum");
    if (type == "Multi") {
      /* MultiTracker::MultiTracker, src/MultiTracker.cpp:35-50 */
      const long long n = blocks.get_integer(name + "number of trackers", -1);
      if (n < 0)
        throw ParameterError("\"" + name + "number of trackers\" not found");
      for (long long i = 0; i < n; ++i) {
        const std::string child = name + "tracker[" + std::to_string(i) + "]:";
        node.children.push_back(generate(blocks, child, x));
        node.names.push_back(blocks.get_string(child + "output name", ""));
      }
      return node;
    }
    if (type != "Spectrum" && type != "Absorption" &&
        type != "WeightedSpectrum")
      throw ParameterError("Unknown Tracker type: \"" + type + "\"");
    if (_kinds.size() == 16)
      throw ParameterError("at most 16 trackers");
    const bool absorption = type == "Absorption";
    const bool weighted = type == "WeightedSpectrum";
    node.leaf = (int)_kinds.size();
    _kinds.push_back(absorption ? CMI_GPU_TRACKER_ABSORPTION
                     : weighted ? CMI_GPU_TRACKER_WEIGHTED_SPECTRUM
                                : CMI_GPU_TRACKER_SPECTRUM);
    double angle = 3.141592653589793;
    double v[3] = {0., 0., 0.};
    int32_t bins = 1; /* (an absorption tracker has no spectrum) */
    int32_t bins_type = CMI_GPU_FREQUENCY_BINS_LINEAR;
    double bins_minimum = 0., bins_maximum = 0.;
    if (weighted) {
      /* WeightedSpectrumTracker(name, blocks) ->
       * FrequencyBinsFactory::generate, src/FrequencyBinsFactory.hpp:57-72;
       * LinearFrequencyBins(name, blocks), src/LinearFrequencyBins.hpp:80-88 */
      const std::string bins_name = name + "FrequencyBins:";
      const std::string bt = blocks.get_string(bins_name + "type", "Linear");
      if (bt == "Level") {
        bins_type = CMI_GPU_FREQUENCY_BINS_LEVEL;
        bins = NUMBER_OF_IONNAMES;
      } else if (bt == "Linear") {
        bins = (int32_t)blocks.get_integer(bins_name + "number of bins", 100);
        if (bins < 1)
          throw ParameterError("a tracker needs at least one bin");
        bins_minimum = blocks.get_physical_value(
            QUANTITY_FREQUENCY, bins_name + "minimum frequency", "13.6 eV");
        bins_maximum = blocks.get_physical_value(
            QUANTITY_FREQUENCY, bins_name + "maximum frequency", "54.4 eV");
      } else {
        throw ParameterError("Unknown FrequencyBins type: \"" + bt + "\".");
      }
    }
    _bins_type.push_back(bins_type);
    _bins_minimum.push_back(bins_minimum);
    _bins_maximum.push_back(bins_maximum);
    if (!absorption && !weighted) {
      bins = (int32_t)blocks.get_integer(name + "number of bins", 100);
      if (bins < 1)
        throw ParameterError("a tracker needs at least one bin");
      angle = blocks.get_physical_value(QUANTITY_ANGLE, name + "opening angle",
                                        "180. degrees");
      /* (a vector of plain numbers) */
      const std::string d =
          blocks.get_string(name + "reference direction", "[0., 0., 0.]");
      if (std::sscanf(d.c_str(), " [ %lf , %lf , %lf ]", &v[0], &v[1],
                      &v[2]) != 3)
        throw ParameterError("bad reference direction \"" + d + "\"");
    }
    _opening_angles.push_back(angle);
    _number_of_bins.push_back(bins);
    for (int a = 0; a < 3; ++a) {
      _positions.push_back(x[a]);
      _reference_directions.push_back(v[a]);
    }
    return node;
  }

  size_t first_bin_of(int leaf) const {
    size_t first = 0;
    for (int t = 0; t < leaf; ++t)
      first += (size_t)_number_of_bins[t];
    return first;
  }

  /* SpectrumTracker::output_tracker, src/SpectrumTracker.hpp:226-238;
   * AbsorptionTracker::output_tracker, src/AbsorptionTracker.hpp:145-160 */
  void output_leaf(int t, const std::string &filename) const {
    const double minimum_frequency = 3.289e15;
    const int32_t nbins = _number_of_bins[t];
    const double frequency_width = 3. * 3.289e15 / nbins;
    const uint64_t *c = _counts.data() + 3 * first_bin_of(t);
    std::ofstream ofile(filename);
    if (_kinds[t] == CMI_GPU_TRACKER_WEIGHTED_SPECTRUM) {
      /* WeightedSpectrumTracker::output_tracker, :325-341 */
      const double *f = _flux.data() + 4 * first_bin_of(t);
      ofile << "# frequency (Hz)";
      for (int type = 0; type < 4; ++type)
        ofile << "\t" << photontype_name(type) << " flux (s^-1 m^-2)";
      ofile << "\n";
      for (int32_t i = 0; i < nbins; ++i) {
        ofile << bin_frequency(t, i);
        for (int type = 0; type < 4; ++type)
          ofile << "\t" << f[(size_t)type * (size_t)nbins + i];
        ofile << "\n";
      }
      return;
    }
    if (_kinds[t] == CMI_GPU_TRACKER_ABSORPTION) {
      ofile << "# Ion ";
      for (int type = 0; type < 4; ++type)
        ofile << "\t" << photontype_name(type);
      ofile << "\n";
      for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion) {
        ofile << ion_name(ion);
        for (int type = 0; type < 4; ++type)
          ofile << "\t"
                << _absorption[((size_t)t * 4 + (size_t)type) *
                                   NUMBER_OF_IONNAMES +
                               ion];
        ofile << "\n";
      }
      return;
    }
    ofile << "# frequency (Hz)\tprimary count\tdiffuse H count\tdiffuse He "
             "count\n";
    for (int32_t i = 0; i < nbins; ++i) {
      const double nu = minimum_frequency + (i + 0.5) * frequency_width;
      ofile << nu << "\t" << c[i] << "\t" << c[nbins + i] << "\t"
            << c[2 * nbins + i] << "\n";
    }
  }
  /* Tracker::describe: SpectrumTracker.hpp:249-259, AbsorptionTracker.hpp:
   * 230-232; a MultiTracker inside a MultiTracker describes nothing
   * (src/Tracker.hpp:148) */
  void describe(const Node &node, const std::string &prefix,
                std::ostream &stream) const {
    if (node.leaf < 0)
      return;
    const int t = node.leaf;
    if (_kinds[t] == CMI_GPU_TRACKER_ABSORPTION) {
      stream << prefix << "type: AbsorptionTracker\n";
      return;
    }
    if (_kinds[t] == CMI_GPU_TRACKER_WEIGHTED_SPECTRUM) {
      /* src/WeightedSpectrumTracker.hpp:441-445 */
      stream << prefix << "type: WeightedSpectrum\n";
      stream << prefix << "number of bins: " << _number_of_bins[t] << "\n";
      return;
    }
    /* (the reference keeps the cosine and the normalised direction) */
    const double *v = &_reference_directions[3 * (size_t)t];
    const double norm2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double scale = norm2 > 0. ? 1. / std::sqrt(norm2) : 1.;
    stream << prefix << "type: Spectrum\n";
    stream << prefix << "number of bins: " << _number_of_bins[t] << "\n";
    stream << prefix << "opening angle: "
           << std::acos(std::cos(_opening_angles[t])) << " radians\n";
    stream << prefix << "reference direction: [" << v[0] * scale << ", "
           << v[1] * scale << ", " << v[2] * scale << "]\n";
  }
  /* Tracker::output_tracker; MultiTracker::output_tracker,
   * src/MultiTracker.cpp:127-143: one file per tracker of the MultiTracker
   * and the named file as their table of contents */
  void output_node(const Node &node, const std::string &filename) const {
    if (node.leaf >= 0) {
      output_leaf(node.leaf, filename);
      return;
    }
    std::ofstream ofile(filename);
    for (size_t i = 0; i < node.children.size(); ++i) {
      std::string this_filename = node.names[i];
      if (this_filename.empty())
        this_filename = filename + "." + std::to_string(i) + ".txt";
      output_node(node.children[i], this_filename);
      ofile << "tracker[" << i << "]:\n";
      ofile << "  output name: " << this_filename << "\n";
      describe(node.children[i], "  ", ofile);
    }
  }

public:
  explicit TrackerManager(ParameterFile &params)
      : _number_of_photons((uint_fast64_t)params.get_integer(
            "TrackerManager:minimum number of photon packets", 0)),
        _hdf5_output(params.get_bool("TrackerManager:HDF5 output", false)),
        _hdf5_name(params.get_string("TrackerManager:HDF5 output name",
                                     "trackers.hdf5")) {
    const std::string filename = params.get_filename("TrackerManager:filename");
    ParameterFile blocks(filename);
    const long long n = blocks.get_integer("number of trackers", -1);
    if (n < 0)
      throw ParameterError("\"number of trackers\" not found in \"" +
                           filename + "\"");
    for (long long i = 0; i < n; ++i) {
      const std::string name = "tracker[" + std::to_string(i) + "]:";
      if (!blocks.has_value(name + "position"))
        throw ParameterError("\"" + name + "position\" not found in \"" +
                             filename + "\"");
      const std::array<double, 3> x =
          blocks.get_physical_vector(QUANTITY_LENGTH, name + "position", "");
      _trackers.push_back(generate(blocks, name, x));
      for (int a = 0; a < 3; ++a)
        _tracker_positions.push_back(x[a]);
      /* src/TrackerManager.hpp:132-138 */
      _output_names.push_back(blocks.get_string(
          name + "output name",
          "Tracker" + std::to_string(i) + (_hdf5_output ? "" : ".txt")));
    }
    if (_hdf5_output) {
      for (const Node &node : _trackers)
        if (node.leaf < 0 || _kinds[node.leaf] == CMI_GPU_TRACKER_SPECTRUM)
          throw ParameterError(
              "TrackerManager:HDF5 output: only Absorption and "
              "WeightedSpectrum trackers have an HDF5 form (as in the "
              "reference, src/Tracker.hpp:112-130)");
      /* src/TrackerManager.hpp:141-161 */
      _group_index.assign(_trackers.size(), 0);
      for (size_t i = 0; i < _trackers.size(); ++i) {
        size_t group_id = 0;
        while (group_id < _tracker_groups.size() &&
               !same_group(_trackers[_tracker_groups[group_id]], _trackers[i]))
          ++group_id;
        if (group_id == _tracker_groups.size()) {
          _tracker_groups.push_back(i);
          _group_size.push_back(0);
        }
        _group_index[i] = group_id;
        ++_group_size[group_id];
      }
    }
    std::ofstream ofile(filename + ".used-values");
    blocks.print_contents(ofile);
  }

  uint_fast64_t get_number_of_photons() const { return _number_of_photons; }
  /* the engine's trackers: the leaves */
  size_t size() const { return _kinds.size(); }
  /* the file's trackers */
  size_t number_of_trackers() const { return _trackers.size(); }

  /* TrackerManager::add_trackers */
  int lower(cmi_gpu_engine *engine) const {
    if (size() == 0)
      return CMI_GPU_OK;
    int rc = cmi_gpu_set_trackers(engine, (int32_t)size(), _positions.data(),
                                  _kinds.data(), _number_of_bins.data(),
                                  _opening_angles.data(),
                                  _reference_directions.data());
    for (size_t t = 0; t < size() && rc == CMI_GPU_OK; ++t)
      if (_kinds[t] == CMI_GPU_TRACKER_WEIGHTED_SPECTRUM)
        rc = cmi_gpu_set_tracker_frequency_bins(engine, (int32_t)t,
                                                _bins_type[t], _bins_minimum[t],
                                                _bins_maximum[t]);
    if (rc == CMI_GPU_OK)
      rc = cmi_gpu_enable_trackers(engine, 1);
    return rc;
  }
  /* Tracker::normalize_for_cell of the cells the trackers sit in
   * (src/TrackerManager.hpp:216,250; WeightedSpectrumTracker::
   * normalize_for_cell, :96-98): the cells of the Cartesian grid all have
   * this volume */
  void normalize_for_cell(const double cell_volume) {
    _side_length = std::cbrt(cell_volume);
  }
  /* the counts of one engine, added to the total (the copies of a tracker
   * are merged, src/TrackerManager.hpp:307-318) */
  int collect(cmi_gpu_engine *engine) {
    if (size() == 0)
      return CMI_GPU_OK; /* a block file without trackers */
    size_t total_bins = 0;
    for (int32_t b : _number_of_bins)
      total_bins += (size_t)b;
    std::vector<uint64_t> part(3 * total_bins);
    int rc = cmi_gpu_get_tracker_counts(engine, part.data());
    if (rc != CMI_GPU_OK)
      return rc;
    _counts.resize(part.size(), 0);
    for (size_t k = 0; k < part.size(); ++k)
      _counts[k] += part[k];
    std::vector<double> sums(4 * NUMBER_OF_IONNAMES * size());
    rc = cmi_gpu_get_tracker_absorption(engine, sums.data());
    if (rc != CMI_GPU_OK)
      return rc;
    _absorption.resize(sums.size(), 0.);
    for (size_t k = 0; k < sums.size(); ++k)
      _absorption[k] += sums[k];
    std::vector<double> flux(4 * total_bins);
    rc = cmi_gpu_get_tracker_flux(engine, flux.data());
    if (rc != CMI_GPU_OK)
      return rc;
    _flux.resize(flux.size(), 0.);
    for (size_t k = 0; k < flux.size(); ++k)
      _flux[k] += flux[k];
    return CMI_GPU_OK;
  }
  /* TrackerManager::normalize -> AbsorptionTracker::normalize
   * (src/AbsorptionTracker.hpp:87-93; a SpectrumTracker keeps its counts) */
  void normalize(const double luminosity_per_weight) {
    for (double &v : _absorption)
      v *= luminosity_per_weight;
    /* WeightedSpectrumTracker::normalize, :106-116 */
    bool weighted = false;
    for (int32_t kind : _kinds)
      weighted = weighted || kind == CMI_GPU_TRACKER_WEIGHTED_SPECTRUM;
    if (!weighted)
      return;
    if (!(_side_length > 0.))
      throw std::runtime_error("Tracker was not normalized!");
    const double norm = luminosity_per_weight / (_side_length * _side_length);
    for (double &v : _flux)
      v *= norm;
  }
  /* TrackerManager::output_trackers, :323-375 */
  void output_trackers() const {
    if (_hdf5_output) {
      /* one group of the file per set of trackers that are the same_group():
       * AbsorptionTracker::create_group / append_to_group
       * (src/AbsorptionTracker.hpp:179-223), WeightedSpectrumTracker's
       * (src/WeightedSpectrumTracker.hpp:366-428) */
      Hdf5Writer file;
      auto table_of = [](const std::vector<double> &table) {
        return [table](std::ostream &os) {
          os.write(reinterpret_cast<const char *>(table.data()),
                   8 * table.size());
        };
      };
      for (size_t igroup = 0; igroup < _tracker_groups.size(); ++igroup) {
        const std::string group = "Group" + std::to_string(igroup);
        std::vector<int> leaves; /* of the group's trackers, in file order */
        std::vector<double> positions;
        std::vector<std::string> labels;
        for (size_t i = 0; i < _trackers.size(); ++i)
          if (_group_index[i] == igroup) {
            leaves.push_back(_trackers[i].leaf);
            for (int a = 0; a < 3; ++a)
              positions.push_back(_tracker_positions[3 * i + a]);
            labels.push_back(_output_names[i]);
          }
        const uint64_t n = leaves.size();
        const int first = leaves[0];
        if (_kinds[first] == CMI_GPU_TRACKER_ABSORPTION) {
          file.attribute(group, "type", std::string("Absorption"));
          std::vector<std::string> ion_names;
          for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
            ion_names.push_back(ion_name(ion));
          file.dataset(group, "ion name", ion_names);
          for (int type = 0; type < 4; ++type) {
            std::vector<double> table(n * NUMBER_OF_IONNAMES);
            for (size_t k = 0; k < n; ++k)
              for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
                table[k * NUMBER_OF_IONNAMES + ion] =
                    _absorption[((size_t)leaves[k] * 4 + (size_t)type) *
                                    NUMBER_OF_IONNAMES +
                                ion];
            file.dataset(group,
                         std::string(photontype_name(type)) + " absorption",
                         {n, (uint64_t)NUMBER_OF_IONNAMES}, table_of(table));
          }
        } else {
          const uint64_t nbins = (uint64_t)_number_of_bins[first];
          file.attribute(group, "type", std::string("WeightedSpectrum"));
          file.attribute(group, "frequency unit", std::string("s^-1"));
          file.attribute(group, "flux unit", std::string("m^-2 s^-1"));
          std::vector<double> frequencies(nbins);
          for (uint64_t i = 0; i < nbins; ++i)
            frequencies[i] = bin_frequency(first, (int32_t)i);
          file.dataset(group, "frequencies", {nbins}, table_of(frequencies));
          if (_bins_type[first] == CMI_GPU_FREQUENCY_BINS_LEVEL) {
            /* LevelFrequencyBins::get_label, src/LevelFrequencyBins.hpp:
             * 106-108 */
            std::vector<std::string> bin_labels;
            for (uint64_t i = 0; i < nbins; ++i)
              bin_labels.push_back(ion_name(level_bins()[i].second));
            file.dataset(group, "bin labels", bin_labels);
          }
          for (int type = 0; type < 4; ++type) {
            std::vector<double> table(n * nbins);
            for (size_t k = 0; k < n; ++k) {
              const double *f = _flux.data() + 4 * first_bin_of(leaves[k]) +
                                (size_t)type * nbins;
              for (uint64_t i = 0; i < nbins; ++i)
                table[k * nbins + i] = f[i];
            }
            file.dataset(group, std::string(photontype_name(type)) + " flux",
                         {n, nbins}, table_of(table));
          }
        }
        file.dataset(group, "positions", {n, 3}, table_of(positions));
        file.dataset(group, "tracker labels", labels);
        file.attribute(group, "position unit", std::string("m"));
      }
      file.write(_hdf5_name);
      return;
    }
    for (size_t i = 0; i < _trackers.size(); ++i)
      output_node(_trackers[i], _output_names[i]);
  }
};

class GpuIonizationSimulation {
  const bool _every_iteration_output;
  const bool _output_statistics;
  const bool _verbose;
  ParameterFile _parameter_file;
  uint_fast32_t _number_of_iterations;
  uint_fast64_t _number_of_photons;
  uint_fast64_t _number_of_photons_init;
  int32_t _random_seed;

  Abundances _abundances;
  std::unique_ptr<CrossSections> _cross_sections;
  std::unique_ptr<RecombinationRates> _recombination_rates;
  std::unique_ptr<DensityFunction> _density_function;
  SimulationBox _simulation_box;
  std::array<long long, 3> _ncell;
  std::unique_ptr<PhotonSourceDistribution> _photon_source_distribution;
  std::unique_ptr<PhotonSourceSpectrum> _photon_source_spectrum;
  std::unique_ptr<ContinuousPhotonSource> _continuous_photon_source;
  std::unique_ptr<PhotonSourceSpectrum> _continuous_photon_source_spectrum;
  DiffuseReemission _reemission;
  cmi_gpu_temperature_params _temperature_params;
  std::unique_ptr<DensityGridWriter> _density_grid_writer;
  std::unique_ptr<DensityGrid> _density_grid;

  cmi_gpu_engine *_engine = nullptr;
  /* domain decomposition (DensitySubGridCreator,
   * src/DensitySubGridCreator.hpp:314-396): one engine per block of the grid,
   * possibly on different devices; _engine is then unused */
  struct Block {
    cmi_gpu_engine *engine = nullptr;
    int32_t offset[3], size[3];
    int device = 0;
    /* a copy of another block (DensitySubGridCreator::create_copies,
     * src/DensitySubGridCreator.hpp:437-531): index of the original */
    int original = -1;
    int64_t ncell() const { return (int64_t)size[0] * size[1] * size[2]; }
  };
  /* the originals first (block (bx,by,bz) at (bx*nby + by)*nbz + bz), then
   * the copies */
  std::vector<Block> _blocks;
  size_t _number_of_copies = 0;
  size_t _number_of_neighbour_copies = 0;
  /* replica mode (the reference's MPI path,
   * src/IonizationSimulation.cpp:394-397,458-529): one engine per device,
   * each holding the whole grid and flying its share of the packets */
  std::vector<cmi_gpu_engine *> _replicas;
  /* the engines of either mode as one group: RCCL all-reduce of the
   * accumulators / device-to-device hand-over of flights */
  cmi_gpu_group *_group = nullptr;
  std::array<int, 3> _nblock = {1, 1, 1};
  std::array<std::vector<int32_t>, 3> _block_edges;
  uint64_t _exchange_rounds = 0, _flights_exchanged = 0;
  double _shoot_seconds = 0., _update_seconds = 0.;
  double _last_totweight = 0.;
  double _last_typecount[4] = {0., 0., 0., 0.};

  void check(int rc, const char *what) const {
    if (rc != CMI_GPU_OK)
      throw std::runtime_error(std::string(what) + ": " + cmi_gpu_last_error());
  }
  void status(const std::string &message) const {
    if (_verbose)
      std::cout << message << std::endl;
  }
  bool decomposed() const { return !_blocks.empty(); }

  /* f(k) for k = 0 .. n - 1, each on its own host thread; the first
   * exception (check() throws with the thread's own error string) is
   * rethrown in the caller */
  template <typename F> static void for_each_engine_in_parallel(size_t n, F f) {
    if (n <= 1) {
      if (n == 1)
        f(0);
      return;
    }
    std::vector<std::exception_ptr> errors(n);
    std::vector<std::thread> threads;
    for (size_t k = 0; k < n; ++k)
      threads.emplace_back([&, k]() {
        try {
          f(k);
        } catch (...) {
          errors[k] = std::current_exception();
        }
      });
    for (std::thread &t : threads)
      t.join();
    for (size_t k = 0; k < n; ++k)
      if (errors[k])
        std::rethrow_exception(errors[k]);
  }

  /* PhotonSource::get_total_luminosity, src/PhotonSource.cpp:95-111 */
  double total_luminosity() const {
    double luminosity = 0.;
    if (_photon_source_distribution)
      luminosity += _photon_source_distribution->get_total_luminosity();
    if (_continuous_photon_source)
      luminosity +=
          _continuous_photon_source->has_total_luminosity()
              ? _continuous_photon_source->get_total_luminosity()
              : _continuous_photon_source->get_total_surface_area() *
                    _continuous_photon_source_spectrum->get_total_flux();
    return luminosity;
  }

  void lower_model(cmi_gpu_engine *engine) {
    if (_photon_source_distribution)
      check(_photon_source_distribution->lower(engine), "sources");
    else
      check(cmi_gpu_set_sources(engine, 0, nullptr, nullptr, 0.), "sources");
    if (_photon_source_spectrum)
      check(_photon_source_spectrum->lower(engine), "spectrum");
    if (_continuous_photon_source) {
      /* PhotonSource ctor, src/PhotonSource.cpp:104-111 */
      const double luminosity =
          _continuous_photon_source->has_total_luminosity()
              ? _continuous_photon_source->get_total_luminosity()
              : _continuous_photon_source->get_total_surface_area() *
                    _continuous_photon_source_spectrum->get_total_flux();
      check(_continuous_photon_source_spectrum->lower_continuous(engine),
            "continuous spectrum");
      check(_continuous_photon_source->lower(engine, luminosity),
            "continuous source");
    }
    check(_cross_sections->lower(engine), "cross sections");
    check(_recombination_rates->lower(engine), "recombination rates");
    check(_abundances.lower(engine), "abundances");
    check(_reemission.lower(engine), "reemission");
    check(cmi_gpu_set_temperature_params(engine, &_temperature_params),
          "temperature parameters");
  }

  /* gather / scatter one field between the whole grid (host) and a block */
  void block_slice(const Block &b, const std::vector<double> &whole,
                   std::vector<double> &part) const {
    part.resize((size_t)b.ncell());
    size_t k = 0;
    for (int32_t ix = 0; ix < b.size[0]; ++ix)
      for (int32_t iy = 0; iy < b.size[1]; ++iy)
        for (int32_t iz = 0; iz < b.size[2]; ++iz)
          part[k++] = whole[(size_t)(((ix + b.offset[0]) * _ncell[1] + iy +
                                      b.offset[1]) *
                                         _ncell[2] +
                                     iz + b.offset[2])];
  }
  void block_unslice(const Block &b, const std::vector<double> &part,
                     std::vector<double> &whole) const {
    size_t k = 0;
    for (int32_t ix = 0; ix < b.size[0]; ++ix)
      for (int32_t iy = 0; iy < b.size[1]; ++iy)
        for (int32_t iz = 0; iz < b.size[2]; ++iz)
          whole[(size_t)(((ix + b.offset[0]) * _ncell[1] + iy + b.offset[1]) *
                             _ncell[2] +
                         iz + b.offset[2])] = part[k++];
  }
  /* the block that owns a cell of the whole grid (long index) */
  size_t owner_of_cell(int64_t cell) const {
    const int64_t c[3] = {cell / (_ncell[1] * _ncell[2]),
                          (cell / _ncell[2]) % _ncell[1], cell % _ncell[2]};
    size_t idx[3];
    for (int a = 0; a < 3; ++a) {
      size_t i = 0;
      while (i + 1 < (size_t)_nblock[a] && c[a] >= _block_edges[a][i + 1])
        ++i;
      idx[a] = i;
    }
    return (idx[0] * _nblock[1] + idx[1]) * _nblock[2] + idx[2];
  }

  void download_field(int field, std::vector<double> &whole) {
    if (!decomposed()) {
      check(cmi_gpu_download_field(_engine, field, whole.data()), "download");
      return;
    }
    std::vector<double> part;
    for (Block &b : _blocks) {
      if (b.original >= 0)
        continue; /* a copy holds what its original holds */
      part.resize((size_t)b.ncell());
      check(cmi_gpu_download_field(b.engine, field, part.data()), "download");
      block_unslice(b, part, whole);
    }
  }

  /* One iteration's transport on a decomposed grid: every block runs through
   * the packet ids and flies those emitted inside it; then hand-over rounds
   * (cmi_gpu_group_exchange_flights) until no flight is left. */
  void shoot_decomposed(uint_fast32_t loop, uint_fast64_t numphoton,
                        double &totweight, double typecount[4]) {
    /* (with re-emission cmi_gpu_shoot blocks its caller while a device
     * works through the generations: one host thread per engine) */
    for_each_engine_in_parallel(_blocks.size(), [&](size_t k) {
      Block &b = _blocks[k];
      check(cmi_gpu_reset_grid(b.engine), "reset_grid");
      check(cmi_gpu_reset_exports(b.engine), "reset_exports");
      check(cmi_gpu_shoot(b.engine, (uint32_t)_random_seed, loop, 0, numphoton),
            "shoot");
    });
    /* rounds of {every flight that left a block into the inbox of the block
     * that owns the cell it enters - written by the source GPU across xGMI -
     * then every block continues what it received} until nothing moves */
    for (;;) {
      uint64_t total = 0;
      check(cmi_gpu_group_exchange_flights(_group, (uint32_t)_random_seed,
                                           loop, 0, &total),
            "exchange_flights");
      if (total == 0)
        break;
      ++_exchange_rounds;
      _flights_exchanged += total;
    }
    /* the copies of a block: sum their integrals into each of them
     * (DensitySubGridCreator::update_original_counters,
     * src/DensitySubGridCreator.hpp:556-574) */
    if (_number_of_copies > 0)
      check(cmi_gpu_group_reduce_accumulators(_group), "reduce_accumulators");
    totweight = 0.;
    for (int i = 0; i < 4; ++i)
      typecount[i] = 0.;
    for (Block &b : _blocks) {
      double tw = 0., tc[4];
      check(cmi_gpu_get_counters(b.engine, &tw, tc, nullptr), "get_counters");
      totweight += tw;
      for (int i = 0; i < 4; ++i)
        typecount[i] += tc[i];
    }
  }

  /* One iteration's transport in replica mode: rank r of P flies the packets
   * MPICommunicator::distribute gives it (src/MPICommunicator.hpp:207-222),
   * then the accumulators are summed over the replicas (one grouped
   * ncclAllReduce) and the counters on the host. */
  void shoot_replicated(uint_fast32_t loop, uint_fast64_t numphoton,
                        double &totweight, double typecount[4]) {
    const uint64_t P = _replicas.size();
    std::vector<uint64_t> first(P + 1, 0);
    for (uint64_t r = 0; r < P; ++r)
      first[r + 1] = first[r] + numphoton / P + (r < numphoton % P ? 1 : 0);
    for_each_engine_in_parallel(P, [&](size_t r) {
      check(cmi_gpu_reset_grid(_replicas[r]), "reset_grid");
      check(cmi_gpu_shoot(_replicas[r], (uint32_t)_random_seed, loop, first[r],
                          first[r + 1] - first[r]),
            "shoot");
    });
    totweight = 0.;
    for (int i = 0; i < 4; ++i)
      typecount[i] = 0.;
    for (cmi_gpu_engine *e : _replicas) {
      double tw = 0., tc[4];
      check(cmi_gpu_get_counters(e, &tw, tc, nullptr), "get_counters");
      totweight += tw;
      for (int i = 0; i < 4; ++i)
        typecount[i] += tc[i];
    }
    check(cmi_gpu_group_reduce_accumulators(_group), "reduce_accumulators");
  }

  void download_state() {
    DensityGrid &g = *_density_grid;
    if (decomposed()) {
      download_field(CMI_GPU_FIELD_NUMBER_DENSITY, g._number_density);
      download_field(CMI_GPU_FIELD_TEMPERATURE, g._temperature);
      for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
        download_field(CMI_GPU_FIELD_IONIC_FRACTION + ion,
                       g._ionic_fraction[ion]);
      return;
    }
    check(cmi_gpu_download_field(_engine, CMI_GPU_FIELD_NUMBER_DENSITY,
                                 g._number_density.data()),
          "download");
    check(cmi_gpu_download_field(_engine, CMI_GPU_FIELD_TEMPERATURE,
                                 g._temperature.data()),
          "download");
    for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion)
      check(cmi_gpu_download_field(_engine, CMI_GPU_FIELD_IONIC_FRACTION + ion,
                                   g._ionic_fraction[ion].data()),
            "download");
  }

  /* EmissivityCalculationSimulation (src/EmissivityCalculationSimulation.cpp:
   * 58-299: reads a snapshot back, computes the lines flagged in
   * "EmissivityValues:<name>" and appends them to the file) folded into the
   * run: the final state is still on the device, so the flagged lines are
   * computed there (cmi_gpu_compute_emissivities) and go into the last
   * snapshot as the same datasets. */
  std::vector<int32_t> _emission_lines;
  std::unique_ptr<TrackerManager> _trackers;

  template <typename F> void for_each_engine(F f) {
    if (decomposed()) {
      for (Block &b : _blocks)
        f(b.engine);
    } else if (!_replicas.empty()) {
      for (cmi_gpu_engine *e : _replicas)
        f(e);
    } else if (_engine) {
      f(_engine);
    }
  }

  void compute_emissivities() {
    DensityGrid &g = *_density_grid;
    g._emissivity.clear();
    if (_emission_lines.empty())
      return;
    const int32_t nlines = (int32_t)_emission_lines.size();
    const size_t n = (size_t)g.get_number_of_cells();
    for (int32_t line : _emission_lines)
      g._emissivity.emplace_back(emission_line_name(line),
                                 std::vector<double>(n));
    std::vector<double> values, part;
    if (!decomposed()) {
      values.resize((size_t)nlines * n);
      check(cmi_gpu_compute_emissivities(_engine, nlines,
                                         _emission_lines.data(), 0,
                                         (int64_t)n, values.data()),
            "compute_emissivities");
      for (int32_t k = 0; k < nlines; ++k)
        std::copy(values.begin() + (size_t)k * n,
                  values.begin() + (size_t)(k + 1) * n,
                  g._emissivity[k].second.begin());
      return;
    }
    for (Block &b : _blocks) {
      if (b.original >= 0)
        continue;
      const size_t nb = (size_t)b.ncell();
      values.resize((size_t)nlines * nb);
      check(cmi_gpu_compute_emissivities(b.engine, nlines,
                                         _emission_lines.data(), 0,
                                         (int64_t)nb, values.data()),
            "compute_emissivities");
      for (int32_t k = 0; k < nlines; ++k) {
        part.assign(values.begin() + (size_t)k * nb,
                    values.begin() + (size_t)(k + 1) * nb);
        block_unslice(b, part, g._emissivity[k].second);
      }
    }
  }

public:
  /* EmissivityValues::get_name, src/EmissivityValues.hpp:126-216 (the
   * dataset names; [SIII] 6312 is spelled "SIII_6213" there) */
  static const char *emission_line_name(int32_t line) {
    static const char *const names[CMI_GPU_NUMBER_OF_EMISSIONLINES] = {
        "Halpha",     "Hbeta",      "HII",        "BaLow",
        "BaHigh",     "OI_6300",    "OI_6364",    "OII_3727",
        "OIII_5007",  "OIII_4959",  "OIII_4363",  "OIII_52mu",
        "OIII_88mu",  "NII_5755",   "NII_6548",   "NII_6584",
        "NeIII_3869", "NeIII_3968", "SII_6725",   "SII_4072",
        "SIII_9405",  "SIII_6213",  "SIII_19mu",  "SIII_33mu",
        "avg_T",      "avg_T_count", "avg_nH_nHe", "avg_nH_nHe_count",
        "NeII_12mu",  "NIII_57mu",  "NeIII_15mu", "NII_122mu",
        "CII_158mu",  "CII_2325",   "CIII_1908",  "OII_7325",
        "SIV_10mu",   "HeI_5876",   "Hrec_s",     "WFC2_F439W",
        "WFC2_F555W", "WFC2_F675W"};
    return names[line];
  }

  /* IonizationSimulation ctor, src/IonizationSimulation.cpp:101-231.
   * num_thread is accepted for command-line compatibility; the device id
   * replaces it as the degree of freedom. */
  GpuIonizationSimulation(const bool write_output,
                          const bool every_iteration_output,
                          const bool output_statistics,
                          const int_fast32_t num_thread,
                          const std::string &parameterfile,
                          const int device = 0, const bool verbose = true,
                          const bool create_engine = true,
                          const std::array<int, 3> blocks = {1, 1, 1},
                          const std::vector<int> &devices = {},
                          const int copies = 0,
                          /* --task-based: the run is controlled by the
                           * parameter block of the reference's task-based
                           * driver (TaskBasedIonizationSimulation ctor,
                           * src/TaskBasedIonizationSimulation.cpp:190-260:
                           * "number of iterations" 10, "number of photons"
                           * 1e6, "random seed" 42, "source copy level" 4;
                           * no "number of photons first loop") instead of
                           * IonizationSimulation's - the engine is the same */
                          const bool task_based = false)
      : _every_iteration_output(every_iteration_output),
        _output_statistics(output_statistics), _verbose(verbose),
        _parameter_file(parameterfile),
        _number_of_iterations((uint_fast32_t)_parameter_file.get_integer(
            task_based ? "TaskBasedIonizationSimulation:number of iterations"
                       : "IonizationSimulation:number of iterations",
            10)),
        _number_of_photons((uint_fast64_t)_parameter_file.get_integer(
            task_based ? "TaskBasedIonizationSimulation:number of photons"
                       : "IonizationSimulation:number of photons",
            task_based ? 1000000 : 100000)),
        _number_of_photons_init(
            task_based
                ? _number_of_photons
                : (uint_fast64_t)_parameter_file.get_integer(
                      "IonizationSimulation:number of photons first loop",
                      (long long)_number_of_photons)),
        _random_seed(0), _abundances(_parameter_file),
        _cross_sections(generate_cross_sections(_parameter_file)),
        _recombination_rates(generate_recombination_rates(_parameter_file)),
        _density_function(generate_density_function(_parameter_file)),
        _simulation_box(_parameter_file),
        _ncell(_parameter_file.get_integer_vector("DensityGrid:number of cells",
                                                  {64, 64, 64})),
        _photon_source_distribution(
            generate_photon_source_distribution(_parameter_file)),
        _photon_source_spectrum(generate_photon_source_spectrum(
            "PhotonSourceSpectrum", _parameter_file)),
        _reemission(_parameter_file) {
    (void)num_thread;
    const std::string grid_type =
        _parameter_file.get_string("DensityGrid:type", "Cartesian");
    if (grid_type != "Cartesian")
      throw ParameterError("DensityGrid type \"" + grid_type +
                           "\" is not on this path (Cartesian only)");
    /* src/IonizationSimulation.cpp:164-176 */
    _continuous_photon_source.reset(generate_continuous_photon_source(
        _simulation_box.sides.data(), _parameter_file));
    _continuous_photon_source_spectrum.reset(generate_photon_source_spectrum(
        "ContinuousPhotonSourceSpectrum", _parameter_file,
        _continuous_photon_source ? "Monochromatic" : "None"));
    if (_continuous_photon_source && !_continuous_photon_source_spectrum)
      throw ParameterError(
          "No spectrum provided for the continuous photon sources!");
    if (_photon_source_distribution && !_photon_source_spectrum)
      throw ParameterError(
          "No spectrum provided for the discrete photon sources!");

    /* src/EmissivityCalculationSimulation.cpp:70-74 */
    for (int32_t line = 0; line < CMI_GPU_NUMBER_OF_EMISSIONLINES; ++line)
      if (_parameter_file.get_bool(
              std::string("EmissivityValues:") + emission_line_name(line),
              false))
        _emission_lines.push_back(line);

    const std::string output_folder = _parameter_file.get_string(
        "IonizationSimulation:output folder", ".");
    if (write_output)
      _density_grid_writer.reset(generate_writer(output_folder, _parameter_file));

    /* TemperatureCalculator parameters, src/TemperatureCalculator.cpp:133-160 */
    cmi_gpu_temperature_params &t = _temperature_params;
    t.do_temperature_calculation = _parameter_file.get_bool(
        "TemperatureCalculator:do temperature calculation", false);
    t.minimum_number_of_iterations = (int32_t)_parameter_file.get_integer(
        "TemperatureCalculator:minimum number of iterations", 3);
    t.epsilon_convergence = _parameter_file.get_double(
        "TemperatureCalculator:epsilon convergence", 1.e-3);
    t.maximum_number_of_iterations = (int32_t)_parameter_file.get_integer(
        "TemperatureCalculator:maximum number of iterations", 100);
    t.pah_heating_factor =
        _parameter_file.get_double("TemperatureCalculator:PAH heating factor", 0.);
    t.cosmic_ray_heating_factor = _parameter_file.get_double(
        "TemperatureCalculator:cosmic ray heating factor", 0.);
    t.cosmic_ray_heating_limit = _parameter_file.get_double(
        "TemperatureCalculator:cosmic ray heating limit", 0.75);
    t.cosmic_ray_heating_scale_length = _parameter_file.get_physical_value(
        QUANTITY_LENGTH, "TemperatureCalculator:cosmic ray heating scale length",
        "1.33333 kpc");
    t.minimum_ionized_temperature = _parameter_file.get_physical_value(
        QUANTITY_TEMPERATURE, "TemperatureCalculator:minimum ionized temperature",
        "4000. K");

    _random_seed = (int32_t)_parameter_file.get_integer(
        task_based ? "TaskBasedIonizationSimulation:random seed"
                   : "IonizationSimulation:random seed",
        42);
    /* src/IonizationSimulation.cpp:208-213 */
    if (_parameter_file.get_bool("IonizationSimulation:enable trackers", false))
      _trackers.reset(new TrackerManager(_parameter_file));

    /* all parameters read: dump them (src/IonizationSimulation.cpp:218-226) */
    if (write_output) {
      std::ofstream pfile(parameterfile + ".used-values");
      _parameter_file.print_contents(pfile);
      status("Wrote used parameters to " + parameterfile + ".used-values.");
    }

    _density_grid.reset(new DensityGrid(_simulation_box, _ncell));

    if (create_engine) {
      cmi_gpu_config config = {};
      for (int a = 0; a < 3; ++a) {
        config.anchor[a] = _simulation_box.anchor[a];
        config.sides[a] = _simulation_box.sides[a];
        config.ncell[a] = (int32_t)_ncell[a];
        config.periodic[a] = _simulation_box.periodicity[a] ? 1 : 0;
      }
      config.device = device;
      config.track_heating = t.do_temperature_calculation;
      config.stream = nullptr;
      config.external_accumulators = nullptr;
      _nblock = blocks;
      if (blocks[0] * blocks[1] * blocks[2] == 1 && devices.size() > 1) {
        /* replicas: the whole grid on every device */
        for (int d : devices) {
          config.device = d;
          cmi_gpu_engine *e = nullptr;
          check(cmi_gpu_create(&config, &e), "cmi_gpu_create");
          _replicas.push_back(e);
          lower_model(e);
        }
        _engine = _replicas[0];
        check(cmi_gpu_group_create((int32_t)_replicas.size(), _replicas.data(),
                                   &_group),
              "group_create");
        status("Replica mode: " + std::to_string(_replicas.size()) +
               " devices, accumulators reduced over RCCL.");
      } else if (blocks[0] * blocks[1] * blocks[2] == 1) {
        check(cmi_gpu_create(&config, &_engine), "cmi_gpu_create");
        lower_model(_engine);
      } else {
        /* blocks as even as possible; block k runs on devices[k % n] */
        for (int a = 0; a < 3; ++a) {
          const long long q = _ncell[a] / blocks[a], r = _ncell[a] % blocks[a];
          _block_edges[a].push_back(0);
          for (int i = 0; i < blocks[a]; ++i)
            _block_edges[a].push_back(_block_edges[a].back() + (int32_t)q +
                                      (i < r ? 1 : 0));
        }
        size_t k = 0;
        for (int bx = 0; bx < blocks[0]; ++bx)
          for (int by = 0; by < blocks[1]; ++by)
            for (int bz = 0; bz < blocks[2]; ++bz, ++k) {
              Block b;
              const int idx[3] = {bx, by, bz};
              for (int a = 0; a < 3; ++a) {
                b.offset[a] = _block_edges[a][idx[a]];
                b.size[a] = _block_edges[a][idx[a] + 1] - b.offset[a];
                config.sub_offset[a] = b.offset[a];
                config.sub_ncell[a] = b.size[a];
              }
              b.device = devices.empty() ? device
                                         : devices[k % devices.size()];
              config.device = b.device;
              check(cmi_gpu_create(&config, &b.engine), "cmi_gpu_create");
              _blocks.push_back(b);
              lower_model(b.engine);
              /* room for every packet of an iteration to leave the block */
              check(cmi_gpu_set_export_buffer(
                        b.engine, nullptr,
                        std::max(_number_of_photons, _number_of_photons_init) +
                            1024),
                    "set_export_buffer");
            }
        /* Copies of the blocks that contain a source: the packets of a source
         * all start in its block, so without copies one device flies the
         * first flight of every packet while the others wait. The reference
         * makes 2^level copies of such a subgrid
         * (TaskBasedIonizationSimulation:source copy level, default 4,
         * src/TaskBasedIonizationSimulation.cpp:199,514-560); here a block
         * is 1/P of the grid, so: one copy per device at most, 2^level at
         * most. --copies K asks for exactly K engines per source block (also
         * on one device: tests). The neighbours of a block with copies get
         * half as many, as in the reference (below). */
        {
          const int level = (int)_parameter_file.get_integer(
              "TaskBasedIonizationSimulation:source copy level", 4);
          std::vector<int> pool = devices;
          if (pool.empty())
            pool.push_back(device);
          size_t want = copies > 0
                            ? (size_t)copies
                            : std::min((size_t)1 << std::min(level, 6),
                                       pool.size());
          const size_t originals = _blocks.size();
          std::vector<char> has_source(originals, 0);
          if (_photon_source_distribution && want > 1) {
            const photonsourcenumber_t ns =
                _photon_source_distribution->get_number_of_sources();
            for (photonsourcenumber_t i = 0; i < ns; ++i) {
              const CoordinateVector pos =
                  _photon_source_distribution->get_position(i);
              int64_t c[3];
              bool inside = true;
              for (int a = 0; a < 3; ++a) {
                c[a] = (int64_t)std::floor((pos[a] - config.anchor[a]) /
                                           config.sides[a] * _ncell[a]);
                inside &= c[a] >= 0 && c[a] < _ncell[a];
              }
              if (inside)
                has_source[owner_of_cell((c[0] * _ncell[1] + c[1]) *
                                             _ncell[2] +
                                         c[2])] = 1;
            }
          }
          /* engines per block: `want` for a block with a source; the blocks
           * next to a block with 2^l engines get 2^(l-1) at least, and so on
           * outwards (the reference's restriction of the copy levels,
           * src/TaskBasedIonizationSimulation.cpp:533-556: a packet that
           * leaves a block with many copies must not find all of them queueing
           * for one neighbour) - capped by the device pool like `want`.
           * `rings_dropped`: the cascade stops that many rings early. */
          auto plan = [&](size_t want_now, int rings_dropped) {
            std::vector<size_t> engines_of(originals, 1);
            int top_level = 0;
            for (size_t o = 0; o < originals; ++o)
              if (has_source[o]) {
                engines_of[o] = want_now;
                while (((size_t)2 << top_level) <= want_now)
                  ++top_level;
              }
            for (int level = top_level; level > 1 + rings_dropped; --level) {
              const size_t here = (size_t)1 << level;
              for (size_t o = 0; o < originals; ++o) {
                if (engines_of[o] < here || engines_of[o] >= 2 * here)
                  continue;
                const int bx = (int)(o / ((size_t)_nblock[1] * _nblock[2]));
                const int by = (int)((o / _nblock[2]) % _nblock[1]);
                const int bz = (int)(o % _nblock[2]);
                const int at[3] = {bx, by, bz};
                for (int axis = 0; axis < 3; ++axis)
                  for (int side = -1; side <= 1; side += 2) {
                    int nb[3] = {at[0], at[1], at[2]};
                    nb[axis] += side;
                    if (nb[axis] < 0 || nb[axis] >= _nblock[axis]) {
                      if (!config.periodic[axis] || _nblock[axis] < 2)
                        continue;
                      nb[axis] = (nb[axis] + _nblock[axis]) % _nblock[axis];
                    }
                    const size_t n =
                        ((size_t)nb[0] * _nblock[1] + nb[1]) * _nblock[2] +
                        nb[2];
                    engines_of[n] = std::max(engines_of[n], here / 2);
                  }
              }
            }
            return engines_of;
          };
          /* A group holds at most 64 engines (and every copy is a whole block
           * engine in device memory). The reference has no such limit, so a
           * cascade that does not fit is cut back instead of refused: first
           * the outermost rings of neighbour copies, one at a time, then the
           * copies of the source blocks themselves are halved - the status
           * line says what was granted. */
          const size_t group_limit = 64;
          if (originals > group_limit)
            throw std::runtime_error(
                "too many blocks (at most 64 engines in a group)");
          size_t granted = want;
          int rings_dropped = 0;
          std::vector<size_t> engines_of;
          for (;;) {
            engines_of = plan(granted, rings_dropped);
            size_t total = 0;
            int rings = 0;
            for (size_t n : engines_of)
              total += n;
            while (((size_t)4 << rings) <= granted)
              ++rings; /* rings of neighbour copies a cascade from `granted`
                          has: levels top ... 2 */
            if (total <= group_limit)
              break;
            if (rings_dropped < rings) {
              ++rings_dropped;
            } else if (granted > 1) {
              size_t half = 1;
              while (2 * half < granted)
                half *= 2;
              granted = half; /* the next power of two below */
              rings_dropped = 0;
            } else {
              break; /* (originals <= limit: cannot happen) */
            }
          }
          if (granted != want || rings_dropped)
            status("Copies: " + std::to_string(want) +
                   " engines per source block asked for, " +
                   std::to_string(granted) + " granted" +
                   (rings_dropped
                        ? ", the outermost " + std::to_string(rings_dropped) +
                              " ring(s) of neighbour copies dropped"
                        : std::string()) +
                   " (at most 64 engines in a group).");
          for (size_t o = 0; o < originals; ++o) {
            const size_t want_here = engines_of[o];
            if (want_here < 2)
              continue;
            /* on the devices after the original's, round-robin */
            size_t at = 0;
            while (at < pool.size() && pool[at] != _blocks[o].device)
              ++at;
            for (size_t k = 1; k < want_here; ++k) {
              Block b = _blocks[o];
              b.original = (int)o;
              b.device = pool[(at + k) % pool.size()];
              for (int a = 0; a < 3; ++a) {
                config.sub_offset[a] = b.offset[a];
                config.sub_ncell[a] = b.size[a];
              }
              config.device = b.device;
              check(cmi_gpu_create(&config, &b.engine), "cmi_gpu_create");
              _blocks.push_back(b);
              lower_model(b.engine);
              check(cmi_gpu_set_export_buffer(
                        b.engine, nullptr,
                        std::max(_number_of_photons, _number_of_photons_init) +
                            1024),
                    "set_export_buffer");
              ++_number_of_copies;
              if (!has_source[o])
                ++_number_of_neighbour_copies;
            }
          }
        }
        {
          std::vector<cmi_gpu_engine *> engines;
          for (Block &b : _blocks)
            engines.push_back(b.engine);
          check(cmi_gpu_group_create((int32_t)engines.size(), engines.data(),
                                     &_group),
                "group_create");
        }
        status("Domain decomposition: " +
               std::to_string(_blocks.size() - _number_of_copies) +
               " blocks and " + std::to_string(_number_of_copies) +
               " copies of source blocks" +
               (_number_of_neighbour_copies
                    ? " (" + std::to_string(_number_of_neighbour_copies) +
                          " of them of their neighbours)"
                    : std::string()) +
               ", flights handed over device to device.");
      }
    }
  }

  ~GpuIonizationSimulation() {
    if (_group)
      cmi_gpu_group_destroy(_group);
    if (!_replicas.empty()) {
      for (cmi_gpu_engine *e : _replicas)
        cmi_gpu_destroy(e);
    } else if (_engine) {
      cmi_gpu_destroy(_engine);
    }
    for (Block &b : _blocks)
      cmi_gpu_destroy(b.engine);
  }
  size_t number_of_copies() const { return _number_of_copies; }
  uint64_t exchange_rounds() const { return _exchange_rounds; }
  uint64_t flights_exchanged() const { return _flights_exchanged; }

  ParameterFile &parameter_file() { return _parameter_file; }
  DensityGrid &grid() { return *_density_grid; }
  uint_fast32_t number_of_iterations() const { return _number_of_iterations; }
  uint_fast64_t number_of_photons() const { return _number_of_photons; }
  const cmi_gpu_temperature_params &temperature_params() const {
    return _temperature_params;
  }
  const DiffuseReemission &reemission() const { return _reemission; }
  const Abundances &abundances() const { return _abundances; }
  PhotonSourceDistribution *sources() { return _photon_source_distribution.get(); }
  PhotonSourceSpectrum *spectrum() { return _photon_source_spectrum.get(); }
  CrossSections *cross_sections() { return _cross_sections.get(); }
  RecombinationRates *recombination_rates() { return _recombination_rates.get(); }
  int32_t random_seed() const { return _random_seed; }

  /* IonizationSimulation::initialize, src/IonizationSimulation.cpp:239-325:
   * evaluate the DensityFunction on every cell (host), upload SoA arrays */
  void initialize(DensityFunction *density_function = nullptr) {
    if (density_function == nullptr)
      density_function = _density_function.get();
    status("Initializing DensityFunction...");
    density_function->initialize();
    status("Done.");
    DensityGrid &g = *_density_grid;
    const int64_t n = g.get_number_of_cells();
    std::vector<double> x((size_t)NUMBER_OF_IONNAMES * n);
    /* DensityGridInitializationFunction, src/DensityGrid.hpp:775-790 */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      DensityGrid::iterator cell(&g, i);
      const DensityValues vals = (*density_function)(cell);
      g._number_density[i] = vals.get_number_density();
      g._temperature[i] = vals.get_temperature();
      for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion) {
        g._ionic_fraction[ion][i] = vals.get_ionic_fraction(ion);
        x[(size_t)ion * n + i] = vals.get_ionic_fraction(ion);
      }
    }
    density_function->free();
    if (!_replicas.empty()) {
      for (cmi_gpu_engine *e : _replicas)
        check(cmi_gpu_upload_cells(e, g._number_density.data(),
                                   g._temperature.data(), x.data()),
              "cmi_gpu_upload_cells");
    } else if (_engine) {
      check(cmi_gpu_upload_cells(_engine, g._number_density.data(),
                                 g._temperature.data(), x.data()),
            "cmi_gpu_upload_cells");
    }
    for (Block &b : _blocks) {
      std::vector<double> dens, temp, part, xb;
      block_slice(b, g._number_density, dens);
      block_slice(b, g._temperature, temp);
      for (int ion = 0; ion < NUMBER_OF_IONNAMES; ++ion) {
        block_slice(b, g._ionic_fraction[ion], part);
        xb.insert(xb.end(), part.begin(), part.end());
      }
      check(cmi_gpu_upload_cells(b.engine, dens.data(), temp.data(),
                                 xb.data()),
            "cmi_gpu_upload_cells");
    }
  }

  /* the initial snapshot (src/IonizationSimulation.cpp:347-350) without an
   * engine: what a dry run can show of the writer and the density function */
  void write_initial_snapshot() {
    if (_density_grid_writer)
      _density_grid_writer->write(*_density_grid, 0, _parameter_file);
  }

  /* IonizationSimulation::run, src/IonizationSimulation.cpp:334-680 */
  void run(DensityGridWriter *density_grid_writer = nullptr) {
    if (!_engine && !decomposed())
      throw std::runtime_error("run() needs an engine (not a dry run)");
    if (_density_grid_writer)
      _density_grid_writer->write(*_density_grid, 0, _parameter_file);

    uint_fast32_t loop = 0;
    while (loop < _number_of_iterations) {
      status("Starting loop " + std::to_string(loop) + ".");
      uint_fast64_t lnumphoton = _number_of_photons;
      if (loop == 0)
        lnumphoton = _number_of_photons_init;
      /* src/IonizationSimulation.cpp:367-370 */
      if (_trackers && loop == _number_of_iterations - 1) {
        _trackers->normalize_for_cell(_density_grid->get_cell_volume());
        for_each_engine([&](cmi_gpu_engine *e) {
          check(_trackers->lower(e), "set_spectrum_trackers");
        });
        lnumphoton = std::max(lnumphoton, _trackers->get_number_of_photons());
      }

      status("Start shooting " + std::to_string(lnumphoton) + " photons...");
      auto t0 = std::chrono::steady_clock::now();
      double totweight = 0.;
      double typecount[4] = {0., 0., 0., 0.};
      if (decomposed()) {
        shoot_decomposed(loop, lnumphoton, totweight, typecount);
      } else if (!_replicas.empty()) {
        shoot_replicated(loop, lnumphoton, totweight, typecount);
      } else {
        check(cmi_gpu_reset_grid(_engine), "reset_grid");
        check(cmi_gpu_shoot(_engine, (uint32_t)_random_seed, loop, 0,
                            lnumphoton),
              "shoot");
        check(cmi_gpu_get_counters(_engine, &totweight, typecount, nullptr),
              "get_counters");
      }
      auto t1 = std::chrono::steady_clock::now();
      _shoot_seconds += std::chrono::duration<double>(t1 - t0).count();
      _last_totweight = totweight;
      for (int i = 0; i < 4; ++i)
        _last_typecount[i] = typecount[i];
      status("Done shooting photons.");
      if (_output_statistics) {
        /* src/IonizationSimulation.cpp:418-447 */
        std::ostringstream s;
        s << 100. * typecount[3] / totweight
          << "% of photons were reemitted as non-ionizing photons.\n"
          << 100. * (typecount[1] + typecount[2]) / totweight
          << "% of photons were scattered.\n"
          << "Escape fraction: "
          << std::max(0., 100. * (totweight - typecount[3]) / totweight)
          << "%.\n"
          << "Diffuse HI escape fraction: " << 100. * typecount[1] / totweight
          << "%.\n"
          << "Diffuse HeI escape fraction: " << 100. * typecount[2] / totweight
          << "%.";
        status(s.str());
      }

      status("Calculating ionization state after shooting " +
             std::to_string(lnumphoton) + " photons...");
      t0 = std::chrono::steady_clock::now();
      if (_group) {
        /* replicas, or the copies of a block: member r of the engines that
         * hold the same cells solves slab r, the slabs are gathered into all
         * of them (src/IonizationSimulation.cpp:532-618); a block without
         * copies updates its own cells */
        check(cmi_gpu_group_update_cells(_group, loop, totweight),
              "group_update_cells");
        for (Block &b : _blocks)
          check(cmi_gpu_synchronize(b.engine), "synchronize");
        for (cmi_gpu_engine *e : _replicas)
          check(cmi_gpu_synchronize(e), "synchronize");
      } else {
        check(cmi_gpu_update_cells(_engine, loop, totweight), "update_cells");
        check(cmi_gpu_synchronize(_engine), "synchronize");
      }
      t1 = std::chrono::steady_clock::now();
      _update_seconds += std::chrono::duration<double>(t1 - t0).count();
      status("Done calculating ionization state.");

      ++loop;
      if (_density_grid_writer && _every_iteration_output &&
          loop < _number_of_iterations) {
        download_state();
        _density_grid_writer->write(*_density_grid, loop, _parameter_file);
      }
    }
    if (loop == _number_of_iterations)
      status("Maximum number of iterations (" +
             std::to_string(_number_of_iterations) + ") reached, stopping.");

    /* src/IonizationSimulation.cpp:651-653 */
    if (_trackers && _number_of_iterations > 0) {
      for_each_engine([&](cmi_gpu_engine *e) {
        check(_trackers->collect(e), "get_tracker_counts");
        check(cmi_gpu_enable_trackers(e, 0), "enable_trackers");
      });
      /* src/IonizationSimulation.cpp:621-624 */
      _trackers->normalize(total_luminosity() / _last_totweight);
      _trackers->output_trackers();
    }
    download_state();
    compute_emissivities();
    if (_density_grid_writer)
      _density_grid_writer->write(*_density_grid, _number_of_iterations,
                                  _parameter_file);
    if (density_grid_writer)
      density_grid_writer->write(*_density_grid, _number_of_iterations,
                                 _parameter_file);
    /* the reference's own packets/s instrument,
     * src/IonizationSimulation.cpp:667-674 */
    std::ostringstream s;
    s << "Total photon shooting time: " << _shoot_seconds << " s.\n"
      << "Total cell update time: " << _update_seconds << " s.";
    if (decomposed())
      s << "\nFlights handed over between blocks: " << _flights_exchanged
        << " in " << _exchange_rounds << " rounds.";
    status(s.str());
  }

  double shoot_seconds() const { return _shoot_seconds; }
  double update_seconds() const { return _update_seconds; }
};

} // namespace cmi

#endif
