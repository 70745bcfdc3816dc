/*
 * SphSnapshots.hpp - DensityFunctions that read the binary dumps of SPH
 * codes (Fortran unformatted files) and map the particles onto the cells:
 *
 *   PhantomSnapshotDensityFunction  src/PhantomSnapshotDensityFunction.{hpp,cpp}
 *   SPHNGSnapshotDensityFunction    src/SPHNGSnapshotDensityFunction.{hpp,cpp},
 *                                   src/SPHNGSnapshotUtilities.hpp
 *
 * Host side only (§8 row f3: the data formats either side of the hot path).
 * Included at the end of Plugins.hpp, whose factory hands the types here to
 * generate_sph_snapshot_density_function().
 */
#ifndef CMI_HOST_SPH_SNAPSHOTS_HPP
#define CMI_HOST_SPH_SNAPSHOTS_HPP

#include "PetkovaMapping.hpp"

#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace cmi {

/* A Fortran unformatted sequential file: every record is its length (4
 * bytes), the bytes, and the length again
 * (PhantomSnapshotDensityFunction::skip_block / read_block,
 * src/PhantomSnapshotDensityFunction.hpp:89-181). */
class FortranRecords {
  std::ifstream _file;
  long long _size = 0;

public:
  explicit FortranRecords(const std::string &filename)
      : _file(filename, std::ios::binary | std::ios::in) {
    if (!_file)
      throw ParameterError("Unable to open file \"" + filename + "\"!");
    _file.seekg(0, std::ios_base::end);
    _size = (long long)_file.tellg();
    _file.seekg(0, std::ios_base::beg);
  }
  /* the next record; if `expected` is given its size must be that */
  std::vector<uint8_t> read(long long expected = -1) {
    uint32_t length1 = 0, length2 = 0;
    _file.read(reinterpret_cast<char *>(&length1), 4);
    if (!_file)
      throw ParameterError("unexpected end of a Fortran unformatted file");
    /* (a damaged length must not become an allocation) */
    if ((long long)_file.tellg() + (long long)length1 + 4 > _size)
      throw ParameterError("Wrong block size!");
    if (expected >= 0 && (long long)length1 != expected)
      throw ParameterError(
          "Wrong number of variables passed on to read_block()! Block size "
          "is " + std::to_string(length1) + ", but size of variables is " +
          std::to_string(expected) + ".");
    std::vector<uint8_t> data(length1);
    _file.read(reinterpret_cast<char *>(data.data()), length1);
    _file.read(reinterpret_cast<char *>(&length2), 4);
    if (!_file || length1 != length2)
      throw ParameterError("Wrong block size!");
    return data;
  }
  void skip() {
    uint32_t length1 = 0, length2 = 0;
    _file.read(reinterpret_cast<char *>(&length1), 4);
    if (!_file || (long long)_file.tellg() + (long long)length1 + 4 > _size)
      throw ParameterError("Wrong block size!");
    _file.seekg(length1, std::ios_base::cur);
    _file.read(reinterpret_cast<char *>(&length2), 4);
    if (!_file || length1 != length2)
      throw ParameterError("Wrong block size!");
  }
  template <typename T> T value() {
    const std::vector<uint8_t> data = read(sizeof(T));
    T v;
    std::memcpy(&v, data.data(), sizeof(T));
    return v;
  }
  template <typename T> std::vector<T> values(size_t n) {
    const std::vector<uint8_t> data = read((long long)(n * sizeof(T)));
    std::vector<T> v(n);
    if (n)
      std::memcpy(v.data(), data.data(), n * sizeof(T));
    return v;
  }
  /* a string without its trailing blanks */
  static std::string stripped(const uint8_t *chars, size_t n) {
    while (n > 0 && chars[n - 1] == ' ')
      --n;
    return std::string(reinterpret_cast<const char *>(chars), n);
  }
  std::string string() {
    const std::vector<uint8_t> data = read();
    return stripped(data.data(), data.size());
  }
  /* tags: 16 characters each */
  std::vector<std::string> tags(size_t n) {
    const std::vector<uint8_t> data = read();
    if (data.size() % 16 != 0)
      throw ParameterError(
          "Block has the wrong size to contain a list of tags!");
    if (n * 16 != data.size())
      throw ParameterError("Vector of wrong size given!");
    std::vector<std::string> out(n);
    for (size_t i = 0; i < n; ++i)
      out[i] = stripped(data.data() + 16 * i, 16);
    return out;
  }
};

/* A set of SPH particles with the search the mappings need: the particles
 * whose kernel (of radius reach[i]) covers a point, or comes within a margin
 * of it - what the reference asks its Octree (get_ngbs, get_ngbs_sphere,
 * src/Octree.hpp:128-211). A uniform grid of bins with the largest reach as
 * their side gives the same sets. */
class SphParticleBins {
  const std::vector<double> &_positions; /* [n][3] */
  const std::vector<double> &_reach;
  std::array<double, 3> _anchor = {0., 0., 0.}, _side = {1., 1., 1.};
  std::array<int, 3> _nbin = {1, 1, 1};
  std::vector<uint32_t> _start, _particles;

  int bin_of(double x, int a) const {
    const int i = (int)std::floor((x - _anchor[a]) / _side[a]);
    return i < 0 ? 0 : (i >= _nbin[a] ? _nbin[a] - 1 : i);
  }

public:
  SphParticleBins(const std::vector<double> &positions,
                  const std::vector<double> &reach)
      : _positions(positions), _reach(reach) {
    const size_t n = reach.size();
    double longest = 0.;
    std::array<double, 3> lo = {DBL_MAX, DBL_MAX, DBL_MAX},
                          hi = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (size_t i = 0; i < n; ++i) {
      longest = std::max(longest, reach[i]);
      for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(lo[a], positions[3 * i + a]);
        hi[a] = std::max(hi[a], positions[3 * i + a]);
      }
    }
    size_t total = 1;
    for (int a = 0; a < 3; ++a) {
      const double extent = n ? hi[a] - lo[a] : 0.;
      _anchor[a] = n ? lo[a] : 0.;
      int nb = longest > 0. ? (int)std::floor(extent / longest) : 1;
      nb = std::max(1, std::min(nb, 256));
      _nbin[a] = nb;
      _side[a] = extent > 0. ? extent / nb : 1.;
      total *= (size_t)nb;
    }
    std::vector<uint32_t> count(total + 1, 0);
    std::vector<size_t> bin(n);
    for (size_t i = 0; i < n; ++i) {
      bin[i] = ((size_t)bin_of(positions[3 * i], 0) * _nbin[1] +
                bin_of(positions[3 * i + 1], 1)) *
                   _nbin[2] +
               bin_of(positions[3 * i + 2], 2);
      ++count[bin[i] + 1];
    }
    for (size_t b = 0; b < total; ++b)
      count[b + 1] += count[b];
    _start = count;
    _particles.resize(n);
    std::vector<uint32_t> cursor(count.begin(), count.end() - 1);
    for (size_t i = 0; i < n; ++i)
      _particles[cursor[bin[i]]++] = (uint32_t)i;
  }
  /* f(index, distance) for every particle with distance <= reach + margin */
  template <typename F>
  void for_neighbours_within(const CoordinateVector &p, double margin,
                             F f) const {
    int from[3], to[3];
    for (int a = 0; a < 3; ++a) {
      const int spread = 1 + (margin > 0. ? (int)std::ceil(margin / _side[a])
                                          : 0);
      /* (a point outside the particles' box: measured from the nearest
       * bin) */
      const int c = bin_of(p[a], a);
      const double beyond =
          std::max(_anchor[a] - p[a],
                   p[a] - (_anchor[a] + _nbin[a] * _side[a]));
      const int extra = beyond > 0. ? (int)std::ceil(beyond / _side[a]) : 0;
      from[a] = std::max(0, c - spread - extra);
      to[a] = std::min(_nbin[a] - 1, c + spread + extra);
    }
    for (int ix = from[0]; ix <= to[0]; ++ix)
      for (int iy = from[1]; iy <= to[1]; ++iy)
        for (int iz = from[2]; iz <= to[2]; ++iz) {
          const size_t bin = ((size_t)ix * _nbin[1] + iy) * _nbin[2] + iz;
          for (uint32_t k = _start[bin]; k < _start[bin + 1]; ++k) {
            const size_t index = _particles[k];
            const double d[3] = {p[0] - _positions[3 * index],
                                 p[1] - _positions[3 * index + 1],
                                 p[2] - _positions[3 * index + 2]};
            const double r =
                std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            if (r <= _reach[index] + margin)
              f(index, r);
          }
        }
  }
};

/* What the Phantom and SPHNG density functions share
 * (src/PhantomSnapshotDensityFunction.cpp:669-832,
 * src/SPHNGSnapshotDensityFunction.cpp:569-580,994-1085): particles with
 * position, mass and smoothing length h, mapped onto the cells either by the
 * kernel at the cell's midpoint (Price 2007, support 2 h) or by the kernel's
 * integral over the cell (Petkova et al. 2018: `use new algorithm`, with the
 * closed form evaluated directly as the reference does in these two classes).
 * Hydrogen only, at `initial temperature`, neutral fraction 1e-6. */
class SphKernelDensityFunction : public DensityFunction {
protected:
  std::vector<double> _positions; /* [n][3], m */
  std::vector<double> _smoothing_lengths, _masses;

private:
  std::vector<double> _reach;
  const double _initial_temperature;
  const bool _use_new_algorithm;
  /* (not in the reference: the faces of the cell oriented towards its
   * midpoint, which makes the sum of the vertex integrals the integral over
   * the cell on a Cartesian grid - INTEGRATION.md 2b, "Petkova_oriented") */
  const bool _oriented_faces;
  std::unique_ptr<SphParticleBins> _bins;
  std::unique_ptr<PetkovaMapping> _petkova;

  /* PhantomSnapshotDensityFunction::kernel, :48-64 (= SPHNG's) */
  static double kernel(double q, double h) {
    if (q < 1.) {
      const double q2 = q * q, h2 = h * h, h3 = h2 * h;
      return (1. - 1.5 * q2 + 0.75 * q2 * q) / (M_PI * h3);
    }
    if (q < 2.) {
      const double c = 2. - q, c2 = c * c, h2 = h * h, h3 = h * h2;
      return 0.25 * c2 * c / (M_PI * h3);
    }
    return 0.;
  }

protected:
  SphKernelDensityFunction(double initial_temperature, bool use_new_algorithm,
                           bool oriented_faces)
      : _initial_temperature(initial_temperature),
        _use_new_algorithm(use_new_algorithm),
        _oriented_faces(oriented_faces) {}
  void add_particle(double x, double y, double z, double mass, double h) {
    _positions.push_back(x);
    _positions.push_back(y);
    _positions.push_back(z);
    _masses.push_back(mass);
    _smoothing_lengths.push_back(h);
  }

public:
  void initialize() override {
    _reach.resize(_smoothing_lengths.size());
    for (size_t i = 0; i < _reach.size(); ++i)
      _reach[i] = 2. * _smoothing_lengths[i];
    _bins.reset(new SphParticleBins(_positions, _reach));
    if (_use_new_algorithm)
      _petkova.reset(new PetkovaMapping(false)); /* closed form only */
  }
  void free() override {
    _bins.reset();
    _petkova.reset();
  }

  size_t get_number_of_particles() const { return _masses.size(); }
  CoordinateVector get_position(size_t index) const {
    return CoordinateVector(_positions[3 * index], _positions[3 * index + 1],
                            _positions[3 * index + 2]);
  }
  double get_mass(size_t index) const { return _masses[index]; }
  double get_smoothing_length(size_t index) const {
    return _smoothing_lengths[index];
  }

  DensityValues operator()(const Cell &cell) override {
    const CoordinateVector position = cell.get_cell_midpoint();
    double density = 0.;
    if (_use_new_algorithm) {
      const std::vector<Face> faces = cell.get_faces();
      if (faces.empty())
        throw ParameterError(
            "the Petkova mapping needs the faces of the grid's cells");
      /* (the reference takes the norm of the vertices' POSITIONS as the
       * radius of the search: at least the distance of the furthest vertex
       * from the midpoint for a box around the origin, more elsewhere. Kept:
       * a larger radius only adds particles whose integral over the cell is
       * zero.) */
      double radius = 0.;
      for (const Face &face : faces)
        for (const CoordinateVector &v : face.vertices)
          radius = std::max(
              radius, std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
      const double midpoint[3] = {position[0], position[1], position[2]};
      _bins->for_neighbours_within(position, radius, [&](size_t i, double) {
        density += _petkova->mass_fraction(
                       faces, &_positions[3 * i], _smoothing_lengths[i],
                       _oriented_faces ? midpoint : nullptr, false) *
                   _masses[i];
      });
      density /= cell.get_volume();
    } else {
      _bins->for_neighbours_within(position, 0., [&](size_t i, double r) {
        density += _masses[i] *
                   kernel(r / _smoothing_lengths[i], _smoothing_lengths[i]);
      });
    }
    DensityValues values;
    values.set_number_density(density / 1.6737236e-27);
    values.set_temperature(_initial_temperature);
    values.set_ionic_fraction(ION_H_n, 1.e-6);
    values.set_ionic_fraction(ION_He_n, 1.e-6);
    return values;
  }
};

/* read_dict of both readers (src/PhantomSnapshotDensityFunction.hpp:196-237,
 * src/SPHNGSnapshotUtilities.hpp:152-183): a count, the tags (absent from an
 * untagged file: every tag is "tag"), the values; duplicate tags get a
 * counter appended */
template <typename T>
inline std::map<std::string, T> read_fortran_dict(FortranRecords &file,
                                                  bool tagged = true,
                                                  bool tags_of_empty = false) {
  const uint32_t size = file.value<uint32_t>();
  std::map<std::string, T> dict;
  if (size == 0 && !tags_of_empty)
    return dict;
  std::vector<std::string> tags =
      tagged ? file.tags(size) : std::vector<std::string>(size, "tag");
  const std::vector<T> vals = file.values<T>(size);
  for (uint32_t i = 0; i < size; ++i) {
    if (dict.count(tags[i]) == 1) {
      unsigned count = 1;
      while (dict.count(tags[i] + std::to_string(count)) == 1)
        ++count;
      tags[i] += std::to_string(count);
    }
    dict[tags[i]] = vals[i];
  }
  return dict;
}

/* src/PhantomSnapshotDensityFunction.cpp:395-668: the gas particles of a
 * Phantom dump (tagged format "FT...": header dictionaries per number type,
 * then the arrays of one block by tag - x, y, z in 8 bytes, h in 4), all of
 * the same mass `massoftype` x `umass`, lengths in `udist` (CGS in the
 * file). */
class PhantomSnapshotDensityFunction : public SphKernelDensityFunction {
public:
  PhantomSnapshotDensityFunction(const std::string &filename,
                                 double initial_temperature,
                                 bool use_new_algorithm, bool use_periodic_box,
                                 bool oriented_faces = false)
      : SphKernelDensityFunction(initial_temperature, use_new_algorithm,
                                 oriented_faces) {
    if (use_periodic_box)
      throw ParameterError(
          "PhantomSnapshot: `use periodic box` is not on this path");
    FortranRecords file(filename);
    file.skip(); /* "it contains garbage" */
    const std::string fileident = file.string();
    if (fileident.empty() || fileident[0] != 'F')
      throw ParameterError("Unsupported Phantom snapshot format: " +
                           fileident + "!");
    if (fileident.size() < 2 || fileident[1] != 'T')
      throw ParameterError("untagged Phantom dumps are not supported (nor by "
                           "the reference)");
    std::map<std::string, int32_t> ints = read_fortran_dict<int32_t>(file);
    read_fortran_dict<int8_t>(file);
    read_fortran_dict<int16_t>(file);
    read_fortran_dict<int32_t>(file);
    std::map<std::string, int64_t> int64s = read_fortran_dict<int64_t>(file);
    std::map<std::string, double> reals = read_fortran_dict<double>(file);
    read_fortran_dict<float>(file);
    std::map<std::string, double> real8s = read_fortran_dict<double>(file);
    if (ints["nblocks"] != 1)
      throw ParameterError("Phantom dumps in several blocks are not supported "
                           "(nor by the reference)");
    const int32_t narraylengths = file.value<int32_t>();
    if (narraylengths < 2 || narraylengths > 3)
      throw ParameterError("unexpected number of array lengths in the Phantom "
                           "dump");
    const size_t numpart = (size_t)int64s["npartoftype"];
    std::vector<std::array<int32_t, 8>> varnums((size_t)narraylengths);
    for (auto &v : varnums) {
      /* the number of entries (8 bytes) and the numbers of arrays of each of
       * the 8 types */
      const std::vector<uint8_t> block = file.read(8 + 8 * 4);
      std::memcpy(v.data(), block.data() + 8, 32);
    }
    std::vector<double> x, y, z;
    std::vector<float> h;
    for (const auto &v : varnums)
      for (int idata = 0; idata < 8; ++idata)
        for (int32_t i = 0; i < v[idata]; ++i) {
          const std::string tag = file.string();
          if (tag == "x")
            x = file.values<double>(numpart);
          else if (tag == "y")
            y = file.values<double>(numpart);
          else if (tag == "z")
            z = file.values<double>(numpart);
          else if (tag == "h")
            h = file.values<float>(numpart);
          else
            file.skip();
        }
    if (x.size() != numpart || y.size() != numpart || z.size() != numpart ||
        h.size() != numpart)
      throw ParameterError("Phantom dump without x, y, z and h of its " +
                           std::to_string(numpart) + " particles");
    /* :592-619 */
    const double pmass = reals["massoftype"] * real8s["umass"] * 0.001;
    const double unit_length_in_SI = real8s["udist"] * 0.01;
    for (size_t i = 0; i < numpart; ++i)
      add_particle(x[i] * unit_length_in_SI, y[i] * unit_length_in_SI,
                   z[i] * unit_length_in_SI, pmass, h[i] * unit_length_in_SI);
  }
  explicit PhantomSnapshotDensityFunction(ParameterFile &params)
      : PhantomSnapshotDensityFunction(
            params.get_filename("DensityFunction:filename"),
            params.get_physical_value(QUANTITY_TEMPERATURE,
                                      "DensityFunction:initial temperature",
                                      "8000. K"),
            params.get_bool("DensityFunction:use new algorithm", false),
            params.get_bool("DensityFunction:use periodic box", false),
            params.get_bool("DensityFunction:oriented cell faces", false)) {}
};

/* src/SPHNGSnapshotDensityFunction.cpp:95-470: the gas particles (iphase 0)
 * of an SPHNG dump, tagged ("FT...") or not: a header of dictionaries -
 * particle numbers, units (`udist` cm, `umass` g) - and per block the arrays
 * isteps, iphase, (iunique,) x, y, z, m, h in a fixed order, everything else
 * skipped. The write-statistics and binary-dump side outputs of the reference
 * are not provided. */
class SPHNGSnapshotDensityFunction : public SphKernelDensityFunction {
public:
  SPHNGSnapshotDensityFunction(const std::string &filename,
                               double initial_temperature,
                               bool use_new_algorithm,
                               bool oriented_faces = false)
      : SphKernelDensityFunction(initial_temperature, use_new_algorithm,
                                 oriented_faces) {
    FortranRecords file(filename);
    file.skip();
    const std::string fileident = file.string();
    if (fileident.empty() || fileident[0] != 'F')
      throw ParameterError("Unsupported SPHNG snapshot format: " + fileident +
                           "!");
    const bool tagged = fileident.size() > 1 && fileident[1] == 'T';
    auto skip_tagged = [&]() { /* a tag, if the file has tags, and its data */
      if (tagged)
        file.skip();
      file.skip();
    };
    /* :146-164 */
    std::map<std::string, uint32_t> numbers =
        read_fortran_dict<uint32_t>(file, tagged, true);
    if (!tagged) {
      const size_t numnumbers = numbers.size();
      numbers["nparttot"] = numbers["tag"];
      numbers["nblocks"] = numnumbers == 6 ? 1u : numbers["tag6"];
    }
    const uint32_t numblock = numbers["nblocks"];
    /* :170-176: three absent blocks */
    file.skip();
    file.skip();
    file.skip();
    /* :178-190: the highest unique index, if there */
    if (file.value<int32_t>() == 1)
      skip_tagged();
    /* :192-209: a dictionary of doubles that is not used */
    file.skip();
    skip_tagged();
    /* :211-213 */
    file.skip();
    /* :215-229 */
    std::map<std::string, double> units =
        read_fortran_dict<double>(file, tagged, true);
    if (!tagged) {
      units["udist"] = units["tag"];
      units["umass"] = units["tag1"];
    }
    /* :231-232 */
    file.skip();
    const double unit_length = units["udist"] * 0.01;
    const double unit_mass = units["umass"] * 0.001;
    for (uint32_t iblock = 0; iblock < numblock; ++iblock) {
      /* :277-283 */
      uint64_t npart;
      uint32_t nums[8];
      {
        const std::vector<uint8_t> block = file.read(8 + 8 * 4);
        std::memcpy(&npart, block.data(), 8);
        std::memcpy(nums, block.data() + 8, 32);
      }
      file.read(8 + 8 * 4); /* the sink particles' numbers */
      if (tagged)
        file.skip();
      file.values<int32_t>(npart); /* isteps */
      if (nums[0] >= 2)
        skip_tagged();
      const std::string tag = tagged ? file.string() : "iphase";
      if (tag != "iphase")
        throw ParameterError("Wrong tag: \"" + tag +
                             "\" (expected \"iphase\")!");
      const std::vector<int8_t> iphase = file.values<int8_t>(npart);
      if (nums[4] >= 1)
        skip_tagged(); /* iunique */
      std::vector<double> columns[5]; /* x y z m h */
      for (auto &column : columns) {
        if (tagged)
          file.skip();
        column = file.values<double>(npart);
      }
      /* :360-401: velocities, thermal energy, density */
      for (int i = 0; i < 5; ++i)
        skip_tagged();
      /* :403-409 */
      for (uint32_t i = 0; i + 1 < nums[6]; ++i)
        skip_tagged();
      /* :411-417: sink particle data */
      for (int i = 0; i < 10; ++i)
        skip_tagged();
      for (uint64_t i = 0; i < npart; ++i)
        if (iphase[i] == 0)
          add_particle(columns[0][i] * unit_length,
                       columns[1][i] * unit_length,
                       columns[2][i] * unit_length, columns[3][i] * unit_mass,
                       columns[4][i] * unit_length);
    }
  }
  explicit SPHNGSnapshotDensityFunction(ParameterFile &params)
      : SPHNGSnapshotDensityFunction(
            params.get_filename("DensityFunction:filename"),
            params.get_physical_value(QUANTITY_TEMPERATURE,
                                      "DensityFunction:initial temperature",
                                      "8000. K"),
            params.get_bool("DensityFunction:use new algorithm", false),
            params.get_bool("DensityFunction:oriented cell faces", false)) {
    if (params.get_bool("DensityFunction:write statistics", false) ||
        params.get_bool("DensityFunction:binary dump", false))
      throw ParameterError("SPHNGSnapshot: `write statistics` and `binary "
                           "dump` are not on this path");
  }
};

inline DensityFunction *
generate_sph_snapshot_density_function(const std::string &type,
                                       ParameterFile &params) {
  if (type == "PhantomSnapshot")
    return new PhantomSnapshotDensityFunction(params);
  if (type == "SPHNGSnapshot")
    return new SPHNGSnapshotDensityFunction(params);
  return nullptr;
}

} // namespace cmi

#endif
