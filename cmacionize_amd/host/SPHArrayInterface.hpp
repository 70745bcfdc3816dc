/*
 * SPHArrayInterface.hpp - the coupling object of the reference's library mode
 * (src/SPHArrayInterface.hpp, src/SPHArrayInterface.cpp): a DensityFunction
 * that maps a cloud of SPH particles (positions, smoothing lengths, masses -
 * arrays owned by the calling code) onto the cells of the grid, and a
 * DensityGridWriter that maps the converged neutral fractions back onto the
 * particles. Host side only: the mapping runs once before and once after the
 * simulation; everything in between is the GPU engine.
 *
 * Mapping types of the reference (get_mapping_type, src/SPHArrayInterface.hpp:
 * 105-118):
 *   "M_over_V"  cell density = mass of a particle / cell volume; a cell hands
 *               its neutral fraction to the particle closest to its midpoint
 *               (src/SPHArrayInterface.cpp:941-942, .hpp:149-155)
 *   "centroid"  cell density = sum of m W(r / h, h) over the particles whose
 *               kernel covers the cell's midpoint, W the cubic spline kernel
 *               (src/CubicSplineKernel.hpp:36-59); back: every such particle
 *               loses W-weighted share of the cell's ionized fraction
 *               (src/SPHArrayInterface.cpp:943-959, .hpp:156-200)
 *   "Petkova"   the exact volume integral of every particle's kernel over
 *               the cell (Petkova, Laibe & Bonnell 2018; PetkovaMapping.hpp:
 *               src/SPHArrayInterface.cpp:208-925), for all particles whose
 *               kernel can reach the cell (within smoothing length + half the
 *               cell's diagonal of its midpoint); back: every such particle
 *               loses its mass-weighted share of the cell's ionized fraction
 *               (src/SPHArrayInterface.cpp:960-1003, .hpp:201-251). On a
 *               Cartesian grid the reference's sum of vertex integrals is NOT
 *               that volume integral (PetkovaMapping::mass_fraction says
 *               why); "Petkova" reproduces the reference's numbers - pinned
 *               by its known answers -, "Petkova_oriented" (not in the
 *               reference) is the integral.
 *
 * The reference finds a cell's neighbours with an Octree; here a uniform bin
 * grid over the particles does (bin side = the largest smoothing length): the
 * neighbour SET of a cell midpoint is the same - all particles with r < h -
 * so the sums are the reference's up to their order.
 */
#ifndef CMI_HOST_SPHARRAYINTERFACE_HPP
#define CMI_HOST_SPHARRAYINTERFACE_HPP

#include "GpuIonizationSimulation.hpp"
#include "PetkovaMapping.hpp"

#include <cfloat>
#include <cmath>
#include <memory>
#include <vector>

namespace cmi {

/* src/CubicSplineKernel.hpp:36-59 */
inline double cubic_spline_kernel(double u, double h) {
  const double KC1 = 2.546479089470;
  const double KC2 = 15.278874536822;
  const double KC5 = 5.092958178941;
  if (u < 1.) {
    if (u < 0.5)
      return (KC1 + KC2 * (u - 1.) * u * u) / (h * h * h);
    return KC5 * (1. - u) * (1. - u) * (1. - u) / (h * h * h);
  }
  return 0.;
}

enum SPHArrayMappingType {
  SPHARRAY_MAPPING_M_OVER_V = 0,
  SPHARRAY_MAPPING_CENTROID,
  SPHARRAY_MAPPING_PETKOVA,
  /* not in the reference: "Petkova" with every face's normal turned into the
   * cell, see PetkovaMapping::mass_fraction */
  SPHARRAY_MAPPING_PETKOVA_ORIENTED
};

class SPHArrayInterface : public DensityFunction, public DensityGridWriter {
  const double _unit_length_in_SI, _unit_mass_in_SI;
  const bool _is_periodic;
  double _box_anchor[3], _box_sides[3]; /* m */
  const SPHArrayMappingType _mapping_type;

  std::vector<double> _positions; /* [n][3], m */
  std::vector<double> _smoothing_lengths, _masses, _neutral_fractions;

  /* the bin grid over the particles */
  int _nbin[3] = {1, 1, 1};
  double _bin_side[3] = {1., 1., 1.};
  std::vector<uint32_t> _bin_start, _bin_particles;

  /* the pre-computed vertex integrals, built by the constructors like the
   * reference's gridding() (182 MB, a few seconds) */
  std::unique_ptr<PetkovaMapping> _petkova;

  static SPHArrayMappingType get_mapping_type(const std::string &name) {
    if (name == "M_over_V")
      return SPHARRAY_MAPPING_M_OVER_V;
    if (name == "centroid")
      return SPHARRAY_MAPPING_CENTROID;
    if (name == "Petkova")
      return SPHARRAY_MAPPING_PETKOVA;
    if (name == "Petkova_oriented")
      return SPHARRAY_MAPPING_PETKOVA_ORIENTED;
    throw ParameterError("Unknown SPHArrayMappingType: \"" + name + "\"!");
  }

  /* Box::periodic_distance, src/Box.hpp:113-128 (or the plain difference) */
  void separation(const double p[3], size_t index, double d[3],
                  bool only_if_periodic = false) const {
    for (int a = 0; a < 3; ++a) {
      d[a] = p[a] - _positions[3 * index + a];
      /* as the reference: its own kernel sums wrap whenever a box is set, its
       * Octree searches only in a periodic box */
      if (only_if_periodic ? _is_periodic : _box_sides[0] != 0.) {
        if (2. * d[a] < -_box_sides[a])
          d[a] += _box_sides[a];
        if (2. * d[a] >= _box_sides[a])
          d[a] -= _box_sides[a];
      }
    }
  }
  int bin_of(double x, int a) const {
    int i = (int)std::floor((x - _box_anchor[a]) / _bin_side[a]);
    if (_is_periodic)
      i = ((i % _nbin[a]) + _nbin[a]) % _nbin[a];
    else
      i = i < 0 ? 0 : (i >= _nbin[a] ? _nbin[a] - 1 : i);
    return i;
  }
  /* the distinct bins along axis a within `reach` bins of bin c */
  int bins_in_reach(int a, int c, int reach, std::vector<int> &out) const {
    out.clear();
    if (2 * reach + 1 >= _nbin[a]) {
      for (int i = 0; i < _nbin[a]; ++i)
        out.push_back(i);
    } else {
      for (int o = -reach; o <= reach; ++o) {
        int i = c + o;
        if (_is_periodic)
          i = ((i % _nbin[a]) + _nbin[a]) % _nbin[a];
        else if (i < 0 || i >= _nbin[a])
          continue;
        out.push_back(i);
      }
    }
    return (int)out.size();
  }
  /* calls f(index, r) for every particle i within margin + h_i of p:
   * Octree::get_ngbs (margin 0, src/Octree.hpp:128-161) and
   * Octree::get_ngbs_sphere (src/Octree.hpp:177-211) */
  template <typename F>
  void for_neighbours_within(const double p[3], double margin, F f) const {
    const int c[3] = {bin_of(p[0], 0), bin_of(p[1], 1), bin_of(p[2], 2)};
    /* a bin side is at least the largest smoothing length */
    std::vector<int> bins[3];
    for (int a = 0; a < 3; ++a) {
      const int reach =
          1 + (margin > 0. ? (int)std::ceil(margin / _bin_side[a]) : 0);
      bins_in_reach(a, c[a], reach, bins[a]);
    }
    for (int ix : bins[0])
      for (int iy : bins[1])
        for (int iz : bins[2]) {
          const size_t bin = ((size_t)ix * _nbin[1] + iy) * _nbin[2] + iz;
          for (uint32_t k = _bin_start[bin]; k < _bin_start[bin + 1]; ++k) {
            const size_t index = _bin_particles[k];
            double d[3];
            separation(p, index, d, true);
            double r = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            if (r > _smoothing_lengths[index] + margin)
              continue;
            if (!_is_periodic) { /* the distance the kernel sums use */
              separation(p, index, d);
              r = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            }
            f(index, r);
          }
        }
  }
  template <typename F> void for_neighbours(const double p[3], F f) const {
    for_neighbours_within(p, 0., f);
  }
  /* the particles the Petkova mapping sums over for a cell, with the mass
   * each of them has inside the cell (src/SPHArrayInterface.cpp:960-1003,
   * src/SPHArrayInterface.hpp:201-240) */
  template <typename F> void for_cell_masses(const Cell &cell, F f) const {
    const CoordinateVector mid = cell.get_cell_midpoint();
    const std::vector<Face> faces = cell.get_faces();
    if (faces.empty())
      throw ParameterError(
          "the Petkova mapping needs the faces of the grid's cells");
    /* the vertex furthest from the midpoint */
    double radius = 0.;
    for (const Face &face : faces)
      for (const CoordinateVector &v : face.vertices) {
        const double d[3] = {v[0] - mid[0], v[1] - mid[1], v[2] - mid[2]};
        radius = std::max(radius,
                          std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]));
      }
    const double p[3] = {mid[0], mid[1], mid[2]};
    for_neighbours_within(p, radius, [&](size_t index, double) {
      /* the kernel's h is half the SPH smoothing length; the particle's own
       * position, also across a periodic face - as the reference */
      const double h = 0.5 * _smoothing_lengths[index];
      f(index, _petkova->mass_fraction(
                   faces, &_positions[3 * index], h,
                   _mapping_type == SPHARRAY_MAPPING_PETKOVA_ORIENTED
                       ? p
                       : nullptr) *
                   _masses[index]);
    });
  }
  /* Octree::get_closest_ngb */
  size_t closest_particle(const double p[3]) const {
    /* rings of bins around p until a particle is found and no closer one can
     * hide in the next ring */
    const int c[3] = {bin_of(p[0], 0), bin_of(p[1], 1), bin_of(p[2], 2)};
    const int reach = std::max(_nbin[0], std::max(_nbin[1], _nbin[2]));
    const double side =
        std::min(_bin_side[0], std::min(_bin_side[1], _bin_side[2]));
    size_t best = 0;
    double best_r2 = DBL_MAX;
    for (int ring = 0; ring <= reach; ++ring) {
      if (best_r2 < DBL_MAX) {
        const double safe = (ring - 1) * side;
        if (safe > 0. && safe * safe > best_r2)
          break;
      }
      for (int ox = -ring; ox <= ring; ++ox)
        for (int oy = -ring; oy <= ring; ++oy)
          for (int oz = -ring; oz <= ring; ++oz) {
            if (std::max(std::abs(ox), std::max(std::abs(oy), std::abs(oz))) !=
                ring)
              continue;
            int i[3] = {c[0] + ox, c[1] + oy, c[2] + oz};
            bool ok = true;
            for (int a = 0; a < 3; ++a) {
              if (_is_periodic) {
                if (2 * ring + 1 > _nbin[a] &&
                    (i[a] < 0 || i[a] >= _nbin[a])) {
                  ok = false; /* the ring already wraps onto itself */
                  break;
                }
                i[a] = ((i[a] % _nbin[a]) + _nbin[a]) % _nbin[a];
              } else if (i[a] < 0 || i[a] >= _nbin[a]) {
                ok = false;
                break;
              }
            }
            if (!ok)
              continue;
            const size_t bin = ((size_t)i[0] * _nbin[1] + i[1]) * _nbin[2] + i[2];
            for (uint32_t k = _bin_start[bin]; k < _bin_start[bin + 1]; ++k) {
              const size_t index = _bin_particles[k];
              double d[3];
              separation(p, index, d);
              const double r2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
              if (r2 < best_r2) {
                best_r2 = r2;
                best = index;
              }
            }
          }
    }
    return best;
  }

  template <typename TX, typename TH>
  void reset_impl(const TX *x, const TX *y, const TX *z, const TH *h,
                  const TH *m, size_t npart) {
    /* SPHArrayInterface::reset, src/SPHArrayInterface.cpp:146-178 */
    _positions.resize(3 * npart);
    _smoothing_lengths.assign(npart, 0.);
    _masses.assign(npart, 0.);
    _neutral_fractions.assign(npart, 0.);
    for (size_t i = 0; i < npart; ++i) {
      _positions[3 * i + 0] = x[i] * _unit_length_in_SI;
      _positions[3 * i + 1] = y[i] * _unit_length_in_SI;
      _positions[3 * i + 2] = z[i] * _unit_length_in_SI;
      _smoothing_lengths[i] = h[i] * _unit_length_in_SI;
      _masses[i] = m[i] * _unit_mass_in_SI;
    }
    if (!_is_periodic) {
      double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX};
      double hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
      for (size_t i = 0; i < npart; ++i)
        for (int a = 0; a < 3; ++a) {
          lo[a] = std::min(lo[a], _positions[3 * i + a]);
          hi[a] = std::max(hi[a], _positions[3 * i + a]);
        }
      for (int a = 0; a < 3; ++a) {
        const double extent = hi[a] - lo[a];
        _box_anchor[a] = lo[a] - 0.005 * extent;
        _box_sides[a] = 1.01 * extent;
      }
    }
  }

public:
  /* src/SPHArrayInterface.cpp:41-60 */
  SPHArrayInterface(double unit_length_in_SI, double unit_mass_in_SI,
                    const std::string &mapping_type)
      : DensityGridWriter(""), _unit_length_in_SI(unit_length_in_SI),
        _unit_mass_in_SI(unit_mass_in_SI), _is_periodic(false),
        _box_anchor{0., 0., 0.}, _box_sides{0., 0., 0.},
        _mapping_type(get_mapping_type(mapping_type)) {
    if (_mapping_type == SPHARRAY_MAPPING_PETKOVA ||
        _mapping_type == SPHARRAY_MAPPING_PETKOVA_ORIENTED)
      _petkova.reset(new PetkovaMapping());
  }
  /* periodic versions, :69-130 (box in the caller's length unit) */
  template <typename T>
  SPHArrayInterface(double unit_length_in_SI, double unit_mass_in_SI,
                    const T *box_anchor, const T *box_sides,
                    const std::string &mapping_type)
      : DensityGridWriter(""), _unit_length_in_SI(unit_length_in_SI),
        _unit_mass_in_SI(unit_mass_in_SI), _is_periodic(true),
        _box_anchor{box_anchor[0] * unit_length_in_SI,
                    box_anchor[1] * unit_length_in_SI,
                    box_anchor[2] * unit_length_in_SI},
        _box_sides{box_sides[0] * unit_length_in_SI,
                   box_sides[1] * unit_length_in_SI,
                   box_sides[2] * unit_length_in_SI},
        _mapping_type(get_mapping_type(mapping_type)) {
    if (_mapping_type == SPHARRAY_MAPPING_PETKOVA ||
        _mapping_type == SPHARRAY_MAPPING_PETKOVA_ORIENTED)
      _petkova.reset(new PetkovaMapping());
  }

  void reset(const double *x, const double *y, const double *z,
             const double *h, const double *m, size_t npart) {
    reset_impl(x, y, z, h, m, npart);
  }
  void reset(const double *x, const double *y, const double *z, const float *h,
             const float *m, size_t npart) {
    reset_impl(x, y, z, h, m, npart);
  }
  void reset(const float *x, const float *y, const float *z, const float *h,
             const float *m, size_t npart) {
    reset_impl(x, y, z, h, m, npart);
  }
  size_t number_of_particles() const { return _masses.size(); }

  /* SPHArrayInterface::initialize, :199-203: the search structure */
  void initialize() override {
    const size_t npart = _masses.size();
    double hmax = 0.;
    for (double h : _smoothing_lengths)
      hmax = std::max(hmax, h);
    for (int a = 0; a < 3; ++a) {
      const double side = _box_sides[a] > 0. ? _box_sides[a] : 1.;
      int n = hmax > 0. ? (int)std::floor(side / hmax) : 1;
      n = std::max(1, std::min(n, 256));
      _nbin[a] = n;
      _bin_side[a] = side / n;
    }
    const size_t nbins = (size_t)_nbin[0] * _nbin[1] * _nbin[2];
    _bin_start.assign(nbins + 1, 0);
    std::vector<uint32_t> which(npart);
    for (size_t i = 0; i < npart; ++i) {
      const size_t bin = ((size_t)bin_of(_positions[3 * i], 0) * _nbin[1] +
                          bin_of(_positions[3 * i + 1], 1)) *
                             _nbin[2] +
                         bin_of(_positions[3 * i + 2], 2);
      which[i] = (uint32_t)bin;
      ++_bin_start[bin + 1];
    }
    for (size_t b = 0; b < nbins; ++b)
      _bin_start[b + 1] += _bin_start[b];
    _bin_particles.resize(npart);
    std::vector<uint32_t> cursor(_bin_start.begin(), _bin_start.end() - 1);
    for (size_t i = 0; i < npart; ++i)
      _bin_particles[cursor[which[i]]++] = (uint32_t)i;
  }

  /* SPHArrayInterface::operator(), :931-1010 */
  DensityValues operator()(const Cell &cell) override {
    DensityValues values;
    const CoordinateVector mid = cell.get_cell_midpoint();
    const double p[3] = {mid[0], mid[1], mid[2]};
    double density = 0.;
    if (_mapping_type == SPHARRAY_MAPPING_M_OVER_V) {
      density = _masses[0] / cell.get_volume();
    } else if (_mapping_type == SPHARRAY_MAPPING_CENTROID) {
      for_neighbours(p, [&](size_t index, double r) {
        const double h = _smoothing_lengths[index];
        density += _masses[index] * cubic_spline_kernel(r / h, h);
      });
    } else {
      for_cell_masses(cell, [&](size_t, double mass) { density += mass; });
      density = density / cell.get_volume();
    }
    /* "Ensure that the density > 0" */
    if (density <= 0.)
      density = _masses[0] / cell.get_volume() * 1.e-6;
    values.set_number_density(density / 1.6737236e-27);
    values.set_temperature(8000.);
    values.set_ionic_fraction(ION_H_n, 1.e-6);
    values.set_ionic_fraction(ION_He_n, 1.e-6);
    return values;
  }

  /* SPHArrayInterface::write, :1049-1075 with InverseMappingFunction,
   * src/SPHArrayInterface.hpp:123-215 */
  void write(DensityGrid &grid, uint_fast32_t, ParameterFile &,
             double = 0.) override {
    for (double &nf : _neutral_fractions)
      nf = 1.;
    const int64_t ncell = grid.get_number_of_cells();
    if (_mapping_type == SPHARRAY_MAPPING_M_OVER_V) {
      /* the last cell that names a particle wins: in cell order */
      for (int64_t c = 0; c < ncell; ++c) {
        DensityGrid::iterator cell(&grid, c);
        const CoordinateVector mid = cell.get_cell_midpoint();
        const double p[3] = {mid[0], mid[1], mid[2]};
        _neutral_fractions[closest_particle(p)] =
            cell.get_ionization_variables().get_ionic_fraction(ION_H_n);
      }
      return;
    }
    /* the reference runs the cells on its worker threads with one lock per
     * particle; atomic subtractions here */
    std::string error;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t c = 0; c < ncell; ++c) {
      DensityGrid::iterator cell(&grid, c);
      const CoordinateVector mid = cell.get_cell_midpoint();
      const double p[3] = {mid[0], mid[1], mid[2]};
      const double xH =
          cell.get_ionization_variables().get_ionic_fraction(ION_H_n);
      std::vector<std::pair<size_t, double>> inside;
      double cell_mass = 0.;
      try {
        if (_mapping_type != SPHARRAY_MAPPING_CENTROID) {
          for_cell_masses(cell, [&](size_t index, double mass) {
            inside.emplace_back(index, mass);
            cell_mass += mass;
          });
        } else {
          for_neighbours(p, [&](size_t index, double r) {
            const double h = _smoothing_lengths[index];
            const double mass = _masses[index] * cubic_spline_kernel(r / h, h);
            inside.emplace_back(index, mass);
            cell_mass += mass;
          });
        }
      } catch (const std::exception &e) {
#pragma omp critical(sph_array_interface_error)
        error = e.what();
        continue;
      }
      for (const auto &part : inside) {
        const double share = part.second / cell_mass * (1. - xH);
#pragma omp atomic
        _neutral_fractions[part.first] -= share;
      }
    }
    if (!error.empty())
      throw std::runtime_error(error);
  }

  /* SPHArrayInterface::fill_array, :1018-1035 */
  template <typename T> void fill_array(T *nH) const {
    for (size_t i = 0; i < _neutral_fractions.size(); ++i)
      nH[i] = (T)_neutral_fractions[i];
  }
};

} // namespace cmi

#endif
