/*
 * PetkovaMapping.hpp - the exact particle-to-cell mapping of the reference's
 * library mode (mapping type "Petkova"; Petkova, Laibe & Bonnell 2018): the
 * fraction of an SPH particle's mass (cubic spline kernel) that lies inside a
 * cell bounded by planar faces, as a sum of signed "vertex integrals" - one
 * per face edge and edge end.
 *
 * Restates, for the host side of the library mode (runs once before and once
 * after the GPU simulation):
 *   SPHArrayInterface::full_integral     src/SPHArrayInterface.cpp:533-737
 *   SPHArrayInterface::gridding          src/SPHArrayInterface.cpp:208-393
 *   SPHArrayInterface::gridded_integral  src/SPHArrayInterface.cpp:409-519
 *   SPHArrayInterface::mass_contribution src/SPHArrayInterface.cpp:739-925
 *
 * The reference evaluates the closed form once on a (distance to the face
 * plane, cosine of the angle under which the edge line is seen, cosine of the
 * azimuth of the vertex) lattice when the interface is constructed and
 * interpolates trilinearly afterwards (is_pre_computed = true,
 * src/SPHArrayInterface.cpp:875); the known answers of
 * test/testSPHArrayInterface.cpp:95,123,151 are answers of the INTERPOLATED
 * form, so the lattice (nodes, its zero row and column, its copied last row)
 * is kept exactly. Only the table's storage (one flat array) and the
 * evaluation (one helper for the angle integrals the reference writes out
 * four times, lattice nodes from two small functions, OpenMP over rows) are
 * this file's own.
 */
#ifndef CMI_HOST_PETKOVAMAPPING_HPP
#define CMI_HOST_PETKOVAMAPPING_HPP

#include "Plugins.hpp"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace cmi {

class PetkovaMapping {
  /* the lattice of SPHArrayInterface::gridding: distances 0 < r0 <= 2 (in
   * units of the kernel's h; 50 nodes up to 0.1, 199 beyond), cosines 0 ...
   * 1 (150 nodes up to 0.98, 150 beyond) for both angles */
  static constexpr int NR_FINE = 50, NR_COARSE = 199, NCOS = 150;
  static constexpr int NROW = NR_FINE + NR_COARSE + 2; /* 251 */
  static constexpr int NCOL = 2 * NCOS + 1;            /* 301 */
  static constexpr double R_SPLIT = 0.1, COS_SPLIT = 0.98;

  std::vector<double> _table; /* [NROW][NCOL][NCOL] */

  static double radius_node(int row) { /* rows 1 ... NROW - 2 */
    if (row <= NR_FINE)
      return (R_SPLIT / NR_FINE) * row * 1.0;
    return R_SPLIT + ((2.0 - R_SPLIT) / NR_COARSE) * (row - NR_FINE) * 1.0;
  }
  static double cosine_node(int col) { /* columns 0 ... 2 NCOS - 1 */
    if (col < NCOS)
      return (COS_SPLIT / (NCOS - 1)) * col;
    return COS_SPLIT + ((1.0 - COS_SPLIT) / NCOS) * (col - NCOS + 1);
  }
  double node(int i, int j, int k) const {
    return _table[((size_t)i * NCOL + j) * NCOL + k];
  }

  /* the integrals over the azimuth the closed form is made of, up to angle
   * phi (cosine cosp, sine sinp) for a = R_0 / r0 */
  struct AzimuthIntegrals {
    double I0, I1, I_1, I_2, I_3, I_4, I_5;
    AzimuthIntegrals(double phi, double cosp, double sinp, double a) {
      const double a2 = a * a;
      const double cosp2 = cosp * cosp;
      const double mu = cosp / std::sqrt(a2 + cosp2);
      const double tanp = sinp / cosp;
      I0 = phi;
      I_2 = phi + a2 * tanp;
      I_4 = phi + 2. * a2 * tanp +
            1. / 3. * a2 * a2 * tanp * (2. + 1. / cosp2);
      const double u = sinp * std::sqrt((1. - mu) * (1. + mu));
      const double u2 = u * u;
      const double logs = std::log((1. + u) / (1. - u));
      I1 = std::atan(u / a);
      I_1 = 0.5 * a * logs + I1;
      I_3 = I_1 + 0.25 * a * (1. + a2) * (2. * u / (1. - u2) + logs);
      I_5 = I_3 + a * (1. + a2) * (1. + a2) / 16. *
                      ((10. * u - 6. * u2 * u) / ((1. - u2) * (1. - u2)) +
                       3. * logs);
    }
  };

public:
  /* SPHArrayInterface::full_integral: the integral of the kernel (smoothing
   * length h, support 2 h) over the cone between the particle, the foot of
   * its perpendicular on the edge line (at distance R_0 from the projected
   * particle, which lies r0 from the face plane) and the vertex at azimuth
   * phi */
  static double vertex_integral(double phi, double cosphi, double r0,
                                double R_0, double h) {
    if (r0 == 0. || R_0 == 0. || phi == 0.)
      return 0.;

    const double h2 = h * h;
    const double r02 = r0 * r0;
    const double r03 = r02 * r0;
    const double r0h = r0 / h; /* (the reference multiplies by 1 / h) */
    const double r0h2 = r0h * r0h;
    const double r0h3 = r0h2 * r0h;
    const double hr0 = (1. / r0);
    const double r0h_2 = h2 * hr0 * hr0;
    const double r0h_3 = r0h_2 * h * hr0;

    /* constants of integration of the three radial pieces of the kernel */
    double B1 = 0., B2 = 0., B3 = 0.;
    if (r0 >= 2. * h) {
      B3 = 0.25 * h2 * h;
    } else if (r0 > h) {
      const double common = -4. / 3. + r0h - 0.3 * r0h2 + 1. / 30. * r0h3 -
                            1. / 15. * r0h_3;
      B3 = 0.25 * r03 * (common + 8. / 5. * r0h_2);
      B2 = 0.25 * r03 * common;
    } else {
      const double common = -2. / 3. + 0.3 * r0h2 - 0.1 * r0h3;
      B3 = 0.25 * r03 * (common + 7. / 5. * r0h_2);
      B2 = 0.25 * r03 * (common - 1. / 5. * r0h_2);
      B1 = 0.25 * r03 * common;
    }

    const double a = R_0 * hr0;
    const double linedist2 = r02 + R_0 * R_0;
    const double R = R_0 / cosphi;
    const double r2 = r02 + R * R;

    /* the parts of the inner pieces cut off by the spheres of radius h and
     * 2 h around the particle, where the edge line crosses them */
    double D2 = 0., D3 = 0.;
    if (linedist2 <= h2) {
      const double c1 = R_0 / std::sqrt(h2 - r02);
      const AzimuthIntegrals p1(std::acos(c1), c1,
                                std::sqrt((1. + c1) * (1. - c1)), a);
      D2 = -1. / 6. * p1.I_2 + 0.25 * r0h * p1.I_3 - 0.15 * r0h2 * p1.I_4 +
           1. / 30. * r0h3 * p1.I_5 - 1. / 60. * r0h_3 * p1.I1 +
           (B1 - B2) / r03 * p1.I0;
    }
    if (linedist2 <= 4. * h2) {
      const double c2 = R_0 / std::sqrt(4. * h2 - r02);
      const AzimuthIntegrals p2(std::acos(c2), c2,
                                std::sqrt((1. - c2) * (1. + c2)), a);
      D3 = 1. / 3. * p2.I_2 - 0.25 * r0h * p2.I_3 + 3. / 40. * r0h2 * p2.I_4 -
           1. / 120. * r0h3 * p2.I_5 + 4. / 15. * r0h_3 * p2.I1 +
           (B2 - B3) / r03 * p2.I0 + D2;
    }

    const AzimuthIntegrals p(phi, cosphi,
                             std::sqrt((1. - cosphi) * (1. + cosphi)), a);
    if (r2 < h2)
      return M_1_PI * r0h3 *
             (1. / 6. * p.I_2 - 3. / 40. * r0h2 * p.I_4 +
              1. / 40. * r0h3 * p.I_5 + B1 / r03 * p.I0);
    if (r2 < 4. * h2)
      return M_1_PI * r0h3 *
             (0.25 * (4. / 3. * p.I_2 - r0h * p.I_3 + 0.3 * r0h2 * p.I_4 -
                      1. / 30. * r0h3 * p.I_5 + 1. / 15. * r0h_3 * p.I1) +
              B2 / r03 * p.I0 + D2);
    return M_1_PI * r0h3 * (-0.25 * r0h_3 * p.I1 + B3 / r03 * p.I0 + D3);
  }

  /* SPHArrayInterface::gridding (called by the reference's constructors for
   * this mapping type) */
  /* tabulate = false: for callers that only use the closed form
   * (mass_fraction(..., tabulated = false): the Phantom / SPHNG density
   * functions) - no 182 MB table, no seconds of tabulation */
  explicit PetkovaMapping(bool tabulate = true)
      : _table(tabulate ? (size_t)NROW * NCOL * NCOL : 0, 0.) {
    if (!tabulate)
      return;
    const int last = NR_FINE + NR_COARSE;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int i = 1; i <= last; ++i) {
      const double r0 = radius_node(i);
      double *row = &_table[(size_t)i * NCOL * NCOL];
      for (int j = 0; j < 2 * NCOS; ++j) {
        const double mu0 = cosine_node(j);
        const double R_0 = r0 * std::sqrt(1. - mu0 * mu0) / mu0;
        for (int k = 0; k < 2 * NCOS; ++k) {
          const double cosphi = cosine_node(k);
          row[(size_t)j * NCOL + k] =
              vertex_integral(std::acos(cosphi), cosphi, r0, R_0, 1.0);
        }
      }
    }
    /* one more row so that the interpolation may read row i + 1 at r0 = 2 */
    for (int j = 0; j < 2 * NCOS; ++j)
      for (int k = 0; k < 2 * NCOS; ++k)
        _table[((size_t)(last + 1) * NCOL + j) * NCOL + k] = node(last, j, k);
  }

  /* SPHArrayInterface::gridded_integral */
  double interpolated_vertex_integral(double phi, double cosphi, double r0_old,
                                      double R_0_old, double h_old) const {
    const double h_inverse = 1. / h_old;
    double r0 = r0_old * h_inverse;
    const double R_0 = R_0_old * h_inverse;
    if (r0 == 0. || R_0 == 0. || phi == 0.)
      return 0.;
    const double mu0 = r0 / std::sqrt(r0 * r0 + R_0 * R_0);
    if (r0 > 2.)
      r0 = 2.;

    /* lattice cell and weight along each axis */
    int i, j, k;
    double fr, fm, fc;
    if (r0 < R_SPLIT) {
      i = (int)(r0 * NR_FINE / R_SPLIT);
      fr = (r0 * NR_FINE - R_SPLIT * i) / R_SPLIT;
    } else {
      i = NR_FINE + (int)((r0 - R_SPLIT) * NR_COARSE / (2. - R_SPLIT));
      fr = (r0 * NR_COARSE - R_SPLIT * NR_COARSE -
            (2. - R_SPLIT) * (i - NR_FINE)) /
           (2. - R_SPLIT);
    }
    if (mu0 < COS_SPLIT) {
      j = (int)(mu0 * (NCOS - 1) / COS_SPLIT);
      fm = (mu0 * (NCOS - 1) - COS_SPLIT * j) / COS_SPLIT;
    } else {
      j = NCOS - 1 + (int)((mu0 - COS_SPLIT) * NCOS / (1. - COS_SPLIT));
      fm = (mu0 * NCOS - COS_SPLIT * NCOS -
            (1. - COS_SPLIT) * (j - NCOS + 1)) /
           (1. - COS_SPLIT);
    }
    if (cosphi < COS_SPLIT) {
      k = (int)(cosphi * (NCOS - 1) / COS_SPLIT);
      fc = (cosphi * (NCOS - 1) - COS_SPLIT * k) / COS_SPLIT;
    } else {
      k = NCOS - 1 + (int)((cosphi - COS_SPLIT) * NCOS / (1. - COS_SPLIT));
      fc = (cosphi * NCOS - COS_SPLIT * NCOS -
            (1. - COS_SPLIT) * (k - NCOS + 1)) /
           (1. - COS_SPLIT);
    }
    /* the reference only warns about indices off the lattice; they cannot
     * leave the table for 0 <= mu0, cosphi <= 1 (rounding may push a cosine
     * one ulp over 1: clamp instead of reading past the table) */
    i = std::min(std::max(i, 0), NROW - 2);
    j = std::min(std::max(j, 0), NCOL - 2);
    k = std::min(std::max(k, 0), NCOL - 2);

    /* the lattice's last column along either cosine is "no contribution" */
    if (j == 2 * NCOS - 1 || k == 2 * NCOS - 1)
      return 0.;

    const double fx1 = fr * node(i + 1, j, k) + (1. - fr) * node(i, j, k);
    const double fx2 =
        fr * node(i + 1, j + 1, k) + (1. - fr) * node(i, j + 1, k);
    const double fx3 =
        fr * node(i + 1, j, k + 1) + (1. - fr) * node(i, j, k + 1);
    const double fx4 =
        fr * node(i + 1, j + 1, k + 1) + (1. - fr) * node(i, j + 1, k + 1);
    const double fy1 = fm * fx2 + (1. - fm) * fx1;
    const double fy2 = fm * fx4 + (1. - fm) * fx3;
    return fc * fy2 + (1. - fc) * fy1;
  }

  /* SPHArrayInterface::mass_contribution: the fraction of a particle (kernel
   * smoothing length h, i.e. HALF the SPH smoothing length the caller
   * stores) inside the cell with the given planar faces (Cell::get_faces).
   *
   * The sign of a face's vertex integrals is the sign of the particle's
   * distance to the face along the normal (v2 - v1) x (v3 - v1) of the face's
   * first three vertices: the sum is the kernel's integral over the cell when
   * all normals point INTO the cell. CartesianDensityGrid::get_faces lists the
   * vertices of two opposite faces in the same order - their normals point the
   * same way - so on a Cartesian grid the reference's sum is not that
   * integral (a particle in the middle of a large cell gets 0, the 1000
   * particles of mass 1 of test/testSPHArrayInterface.cpp come to 1.49); with
   * `oriented_towards` = NULL this function returns the reference's number
   * whatever it means - the known answers pin exactly that - and with a
   * point inside the cell (its midpoint) every face's normal is taken to
   * point towards it (and the side of an edge is judged from the face's
   * midpoint, below), which makes the sum the integral (the mapping type
   * "Petkova_oriented", not in the reference). */
  /* tabulated = false: the closed form itself at every vertex, as
   * PhantomSnapshotDensityFunction::mass_contribution does
   * (src/PhantomSnapshotDensityFunction.cpp:279-378) */
  double mass_fraction(const std::vector<Face> &faces,
                       const double particle[3], double h,
                       const double *oriented_towards = nullptr,
                       bool tabulated = true) const {
    auto det3 = [](const double *p, const double *q, const double *r) {
      return p[0] * (q[1] * r[2] - q[2] * r[1]) +
             p[1] * (q[2] * r[0] - q[0] * r[2]) +
             p[2] * (q[0] * r[1] - q[1] * r[0]);
    };
    auto distance = [](const double *p, const double *q) {
      const double d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
      return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    };
    double total = 0.;
    for (const Face &face : faces) {
      const size_t nvert = face.vertices.size();
      const double *v1 = face.vertices[0].v, *v2 = face.vertices[1].v,
                   *v3 = face.vertices[2].v;
      /* the face's plane A x + B y + C z + D = 0, the particle's signed
       * distance to it and its projection onto it */
      const double e2[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
      const double e3[3] = {v3[0] - v1[0], v3[1] - v1[1], v3[2] - v1[2]};
      const double n[3] = {e2[1] * e3[2] - e3[1] * e2[2],
                           e2[2] * e3[0] - e3[2] * e2[0],
                           e2[0] * e3[1] - e3[0] * e2[1]};
      const double D = -n[0] * v1[0] - n[1] * v1[1] - n[2] * v1[2];
      const double inverse_norm =
          1. / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      const double r0 =
          (n[0] * particle[0] + n[1] * particle[1] + n[2] * particle[2] + D) *
          inverse_norm;
      const double ar0 = std::fabs(r0);
      const double projected[3] = {particle[0] - r0 * n[0] * inverse_norm,
                                   particle[1] - r0 * n[1] * inverse_norm,
                                   particle[2] - r0 * n[2] * inverse_norm};
      const double face_orientation = det3(v1, v2, v3);
      double side = r0;
      if (oriented_towards &&
          n[0] * (oriented_towards[0] - v1[0]) +
                  n[1] * (oriented_towards[1] - v1[1]) +
                  n[2] * (oriented_towards[2] - v1[2]) <
              0.)
        side = -r0;

      for (size_t e = 0; e < nvert; ++e) {
        const double *a = face.vertices[e].v,
                     *b = face.vertices[(e + 1) % nvert].v;
        const double rab = distance(a, b);
        const double rpa = distance(projected, a);
        const double rpb = distance(projected, b);
        const double cosa = ((b[0] - a[0]) * (projected[0] - a[0]) +
                             (b[1] - a[1]) * (projected[1] - a[1]) +
                             (b[2] - a[2]) * (projected[2] - a[2])) /
                            (rpa * rab);
        /* cosine of the azimuth of end a seen from the foot of the
         * perpendicular = sine of the angle at a */
        double cosphi_a = 0.;
        if (std::fabs(cosa) < 1.)
          cosphi_a = std::sqrt((1. - cosa) * (1. + cosa));
        const double R_0 = rpa * cosphi_a;
        const double cosphi_b = R_0 / rpb;
        const double phi_a = R_0 < rpa ? std::acos(cosphi_a) : 0.;
        const double phi_b = R_0 < rpb ? std::acos(cosphi_b) : 0.;

        /* does the projected particle lie on the face's side of the edge?
         * The reference compares two determinants of POSITION vectors
         * (projected, a, b) and (v1, v2, v3): fine unless the face lies in a
         * plane through the origin of the coordinates, where both vanish and
         * every vertex integral of the face counts negative - a box centred
         * on the origin with an even number of cells has such faces. The
         * oriented variant asks the face's midpoint instead. */
        double same_side = det3(projected, a, b) * face_orientation;
        if (oriented_towards) {
          const double *c = face.midpoint.v;
          const double e[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
          const double p[3] = {projected[0] - a[0], projected[1] - a[1],
                               projected[2] - a[2]};
          const double q[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
          const double ep[3] = {e[1] * p[2] - e[2] * p[1],
                                e[2] * p[0] - e[0] * p[2],
                                e[0] * p[1] - e[1] * p[0]};
          const double eq[3] = {e[1] * q[2] - e[2] * q[1],
                                e[2] * q[0] - e[0] * q[2],
                                e[0] * q[1] - e[1] * q[0]};
          same_side = ep[0] * eq[0] + ep[1] * eq[1] + ep[2] * eq[2];
        }
        const double sign = same_side * side <= 0. ? -1. : 1.;
        const double Ia =
            tabulated
                ? interpolated_vertex_integral(phi_a, cosphi_a, ar0, R_0, h)
                : vertex_integral(phi_a, cosphi_a, ar0, R_0, h);
        const double Ib =
            tabulated
                ? interpolated_vertex_integral(phi_b, cosphi_b, ar0, R_0, h)
                : vertex_integral(phi_b, cosphi_b, ar0, R_0, h);
        const double sinphi_a = std::sqrt((1. - cosphi_a) * (1. + cosphi_a));
        const double sinphi_b = std::sqrt((1. - cosphi_b) * (1. + cosphi_b));
        /* the foot of the perpendicular outside the edge: the difference of
         * the two cones, otherwise their sum */
        if (rpa * sinphi_a >= rab || rpb * sinphi_b >= rab)
          total += sign * (phi_a >= phi_b ? Ia - Ib : Ib - Ia);
        else
          total += sign * (Ia + Ib);
      }
    }
    /* "Ensure there is no negative mass" (SPHArrayInterface's version; the
     * Phantom density function's returns the sum as it is) */
    return tabulated ? std::max(total, 1.e-6) : total;
  }
};

} // namespace cmi

#endif
