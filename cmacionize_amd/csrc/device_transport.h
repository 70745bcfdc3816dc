/*
 * device_transport.h - device functions for packet emission and the DDA
 * march through the regular grid.
 *
 * Two marchers implement CartesianDensityGrid::interact
 * (src/CartesianDensityGrid.cpp:375-452):
 *
 *  EXACT  restates the reference's arithmetic operation by operation: every
 *         step recomputes the cell walls from the cell index and the wall
 *         distances from the current position. Path lengths are bit-identical
 *         to the CPU oracle's. Used by the trace probe and selectable for the
 *         transport kernel (tuning "exact_dda").
 *
 *  FAST   the incremental form of the same traversal (Amanatides & Woo): the
 *         ray is parametrised as origin + t * direction, the parameter of the
 *         next wall crossing per axis is initialised exactly like the
 *         reference's first step and then advanced by cellside / |direction|.
 *         Same cells, same tie rule (every axis that ties the minimum
 *         advances), path lengths equal to the reference's up to rounding
 *         (tested: |ds - ds_exact| <= 1e-12 x cellside). ~5x fewer
 *         instructions per step; the default.
 */
#ifndef CMI_DEVICE_TRANSPORT_H
#define CMI_DEVICE_TRANSPORT_H

#include "device_physics.h"
#include "device_spectra.h"

#include <float.h>

/* State of one packet in flight; Photon of src/Photon.hpp:36-69 minus the
 * fields the path never reads (Stokes / direction parameters). FULL = all 14
 * cross sections are carried; otherwise only hydrogen's. Members a kernel
 * variant does not use cost no registers. */
template <bool FULL> struct Packet {
  double pos[3]; /* EXACT: current position; FAST: origin of the flight */
  double dir[3];
  double inv_dir[3];
  double tau;    /* remaining optical depth */
  double nu;     /* frequency (Hz) */
  double weight;
  double sigma_H;
  double sigma_He_corr; /* A_He * sigma_He */
  double sigma_He;      /* for the re-emission decision */
  /* FAST marcher */
  double t;         /* path parameter reached so far */
  double tmax[3];   /* parameter of the next wall crossing per axis */
  double tdelta[3]; /* parameter distance between walls (0 if dir == 0) */
  int32_t cell;     /* long index of the current cell */
  int32_t cstep[3]; /* change of the long index when the axis advances */
  int32_t rem[3];   /* cells left before the box face in the travel direction;
                     * negative = the packet has left through that face */
  /* tile rounds: coordinates of the current cell inside its tile, and their
   * change (+1 / -1) when the axis advances */
  int32_t lc[3];
  int32_t lsgn[3];
  /* EXACT marcher */
  int32_t index[3];
  int32_t type;
};

/* PhotonSource::get_random_direction + Photon::set_direction
 * (src/PhotonSource.hpp:140-148, src/Photon.hpp:165-170) */
template <bool FULL>
__device__ __forceinline__ void random_direction(Packet<FULL> &p,
                                                 PacketRng &rng) {
  const double cost = 2. * rng.next() - 1.;
  const double sint = sqrt(fmax(1. - cost * cost, 0.));
  const double phi = 2. * M_PI * rng.next();
  double sinp, cosp;
  sincos(phi, &sinp, &cosp);
  p.dir[0] = sint * cosp;
  p.dir[1] = sint * sinp;
  p.dir[2] = cost;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    p.inv_dir[a] = 1. / p.dir[a];
}

/* PhotonSource::set_cross_sections, src/PhotonSource.cpp:189-199. The packet
 * keeps only what the march needs (sigma_H, A_He sigma_He); the 16 per-step
 * accumulation weights of DensityGrid::update_integrals
 * (src/DensityGrid.hpp:162-173) - sigma_ion for the 14 ions, and
 * sigma_H (nu - nu_H), sigma_He (nu - nu_He) for the two heating terms - are
 * returned in `weights` for the caller to keep where it accumulates from. */
template <bool FULL>
__device__ __forceinline__ void
set_cross_sections(const ModelDev &m, Packet<FULL> &p,
                   double (&weights)[CMI_NACC]) {
  if (FULL) {
    cmi_cross_sections(m, p.nu, weights);
    p.sigma_H = weights[ION_H_n];
    p.sigma_He = weights[ION_He_n];
    p.sigma_He_corr = m.abundance[0] * p.sigma_He;
    weights[CMI_NION] = p.sigma_H * (p.nu - m.nu_H);
    weights[CMI_NION + 1] = p.sigma_He * (p.nu - m.nu_He);
  } else {
    /* only reached with FixedValueCrossSections and sigma[1..13] == 0 */
    p.sigma_H = m.xsec_fixed[ION_H_n];
    p.sigma_He = m.xsec_fixed[ION_He_n];
    p.sigma_He_corr = m.abundance[0] * p.sigma_He;
    weights[ION_H_n] = p.sigma_H;
    weights[CMI_NION] = p.sigma_H * (p.nu - m.nu_H);
  }
}

/* CartesianDensityGrid::get_cell_indices, src/CartesianDensityGrid.cpp:152-161
 * (truncating conversion, like the reference's implicit double -> int) */
template <bool FULL>
__device__ __forceinline__ void locate_cell(const GridDev &g,
                                            Packet<FULL> &p) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
    p.index[a] = (int32_t)((p.pos[a] - g.anchor[a]) * g.inv_cellside[a]) -
                 g.offset[a];
}

/* CartesianDensityGrid::is_inside, src/CartesianDensityGrid.cpp:187-227 */
template <bool FULL>
__device__ __forceinline__ bool is_inside(const GridDev &g, Packet<FULL> &p) {
  bool inside = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (!g.periodic[a]) {
      inside &= (p.index[a] >= 0 && p.index[a] < g.ncell[a]);
    } else {
      if (p.index[a] < 0) {
        p.index[a] = g.ncell[a] - 1;
        p.pos[a] += g.box_sides[a];
      }
      if (p.index[a] >= g.ncell[a]) {
        p.index[a] = 0;
        p.pos[a] -= g.box_sides[a];
      }
    }
  }
  return inside;
}

/* Begin a flight from p.pos along p.dir: interact() starts from the cell that
 * contains the position (src/CartesianDensityGrid.cpp:386). FAST also sets up
 * the wall-crossing parameters, with the first crossing computed exactly as
 * the reference's first get_wall_intersection (:280-318). */
template <bool FULL, bool EXACT>
__device__ __forceinline__ void start_flight(const GridDev &g,
                                             Packet<FULL> &p) {
  locate_cell(g, p);
  if (!EXACT) {
    /* a start outside a non-periodic box ends the flight at once
     * (is_inside(), :187-227); periodic axes wrap the start position */
    const bool inside = is_inside(g, p);
    p.t = 0.;
    const int32_t stride[3] = {g.ncell[1] * g.ncell[2], g.ncell[2], 1};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double lo =
          g.anchor[a] + g.cellside[a] * (p.index[a] + g.offset[a]);
      const double hi = lo + g.cellside[a];
      p.tmax[a] =
          (p.dir[a] > 0.)
              ? (hi - p.pos[a]) * p.inv_dir[a]
              : ((p.dir[a] < 0.) ? (lo - p.pos[a]) * p.inv_dir[a] : DBL_MAX);
      /* an axis the packet does not move along never ties the minimum */
      p.tdelta[a] = (p.dir[a] != 0.) ? g.cellside[a] * fabs(p.inv_dir[a]) : 0.;
      p.cstep[a] = (p.dir[a] > 0.) ? stride[a] : -stride[a];
      p.rem[a] = (p.dir[a] > 0.) ? g.ncell[a] - 1 - p.index[a] : p.index[a];
    }
    if (!inside)
      p.rem[0] = -1;
    p.cell = (p.index[0] * g.ncell[1] + p.index[1]) * g.ncell[2] + p.index[2];
  }
}

/* PhotonSource::get_random_photon (discrete branch) + the first optical depth
 * of IonizationPhotonShootJob::execute
 * (src/PhotonSource.cpp:208-249, src/IonizationPhotonShootJob.hpp:119-135) */
/* ... in two parts, so that a block of a decomposed grid can drop a packet
 * that starts elsewhere before paying for its frequency and cross sections:
 * where and in which direction (draws 1-4) ... */
template <bool FULL, bool EXACT>
__device__ inline uint32_t emit_geometry(const GridDev &g, const ModelDev &m,
                                         PacketRng &rng, Packet<FULL> &p) {
  /* first uniform: continuous or discrete source (drawn also when there is
   * no continuous source: continuous_probability = 0) */
  double x = rng.next();
  uint32_t origin = 0;
  if (x >= m.continuous_probability) {
    x = rng.next();
    int i = 0;
    while (x > m.source_cumulative[i])
      ++i;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      p.pos[a] = m.source_position[3 * i + a];
    random_direction(p, rng);
  } else if (m.continuous_type == 2) {
    /* PlanarContinuousPhotonSource::get_random_incoming_direction
     * (src/PlanarContinuousPhotonSource.hpp:165-188): a point of a rectangle
     * in the plane x[axis] = intercept, an isotropic direction */
    origin = 1;
    const int i0 = m.continuous_axis == 0 ? 1 : 0;
    const int i1 = m.continuous_axis == 2 ? 1 : 2;
    const double u0 = rng.next();
    const double u1 = rng.next();
#pragma unroll
    for (int a = 0; a < 3; ++a)
      p.pos[a] = a == m.continuous_axis
                     ? m.continuous_intercept
                     : (a == i0 ? m.continuous_anchor[0] + u0 * m.continuous_side[0]
                                : m.continuous_anchor[1] + u1 * m.continuous_side[1]);
    (void)i1;
    random_direction(p, rng);
  } else {
    /* IsotropicContinuousPhotonSource::get_random_incoming_direction
     * (src/IsotropicContinuousPhotonSource.hpp:95-191): a focus point in the
     * box, an isotropic direction through it, and the point where that line
     * enters the box */
    origin = 1;
    double focus[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
      focus[a] = g.anchor[a] + g.box_sides[a] * rng.next();
    random_direction(p, rng);
    double l[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double top = g.anchor[a] + g.box_sides[a];
      l[a] = (p.dir[a] < 0.)
                 ? (top - focus[a]) / p.dir[a]
                 : ((p.dir[a] > 0.) ? (g.anchor[a] - focus[a]) / p.dir[a]
                                    : -DBL_MAX);
    }
    const double maxl = fmax(fmax(l[0], l[1]), l[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double top = g.anchor[a] + g.box_sides[a];
      double q = focus[a] + maxl * p.dir[a];
      /* the top anchor itself lies outside the box */
      q = fmin(q, top - 2.220446049250313e-16 * g.box_sides[a]);
      p.pos[a] = fmax(q, g.anchor[a]);
    }
  }
  p.type = TYPE_PRIMARY;
  p.weight = m.photon_weight[origin];
  start_flight<FULL, EXACT>(g, p);
  return origin;
}

/* ... and what it is: frequency, cross sections, first optical depth */
template <bool FULL>
__device__ inline void emit_physics(const ModelDev &m, PacketRng &rng,
                                    Packet<FULL> &p,
                                    double (&weights)[CMI_NACC],
                                    uint32_t origin = 0) {
  p.nu = sample_source_spectrum(m, rng, origin);
  set_cross_sections(m, p, weights);
  p.tau = -log(rng.next());
}

/* the same from a row written by direction_key_kernel - {sigma[14], nu, tau}
 * of this packet, computed there with the same functions: the random numbers
 * the spectrum and the optical depth consumed are drawn and dropped */
template <bool FULL>
__device__ __forceinline__ void
emit_physics_from_row(const ModelDev &m, PacketRng &rng, Packet<FULL> &p,
                      double (&weights)[CMI_NACC], uint32_t origin,
                      const double *row) {
  const double4 *r4 = reinterpret_cast<const double4 *>(row);
#pragma unroll
  for (int k = 0; k < CMI_NACC; k += 4) {
    const double4 v = r4[k >> 2];
    weights[k] = v.x;
    weights[k + 1] = v.y;
    weights[k + 2] = v.z;
    weights[k + 3] = v.w;
  }
  p.nu = weights[CMI_NION];
  p.tau = weights[CMI_NION + 1];
  p.sigma_H = weights[ION_H_n];
  p.sigma_He = weights[ION_He_n];
  p.sigma_He_corr = m.abundance[0] * p.sigma_He;
  weights[CMI_NION] = p.sigma_H * (p.nu - m.nu_H);
  weights[CMI_NION + 1] = p.sigma_He * (p.nu - m.nu_He);
  /* sample_source_spectrum: one uniform for a Planck spectrum, none for a
   * monochromatic one; then the optical depth's */
  const bool draws = origin != 0 ? m.continuous_spectrum_type != 0
                                 : m.spectrum_type != 0;
  if (draws)
    (void)rng.next();
  (void)rng.next();
}

template <bool FULL, bool EXACT>
__device__ inline uint32_t emit_packet(const GridDev &g, const ModelDev &m,
                                       PacketRng &rng, Packet<FULL> &p,
                                       double (&weights)[CMI_NACC]) {
  const uint32_t origin = emit_geometry<FULL, EXACT>(g, m, rng, p);
  emit_physics<FULL>(m, rng, p, weights, origin);
  return origin;
}

/* EXACT: one iteration of the loop of CartesianDensityGrid::interact
 * (src/CartesianDensityGrid.cpp:396-432) with get_wall_intersection
 * (:280-318) and get_optical_depth (src/DensityGrid.hpp:117-140) inlined.
 * The caller has checked that the packet is inside and tau > 0.
 * Returns the path length travelled in the cell; `cell` receives the long
 * index, `kappa` the cell's transport record. */
template <bool FULL>
__device__ __forceinline__ double dda_step(const GridDev &g,
                                           const double2 *__restrict__ opacity,
                                           Packet<FULL> &p, int64_t &cell,
                                           double2 &kappa) {
  cell = ((int64_t)p.index[0] * g.ncell[1] + p.index[1]) * g.ncell[2] +
         p.index[2];
  kappa = opacity[cell];

  double d[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double lo = g.anchor[a] + g.cellside[a] * (p.index[a] + g.offset[a]);
    const double hi = lo + g.cellside[a];
    d[a] = (p.dir[a] > 0.)
               ? (hi - p.pos[a]) * p.inv_dir[a]
               : ((p.dir[a] < 0.) ? (lo - p.pos[a]) * p.inv_dir[a] : DBL_MAX);
  }
  double ds = fmin(d[0], fmin(d[1], d[2]));
  double wall[3];
  int32_t step[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    /* every axis that ties the minimum advances (edges, corners) */
    step[a] = (d[a] == ds) ? ((p.dir[a] > 0.) ? 1 : -1) : 0;
    wall[a] = p.pos[a] + ds * p.dir[a];
  }

  /* kappa = {n x_H, n x_He}; a negative .x marks vacuum */
  const double kH = fmax(kappa.x, 0.);
  const double tau_cell = ds * (p.sigma_H * kH + p.sigma_He_corr * kappa.y);
  p.tau -= tau_cell;
  if (p.tau < 0.) {
    const double Scorr = ds * p.tau / tau_cell;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      p.pos[a] += (wall[a] - p.pos[a]) * (ds + Scorr) / ds;
    ds += Scorr;
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      p.pos[a] = wall[a];
      p.index[a] += step[a];
    }
  }
  return ds;
}

/* FAST marcher: has the packet left the box? */
template <bool FULL>
__device__ __forceinline__ bool fast_outside(const Packet<FULL> &p) {
  return (p.rem[0] | p.rem[1] | p.rem[2]) < 0;
}

/* v_min_f64 without the canonicalisation the compiler wraps around fmin() in
 * IEEE mode (the operands are never NaN here) */
__device__ __forceinline__ double min_f64(double x, double y) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ double max_f64(double x, double y) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

/* FAST: the same loop iteration in incremental form, written without
 * branches on the common path. Precondition: not outside and tau > 0. Returns
 * the path length in `cell`. Afterwards tau < 0 means absorbed in `cell` at
 * parameter p.t (the marcher state has still advanced to the wall - it is not
 * read again, end_flight() / a new start_flight() follow); otherwise the
 * packet is on the wall, in p.cell, or outside (fast_outside()). Periodic
 * wrapping is left to fast_wrap(), which the caller runs when any axis of the
 * grid is periodic. */
/* (the byte offset is kept in 32 bits so that the load can use the scalar
 * base + 32-bit vector offset addressing form: the FAST marcher therefore
 * serves engines of fewer than CMI_FAST_MARCHER_MAX_CELLS cells; the host
 * selects the EXACT marcher above that) */
#define CMI_FAST_MARCHER_MAX_CELLS (1ll << 28)
template <bool FULL>
__device__ __forceinline__ double2
fast_load_record(const double2 *__restrict__ opacity, const Packet<FULL> &p) {
  return *reinterpret_cast<const double2 *>(
      reinterpret_cast<const char *>(opacity) + ((uint32_t)p.cell << 4));
}

/* `kappa` is the transport record of p.cell, loaded by the caller
 * (fast_load_record) - early, so that the load overlaps other work. */
template <bool FULL, bool TILE = false>
__device__ __forceinline__ double fast_step(Packet<FULL> &p, int32_t &cell,
                                            const double2 kappa) {
  cell = p.cell;
  const double tmin = min_f64(p.tmax[0], min_f64(p.tmax[1], p.tmax[2]));
  double ds = tmin - p.t;
  /* kappa = {n x_H, n x_He}; a negative .x marks vacuum */
  const double kH = max_f64(kappa.x, 0.);
  const double tau_cell =
      FULL ? ds * (p.sigma_H * kH + p.sigma_He_corr * kappa.y)
           : ds * (p.sigma_H * kH);
  p.tau -= tau_cell;
  const double t_old = p.t;
  p.t = tmin;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    /* every tied axis advances - under the execution mask (round 4): the
     * compare writes the lanes' mask to a scalar register pair, and the wall
     * parameter, the cell index and the cells left before the face take ONE
     * add each (tmax + tdelta is the exactly rounded sum the fused
     * multiply-add with hit = 1. gave). As selects these were eight vector
     * instructions per axis; the mask costs two scalar ones. */
    const unsigned long long hit =
        __builtin_amdgcn_fcmp(p.tmax[a], tmin, 1 /* ordered, equal */);
    unsigned long long saved;
    if (TILE)
      asm volatile("s_and_saveexec_b64 %4, %5\n\t"
                   "v_add_f64 %0, %0, %6\n\t"
                   "v_add_u32 %1, %1, %7\n\t"
                   "v_add_u32 %2, -1, %2\n\t"
                   "v_add_u32 %3, %3, %8\n\t"
                   "s_mov_b64 exec, %4"
                   : "+v"(p.tmax[a]), "+v"(p.cell), "+v"(p.rem[a]),
                     "+v"(p.lc[a]), "=&s"(saved)
                   : "s"(hit), "v"(p.tdelta[a]), "v"(p.cstep[a]),
                     "v"(p.lsgn[a])
                   : "scc");
    else
      asm volatile("s_and_saveexec_b64 %3, %4\n\t"
                   "v_add_f64 %0, %0, %5\n\t"
                   "v_add_u32 %1, %1, %6\n\t"
                   "v_add_u32 %2, -1, %2\n\t"
                   "s_mov_b64 exec, %3"
                   : "+v"(p.tmax[a]), "+v"(p.cell), "+v"(p.rem[a]),
                     "=&s"(saved)
                   : "s"(hit), "v"(p.tdelta[a]), "v"(p.cstep[a])
                   : "scc");
  }
  if (p.tau < 0.) {
    ds += ds * p.tau / tau_cell; /* Scorr */
    p.t = t_old + ds;
  }
  return ds;
}

/* FAST: is_inside() for periodic axes - wrap the cell index and shift the
 * flight's origin by a box side, so that origin + t * dir stays the wrapped
 * position (src/CartesianDensityGrid.cpp:187-227) */
template <bool FULL>
__device__ __forceinline__ void fast_wrap(const GridDev &g, Packet<FULL> &p) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (g.periodic[a] && p.rem[a] < 0) {
      p.cell -= p.cstep[a] * g.ncell[a];
      p.rem[a] = g.ncell[a] - 1;
      p.pos[a] -= (p.cstep[a] > 0 ? 1. : -1.) * g.box_sides[a];
    }
  }
}

/* FAST, decomposed grids: continue a flight that another block handed over.
 * The marcher's parametric state (origin, direction, t, tmax) travels with the
 * packet, so the arithmetic goes on exactly as if the grid were whole; only
 * the block-local bookkeeping is rebuilt from the cell being entered. */
template <bool FULL>
__device__ __forceinline__ void resume_flight(const GridDev &g, Packet<FULL> &p,
                                              int64_t cell_global) {
  const int64_t gz = cell_global % g.global_ncell[2];
  const int64_t gy = (cell_global / g.global_ncell[2]) % g.global_ncell[1];
  const int64_t gx = cell_global / ((int64_t)g.global_ncell[2] *
                                    g.global_ncell[1]);
  p.index[0] = (int32_t)gx - g.offset[0];
  p.index[1] = (int32_t)gy - g.offset[1];
  p.index[2] = (int32_t)gz - g.offset[2];
  bool inside = true;
  const int32_t stride[3] = {g.ncell[1] * g.ncell[2], g.ncell[2], 1};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    inside &= (p.index[a] >= 0 && p.index[a] < g.ncell[a]);
    p.tdelta[a] = (p.dir[a] != 0.) ? g.cellside[a] * fabs(p.inv_dir[a]) : 0.;
    p.cstep[a] = (p.dir[a] > 0.) ? stride[a] : -stride[a];
    p.rem[a] = (p.dir[a] > 0.) ? g.ncell[a] - 1 - p.index[a] : p.index[a];
  }
  if (!inside)
    p.rem[0] = -1;
  p.cell = (p.index[0] * g.ncell[1] + p.index[1]) * g.ncell[2] + p.index[2];
}

/* FAST: the same for a flight that waited inside this engine (tile rounds):
 * `cell` is the long index in THIS engine's grid */
template <bool FULL>
__device__ __forceinline__ void resume_flight_local(const GridDev &g,
                                                    Packet<FULL> &p,
                                                    int32_t cell) {
  const int32_t iz = cell % g.ncell[2];
  const int32_t iy = (cell / g.ncell[2]) % g.ncell[1];
  const int32_t ix = cell / (g.ncell[2] * g.ncell[1]);
  p.index[0] = ix;
  p.index[1] = iy;
  p.index[2] = iz;
  const int32_t stride[3] = {g.ncell[1] * g.ncell[2], g.ncell[2], 1};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p.tdelta[a] = (p.dir[a] != 0.) ? g.cellside[a] * fabs(p.inv_dir[a]) : 0.;
    p.cstep[a] = (p.dir[a] > 0.) ? stride[a] : -stride[a];
    p.rem[a] = (p.dir[a] > 0.) ? g.ncell[a] - 1 - p.index[a] : p.index[a];
  }
  p.cell = cell;
}

/* FAST, decomposed grids: the packet has stepped out of this block from
 * `last_cell` into p.cell (a long index one past a face). Returns the long
 * index of that cell in the WHOLE grid, or -1 if it lies outside the whole
 * grid; across a periodic face of the whole box the flight's origin is
 * shifted by a box side. Every tied axis advanced by one cell, so the
 * difference of the two long indices decodes uniquely (blocks are at least 3
 * cells wide). */
template <bool FULL>
__device__ __forceinline__ int64_t
exit_cell_global(const GridDev &g, Packet<FULL> &p, int32_t last_cell) {
  const int32_t ny = g.ncell[1], nz = g.ncell[2];
  int32_t iz = last_cell % nz;
  int32_t iy = (last_cell / nz) % ny;
  int32_t ix = last_cell / (nz * ny);
  int32_t d = p.cell - last_cell;
  int32_t dz = ((d % nz) + nz) % nz; /* 0, 1 or nz - 1 */
  dz = (dz == nz - 1) ? -1 : dz;
  d = (d - dz) / nz;
  int32_t dy = ((d % ny) + ny) % ny;
  dy = (dy == ny - 1) ? -1 : dy;
  const int32_t dx = (d - dy) / ny;
  int64_t gc[3] = {(int64_t)ix + dx + g.offset[0],
                   (int64_t)iy + dy + g.offset[1],
                   (int64_t)iz + dz + g.offset[2]};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (gc[a] < 0 || gc[a] >= g.global_ncell[a]) {
      if (!g.global_periodic[a])
        return -1;
      /* across a periodic face of the whole box: is_inside()'s wrap */
      p.pos[a] += (gc[a] < 0 ? 1. : -1.) * g.box_sides[a];
      gc[a] = gc[a] < 0 ? g.global_ncell[a] - 1 : 0;
    }
  }
  return (gc[0] * g.global_ncell[1] + gc[1]) * g.global_ncell[2] + gc[2];
}

/* FAST: materialise the current position (end of a flight) */
template <bool FULL>
__device__ __forceinline__ void end_flight(Packet<FULL> &p) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
    p.pos[a] = p.pos[a] + p.t * p.dir[a];
  p.t = 0.;
}

#endif
