/*
 * temperature_pipeline.h - TemperatureCalculator::calculate_temperature
 * (src/TemperatureCalculator.cpp:567-931) over the grid as a pipeline of
 * small kernels instead of one: the same functions of device_thermal.h, the
 * same arithmetic in the same order - results equal those of
 * temperature_kernel bit for bit (tests/test_gpu_physics.py) - but
 *
 *   temp_begin_kernel     every cell: cells that need no solve are stored,
 *                         the others get a slot (solve state in global memory)
 *   per secant step, for the slots still iterating:
 *     temp_eval_kernel      one lane per (slot, evaluation at 1.1 T0 / 0.9 T0
 *                           / T0): the ionization balance, the gains, the
 *                           continuum losses, the 13 line-cooling abundances
 *     temp_linecool_kernel  one lane per (slot, evaluation): the ten 5-level
 *                           and three 2-level populations -> the line cooling
 *     temp_secant_kernel    one lane per slot: the step's new temperature;
 *                           converged cells are finished and stored, the rest
 *                           is listed for the next step
 *   temp_finish_kernel    the last few slots, all their remaining steps, one
 *                         wave per slot
 *
 * One kernel holding a whole solve keeps ~400 values alive (256 registers +
 * 137 spilled, 2 waves/SIMD) and its lanes take 1 to ~10 steps each; here
 * every kernel is dense in work that is alike, the line cooling - two thirds
 * of the instructions - runs three evaluations per cell side by side with
 * its own small register budget, and a step's intermediate values (21 doubles
 * per evaluation) travel through HBM.
 */
#ifndef CMI_TEMPERATURE_PIPELINE_H
#define CMI_TEMPERATURE_PIPELINE_H

/* fields of a slot's solve state (SoA: field * capacity + slot) */
enum {
  TS_T0 = 0, TS_GAIN0, TS_LOSS0, TS_H0, TS_HE0, TS_TLAST, TS_HEAT0, TS_HEAT1,
  TS_NTOT, TS_ZMID, TS_NFIELD
};
/* fields of an evaluation (SoA: field * 3 capacity + k * capacity + a) */
enum {
  TE_ABUND = 0, /* 13 */
  TE_T = 13, TE_NE, TE_N, TE_GAIN, TE_LOSS_FF, TE_LOSS_REC, TE_H0, TE_HE0,
  TE_LINES, TE_NFIELD
};

struct TempPipeArgs {
  UpdateArgs u;
  int64_t chunk_first, chunk_count; /* cells of this pass */
  uint32_t capacity;                /* slots */
  uint32_t *slot_cell;              /* cell - chunk_first of a slot */
  double *state;
  int32_t *niter;
  double *eval;
  /* counts[0]: slots handed out; counts[1 + w]: length of list[w] */
  unsigned int *counts;
  uint32_t *list[2];
  int32_t current;   /* which list holds the slots of this step */
  uint32_t nactive;  /* its length (host copy) */
};

__device__ __forceinline__ void
temp_store_cell(const UpdateArgs &a, int64_t cell, double ntot, double T,
                const double (&x)[CMI_NION], const double (&heating)[2]) {
  a.cells.temperature[cell] = T;
#pragma unroll
  for (int i = 0; i < CMI_NION; ++i)
    a.cells.x[i][cell] = x[i];
  (*acc_at(a.cells, CMI_NION, cell)) = heating[0];
  (*acc_at(a.cells, CMI_NION + 1, cell)) = heating[1];
  a.cells.opacity[cell] =
      (ntot > 0.) ? make_double2(ntot * x[ION_H_n], ntot * x[ION_He_n])
                  : make_double2(-1., 0.);
}

__device__ __forceinline__ void temp_load_state(const TempPipeArgs &a,
                                                uint32_t slot,
                                                TemperatureSolve &s) {
  const double *st = a.state + slot;
  const size_t cap = a.capacity;
  s.T0 = st[TS_T0 * cap];
  s.gain0 = st[TS_GAIN0 * cap];
  s.loss0 = st[TS_LOSS0 * cap];
  s.h0 = st[TS_H0 * cap];
  s.he0 = st[TS_HE0 * cap];
  s.Tlast = st[TS_TLAST * cap];
  s.h[0] = st[TS_HEAT0 * cap];
  s.h[1] = st[TS_HEAT1 * cap];
  s.niter = a.niter[slot];
}

__device__ __forceinline__ void temp_store_state(const TempPipeArgs &a,
                                                 uint32_t slot,
                                                 const TemperatureSolve &s) {
  double *st = a.state + slot;
  const size_t cap = a.capacity;
  st[TS_T0 * cap] = s.T0;
  st[TS_GAIN0 * cap] = s.gain0;
  st[TS_LOSS0 * cap] = s.loss0;
  st[TS_H0 * cap] = s.h0;
  st[TS_HE0 * cap] = s.he0;
  st[TS_TLAST * cap] = s.Tlast;
  st[TS_HEAT0 * cap] = s.h[0];
  st[TS_HEAT1 * cap] = s.h[1];
  a.niter[slot] = s.niter;
}

__device__ __forceinline__ CellIntegrals temp_integrals(const UpdateArgs &a,
                                                        int64_t cell) {
  CellIntegrals J;
  J.J = acc_at(a.cells, 0, cell);
  J.stride = a.cells.acc_field_stride;
  J.row = a.cells.acc_cell_stride != 1;
  J.jfac = a.jfac;
  return J;
}

/* the coefficient tables in LDS (read hundreds of times per evaluation) */
__device__ __forceinline__ void temp_stage_tables(const TablesDev *global,
                                                  TablesDev *lds) {
  const uint64_t *src = reinterpret_cast<const uint64_t *>(global);
  uint64_t *dst = reinterpret_cast<uint64_t *>(lds);
  for (unsigned k = threadIdx.x; k < sizeof(TablesDev) / 8; k += blockDim.x)
    dst[k] = src[k];
  __syncthreads();
}

__global__ void __launch_bounds__(CMI_BLOCK)
    temp_begin_kernel(const TempPipeArgs a_in) {
  __shared__ unsigned int s_count[CMI_BLOCK / 64], s_base;
  __shared__ TablesDev lds_tables;
  TempPipeArgs a = a_in;
  temp_stage_tables(a_in.u.model.tables, &lds_tables);
  a.u.model.tables = &lds_tables;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  /* (the trip count is the same for every thread of a workgroup) */
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < a.chunk_count;
       base += stride) {
    const int64_t k = base + threadIdx.x;
    bool solve = false;
    TemperatureSolve s;
    double ntot = 0., zmid = 0.;
    if (k < a.chunk_count) {
      const int64_t c = a.chunk_first + k;
      ntot = a.u.cells.number_density[c];
      double T = a.u.cells.temperature[c];
      const CellIntegrals J = temp_integrals(a.u, c);
      double x[CMI_NION], heating[2];
      heating[0] = (*acc_at(a.u.cells, CMI_NION, c));
      heating[1] = (*acc_at(a.u.cells, CMI_NION + 1, c));
      /* z of the cell midpoint, src/CartesianDensityGrid.hpp:85-89 */
      const int64_t iz = c % a.u.grid.ncell[2];
      zmid = (a.u.grid.anchor[2] +
              a.u.grid.cellside[2] * (iz + a.u.grid.offset[2])) +
             0.5 * a.u.grid.cellside[2];
      solve = temperature_begin(a.u.model, J, a.u.hfac, ntot, T, heating, x, s);
      if (solve && !temperature_goes_on(a.u.model, s)) {
        /* (no iterations allowed: the cell keeps its fractions) */
#pragma unroll
        for (int i = 0; i < CMI_NION; ++i)
          x[i] = a.u.cells.x[i][c];
        temperature_end(a.u.model, ntot, J, s, T, heating, x);
        solve = false;
      }
      if (!solve)
        temp_store_cell(a.u, c, ntot, T, x, heating);
    }
    const unsigned int slot =
        block_reserve(solve, a.counts, s_count, &s_base);
    if (solve) {
      a.slot_cell[slot] = (uint32_t)k;
      temp_store_state(a, slot, s);
      a.state[(size_t)TS_NTOT * a.capacity + slot] = ntot;
      a.state[(size_t)TS_ZMID * a.capacity + slot] = zmid;
      a.list[0][slot] = slot;
    }
  }
}

/* cooling_and_heating_balance (device_thermal.h) without its line cooling:
 * the same statements in the same order; what the line cooling is added to
 * comes back in pieces (loss = lines * n; loss += free-free; loss +=
 * recombination; then the clamps - temp_secant_kernel) */
__device__ inline void
balance_without_lines(const ModelDev &m, double &h0, double &he0, double &gain,
                      double &ne_out, double &loss_ff, double &loss_rec,
                      double T, double n, double midpoint_z,
                      const CellIntegrals &j, const double h[2], double *abund,
                      size_t abund_stride) {
  enum { NI = 0, NII, OI, OII, OIII, NeIII, SII, SIII, CII, CIII, NIII, NeII,
         SIV };
  double x[CMI_NION];
  const double alphaH = cmi_recombination_rate(m, ION_H_n, T);
  const double alphaHe = cmi_recombination_rate(m, ION_He_n, T);
  const double T4 = T * 1.e-4;
  const double sqrtT = sqrt(T);
  const double logT = log(T);
  const double AHe = m.abundance[0];

  cmi_ionization_states_hydrogen_helium(alphaH, alphaHe, j(ION_H_n),
                                        j(ION_He_n), n, AHe, T, h0, he0);
  const double ne = n * (1. - h0 + AHe * (1. - he0));
  const double nhp = n * (1. - h0);
  const double nhep = (1. - he0) * n * AHe;
  const double nenhp = ne * nhp;
  const double nenhep = ne * nhep;

  gain = n * (h[0] * h0 + h[1] * AHe * he0);
  const double alpha_e_2sP =
      4.17e-20 * exp(-0.861 * (logT - 9.210340371976184));
  const double pHots = 1. / (1. + 77. * he0 / (sqrtT * h0));
  gain += pHots * 1.21765423e-18 * alpha_e_2sP * nenhep;
  gain += 1.5e-37 * n * ne * m.pahfac;
  double heatcr = 0.;
  if (m.crfac > 0.) {
    heatcr = m.crfac * 1.2e-25 / sqrt(ne);
    if (m.crscale > 0.)
      heatcr *= exp(-fabs(midpoint_z) / m.crscale);
  }
  gain += heatcr;

  const double nh0 = n * h0;
  const double nhe0 = n * he0 * AHe;
  cmi_ionization_states_metals(m, j, ne, T, T4, nh0, nhe0, nhp, x);

  const double AC = m.abundance[1], AN = m.abundance[2], AO = m.abundance[3],
               ANe = m.abundance[4], AS = m.abundance[5];
#define AB(k) abund[(size_t)(k)*abund_stride]
  AB(CII) = AC * (1. - x[ION_C_p1] - x[ION_C_p2]);
  AB(CIII) = AC * x[ION_C_p1];
  AB(NI) = AN * (1. - x[ION_N_n] - x[ION_N_p1] - x[ION_N_p2]);
  AB(NII) = AN * x[ION_N_n];
  AB(NIII) = AN * x[ION_N_p1];
  AB(OI) = AO * (1. - x[ION_O_n] - x[ION_O_p1]);
  AB(OII) = AO * x[ION_O_n];
  AB(OIII) = AO * x[ION_O_p1];
  AB(NeII) = ANe * x[ION_Ne_n];
  AB(NeIII) = ANe * x[ION_Ne_p1];
  AB(SII) = AS * (1. - x[ION_S_p1] - x[ION_S_p2] - x[ION_S_p3]);
  AB(SIII) = AS * x[ION_S_p1];
  AB(SIV) = AS * x[ION_S_p2];
#undef AB

  ne_out = ne;
  const double c = 5.5 - logT;
  const double gff = 1.1 + 0.34 * exp(-c * c / 3.);
  loss_ff = 1.42e-40 * gff * sqrtT * (nenhp + nenhep);
  const double Lhp =
      2.85e-40 * nenhp * sqrtT * (5.914 - 0.5 * logT + 0.01184 * cbrt(T));
  const double Lhep = 1.55e-39 * nenhep * exp(0.3647 * logT);
  loss_rec = Lhp + Lhep;
}

/* lane t: evaluation k = t / nactive of the a = t % nactive -th active slot */
__global__ void __launch_bounds__(CMI_BLOCK)
    temp_eval_kernel(const TempPipeArgs a_in) {
  __shared__ TablesDev lds_tables;
  TempPipeArgs a = a_in;
  temp_stage_tables(a_in.u.model.tables, &lds_tables);
  a.u.model.tables = &lds_tables;
  const uint64_t total = 3ull * a.nactive;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const size_t estride = 3 * (size_t)a.capacity;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const uint32_t k = (uint32_t)(t / a.nactive);
    const uint32_t i = (uint32_t)(t % a.nactive);
    const uint32_t slot = a.list[a.current][i];
    const size_t cap = a.capacity;
    const double T0 = a.state[(size_t)TS_T0 * cap + slot];
    const double ntot = a.state[(size_t)TS_NTOT * cap + slot];
    const double zmid = a.state[(size_t)TS_ZMID * cap + slot];
    double h[2];
    h[0] = a.state[(size_t)TS_HEAT0 * cap + slot];
    h[1] = a.state[(size_t)TS_HEAT1 * cap + slot];
    /* temperature_step's three temperatures, in its order */
    const double T1 = 1.1 * T0;
    const double Tk = (k == 0) ? T1 : ((k == 1) ? 0.9 * T0 : T0);
    const CellIntegrals J =
        temp_integrals(a.u, a.chunk_first + a.slot_cell[slot]);
    double *e = a.eval + (size_t)k * cap + i;
    double h0, he0, gain, ne, loss_ff, loss_rec;
    balance_without_lines(a.u.model, h0, he0, gain, ne, loss_ff, loss_rec, Tk,
                          ntot, zmid, J, h, e + (size_t)TE_ABUND * estride,
                          estride);
    e[(size_t)TE_T * estride] = Tk;
    e[(size_t)TE_NE * estride] = ne;
    e[(size_t)TE_N * estride] = ntot;
    e[(size_t)TE_GAIN * estride] = gain;
    e[(size_t)TE_LOSS_FF * estride] = loss_ff;
    e[(size_t)TE_LOSS_REC * estride] = loss_rec;
    e[(size_t)TE_H0 * estride] = h0;
    e[(size_t)TE_HE0 * estride] = he0;
  }
}

/* waves per SIMD temp_linecool_kernel is built for, with the line cooling
 * inlined (measured, ms per launch of a 256^3 lexington update: the function
 * called, 246 VGPRs / 2 waves 7.10; inlined at 2 / 3 / 4 waves per SIMD 6.90 /
 * 6.25 / 8.70) */
#ifndef CMI_LINECOOL_WAVES
#define CMI_LINECOOL_WAVES 3
#endif
__global__ void __launch_bounds__(CMI_BLOCK, CMI_LINECOOL_WAVES)
    temp_linecool_kernel(const TempPipeArgs a) {
  __shared__ LineCoolingDev lds_lc;
  {
    const uint64_t *src =
        reinterpret_cast<const uint64_t *>(&a.u.model.tables->lc);
    uint64_t *dst = reinterpret_cast<uint64_t *>(&lds_lc);
    for (unsigned k = threadIdx.x; k < sizeof(LineCoolingDev) / 8;
         k += blockDim.x)
      dst[k] = src[k];
    __syncthreads();
  }
  const uint64_t total = 3ull * a.nactive;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const size_t estride = 3 * (size_t)a.capacity;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += stride) {
    const uint32_t k = (uint32_t)(t / a.nactive);
    const uint32_t i = (uint32_t)(t % a.nactive);
    double *e = a.eval + (size_t)k * a.capacity + i;
    e[(size_t)TE_LINES * estride] =
        line_cooling_inlined(lds_lc, e[(size_t)TE_T * estride],
                     e[(size_t)TE_NE * estride],
                     e + (size_t)TE_ABUND * estride, (int)estride);
  }
}

/* the rest of temperature_step after its three evaluations (:716-745) */
__device__ __forceinline__ void
temperature_step_finish(const ModelDev &m, TemperatureSolve &s, double T0,
                        double gain1, double loss1, double gain2,
                        double loss2) {
  const double logtt = log(1.1 / 0.9);
  const double T1 = 1.1 * T0;
  double expgain;
  if (gain2 > 0.)
    expgain = (gain1 > 0.) ? log(gain1 / gain2) : -99.;
  else
    expgain = (gain1 > 0.) ? 99. : 0.;
  double exploss;
  if (loss2 > 0.)
    exploss = (loss1 > 0.) ? log(loss1 / loss2) : -99.;
  else
    exploss = (loss1 > 0.) ? 99. : 0.;
  const double expdiff = expgain - exploss;
  if (s.gain0 > 0. && expdiff != 0.)
    s.T0 = T0 * pow(s.loss0 / s.gain0, logtt / expdiff);
  else
    s.T0 = T1;
  if (s.T0 < m.t_min_ionized) {
    s.T0 = 500.;
    s.h0 = 1.;
    s.he0 = 1.;
    s.gain0 = 1.;
    s.loss0 = 1.;
  }
  if (s.T0 > 1.e10) {
    s.T0 = 1.e10;
    s.h0 = 1.e-10;
    s.he0 = 1.e-10;
    s.gain0 = 1.;
    s.loss0 = 1.;
  }
}

/* (measured, ms per launch of a 256^3 lexington update: the compiler's choice
 * 1.50, built for 3 waves per SIMD 1.34, for 4 1.44) */
#ifndef CMI_SECANT_WAVES
#define CMI_SECANT_WAVES 3
#endif
__global__ void __launch_bounds__(CMI_BLOCK, CMI_SECANT_WAVES)
    temp_secant_kernel(const TempPipeArgs a_in) {
  __shared__ unsigned int s_count[CMI_BLOCK / 64], s_base;
  __shared__ TablesDev lds_tables;
  TempPipeArgs a = a_in;
  temp_stage_tables(a_in.u.model.tables, &lds_tables);
  a.u.model.tables = &lds_tables;
  const int next = 1 - a.current;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const size_t estride = 3 * (size_t)a.capacity;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x; base < a.nactive;
       base += stride) {
    const uint64_t i = base + threadIdx.x;
    bool again = false;
    uint32_t slot = 0;
    if (i < a.nactive) {
      slot = a.list[a.current][i];
      TemperatureSolve s;
      temp_load_state(a, slot, s);
      const double T0 = s.T0;
      ++s.niter;
      s.Tlast = T0;
      double gain[3], loss[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double *e = a.eval + (size_t)k * a.capacity + i;
        /* loss = line_cooling * n; += free-free; += recombination; clamps */
        double l = e[(size_t)TE_LINES * estride] * e[(size_t)TE_N * estride];
        l += e[(size_t)TE_LOSS_FF * estride];
        l += e[(size_t)TE_LOSS_REC * estride];
        loss[k] = fmax(l, 0.);
        gain[k] = fmax(e[(size_t)TE_GAIN * estride], 0.);
      }
      {
        const double *e = a.eval + (size_t)2 * a.capacity + i;
        s.h0 = e[(size_t)TE_H0 * estride];
        s.he0 = e[(size_t)TE_HE0 * estride];
      }
      s.gain0 = gain[2];
      s.loss0 = loss[2];
      temperature_step_finish(a.u.model, s, T0, gain[0], loss[0], gain[1],
                              loss[1]);
      again = temperature_goes_on(a.u.model, s);
      if (again) {
        temp_store_state(a, slot, s);
      } else {
        const int64_t c = a.chunk_first + a.slot_cell[slot];
        const double ntot = a.state[(size_t)TS_NTOT * a.capacity + slot];
        const CellIntegrals J = temp_integrals(a.u, c);
        double T, heating[2], x[CMI_NION];
        temperature_end(a.u.model, ntot, J, s, T, heating, x);
        temp_store_cell(a.u, c, ntot, T, x, heating);
      }
    }
    const unsigned int q =
        block_reserve(again, a.counts + 1 + next, s_count, &s_base);
    if (again)
      a.list[next][q] = slot;
  }
}

/* The last slots - a handful of cells whose secant iteration wanders until
 * t_max_iterations ends it; as pipeline steps they would cost a hundred rounds
 * of three nearly empty launches each - are finished by one launch. A slot's
 * remaining steps are ONE dependent chain, so what counts is the length of a
 * step, not the lanes it occupies: a slot gets a whole wave. Lanes 16 k ..
 * 16 k + 15 hold evaluation k of the step (at 1.1 T0, 0.9 T0, T0; the fourth
 * quarter repeats the third); inside a quarter
 *   - the hydrogen / helium balance and the gains run on every lane alike
 *     (the same numbers 16 times: nothing to exchange),
 *   - lane r < 12 takes the ratio of metal ion r (cmi_metal_ratio: its
 *     recombination and charge transfer fits), the twelve ratios are then
 *     read by every lane,
 *   - lane r < 10 solves the level populations of five-level ion r, lanes 10
 *     .. 12 the two-level ions; the thirteen cooling terms are summed by
 *     every lane in line_cooling()'s order.
 * Every number is computed by the statements of cooling_and_heating_balance
 * (device_thermal.h) in their order - the results equal temperature_kernel's
 * bit for bit - but a step is as long as one ion's fits, not thirteen ions':
 * round 4's form (four lanes per slot, an evaluation whole on one lane) spent
 * 7.2 ms per 256^3 update on a few dozen cells while the GPU idled. */
__global__ void __launch_bounds__(CMI_BLOCK)
    temp_finish_kernel(const TempPipeArgs a_in) {
  __shared__ TablesDev lds_tables;
  TempPipeArgs a = a_in;
  temp_stage_tables(a_in.u.model.tables, &lds_tables);
  a.u.model.tables = &lds_tables;
  const ModelDev &m = a.u.model;
  const int lane = threadIdx.x & 63;
  const int role = lane & 15;
  const int quarter = lane & ~15;        /* first lane of the evaluation */
  const int k = (lane >> 4) < 3 ? (lane >> 4) : 2;
  const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w >= a.nactive)
    return; /* (the whole wave; no barrier follows) */
  const uint32_t slot = a.list[a.current][w];
  TemperatureSolve s;
  temp_load_state(a, slot, s);
  const double n = a.state[(size_t)TS_NTOT * a.capacity + slot];
  const double midpoint_z = a.state[(size_t)TS_ZMID * a.capacity + slot];
  const int64_t cell = a.chunk_first + a.slot_cell[slot];
  const CellIntegrals j = temp_integrals(a.u, cell);
  enum { NI = 0, NII, OI, OII, OIII, NeIII, SII, SIII, CII, CIII, NIII, NeII,
         SIV };
  const double AHe = m.abundance[0];
  const double AC = m.abundance[1], AN = m.abundance[2], AO = m.abundance[3],
               ANe = m.abundance[4], AS = m.abundance[5];
  /* (listed: its last step left it unconverged) */
  do {
    const double T0 = s.T0;
    const double T = (k == 0) ? 1.1 * T0 : ((k == 1) ? 0.9 * T0 : T0);
    /* ---- cooling_and_heating_balance(m, h0, he0, gain, loss, T, n, ...) */
    double h0, he0, gain, loss;
    const double alphaH = cmi_recombination_rate(m, ION_H_n, T);
    const double alphaHe = cmi_recombination_rate(m, ION_He_n, T);
    const double T4 = T * 1.e-4;
    const double sqrtT = sqrt(T);
    const double logT = log(T);
    cmi_ionization_states_hydrogen_helium(alphaH, alphaHe, j(ION_H_n),
                                          j(ION_He_n), n, AHe, T, h0, he0);
    const double ne = n * (1. - h0 + AHe * (1. - he0));
    const double nhp = n * (1. - h0);
    const double nhep = (1. - he0) * n * AHe;
    const double nenhp = ne * nhp;
    const double nenhep = ne * nhep;
    gain = n * (s.h[0] * h0 + s.h[1] * AHe * he0);
    const double alpha_e_2sP =
        4.17e-20 * exp(-0.861 * (logT - 9.210340371976184));
    const double pHots = 1. / (1. + 77. * he0 / (sqrtT * h0));
    gain += pHots * 1.21765423e-18 * alpha_e_2sP * nenhep;
    gain += 1.5e-37 * n * ne * m.pahfac;
    double heatcr = 0.;
    if (m.crfac > 0.) {
      heatcr = m.crfac * 1.2e-25 / sqrt(ne);
      if (m.crscale > 0.)
        heatcr *= exp(-fabs(midpoint_z) / m.crscale);
    }
    gain += heatcr;
    const double nh0 = n * h0;
    const double nhe0 = n * he0 * AHe;
    /* cmi_ionization_states_metals: one ion per lane */
    double my_ratio = 0.;
    if (role < 12)
      my_ratio = cmi_metal_ratio(m, j, ION_C_p1 + role, ne, T, T4, nh0, nhe0,
                                 nhp);
    double ratio[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
      ratio[i] = __shfl(my_ratio, quarter + i, 64);
    double x[CMI_NION];
    cmi_metal_fractions(ratio, x);
    double ab[13];
    ab[CII] = AC * (1. - x[ION_C_p1] - x[ION_C_p2]);
    ab[CIII] = AC * x[ION_C_p1];
    ab[NI] = AN * (1. - x[ION_N_n] - x[ION_N_p1] - x[ION_N_p2]);
    ab[NII] = AN * x[ION_N_n];
    ab[NIII] = AN * x[ION_N_p1];
    ab[OI] = AO * (1. - x[ION_O_n] - x[ION_O_p1]);
    ab[OII] = AO * x[ION_O_n];
    ab[OIII] = AO * x[ION_O_p1];
    ab[NeII] = ANe * x[ION_Ne_n];
    ab[NeIII] = ANe * x[ION_Ne_p1];
    ab[SII] = AS * (1. - x[ION_S_p1] - x[ION_S_p2] - x[ION_S_p3]);
    ab[SIII] = AS * x[ION_S_p1];
    ab[SIV] = AS * x[ION_S_p2];
    double my_ab = 0.;
#pragma unroll
    for (int i = 0; i < 13; ++i)
      if (role == i)
        my_ab = ab[i];
    /* line_cooling(lc, T, ne, abund): one ion per lane */
    const LineCoolingDev &lc = m.tables->lc;
    const double kb = CMI_BOLTZMANN;
    double term = 0.;
    if (ne != 0.) {
      const double prefactor = lc.prefactor * ne / sqrt(T);
      const double Tinv = 1. / T;
      const double logT_lc = log(T);
      if (role < CMI_LC_NFIVE_DEV)
        term = my_ab * kb *
               lc_five_level_cooling(lc, role, prefactor, T, Tinv, logT_lc);
      else if (role < CMI_LC_NFIVE_DEV + CMI_LC_NTWO_DEV) {
        const int i = role - CMI_LC_NFIVE_DEV;
        term = my_ab * kb * lc.two_energy[i] * lc.two_A[i] *
               lc_two_level_cooling(lc, i, prefactor, T, Tinv, logT_lc);
      }
    }
    double cooling = 0.;
#pragma unroll
    for (int e = 0; e < CMI_LC_NFIVE_DEV + CMI_LC_NTWO_DEV; ++e)
      cooling += __shfl(term, quarter + e, 64);
    loss = ((ne == 0.) ? 1.e-99 : cooling) * n;
    const double c = 5.5 - logT;
    const double gff = 1.1 + 0.34 * exp(-c * c / 3.);
    loss += 1.42e-40 * gff * sqrtT * (nenhp + nenhep);
    const double Lhp =
        2.85e-40 * nenhp * sqrtT * (5.914 - 0.5 * logT + 0.01184 * cbrt(T));
    const double Lhep = 1.55e-39 * nenhep * exp(0.3647 * logT);
    loss += Lhp + Lhep;
    loss = fmax(loss, 0.);
    gain = fmax(gain, 0.);
    /* ---- temperature_step's bookkeeping, on every lane alike */
    const double gain1 = __shfl(gain, 0, 64);
    const double loss1 = __shfl(loss, 0, 64);
    const double gain2 = __shfl(gain, 16, 64);
    const double loss2 = __shfl(loss, 16, 64);
    ++s.niter;
    s.Tlast = T0;
    s.h0 = __shfl(h0, 32, 64);
    s.he0 = __shfl(he0, 32, 64);
    s.gain0 = __shfl(gain, 32, 64);
    s.loss0 = __shfl(loss, 32, 64);
    temperature_step_finish(m, s, T0, gain1, loss1, gain2, loss2);
  } while (temperature_goes_on(m, s));
  if (lane == 0) {
    double T, heating[2], x[CMI_NION];
    temperature_end(m, n, j, s, T, heating, x);
    temp_store_cell(a.u, cell, n, T, x, heating);
  }
}

#endif
