/*
 * engine.hip - host side of the MI355X photoionization engine and its C ABI
 * (include/cmi_gpu.h). Owns the device memory, lowers the plugin descriptors
 * into device tables and launches the kernels of kernels.h on one HIP stream.
 */
#include "../../include/cmi_gpu.h"

#include "atomic_data.h"
#include "linecooling_data.h"
#include "kernels.h"
#include "sort.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                          \
  do {                                                                         \
    hipError_t err__ = (expr);                                                 \
    if (err__ != hipSuccess)                                                   \
      return fail(CMI_GPU_EDEVICE, "%s failed: %s (%s:%d)", #expr,             \
                  hipGetErrorString(err__), __FILE__, __LINE__);               \
  } while (0)

struct EventPair {
  hipEvent_t start = nullptr, stop = nullptr;
  uint64_t packets = 0; /* flights started by the launch */
};

} // namespace

struct cmi_gpu_engine {
  cmi_gpu_config config;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cu = 0;

  GridDev grid;
  ModelDev model;
  CellsDev cells;
  int64_t ncell = 0;

  /* device allocations */
  double *state_block = nullptr;   /* n, T, x[14] : 16 fields */
  double *acc_block = nullptr;     /* J[14], heating[2] : 16 fields */
  bool own_acc = false;
  double2 *opacity = nullptr;
  CountersDev *counters = nullptr;
  TablesDev *tables = nullptr;
  TablesDev *host_tables = nullptr; /* host copy, for host-side tabulation */
  SpectraDev *spectra = nullptr;
  bool spectra_dirty = true; /* cross sections / spectrum changed */
  /* caller-supplied tables (cmi_gpu_set_*_table): [0], [1] the spectra of the
   * discrete / continuous sources, [2] cross sections, [3] recombination
   * rates - device copy {x[n], y[rows][n]} and the host's */
  double *user_table[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<double> user_table_host[4];
  /* counters the host needs between launches: pinned host memory, mapped */
  unsigned int *mailbox = nullptr, *mailbox_dev = nullptr;
  double *source_position = nullptr;
  double *source_cumulative = nullptr;
  std::vector<double> source_position_host;
  double discrete_luminosity = 0., continuous_luminosity = 0.;
  bool have_continuous_spectrum = false;

  bool have_sources = false, have_spectrum = false, have_xsec = false,
       have_recomb = false, have_cells = false;
  bool full_ions = false; /* transport carries all 14 cross sections */
  cmi_gpu_temperature_params tparams;

  /* direction sort of the packet order (keys/ids double buffered) */
  unsigned int *select_count = nullptr; /* block_select_kernel's counter */
  uint32_t *select_ids = nullptr;       /* ... and its list */
  uint64_t select_capacity = 0;
  double *select_rows = nullptr; /* emission rows of a selection, by id */
  uint64_t select_rows_capacity = 0;
  uint32_t *sort_keys[2] = {nullptr, nullptr};
  uint32_t *sort_ids[2] = {nullptr, nullptr};
  void *sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  uint64_t sort_capacity = 0;

  /* re-emission queues (ping-pong) */
  double *queue_block = nullptr;
  uint64_t queue_capacity = 0;
  QueueDev ended_queue, ready_queue;
  /* decomposed grids: caller-owned buffer for the flights that leave */
  double *export_rows = nullptr;
  uint64_t export_capacity = 0;
  unsigned int *export_count = nullptr;
  bool own_export_rows = false;
  double *import_rows = nullptr; /* staging for flights given in host memory */
  uint64_t import_capacity = 0;
  unsigned int *queue_counts = nullptr; /* [2]: ended, ready */
  /* tile rounds: two sets of flight rows (in / out of a round), the plan */
  char *tile_block = nullptr;
  uint64_t tile_capacity = 0;
  bool tile_has_weights = false;
  FlightRowsDev tile_rows[2];
  uint32_t *tile_iota = nullptr;
  TileItemDev *tile_items = nullptr;
  uint32_t *tile_begin = nullptr;      /* [ntiles + 2] */
  /* counting sort of the slots by tile (null: too many tiles, radix sort) */
  uint32_t *tile_blockhist = nullptr, *tile_total = nullptr;
  uint32_t *tile_new_slots = nullptr; /* slots filled by a round's re-emissions */
  /* SpectrumTrackers: counted while enabled (the exact marcher then) */
  TrackersDev trackers = {};
  bool trackers_enabled = false;
  /* PAD transport kernels: n x_H inside a layer of ghost cells */
  double *pad_H = nullptr;
  /* the temperature solve as a pipeline (temperature_pipeline.h) */
  char *temp_pipe_block = nullptr;
  uint32_t temp_pipe_capacity = 0;
  unsigned int *temp_pipe_counts = nullptr;
  uint32_t *tile_ended_slot = nullptr; /* slot of each absorption record */
  uint32_t *tile_ended_pos = nullptr;  /* its position in the next round */
  uint32_t *tile_slot_of[2] = {nullptr, nullptr}; /* position -> slot */
  unsigned int *tile_absorbed_count = nullptr; /* [units of work] */
  unsigned int *tile_absorbed_before = nullptr; /* their running totals */
  unsigned int *tile_counts = nullptr; /* [8]: rows, live, nitems, next item,
                                          absorbed */
  uint64_t tile_rounds_run = 0;
  /* hydrogen-only runs: something other than a transport step may have
   * written one of the accumulator fields such a run never adds to (a field
   * upload, a caller-owned block, a failed clear at a layout switch):
   * cmi_gpu_reset_grid then clears the whole block once */
  bool acc_block_dirty = false;

  struct Tuning {
    bool sort_packets = true;
    int sort_tau_bits = -1;  /* tau classes per direction bin; -1 = auto */
    /* direction bits of the sort key (2 x 11 at most); -1 = auto: one or two
     * fewer than 22 where that saves the radix sort a pass of 8 bits */
    int sort_dir_bits = -1;
    int aggregate = CMI_AGG_BLOCK;      /* first generation (sorted bundles) */
    int aggregate_reemit = CMI_AGG_NONE; /* later generations (random flights) */
    int refill_threshold = CMI_REFILL_THRESHOLD;
    uint32_t chunk = 64;
    int max_blocks_per_cu = 8;
    uint64_t max_packets_per_launch = 1ull << 27;
    int exp_no_atomics = 0;
    bool exact_dda = false;
    bool reemit_passes = true;
    int refill_threshold_reemit = 32;
    /* -1: 4096 on a whole grid; 262144 on a block of a decomposed grid, whose
     * hand-over rounds are many launches of few flights - each costs the
     * latency of its longest flight - (measured on config 5's workload on one
     * GPU, a device per block: 4096 / 32768 / 262144 / 2e6 -> 52 / 47 / 44 /
     * 49 ms per iteration; the blocks' calls in series 360 -> 325 ms) */
    int64_t reemit_inline_below = -1;
    int reemit_max_passes = 12;
    /* later generations in tile rounds (tile_kernels.h) instead of passes of
     * the transport kernel; below tile_min_flights flights the transport
     * kernel finishes them with single atomics */
    bool tile_rounds = true;
    uint64_t tile_min_flights = 100000;
    int tile_min_per_item = -1; /* flights per unit of work; -1 = auto */
    int tile_refill_threshold = 48;
    int tile_max_rounds = 1000;
    bool tile_counting_sort = true; /* false: rocPRIM radix sort of the slots */
    /* multi-ion runs: the cross sections of re-emitted flights in a kernel of
     * their own (flight_weights_kernel) instead of inside the interaction
     * kernels */
    bool defer_weights = true;
    /* multi-ion runs: the emission physics of the new packets (spectrum,
     * cross sections, optical depth) in the sort-key kernel, read back by the
     * transport kernel (shoot_kernel<..., PRE>) */
    bool pre_emission = true;
    /* hydrogen-only first generation on a whole non-periodic grid: march
     * through the padded records (shoot_kernel<..., PAD>) */
    bool pad_march = true;
    /* sorted first generation: the blocks of an XCD take neighbouring
     * positions of the packet order */
    bool xcd_remap = false;
    /* the temperature solve as a pipeline of kernels (0: one kernel) */
    bool temperature_pipeline = true;
    /* ... whose last slots one launch finishes (temp_finish_kernel: a wave
     * per slot; measured at 256^3, ms per update with 32768 / 8192 / 2048 /
     * 512: 51.2 / 50.7 / 50.7 / 49.8 - the wide steps are the cheaper way
     * while many slots are left) */
    uint32_t temperature_finish_slots = 1024;
    /* the live rows are copied into fresh rows, in tile order, once the
     * flights are spread over this many slots per flight; 0: never; -1: 2 for
     * multi-ion transport (two rows per visit, 25 GB at 1e8 packets: a sparse
     * footprint costs more than the copies), never for hydrogen-only */
    int tile_compact_ratio = -1;
    /* the first generation parks an absorbed packet at the place of its
     * position in the launch's order (no queue counter) */
    bool park_in_place = true;
    /* a block of a decomposed grid picks its own packets out of a launch's
     * ids before keys, sort and transport (block_select_kernel); 0: every
     * packet goes through them and the transport kernel drops the others */
    bool block_select = true;
    /* ... and flies them with the kernels built for a whole grid's first
     * generation (padded march / pre-computed emission rows) */
    bool block_first_kernels = true;
  } tune;

  /* device timing (HIP events around launches) is opt-in: set_tuning
   * ("timing", 1). Events are recycled through a pool; without timing a run
   * of any length creates none. */
  bool timing = false;
  /* with timing: the DDA step counter after every transport launch */
  unsigned long long *launch_steps = nullptr;
  std::vector<EventPair> shoot_events, update_events, kernel_events;
  std::vector<EventPair> event_pool;
};

namespace {

#define CMI_MAX_TIMED_LAUNCHES 65536
/* start / stop of a timed region on the engine's stream; no-ops unless timing
 * is on */
int timer_begin(cmi_gpu_engine *e, EventPair &ev) {
  ev = EventPair();
  if (!e->timing)
    return CMI_GPU_OK;
  if (!e->event_pool.empty()) {
    ev = e->event_pool.back();
    e->event_pool.pop_back();
  } else {
    HIP_TRY(hipEventCreate(&ev.start));
    hipError_t err = hipEventCreate(&ev.stop);
    if (err != hipSuccess) {
      (void)hipEventDestroy(ev.start);
      HIP_TRY(err);
    }
  }
  hipError_t err = hipEventRecord(ev.start, e->stream);
  if (err != hipSuccess) {
    e->event_pool.push_back(ev);
    HIP_TRY(err);
  }
  return CMI_GPU_OK;
}

/* A few counters from device memory into the engine's mailbox - pinned host
 * memory the device writes directly - so that reading them costs a tiny
 * kernel and a stream synchronisation, not a staged copy. */
__global__ void mailbox_kernel(const unsigned int *src, unsigned int *dst,
                               int n) {
  if ((int)threadIdx.x < n)
    dst[threadIdx.x] = src[threadIdx.x];
}
/* the steps counted so far (summed over the shards) */
__global__ void snapshot_steps_kernel(const CountersDev *counters,
                                      unsigned long long *dst) {
  unsigned long long sum = 0;
  for (int k = threadIdx.x; k < CMI_COUNTER_SHARDS; k += 64)
    sum += counters[k].nsteps;
  for (int off = 32; off > 0; off >>= 1)
    sum += __shfl_down(sum, off, 64);
  if (threadIdx.x == 0)
    *dst = sum;
}

/* all shards of the counters, added up (shard k by thread k % 256, then the
 * 256 partial sums in a fixed order: the same result every time) into the
 * engine's mailbox - pinned host memory the device writes directly; a staged
 * copy of the 64 KB of shards into pageable memory cost 0.25 ms */
__global__ void __launch_bounds__(256)
    counters_sum_kernel(const CountersDev *shards, CountersDev *out) {
  __shared__ CountersDev partial[256];
  CountersDev mine = CountersDev();
  for (int k = threadIdx.x; k < CMI_COUNTER_SHARDS; k += 256) {
    const CountersDev c = shards[k];
    mine.totweight += c.totweight;
    for (int i = 0; i < 4; ++i)
      mine.typecount[i] += c.typecount[i];
    mine.nsteps += c.nsteps;
    mine.natomics += c.natomics;
    mine.nwavesteps += c.nwavesteps;
  }
  partial[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    CountersDev sum = CountersDev();
    for (int t = 0; t < 256; ++t) {
      sum.totweight += partial[t].totweight;
      for (int i = 0; i < 4; ++i)
        sum.typecount[i] += partial[t].typecount[i];
      sum.nsteps += partial[t].nsteps;
      sum.natomics += partial[t].natomics;
      sum.nwavesteps += partial[t].nwavesteps;
    }
    *out = sum;
  }
}

static int ensure_mailbox(cmi_gpu_engine *e) {
  if (!e->mailbox) {
    HIP_TRY(hipHostMalloc(&e->mailbox, 16 * sizeof(unsigned int),
                          hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&e->mailbox_dev, e->mailbox, 0));
  }
  return CMI_GPU_OK;
}

static int download_counters(cmi_gpu_engine *e, CountersDev &sum) {
  {
    int rc = ensure_mailbox(e);
    if (rc)
      return rc;
  }
  static_assert(sizeof(CountersDev) <= 16 * sizeof(unsigned int),
                "the counters fit the mailbox");
  counters_sum_kernel<<<1, 256, 0, e->stream>>>(
      e->counters, reinterpret_cast<CountersDev *>(e->mailbox_dev));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  memcpy(&sum, (const void *)e->mailbox, sizeof sum);
  return CMI_GPU_OK;
}

int timer_end(cmi_gpu_engine *e, std::vector<EventPair> &list, EventPair &ev,
              uint64_t packets) {
  if (!ev.start)
    return CMI_GPU_OK;
  ev.packets = packets;
  hipError_t err = hipEventRecord(ev.stop, e->stream);
  if (err != hipSuccess) {
    e->event_pool.push_back(ev);
    HIP_TRY(err);
  }
  /* a caller that never reads its timings does not grow without bound */
  if (list.size() >= CMI_MAX_TIMED_LAUNCHES) {
    e->event_pool.insert(e->event_pool.end(), list.begin(), list.end());
    list.clear();
  }
  if (&list == &e->kernel_events) {
    if (!e->launch_steps)
      HIP_TRY(hipMalloc(&e->launch_steps,
                        sizeof(unsigned long long) * CMI_MAX_TIMED_LAUNCHES));
    snapshot_steps_kernel<<<1, 64, 0, e->stream>>>(
        e->counters, e->launch_steps + list.size());
    HIP_TRY(hipGetLastError());
  }
  list.push_back(ev);
  return CMI_GPU_OK;
}

void release_events(cmi_gpu_engine *e, std::vector<EventPair> &list) {
  e->event_pool.insert(e->event_pool.end(), list.begin(), list.end());
  list.clear();
}

double eV_to_Hz(double eV) {
  /* UnitConverter::to_SI<QUANTITY_FREQUENCY>(eV, "eV"),
   * src/UnitConverter.hpp:156-159,266-300 */
  return eV * CMI_ELECTRONVOLT * (1. / CMI_PLANCK) / 1.;
}

double *field_pointer(cmi_gpu_engine *e, int field) {
  if (field < 0 || field >= CMI_GPU_NFIELD)
    return nullptr;
  if (field < CMI_GPU_FIELD_MEAN_INTENSITY)
    return e->state_block + (int64_t)field * e->ncell;
  const int f = field - CMI_GPU_FIELD_MEAN_INTENSITY;
  if (e->cells.acc_cell_stride != 1) /* [ncell][16], rows in threshold order */
    return e->acc_block + cmi_acc_column(f);
  return e->acc_block + (int64_t)f * e->cells.acc_field_stride;
}

/* element stride of a field: 1 for the state fields, the accumulator layout's
 * cell stride for the accumulator fields */
int64_t field_stride(cmi_gpu_engine *e, int field) {
  return field < CMI_GPU_FIELD_MEAN_INTENSITY ? 1 : e->cells.acc_cell_stride;
}

/* lower the generated raw tables into the device layout, applying the unit
 * conversions of the reference constructors */
void build_tables(TablesDev &t) {
  memset(&t, 0, sizeof t);
  /* src/VernerCrossSections.cpp:36-154 */
  const double eV_to_Hz_fac = CMI_ELECTRONVOLT / CMI_PLANCK;
  static_assert(CMI_VERNER_NTERM == CMI_VERNER_NTERM_DEV, "term count");
  for (int i = 0; i < CMI_VERNER_NTERM; ++i) {
    const cmi_verner_term &raw = cmi_verner_terms[i];
    VernerTermDev &d = t.verner[i];
    d.ion = raw.ion;
    d.shell = raw.shell;
    d.ninn = raw.ninn;
    d.ntot = raw.ntot;
    const double E_th = raw.A[0], E_0 = raw.A[1], sigma_0 = raw.A[2],
                 y_a = raw.A[3], P = raw.A[4], y_w = raw.A[5];
    d.E_th = E_th * eV_to_Hz_fac;
    d.einn = (raw.N < 3) ? 1.e30 : raw.einn_eV * eV_to_Hz_fac;
    d.A_Plconst = 0.5 * P - 5.5 - raw.l;
    d.A_E_0_inv = 1. / (E_0 * eV_to_Hz_fac);
    d.A_sigma_0 = 1.e-22 * sigma_0;
    d.A_y_a_inv = 1. / y_a;
    d.A_P = P;
    d.A_y_w_sq = y_w * y_w;
    d.B_E_0_inv = 1. / (raw.B[2] * eV_to_Hz_fac);
    d.B_sigma_0 = 1.e-22 * raw.B[3];
    d.B_y_a_inv = 1. / raw.B[4];
    d.B_P = raw.B[5];
    d.B_y_w_sq = raw.B[6] * raw.B[6];
    d.B_y_0 = raw.B[7];
    d.B_y_1_sq = raw.B[8] * raw.B[8];
  }
  /* src/VernerRecombinationRates.cpp:38-90 */
  for (int i = 0; i < CMI_VERNER_NREC; ++i) {
    const cmi_verner_rec &raw = cmi_verner_recs[i];
    VernerRecDev &d = t.verner_rec[raw.ion];
    d.kind = raw.kind;
    d.p[0] = raw.p[0];
    d.p[1] = raw.p[1];
    d.p[2] = (raw.kind == 0 && raw.p[2] != 0.) ? 1. / raw.p[2] : raw.p[2];
    d.p[3] = (raw.kind == 0 && raw.p[3] != 0.) ? 1. / raw.p[3] : raw.p[3];
  }
  /* hydrogen and helium: the same fit with their own constants, :165-190 */
  {
    VernerRecDev &h = t.verner_rec[ION_H_n];
    h.kind = 0;
    h.p[0] = 7.982e-11;
    h.p[1] = 0.748;
    h.p[2] = 1. / 3.148;
    h.p[3] = 1. / 7.036e5;
    VernerRecDev &he = t.verner_rec[ION_He_n];
    he.kind = 0;
    he.p[0] = 3.294e-11;
    he.p[1] = 0.691;
    he.p[2] = 1. / 15.54;
    he.p[3] = 1. / 3.676e7;
  }
  /* dielectronic terms: Nussbaumer & Storey (1983) rows {a/t, 1, t, t^2,
   * exponent}, :197-285 */
  auto ns = [&t](int ion, double a, double b, double c, double d, double f) {
    VernerRecDev &r = t.verner_rec[ion];
    r.dkind = 1;
    r.d[0] = a;
    r.d[1] = b;
    r.d[2] = c;
    r.d[3] = d;
    r.d[4] = f;
  };
  ns(ION_C_p1, 1.8267, 4.1012, 4.8443, 0.2261, 0.5960);
  ns(ION_C_p2, 2.3196, 10.7328, 6.8830, -0.1824, 0.4101);
  ns(ION_N_n, 0., 0.6310, 0.1990, -0.0197, 0.4398);
  ns(ION_N_p1, 0.0320, -0.6624, 4.3191, 0.0003, 0.5946);
  ns(ION_N_p2, -0.8806, 11.2406, 30.7066, -1.1721, 0.6127);
  ns(ION_O_n, -0.0001, 0.0001, 0.0956, 0.0193, 0.4106);
  ns(ION_O_p1, -0.0036, 0.7519, 1.5252, -0.0838, 0.2769);
  ns(ION_Ne_p1, 0.0129, -0.1779, 0.9353, -0.0682, 0.4156);
  /* sulphur: sums of exponentials, in eV (:288-303) and in K (:304-312) */
  auto esum = [&t](int ion, double unit, int n, const double *c,
                   const double *E) {
    VernerRecDev &r = t.verner_rec[ion];
    r.dkind = 2;
    r.dunit = unit;
    r.dn = n;
    for (int k = 0; k < n; ++k) {
      r.dc[k] = c[k];
      r.dE[k] = E[k];
    }
  };
  {
    const double c1[] = {1.37e-9}, E1[] = {14.95};
    esum(ION_S_p1, 1. / 1.16045221e4, 1, c1, E1);
    const double c2[] = {8.0729e-9, 1.1012e-10}, E2[] = {17.56, 7.07};
    esum(ION_S_p2, 1. / 1.16045221e4, 2, c2, E2);
    const double c3[] = {5.817e-7, 1.391e-6, 1.123e-5,
                         1.521e-4, 1.875e-3, 2.097e-2};
    const double E3[] = {362.8, 1058., 7160., 3.26e4, 1.235e5, 2.07e5};
    esum(ION_S_p3, 1., 6, c3, E3);
  }
  /* src/ChargeTransferRates.cpp: {kind, a, b, c, d, e, lo, hi};
   * kind 0 zero, 1 constant, 2 a t^b (1 + c e^{d t}), 3 same * e^{e/t},
   * 4 a t^2 */
  auto set = [](CTFitDev &f, int kind, double a, double b, double c, double d,
                double e, double lo, double hi) {
    f.kind = kind;
    f.a = a;
    f.b = b;
    f.c = c;
    f.d = d;
    f.e = e;
    f.lo = lo;
    f.hi = hi;
  };
  /* recombination with H, :44-157 */
  set(t.ct_recomb_H[ION_He_n], 2, 7.47e-21, 2.06, 9.93, -3.89, 0, 0.6, 10.);
  set(t.ct_recomb_H[ION_C_p1], 2, 1.67e-19, 2.79, 304.74, -4.07, 0, 0.5, 5.);
  set(t.ct_recomb_H[ION_C_p2], 2, 3.25e-15, 0.21, 0.19, -3.29, 0, 0.1, 10.);
  set(t.ct_recomb_H[ION_N_n], 2, 1.01e-18, -0.29, -0.92, -8.38, 0, 0.01, 5.);
  set(t.ct_recomb_H[ION_N_p1], 2, 3.05e-16, 0.6, 2.65, -0.93, 0, 0.1, 10.);
  set(t.ct_recomb_H[ION_N_p2], 2, 4.54e-15, 0.57, -0.65, -0.89, 0, 0.001,
      10.);
  set(t.ct_recomb_H[ION_O_n], 2, 1.04e-15, 3.15e-2, -0.61, -9.73, 0, 0.001,
      1.);
  set(t.ct_recomb_H[ION_O_p1], 2, 1.04e-15, 0.27, 2.02, -5.92, 0, 0.01, 10.);
  set(t.ct_recomb_H[ION_Ne_n], 0, 0, 0, 0, 0, 0, 0, 0);
  set(t.ct_recomb_H[ION_Ne_p1], 1, 1.e-20, 0, 0, 0, 0, 0, 0);
  set(t.ct_recomb_H[ION_S_p1], 1, 1.e-20, 0, 0, 0, 0, 0, 0);
  set(t.ct_recomb_H[ION_S_p2], 2, 2.29e-15, 4.02e-2, 1.59, -6.06, 0, 0.1, 3.);
  set(t.ct_recomb_H[ION_S_p3], 2, 6.44e-15, 0.13, 2.69, -5.69, 0, 0.1, 3.);
  /* ionization by H+, :169-250 (all others zero) */
  set(t.ct_ion_H[ION_N_n], 3, 4.55e-18, -0.29, -0.92, -8.38, -1.086, 0.01,
      5.);
  set(t.ct_ion_H[ION_O_n], 3, 7.4e-17, 0.47, 24.37, -0.74, -0.023, 0.001, 1.);
  /* recombination with He, :262-395 */
  set(t.ct_recomb_He[ION_C_p2], 4, 4.6e-17, 0, 0, 0, 0, 0.1, 3.);
  set(t.ct_recomb_He[ION_N_p1], 2, 3.3e-16, 0.29, 1.3, -4.5, 0, 0.1, 3.);
  set(t.ct_recomb_He[ION_N_p2], 1, 1.5e-16, 0, 0, 0, 0, 0, 0);
  set(t.ct_recomb_He[ION_O_p1], 2, 2.e-16, 0.95, 0., 0., 0, 0.5, 5.);
  set(t.ct_recomb_He[ION_Ne_p1], 1, 1.e-20, 0, 0, 0, 0, 0, 0);
  set(t.ct_recomb_He[ION_S_p2], 2, 1.1e-15, 0.56, 0., 0., 0, 0.1, 3.);
  set(t.ct_recomb_He[ION_S_p3], 2, 7.6e-19, 0.32, 3.4, -5.25, 0, 0.1, 3.);

  /* which charge transfer terms the balance of each metal ion contains,
   * src/IonizationStateCalculator.cpp:323-501 */
  {
    const int with_rH[] = {ION_C_p2, ION_N_n,  ION_N_p1, ION_N_p2, ION_O_n,
                           ION_O_p1, ION_Ne_p1, ION_S_p1, ION_S_p2, ION_S_p3};
    for (int ion : with_rH)
      t.metal_ct[ion][0] = t.ct_recomb_H[ion];
    t.metal_ct[ION_N_n][1] = t.ct_ion_H[ION_N_n];
    t.metal_ct[ION_O_n][1] = t.ct_ion_H[ION_O_n];
    const int with_rHe[] = {ION_C_p2,  ION_N_p1, ION_N_p2, ION_O_p1,
                            ION_Ne_p1, ION_S_p2, ION_S_p3};
    for (int ion : with_rHe)
      t.metal_ct[ion][2] = t.ct_recomb_He[ion];
  }

  /* line cooling data, src/LineCoolingData.cpp:42-1399: energy levels to
   * energy differences in K, everything else copied */
  static_assert(CMI_LC_NFIVE == CMI_LC_NFIVE_DEV && CMI_LC_NTWO == CMI_LC_NTWO_DEV,
                "line cooling element counts");
  auto unit_factor = [](int unit) {
    /* :46-61: cm^-1, eV, Ry -> K */
    if (unit == 0)
      return 100. * CMI_PLANCK * CMI_LIGHTSPEED / CMI_BOLTZMANN;
    if (unit == 1)
      return CMI_ELECTRONVOLT / CMI_BOLTZMANN;
    return 2.179872325e-18 / CMI_BOLTZMANN;
  };
  static const int TR[5][5] = {{-1, 0, 1, 2, 3},
                               {-1, -1, 4, 5, 6},
                               {-1, -1, -1, 7, 8},
                               {-1, -1, -1, -1, 9},
                               {-1, -1, -1, -1, -1}};
  LineCoolingDev &lc = t.lc;
  for (int el = 0; el < CMI_LC_NFIVE; ++el) {
    const cmi_lc_five_level &d = cmi_lc_five[el];
    const double f = unit_factor(d.unit);
    for (int j = 1; j < 5; ++j) {
      lc.energy[el][TR[0][j]] = d.levels[j - 1] * f;
      for (int i = 1; i < j; ++i)
        lc.energy[el][TR[i][j]] = (d.levels[j - 1] - d.levels[i - 1]) * f;
    }
    for (int tr = 0; tr < CMI_LC_NTRANS; ++tr) {
      lc.A[el][tr] = d.A[tr];
      for (int k = 0; k < 7; ++k)
        lc.cs[el][tr][k] = d.cs[tr][k];
    }
    for (int k = 0; k < 5; ++k)
      lc.inv_weight[el][k] = d.inv_weight[k];
  }
  for (int el = 0; el < CMI_LC_NTWO; ++el) {
    const cmi_lc_two_level &d = cmi_lc_two[el];
    lc.two_energy[el] = d.energy * unit_factor(d.unit);
    lc.two_A[el] = d.A;
    for (int k = 0; k < 7; ++k)
      lc.two_cs[el][k] = d.cs[k];
    lc.two_inv_weight[el][0] = d.inv_weight[0];
    lc.two_inv_weight[el][1] = d.inv_weight[1];
  }
  /* :1390-1398 */
  lc.prefactor = CMI_PLANCK * CMI_PLANCK /
                 (std::sqrt(CMI_BOLTZMANN) *
                  std::pow(2. * M_PI * CMI_ELECTRON_MASS, 1.5));
}

/* the model as the HOST evaluates it (tabulating spectra, the reference cross
 * section of the sort key): the tables' host copies instead of the device's */
ModelDev host_model_of(const cmi_gpu_engine *e) {
  ModelDev m = e->model;
  m.tables = e->host_tables;
  TableDev *t[4] = {&m.spectrum_table[0], &m.spectrum_table[1], &m.xsec_table,
                    &m.recomb_table};
  for (int k = 0; k < 4; ++k) {
    if (t[k]->n > 0) {
      t[k]->x = e->user_table_host[k].data();
      t[k]->y = e->user_table_host[k].data() + t[k]->n;
    }
  }
  return m;
}

/* store a caller-supplied table: n abscissae and rows x n values */
int store_user_table(cmi_gpu_engine *e, int which, TableDev &t, int32_t n,
                     int rows, const double *x, const double *y,
                     int32_t interpolation) {
  HIP_TRY(hipSetDevice(e->device));
  /* (kernels of an earlier call may still read the old table) */
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::vector<double> &h = e->user_table_host[which];
  h.assign(x, x + n);
  h.insert(h.end(), y, y + (size_t)rows * (size_t)n);
  (void)hipFree(e->user_table[which]);
  e->user_table[which] = nullptr;
  HIP_TRY(hipMalloc(&e->user_table[which], sizeof(double) * h.size()));
  HIP_TRY(hipMemcpy(e->user_table[which], h.data(), sizeof(double) * h.size(),
                    hipMemcpyHostToDevice));
  t.x = e->user_table[which];
  t.y = e->user_table[which] + n;
  t.n = n;
  t.interpolation = interpolation;
  return CMI_GPU_OK;
}

/* n >= 2 abscissae, finite and in strictly ascending order */
bool table_abscissae_ok(int32_t n, const double *x) {
  if (n < 2 || !x)
    return false;
  for (int32_t i = 0; i < n; ++i)
    if (!std::isfinite(x[i]) || (i > 0 && !(x[i] > x[i - 1])))
      return false;
  return true;
}

/* the guide table of a cumulative distribution (SpectraDev) */
void build_guide(const double *cdf, uint16_t *guide) {
  uint32_t last = 0; /* last entry below the current k / G */
  for (uint32_t k = 0; k <= CMI_NGUIDE + 1; ++k) {
    const double edge = (double)k / CMI_NGUIDE;
    while (last + 1 < CMI_NFREQ && cdf[last + 1] < edge)
      ++last;
    guide[k] = (uint16_t)((cdf[last] < edge) ? last : 0u);
  }
}

/* Tabulate the sampled spectra on the host: the constructors of
 * PlanckPhotonSourceSpectrum (src/PlanckPhotonSourceSpectrum.cpp:53-113),
 * Hydrogen/HeliumLymanContinuumSpectrum
 * (src/HydrogenLymanContinuumSpectrum.cpp:40-122,
 * src/HeliumLymanContinuumSpectrum.cpp:45-133) and
 * HeliumTwoPhotonContinuumSpectrum
 * (src/HeliumTwoPhotonContinuumSpectrum.cpp:44-101). */
void build_spectra(const ModelDev &host_model, SpectraDev &s) {
  memset(&s, 0, sizeof s);
  const double h = CMI_PLANCK, k = CMI_BOLTZMANN;
  for (int which_planck = 0; which_planck < 2; ++which_planck) {
    /* 0: the discrete sources' spectrum, 1: the continuous source's */
    const bool wanted =
        which_planck == 0
            ? host_model.spectrum_type == CMI_GPU_SPECTRUM_PLANCK
            : (host_model.continuous_type != 0 &&
               host_model.continuous_spectrum_type == CMI_GPU_SPECTRUM_PLANCK);
    if (!wanted)
      continue;
    const double temperature = which_planck == 0
                                   ? host_model.planck_temperature
                                   : host_model.continuous_planck_temperature;
    double *planck_cdf = which_planck == 0 ? s.planck_cdf : s.planck2_cdf;
    double *planck_logcdf =
        which_planck == 0 ? s.planck_logcdf : s.planck2_logcdf;
    double *planck_logfreq =
        which_planck == 0 ? s.planck_logfreq : s.planck2_logfreq;
    const double max_frequency = 4.;
    const double min_frequency = 3.289e15;
    std::vector<double> frequency(CMI_NFREQ), luminosity(CMI_NFREQ);
    for (int i = 0; i < CMI_NFREQ; ++i) {
      frequency[i] = 1. + i * (max_frequency - 1.) / (CMI_NFREQ - 1.);
      luminosity[i] = frequency[i] * frequency[i] * frequency[i] /
                      (std::exp(h * frequency[i] * min_frequency /
                                (k * temperature)) -
                       1.);
    }
    planck_cdf[0] = 0.;
    for (int i = 1; i < CMI_NFREQ; ++i)
      planck_cdf[i] = planck_cdf[i - 1] +
                        0.5 *
                            (luminosity[i] / frequency[i] +
                             luminosity[i - 1] / frequency[i - 1]) *
                            (frequency[i] - frequency[i - 1]);
    planck_logcdf[0] = -10.;
    planck_logfreq[0] = 0.;
    for (int i = 1; i < CMI_NFREQ; ++i) {
      planck_cdf[i] /= planck_cdf[CMI_NFREQ - 1];
      planck_logcdf[i] = std::log10(planck_cdf[i]);
      planck_logfreq[i] = std::log10(frequency[i]);
    }
  }
  for (int which = 0; which < 2; ++which) {
    const int ion = which == 0 ? ION_H_n : ION_He_n;
    const double min_frequency =
        which == 0 ? 3.289e15 : 1.81 * 3.288465385e15;
    const double max_frequency =
        which == 0 ? 4. * min_frequency : 4. * 3.288465385e15;
    double *nu = s.lyc_freq[which];
    std::vector<double> xsec(CMI_NFREQ);
    for (int i = 0; i < CMI_NFREQ; ++i) {
      nu[i] = min_frequency +
              i * (max_frequency - min_frequency) / (CMI_NFREQ - 1.);
      double sigma[CMI_NION];
      cmi_cross_sections(host_model, nu[i], sigma);
      xsec[i] = sigma[ion];
    }
    for (int iT = 0; iT < CMI_NTEMP; ++iT) {
      double *cdf = s.lyc_cdf[which][iT];
      cdf[0] = 0.;
      s.lyc_T[iT] = 1500. + (iT + 0.5) * 13500. / CMI_NTEMP;
      for (int inu = 1; inu < CMI_NFREQ; ++inu) {
        const double j1 =
            nu[inu - 1] * nu[inu - 1] * nu[inu - 1] * xsec[inu - 1] *
            std::exp(-(h * (nu[inu - 1] - min_frequency)) / (k * s.lyc_T[iT]));
        const double j2 =
            nu[inu] * nu[inu] * nu[inu] * xsec[inu] *
            std::exp(-(h * (nu[inu] - min_frequency)) / (k * s.lyc_T[iT]));
        cdf[inu] =
            0.5 * (j1 / nu[inu] + j2 / nu[inu - 1]) * (nu[inu] - nu[inu - 1]);
      }
      for (int inu = 1; inu < CMI_NFREQ; ++inu)
        cdf[inu] = cdf[inu - 1] + cdf[inu];
      const double total = cdf[CMI_NFREQ - 1];
      for (int inu = 0; inu < CMI_NFREQ; ++inu)
        cdf[inu] /= total; /* NaN rows if the ion's cross section is zero */
    }
  }
  {
    const double min_frequency = 3.288465385e15;
    const double max_frequency = 1.6 * min_frequency;
    const double nu0 = 4.98e15;
    auto A_of = [](double y) {
      if (!(y < 1.))
        return 0.;
      /* Utilities::locate on the 41-point table + linear interpolation */
      uint32_t lo = 0, hi = CMI_HE2Q_N;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (y > cmi_he2q_y[mid])
          lo = mid;
        else
          hi = mid;
      }
      if (lo == CMI_HE2Q_N - 1)
        --lo;
      const double f =
          (y - cmi_he2q_y[lo]) / (cmi_he2q_y[lo + 1] - cmi_he2q_y[lo]);
      return cmi_he2q_A[lo] + f * (cmi_he2q_A[lo + 1] - cmi_he2q_A[lo]);
    };
    for (int i = 0; i < CMI_NFREQ; ++i)
      s.he2pc_freq[i] = min_frequency + i * (max_frequency - min_frequency) /
                                            (CMI_NFREQ - 1.);
    s.he2pc_cdf[0] = 0.;
    for (int i = 1; i < CMI_NFREQ; ++i) {
      const double A1 = A_of(s.he2pc_freq[i - 1] / nu0);
      const double A2 = A_of(s.he2pc_freq[i] / nu0);
      s.he2pc_cdf[i] =
          0.5 * (A1 + A2) * (s.he2pc_freq[i] - s.he2pc_freq[i - 1]);
    }
    for (int i = 1; i < CMI_NFREQ; ++i)
      s.he2pc_cdf[i] = s.he2pc_cdf[i - 1] + s.he2pc_cdf[i];
    const double total = s.he2pc_cdf[CMI_NFREQ - 1];
    for (int i = 0; i < CMI_NFREQ; ++i)
      s.he2pc_cdf[i] /= total;
  }
  build_guide(s.planck_cdf, s.planck_guide);
  build_guide(s.planck2_cdf, s.planck2_guide);
  build_guide(s.he2pc_cdf, s.he2pc_guide);
  for (int which = 0; which < 2; ++which)
    for (int iT = 0; iT < CMI_NTEMP; ++iT)
      build_guide(s.lyc_cdf[which][iT], s.lyc_guide[which][iT]);
}

/* (re)build and upload the spectra tables if a sampled spectrum is in use */
int ensure_spectra(cmi_gpu_engine *e) {
  const bool needed =
      e->model.spectrum_type == CMI_GPU_SPECTRUM_PLANCK ||
      (e->model.continuous_type != 0 &&
       e->model.continuous_spectrum_type == CMI_GPU_SPECTRUM_PLANCK) ||
      e->model.reemit_type == CMI_GPU_REEMIT_PHYSICAL;
  if (!needed || !e->spectra_dirty)
    return CMI_GPU_OK;
  if (!e->spectra)
    HIP_TRY(hipMalloc(&e->spectra, sizeof(SpectraDev)));
  const ModelDev host_model = host_model_of(e);
  SpectraDev *host = new SpectraDev;
  build_spectra(host_model, *host);
  HIP_TRY(hipStreamSynchronize(e->stream));
  hipError_t err =
      hipMemcpy(e->spectra, host, sizeof(SpectraDev), hipMemcpyHostToDevice);
  delete host;
  HIP_TRY(err);
  e->model.spectra = e->spectra;
  e->spectra_dirty = false;
  return CMI_GPU_OK;
}

__global__ void gather_strided_kernel(const double *src, int64_t stride,
                                      double *dst, int64_t n) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    dst[i] = src[i * stride];
}
__global__ void scatter_strided_kernel(const double *src, double *dst,
                                       int64_t stride, int64_t n) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += step)
    dst[i * stride] = src[i];
}

int grid_blocks(cmi_gpu_engine *e, int64_t work_items, int blocks_per_cu) {
  int64_t want = (work_items + CMI_BLOCK - 1) / CMI_BLOCK;
  int64_t cap = (int64_t)e->num_cu * blocks_per_cu;
  if (want < 1)
    want = 1;
  return (int)(want < cap ? want : cap);
}

void update_full_flag(cmi_gpu_engine *e) {
  /* the light transport kernel is exact iff no ion other than H0 can ever
   * have a non-zero cross section */
  bool full = e->model.xsec_verner != 0;
  if (!full)
    for (int i = 1; i < CMI_NION; ++i)
      if (e->model.xsec_fixed[i] != 0.)
        full = true;
  /* accumulator layout follows the transport kernel: [16][ncell] when only
   * hydrogen is accumulated (neighbouring cells share 64-B lines), [ncell][16]
   * when every step updates all 16 values of a cell. A switch zeroes the
   * whole block (reset_grid of a hydrogen-only run clears only the fields
   * such a run adds to), so switching between iterations is safe. */
  if (full != e->full_ions && e->acc_block) {
    /* (a clear that could not be enqueued is made up for, with its error
     * reported, by the next cmi_gpu_reset_grid) */
    if (hipSetDevice(e->device) != hipSuccess ||
        hipMemsetAsync(e->acc_block, 0,
                       (size_t)CMI_NACC * e->ncell * sizeof(double),
                       e->stream) != hipSuccess)
      e->acc_block_dirty = true;
  }
  e->full_ions = full;
  if (full) {
    e->cells.acc_field_stride = 1;
    e->cells.acc_cell_stride = CMI_NACC;
  } else {
    e->cells.acc_field_stride = e->ncell;
    e->cells.acc_cell_stride = 1;
  }
}

int rebuild_opacity(cmi_gpu_engine *e) {
  opacity_kernel<<<grid_blocks(e, e->ncell, 8), CMI_BLOCK, 0, e->stream>>>(
      e->cells, e->ncell);
  HIP_TRY(hipGetLastError());
  return CMI_GPU_OK;
}

} // namespace

extern "C" {

const char *cmi_gpu_last_error(void) { return g_last_error.c_str(); }

int cmi_gpu_create(const cmi_gpu_config *config, cmi_gpu_engine **out) {
  if (!config || !out)
    return fail(CMI_GPU_EINVAL, "cmi_gpu_create: null argument");
  for (int a = 0; a < 3; ++a) {
    if (config->ncell[a] <= 0)
      return fail(CMI_GPU_EINVAL, "number of cells must be positive");
    if (!(config->sides[a] > 0.))
      return fail(CMI_GPU_EINVAL, "box sides must be positive");
  }
  if (config->sub_ncell[0] > 0 || config->sub_ncell[1] > 0 ||
      config->sub_ncell[2] > 0) {
    for (int a = 0; a < 3; ++a) {
      if (config->sub_ncell[a] < 3 || config->sub_offset[a] < 0 ||
          config->sub_offset[a] + config->sub_ncell[a] > config->ncell[a])
        return fail(CMI_GPU_EINVAL,
                    "a block of a decomposed grid must lie inside the grid "
                    "and be at least 3 cells wide");
    }
  }
  {
    /* the kernels index the cells of an engine with 32 bits (2^31 cells of
     * 272 B would not fit one device anyway); a larger grid has to be
     * decomposed into blocks */
    int64_t local = 1;
    for (int a = 0; a < 3; ++a)
      local *= config->sub_ncell[0] > 0 ? config->sub_ncell[a]
                                        : config->ncell[a];
    if (local >= (1ll << 31))
      return fail(CMI_GPU_EINVAL,
                  "more than 2^31 - 1 cells per engine are not supported");
  }
  int ndev = 0;
  hipError_t err = hipGetDeviceCount(&ndev);
  if (err != hipSuccess || ndev == 0)
    return fail(CMI_GPU_EDEVICE,
                "no HIP device available (%s); this engine has no CPU path",
                err == hipSuccess ? "device count is 0"
                                  : hipGetErrorString(err));
  if (config->device < 0 || config->device >= ndev)
    return fail(CMI_GPU_EINVAL, "device %d out of range [0,%d)",
                config->device, ndev);
  HIP_TRY(hipSetDevice(config->device));

  cmi_gpu_engine *e = new cmi_gpu_engine();
  e->config = *config;
  e->device = config->device;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, e->device));
  e->num_cu = prop.multiProcessorCount;
  if (config->stream) {
    e->stream = (hipStream_t)config->stream;
  } else {
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->own_stream = true;
  }

  /* CartesianDensityGrid ctor, src/CartesianDensityGrid.cpp:72-79; a block
   * of a decomposed grid keeps the whole grid's anchor and cell size
   * (DensitySubGridCreator::create_subgrid,
   * src/DensitySubGridCreator.hpp:314-396) */
  GridDev &g = e->grid;
  const bool decomposed = config->sub_ncell[0] > 0 || config->sub_ncell[1] > 0 ||
                          config->sub_ncell[2] > 0;
  for (int a = 0; a < 3; ++a) {
    g.anchor[a] = config->anchor[a];
    g.box_sides[a] = config->sides[a];
    g.global_ncell[a] = config->ncell[a];
    g.ncell[a] = decomposed ? config->sub_ncell[a] : config->ncell[a];
    g.offset[a] = decomposed ? config->sub_offset[a] : 0;
    g.global_periodic[a] = config->periodic[a] ? 1 : 0;
    /* a block of a decomposed grid is not periodic itself: a flight across a
     * periodic face of the whole box is handed over like any other */
    g.periodic[a] = (config->periodic[a] && !decomposed) ? 1 : 0;
    g.cellside[a] = config->sides[a] / config->ncell[a];
    g.inv_cellside[a] = 1. / g.cellside[a];
  }
  g.decomposed = decomposed ? 1 : 0;
  g.copy_rank = 0;
  g.copy_count = 1;
  e->ncell = (int64_t)g.ncell[0] * g.ncell[1] * g.ncell[2];
  g.ncell_total = e->ncell;

  const size_t field_bytes = (size_t)e->ncell * sizeof(double);
  HIP_TRY(hipMalloc(&e->state_block, 16 * field_bytes));
  HIP_TRY(hipMemsetAsync(e->state_block, 0, 16 * field_bytes, e->stream));
  if (config->external_accumulators) {
    e->acc_block = (double *)config->external_accumulators;
  } else {
    HIP_TRY(hipMalloc(&e->acc_block, CMI_NACC * field_bytes));
    e->own_acc = true;
  }
  HIP_TRY(hipMemsetAsync(e->acc_block, 0, CMI_NACC * field_bytes, e->stream));
  HIP_TRY(hipMalloc(&e->opacity, (size_t)e->ncell * sizeof(double2)));
  HIP_TRY(hipMalloc(&e->counters, sizeof(CountersDev) * CMI_COUNTER_SHARDS));
  HIP_TRY(hipMemsetAsync(e->counters, 0,
                         sizeof(CountersDev) * CMI_COUNTER_SHARDS, e->stream));
  HIP_TRY(hipMalloc(&e->tables, sizeof(TablesDev)));
  {
    e->host_tables = new TablesDev;
    build_tables(*e->host_tables);
    HIP_TRY(hipMemcpy(e->tables, e->host_tables, sizeof(TablesDev),
                      hipMemcpyHostToDevice));
  }

  CellsDev &c = e->cells;
  c.number_density = e->state_block;
  c.temperature = e->state_block + e->ncell;
  for (int i = 0; i < CMI_NION; ++i)
    c.x[i] = e->state_block + (int64_t)(2 + i) * e->ncell;
  c.acc_base = e->acc_block;
  c.acc_field_stride = e->ncell; /* SoA until all 16 fields are in use */
  c.acc_cell_stride = 1;
  c.opacity = e->opacity;

  ModelDev &m = e->model;
  memset(&m, 0, sizeof m);
  m.tables = e->tables;
  /* DensityGrid ctor, src/DensityGrid.hpp:219-222 */
  m.nu_H = eV_to_Hz(13.6);
  m.nu_He = eV_to_Hz(24.6);
  m.reemit_type = CMI_GPU_REEMIT_NONE;
  m.photon_weight[0] = 1.;
  m.photon_weight[1] = 1.;

  cmi_gpu_temperature_params &tp = e->tparams;
  tp.do_temperature_calculation = 0;
  tp.minimum_number_of_iterations = 3;
  tp.epsilon_convergence = 1.e-3;
  tp.maximum_number_of_iterations = 100;
  tp.pah_heating_factor = 0.;
  tp.cosmic_ray_heating_factor = 0.;
  tp.cosmic_ray_heating_limit = 0.75;
  tp.cosmic_ray_heating_scale_length = 1.33333 * 3.086e19;
  tp.minimum_ionized_temperature = 4000.;
  m.t_epsilon = tp.epsilon_convergence;
  m.t_max_iterations = tp.maximum_number_of_iterations;
  m.crlim = tp.cosmic_ray_heating_limit;
  m.crscale = tp.cosmic_ray_heating_scale_length;
  m.t_min_ionized = tp.minimum_ionized_temperature;

  HIP_TRY(hipStreamSynchronize(e->stream));
  /* CMI_GPU_TUNING="key=value,key=value": cmi_gpu_set_tuning for hosts that
   * have no way to call it (the cmi-gpu executable, a code that links the
   * library mode): experiments and bisections, not configuration */
  if (const char *env = std::getenv("CMI_GPU_TUNING")) {
    std::string all(env);
    size_t at = 0;
    while (at < all.size()) {
      size_t end = all.find(',', at);
      if (end == std::string::npos)
        end = all.size();
      const std::string item = all.substr(at, end - at);
      const size_t eq = item.find('=');
      if (eq != std::string::npos) {
        const int rc = cmi_gpu_set_tuning(e, item.substr(0, eq).c_str(),
                                          std::atoll(item.c_str() + eq + 1));
        if (rc) {
          cmi_gpu_destroy(e);
          return rc;
        }
      }
      at = end + 1;
    }
  }
  *out = e;
  return CMI_GPU_OK;
}

int cmi_gpu_destroy(cmi_gpu_engine *e) {
  if (!e)
    return CMI_GPU_OK;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  release_events(e, e->shoot_events);
  release_events(e, e->update_events);
  release_events(e, e->kernel_events);
  for (auto &p : e->event_pool) {
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
  }
  (void)hipFree(e->state_block);
  if (e->own_acc)
    (void)hipFree(e->acc_block);
  (void)hipFree(e->opacity);
  (void)hipFree(e->counters);
  (void)hipFree(e->trackers.counts);
  (void)hipFree(e->trackers.absorption);
  (void)hipFree(e->trackers.flux);
  (void)hipFree(e->temp_pipe_block);
  (void)hipFree(e->pad_H);
  (void)hipFree(e->temp_pipe_counts);
  (void)hipFree(e->tables);
  (void)hipFree(e->spectra);
  for (int k = 0; k < 4; ++k)
    (void)hipFree(e->user_table[k]);
  delete e->host_tables;
  (void)hipFree(e->source_position);
  (void)hipFree(e->source_cumulative);
  (void)hipFree(e->sort_keys[0]);
  (void)hipFree(e->select_count);
  (void)hipFree(e->select_ids);
  (void)hipFree(e->select_rows);
  (void)hipFree(e->sort_temp);
  (void)hipFree(e->queue_block);
  (void)hipFree(e->queue_counts);
  if (e->mailbox)
    (void)hipHostFree(e->mailbox);
  (void)hipFree(e->export_count);
  if (e->own_export_rows)
    (void)hipFree(e->export_rows);
  (void)hipFree(e->import_rows);
  (void)hipFree(e->tile_block);
  (void)hipFree(e->tile_counts);
  (void)hipFree(e->launch_steps);
  if (e->own_stream)
    (void)hipStreamDestroy(e->stream);
  delete e;
  return CMI_GPU_OK;
}

int cmi_gpu_synchronize(cmi_gpu_engine *e) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  HIP_TRY(hipStreamSynchronize(e->stream));
  return CMI_GPU_OK;
}

int64_t cmi_gpu_number_of_cells(const cmi_gpu_engine *e) {
  return e ? e->ncell : -1;
}

/* PhotonSource ctor, src/PhotonSource.cpp:104-130: how the packets are
 * shared between the discrete sources and the continuous one, and the weight
 * a packet of either kind carries */
static void mix_sources(cmi_gpu_engine *e) {
  ModelDev &m = e->model;
  const double discrete = m.nsource > 0 ? e->discrete_luminosity : 0.;
  const double continuous =
      m.continuous_type != 0 ? e->continuous_luminosity : 0.;
  m.total_luminosity = discrete + continuous;
  m.continuous_probability = 0.;
  m.photon_weight[0] = 1.;
  m.photon_weight[1] = 1.;
  if (m.total_luminosity > 0.) {
    if (discrete > 0.) {
      m.continuous_probability = continuous > 0. ? 0.5 : 0.;
      m.photon_weight[0] = 1.;
      m.photon_weight[1] =
          continuous > 0. ? (1. - m.continuous_probability) * continuous /
                                m.continuous_probability / discrete
                          : 1.;
    } else {
      m.continuous_probability = 1.;
      m.photon_weight[0] = 0.;
      m.photon_weight[1] = 1.;
    }
  }
  e->have_sources = m.total_luminosity > 0.;
}

int cmi_gpu_set_sources(cmi_gpu_engine *e, int32_t n, const double *positions,
                        const double *weights, double total_luminosity) {
  if (!e || n < 0 || (n > 0 && (!positions || !weights)))
    return fail(CMI_GPU_EINVAL, "cmi_gpu_set_sources: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(e->source_position);
  (void)hipFree(e->source_cumulative);
  e->source_position = nullptr;
  e->source_cumulative = nullptr;
  e->source_position_host.clear();
  e->model.nsource = 0;
  e->model.source_position = nullptr;
  e->model.source_cumulative = nullptr;
  e->discrete_luminosity = 0.;
  if (n == 0) { /* no discrete sources (a continuous source only) */
    mix_sources(e);
    return CMI_GPU_OK;
  }
  /* PhotonSource ctor, src/PhotonSource.cpp:74-93 */
  std::vector<double> cumulative(n);
  for (int i = 0; i < n; ++i)
    cumulative[i] = (i > 0 ? cumulative[i - 1] : 0.) + weights[i];
  if (std::abs(cumulative.back() - 1.) > 1.e-9) {
    mix_sources(e);
    return fail(CMI_GPU_EINVAL,
                "Discrete source weights do not sum to 1.0 (%g)!",
                cumulative.back());
  }
  cumulative.back() = 1.;
  HIP_TRY(hipMalloc(&e->source_position, sizeof(double) * 3 * n));
  HIP_TRY(hipMalloc(&e->source_cumulative, sizeof(double) * n));
  HIP_TRY(hipMemcpy(e->source_position, positions, sizeof(double) * 3 * n,
                    hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->source_cumulative, cumulative.data(),
                    sizeof(double) * n, hipMemcpyHostToDevice));
  e->source_position_host.assign(positions, positions + 3 * (size_t)n);
  e->model.nsource = n;
  e->model.source_position = e->source_position;
  e->model.source_cumulative = e->source_cumulative;
  e->discrete_luminosity = total_luminosity;
  mix_sources(e);
  return CMI_GPU_OK;
}

int cmi_gpu_set_continuous_source(cmi_gpu_engine *e, int32_t type,
                                  double luminosity) {
  if (!e || (type != CMI_GPU_CONTINUOUS_NONE &&
             type != CMI_GPU_CONTINUOUS_ISOTROPIC) ||
      (type != CMI_GPU_CONTINUOUS_NONE && !(luminosity > 0.)))
    return fail(CMI_GPU_EINVAL, "cmi_gpu_set_continuous_source: bad argument");
  e->model.continuous_type = type;
  e->continuous_luminosity = type != CMI_GPU_CONTINUOUS_NONE ? luminosity : 0.;
  e->spectra_dirty = true;
  mix_sources(e);
  return CMI_GPU_OK;
}

int cmi_gpu_set_continuous_source_planar(cmi_gpu_engine *e, int32_t axis,
                                         double intercept,
                                         const double *anchor,
                                         const double *sides,
                                         double luminosity) {
  if (!e || axis < 0 || axis > 2 || !anchor || !sides || !(sides[0] > 0.) ||
      !(sides[1] > 0.) || !(luminosity > 0.))
    return fail(CMI_GPU_EINVAL,
                "cmi_gpu_set_continuous_source_planar: bad argument");
  ModelDev &m = e->model;
  m.continuous_type = CMI_GPU_CONTINUOUS_PLANAR;
  m.continuous_axis = axis;
  m.continuous_intercept = intercept;
  for (int k = 0; k < 2; ++k) {
    m.continuous_anchor[k] = anchor[k];
    m.continuous_side[k] = sides[k];
  }
  e->continuous_luminosity = luminosity;
  e->spectra_dirty = true;
  mix_sources(e);
  return CMI_GPU_OK;
}

int cmi_gpu_set_continuous_spectrum_monochromatic(cmi_gpu_engine *e,
                                                  double frequency) {
  if (!e || !(frequency > 0.))
    return fail(CMI_GPU_EINVAL, "monochromatic spectrum: bad argument");
  e->model.continuous_spectrum_type = CMI_GPU_SPECTRUM_MONOCHROMATIC;
  e->model.continuous_mono_frequency = frequency;
  e->have_continuous_spectrum = true;
  return CMI_GPU_OK;
}

int cmi_gpu_set_continuous_spectrum_planck(cmi_gpu_engine *e,
                                           double temperature) {
  if (!e || !(temperature > 0.))
    return fail(CMI_GPU_EINVAL, "Planck spectrum: bad argument");
  e->model.continuous_spectrum_type = CMI_GPU_SPECTRUM_PLANCK;
  e->model.continuous_planck_temperature = temperature;
  e->have_continuous_spectrum = true;
  e->spectra_dirty = true;
  return CMI_GPU_OK;
}

int cmi_gpu_set_spectrum_monochromatic(cmi_gpu_engine *e, double frequency) {
  if (!e || !(frequency > 0.))
    return fail(CMI_GPU_EINVAL, "monochromatic spectrum: bad argument");
  e->model.spectrum_type = CMI_GPU_SPECTRUM_MONOCHROMATIC;
  e->model.mono_frequency = frequency;
  e->have_spectrum = true;
  return CMI_GPU_OK;
}

int cmi_gpu_set_spectrum_planck(cmi_gpu_engine *e, double temperature) {
  if (!e || !(temperature > 0.))
    return fail(CMI_GPU_EINVAL, "Planck spectrum: bad argument");
  e->model.spectrum_type = CMI_GPU_SPECTRUM_PLANCK;
  e->model.planck_temperature = temperature;
  e->have_spectrum = true;
  e->spectra_dirty = true;
  return CMI_GPU_OK;
}

int cmi_gpu_set_spectrum_table(cmi_gpu_engine *e, int32_t role, int32_t n,
                               const double *frequency,
                               const double *cumulative,
                               int32_t interpolation) {
  if (!e || (role != CMI_GPU_ROLE_SOURCE && role != CMI_GPU_ROLE_CONTINUOUS) ||
      !frequency || !table_abscissae_ok(n, cumulative) ||
      (interpolation != CMI_GPU_TABLE_LINEAR &&
       interpolation != CMI_GPU_TABLE_LOGLOG))
    return fail(CMI_GPU_EINVAL,
                "cmi_gpu_set_spectrum_table: needs n >= 2 frequencies and a "
                "strictly ascending cumulative distribution");
  if (cumulative[0] < 0. || cumulative[0] > 1.e-9 ||
      std::abs(cumulative[n - 1] - 1.) > 1.e-9)
    return fail(CMI_GPU_EINVAL,
                "cmi_gpu_set_spectrum_table: the cumulative distribution must "
                "run from 0 to 1 (%g ... %g)",
                cumulative[0], cumulative[n - 1]);
  for (int32_t i = 0; i < n; ++i)
    if (!(frequency[i] > 0.) || !std::isfinite(frequency[i]))
      return fail(CMI_GPU_EINVAL,
                  "cmi_gpu_set_spectrum_table: frequency %d is not positive",
                  i);
  int rc = store_user_table(e, role, e->model.spectrum_table[role], n, 1,
                            cumulative, frequency, interpolation);
  if (rc)
    return rc;
  if (role == CMI_GPU_ROLE_SOURCE) {
    e->model.spectrum_type = CMI_GPU_SPECTRUM_TABLE;
    e->have_spectrum = true;
  } else {
    e->model.continuous_spectrum_type = CMI_GPU_SPECTRUM_TABLE;
    e->have_continuous_spectrum = true;
  }
  return CMI_GPU_OK;
}

int cmi_gpu_set_cross_sections_table(cmi_gpu_engine *e, int32_t n,
                                     const double *frequency,
                                     const double *sigma,
                                     int32_t interpolation) {
  if (!e || !sigma || !table_abscissae_ok(n, frequency) ||
      !(frequency[0] > 0.) ||
      (interpolation != CMI_GPU_TABLE_LINEAR &&
       interpolation != CMI_GPU_TABLE_LOGLOG))
    return fail(CMI_GPU_EINVAL,
                "cmi_gpu_set_cross_sections_table: needs n >= 2 positive, "
                "strictly ascending frequencies and sigma[14][n]");
  for (size_t i = 0; i < (size_t)CMI_NION * (size_t)n; ++i)
    if (!(sigma[i] >= 0.) || !std::isfinite(sigma[i]))
      return fail(CMI_GPU_EINVAL,
                  "cmi_gpu_set_cross_sections_table: cross section %zu of ion "
                  "%zu is negative or not finite",
                  i % (size_t)n, i / (size_t)n);
  int rc = store_user_table(e, 2, e->model.xsec_table, n, CMI_NION, frequency,
                            sigma, interpolation);
  if (rc)
    return rc;
  e->model.xsec_verner = 2;
  e->have_xsec = true;
  e->spectra_dirty = true;
  update_full_flag(e);
  return CMI_GPU_OK;
}

int cmi_gpu_set_recombination_rates_table(cmi_gpu_engine *e, int32_t n,
                                          const double *temperature,
                                          const double *alpha,
                                          int32_t interpolation) {
  if (!e || !alpha || !table_abscissae_ok(n, temperature) ||
      !(temperature[0] > 0.) ||
      (interpolation != CMI_GPU_TABLE_LINEAR &&
       interpolation != CMI_GPU_TABLE_LOGLOG))
    return fail(CMI_GPU_EINVAL,
                "cmi_gpu_set_recombination_rates_table: needs n >= 2 positive, "
                "strictly ascending temperatures and alpha[14][n]");
  for (size_t i = 0; i < (size_t)CMI_NION * (size_t)n; ++i)
    if (!(alpha[i] >= 0.) || !std::isfinite(alpha[i]))
      return fail(CMI_GPU_EINVAL,
                  "cmi_gpu_set_recombination_rates_table: rate %zu of ion %zu "
                  "is negative or not finite",
                  i % (size_t)n, i / (size_t)n);
  int rc = store_user_table(e, 3, e->model.recomb_table, n, CMI_NION,
                            temperature, alpha, interpolation);
  if (rc)
    return rc;
  e->model.recomb_verner = 2;
  e->have_recomb = true;
  return CMI_GPU_OK;
}

int cmi_gpu_set_cross_sections_fixed(cmi_gpu_engine *e, const double *sigma) {
  if (!e || !sigma)
    return fail(CMI_GPU_EINVAL, "fixed cross sections: bad argument");
  e->model.xsec_verner = 0;
  for (int i = 0; i < CMI_NION; ++i)
    e->model.xsec_fixed[i] = sigma[i];
  e->have_xsec = true;
  e->spectra_dirty = true;
  update_full_flag(e);
  return CMI_GPU_OK;
}

int cmi_gpu_set_cross_sections_verner(cmi_gpu_engine *e) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  e->model.xsec_verner = 1;
  e->have_xsec = true;
  e->spectra_dirty = true;
  update_full_flag(e);
  return CMI_GPU_OK;
}

int cmi_gpu_set_recombination_rates_fixed(cmi_gpu_engine *e,
                                          const double *alpha) {
  if (!e || !alpha)
    return fail(CMI_GPU_EINVAL, "fixed recombination rates: bad argument");
  e->model.recomb_verner = 0;
  for (int i = 0; i < CMI_NION; ++i)
    e->model.recomb_fixed[i] = alpha[i];
  e->have_recomb = true;
  return CMI_GPU_OK;
}

int cmi_gpu_set_recombination_rates_verner(cmi_gpu_engine *e) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  e->model.recomb_verner = 1;
  e->have_recomb = true;
  return CMI_GPU_OK;
}

int cmi_gpu_set_abundances(cmi_gpu_engine *e, const double *abundances) {
  if (!e || !abundances)
    return fail(CMI_GPU_EINVAL, "abundances: bad argument");
  for (int i = 0; i < 6; ++i)
    e->model.abundance[i] = abundances[i];
  return CMI_GPU_OK;
}

int cmi_gpu_set_reemission(cmi_gpu_engine *e, int32_t type,
                           double fixed_probability, double fixed_frequency) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (type != CMI_GPU_REEMIT_NONE && type != CMI_GPU_REEMIT_PHYSICAL &&
      type != CMI_GPU_REEMIT_FIXED)
    return fail(CMI_GPU_EINVAL,
                "Unknown DiffuseReemissionHandler type: %d", type);
  e->model.reemit_type = type;
  e->spectra_dirty = true;
  e->model.reemit_fixed_probability = fixed_probability;
  e->model.reemit_fixed_frequency = fixed_frequency;
  return CMI_GPU_OK;
}

int cmi_gpu_set_temperature_params(cmi_gpu_engine *e,
                                   const cmi_gpu_temperature_params *params) {
  if (!e || !params)
    return fail(CMI_GPU_EINVAL, "temperature params: bad argument");
  e->tparams = *params;
  ModelDev &m = e->model;
  m.t_epsilon = params->epsilon_convergence;
  m.t_max_iterations = params->maximum_number_of_iterations;
  m.pahfac = params->pah_heating_factor;
  m.crfac = params->cosmic_ray_heating_factor;
  m.crlim = params->cosmic_ray_heating_limit;
  m.crscale = params->cosmic_ray_heating_scale_length;
  m.t_min_ionized = params->minimum_ionized_temperature;
  return CMI_GPU_OK;
}

int cmi_gpu_upload_cells(cmi_gpu_engine *e, const double *number_density,
                         const double *temperature,
                         const double *ionic_fractions) {
  if (!e || !number_density || !temperature)
    return fail(CMI_GPU_EINVAL, "upload_cells: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  const size_t bytes = (size_t)e->ncell * sizeof(double);
  HIP_TRY(hipMemcpyAsync(e->cells.number_density, number_density, bytes,
                         hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->cells.temperature, temperature, bytes,
                         hipMemcpyHostToDevice, e->stream));
  if (ionic_fractions) {
    HIP_TRY(hipMemcpyAsync(e->cells.x[0], ionic_fractions, CMI_NION * bytes,
                           hipMemcpyHostToDevice, e->stream));
  } else {
    HIP_TRY(hipMemsetAsync(e->cells.x[0], 0, CMI_NION * bytes, e->stream));
  }
  int rc = rebuild_opacity(e);
  if (rc)
    return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->have_cells = true;
  return CMI_GPU_OK;
}

int cmi_gpu_upload_field(cmi_gpu_engine *e, int32_t field,
                         const double *values) {
  if (!e || !values)
    return fail(CMI_GPU_EINVAL, "upload_field: bad argument");
  double *dst = field_pointer(e, field);
  if (!dst)
    return fail(CMI_GPU_EINVAL, "upload_field: unknown field %d", field);
  HIP_TRY(hipSetDevice(e->device));
  const int64_t stride = field_stride(e, field);
  if (stride == 1) {
    HIP_TRY(hipMemcpyAsync(dst, values, (size_t)e->ncell * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
  } else {
    double *tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)e->ncell * sizeof(double)));
    hipError_t err = hipMemcpyAsync(tmp, values,
                                    (size_t)e->ncell * sizeof(double),
                                    hipMemcpyHostToDevice, e->stream);
    if (err == hipSuccess) {
      scatter_strided_kernel<<<grid_blocks(e, e->ncell, 8), CMI_BLOCK, 0,
                               e->stream>>>(tmp, dst, stride, e->ncell);
      err = hipGetLastError();
    }
    if (err == hipSuccess)
      err = hipStreamSynchronize(e->stream);
    (void)hipFree(tmp);
    HIP_TRY(err);
  }
  if (field >= CMI_GPU_FIELD_MEAN_INTENSITY)
    e->acc_block_dirty = true; /* an accumulator field */
  if (field == CMI_GPU_FIELD_NUMBER_DENSITY ||
      field == CMI_GPU_FIELD_IONIC_FRACTION + ION_H_n ||
      field == CMI_GPU_FIELD_IONIC_FRACTION + ION_He_n) {
    int rc = rebuild_opacity(e);
    if (rc)
      return rc;
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  return CMI_GPU_OK;
}

int cmi_gpu_download_field(cmi_gpu_engine *e, int32_t field, double *values) {
  if (!e || !values)
    return fail(CMI_GPU_EINVAL, "download_field: bad argument");
  double *src = field_pointer(e, field);
  if (!src)
    return fail(CMI_GPU_EINVAL, "download_field: unknown field %d", field);
  HIP_TRY(hipSetDevice(e->device));
  const int64_t stride = field_stride(e, field);
  if (stride == 1) {
    HIP_TRY(hipMemcpyAsync(values, src, (size_t)e->ncell * sizeof(double),
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  } else {
    double *tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)e->ncell * sizeof(double)));
    gather_strided_kernel<<<grid_blocks(e, e->ncell, 8), CMI_BLOCK, 0,
                            e->stream>>>(src, stride, tmp, e->ncell);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess)
      err = hipMemcpyAsync(values, tmp, (size_t)e->ncell * sizeof(double),
                           hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess)
      err = hipStreamSynchronize(e->stream);
    (void)hipFree(tmp);
    HIP_TRY(err);
  }
  return CMI_GPU_OK;
}

int cmi_gpu_accumulator_layout(cmi_gpu_engine *e, int64_t *field_stride_out,
                               int64_t *cell_stride_out) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (field_stride_out)
    *field_stride_out = e->cells.acc_field_stride;
  if (cell_stride_out)
    *cell_stride_out = e->cells.acc_cell_stride;
  return CMI_GPU_OK;
}

void *cmi_gpu_field_device_pointer(cmi_gpu_engine *e, int32_t field) {
  if (!e)
    return nullptr;
  return field_pointer(e, field);
}

int cmi_gpu_reset_grid(cmi_gpu_engine *e) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  HIP_TRY(hipSetDevice(e->device));
  const size_t field_bytes = (size_t)e->ncell * sizeof(double);
  if (e->full_ions || e->acc_block_dirty) {
    HIP_TRY(hipMemsetAsync(e->acc_block, 0, CMI_NACC * field_bytes,
                           e->stream));
    e->acc_block_dirty = false;
  } else {
    /* hydrogen-only runs add to J_H (and the two heating fields) only - the
     * other thirteen fields of the [16][ncell] block stay as zero as the
     * layout switch left them (update_full_flag); whatever else writes one
     * of them marks the block dirty */
    HIP_TRY(hipMemsetAsync(e->acc_block, 0, field_bytes, e->stream));
    if (e->config.track_heating)
      HIP_TRY(hipMemsetAsync(e->acc_block + (size_t)CMI_NION * e->ncell, 0,
                             2 * field_bytes, e->stream));
  }
  HIP_TRY(hipMemsetAsync(e->counters, 0,
                         sizeof(CountersDev) * CMI_COUNTER_SHARDS, e->stream));
  return CMI_GPU_OK;
}

int cmi_gpu_set_tuning(cmi_gpu_engine *e, const char *key, int64_t value) {
  if (!e || !key)
    return fail(CMI_GPU_EINVAL, "set_tuning: bad argument");
  const std::string k(key);
  if (k == "sort_packets")
    e->tune.sort_packets = value != 0;
  else if (k == "sort_dir_bits")
    e->tune.sort_dir_bits =
        (int)(value < 0 ? -1 : (value < 2 ? 2 : (value > 22 ? 22 : value)));
  else if (k == "sort_tau_bits")
    e->tune.sort_tau_bits = (int)(value < 0 ? -1 : (value > 3 ? 3 : value));
  else if (k == "aggregate")
    e->tune.aggregate = (int)(value < 0 ? 0 : (value > 2 ? 2 : value));
  else if (k == "aggregate_reemit")
    e->tune.aggregate_reemit = (int)(value < 0 ? 0 : (value > 2 ? 2 : value));
  else if (k == "refill_threshold")
    e->tune.refill_threshold = (int)(value < 1 ? 1 : (value > 64 ? 64 : value));
  else if (k == "chunk")
    e->tune.chunk = (uint32_t)(value < 64 ? 64 : value);
  else if (k == "max_blocks_per_cu")
    e->tune.max_blocks_per_cu = (int)(value < 1 ? 1 : value);
  else if (k == "max_packets_per_launch")
    e->tune.max_packets_per_launch =
        (uint64_t)(value < 1024 ? 1024 : (value > (1ll << 30) ? (1ll << 30) : value));
  else if (k == "exp_no_atomics")
    e->tune.exp_no_atomics = (int)value;
  else if (k == "exact_dda")
    e->tune.exact_dda = value != 0;
  else if (k == "reemit_passes")
    e->tune.reemit_passes = value != 0;
  else if (k == "refill_threshold_reemit")
    e->tune.refill_threshold_reemit =
        (int)(value < 1 ? 1 : (value > 64 ? 64 : value));
  else if (k == "reemit_inline_below")
    e->tune.reemit_inline_below = value < 0 ? -1 : value;
  else if (k == "reemit_max_passes")
    e->tune.reemit_max_passes = (int)(value < 1 ? 1 : value);
  else if (k == "timing")
    e->timing = value != 0;
  else if (k == "tile_rounds")
    e->tune.tile_rounds = value != 0;
  else if (k == "tile_min_flights")
    e->tune.tile_min_flights = (uint64_t)(value < 0 ? 0 : value);
  else if (k == "tile_min_per_item")
    e->tune.tile_min_per_item = (int)(value < 0 ? -1 : value);
  else if (k == "tile_refill_threshold")
    e->tune.tile_refill_threshold =
        (int)(value < 1 ? 1 : (value > 64 ? 64 : value));
  else if (k == "tile_compact_ratio")
    e->tune.tile_compact_ratio = (int)(value < -1 ? -1 : value);
  else if (k == "park_in_place")
    e->tune.park_in_place = value != 0;
  else if (k == "block_select")
    e->tune.block_select = value != 0;
  else if (k == "block_first_kernels")
    e->tune.block_first_kernels = value != 0;
  else if (k == "accumulators_dirty")
    e->acc_block_dirty = e->acc_block_dirty || value != 0;
  else if (k == "temperature_finish_slots")
    e->tune.temperature_finish_slots = (uint32_t)(value < 0 ? 0 : value);
  else if (k == "temperature_pipeline")
    e->tune.temperature_pipeline = value != 0;
  else if (k == "pad_march")
    e->tune.pad_march = value != 0;
  else if (k == "xcd_remap")
    e->tune.xcd_remap = value != 0;
  else if (k == "pre_emission")
    e->tune.pre_emission = value != 0;
  else if (k == "defer_weights")
    e->tune.defer_weights = value != 0;
  else if (k == "tile_counting_sort")
    e->tune.tile_counting_sort = value != 0;
  else if (k == "tile_max_rounds")
    e->tune.tile_max_rounds = (int)(value < 0 ? 0 : value);
  else
    return fail(CMI_GPU_EINVAL, "set_tuning: unknown key '%s'", key);
  return CMI_GPU_OK;
}

/* make sure the sort buffers hold n packets */
static int reserve_sort_buffers(cmi_gpu_engine *e, uint64_t n) {
  if (e->sort_capacity >= n)
    return CMI_GPU_OK;
  HIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(e->sort_keys[0]);
  (void)hipFree(e->sort_temp);
  e->sort_keys[0] = nullptr;
  e->sort_temp = nullptr;
  e->sort_capacity = 0;
  uint32_t *block = nullptr;
  HIP_TRY(hipMalloc(&block, sizeof(uint32_t) * 4 * n));
  e->sort_keys[0] = block;
  e->sort_keys[1] = block + n;
  e->sort_ids[0] = block + 2 * n;
  e->sort_ids[1] = block + 3 * n;
  size_t bytes = 0;
  HIP_TRY(cmi_sort_pairs_temp_bytes(n, 32, &bytes));
  e->sort_temp_bytes = bytes;
  HIP_TRY(hipMalloc(&e->sort_temp, bytes ? bytes : 16));
  e->sort_capacity = n;
  return CMI_GPU_OK;
}

/* make sure the two re-emission queues hold n packets each */
static int reserve_queues(cmi_gpu_engine *e, uint64_t n) {
  if (e->queue_capacity >= n)
    return CMI_GPU_OK;
  HIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(e->queue_block);
  e->queue_block = nullptr;
  e->queue_capacity = 0;
  /* ended flights: pos[3], nu + cell, id, meta = 5.5 doubles per packet;
   * ready flights: pos[3], dir[3], tau, nu + id, meta = 9 doubles per packet;
   * every packet of a launch can be in either */
  const size_t ended_doubles = 6, ready_doubles = 9;
  HIP_TRY(hipMalloc(&e->queue_block,
                    sizeof(double) * (ended_doubles + ready_doubles) * n));
  if (!e->queue_counts) {
    HIP_TRY(hipMalloc(&e->queue_counts, 2 * sizeof(unsigned int)));
  }
  QueueDev &q = e->ended_queue;
  double *base = e->queue_block;
  for (int a = 0; a < 3; ++a) {
    q.pos[a] = base + (size_t)a * n;
    q.dir[a] = nullptr;
  }
  q.tau = nullptr;
  q.nu = base + (size_t)3 * n;
  q.cell = (int32_t *)(base + (size_t)4 * n);
  q.id = (uint32_t *)q.cell + n;
  q.meta = q.id + n;
  q.count = e->queue_counts;
  QueueDev &r = e->ready_queue;
  base = e->queue_block + ended_doubles * n;
  for (int a = 0; a < 3; ++a) {
    r.pos[a] = base + (size_t)a * n;
    r.dir[a] = base + (size_t)(3 + a) * n;
  }
  r.tau = base + (size_t)6 * n;
  r.nu = base + (size_t)7 * n;
  r.cell = nullptr;
  r.id = (uint32_t *)(base + (size_t)8 * n);
  r.meta = r.id + n;
  r.count = e->queue_counts + 1;
  e->queue_capacity = n;
  return CMI_GPU_OK;
}

__global__ void iota_kernel(uint32_t *out, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    out[i] = (uint32_t)i;
}

/* tiles of the engine's grid for the current transport flavour */
static TileGridDev tile_grid(const cmi_gpu_engine *e) {
  TileGridDev t;
  const bool heat = e->config.track_heating != 0;
  if (e->full_ions) {
    t.log2[0] = TileShape<true, true>::LX;
    t.log2[1] = TileShape<true, true>::LY;
    t.log2[2] = TileShape<true, true>::LZ;
  } else if (heat) {
    t.log2[0] = TileShape<false, true>::LX;
    t.log2[1] = TileShape<false, true>::LY;
    t.log2[2] = TileShape<false, true>::LZ;
  } else {
    t.log2[0] = TileShape<false, false>::LX;
    t.log2[1] = TileShape<false, false>::LY;
    t.log2[2] = TileShape<false, false>::LZ;
  }
  int64_t total = 1;
  for (int a = 0; a < 3; ++a) {
    const int side = 1 << t.log2[a];
    t.ntile[a] = (e->grid.ncell[a] + side - 1) / side;
    total *= t.ntile[a];
  }
  t.ntiles = (int32_t)total;
  return t;
}

/* make sure the flight rows of the tile rounds hold n flights each */
static int reserve_tile_buffers(cmi_gpu_engine *e, uint64_t n) {
  const bool weights = e->full_ions;
  if (e->tile_capacity >= n && (e->tile_has_weights || !weights))
    return CMI_GPU_OK;
  HIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(e->tile_block);
  e->tile_block = nullptr;
  e->tile_capacity = 0;
  const TileGridDev t = tile_grid(e);
  const size_t row_bytes = sizeof(double) * CMI_FLIGHT_DOUBLES * n;
  const size_t weight_bytes = weights ? sizeof(double) * CMI_NACC * n : 0;
  const size_t key_bytes = (sizeof(uint32_t) * n + 255) & ~(size_t)255;
  const size_t nitems =
      (size_t)t.ntiles + n / CMI_TILE_ITEM_FLIGHTS_FULL + 4;
  const size_t item_bytes = sizeof(TileItemDev) * nitems;
  const size_t begin_bytes =
      (sizeof(uint32_t) * ((size_t)t.ntiles + 2) + 255) & ~(size_t)255;
  const size_t count_bytes =
      (sizeof(unsigned int) * nitems + 255) & ~(size_t)255;
  const bool counting = t.ntiles <= CMI_TILE_SORT_MAX_TILES;
  const size_t hist_bytes =
      counting ? sizeof(uint32_t) * (size_t)t.ntiles * CMI_TILE_SORT_BLOCKS : 0;
  const size_t total = 2 * (row_bytes + weight_bytes + key_bytes) +
                       6 * key_bytes + 2 * begin_bytes + 2 * count_bytes +
                       item_bytes + hist_bytes;
  HIP_TRY(hipMalloc(&e->tile_block, total));
  if (!e->tile_counts)
    HIP_TRY(hipMalloc(&e->tile_counts, 8 * sizeof(unsigned int)));
  char *at = e->tile_block;
  for (int k = 0; k < 2; ++k) {
    FlightRowsDev &r = e->tile_rows[k];
    r.rows = (double *)at;
    at += row_bytes;
    r.weights = weights ? (double *)at : nullptr;
    at += weight_bytes;
    r.keys = (uint32_t *)at;
    at += key_bytes;
    r.count = e->tile_counts + k;
    r.capacity = (unsigned int)n;
  }
  e->tile_iota = (uint32_t *)at;
  at += key_bytes;
  e->tile_ended_slot = (uint32_t *)at;
  at += key_bytes;
  e->tile_ended_pos = (uint32_t *)at;
  at += key_bytes;
  for (int k = 0; k < 2; ++k) {
    e->tile_slot_of[k] = (uint32_t *)at;
    at += key_bytes;
  }
  e->tile_new_slots = (uint32_t *)at;
  at += key_bytes;
  e->tile_begin = (uint32_t *)at;
  at += begin_bytes;
  e->tile_total = counting ? (uint32_t *)at : nullptr;
  at += begin_bytes;
  e->tile_absorbed_count = (unsigned int *)at;
  at += count_bytes;
  e->tile_absorbed_before = (unsigned int *)at;
  at += count_bytes;
  e->tile_items = (TileItemDev *)at;
  at += item_bytes;
  e->tile_blockhist = counting ? (uint32_t *)at : nullptr;
  iota_kernel<<<grid_blocks(e, (int64_t)n, 8), CMI_BLOCK, 0, e->stream>>>(
      e->tile_iota, n);
  HIP_TRY(hipGetLastError());
  e->tile_capacity = n;
  e->tile_has_weights = weights;
  return CMI_GPU_OK;
}

/* Transport of n_packets flights and of everything they re-emit: new packets
 * (flights == NULL) or flights handed over by other blocks of a decomposed
 * grid (device rows of CMI_FLIGHT_DOUBLES doubles). */
/* A block of a decomposed grid flies the packets that start in it (after at
 * most one step of length zero, see shoot_kernel): no source within one cell
 * of the block - and, for the block at the grid's origin, none outside the
 * whole grid - means nothing to emit, and the pass over the packet ids can be
 * skipped altogether. */
/* n <= 16 counters at `src` (device memory) as they are once the stream has
 * run dry */
static int read_counters(cmi_gpu_engine *e, const unsigned int *src, int n,
                         unsigned int *out) {
  {
    int rc = ensure_mailbox(e);
    if (rc)
      return rc;
  }
  mailbox_kernel<<<1, 16, 0, e->stream>>>(src, e->mailbox_dev, n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int k = 0; k < n; ++k)
    out[k] = ((volatile unsigned int *)e->mailbox)[k];
  return CMI_GPU_OK;
}

static bool block_emits_nothing(const cmi_gpu_engine *e) {
  if (e->model.continuous_type != 0)
    return false; /* its packets enter through every face of the box */
  const GridDev &g = e->grid;
  const bool at_origin = (g.offset[0] | g.offset[1] | g.offset[2]) == 0;
  for (int32_t s = 0; s < e->model.nsource; ++s) {
    bool near = true, in_grid = true;
    for (int a = 0; a < 3; ++a) {
      const double x = e->source_position_host[3 * (size_t)s + a];
      const double lo = g.anchor[a] + g.cellside[a] * (g.offset[a] - 1);
      const double hi =
          g.anchor[a] + g.cellside[a] * (g.offset[a] + g.ncell[a] + 1);
      near &= (x >= lo && x <= hi);
      /* (a cell of margin: the kernel decides by cell index) */
      in_grid &= (x >= g.anchor[a] + g.cellside[a] &&
                  x <= g.anchor[a] + g.box_sides[a] - g.cellside[a]);
    }
    if (near || (!in_grid && at_origin))
      return false;
  }
  return e->model.nsource > 0;
}

static uint64_t reemit_inline_below(const cmi_gpu_engine *e) {
  if (e->tune.reemit_inline_below >= 0)
    return (uint64_t)e->tune.reemit_inline_below;
  return e->grid.decomposed ? 262144u : 4096u;
}

static int shoot_impl(cmi_gpu_engine *e, uint32_t seed, uint32_t iteration,
                      uint64_t first_packet, uint64_t n_packets,
                      const double *flights) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (!e->have_sources || (e->model.nsource > 0 && !e->have_spectrum) ||
      (e->model.continuous_type != 0 && !e->have_continuous_spectrum) ||
      !e->have_xsec || !e->have_cells)
    return fail(CMI_GPU_ESTATE,
                "cmi_gpu_shoot: sources, their spectra, cross sections and "
                "cell data must be set first");
  if (n_packets == 0)
    return CMI_GPU_OK;
  /* (packet ids of a call are 32-bit, and 0xffffffff marks a place of the
   * ended queue that holds no flight: CMI_QUEUE_HOLE) */
  if (n_packets >= 0xffffffffull)
    return fail(CMI_GPU_EINVAL,
                "cmi_gpu_shoot: at most 2^32 - 2 packets per call");
  if (e->grid.decomposed &&
      (e->tune.exact_dda || !e->export_rows ||
       e->ncell >= CMI_FAST_MARCHER_MAX_CELLS))
    return fail(CMI_GPU_ESTATE,
                "cmi_gpu_shoot: a block of a decomposed grid needs an export "
                "buffer (cmi_gpu_set_export_buffer) and the incremental "
                "marcher (fewer than 2^28 cells per block)");
  if (!flights && e->grid.decomposed && block_emits_nothing(e))
    return CMI_GPU_OK;
  HIP_TRY(hipSetDevice(e->device));
  {
    int rc = ensure_spectra(e);
    if (rc)
      return rc;
  }

  const bool heat = e->config.track_heating != 0;
  const bool reemit = e->model.reemit_type != CMI_GPU_REEMIT_NONE;
  /* cross-lane aggregation keys and the fast marcher use 32-bit cell
   * indices, and the marcher a 32-bit BYTE offset into the 16-B transport
   * records (fast_load_record): 2^28 cells. Larger engines (768^3 and up;
   * 2^28 cells are 73 GB of state) march with the exact marcher. */
  const bool small_grid = e->ncell < CMI_FAST_MARCHER_MAX_CELLS;
  const int agg = small_grid ? e->tune.aggregate : CMI_AGG_NONE;
  const int agg_reemit = small_grid ? e->tune.aggregate_reemit : CMI_AGG_NONE;
  const bool tracking = e->trackers_enabled && e->trackers.n != 0;
  /* (trackers count in the exact marcher on an undivided grid - the counts
   * equal the oracle's one by one - and in the incremental one on the blocks
   * of a decomposed grid, whose hand-overs carry its state) */
  const bool exact = e->tune.exact_dda || !small_grid ||
                     (tracking && !e->grid.decomposed);
  /* with re-emission in passes the transport launches use the variant WITHOUT
   * the re-emission code (absorbed packets go to the interaction kernel); the
   * variant with it follows re-emissions in place */
  /* (a handful of handed-over flights - the late hand-over rounds of a
   * decomposed grid - are followed in place, re-emissions and all, by ONE
   * launch: a pass, the interaction kernel and the in-place kernel after it
   * each cost the latency of the longest flight, ~1 ms, whatever their
   * number) */
  const bool passes = reemit && e->tune.reemit_passes &&
                      !(flights && n_packets < reemit_inline_below(e));
  void (*kernel)(const ShootArgs) = nullptr;
  void (*kernel_inline)(const ShootArgs) = nullptr;
#define PICK(F, H, X)                                                          \
  if (e->full_ions == F && heat == H && exact == X) {                          \
    kernel = shoot_kernel<F, H, false, X>;                                     \
    kernel_inline = shoot_kernel<F, H, true, X>;                               \
  }
  PICK(false, false, false)
  PICK(false, true, false)
  PICK(true, false, false)
  PICK(true, true, false)
  PICK(false, false, true)
  PICK(false, true, true)
  PICK(true, false, true)
  PICK(true, true, true)
#undef PICK
  if (tracking && !exact) {
    /* (a block of a decomposed grid: the hook in the incremental marcher) */
#define PICK_TRACK(F, H)                                                       \
  if (e->full_ions == F && heat == H) {                                        \
    kernel = shoot_kernel<F, H, false, false, false, false, false, true>;     \
    kernel_inline = shoot_kernel<F, H, true, false, false, false, false, true>; \
  }
    PICK_TRACK(false, false)
    PICK_TRACK(false, true)
    PICK_TRACK(true, false)
    PICK_TRACK(true, true)
#undef PICK_TRACK
  }
  if (reemit && !passes)
    kernel = kernel_inline;
  /* the first generation of new packets on a non-periodic grid with the
   * block combining table (every benchmark config): the specialised build of
   * the same kernel */
  void (*kernel_first)(const ShootArgs) = kernel;
  if (!flights && !exact && !tracking && agg == CMI_AGG_BLOCK &&
      kernel != kernel_inline &&
      !(e->grid.periodic[0] | e->grid.periodic[1] | e->grid.periodic[2])) {
    if (e->full_ions)
      kernel_first = heat ? shoot_kernel<true, true, false, false, true>
                          : shoot_kernel<true, false, false, false, true>;
    else
      kernel_first = heat ? shoot_kernel<false, true, false, false, true>
                          : shoot_kernel<false, false, false, false, true>;
  }
  /* ... and, hydrogen only on a whole grid, marching through padded records */
  /* (the kernel addresses padded records by 32-bit byte offsets:
   * (nx + 2)(ny + 2)(nz + 2) < 2^29, which a flat grid of fewer than 2^28
   * cells - 1 x 13400 x 13400 - can exceed) */
  const int64_t padded_cells =
      ((int64_t)e->grid.ncell[0] + 2 * CMI_PAD_LAYERS) *
      ((int64_t)e->grid.ncell[1] + 2 * CMI_PAD_LAYERS) *
      ((int64_t)e->grid.ncell[2] + 2 * CMI_PAD_LAYERS);
  /* (a block of a decomposed grid: its ghost layer says "left the block", the
   * end of the flight then decides between "left the box" and a hand-over) */
  const bool block_ok = !e->grid.decomposed || e->tune.block_first_kernels;
  const bool pad = kernel_first != kernel && !e->full_ions &&
                   e->tune.pad_march && block_ok &&
                   padded_cells < ((int64_t)1 << 29);
  const bool pad_big = pad && e->ncell > CMI_TABLE_BIG_CELLS;
  if (pad_big)
    kernel_first = heat ? shoot_kernel<false, true, false, false, true, false,
                                       true, false, true>
                        : shoot_kernel<false, false, false, false, true, false,
                                       true, false, true>;
  else if (pad)
    kernel_first =
        heat ? shoot_kernel<false, true, false, false, true, false, true>
             : shoot_kernel<false, false, false, false, true, false, true>;
  /* ... and, for multi-ion runs whose packets are sorted anyway, with the
   * emission physics done by the key kernel (the rows live in the second
   * weights buffer of the tile rounds, idle during the first generation) */
  void (*kernel_first_pre)(const ShootArgs) = nullptr;
  if (kernel_first != kernel && e->full_ions && e->tune.pre_emission &&
      e->tune.sort_packets && passes && e->tune.tile_rounds && block_ok)
    kernel_first_pre =
        heat ? shoot_kernel<true, true, false, false, true, true>
             : shoot_kernel<true, false, false, false, true, true>;

  /* (the hydrogen-only kernels built for the table run in larger blocks) */
  const int first_threads =
      (kernel_first != kernel && !e->full_ions)
          ? (pad_big ? shoot_block_threads<false, true, true>()
                     : shoot_block_threads<false, true>())
          : CMI_BLOCK;
  auto occupancy = [&](void (*k)(const ShootArgs), int &blocks_per_cu,
                       int threads = CMI_BLOCK) -> int {
    blocks_per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k,
                                                         threads, 0));
    if (blocks_per_cu < 1)
      blocks_per_cu = 1;
    if (blocks_per_cu > e->tune.max_blocks_per_cu)
      blocks_per_cu = e->tune.max_blocks_per_cu;
    return CMI_GPU_OK;
  };
  int blocks_per_cu = 0, blocks_per_cu_inline = 0;
  {
    int rc = occupancy(kernel, blocks_per_cu);
    if (rc)
      return rc;
    rc = occupancy(kernel_inline, blocks_per_cu_inline);
    if (rc)
      return rc;
  }

  /* (the specialised first-generation kernels may fit more blocks per CU) */
  int blocks_per_cu_first = blocks_per_cu;
  if (kernel_first != kernel) {
    int rc = occupancy(kernel_first_pre ? kernel_first_pre : kernel_first,
                       blocks_per_cu_first, first_threads);
    if (rc)
      return rc;
  }
  const bool sorted = e->tune.sort_packets && !flights;
  const uint64_t max_launch = e->tune.max_packets_per_launch;
  /* later generations in tile rounds: needs the incremental marcher */
  const bool tiles = passes && e->tune.tile_rounds && !exact && !tracking;
  /* a block of a decomposed grid picks its own packets out of each launch's
   * ids first (block_select_kernel): buffers for what it picks, not for all
   * ids (1 / 8 of them in config 5 - sort buffers, queues and flight slots
   * are 40 GB per 1e8 packets) */
  const bool select_mode =
      sorted && e->grid.decomposed && e->tune.block_select;
  auto reserve_for = [&](uint64_t cap) -> int {
    if (sorted || tiles) {
      int rc = reserve_sort_buffers(e, cap);
      if (rc)
        return rc;
    }
    if (passes) {
      int rc = reserve_queues(e, cap);
      if (rc)
        return rc;
    }
    if (tiles) {
      int rc = reserve_tile_buffers(e, cap);
      if (rc)
        return rc;
    }
    return CMI_GPU_OK;
  };
  if (!select_mode) {
    int rc = reserve_for(n_packets < max_launch ? n_packets : max_launch);
    if (rc)
      return rc;
  }
  void (*tkernel)(const TileArgs) = nullptr;
  int tile_threads = 0, tile_blocks_per_cu = 0;
  if (tiles) {
    if (e->full_ions)
      tkernel = heat ? tile_kernel<true, true> : tile_kernel<true, false>;
    else
      tkernel = heat ? tile_kernel<false, true> : tile_kernel<false, false>;
    tile_threads = e->full_ions
                       ? TileShape<true, true>::THREADS
                       : (heat ? TileShape<false, true>::THREADS
                               : TileShape<false, false>::THREADS);
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
        &tile_blocks_per_cu, tkernel, tile_threads, 0));
    if (tile_blocks_per_cu < 1)
      tile_blocks_per_cu = 1;
  }
  QueueDev no_queue;
  memset(&no_queue, 0, sizeof no_queue);
  /* sort key: 22 direction bits, the tau class and the source index. The
   * tau classes only help when a packet's range follows from its optical
   * depth alone (one cross section for all packets). */
  uint32_t tau_bits = 0;
  {
    const uint64_t per_source =
        (n_packets < max_launch ? n_packets : max_launch) /
        (uint64_t)(e->model.nsource > 0 ? e->model.nsource : 1);
    if (e->tune.sort_tau_bits >= 0)
      tau_bits = (uint32_t)e->tune.sort_tau_bits;
    else
      /* measured on 256^3: the classes pay off once a direction bin of
       * 64 x 2^bits packets is still narrower than a few cells */
      /* (multi-ion runs since round 5's cell-by-cell sums: 0 / 1 / 2 / 3 /
       * 4 class bits 71.8 / 70.0 / 66.3 / 68.0 / 68.0 ms per 1e8 packets) */
      tau_bits = per_source >= (1ull << 24)
                     ? (e->full_ions ? 2u : 3u)
                     : (per_source >= (1ull << 22) ? 2u : 0u);
  }
  double sigma_ref = 1.;
  if (e->full_ions) {
    double sigma_He;
    const ModelDev host_model = host_model_of(e);
    cmi_cross_sections_H_He(host_model, 1.0001 * e->model.nu_H, sigma_ref,
                            sigma_He);
  }
  uint32_t source_bits = 0;
  /* (the continuous source counts as one more) */
  for (int s = e->model.nsource - (e->model.continuous_type != 0 ? 0 : 1);
       s > 0; s >>= 1)
    ++source_bits;
  if (source_bits > 10u - tau_bits)
    source_bits = 10u - tau_bits;
  uint32_t dir_bits = 22u;
  if (e->tune.sort_dir_bits >= 0) {
    dir_bits = (uint32_t)e->tune.sort_dir_bits;
  } else {
    /* measured on 256^3, 1e8 packets: 21 bits order the packets as well as
     * 22 (20 nearly, 18 not), and 21 + 3 tau bits are three passes, not four */
    const uint32_t over = (22u + tau_bits + source_bits) % 8u;
    if (over == 1u || over == 2u)
      dir_bits = 22u - over;
  }
  const int key_bits = (int)(dir_bits + tau_bits + source_bits);

  if (pad) {
    /* the padded records of this call's cell state (0.1 ms at 256^3) */
    const GridDev &g = e->grid;
    const int64_t padded = (int64_t)(g.ncell[0] + 2 * CMI_PAD_LAYERS) *
                           (g.ncell[1] + 2 * CMI_PAD_LAYERS) *
                           (g.ncell[2] + 2 * CMI_PAD_LAYERS);
    if (!e->pad_H)
      HIP_TRY(hipMalloc(&e->pad_H, sizeof(double) * (size_t)padded));
    pad_record_kernel<<<grid_blocks(e, padded, 8), CMI_BLOCK, 0, e->stream>>>(
        e->cells.opacity, e->pad_H, g.ncell[0], g.ncell[1], g.ncell[2]);
    HIP_TRY(hipGetLastError());
  }

  for (uint64_t done = 0; done < n_packets; done += max_launch) {
    /* the launch's packet ids ... */
    const uint64_t nids = n_packets - done < max_launch ? n_packets - done
                                                        : max_launch;
    /* ... and what it flies: all of them, or - a block of a decomposed grid -
     * those that start in the block */
    uint64_t n = nids;
    const uint32_t *select = nullptr;
    if (select_mode) {
      if (!e->select_count)
        HIP_TRY(hipMalloc(&e->select_count, sizeof(unsigned int)));
      if (e->select_capacity < nids) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        (void)hipFree(e->select_ids);
        e->select_ids = nullptr;
        e->select_capacity = 0;
        HIP_TRY(hipMalloc(&e->select_ids, sizeof(uint32_t) * nids));
        e->select_capacity = nids;
      }
      HIP_TRY(hipMemsetAsync(e->select_count, 0, sizeof(unsigned int),
                             e->stream));
      SelectArgs sa;
      sa.grid = e->grid;
      sa.model = e->model;
      sa.first_packet = first_packet + done;
      sa.batch_offset = done;
      sa.n_packets = nids;
      sa.seed = seed;
      sa.iteration = iteration;
      sa.select = e->select_ids;
      sa.count = e->select_count;
      block_select_kernel<false>
          <<<grid_blocks(e, (int64_t)nids, 8), CMI_BLOCK, 0, e->stream>>>(sa);
      HIP_TRY(hipGetLastError());
      unsigned int mine = 0;
      int rrc = read_counters(e, e->select_count, 1, &mine);
      if (rrc)
        return rrc;
      if (mine == 0)
        continue;
      n = mine;
      select = e->select_ids;
      /* (some headroom: the count differs by a per cent from one iteration
       * to the next, and growing means freeing and allocating again) */
      int rc = reserve_for(n + n / 16 + 1024);
      if (rc)
        return rc;
    }
    ShootArgs a;
    a.grid = e->grid;
    a.model = e->model;
    a.cells = e->cells;
    a.counters = e->counters;
    a.first_packet = first_packet;
    a.batch_offset = done;
    a.n_packets = n;
    a.order = nullptr;
    a.pre_rows = nullptr;
    a.xin = flights ? flights + (size_t)CMI_FLIGHT_DOUBLES * done : nullptr;
    a.xin_local = 0;
    a.xout.rows = e->export_rows;
    a.xout.count = e->export_count;
    a.xout.capacity = (unsigned int)e->export_capacity;
    a.pad_H = pad ? e->pad_H : nullptr;
    a.xcd_remap = (sorted && e->tune.xcd_remap) ? 1 : 0;
    a.pad_ny = e->grid.ncell[1] + 2 * CMI_PAD_LAYERS;
    a.pad_nz = e->grid.ncell[2] + 2 * CMI_PAD_LAYERS;
    a.pad_inv_yz = 1. / ((double)a.pad_ny * (double)a.pad_nz);
    a.pad_inv_z = 1. / (double)a.pad_nz;
    a.chunk = e->tune.chunk;
    a.seed = seed;
    a.iteration = iteration;
    /* handed-over flights are no ray bundles worth keeping together */
    a.refill_threshold = flights ? e->tune.refill_threshold_reemit
                                 : e->tune.refill_threshold;
    a.exp_no_atomics = e->tune.exp_no_atomics;
    a.trackers = e->trackers;
    if (!tracking)
      a.trackers.n = 0;
    a.aggregate = flights ? agg_reemit : agg;
    a.qin = no_queue;
    a.qout = no_queue;
    a.park_in_place = 0;
    if (passes) {
      HIP_TRY(hipMemsetAsync(e->queue_counts, 0, 2 * sizeof(unsigned int),
                             e->stream));
      a.qout = e->ended_queue;
      if (!flights && e->tune.park_in_place) {
        /* new packets: parked at their place in the launch's order */
        a.park_in_place = 1;
        HIP_TRY(hipMemsetAsync(e->ended_queue.id, 0xff,
                               sizeof(uint32_t) * (size_t)n, e->stream));
      }
    }

    EventPair ev;
    {
      int trc = timer_begin(e, ev);
      if (trc)
        return trc;
    }
    if (sorted) {
      KeyArgs k;
      k.model = e->model;
      k.first_packet = first_packet + done;
      k.n_packets = n;
      k.seed = seed;
      k.iteration = iteration;
      k.tau_bits = tau_bits;
      k.full_ions = e->full_ions ? 1 : 0;
      k.sigma_ref = sigma_ref;
      k.source_mask = (1u << source_bits) - 1u;
      /* coarse direction bins of ~64 x 2^tau_bits packets per source */
      k.dir_hi_bits = 0;
      k.dir_bits = dir_bits;
      if (tau_bits != 0) {
        const uint64_t per_bin = 64ull << tau_bits;
        /* (a block's own packets fill its part of the sphere as densely as
         * the launch's ids fill the whole) */
        const uint64_t per_source =
            nids / (uint64_t)(e->model.nsource > 0 ? e->model.nsource : 1);
        while (k.dir_hi_bits < dir_bits &&
               (per_source >> (k.dir_hi_bits + 1u)) >= per_bin)
          ++k.dir_hi_bits;
      }
      k.keys = e->sort_keys[0];
      k.ids = e->sort_ids[0];
      k.pre_rows = kernel_first_pre ? e->tile_rows[1].weights : nullptr;
      if (kernel_first_pre && select) {
        /* the rows are addressed by packet id within the launch (that is what
         * the transport kernel knows): a block that flies a selection of the
         * ids needs room for all of them - the flight slots it borrows the
         * room from otherwise are sized for its selection */
        if (e->select_rows_capacity < nids) {
          HIP_TRY(hipStreamSynchronize(e->stream));
          (void)hipFree(e->select_rows);
          e->select_rows = nullptr;
          e->select_rows_capacity = 0;
          HIP_TRY(hipMalloc(&e->select_rows,
                            sizeof(double) * CMI_NACC * (size_t)nids));
          e->select_rows_capacity = nids;
        }
        k.pre_rows = e->select_rows;
      }
      k.select = select;
      a.pre_rows = k.pre_rows;
      if (k.pre_rows)
        emission_key_kernel<<<grid_blocks(e, (int64_t)n, 8), CMI_BLOCK, 0,
                              e->stream>>>(k);
      else
        direction_key_kernel<<<grid_blocks(e, (int64_t)n, 8), CMI_BLOCK, 0,
                               e->stream>>>(k);
      HIP_TRY(hipGetLastError());
      HIP_TRY(cmi_sort_pairs(e->sort_temp, e->sort_temp_bytes, e->sort_keys[0],
                             e->sort_keys[1], e->sort_ids[0], e->sort_ids[1],
                             n, key_bits, e->stream));
      a.order = e->sort_ids[1];
    }
    /* enough chunks for every wave of a full grid, else fewer blocks */
    const uint64_t nchunks = (n + a.chunk - 1) / a.chunk;
    int64_t blocks = (int64_t)e->num_cu * blocks_per_cu_first;
    const int64_t need = (int64_t)((nchunks + (first_threads / 64) - 1) /
                                   (first_threads / 64));
    if (blocks > need)
      blocks = need;
    if (blocks < 1)
      blocks = 1;
    EventPair kev;
    {
      int trc = timer_begin(e, kev);
      if (trc)
        return trc;
    }
    if (sorted && kernel_first_pre)
      kernel_first_pre<<<(unsigned)blocks, CMI_BLOCK, 0, e->stream>>>(a);
    else
      kernel_first<<<(unsigned)blocks, first_threads, 0, e->stream>>>(a);
    HIP_TRY(hipGetLastError());
    {
      int trc = timer_end(e, e->kernel_events, kev, n);
      if (trc)
        return trc;
    }
    /* later generations, in tile rounds (tile_kernels.h): the interaction
     * kernel turns the ended flights into flight rows keyed by the tile they
     * start in; every round sorts the rows by tile, flies each flight through
     * ONE tile with the tile's accumulators in LDS, and collects the flights
     * that go on (into another tile, or re-emitted) for the next round */
    const double *handover = nullptr; /* flights the tile rounds leave over */
    unsigned int handover_count = 0;
    if (tiles) {
      const TileGridDev tg = tile_grid(e);
      int tile_bits = 1; /* keys: tiles and the "free slot" key ntiles */
      while ((1ll << tile_bits) < (int64_t)tg.ntiles + 1)
        ++tile_bits;
      InteractArgs ia;
      ia.model = e->model;
      ia.cells = e->cells;
      ia.counters = e->counters;
      ia.first_packet = a.first_packet;
      ia.seed = seed;
      ia.iteration = iteration;
      ia.qin = e->ended_queue;
      ia.n_in = a.park_in_place ? (uint32_t)n : 0u;
      ia.qout = no_queue;
      ia.grid = e->grid;
      ia.tiles = tg;
      ia.items = e->tile_items;
      ia.nitems = e->tile_counts + 2;
      ia.absorbed_before = e->tile_absorbed_before;
      ia.ended_slot = e->tile_ended_slot;
      const uint32_t item_flights = e->full_ions ? CMI_TILE_ITEM_FLIGHTS_FULL
                                                 : CMI_TILE_ITEM_FLIGHTS_H;
      const bool defer = e->full_ions && e->tune.defer_weights;
      unsigned int *const d_new = e->tile_counts + 4;
      ia.new_slots = e->tile_new_slots;
      ia.new_count = d_new;
      unsigned int *const d_nrows = e->tile_counts;
      unsigned int *const d_nlive = e->tile_counts + 1;
      unsigned int *const d_nitems = e->tile_counts + 2;
      unsigned int *const d_next = e->tile_counts + 3;
      HIP_TRY(hipMemsetAsync(e->tile_counts, 0, 8 * sizeof(unsigned int),
                             e->stream));
      /* the absorbed packets of the first generation -> flights in slots */
      int cur = 0; /* which set of rows holds the flights */
      FlightRowsDev rows = e->tile_rows[cur];
      ia.rows = rows;
      ia.rows.count = d_nrows;
      ia.ended_pos = nullptr;
      ia.key_out = nullptr;
      {
        const int iblocks = e->num_cu * 8;
        if (e->full_ions && defer)
          interaction_kernel<true, true, true>
              <<<iblocks, CMI_BLOCK, 0, e->stream>>>(ia);
        else if (e->full_ions)
          interaction_kernel<true, true>
              <<<iblocks, CMI_BLOCK, 0, e->stream>>>(ia);
        else
          interaction_kernel<false, true>
              <<<iblocks, CMI_BLOCK, 0, e->stream>>>(ia);
        HIP_TRY(hipGetLastError());
      }
      unsigned int nslots = 0;
      {
        int rrc = read_counters(e, d_nrows, 1, &nslots);
        if (rrc)
          return rrc;
      }
      if (nslots > rows.capacity)
        return fail(CMI_GPU_ENOMEM,
                    "tile rounds: %u flights, room for %u - flights were "
                    "lost, the iteration is invalid",
                    nslots, rows.capacity);
      if (defer && nslots != 0) {
        FlightWeightsArgs wa;
        wa.model = e->model;
        wa.rows = rows;
        wa.slots = nullptr;
        wa.count = nullptr;
        wa.n = nslots;
        flight_weights_kernel<<<grid_blocks(e, (int64_t)nslots, 8), CMI_BLOCK,
                                0, e->stream>>>(wa);
        HIP_TRY(hipGetLastError());
      }
      /* The flights of a round by position (TileArgs): keys[i] and the slot
       * of position i < npos. Round 0: the slots as the interaction kernel
       * filled them, position = slot; every round writes the arrays of the
       * next one in its tile order, flights only - the rows never move, and
       * what has ended is gone from the arrays one round later. */
      uint32_t *keys = rows.keys;
      uint32_t *keys_next = e->tile_rows[1].keys;
      const uint32_t *slot_of = nullptr;
      int next_slot_of = 0;
      unsigned int npos = nslots;
      /* slots the flights are spread over (since the last compaction) */
      unsigned int extent = nslots;
      const unsigned int compact_ratio =
          e->tune.tile_compact_ratio >= 0
              ? (unsigned int)e->tune.tile_compact_ratio
              : (e->full_ions ? 2u : 0u);
      for (int round = 0; npos != 0; ++round) {
        /* the positions in tile order (ended flights last), cut into units
         * of work */
        TilePlanArgs pa;
        pa.tiles = tg;
        pa.sorted_keys = e->sort_keys[1];
        pa.nslots = npos;
        pa.tile_begin = e->tile_begin;
        pa.item_flights = item_flights;
        pa.items = e->tile_items;
        pa.nitems = d_nitems;
        pa.next_item = d_next;
        pa.nlive = d_nlive;
        if (e->tile_blockhist && e->tune.tile_counting_sort) {
          TileSortArgs sa;
          sa.keys = keys;
          sa.nslots = npos;
          sa.ntiles = (uint32_t)tg.ntiles;
          /* (about a counter per slot and workgroup at least) */
          {
            uint64_t nb = ((uint64_t)npos / sa.ntiles + 7) / 8 * 8;
            if (nb < 8)
              nb = 8;
            if (nb > CMI_TILE_SORT_BLOCKS)
              nb = CMI_TILE_SORT_BLOCKS;
            sa.nblocks = (uint32_t)nb;
          }
          sa.blockhist = e->tile_blockhist;
          sa.total = e->tile_total;
          sa.tile_begin = e->tile_begin;
          sa.order = e->sort_ids[1];
          tile_count_kernel<<<sa.nblocks, CMI_TILE_SORT_THREADS, 0,
                              e->stream>>>(sa);
          tile_column_kernel<<<(sa.ntiles + CMI_BLOCK - 1) / CMI_BLOCK,
                               CMI_BLOCK, 0, e->stream>>>(sa);
          tile_offsets_kernel<<<1, CMI_TILE_SORT_THREADS, 0, e->stream>>>(sa);
          tile_scatter_kernel<<<sa.nblocks, CMI_TILE_SORT_THREADS, 0,
                                e->stream>>>(sa);
          HIP_TRY(hipGetLastError());
        } else {
          HIP_TRY(cmi_sort_pairs(e->sort_temp, e->sort_temp_bytes, keys,
                                 e->sort_keys[1], e->tile_iota,
                                 e->sort_ids[1], npos, tile_bits, e->stream));
          tile_begin_kernel<<<grid_blocks(e, (int64_t)npos + 1, 8),
                              CMI_BLOCK, 0, e->stream>>>(pa);
          HIP_TRY(hipGetLastError());
        }
        tile_plan_kernel<<<1, CMI_TILE_PLAN_THREADS, 0, e->stream>>>(pa);
        HIP_TRY(hipGetLastError());
        unsigned int plan[2] = {0, 0}; /* flights, units of work */
        {
          int rrc = read_counters(e, d_nlive, 2, plan);
          if (rrc)
            return rrc;
        }
        const unsigned int nlive = plan[0];
        if (nlive == 0)
          break;
        /* a unit of work costs a fixed ~10-20 us (tile records in, tile
         * accumulators out); measured on MI355X the round beats single
         * atomics while a unit has a few hundred flights to share that */
        const uint64_t per_item =
            e->tune.tile_min_per_item >= 0
                ? (uint64_t)e->tune.tile_min_per_item
                : 200u;
        const bool finish = nlive < e->tune.tile_min_flights ||
                            (uint64_t)nlive < per_item * plan[1] ||
                            round >= e->tune.tile_max_rounds;
        const uint32_t *order = e->sort_ids[1];
        if (finish || (compact_ratio != 0 &&
                       (uint64_t)compact_ratio * nlive < (uint64_t)extent)) {
          /* the live rows into the other set of rows, in tile order: position
           * j of this round is then place j and slot j */
          TileCompactArgs ca;
          ca.from = rows;
          ca.to = e->tile_rows[1 - cur];
          /* (the two key arrays change hands every round, whatever set of
           * rows is in use: the copies' keys go to the one that is free) */
          ca.to.keys = keys_next;
          ca.order = e->sort_ids[1];
          ca.slot_in = slot_of;
          ca.keys_in = keys;
          ca.nlive = d_nlive;
          ca.with_weights = e->full_ions ? 1 : 0;
          tile_compact_kernel<<<grid_blocks(e, 8ll * nlive, 8), CMI_BLOCK, 0,
                                e->stream>>>(ca);
          HIP_TRY(hipGetLastError());
          cur = 1 - cur;
          rows = e->tile_rows[cur];
          {
            uint32_t *t = keys;
            keys = keys_next;
            keys_next = t;
          }
          slot_of = nullptr;
          order = e->tile_iota;
          extent = nlive;
        }
        if (finish) {
          /* too few flights per tile for the LDS accumulators to pay: the
           * rest goes on as passes of the transport kernel (below), the first
           * of which resumes the flights from their (fresh, dense) rows */
          handover = rows.rows;
          handover_count = nlive;
          break;
        }
        TileArgs ta;
        ta.grid = e->grid;
        ta.model = e->model;
        ta.cells = e->cells;
        ta.counters = e->counters;
        ta.tiles = tg;
        ta.refill_threshold = e->tune.tile_refill_threshold;
        ta.rows = rows;
        ta.order = order;
        ta.slot_in = slot_of;
        ta.keys_out = keys_next;
        ta.slot_out = e->tile_slot_of[next_slot_of];
        ta.items = e->tile_items;
        ta.nitems = d_nitems;
        ta.next_item = d_next;
        ta.ended = e->ended_queue;
        ta.ended_slot = e->tile_ended_slot;
        ta.ended_pos = e->tile_ended_pos;
        ta.absorbed_count = e->tile_absorbed_count;
        ta.xout = a.xout;
        /* no more workgroups than units of work can exist */
        int64_t tb = (int64_t)e->num_cu * tile_blocks_per_cu;
        const int64_t most =
            (int64_t)tg.ntiles + (int64_t)nlive / item_flights + 1;
        if (tb > most)
          tb = most;
        EventPair tev;
        {
          int trc = timer_begin(e, tev);
          if (trc)
            return trc;
        }
        tkernel<<<(unsigned)tb, tile_threads, 0, e->stream>>>(ta);
        HIP_TRY(hipGetLastError());
        {
          int trc = timer_end(e, e->kernel_events, tev, nlive);
          if (trc)
            return trc;
        }
        ++e->tile_rounds_run;
        /* the packets absorbed in this round: re-emitted into their slots,
         * their new keys at their positions of the next round */
        ia.rows = rows;
        ia.ended_pos = e->tile_ended_pos;
        ia.key_out = keys_next;
        absorbed_scan_kernel<<<1, CMI_TILE_PLAN_THREADS, 0, e->stream>>>(
            d_nitems, e->tile_absorbed_count, e->tile_absorbed_before);
        HIP_TRY(hipGetLastError());
        /* (about a quarter of a round's flights are absorbed) */
        const int sblocks = grid_blocks(e, (int64_t)nlive / 2 + 1, 8);
        if (defer) {
          HIP_TRY(hipMemsetAsync(d_new, 0, sizeof(unsigned int), e->stream));
          interaction_slots_kernel<true, true>
              <<<sblocks, CMI_BLOCK, 0, e->stream>>>(ia);
          HIP_TRY(hipGetLastError());
          FlightWeightsArgs wa;
          wa.model = e->model;
          wa.rows = rows;
          wa.slots = e->tile_new_slots;
          wa.count = d_new;
          wa.n = 0;
          /* (about a tenth of a round's flights are re-emitted) */
          flight_weights_kernel<<<grid_blocks(e, (int64_t)nlive / 4 + 1, 8),
                                  CMI_BLOCK, 0, e->stream>>>(wa);
        } else if (e->full_ions)
          interaction_slots_kernel<true>
              <<<sblocks, CMI_BLOCK, 0, e->stream>>>(ia);
        else
          interaction_slots_kernel<false>
              <<<sblocks, CMI_BLOCK, 0, e->stream>>>(ia);
        HIP_TRY(hipGetLastError());
        /* the next round: this round's places are its positions */
        {
          uint32_t *t = keys;
          keys = keys_next;
          keys_next = t;
        }
        slot_of = e->tile_slot_of[next_slot_of];
        next_slot_of = 1 - next_slot_of;
        npos = nlive;
      }
    }
    /* ... or as passes of the transport kernel: the interaction kernel turns
     * the ended flights of one launch into the ready flights of the next.
     * Those start all over the grid in random directions, so these launches
     * refill eagerly instead of keeping ray bundles together. */
    if (handover_count != 0) {
      /* pass 0 of the tail: the flights resume from their slots, absorbed
       * ones are parked for the interaction kernel as in every pass */
      const bool last = handover_count < reemit_inline_below(e);
      ShootArgs b = a;
      b.park_in_place = 0;
      b.order = nullptr;
      b.xin = handover;
      b.xin_local = 1;
      b.n_packets = handover_count;
      b.refill_threshold = e->tune.refill_threshold_reemit;
      b.aggregate = agg_reemit;
      b.qin = no_queue;
      b.qout = last ? no_queue : e->ended_queue;
      if (!last)
        HIP_TRY(hipMemsetAsync(e->ended_queue.count, 0, sizeof(unsigned int),
                               e->stream));
      const uint64_t nch = ((uint64_t)handover_count + b.chunk - 1) / b.chunk;
      int64_t nb =
          (int64_t)e->num_cu * (last ? blocks_per_cu_inline : blocks_per_cu);
      const int64_t nneed =
          (int64_t)((nch + (CMI_BLOCK / 64) - 1) / (CMI_BLOCK / 64));
      if (nb > nneed)
        nb = nneed;
      if (nb < 1)
        nb = 1;
      EventPair gev;
      {
        int trc = timer_begin(e, gev);
        if (trc)
          return trc;
      }
      (last ? kernel_inline : kernel)<<<(unsigned)nb, CMI_BLOCK, 0,
                                        e->stream>>>(b);
      HIP_TRY(hipGetLastError());
      {
        int trc = timer_end(e, e->kernel_events, gev, handover_count);
        if (trc)
          return trc;
      }
      if (last)
        handover_count = 0;
    }
    for (int gen = 0; passes && (!tiles || handover_count != 0); ++gen) {
      InteractArgs ia;
      ia.model = e->model;
      ia.cells = e->cells;
      ia.counters = e->counters;
      ia.first_packet = a.first_packet;
      ia.seed = seed;
      ia.iteration = iteration;
      ia.qin = e->ended_queue;
      /* (the first pass reads what the first generation parked) */
      ia.n_in = (gen == 0 && !tiles && a.park_in_place) ? (uint32_t)n : 0u;
      ia.qout = e->ready_queue;
      HIP_TRY(hipMemsetAsync(e->ready_queue.count, 0, sizeof(unsigned int),
                             e->stream));
      const int iblocks = e->num_cu * 8;
      memset(&ia.rows, 0, sizeof ia.rows);
      ia.grid = e->grid;
      memset(&ia.tiles, 0, sizeof ia.tiles);
      ia.items = nullptr;
      ia.nitems = nullptr;
      ia.absorbed_before = nullptr;
      ia.ended_slot = nullptr;
      ia.ended_pos = nullptr;
      ia.key_out = nullptr;
      if (e->full_ions)
        interaction_kernel<true, false>
            <<<iblocks, CMI_BLOCK, 0, e->stream>>>(ia);
      else
        interaction_kernel<false, false>
            <<<iblocks, CMI_BLOCK, 0, e->stream>>>(ia);
      HIP_TRY(hipGetLastError());
      unsigned int count = 0;
      {
        int rrc = read_counters(e, e->ready_queue.count, 1, &count);
        if (rrc)
          return rrc;
      }
      if (count == 0)
        break;
      const bool last = count < reemit_inline_below(e) ||
                        gen + 2 >= e->tune.reemit_max_passes;
      ShootArgs b = a;
      b.park_in_place = 0;
      b.order = nullptr;
      b.xin = nullptr;
      b.n_packets = count;
      b.refill_threshold = e->tune.refill_threshold_reemit;
      b.aggregate = agg_reemit;
      b.qin = e->ready_queue;
      b.qout = last ? no_queue : e->ended_queue;
      if (!last)
        HIP_TRY(hipMemsetAsync(e->ended_queue.count, 0, sizeof(unsigned int),
                               e->stream));
      const uint64_t nch = ((uint64_t)count + b.chunk - 1) / b.chunk;
      int64_t nb =
          (int64_t)e->num_cu * (last ? blocks_per_cu_inline : blocks_per_cu);
      const int64_t nneed =
          (int64_t)((nch + (CMI_BLOCK / 64) - 1) / (CMI_BLOCK / 64));
      if (nb > nneed)
        nb = nneed;
      if (nb < 1)
        nb = 1;
      EventPair gev;
      {
        int trc = timer_begin(e, gev);
        if (trc)
          return trc;
      }
      /* the last pass follows whatever is still re-emitted in place */
      (last ? kernel_inline : kernel)<<<(unsigned)nb, CMI_BLOCK, 0,
                                        e->stream>>>(b);
      HIP_TRY(hipGetLastError());
      {
        int trc = timer_end(e, e->kernel_events, gev, count);
        if (trc)
          return trc;
      }
      if (last)
        break;
    }
    {
      int trc = timer_end(e, e->shoot_events, ev, 0);
      if (trc)
        return trc;
    }
  }
  return CMI_GPU_OK;
}

int cmi_gpu_shoot(cmi_gpu_engine *e, uint32_t seed, uint32_t iteration,
                  uint64_t first_packet, uint64_t n_packets) {
  return shoot_impl(e, seed, iteration, first_packet, n_packets, nullptr);
}

int cmi_gpu_shoot_flights(cmi_gpu_engine *e, uint32_t seed, uint32_t iteration,
                          uint64_t first_packet, const void *flights,
                          uint64_t n_flights) {
  if (!e || (!flights && n_flights))
    return fail(CMI_GPU_EINVAL, "shoot_flights: bad argument");
  if (!e->grid.decomposed)
    return fail(CMI_GPU_ESTATE,
                "shoot_flights: the engine is not a block of a decomposed "
                "grid");
  return shoot_impl(e, seed, iteration, first_packet, n_flights,
                    (const double *)flights);
}

int cmi_gpu_set_export_buffer(cmi_gpu_engine *e, void *rows,
                              uint64_t capacity) {
  if (!e || capacity >= (1ull << 32))
    return fail(CMI_GPU_EINVAL, "set_export_buffer: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (!e->export_count)
    HIP_TRY(hipMalloc(&e->export_count, sizeof(unsigned int)));
  HIP_TRY(hipMemsetAsync(e->export_count, 0, sizeof(unsigned int), e->stream));
  if (e->own_export_rows) {
    (void)hipFree(e->export_rows);
    e->own_export_rows = false;
  }
  e->export_rows = (double *)rows;
  if (!rows && capacity) {
    /* engine-owned buffer, for hosts that exchange through host memory */
    HIP_TRY(hipMalloc(&e->export_rows,
                      sizeof(double) * CMI_FLIGHT_DOUBLES * capacity));
    e->own_export_rows = true;
  }
  e->export_capacity = capacity;
  return CMI_GPU_OK;
}

int cmi_gpu_get_export_count(cmi_gpu_engine *e, uint64_t *count) {
  if (!e || !count)
    return fail(CMI_GPU_EINVAL, "get_export_count: bad argument");
  *count = 0;
  if (!e->export_count)
    return CMI_GPU_OK;
  HIP_TRY(hipSetDevice(e->device));
  unsigned int n = 0;
  HIP_TRY(hipMemcpyAsync(&n, e->export_count, sizeof n, hipMemcpyDeviceToHost,
                         e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (n > e->export_capacity)
    return fail(CMI_GPU_ENOMEM,
                "export buffer overflow: %u flights left the block, room for "
                "%llu - flights were lost, the iteration is invalid",
                n, (unsigned long long)e->export_capacity);
  *count = n;
  return CMI_GPU_OK;
}

int cmi_gpu_download_exports(cmi_gpu_engine *e, double *host_rows,
                             uint64_t capacity, uint64_t *count) {
  if (!e || !count || (!host_rows && capacity))
    return fail(CMI_GPU_EINVAL, "download_exports: bad argument");
  uint64_t n = 0;
  int rc = cmi_gpu_get_export_count(e, &n);
  if (rc)
    return rc;
  *count = n;
  if (n > capacity)
    return fail(CMI_GPU_ENOMEM,
                "download_exports: %llu flights, room for %llu",
                (unsigned long long)n, (unsigned long long)capacity);
  if (n) {
    HIP_TRY(hipMemcpyAsync(host_rows, e->export_rows,
                           sizeof(double) * CMI_FLIGHT_DOUBLES * n,
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return CMI_GPU_OK;
}

int cmi_gpu_shoot_flights_host(cmi_gpu_engine *e, uint32_t seed,
                               uint32_t iteration, uint64_t first_packet,
                               const double *host_rows, uint64_t n_flights) {
  if (!e || (!host_rows && n_flights))
    return fail(CMI_GPU_EINVAL, "shoot_flights_host: bad argument");
  if (n_flights == 0)
    return CMI_GPU_OK;
  HIP_TRY(hipSetDevice(e->device));
  if (e->import_capacity < n_flights) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipFree(e->import_rows);
    e->import_rows = nullptr;
    e->import_capacity = 0;
    HIP_TRY(hipMalloc(&e->import_rows,
                      sizeof(double) * CMI_FLIGHT_DOUBLES * n_flights));
    e->import_capacity = n_flights;
  }
  /* launches of an earlier call may still read the staging buffer */
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpyAsync(e->import_rows, host_rows,
                         sizeof(double) * CMI_FLIGHT_DOUBLES * n_flights,
                         hipMemcpyHostToDevice, e->stream));
  return cmi_gpu_shoot_flights(e, seed, iteration, first_packet, e->import_rows,
                               n_flights);
}

int cmi_gpu_reset_exports(cmi_gpu_engine *e) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (!e->export_count)
    return CMI_GPU_OK;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemsetAsync(e->export_count, 0, sizeof(unsigned int), e->stream));
  return CMI_GPU_OK;
}

int cmi_gpu_get_counters(cmi_gpu_engine *e, double *totweight,
                         double *typecount, uint64_t *nsteps) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  HIP_TRY(hipSetDevice(e->device));
  CountersDev host;
  {
    int rc = download_counters(e, host);
    if (rc)
      return rc;
  }
  if (totweight)
    *totweight = host.totweight;
  if (typecount)
    for (int i = 0; i < 4; ++i)
      typecount[i] = host.typecount[i];
  if (nsteps)
    *nsteps = host.nsteps;
  return CMI_GPU_OK;
}

int cmi_gpu_get_atomic_count(cmi_gpu_engine *e, uint64_t *natomics) {
  if (!e || !natomics)
    return fail(CMI_GPU_EINVAL, "get_atomic_count: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  CountersDev host;
  {
    int rc = download_counters(e, host);
    if (rc)
      return rc;
  }
  *natomics = host.natomics;
  return CMI_GPU_OK;
}

int cmi_gpu_get_wave_steps(cmi_gpu_engine *e, uint64_t *nwavesteps) {
  if (!e || !nwavesteps)
    return fail(CMI_GPU_EINVAL, "get_wave_steps: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  CountersDev host;
  {
    int rc = download_counters(e, host);
    if (rc)
      return rc;
  }
  *nwavesteps = host.nwavesteps;
  return CMI_GPU_OK;
}

/* the temperature solve of the cells [a.first, a.first + a.count) as the
 * pipeline of temperature_pipeline.h, in passes of at most 2^24 cells (the
 * solve state and a step's evaluations of a pass: 10 GB) */
static int temperature_pipeline(cmi_gpu_engine *e, const UpdateArgs &a) {
  const uint32_t want =
      (uint32_t)(a.count < (1ll << 24) ? a.count : (1ll << 24));
  const size_t state_doubles = (size_t)TS_NFIELD;
  const size_t eval_doubles = 3 * (size_t)TE_NFIELD;
  auto bytes_for = [&](uint32_t cap) {
    return sizeof(double) * (state_doubles + eval_doubles) * cap +
           sizeof(uint32_t) * 4 * (size_t)cap;
  };
  if (e->temp_pipe_capacity < want) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipFree(e->temp_pipe_block);
    e->temp_pipe_block = nullptr;
    e->temp_pipe_capacity = 0;
    HIP_TRY(hipMalloc(&e->temp_pipe_block, bytes_for(want)));
    e->temp_pipe_capacity = want;
  }
  if (!e->temp_pipe_counts)
    HIP_TRY(hipMalloc(&e->temp_pipe_counts, 4 * sizeof(unsigned int)));
  const uint32_t cap = e->temp_pipe_capacity;
  TempPipeArgs p;
  p.u = a;
  p.capacity = cap;
  char *at = e->temp_pipe_block;
  p.state = (double *)at;
  at += sizeof(double) * state_doubles * cap;
  p.eval = (double *)at;
  at += sizeof(double) * eval_doubles * cap;
  p.slot_cell = (uint32_t *)at;
  at += sizeof(uint32_t) * (size_t)cap;
  p.niter = (int32_t *)at;
  at += sizeof(uint32_t) * (size_t)cap;
  p.list[0] = (uint32_t *)at;
  at += sizeof(uint32_t) * (size_t)cap;
  p.list[1] = (uint32_t *)at;
  p.counts = e->temp_pipe_counts;
  for (int64_t done = 0; done < a.count; done += cap) {
    p.chunk_first = a.first + done;
    p.chunk_count = a.count - done < (int64_t)cap ? a.count - done : cap;
    p.current = 0;
    p.nactive = 0;
    HIP_TRY(hipMemsetAsync(p.counts, 0, 4 * sizeof(unsigned int), e->stream));
    temp_begin_kernel<<<grid_blocks(e, p.chunk_count, 8), CMI_BLOCK, 0,
                        e->stream>>>(p);
    HIP_TRY(hipGetLastError());
    unsigned int nactive = 0;
    {
      int rrc = read_counters(e, p.counts, 1, &nactive);
      if (rrc)
        return rrc;
    }
    /* (a solve ends after t_max_iterations steps at the latest) */
    while (nactive != 0) {
      p.nactive = nactive;
      if (nactive <= e->tune.temperature_finish_slots) {
        /* the stragglers: one launch, a wave per slot */
        temp_finish_kernel<<<(unsigned)((64ull * nactive + CMI_BLOCK - 1) /
                                        CMI_BLOCK),
                             CMI_BLOCK, 0, e->stream>>>(p);
        HIP_TRY(hipGetLastError());
        break;
      }
      const int eblocks = grid_blocks(e, 3ll * nactive, 8);
      temp_eval_kernel<<<eblocks, CMI_BLOCK, 0, e->stream>>>(p);
      temp_linecool_kernel<<<eblocks, CMI_BLOCK, 0, e->stream>>>(p);
      unsigned int *next_count = p.counts + 1 + (1 - p.current);
      HIP_TRY(hipMemsetAsync(next_count, 0, sizeof(unsigned int), e->stream));
      temp_secant_kernel<<<grid_blocks(e, (int64_t)nactive, 8), CMI_BLOCK, 0,
                           e->stream>>>(p);
      HIP_TRY(hipGetLastError());
      int rrc = read_counters(e, next_count, 1, &nactive);
      if (rrc)
        return rrc;
      p.current = 1 - p.current;
    }
  }
  return CMI_GPU_OK;
}

int cmi_gpu_update_cells(cmi_gpu_engine *e, uint32_t loop, double totweight) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  return cmi_gpu_update_cells_range(e, loop, totweight, 0, e->ncell);
}

int cmi_gpu_update_cells_range(cmi_gpu_engine *e, uint32_t loop,
                               double totweight, int64_t first_cell,
                               int64_t ncell) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (first_cell < 0 || ncell < 0 || first_cell + ncell > e->ncell)
    return fail(CMI_GPU_EINVAL, "update_cells_range: cells [%lld, %lld) are "
                "not inside the engine's %lld cells", (long long)first_cell,
                (long long)(first_cell + ncell), (long long)e->ncell);
  if (ncell == 0)
    return CMI_GPU_OK;
  if (!e->have_sources || !e->have_recomb || !e->have_cells)
    return fail(CMI_GPU_ESTATE,
                "cmi_gpu_update_cells: sources, recombination rates and cell "
                "data must be set first");
  if (!(totweight > 0.))
    return fail(CMI_GPU_EINVAL, "update_cells: totweight must be positive");
  HIP_TRY(hipSetDevice(e->device));
  const bool solve_temperature =
      e->tparams.do_temperature_calculation &&
      loop > (uint32_t)e->tparams.minimum_number_of_iterations;

  UpdateArgs a;
  a.grid = e->grid;
  a.model = e->model;
  a.cells = e->cells;
  /* src/IonizationStateCalculator.cpp:519-522 and :136-137 */
  const double jfac = e->model.total_luminosity / totweight;
  const double hfac = jfac * CMI_PLANCK;
  const double volume =
      e->grid.cellside[0] * e->grid.cellside[1] * e->grid.cellside[2];
  a.jfac = jfac / volume;
  a.hfac = hfac / volume;
  a.first = first_cell;
  a.count = ncell;

  EventPair ev;
  {
    int trc = timer_begin(e, ev);
    if (trc)
      return trc;
  }
  const int blocks = grid_blocks(e, ncell, 8);
  if (solve_temperature && e->tune.temperature_pipeline) {
    int prc = temperature_pipeline(e, a);
    if (prc)
      return prc;
  } else if (solve_temperature)
    temperature_kernel<<<blocks, CMI_BLOCK, 0, e->stream>>>(a);
  else if (e->full_ions)
    ionization_kernel<true><<<blocks, CMI_BLOCK, 0, e->stream>>>(a);
  else if (e->config.track_heating)
    ionization_kernel<false><<<blocks, CMI_BLOCK, 0, e->stream>>>(a);
  else
    ionization_kernel<false, false><<<blocks, CMI_BLOCK, 0, e->stream>>>(a);
  HIP_TRY(hipGetLastError());
  {
    int trc = timer_end(e, e->update_events, ev, 0);
    if (trc)
      return trc;
  }
  return CMI_GPU_OK;
}

int cmi_gpu_compute_emissivities(cmi_gpu_engine *e, int32_t nlines,
                                 const int32_t *lines, int64_t first_cell,
                                 int64_t ncell, double *emissivities) {
  if (!e || !lines || !emissivities)
    return fail(CMI_GPU_EINVAL, "compute_emissivities: null argument");
  if (nlines < 1 || nlines > CMI_NEMISSIONLINE)
    return fail(CMI_GPU_EINVAL, "compute_emissivities: %d lines asked for, "
                "there are %d", (int)nlines, CMI_NEMISSIONLINE);
  if (first_cell < 0 || ncell < 0 || first_cell + ncell > e->ncell)
    return fail(CMI_GPU_EINVAL, "compute_emissivities: cells [%lld, %lld) are "
                "not inside the engine's %lld cells", (long long)first_cell,
                (long long)(first_cell + ncell), (long long)e->ncell);
  if (!e->have_cells)
    return fail(CMI_GPU_ESTATE,
                "compute_emissivities: cell data must be set first");
  EmissivityArgs a;
  a.model = e->model;
  a.cells = e->cells;
  a.first = first_cell;
  a.count = ncell;
  a.nlines = nlines;
  for (int32_t l = 0; l < nlines; ++l) {
    if (lines[l] < 0 || lines[l] >= CMI_NEMISSIONLINE)
      return fail(CMI_GPU_EINVAL, "compute_emissivities: no emission line %d",
                  (int)lines[l]);
    a.lines[l] = lines[l];
  }
  if (ncell == 0)
    return CMI_GPU_OK;
  HIP_TRY(hipSetDevice(e->device));
  double *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(double) * (size_t)nlines * (size_t)ncell));
  a.out = d;
  emissivity_kernel<<<grid_blocks(e, ncell, 8), CMI_BLOCK, 0, e->stream>>>(a);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess)
    err = hipStreamSynchronize(e->stream);
  if (err == hipSuccess)
    err = hipMemcpy(emissivities, d,
                    sizeof(double) * (size_t)nlines * (size_t)ncell,
                    hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (err != hipSuccess)
    return fail(CMI_GPU_EDEVICE, "compute_emissivities: %s",
                hipGetErrorString(err));
  return CMI_GPU_OK;
}

int cmi_gpu_set_spectrum_trackers(cmi_gpu_engine *e, int32_t n,
                                  const double *positions, int32_t nbins,
                                  const double *opening_angles,
                                  const double *reference_directions) {
  if (n > CMI_MAX_TRACKERS)
    return fail(CMI_GPU_EINVAL, "set_spectrum_trackers: %d trackers, at most "
                "%d", (int)n, CMI_MAX_TRACKERS);
  int32_t bins[CMI_MAX_TRACKERS];
  for (int32_t k = 0; k < n && k < CMI_MAX_TRACKERS; ++k)
    bins[k] = nbins;
  return cmi_gpu_set_trackers(e, n, positions, nullptr, bins, opening_angles,
                              reference_directions);
}

int cmi_gpu_set_trackers(cmi_gpu_engine *e, int32_t n, const double *positions,
                         const int32_t *kinds, const int32_t *nbins,
                         const double *opening_angles,
                         const double *reference_directions) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (n < 0 || n > CMI_MAX_TRACKERS)
    return fail(CMI_GPU_EINVAL, "set_trackers: %d trackers, at most "
                "%d", (int)n, CMI_MAX_TRACKERS);
  if (n > 0 && (!positions || !nbins))
    return fail(CMI_GPU_EINVAL, "set_trackers: bad argument");
  for (int32_t k = 0; k < n; ++k)
    if (nbins[k] < 1)
      return fail(CMI_GPU_EINVAL, "set_trackers: tracker %d with %d bins",
                  (int)k, (int)nbins[k]);
  for (int32_t k = 0; kinds && k < n; ++k)
    if (kinds[k] != CMI_TRACKER_SPECTRUM &&
        kinds[k] != CMI_TRACKER_ABSORPTION && kinds[k] != CMI_TRACKER_WEIGHTED)
      return fail(CMI_GPU_EINVAL, "set_trackers: unknown kind %d of tracker "
                  "%d", (int)kinds[k], (int)k);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  (void)hipFree(e->trackers.counts);
  (void)hipFree(e->trackers.absorption);
  (void)hipFree(e->trackers.flux);
  e->trackers = TrackersDev();
  if (n == 0)
    return CMI_GPU_OK;
  TrackersDev t = TrackersDev();
  t.n = n;
  /* src/SpectrumTracker.hpp:88-90: nbins bins over three Rydberg frequencies */
  t.minimum_frequency = 3.289e15;
  t.first_bin[0] = 0;
  for (int32_t k = 0; k < n; ++k) {
    t.nbins[k] = nbins[k];
    t.first_bin[k + 1] = t.first_bin[k] + nbins[k];
    t.inverse_frequency_width[k] = 1. / (3. * 3.289e15 / nbins[k]);
    if (kinds && kinds[k] == CMI_TRACKER_WEIGHTED) {
      /* LinearFrequencyBins' defaults, src/LinearFrequencyBins.hpp:80-88:
       * from 13.6 eV to 54.4 eV (in Hz: the electronvolt over Planck's
       * constant, src/UnitConverter.hpp:156-159,271 with the values of
       * src/PhysicalConstants.hpp); cmi_gpu_set_tracker_frequency_bins for
       * others */
      const double ev = 1.6021766208e-19 / 6.626070040e-34;
      t.bins_type[k] = CMI_BINS_LINEAR;
      t.bins_min[k] = 13.6 * ev;
      t.bins_max[k] = 54.4 * ev;
      t.inverse_frequency_width[k] = nbins[k] / (t.bins_max[k] - t.bins_min[k]);
    }
  }
  /* LevelFrequencyBins::LevelFrequencyBins, src/LevelFrequencyBins.hpp:52-66:
   * the ionization energies of src/ElementData.hpp:39-105 (Hz) in ascending
   * order, closed by four times hydrogen's */
  {
    const double energies[CMI_NION] = {
        3.28810279e+15, 5.94523574e+15, 5.89588678e+15, 1.15792700e+16,
        3.51435505e+15, 7.15759434e+15, 1.14732262e+16, 3.29284691e+15,
        8.49136314e+15, 5.21432028e+15, 9.90492110e+15, 5.64310422e+15,
        8.41222200e+15, 1.14182796e+16};
    for (int i = 0; i < CMI_NION; ++i)
      t.level_edges[i] = energies[i];
    std::sort(t.level_edges, t.level_edges + CMI_NION);
    t.level_edges[CMI_NION] = 4. * energies[ION_H_n];
  }
  const GridDev &g = e->grid;
  for (int32_t k = 0; k < n; ++k) {
    /* the cell that holds the position (CartesianDensityGrid::get_cell_indices,
     * src/CartesianDensityGrid.cpp:152-161); TrackerManager::add_trackers
     * aborts for a position outside the box (src/TrackerManager.hpp:181-184) */
    int64_t cell = 0;
    bool here = true;
    for (int a = 0; a < 3; ++a) {
      const double x = positions[3 * k + a];
      if (!(x >= g.anchor[a] && x <= g.anchor[a] + g.box_sides[a]))
        return fail(CMI_GPU_EINVAL, "Tracker is not inside grid!");
      int64_t i = (int64_t)((x - g.anchor[a]) * g.inv_cellside[a]);
      if (i >= g.global_ncell[a])
        i = g.global_ncell[a] - 1;
      i -= g.offset[a];
      here = here && i >= 0 && i < g.ncell[a];
      cell = cell * g.ncell[a] + i;
    }
    t.cell[k] = here ? cell : -1;
    t.kind[k] = kinds ? kinds[k] : CMI_TRACKER_SPECTRUM;
    t.cos_opening_angle[k] = opening_angles ? cos(opening_angles[k]) : -1.;
    double norm2 = 0.;
    for (int a = 0; a < 3; ++a) {
      t.direction[k][a] =
          reference_directions ? reference_directions[3 * k + a] : 0.;
      norm2 += t.direction[k][a] * t.direction[k][a];
    }
    if (norm2 > 0.)
      for (int a = 0; a < 3; ++a)
        t.direction[k][a] /= sqrt(norm2);
  }
  const size_t bytes =
      sizeof(unsigned long long) * 3 * (size_t)t.first_bin[n];
  HIP_TRY(hipMalloc(&t.counts, bytes));
  HIP_TRY(hipMemsetAsync(t.counts, 0, bytes, e->stream));
  const size_t abytes = sizeof(double) * 4 * CMI_NION * (size_t)n;
  HIP_TRY(hipMalloc(&t.absorption, abytes));
  HIP_TRY(hipMemsetAsync(t.absorption, 0, abytes, e->stream));
  const size_t fbytes = sizeof(double) * 4 * (size_t)t.first_bin[n];
  HIP_TRY(hipMalloc(&t.flux, fbytes));
  HIP_TRY(hipMemsetAsync(t.flux, 0, fbytes, e->stream));
  e->trackers = t;
  return CMI_GPU_OK;
}

int cmi_gpu_set_tracker_frequency_bins(cmi_gpu_engine *e, int32_t tracker,
                                       int32_t type, double minimum_frequency,
                                       double maximum_frequency) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  if (tracker < 0 || tracker >= e->trackers.n ||
      e->trackers.kind[tracker] != CMI_TRACKER_WEIGHTED)
    return fail(CMI_GPU_EINVAL, "set_tracker_frequency_bins: tracker %d is "
                "not a weighted spectrum tracker", (int)tracker);
  TrackersDev &t = e->trackers;
  if (type == CMI_BINS_LEVEL) {
    if (t.nbins[tracker] != CMI_NION)
      return fail(CMI_GPU_EINVAL, "set_tracker_frequency_bins: level bins are "
                  "%d bins, tracker %d has %d", CMI_NION, (int)tracker,
                  (int)t.nbins[tracker]);
    t.bins_type[tracker] = CMI_BINS_LEVEL;
    return CMI_GPU_OK;
  }
  if (type != CMI_BINS_LINEAR)
    return fail(CMI_GPU_EINVAL, "Unknown FrequencyBins type: %d", (int)type);
  if (!(maximum_frequency > minimum_frequency))
    return fail(CMI_GPU_EINVAL, "set_tracker_frequency_bins: empty frequency "
                "range");
  t.bins_type[tracker] = CMI_BINS_LINEAR;
  t.bins_min[tracker] = minimum_frequency;
  t.bins_max[tracker] = maximum_frequency;
  t.inverse_frequency_width[tracker] =
      t.nbins[tracker] / (maximum_frequency - minimum_frequency);
  return CMI_GPU_OK;
}

int cmi_gpu_projected_areas(const double *directions, int64_t n,
                            double *areas) {
  if (n < 0 || (n > 0 && (!directions || !areas)))
    return fail(CMI_GPU_EINVAL, "projected_areas: bad argument");
  for (int64_t i = 0; i < n; ++i)
    areas[i] = cmi_projected_area(directions + 3 * i);
  return CMI_GPU_OK;
}

int cmi_gpu_get_tracker_flux(cmi_gpu_engine *e, double *flux) {
  if (!e || !flux)
    return fail(CMI_GPU_EINVAL, "get_tracker_flux: bad argument");
  if (e->trackers.n == 0)
    return fail(CMI_GPU_ESTATE, "get_tracker_flux: no trackers set");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(flux, e->trackers.flux,
                    sizeof(double) * 4 *
                        (size_t)e->trackers.first_bin[e->trackers.n],
                    hipMemcpyDeviceToHost));
  return CMI_GPU_OK;
}

int cmi_gpu_get_tracker_absorption(cmi_gpu_engine *e, double *absorption) {
  if (!e || !absorption)
    return fail(CMI_GPU_EINVAL, "get_tracker_absorption: bad argument");
  if (e->trackers.n == 0)
    return fail(CMI_GPU_ESTATE, "get_tracker_absorption: no trackers set");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(absorption, e->trackers.absorption,
                    sizeof(double) * 4 * CMI_NION * (size_t)e->trackers.n,
                    hipMemcpyDeviceToHost));
  return CMI_GPU_OK;
}

int cmi_gpu_enable_trackers(cmi_gpu_engine *e, int32_t enable) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  e->trackers_enabled = enable != 0;
  return CMI_GPU_OK;
}

int cmi_gpu_get_tracker_counts(cmi_gpu_engine *e, uint64_t *counts) {
  if (!e || !counts)
    return fail(CMI_GPU_EINVAL, "get_tracker_counts: bad argument");
  if (e->trackers.n == 0)
    return fail(CMI_GPU_ESTATE, "get_tracker_counts: no trackers set");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(counts, e->trackers.counts,
                         sizeof(unsigned long long) * 3 *
                             (size_t)e->trackers.first_bin[e->trackers.n],
                         hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return CMI_GPU_OK;
}

int cmi_gpu_refresh_transport_records(cmi_gpu_engine *e) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  HIP_TRY(hipSetDevice(e->device));
  return rebuild_opacity(e);
}

int cmi_gpu_emit_packets(cmi_gpu_engine *e, uint32_t seed, uint32_t iteration,
                         uint64_t first_packet, uint64_t n, double *position,
                         double *direction, double *frequency,
                         double *cross_sections, double *tau) {
  if (!e || !position || !direction || !frequency || !cross_sections || !tau)
    return fail(CMI_GPU_EINVAL, "emit_packets: bad argument");
  if (!e->have_sources || (e->model.nsource > 0 && !e->have_spectrum) ||
      (e->model.continuous_type != 0 && !e->have_continuous_spectrum) ||
      !e->have_xsec)
    return fail(CMI_GPU_ESTATE, "emit_packets: model not complete");
  if (n == 0)
    return CMI_GPU_OK;
  HIP_TRY(hipSetDevice(e->device));
  {
    int rc = ensure_spectra(e);
    if (rc)
      return rc;
  }
  double *d = nullptr;
  const size_t per = 3 + 3 + 1 + CMI_NION + 1;
  HIP_TRY(hipMalloc(&d, sizeof(double) * per * n));
  double *dpos = d, *ddir = d + 3 * n, *dnu = d + 6 * n, *dsig = d + 7 * n,
         *dtau = d + (7 + CMI_NION) * n;
  emit_probe_kernel<<<(unsigned)((n + 63) / 64), 64, 0, e->stream>>>(
      e->grid, e->model, seed, iteration, first_packet, n, dpos, ddir, dnu,
      dsig, dtau);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess)
    err = hipStreamSynchronize(e->stream);
  if (err == hipSuccess)
    err = hipMemcpy(position, dpos, sizeof(double) * 3 * n,
                    hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(direction, ddir, sizeof(double) * 3 * n,
                    hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(frequency, dnu, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(cross_sections, dsig, sizeof(double) * CMI_NION * n,
                    hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(tau, dtau, sizeof(double) * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(err);
  return CMI_GPU_OK;
}

int cmi_gpu_trace_packets(cmi_gpu_engine *e, uint64_t n,
                          const double *position, const double *direction,
                          const double *tau, const double *sigma_H,
                          const double *sigma_He_corr, int32_t max_steps,
                          int64_t *out_cell, double *out_ds,
                          int32_t *out_nsteps, int64_t *out_last_cell,
                          double *out_position) {
  if (!e || !position || !direction || !tau || !sigma_H || !sigma_He_corr ||
      !out_cell || !out_ds || !out_nsteps || !out_last_cell || !out_position ||
      max_steps <= 0)
    return fail(CMI_GPU_EINVAL, "trace_packets: bad argument");
  if (!e->have_cells)
    return fail(CMI_GPU_ESTATE, "trace_packets: no cell data");
  if (n == 0)
    return CMI_GPU_OK;
  HIP_TRY(hipSetDevice(e->device));
  char *buf = nullptr;
  const size_t in_bytes = sizeof(double) * (3 + 3 + 1 + 1 + 1) * n;
  const size_t cell_bytes = sizeof(int64_t) * (size_t)max_steps * n;
  const size_t ds_bytes = sizeof(double) * (size_t)max_steps * n;
  const size_t tail = (sizeof(int64_t) + 3 * sizeof(double)) * n +
                      sizeof(int32_t) * n;
  HIP_TRY(hipMalloc(&buf, in_bytes + cell_bytes + ds_bytes + tail));
  double *dpos = (double *)buf, *ddir = dpos + 3 * n, *dtau = ddir + 3 * n,
         *dsh = dtau + n, *dshe = dsh + n;
  int64_t *dcell = (int64_t *)(buf + in_bytes);
  double *dds = (double *)(buf + in_bytes + cell_bytes);
  int64_t *dlast = (int64_t *)(buf + in_bytes + cell_bytes + ds_bytes);
  double *dfinal = (double *)(dlast + n);
  int32_t *dnsteps = (int32_t *)(dfinal + 3 * n);
  hipError_t err = hipMemcpy(dpos, position, sizeof(double) * 3 * n,
                             hipMemcpyHostToDevice);
  if (err == hipSuccess)
    err = hipMemcpy(ddir, direction, sizeof(double) * 3 * n,
                    hipMemcpyHostToDevice);
  if (err == hipSuccess)
    err = hipMemcpy(dtau, tau, sizeof(double) * n, hipMemcpyHostToDevice);
  if (err == hipSuccess)
    err = hipMemcpy(dsh, sigma_H, sizeof(double) * n, hipMemcpyHostToDevice);
  if (err == hipSuccess)
    err = hipMemcpy(dshe, sigma_He_corr, sizeof(double) * n,
                    hipMemcpyHostToDevice);
  if (err == hipSuccess)
    err = hipMemsetAsync(dcell, 0xff, cell_bytes, e->stream);
  if (err == hipSuccess)
    err = hipMemsetAsync(dds, 0, ds_bytes, e->stream);
  if (err == hipSuccess) {
    if (e->tune.exact_dda || e->ncell >= CMI_FAST_MARCHER_MAX_CELLS)
      trace_probe_kernel<true><<<(unsigned)((n + 63) / 64), 64, 0, e->stream>>>(
          e->grid, e->opacity, n, dpos, ddir, dtau, dsh, dshe, max_steps,
          dcell, dds, dnsteps, dlast, dfinal);
    else
      trace_probe_kernel<false>
          <<<(unsigned)((n + 63) / 64), 64, 0, e->stream>>>(
              e->grid, e->opacity, n, dpos, ddir, dtau, dsh, dshe, max_steps,
              dcell, dds, dnsteps, dlast, dfinal);
    err = hipGetLastError();
  }
  if (err == hipSuccess)
    err = hipStreamSynchronize(e->stream);
  if (err == hipSuccess)
    err = hipMemcpy(out_cell, dcell, cell_bytes, hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(out_ds, dds, ds_bytes, hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(out_nsteps, dnsteps, sizeof(int32_t) * n,
                    hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(out_last_cell, dlast, sizeof(int64_t) * n,
                    hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(out_position, dfinal, sizeof(double) * 3 * n,
                    hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  HIP_TRY(err);
  return CMI_GPU_OK;
}

int cmi_gpu_sample_spectrum(cmi_gpu_engine *e, int32_t kind,
                            double temperature, uint32_t seed, uint64_t n,
                            double *frequencies) {
  if (!e || !frequencies || kind < 0 || kind > 3)
    return fail(CMI_GPU_EINVAL, "sample_spectrum: bad argument");
  if (!e->have_xsec)
    return fail(CMI_GPU_ESTATE, "sample_spectrum: cross sections not set");
  if (n == 0)
    return CMI_GPU_OK;
  HIP_TRY(hipSetDevice(e->device));
  /* force the tables even if no sampled spectrum is configured */
  const int32_t saved = e->model.reemit_type;
  e->model.reemit_type = CMI_GPU_REEMIT_PHYSICAL;
  int rc = ensure_spectra(e);
  e->model.reemit_type = saved;
  if (rc)
    return rc;
  if (kind == 0 && e->model.spectrum_type != CMI_GPU_SPECTRUM_PLANCK)
    return fail(CMI_GPU_ESTATE, "sample_spectrum: no Planck spectrum set");
  double *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(double) * n));
  spectrum_probe_kernel<<<(unsigned)((n + 255) / 256), 256, 0, e->stream>>>(
      e->model, kind, temperature, seed, n, d);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess)
    err = hipStreamSynchronize(e->stream);
  if (err == hipSuccess)
    err = hipMemcpy(frequencies, d, sizeof(double) * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(err);
  return CMI_GPU_OK;
}

int cmi_gpu_thermal_probe(cmi_gpu_engine *e, int64_t n, int32_t solve,
                          const double *J, const double *heating,
                          const double *temperature,
                          const double *number_density, double *out_fractions,
                          double *out_temperature, double *out_pair) {
  if (!e || n <= 0 || !J || !heating || !temperature || !number_density ||
      !out_fractions || !out_temperature || !out_pair)
    return fail(CMI_GPU_EINVAL, "thermal_probe: bad argument");
  if (!e->have_recomb)
    return fail(CMI_GPU_ESTATE, "thermal_probe: recombination rates not set");
  HIP_TRY(hipSetDevice(e->device));
  double *d = nullptr;
  const size_t in_count = (size_t)n * (CMI_NION + 2 + 1 + 1);
  const size_t out_count = (size_t)n * (CMI_NION + 1 + 2);
  HIP_TRY(hipMalloc(&d, sizeof(double) * (in_count + out_count)));
  double *dJ = d, *dh = dJ + CMI_NION * n, *dT = dh + 2 * n, *dn = dT + n,
         *dx = dn + n, *dTo = dx + CMI_NION * n, *dp = dTo + n;
  hipError_t err =
      hipMemcpy(dJ, J, sizeof(double) * CMI_NION * n, hipMemcpyHostToDevice);
  if (err == hipSuccess)
    err = hipMemcpy(dh, heating, sizeof(double) * 2 * n, hipMemcpyHostToDevice);
  if (err == hipSuccess)
    err = hipMemcpy(dT, temperature, sizeof(double) * n, hipMemcpyHostToDevice);
  if (err == hipSuccess)
    err = hipMemcpy(dn, number_density, sizeof(double) * n,
                    hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    thermal_probe_kernel<<<(unsigned)((n + 63) / 64), 64, 0, e->stream>>>(
        e->model, n, solve, dJ, dh, dT, dn, dx, dTo, dp);
    err = hipGetLastError();
  }
  if (err == hipSuccess)
    err = hipStreamSynchronize(e->stream);
  if (err == hipSuccess)
    err = hipMemcpy(out_fractions, dx, sizeof(double) * CMI_NION * n,
                    hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(out_temperature, dTo, sizeof(double) * n,
                    hipMemcpyDeviceToHost);
  if (err == hipSuccess)
    err = hipMemcpy(out_pair, dp, sizeof(double) * 2 * n,
                    hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(err);
  return CMI_GPU_OK;
}

int cmi_gpu_physics_probe(cmi_gpu_engine *e, int32_t kind, int64_t n,
                          const double *in, double *out) {
  static const int in_width[5] = {1, 1, 15, 1, 1};
  static const int out_width[5] = {CMI_NION, CMI_NION, 1, 5, 3 * CMI_NION};
  if (!e || n <= 0 || !in || !out || kind < 0 || kind > 4)
    return fail(CMI_GPU_EINVAL, "physics_probe: bad argument");
  if (kind == 0 && !e->have_xsec)
    return fail(CMI_GPU_ESTATE, "physics_probe: cross sections not set");
  if (kind == 1 && !e->have_recomb)
    return fail(CMI_GPU_ESTATE, "physics_probe: recombination rates not set");
  HIP_TRY(hipSetDevice(e->device));
  const size_t nin = (size_t)n * in_width[kind];
  const size_t nout = (size_t)n * out_width[kind];
  double *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(double) * (nin + nout)));
  hipError_t err =
      hipMemcpy(d, in, sizeof(double) * nin, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    physics_probe_kernel<<<(unsigned)((n + 63) / 64), 64, 0, e->stream>>>(
        e->model, kind, n, d, in_width[kind], d + nin, out_width[kind]);
    err = hipGetLastError();
  }
  if (err == hipSuccess)
    err = hipStreamSynchronize(e->stream);
  if (err == hipSuccess)
    err = hipMemcpy(out, d + nin, sizeof(double) * nout,
                    hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(err);
  return CMI_GPU_OK;
}

int cmi_gpu_get_timing(cmi_gpu_engine *e, int32_t reset, double *shoot_ms,
                       uint64_t *shoot_launches, double *update_ms,
                       uint64_t *update_launches) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  double s = 0., u = 0.;
  for (auto &p : e->shoot_events) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p.start, p.stop));
    s += ms;
  }
  for (auto &p : e->update_events) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p.start, p.stop));
    u += ms;
  }
  if (shoot_ms)
    *shoot_ms = s;
  if (shoot_launches)
    *shoot_launches = e->shoot_events.size();
  if (update_ms)
    *update_ms = u;
  if (update_launches)
    *update_launches = e->update_events.size();
  if (reset) {
    release_events(e, e->shoot_events);
    release_events(e, e->update_events);
    release_events(e, e->kernel_events);
  }
  return CMI_GPU_OK;
}

int cmi_gpu_get_launch_times(cmi_gpu_engine *e, uint64_t capacity,
                             double *ms, uint64_t *packets, uint64_t *count) {
  if (!e || !count)
    return fail(CMI_GPU_EINVAL, "get_launch_times: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  *count = e->kernel_events.size();
  for (uint64_t i = 0; i < e->kernel_events.size() && i < capacity; ++i) {
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, e->kernel_events[i].start,
                                e->kernel_events[i].stop));
    if (ms)
      ms[i] = t;
    if (packets)
      packets[i] = e->kernel_events[i].packets;
  }
  return CMI_GPU_OK;
}

int cmi_gpu_get_launch_steps(cmi_gpu_engine *e, uint64_t capacity,
                             uint64_t *steps, uint64_t *count) {
  if (!e || !count)
    return fail(CMI_GPU_EINVAL, "get_launch_steps: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  *count = e->kernel_events.size();
  const uint64_t n = *count < capacity ? *count : capacity;
  if (n && steps)
    HIP_TRY(hipMemcpy(steps, e->launch_steps, sizeof(uint64_t) * n,
                      hipMemcpyDeviceToHost));
  return CMI_GPU_OK;
}

int cmi_gpu_get_kernel_timing(cmi_gpu_engine *e, double *kernel_ms,
                              uint64_t *kernel_launches) {
  if (!e)
    return fail(CMI_GPU_EINVAL, "null engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  double k = 0.;
  for (auto &p : e->kernel_events) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p.start, p.stop));
    k += ms;
  }
  if (kernel_ms)
    *kernel_ms = k;
  if (kernel_launches)
    *kernel_launches = e->kernel_events.size();
  return CMI_GPU_OK;
}

} // extern "C"

#include "group.h"
