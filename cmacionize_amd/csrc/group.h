/*
 * group.h - several engines driven by ONE host process (included at the end
 * of engine.hip): the native multi-GPU layer of the C ABI.
 *
 *  replica mode   every engine holds the whole grid and flies its share of
 *                 the packets; cmi_gpu_group_reduce_accumulators sums the
 *                 accumulator blocks into every engine - the reference's
 *                 MPI_Allreduce of each accumulator field
 *                 (src/IonizationSimulation.cpp:459-528) as ONE grouped
 *                 ncclAllReduce over RCCL / xGMI.
 *  domain mode    every engine holds one block of the grid (one block per
 *                 GPU); cmi_gpu_group_exchange_flights hands the flights that
 *                 left a block to the engine that owns the cell they enter -
 *                 the photon buffers of the reference's task-based path
 *                 (src/PhotonTraversalTaskContext.hpp:100-278,
 *                 src/MemorySpace.hpp:96-127) - device to device: a routing
 *                 kernel on the SOURCE device writes every row straight into
 *                 the destination engine's inbox over xGMI (peer access);
 *                 nothing passes through host memory except n x n counts.
 *                 Engines of a group that hold the SAME block are copies of
 *                 each other (DensitySubGridCreator::create_copies,
 *                 src/DensitySubGridCreator.hpp:437-531): the packets emitted
 *                 in the block and the flights that enter it are dealt to its
 *                 copies by packet id, and cmi_gpu_group_reduce_accumulators
 *                 sums the copies' accumulators into every copy
 *                 (update_original_counters, :556-574) so that each then
 *                 solves the same cells from the same integrals
 *                 (update_copy_properties, :580-598, without the copy).
 *
 * RCCL is loaded at run time (dlopen) by the first reduce: a process that
 * already has another copy of RCCL (PyTorch ships its own) never maps a
 * second one through this library, and a host without RCCL can still use
 * everything else.
 */
#ifndef CMI_GROUP_H
#define CMI_GROUP_H

#include <dlfcn.h>
#include <stdlib.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <string>
#include <mutex>
#include <thread>

/* the handful of RCCL entry points used, by their rccl.h signatures */
namespace {
struct RcclApi {
  void *handle = nullptr;
  typedef struct ncclComm *comm_t;
  int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, comm_t,
                   hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, comm_t,
                   hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::mutex mutex;
  bool ready = false;
  /* (callable from several host threads at once: the classes of a group are
   * updated on a thread each) */
  bool load() {
    std::lock_guard<std::mutex> lock(mutex);
    if (ready)
      return true;
    if (!handle)
      for (const char *name : {"librccl.so.1", "librccl.so",
                               "/opt/rocm/lib/librccl.so.1"}) {
        handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (handle)
          break;
      }
    if (!handle)
      return false;
    if (!resolve())
      return false;
    ready = true; /* only once every symbol is there */
    return true;
  }
  bool resolve() {
#define CMI_RCCL_SYM(member, symbol)                                           \
  member = reinterpret_cast<decltype(member)>(dlsym(handle, symbol));          \
  if (!member)                                                                 \
    return false;
    CMI_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    CMI_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    CMI_RCCL_SYM(GroupStart, "ncclGroupStart")
    CMI_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    CMI_RCCL_SYM(AllReduce, "ncclAllReduce")
    CMI_RCCL_SYM(AllGather, "ncclAllGather")
    CMI_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef CMI_RCCL_SYM
    return true;
  }
};
RcclApi g_rccl;
constexpr int kNcclDouble = 8; /* ncclFloat64 */
constexpr int kNcclSum = 0;    /* ncclSum */
} // namespace

#define CMI_GROUP_MAX 64

/* block of the whole grid an engine owns */
struct GroupBoxDev {
  int32_t offset[3], size[3];
  int32_t copy_rank, copy_count;
};

struct GroupRouteArgs {
  const double *rows;        /* exports of the source engine */
  const unsigned int *count; /* how many */
  unsigned int capacity;
  int32_t n;                 /* engines */
  GroupBoxDev box[CMI_GROUP_MAX];
  int32_t global_ncell[3];
  uint32_t *dest;            /* [capacity] owner of each row */
  unsigned int *hist;        /* [n] rows per owner */
  /* second pass */
  double *inbox[CMI_GROUP_MAX];      /* destination buffers (peer memory) */
  unsigned int start[CMI_GROUP_MAX]; /* first slot of this source in each */
  unsigned int room[CMI_GROUP_MAX];  /* capacity of each inbox */
  unsigned int *cursor;              /* [n] rows placed per owner */
};

/* pass 1: who owns the cell each exported flight enters */
__global__ void __launch_bounds__(CMI_BLOCK)
    group_route_count_kernel(const GroupRouteArgs a) {
  __shared__ unsigned int s_hist[CMI_GROUP_MAX];
  if (threadIdx.x < CMI_GROUP_MAX)
    s_hist[threadIdx.x] = 0;
  __syncthreads();
  const unsigned int n = *a.count < a.capacity ? *a.count : a.capacity;
  const unsigned int stride = gridDim.x * blockDim.x;
  for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const int64_t cell =
        __double_as_longlong(a.rows[(size_t)CMI_FLIGHT_DOUBLES * i + 12]);
    const int32_t gz = (int32_t)(cell % a.global_ncell[2]);
    const int32_t gy =
        (int32_t)((cell / a.global_ncell[2]) % a.global_ncell[1]);
    const int32_t gx =
        (int32_t)(cell / ((int64_t)a.global_ncell[2] * a.global_ncell[1]));
    /* (the low half of column 13 is the packet id: which copy of the block) */
    const uint32_t packet_id = (uint32_t)__double_as_longlong(
        a.rows[(size_t)CMI_FLIGHT_DOUBLES * i + 13]);
    uint32_t owner = 0;
    for (int k = 0; k < a.n; ++k) {
      const GroupBoxDev &b = a.box[k];
      if (gx >= b.offset[0] && gx < b.offset[0] + b.size[0] &&
          gy >= b.offset[1] && gy < b.offset[1] + b.size[1] &&
          gz >= b.offset[2] && gz < b.offset[2] + b.size[2] &&
          (int32_t)(packet_id % (uint32_t)b.copy_count) == b.copy_rank)
        owner = (uint32_t)k;
    }
    a.dest[i] = owner;
    atomicAdd(&s_hist[owner], 1u);
  }
  __syncthreads();
  if (threadIdx.x < (unsigned)a.n && s_hist[threadIdx.x] != 0)
    atomicAdd(&a.hist[threadIdx.x], s_hist[threadIdx.x]);
}

/* pass 2: every row into the inbox of its owner (8 lanes per row: a row is
 * 8 x 16 B), written across xGMI where the owner is another device */
__global__ void __launch_bounds__(CMI_BLOCK)
    group_route_scatter_kernel(const GroupRouteArgs a) {
  const unsigned int n = *a.count < a.capacity ? *a.count : a.capacity;
  const unsigned int stride = (gridDim.x * blockDim.x) >> 3;
  const int part = threadIdx.x & 7;
  for (unsigned int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n;
       i += stride) {
    const uint32_t owner = a.dest[i];
    unsigned int slot = 0;
    if (part == 0)
      slot = a.start[owner] + atomicAdd(&a.cursor[owner], 1u);
    slot = __shfl(slot, (threadIdx.x & 63) & ~7, 64);
    if (slot < a.room[owner])
      reinterpret_cast<double2 *>(a.inbox[owner] +
                                  (size_t)CMI_FLIGHT_DOUBLES * slot)[part] =
          reinterpret_cast<const double2 *>(
              a.rows + (size_t)CMI_FLIGHT_DOUBLES * i)[part];
  }
}

/* out[i] += in[i] (engines of a group that share a device: no collective) */
__global__ void __launch_bounds__(CMI_BLOCK)
    group_add_kernel(double *out, const double *in, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    out[i] += in[i];
}

/* engines of a group that hold the same cells: the replicas of replica
 * mode, the copies of a block in domain mode */
struct GroupClass {
  std::vector<int> member;
  bool distinct_devices = true, same_device = true;
  std::vector<RcclApi::comm_t> comm;
};

/* dst[f][i] = src[f][i] for the state fields a cell update writes
 * (temperature and the 14 ionic fractions: fields 1..15 of the state block)
 * and the transport records, cells [first, first + count): an engine pulls
 * the slab another engine of its class has solved (peer memory) */
__global__ void __launch_bounds__(CMI_BLOCK)
    group_pull_slab_kernel(double *dst_state, double2 *dst_opacity,
                           const double *src_state, const double2 *src_opacity,
                           int64_t ncell, int64_t first, int64_t count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count;
       k += stride) {
    const int64_t i = first + k;
#pragma unroll
    for (int f = 1; f < 16; ++f)
      dst_state[f * ncell + i] = src_state[f * ncell + i];
    dst_opacity[i] = src_opacity[i];
  }
}

/* One host thread per engine, started once and parked between jobs: the
 * calls that continue flights (with re-emission cmi_gpu_shoot_flights reads a
 * few bytes back per generation and blocks its caller) and the sharded cell
 * update run on them, so that the devices of a group work at the same time.
 * Round 3 started and joined fresh threads in every exchange round (100-240
 * us of host time per round, profiles/r04/group_rounds.txt). */
class GroupWorkers {
  std::vector<std::thread> _threads;
  std::mutex _mutex;
  std::condition_variable _work, _done;
  const std::function<void(int)> *_job = nullptr;
  std::vector<char> _wanted;
  uint64_t _generation = 0;
  int _pending = 0;
  bool _stop = false;

  void loop(int k) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)> *job = nullptr;
      {
        std::unique_lock<std::mutex> lock(_mutex);
        _work.wait(lock, [&] { return _stop || _generation != seen; });
        if (_stop)
          return;
        seen = _generation;
        if (_wanted[k])
          job = _job;
      }
      if (!job)
        continue;
      (*job)(k);
      {
        std::lock_guard<std::mutex> lock(_mutex);
        if (--_pending == 0)
          _done.notify_all();
      }
    }
  }

public:
  ~GroupWorkers() {
    {
      std::lock_guard<std::mutex> lock(_mutex);
      _stop = true;
    }
    _work.notify_all();
    for (std::thread &t : _threads)
      t.join();
  }
  /* job(k) for every k with wanted[k], each on worker k; returns when all
   * are done. (Called from one thread at a time.) */
  void run(const std::vector<char> &wanted,
           const std::function<void(int)> &job) {
    const int n = (int)wanted.size();
    int count = 0, only = -1;
    for (int k = 0; k < n; ++k)
      if (wanted[k]) {
        ++count;
        only = k;
      }
    if (count == 0)
      return;
    if (count == 1) {
      job(only); /* nothing to overlap with */
      return;
    }
    if ((int)_threads.size() < n) {
      {
        std::lock_guard<std::mutex> lock(_mutex);
        _wanted.resize(n, 0);
      }
      while ((int)_threads.size() < n) {
        const int k = (int)_threads.size();
        _threads.emplace_back([this, k] { loop(k); });
      }
    }
    {
      std::lock_guard<std::mutex> lock(_mutex);
      _wanted.assign(wanted.begin(), wanted.end());
      _wanted.resize(_threads.size(), 0);
      _job = &job;
      _pending = count;
      ++_generation;
    }
    _work.notify_all();
    std::unique_lock<std::mutex> lock(_mutex);
    _done.wait(lock, [&] { return _pending == 0; });
  }
};

struct cmi_gpu_group {
  int n = 0;
  cmi_gpu_engine *engine[CMI_GROUP_MAX];
  std::vector<GroupClass> classes;
  /* domain mode */
  double *inbox[CMI_GROUP_MAX];
  uint64_t inbox_capacity[CMI_GROUP_MAX];
  uint32_t *dest[CMI_GROUP_MAX];
  unsigned int *hist[CMI_GROUP_MAX]; /* [2 n]: rows per owner, cursors */
  hipEvent_t routed[CMI_GROUP_MAX];
  hipEvent_t solved[CMI_GROUP_MAX];
  uint64_t rounds = 0, flights = 0;
  /* host-side cost of the exchange rounds (cmi_gpu_group_exchange_stats):
   * microseconds until the n x n counts are known on the host, and spent
   * starting / joining the owners' host threads beyond the longest flight
   * call, summed over the rounds that moved flights */
  double stats_counts_us = 0., stats_threads_us = 0., stats_total_us = 0.;
  uint64_t stats_rounds = 0;
  GroupWorkers workers;
  /* pinned host memory for the rounds' counts: per source the rows it
   * exported and how many of them go to each engine ([n][n + 1]) */
  unsigned int *host_counts = nullptr;
};

/* fn(k), k < n, on one host thread each: the engines' cell updates block
 * their callers (the temperature pipeline reads a count back per secant
 * step), and engines on different devices should not wait for each other.
 * The first error becomes the caller's. */
template <typename F> static int group_in_parallel(int n, F fn) {
  if (n == 1)
    return fn(0);
  std::vector<int> rc((size_t)n, 0);
  std::vector<std::string> message((size_t)n);
  std::vector<std::thread> threads;
  for (int k = 0; k < n; ++k)
    threads.emplace_back([&, k]() {
      rc[k] = fn(k);
      if (rc[k])
        message[k] = cmi_gpu_last_error(); /* thread-local */
    });
  for (std::thread &t : threads)
    t.join();
  for (int k = 0; k < n; ++k)
    if (rc[k])
      return fail(rc[k], "%s", message[k].c_str());
  return CMI_GPU_OK;
}

extern "C" {

int cmi_gpu_group_create(int32_t n, cmi_gpu_engine *const *engines,
                         cmi_gpu_group **out) {
  if (n < 1 || n > CMI_GROUP_MAX || !engines || !out)
    return fail(CMI_GPU_EINVAL, "group_create: 1 to %d engines", CMI_GROUP_MAX);
  cmi_gpu_group *g = new cmi_gpu_group();
  g->n = n;
  for (int i = 0; i < n; ++i) {
    if (!engines[i]) {
      delete g;
      return fail(CMI_GPU_EINVAL, "group_create: null engine");
    }
    g->engine[i] = engines[i];
    g->inbox[i] = nullptr;
    g->inbox_capacity[i] = 0;
    g->dest[i] = nullptr;
    g->hist[i] = nullptr;
    g->routed[i] = nullptr;
    g->solved[i] = nullptr;
    /* the class of engines that hold the same cells */
    const GridDev &gi = engines[i]->grid;
    GroupClass *cls = nullptr;
    for (GroupClass &c : g->classes) {
      const cmi_gpu_engine *first = engines[c.member[0]];
      const GridDev &gf = first->grid;
      bool same = gf.decomposed == gi.decomposed &&
                  first->cells.acc_cell_stride ==
                      engines[i]->cells.acc_cell_stride &&
                  first->config.track_heating ==
                      engines[i]->config.track_heating;
      for (int a = 0; a < 3; ++a)
        same &= gf.ncell[a] == gi.ncell[a] && gf.offset[a] == gi.offset[a] &&
                gf.global_ncell[a] == gi.global_ncell[a];
      if (same)
        cls = &c;
    }
    if (!cls) {
      g->classes.emplace_back();
      cls = &g->classes.back();
    }
    for (int k : cls->member) {
      if (engines[k]->device == engines[i]->device)
        cls->distinct_devices = false;
      else
        cls->same_device = false;
    }
    cls->member.push_back(i);
  }
  for (GroupClass &c : g->classes) {
    if (c.member.size() > 1 && !c.distinct_devices && !c.same_device) {
      delete g;
      return fail(CMI_GPU_EINVAL,
                  "group_create: the engines that hold the same cells must "
                  "sit on different devices (or, for tests, all on one)");
    }
    /* copies of a block share its packets by packet id */
    for (size_t r = 0; r < c.member.size(); ++r) {
      GridDev &grid = engines[c.member[r]]->grid;
      grid.copy_rank = (int32_t)r;
      grid.copy_count = (int32_t)c.member.size();
    }
  }
  /* peer access between every pair of devices (flights are written straight
   * into the owner's inbox) */
  for (int i = 0; i < n; ++i) {
    hipError_t err = hipSetDevice(g->engine[i]->device);
    for (int k = 0; k < n && err == hipSuccess; ++k) {
      const int other = g->engine[k]->device;
      if (other == g->engine[i]->device)
        continue;
      int can = 0;
      err = hipDeviceCanAccessPeer(&can, g->engine[i]->device, other);
      if (err != hipSuccess)
        break;
      if (!can) {
        delete g;
        return fail(CMI_GPU_EDEVICE,
                    "group_create: device %d cannot access device %d",
                    g->engine[i]->device, other);
      }
      err = hipDeviceEnablePeerAccess(other, 0);
      if (err == hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
        err = hipSuccess;
      }
    }
    if (err == hipSuccess)
      err = hipEventCreateWithFlags(&g->routed[i], hipEventDisableTiming);
    if (err == hipSuccess)
      err = hipEventCreateWithFlags(&g->solved[i], hipEventDisableTiming);
    if (err != hipSuccess) {
      delete g;
      HIP_TRY(err);
    }
  }
  *out = g;
  return CMI_GPU_OK;
}

int cmi_gpu_group_destroy(cmi_gpu_group *g) {
  if (!g)
    return CMI_GPU_OK;
  for (int i = 0; i < g->n; ++i) {
    (void)hipSetDevice(g->engine[i]->device);
    (void)hipStreamSynchronize(g->engine[i]->stream);
    (void)hipFree(g->inbox[i]);
    (void)hipFree(g->dest[i]);
    (void)hipFree(g->hist[i]);
    if (g->routed[i])
      (void)hipEventDestroy(g->routed[i]);
    if (g->solved[i])
      (void)hipEventDestroy(g->solved[i]);
    g->engine[i]->grid.copy_rank = 0;
    g->engine[i]->grid.copy_count = 1;
  }
  for (GroupClass &c : g->classes)
    for (RcclApi::comm_t comm : c.comm)
      (void)g_rccl.CommDestroy(comm);
  if (g->host_counts)
    (void)hipHostFree(g->host_counts);
  delete g;
  return CMI_GPU_OK;
}

/* the part of an engine's accumulator block a transport step can have
 * written: {pointer, doubles} pieces */
static int active_accumulators(cmi_gpu_engine *e, double *ptr[2],
                               int64_t count[2]) {
  if (e->cells.acc_cell_stride != 1) { /* [ncell][16]: everything */
    ptr[0] = e->acc_block;
    count[0] = (int64_t)CMI_NACC * e->ncell;
    return 1;
  }
  ptr[0] = e->acc_block; /* hydrogen-only: J_H (+ the heating terms) */
  count[0] = e->ncell;
  if (!e->config.track_heating)
    return 1;
  ptr[1] = e->acc_block + (int64_t)CMI_NION * e->ncell;
  count[1] = 2 * e->ncell;
  return 2;
}

/* the RCCL communicator of a class (one rank per member), made on first use */
static int class_comm(cmi_gpu_group *g, GroupClass &c) {
  /* one communicator is made at a time (ncclCommInitAll of two classes from
   * two host threads at once is not something RCCL promises) */
  static std::mutex comm_mutex;
  std::lock_guard<std::mutex> lock(comm_mutex);
  if (!c.comm.empty())
    return CMI_GPU_OK;
  const int n = (int)c.member.size();
  if (!g_rccl.load())
    return fail(CMI_GPU_EDEVICE, "cannot load RCCL (librccl.so.1): %s",
                dlerror());
  int devices[CMI_GROUP_MAX];
  for (int i = 0; i < n; ++i)
    devices[i] = g->engine[c.member[i]]->device;
  c.comm.resize(n);
  const int rc = g_rccl.CommInitAll(c.comm.data(), n, devices);
  if (rc != 0) {
    c.comm.clear();
    return fail(CMI_GPU_EDEVICE, "ncclCommInitAll failed: %s",
                g_rccl.GetErrorString(rc));
  }
  return CMI_GPU_OK;
}

/* sum the accumulators of one class of engines into every member */
static int reduce_class(cmi_gpu_group *g, GroupClass &c) {
  const int n = (int)c.member.size();
  double *ptr[CMI_GROUP_MAX][2];
  int64_t count[2] = {0, 0};
  int pieces = 0;
  for (int i = 0; i < n; ++i)
    pieces = active_accumulators(g->engine[c.member[i]], ptr[i], count);
  if (n > 1 && c.same_device) {
    /* engines that share a device (tests on one GPU): plain sums, in engine
     * order, then copies back */
    cmi_gpu_engine *e0 = g->engine[c.member[0]];
    HIP_TRY(hipSetDevice(e0->device));
    for (int i = 1; i < n; ++i)
      HIP_TRY(hipStreamSynchronize(g->engine[c.member[i]]->stream));
    for (int p = 0; p < pieces; ++p) {
      for (int i = 1; i < n; ++i) {
        group_add_kernel<<<grid_blocks(e0, count[p], 8), CMI_BLOCK, 0,
                           e0->stream>>>(ptr[0][p], ptr[i][p], count[p]);
        HIP_TRY(hipGetLastError());
      }
      for (int i = 1; i < n; ++i)
        HIP_TRY(hipMemcpyAsync(ptr[i][p], ptr[0][p],
                               sizeof(double) * count[p],
                               hipMemcpyDeviceToDevice, e0->stream));
    }
    HIP_TRY(hipStreamSynchronize(e0->stream));
    return CMI_GPU_OK;
  }
  {
    const int rc = class_comm(g, c);
    if (rc)
      return rc;
  }
  /* one grouped all-reduce per piece: every engine's call is enqueued on its
   * own stream, behind its transport kernels */
  for (int p = 0; p < pieces; ++p) {
    /* (no return between GroupStart and GroupEnd: an open group would queue
     * this thread's later collectives for ever) */
    int rc = g_rccl.GroupStart();
    hipError_t herr = hipSuccess;
    for (int i = 0; i < n && rc == 0 && herr == hipSuccess; ++i) {
      cmi_gpu_engine *e = g->engine[c.member[i]];
      herr = hipSetDevice(e->device);
      if (herr == hipSuccess)
        rc = g_rccl.AllReduce(ptr[i][p], ptr[i][p], (size_t)count[p],
                              kNcclDouble, kNcclSum, c.comm[i], e->stream);
    }
    const int rc_end = g_rccl.GroupEnd();
    HIP_TRY(herr);
    if (rc == 0)
      rc = rc_end;
    if (rc != 0)
      return fail(CMI_GPU_EDEVICE, "ncclAllReduce failed: %s",
                  g_rccl.GetErrorString(rc));
  }
  return CMI_GPU_OK;
}

int cmi_gpu_group_reduce_accumulators(cmi_gpu_group *g) {
  if (!g)
    return fail(CMI_GPU_EINVAL, "null group");
  /* (CMI_GPU_FORCE_RCCL: run the collective even for a class of one - the
   * only way to exercise the RCCL path on a single-GPU box) */
  const bool force = getenv("CMI_GPU_FORCE_RCCL") != nullptr;
  for (GroupClass &c : g->classes) {
    if (c.member.size() == 1 && !force)
      continue;
    const int rc = reduce_class(g, c);
    if (rc)
      return rc;
  }
  return CMI_GPU_OK;
}

/* the cell update of one class: member r solves the r-th slab of the cells
 * (MPICommunicator::distribute, src/MPICommunicator.hpp:207-222), then every
 * member gets the slabs of the others */
static int update_class(cmi_gpu_group *g, GroupClass &c, uint32_t loop,
                        double totweight) {
  const int n = (int)c.member.size();
  cmi_gpu_engine *e0 = g->engine[c.member[0]];
  const bool force = getenv("CMI_GPU_FORCE_RCCL") != nullptr;
  if (n == 1 && !force)
    return cmi_gpu_update_cells(e0, loop, totweight);
  const int64_t ncell = e0->ncell;
  std::vector<int64_t> first(n + 1, 0);
  for (int r = 0; r < n; ++r)
    first[r + 1] = first[r] + ncell / n + (r < ncell % n ? 1 : 0);
  {
    const int rc = group_in_parallel(n, [&](int r) -> int {
      cmi_gpu_engine *e = g->engine[c.member[r]];
      /* (a fresh host thread has device 0 current, and update_cells_range
       * returns before its own hipSetDevice for an empty slab) */
      HIP_TRY(hipSetDevice(e->device));
      const int urc = cmi_gpu_update_cells_range(e, loop, totweight, first[r],
                                                 first[r + 1] - first[r]);
      if (urc)
        return urc;
      HIP_TRY(hipEventRecord(g->solved[c.member[r]], e->stream));
      return CMI_GPU_OK;
    });
    if (rc)
      return rc;
  }
  if ((c.distinct_devices || force) && ncell % n == 0) {
    /* MPICommunicator::gather of the temperature and the ionic fractions
     * (src/IonizationSimulation.cpp:540-618) as grouped in-place
     * ncclAllGathers: slab r of a field sits at its place in every engine */
    const int rc0 = class_comm(g, c);
    if (rc0)
      return rc0;
    const int64_t count = ncell / n;
    int rc = g_rccl.GroupStart();
    hipError_t herr = hipSuccess;
    for (int r = 0; r < n && rc == 0 && herr == hipSuccess; ++r) {
      cmi_gpu_engine *e = g->engine[c.member[r]];
      herr = hipSetDevice(e->device);
      if (herr != hipSuccess)
        break;
      for (int f = 1; f < 16 && rc == 0; ++f) {
        double *field = e->state_block + (int64_t)f * ncell;
        rc = g_rccl.AllGather(field + r * count, field, (size_t)count,
                              kNcclDouble, c.comm[r], e->stream);
      }
      if (rc == 0) {
        double *records = reinterpret_cast<double *>(e->opacity);
        rc = g_rccl.AllGather(records + 2 * r * count, records,
                              (size_t)(2 * count), kNcclDouble, c.comm[r],
                              e->stream);
      }
    }
    const int rc_end = g_rccl.GroupEnd();
    HIP_TRY(herr);
    if (rc == 0)
      rc = rc_end;
    if (rc != 0)
      return fail(CMI_GPU_EDEVICE, "ncclAllGather failed: %s",
                  g_rccl.GetErrorString(rc));
    return CMI_GPU_OK;
  }
  /* engines that share a device, or slabs of unequal size: every engine
   * pulls the others' slabs itself (peer reads over xGMI) */
  for (int d = 0; d < n; ++d) {
    cmi_gpu_engine *e = g->engine[c.member[d]];
    HIP_TRY(hipSetDevice(e->device));
    for (int r = 0; r < n; ++r) {
      if (r == d)
        continue;
      cmi_gpu_engine *src = g->engine[c.member[r]];
      const int64_t count = first[r + 1] - first[r];
      if (count == 0)
        continue;
      HIP_TRY(hipStreamWaitEvent(e->stream, g->solved[c.member[r]], 0));
      group_pull_slab_kernel<<<grid_blocks(e, count, 8), CMI_BLOCK, 0,
                               e->stream>>>(e->state_block, e->opacity,
                                            src->state_block, src->opacity,
                                            ncell, first[r], count);
      HIP_TRY(hipGetLastError());
    }
  }
  /* nobody's next transport may overwrite what a peer still reads */
  for (int d = 0; d < n; ++d) {
    cmi_gpu_engine *e = g->engine[c.member[d]];
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return CMI_GPU_OK;
}

int cmi_gpu_group_update_cells(cmi_gpu_group *g, uint32_t loop,
                               double totweight) {
  if (!g)
    return fail(CMI_GPU_EINVAL, "null group");
  return group_in_parallel((int)g->classes.size(), [&](int k) -> int {
    return update_class(g, g->classes[(size_t)k], loop, totweight);
  });
}

int cmi_gpu_group_exchange_flights(cmi_gpu_group *g, uint32_t seed,
                                   uint32_t iteration, uint64_t first_packet,
                                   uint64_t *total_flights) {
  if (!g || !total_flights)
    return fail(CMI_GPU_EINVAL, "exchange_flights: bad argument");
  *total_flights = 0;
  const int n = g->n;
  typedef std::chrono::steady_clock clock;
  auto microseconds = [](clock::time_point from, clock::time_point to) {
    return std::chrono::duration<double, std::micro>(to - from).count();
  };
  const clock::time_point t_begin = clock::now();
  GroupRouteArgs proto;
  memset(&proto, 0, sizeof proto);
  proto.n = n;
  for (int i = 0; i < n; ++i) {
    cmi_gpu_engine *e = g->engine[i];
    if (!e->grid.decomposed || !e->export_rows)
      return fail(CMI_GPU_ESTATE,
                  "exchange_flights: every engine must be a block of a "
                  "decomposed grid with an export buffer");
    for (int a = 0; a < 3; ++a) {
      proto.box[i].offset[a] = e->grid.offset[a];
      proto.box[i].size[a] = e->grid.ncell[a];
      proto.global_ncell[a] = e->grid.global_ncell[a];
    }
    proto.box[i].copy_rank = e->grid.copy_rank;
    proto.box[i].copy_count = e->grid.copy_count;
  }
  /* pass 1 on every source: owners and counts */
  for (int s = 0; s < n; ++s) {
    cmi_gpu_engine *e = g->engine[s];
    HIP_TRY(hipSetDevice(e->device));
    if (!g->dest[s]) {
      HIP_TRY(hipMalloc(&g->dest[s], sizeof(uint32_t) * e->export_capacity));
      HIP_TRY(hipMalloc(&g->hist[s], sizeof(unsigned int) * 2 * CMI_GROUP_MAX));
    }
    HIP_TRY(hipMemsetAsync(g->hist[s], 0,
                           sizeof(unsigned int) * 2 * CMI_GROUP_MAX,
                           e->stream));
    GroupRouteArgs a = proto;
    a.rows = e->export_rows;
    a.count = e->export_count;
    a.capacity = (unsigned int)e->export_capacity;
    a.dest = g->dest[s];
    a.hist = g->hist[s];
    group_route_count_kernel<<<e->num_cu * 4, CMI_BLOCK, 0, e->stream>>>(a);
    HIP_TRY(hipGetLastError());
  }
  /* the n x n counts (and the overflow check of every export buffer): into
   * pinned host memory, all copies enqueued before the first wait */
  /* (portable: the engines of a group sit on different devices and every
   * one of them copies into this buffer on its own stream) */
  if (!g->host_counts)
    HIP_TRY(hipHostMalloc(&g->host_counts,
                          sizeof(unsigned int) * CMI_GROUP_MAX *
                              (CMI_GROUP_MAX + 1),
                          hipHostMallocPortable));
  for (int s = 0; s < n; ++s) {
    cmi_gpu_engine *e = g->engine[s];
    HIP_TRY(hipSetDevice(e->device));
    unsigned int *row = g->host_counts + (size_t)s * (n + 1);
    HIP_TRY(hipMemcpyAsync(row, g->hist[s], sizeof(unsigned int) * n,
                           hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(row + n, e->export_count, sizeof(unsigned int),
                           hipMemcpyDeviceToHost, e->stream));
  }
  std::vector<unsigned int> counts((size_t)n * n);
  std::vector<uint64_t> incoming(n, 0);
  for (int s = 0; s < n; ++s) {
    cmi_gpu_engine *e = g->engine[s];
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const unsigned int *row = g->host_counts + (size_t)s * (n + 1);
    const unsigned int exported = row[n];
    if (exported > e->export_capacity)
      return fail(CMI_GPU_ENOMEM,
                  "export buffer overflow: %u flights left a block, room for "
                  "%llu - flights were lost, the iteration is invalid",
                  exported, (unsigned long long)e->export_capacity);
    for (int d = 0; d < n; ++d) {
      counts[(size_t)s * n + d] = row[d];
      incoming[d] += row[d];
    }
  }
  uint64_t total = 0;
  for (int d = 0; d < n; ++d)
    total += incoming[d];
  *total_flights = total;
  if (total == 0)
    return CMI_GPU_OK;
  const clock::time_point t_counts = clock::now();
  /* inboxes */
  for (int d = 0; d < n; ++d) {
    if (g->inbox_capacity[d] >= incoming[d])
      continue;
    cmi_gpu_engine *e = g->engine[d];
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    (void)hipFree(g->inbox[d]);
    g->inbox[d] = nullptr;
    const uint64_t cap = incoming[d] + incoming[d] / 4 + 1024;
    HIP_TRY(hipMalloc(&g->inbox[d],
                      sizeof(double) * CMI_FLIGHT_DOUBLES * cap));
    g->inbox_capacity[d] = cap;
  }
  /* pass 2 on every source: rows into the owners' inboxes */
  std::vector<unsigned int> placed(n, 0);
  for (int s = 0; s < n; ++s) {
    cmi_gpu_engine *e = g->engine[s];
    HIP_TRY(hipSetDevice(e->device));
    GroupRouteArgs a = proto;
    a.rows = e->export_rows;
    a.count = e->export_count;
    a.capacity = (unsigned int)e->export_capacity;
    a.dest = g->dest[s];
    a.hist = g->hist[s];
    a.cursor = g->hist[s] + CMI_GROUP_MAX;
    for (int d = 0; d < n; ++d) {
      a.inbox[d] = g->inbox[d];
      a.start[d] = placed[d];
      a.room[d] = (unsigned int)g->inbox_capacity[d];
      placed[d] += counts[(size_t)s * n + d];
    }
    group_route_scatter_kernel<<<e->num_cu * 4, CMI_BLOCK, 0, e->stream>>>(a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(e->export_count, 0, sizeof(unsigned int),
                           e->stream));
    HIP_TRY(hipEventRecord(g->routed[s], e->stream));
  }
  /* every owner waits for all sources, then flies what it received. With
   * re-emission cmi_gpu_shoot_flights reads a few bytes back per generation
   * and blocks its caller: one host thread per owner, so that the devices
   * work at the same time (the error string of the C ABI is per thread: a
   * failure is carried back to the caller's) */
  for (int d = 0; d < n; ++d) {
    cmi_gpu_engine *e = g->engine[d];
    HIP_TRY(hipSetDevice(e->device));
    for (int s = 0; s < n; ++s)
      if (s != d)
        HIP_TRY(hipStreamWaitEvent(e->stream, g->routed[s], 0));
  }
  std::vector<int> rcs(n, CMI_GPU_OK);
  std::vector<std::string> messages(n);
  std::vector<double> fly_us(n, 0.);
  auto fly = [&](int d) {
    const clock::time_point t0 = clock::now();
    rcs[d] = cmi_gpu_shoot_flights(g->engine[d], seed, iteration, first_packet,
                                   g->inbox[d], incoming[d]);
    if (rcs[d])
      messages[d] = cmi_gpu_last_error();
    fly_us[d] = microseconds(t0, clock::now());
  };
  const clock::time_point t_fly = clock::now();
  std::vector<char> wanted(n, 0);
  for (int d = 0; d < n; ++d)
    wanted[d] = incoming[d] != 0;
  const std::function<void(int)> job = fly;
  g->workers.run(wanted, job);
  const clock::time_point t_end = clock::now();
  for (int d = 0; d < n; ++d)
    if (rcs[d])
      return fail(rcs[d], "%s", messages[d].c_str());
  ++g->rounds;
  g->flights += total;
  double longest = 0.;
  for (int d = 0; d < n; ++d)
    longest = fly_us[d] > longest ? fly_us[d] : longest;
  g->stats_counts_us += microseconds(t_begin, t_counts);
  g->stats_threads_us += microseconds(t_fly, t_end) - longest;
  g->stats_total_us += microseconds(t_begin, t_end);
  ++g->stats_rounds;
  return CMI_GPU_OK;
}

int cmi_gpu_group_exchange_stats(cmi_gpu_group *g, uint64_t *rounds,
                                 double *microseconds, int32_t reset) {
  if (!g || !rounds || !microseconds)
    return fail(CMI_GPU_EINVAL, "exchange_stats: bad argument");
  *rounds = g->stats_rounds;
  microseconds[0] = g->stats_counts_us;
  microseconds[1] = g->stats_threads_us;
  microseconds[2] = g->stats_total_us;
  if (reset) {
    g->stats_rounds = 0;
    g->stats_counts_us = g->stats_threads_us = g->stats_total_us = 0.;
  }
  return CMI_GPU_OK;
}

} // extern "C"

#endif
