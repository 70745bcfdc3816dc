"""Host-side mirror of the reference's iteration loop
(IonizationSimulation::run, src/IonizationSimulation.cpp:359-643) on top of
the C ABI, including the replicated-grid multi-process mode of the reference's
MPI path (src/IonizationSimulation.cpp:394-397,410-414,458-529): every rank
holds the whole grid, shoots its share of the packets, the accumulators are
sum-reduced, every rank then updates all cells.

The compute object ("backend") only needs the methods reset_grid / shoot /
get_counters / update_cells and an `accumulators` tensor; on the GPU it is
GpuBackend below (the HIP engine), in the CPU tests of the distributed logic
it is an oracle-backed stand-in defined under tests/.
"""
import numpy as np

PC = 3.086e16  # src/UnitConverter.hpp:110
ELECTRONVOLT = 1.6021766208e-19
PLANCK = 6.626070040e-34


def eV_to_Hz(ev):
    """UnitConverter::to_SI<QUANTITY_FREQUENCY>(x, "eV")"""
    return ev * ELECTRONVOLT * (1. / PLANCK) / 1.


# benchmarks/stromgren.param
STROMGREN = dict(
    anchor=(-5. * PC,) * 3, sides=(10. * PC,) * 3, periodic=(0, 0, 0),
    density=100. * 1.e6, temperature=8000., xH=1.e-6, xHe=1.e-6,
    source_position=[[0., 0., 0.]], source_weight=[1.], luminosity=4.26e49,
    frequency=eV_to_Hz(13.6), sigma_H=6.3e-18 * 1.e-4,
    alpha_H=4.e-13 * 1.e-6)


def _all_gather_blocks(dist, full, first, count, world):
    """MPICommunicator::gather (src/MPICommunicator.hpp:647-720): every rank
    contributes cells [first, first + count) of `full` (its block of
    distribute_packets(ncell, rank, world)), all end up with the whole array.
    Blocks differ by at most one cell; they travel padded to equal length."""
    import torch
    n = full.numel()
    longest = n // world + (1 if n % world else 0)
    part = torch.zeros(longest, dtype=full.dtype, device=full.device)
    part[:count] = full[first:first + count]
    parts = [torch.empty_like(part) for _ in range(world)]
    dist.all_gather(parts, part)
    for r in range(world):
        f, c = distribute_packets(n, r, world)
        full[f:f + c] = parts[r][:c]


def distribute_packets(n_packets, rank, world):
    """[first, first + count) of rank `rank`: MPICommunicator::distribute /
    distribute_block (src/MPICommunicator.hpp:197-239)."""
    q, r = divmod(int(n_packets), int(world))
    first = rank * q + min(rank, r)
    count = q + (1 if rank < r else 0)
    return first, count


class GpuBackend:
    """The HIP engine with torch-owned accumulators (so that torch.distributed
    can reduce them in place). The engine enqueues on torch's current stream,
    or on a stream of its own when that is the default stream; the drivers
    below put a host synchronisation between engine work and torch work
    (get_counters / get_export_count wait for the engine, .cpu() / .item() /
    torch.cuda.synchronize() for torch and RCCL)."""

    def __init__(self, ncell, anchor, sides, periodic=(0, 0, 0), device=0,
                 track_heating=False):
        import torch
        from .engine import GpuEngine, NACC
        self.torch = torch
        torch.cuda.set_device(device)
        n = int(np.prod(ncell))
        self.ncell = n
        self.track_heating = bool(track_heating)
        self.accumulators = torch.zeros(NACC * n, dtype=torch.float64,
                                        device="cuda:%d" % device)
        stream = torch.cuda.current_stream().cuda_stream
        self.engine = GpuEngine(
            ncell, anchor, sides, periodic, device=device,
            track_heating=track_heating, stream=stream,
            external_accumulators=self.accumulators.data_ptr())

    def active_accumulators(self):
        """The parts of the accumulator block a transport step can have
        written: with hydrogen-only transport (SoA block [16][ncell]) that is
        J_H and, if tracked, the two heating terms; the other 13 fields stay
        zero on every rank and need no reduction. With all ions transported
        (AoS [ncell][16]) it is the whole block."""
        _, cell_stride = self.engine.accumulator_layout()
        n = self.ncell
        if cell_stride != 1:
            return [self.accumulators]
        views = [self.accumulators[:n]]
        if self.track_heating:
            views.append(self.accumulators[14 * n:16 * n])
        return views

    def reset_grid(self):
        self.engine.reset_grid()

    def shoot(self, seed, iteration, first, count):
        self.engine.shoot(seed, iteration, first, count)

    def get_counters(self):
        tw, tc, ns = self.engine.get_counters()
        return tw, tc, ns

    def update_cells(self, loop, totweight):
        self.engine.update_cells(loop, totweight)

    # sharded cell update (src/IonizationSimulation.cpp:532-618) -------------
    def sharded_update_pays(self):
        """Solving 1/P of the cells and gathering T + 14 fractions beats every
        rank solving all cells only when the solve is expensive: multi-ion
        runs (the temperature solve is ~9 ns per cell, the gather moves 120 B
        per cell); the hydrogen-only closed form (0.1 ns per cell) does not."""
        _, cell_stride = self.engine.accumulator_layout()
        return cell_stride != 1

    def update_cells_range(self, loop, totweight, first, count):
        self.engine.update_cells_range(loop, totweight, first, count)

    def state_fields(self):
        """The fields a cell update writes and the next iteration reads:
        temperature and the 14 ionic fractions, as torch views of the engine's
        memory."""
        from .engine import FIELD_TEMPERATURE, FIELD_IONIC_FRACTION
        return [self.engine.field_tensor(f) for f in
                [FIELD_TEMPERATURE] +
                [FIELD_IONIC_FRACTION + i for i in range(14)]]

    def refresh_transport_records(self):
        self.engine.refresh_transport_records()

    def synchronize(self):
        self.engine.synchronize()


class ReplicaIterationDriver:
    """One iteration = reset -> shoot my share -> sum-reduce -> update (every
    rank all cells, or - shard_update - every rank its block of cells followed
    by a gather of the new state, the reference's MPI scheme)."""

    def __init__(self, backend, rank=0, world=1, dist=None,
                 shard_update=None):
        self.backend = backend
        self.rank = rank
        self.world = world
        self.dist = dist
        self.totweight = 0.
        self.typecount = np.zeros(4)
        self.nsteps = 0
        if shard_update is None:
            pays = getattr(backend, "sharded_update_pays", None)
            shard_update = bool(pays and pays())
        self.shard_update = bool(shard_update) and world > 1 and \
            hasattr(backend, "update_cells_range")

    def iteration(self, loop, n_packets, seed):
        b = self.backend
        b.reset_grid()
        first, count = distribute_packets(n_packets, self.rank, self.world)
        b.shoot(seed, loop, first, count)
        tw, tc, ns = b.get_counters()
        if self.world > 1:
            import torch
            d = self.dist
            # MPI_Allreduce(SUM) of each accumulator field
            # (src/IonizationSimulation.cpp:459-528): one collective over the
            # contiguous block of fields that can be non-zero instead of 16
            # chunked ones
            active = getattr(b, "active_accumulators", None)
            for view in (active() if active else [b.accumulators]):
                d.all_reduce(view, op=d.ReduceOp.SUM)
            # totweight + typecount (src/IonizationSimulation.cpp:410-414)
            small = torch.tensor([tw, tc[0], tc[1], tc[2], tc[3], float(ns)],
                                 dtype=torch.float64,
                                 device=b.accumulators.device)
            d.all_reduce(small, op=d.ReduceOp.SUM)
            small = small.cpu().numpy()
            tw, tc, ns = small[0], small[1:5], int(small[5])
        self.totweight, self.typecount, self.nsteps = tw, np.asarray(tc), ns
        if not self.shard_update:
            b.update_cells(loop, tw)
            return tw
        # TemperatureCalculator::calculate_temperature on this rank's block of
        # cells, then MPICommunicator::gather of the temperature and the ionic
        # fractions (src/IonizationSimulation.cpp:532-618)
        fields = b.state_fields()
        first, count = distribute_packets(fields[0].numel(), self.rank,
                                          self.world)
        b.update_cells_range(loop, tw, first, count)
        b.synchronize()
        for f in fields:
            _all_gather_blocks(self.dist, f, first, count, self.world)
        sync = getattr(b, "torch", None)
        if sync is not None and sync.cuda.is_available():
            sync.cuda.synchronize()
        b.refresh_transport_records()
        return tw


# ---------------------------------------------------------------------------
# Domain-decomposed mode: the grid is cut into blocks, one per process (the
# reference's DensitySubGridCreator decomposition + the MPI photon-buffer
# exchange of its task-based path, src/DensitySubGridCreator.hpp:314-396,
# src/TaskBasedIonizationSimulation.cpp:643-1073). Every cell has one owner,
# so there is no accumulator reduction; packets that leave a block travel to
# the block that owns the cell they enter.
# ---------------------------------------------------------------------------

FLIGHT_DOUBLES = 16   # include/cmi_gpu.h: CMI_GPU_FLIGHT_DOUBLES
FLIGHT_CELL = 12      # column holding the int64 long index of the entered cell


def default_blocks(world):
    """(bx, by, bz) with bx*by*bz == world, as cubic as possible, larger
    factors first: 8 -> (2,2,2), 4 -> (2,2,1), 2 -> (2,1,1), 6 -> (3,2,1)."""
    best = None
    for bx in range(1, world + 1):
        if world % bx:
            continue
        for by in range(1, world // bx + 1):
            if (world // bx) % by:
                continue
            bz = world // bx // by
            dims = tuple(sorted((bx, by, bz), reverse=True))
            score = (dims[0] - dims[2], dims)
            if best is None or score < best[0]:
                best = (score, dims)
    return best[1]


class DomainDecomposition:
    """Block decomposition of an (nx, ny, nz) grid over bx*by*bz ranks; rank
    r owns block (r // (by*bz), (r // bz) % by, r % bz)."""

    def __init__(self, ncell, blocks):
        self.ncell = tuple(int(n) for n in ncell)
        self.blocks = tuple(int(b) for b in blocks)
        self.world = int(np.prod(self.blocks))
        # first cell of every block per axis (+ the end): as even as possible
        self.edges = []
        for n, b in zip(self.ncell, self.blocks):
            q, r = divmod(n, b)
            e = [0]
            for i in range(b):
                e.append(e[-1] + q + (1 if i < r else 0))
            self.edges.append(e)
        self._edge_tensors = {}

    def block(self, rank):
        """(offset, sub_ncell) of rank's block."""
        bx, by, bz = self.blocks
        idx = (rank // (by * bz), (rank // bz) % by, rank % bz)
        offset = tuple(self.edges[a][idx[a]] for a in range(3))
        size = tuple(self.edges[a][idx[a] + 1] - self.edges[a][idx[a]]
                     for a in range(3))
        return offset, size

    def rank_of_cell(self, cell):
        """Owner of each long index (whole-grid, int64 torch tensor)."""
        import torch
        nx, ny, nz = self.ncell
        key = (cell.device, cell.dtype)
        if key not in self._edge_tensors:
            self._edge_tensors[key] = [
                torch.tensor(self.edges[a][1:-1], dtype=cell.dtype,
                             device=cell.device) for a in range(3)]
        ex, ey, ez = self._edge_tensors[key]
        gz = cell % nz
        gy = (cell // nz) % ny
        gx = cell // (nz * ny)
        ix = torch.bucketize(gx, ex, right=True)
        iy = torch.bucketize(gy, ey, right=True)
        iz = torch.bucketize(gz, ez, right=True)
        return (ix * self.blocks[1] + iy) * self.blocks[2] + iz


class DomainGpuBackend:
    """One block of the grid on one GPU: the HIP engine with a torch-owned
    export buffer (stream handling as in GpuBackend)."""

    def __init__(self, decomposition, rank, anchor, sides, device=0,
                 track_heating=False, export_capacity=1 << 22,
                 periodic=(0, 0, 0)):
        import torch
        from .engine import GpuEngine
        self.torch = torch
        self.decomposition = decomposition
        self.rank = rank
        torch.cuda.set_device(device)
        self.offset, self.sub_ncell = decomposition.block(rank)
        stream = torch.cuda.current_stream().cuda_stream
        self.engine = GpuEngine(
            decomposition.ncell, anchor, sides, periodic, device=device,
            track_heating=track_heating, stream=stream,
            sub_offset=self.offset, sub_ncell=self.sub_ncell)
        self.exports = torch.zeros((int(export_capacity), FLIGHT_DOUBLES),
                                   dtype=torch.float64,
                                   device="cuda:%d" % device)
        self.engine.set_export_buffer(self.exports.data_ptr(),
                                      int(export_capacity))

    def reset_grid(self):
        self.engine.reset_grid()
        self.engine.reset_exports()

    def shoot(self, seed, iteration, first, count):
        self.engine.shoot(seed, iteration, first, count)

    def take_exports(self):
        """The flights that left the block since the last reset_exports():
        a view of the export buffer, valid until the next transport call."""
        return self.exports[:self.engine.get_export_count()]

    def reset_exports(self):
        self.engine.reset_exports()

    def continue_flights(self, seed, iteration, first, rows):
        if rows.shape[0]:
            rows = rows.contiguous()
            # the rows come out of torch work (sort, copies, the all-to-all on
            # RCCL's stream); the engine may enqueue on another stream than
            # torch's current one, so wait for them on the host
            self.torch.cuda.synchronize()
            self.engine.shoot_flights(seed, iteration, first, rows.data_ptr(),
                                      rows.shape[0])
            # the rows must outlive the asynchronous launches that read them
            self._inflight = rows

    def get_counters(self):
        return self.engine.get_counters()

    def update_cells(self, loop, totweight):
        self.engine.update_cells(loop, totweight)

    def synchronize(self):
        self.engine.synchronize()


def route_flights(decomposition, rows):
    """Sort exported flights by the rank that owns the cell they enter:
    (rows sorted by destination, flights per destination rank [world])."""
    import torch
    if rows.shape[0] == 0:
        return rows, torch.zeros(decomposition.world, dtype=torch.int64)
    cell = rows.view(torch.int64)[:, FLIGHT_CELL]
    dest = decomposition.rank_of_cell(cell)
    order = torch.argsort(dest)
    counts = torch.bincount(dest, minlength=decomposition.world)
    return rows[order], counts.cpu()


class DomainIterationDriver:
    """One iteration of the domain-decomposed mode, one block per process:
    reset -> every rank runs through the iteration's packets and flies those
    emitted in its block -> rounds of {all-to-all of the flights that crossed a
    block face, continue them} until no flight is left anywhere -> counters
    reduce -> every rank updates its own cells."""

    def __init__(self, backend, decomposition, rank=0, world=1, dist=None):
        self.backend = backend
        self.decomposition = decomposition
        self.rank = rank
        self.world = world
        self.dist = dist
        self.totweight = 0.
        self.typecount = np.zeros(4)
        self.nsteps = 0
        self.rounds = 0
        self.flights_exchanged = 0
        # seconds this rank spent inside the rounds' collectives (waiting for
        # the slowest rank + the transfer), summed until the caller resets it
        self.idle_s = 0.
        # set by bench.py: time the hand-over collectives (adds a device-wide
        # synchronize per round)
        self.measure_idle = False

    def _exchange(self, rows, counts):
        """One hand-over round: ONE all-gather of every rank's per-owner
        counts (the n x n matrix: what this rank receives, and whether anybody
        sends anything at all - the round's control in a single collective
        and a single read-back), then the all-to-all of the rows. Returns
        (incoming rows, flights moved by all ranks) - (None, 0) when no rank
        has a flight left."""
        import torch
        d = self.dist
        dev = rows.device
        send = counts.to(torch.int64).to(dev)
        flat = torch.empty(self.world * self.world, dtype=torch.int64,
                           device=dev)
        d.all_gather_into_tensor(flat, send.contiguous())
        matrix = flat.view(self.world, self.world).cpu()
        total = int(matrix.sum())
        if total == 0:
            return None, 0
        recv = matrix[:, self.rank]
        incoming = torch.empty((int(recv.sum()), FLIGHT_DOUBLES),
                               dtype=rows.dtype, device=dev)
        d.all_to_all_single(incoming, rows.contiguous(),
                            output_split_sizes=recv.tolist(),
                            input_split_sizes=matrix[self.rank].tolist())
        return incoming, total

    def iteration(self, loop, n_packets, seed):
        import torch
        b = self.backend
        b.reset_grid()
        b.shoot(seed, loop, 0, n_packets)
        self.rounds = 0
        self.flights_exchanged = 0
        import time
        while True:
            rows, counts = route_flights(self.decomposition, b.take_exports())
            if self.world > 1:
                # (take_exports waited for this rank's own flights: from here
                # to the end of the all-to-all the rank waits for the others)
                t0 = time.perf_counter()
                incoming, total = self._exchange(rows, counts)
                if self.measure_idle:
                    # (bench.py's idle_ms figure: wait for the all-to-all
                    # itself. Off by default - the device-wide synchronize
                    # serialises the collective against the next launch)
                    if total != 0 and torch.cuda.is_available():
                        torch.cuda.synchronize()
                    self.idle_s += time.perf_counter() - t0
                if total == 0:
                    break
            else:
                total = rows.shape[0]
                if total == 0:
                    break
                incoming = rows.clone()
            b.reset_exports()
            b.continue_flights(seed, loop, 0, incoming)
            self.rounds += 1
            self.flights_exchanged += total
        tw, tc, ns = b.get_counters()
        if self.world > 1:
            small = torch.tensor([tw, tc[0], tc[1], tc[2], tc[3], float(ns)],
                                 dtype=torch.float64, device=rows.device)
            self.dist.all_reduce(small, op=self.dist.ReduceOp.SUM)
            small = small.cpu().numpy()
            tw, tc, ns = small[0], small[1:5], int(small[5])
        self.totweight, self.typecount, self.nsteps = tw, np.asarray(tc), ns
        b.update_cells(loop, tw)
        return tw


class LocalDomainDriver:
    """The same iteration with all blocks in ONE process (several engines on
    one device): used to check the decomposed mode against the undivided grid
    on a single GPU, and to run grids in blocks without a second process."""

    def __init__(self, backends, decomposition):
        self.backends = backends
        self.decomposition = decomposition
        self.totweight = 0.
        self.typecount = np.zeros(4)
        self.nsteps = 0
        self.rounds = 0
        self.flights_exchanged = 0
        # set measure_parallel to time every block's call with a device
        # synchronize around it: parallel_s is then the time the iteration's
        # transport would take with one device PER block - per round the
        # slowest block, rounds one after the other (bulk-synchronous, as
        # DomainIterationDriver runs them) - and serial_s what it took here
        self.measure_parallel = False
        self.parallel_s = 0.
        self.serial_s = 0.
        self.round_parallel_s = []

    def _timed(self, call):
        import time
        import torch
        if not self.measure_parallel:
            call()
            return 0.
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def iteration(self, loop, n_packets, seed, update=True):
        import torch
        times = []
        for b in self.backends:
            b.reset_grid()
            times.append(self._timed(
                lambda b=b: b.shoot(seed, loop, 0, n_packets)))
        self.parallel_s = max(times)
        self.serial_s = sum(times)
        self.round_parallel_s = [max(times)]  # emission, then every round
        self.rounds = 0
        self.flights_exchanged = 0
        while True:
            routed = [route_flights(self.decomposition, b.take_exports())
                      for b in self.backends]
            total = sum(r[0].shape[0] for r in routed)
            if total == 0:
                break
            incoming = []
            for dest in range(len(self.backends)):
                parts = []
                for rows, counts in routed:
                    start = int(counts[:dest].sum())
                    parts.append(rows[start:start + int(counts[dest])])
                incoming.append(torch.cat(parts).clone())
            times = []
            for b, rows in zip(self.backends, incoming):
                b.reset_exports()
                times.append(self._timed(
                    lambda b=b, rows=rows: b.continue_flights(seed, loop, 0,
                                                              rows)))
            self.parallel_s += max(times)
            self.serial_s += sum(times)
            self.round_parallel_s.append(max(times))
            self.rounds += 1
            self.flights_exchanged += total
        tw, tc, ns = 0., np.zeros(4), 0
        for b in self.backends:
            t, c, n = b.get_counters()
            tw += t
            tc += np.asarray(c)
            ns += n
        self.totweight, self.typecount, self.nsteps = tw, tc, ns
        if update:
            for b in self.backends:
                b.update_cells(loop, tw)
        return tw
