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


def distribute_packets(n_packets, rank, world):
    """[first, first + count) of rank `rank`: MPICommunicator::distribute /
    distribute_block (src/MPICommunicator.hpp:197-239)."""
    q, r = divmod(int(n_packets), int(world))
    first = rank * q + min(rank, r)
    count = q + (1 if rank < r else 0)
    return first, count


class GpuBackend:
    """The HIP engine with torch-owned accumulators (so that torch.distributed
    can reduce them in place) enqueuing on torch's current stream."""

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

    def synchronize(self):
        self.engine.synchronize()


class ReplicaIterationDriver:
    """One iteration = reset -> shoot my share -> sum-reduce -> update."""

    def __init__(self, backend, rank=0, world=1, dist=None):
        self.backend = backend
        self.rank = rank
        self.world = world
        self.dist = dist
        self.totweight = 0.
        self.typecount = np.zeros(4)
        self.nsteps = 0

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
        b.update_cells(loop, tw)
        return tw
