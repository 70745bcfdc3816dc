"""CPU tests (gloo, world_size 2 and 3) of the multi-process replica mode: the
host logic of cmacionize_amd.simulation.ReplicaIterationDriver - packet
partition with disjoint counters, sum all-reduce of the [16][ncell]
accumulator block and of the packet counters, redundant cell update - with
the CPU oracle standing in for the HIP engine as the compute backend.

Mirrors the reference's MPI tests in spirit (test/testMPICommunicator.cpp:
reduce + distribute on 3 local ranks)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCELL = 16
NPACKET = 30001  # not divisible by 2 or 3: exercises the remainder logic
NITER = 3
SEED = 9


class OracleBackend:
    """Same surface as simulation.GpuBackend, computed by the oracle."""

    def __init__(self, ncell):
        import torch
        import oracle_lib
        n = ncell ** 3
        self.accumulators = torch.zeros(16 * n, dtype=torch.float64)
        acc = self.accumulators.numpy().reshape(16, n)
        self.sim = oracle_lib.stromgren_simulation(ncell, accumulators=acc)
        self.nsteps = 0

    def reset_grid(self):
        self.sim.reset()
        self.sim.totweight = 0.
        self.sim.typecount[:] = 0.

    def shoot(self, seed, iteration, first, count):
        self.sim.shoot(seed, iteration, first, count)

    def get_counters(self):
        return self.sim.totweight, self.sim.typecount.copy(), 0

    def update_cells(self, loop, totweight):
        self.sim.update(loop, totweight)

    # the sharded cell update (src/IonizationSimulation.cpp:532-618)
    def update_cells_range(self, loop, totweight, first, count):
        self.sim.update_range(loop, totweight, first, count)

    def state_fields(self):
        import torch
        return [torch.from_numpy(self.sim.temperature)] + \
            [torch.from_numpy(self.sim.x[i]) for i in range(14)]

    def refresh_transport_records(self):
        pass

    def synchronize(self):
        pass


def run_rank(rank, world, port, out_dir, shard=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from cmacionize_amd.simulation import ReplicaIterationDriver
    os.environ["OMP_NUM_THREADS"] = "1"
    # (the oracle's sums in a fixed order: one thread, whatever the
    # environment was when libgomp started)
    import oracle_lib
    oracle_lib.set_num_threads(1)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                            rank=rank, world_size=world)
    backend = OracleBackend(NCELL)
    driver = ReplicaIterationDriver(backend, rank, world, dist,
                                    shard_update=shard)
    assert driver.shard_update == shard
    first = None
    for loop in range(NITER):
        tw = driver.iteration(loop, NPACKET, SEED)
        if first is None:
            first = (backend.accumulators.numpy().copy(),
                     backend.sim.x[0].copy())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), xH=backend.sim.x[0],
             J=backend.accumulators.numpy().copy(), tw=tw,
             tc=driver.typecount, J_first=first[0], xH_first=first[1])
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_distribute_packets():
    """MPICommunicator::distribute / distribute_block semantics
    (src/MPICommunicator.hpp:197-239): contiguous, disjoint, complete, the
    first `remainder` ranks get one extra."""
    from cmacionize_amd.simulation import distribute_packets
    for n in (0, 1, 7, 100, 30001, 10 ** 8):
        for world in (1, 2, 3, 8):
            nxt = 0
            for r in range(world):
                first, count = distribute_packets(n, r, world)
                assert first == nxt
                assert count == n // world + (1 if r < n % world else 0)
                nxt += count
            assert nxt == n


@pytest.mark.parametrize("shard", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_replica_mode_matches_single_process(world, shard, tmp_path, oracle):
    """shard: every rank updates its block of cells only and the new state is
    gathered (the reference's MPI scheme) instead of every rank updating all
    cells."""
    import torch.multiprocessing as mp
    from cmacionize_amd.simulation import ReplicaIterationDriver
    port = free_port()
    mp.spawn(run_rank, args=(world, port, str(tmp_path), shard), nprocs=world,
             join=True)
    # single process reference (one thread: the oracle library is loaded
    # already, the environment variable would come too late - and with the
    # order of its sums left to chance this comparison of a chaotic system
    # fails once in a long while)
    threads = oracle.num_threads()
    oracle.set_num_threads(1)
    try:  # (an assertion that fails must not leave the oracle on one thread)
        backend = OracleBackend(NCELL)
        driver = ReplicaIterationDriver(backend, 0, 1, None)
        first = None
        for loop in range(NITER):
            tw = driver.iteration(loop, NPACKET, SEED)
            if first is None:
                first = (backend.accumulators.numpy().copy(),
                         backend.sim.x[0].copy())
        ref_x = backend.sim.x[0]
        ref_J = backend.accumulators.numpy()
        ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
                 for r in range(world)]
        for r in ranks:
            assert r["tw"] == tw == NPACKET
            assert np.array_equal(r["tc"], driver.typecount)
            # all ranks hold the same reduced accumulators and the same new state
            assert np.array_equal(r["J"], ranks[0]["J"])
            assert np.array_equal(r["xH"], ranks[0]["xH"])
            # first iteration (identical start state): equal to the single-process
            # result up to the summation order of the reduce
            assert np.allclose(r["J_first"], first[0], rtol=1e-12, atol=0.)
            assert np.allclose(r["xH_first"], first[1], rtol=1e-9, atol=0.)
            # later iterations: rounding differences of x_H (the closed form
            # cancels) flip a few absorption events of the next iteration - a
            # Monte Carlo code is chaotic at the ulp level - so only loosely equal
            assert np.allclose(r["J"], ref_J, rtol=1e-3, atol=1e-6 * ref_J.max())
            assert np.allclose(r["xH"], ref_x, rtol=1e-3, atol=0.)
    finally:
        oracle.set_num_threads(threads)
