"""CPU tests of the domain-decomposed host logic (cmacionize_amd.simulation):
block geometry, routing of flights to the owner of the cell they enter, and
the all-to-all rounds of DomainIterationDriver under torch.distributed (gloo,
world sizes 2 and 3). A toy backend stands in for the engine: its "flights"
hop from cell to cell along +x, so the expected tallies are known exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cmacionize_amd.simulation import (FLIGHT_CELL, FLIGHT_DOUBLES,
                                       DomainDecomposition,
                                       DomainIterationDriver, default_blocks,
                                       route_flights)

NCELL = (12, 4, 4)


def test_block_geometry_covers_the_grid_once():
    for blocks in ((2, 2, 2), (3, 1, 1), (1, 2, 3), (5, 3, 1)):
        dec = DomainDecomposition((17, 9, 6), blocks)
        owner = -np.ones((17, 9, 6), dtype=int)
        for rank in range(dec.world):
            off, size = dec.block(rank)
            sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
            assert (owner[sl] == -1).all()
            owner[sl] = rank
        assert (owner >= 0).all()
        cells = torch.arange(17 * 9 * 6, dtype=torch.int64)
        assert np.array_equal(dec.rank_of_cell(cells).numpy(), owner.ravel())
    assert default_blocks(8) == (2, 2, 2) and default_blocks(4) == (2, 2, 1)
    assert default_blocks(2) == (2, 1, 1) and default_blocks(1) == (1, 1, 1)


def test_route_flights_sorts_by_owner():
    dec = DomainDecomposition(NCELL, (3, 1, 1))
    rows = torch.zeros((50, FLIGHT_DOUBLES), dtype=torch.float64)
    cells = torch.randint(0, int(np.prod(NCELL)), (50,), dtype=torch.int64)
    rows.view(torch.int64)[:, FLIGHT_CELL] = cells
    rows[:, 0] = torch.arange(50, dtype=torch.float64)
    routed, counts = route_flights(dec, rows)
    assert counts.sum() == 50
    dest = dec.rank_of_cell(routed.view(torch.int64)[:, FLIGHT_CELL])
    assert (dest[1:] >= dest[:-1]).all()
    assert np.array_equal(np.bincount(dest.numpy(), minlength=3),
                          counts.numpy())
    assert sorted(routed[:, 0].tolist()) == list(range(50))


def start_cell(p):
    return (p % NCELL[0], (p // NCELL[0]) % NCELL[1], (p * 3) % NCELL[2])


def hops(p):
    return (p * 7) % NCELL[0] + 1


def expected(n_packets):
    J = np.zeros(NCELL)
    tc = np.zeros(4)
    for p in range(n_packets):
        x, y, z = start_cell(p)
        for _ in range(hops(p)):
            J[x, y, z] += 1.
            x += 1
            if x >= NCELL[0]:
                tc[0] += 1
                break
        else:
            tc[3] += 1
    return J, tc


class ToyBackend:
    """Flights hop along +x; one tally per cell visited; a flight leaves the
    grid (type 0) or runs out of hops (type 3)."""

    def __init__(self, decomposition, rank):
        self.dec = decomposition
        self.rank = rank
        self.offset, self.size = decomposition.block(rank)
        self.J = np.zeros(NCELL)
        self.exports = []
        self.tc = np.zeros(4)
        self.updated = None

    def _mine(self, x, y, z):
        return all(self.offset[a] <= c < self.offset[a] + self.size[a]
                   for a, c in enumerate((x, y, z)))

    def _fly(self, pid, x, y, z, left):
        while left > 0:
            self.J[x, y, z] += 1.
            left -= 1
            x += 1
            if x >= NCELL[0]:
                self.tc[0] += 1
                return
            if left > 0 and not self._mine(x, y, z):
                row = np.zeros(FLIGHT_DOUBLES)
                row[6] = left
                row[13] = pid
                row.view(np.int64)[FLIGHT_CELL] = \
                    (x * NCELL[1] + y) * NCELL[2] + z
                self.exports.append(row)
                return
        self.tc[3] += 1

    def reset_grid(self):
        self.J[:] = 0.
        self.tc[:] = 0.
        self.exports = []

    def reset_exports(self):
        self.exports = []

    def shoot(self, seed, iteration, first, count):
        for p in range(first, first + count):
            x, y, z = start_cell(p)
            if self._mine(x, y, z):
                self._fly(p, x, y, z, hops(p))

    def take_exports(self):
        if not self.exports:
            return torch.zeros((0, FLIGHT_DOUBLES), dtype=torch.float64)
        return torch.from_numpy(np.array(self.exports))

    def continue_flights(self, seed, iteration, first, rows):
        for row in rows.numpy():
            cell = int(row.view(np.int64)[FLIGHT_CELL])
            z = cell % NCELL[2]
            y = (cell // NCELL[2]) % NCELL[1]
            x = cell // (NCELL[2] * NCELL[1])
            assert self._mine(x, y, z)
            self._fly(int(row[13]), x, y, z, int(row[6]))

    def get_counters(self):
        return float(self.tc.sum()), self.tc.copy(), int(self.J.sum())

    def update_cells(self, loop, totweight):
        self.updated = (loop, totweight)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, blocks, n_packets, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dec = DomainDecomposition(NCELL, blocks)
    backend = ToyBackend(dec, rank)
    driver = DomainIterationDriver(backend, dec, rank, world, dist)
    tw = driver.iteration(2, n_packets, 42)
    J = torch.from_numpy(backend.J.copy())
    dist.all_reduce(J)  # only to collect the blocks for the check
    if rank == 0:
        np.savez(out, J=J.numpy(), tw=tw, tc=driver.typecount,
                 ns=driver.nsteps, rounds=driver.rounds,
                 exchanged=driver.flights_exchanged)
    assert backend.updated == (2, tw)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,blocks", [(2, (2, 1, 1)), (3, (3, 1, 1))])
def test_domain_iteration_over_gloo(tmp_path, world, blocks):
    n_packets = 500
    out = str(tmp_path / "result.npz")
    mp.spawn(worker, args=(world, free_port(), blocks, n_packets, out),
             nprocs=world, join=True)
    got = np.load(out)
    J, tc = expected(n_packets)
    assert np.array_equal(got["J"], J)
    assert got["tw"] == n_packets and np.array_equal(got["tc"], tc)
    assert got["ns"] == J.sum()
    # flights cross at most world - 1 faces
    assert 1 <= got["rounds"] <= world - 1 and got["exchanged"] > 0


def test_single_process_driver_without_exchange():
    dec = DomainDecomposition(NCELL, (1, 1, 1))
    backend = ToyBackend(dec, 0)
    driver = DomainIterationDriver(backend, dec)
    driver.iteration(0, 200, 1)
    J, tc = expected(200)
    assert np.array_equal(backend.J, J) and np.array_equal(driver.typecount, tc)
    assert driver.rounds == 0
