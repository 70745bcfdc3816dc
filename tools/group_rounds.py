#!/usr/bin/env python3
"""GPU box: what the exchange rounds of a decomposed grid cost on the HOST
(cmi_gpu_group_exchange_flights -> cmi_gpu_group_exchange_stats), with all
engines of the group on the one device of the box - the 8-GPU run's control
flow, its device work serialised.

  1. lexingtonHII40 as 2 x 2 x 2 blocks (config 5's decomposition; the star
     sits on the corner the octants share, every block emits)
  2. stromgren_diffuse with an off-centre star on 4 x 1 x 1 blocks and copies
     of the busy blocks: 4 engines for the star's block, 2 and 1 for its
     neighbours, 1 for the far block (the reference's copy levels,
     src/TaskBasedIonizationSimulation.cpp:514-560): 8 engines

Per iteration: the number of exchange rounds, flights handed over, and per
round the host microseconds until the n x n counts are known, the host
microseconds spent on the owners' threads beyond the longest flight call, and
the whole round.

    python tools/group_rounds.py [NCELL] [PACKETS] [ITERATIONS]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from cmacionize_amd import STROMGREN as S  # noqa: E402
from cmacionize_amd.engine import EngineGroup  # noqa: E402
from cmacionize_amd.simulation import (DomainDecomposition,  # noqa: E402
                                       DomainGpuBackend)
from run_config5 import configure as configure_lexington  # noqa: E402


def configure_diffuse(eng, ncells, source):
    eng.set_sources(source, [1.], S["luminosity"])
    eng.set_spectrum_monochromatic(S["frequency"])
    sigma = np.zeros(14)
    sigma[0] = S["sigma_H"]
    alpha = np.zeros(14)
    alpha[0] = S["alpha_H"]
    eng.set_cross_sections_fixed(sigma)
    eng.set_recombination_rates_fixed(alpha)
    eng.set_reemission(1)
    x = np.zeros((14, ncells))
    x[0] = 1.e-6
    x[1] = 1.e-6
    eng.upload_cells(np.full(ncells, S["density"]),
                     np.full(ncells, S["temperature"]), x)


def run(label, backends, group, npk, iterations):
    print("== %s: %d engines" % (label, len(backends)), flush=True)
    for loop in range(iterations):
        for b in backends:
            b.reset_grid()
        group.exchange_stats(reset=True)
        t0 = time.perf_counter()
        for b in backends:
            b.shoot(42, loop, 0, npk)
        flights = 0
        rounds = 0
        while True:
            n = group.exchange_flights(42, loop)
            if n == 0:
                break
            flights += n
            rounds += 1
        group.reduce_accumulators()
        tw = 0.
        for b in backends:
            b.synchronize()
            tw += b.get_counters()[0]
        t1 = time.perf_counter()
        st = group.exchange_stats(reset=True)
        group.update_cells(loop, tw)
        for b in backends:
            b.synchronize()
        per = max(st["rounds"], 1)
        print("it %d: %2d rounds, %9d flights handed over, transport %7.1f ms "
              "(%5.1f Mpk/s); per round: counts on the host after %7.1f us, "
              "threads %7.1f us, whole round %9.1f us" %
              (loop, rounds, flights, 1e3 * (t1 - t0), npk / (t1 - t0) / 1e6,
               st["counts_us"] / per, st["threads_us"] / per,
               st["total_us"] / per), flush=True)


def main():
    ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    npk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2000000
    iterations = int(sys.argv[3]) if len(sys.argv) > 3 else 4

    dec = DomainDecomposition((ncell,) * 3, (2, 2, 2))
    backends = []
    for rank in range(dec.world):
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             track_heating=True, export_capacity=npk)
        off, size = dec.block(rank)
        configure_lexington(b.engine, ncell, off, size)
        backends.append(b)
    group = EngineGroup([b.engine for b in backends])
    run("lexingtonHII40 %d^3, 2 x 2 x 2 blocks, %.0e packets" % (ncell, npk),
        backends, group, npk, iterations)
    group.close()
    for b in backends:
        b.engine.close()

    # off-centre star in the second of four blocks along x
    side = S["sides"][0]
    source = [[-0.11 * side, 0.03 * side, -0.02 * side]]
    dec = DomainDecomposition((ncell, ncell // 2, ncell // 2), (4, 1, 1))
    ranks = [1, 1, 1, 1, 0, 0, 2, 3]
    # (rank list: four engines for block 1, two for block 0, one each for 2, 3;
    # block 2 is a neighbour of block 1 as well - the reference gives it two,
    # here the eighth engine goes to the far block so that every block has one)
    backends = []
    sides = (S["sides"][0], S["sides"][1] / 2., S["sides"][2] / 2.)
    anchor = (S["anchor"][0], S["anchor"][1] / 2., S["anchor"][2] / 2.)
    for rank in ranks:
        b = DomainGpuBackend(dec, rank, anchor, sides, device=0,
                             export_capacity=2 * npk)
        off, size = dec.block(rank)
        configure_diffuse(b.engine, int(np.prod(size)), source)
        backends.append(b)
    group = EngineGroup([b.engine for b in backends])
    run("stromgren_diffuse %d x %d x %d, 4 x 1 x 1 blocks with copies "
        "(4 + 2 + 1 + 1 engines), %.0e packets" %
        (ncell, ncell // 2, ncell // 2, npk), backends, group, npk,
        iterations)
    group.close()
    for b in backends:
        b.engine.close()


if __name__ == "__main__":
    main()
