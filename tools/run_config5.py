#!/usr/bin/env python3
"""Config 5's shape on ONE GPU (GPU box): lexingtonHII40 on a 512^3 grid as
2 x 2 x 2 blocks of 256^3 - eight engines in one process handing flights over
through device buffers (LocalDomainDriver) - next to the same grid held by a
single engine. Prints per-iteration timings and the hand-over statistics.

    python tools/run_config5.py [NCELL] [PACKETS] [ITERATIONS] [whole|blocks|both]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cmacionize_amd import GpuEngine, STROMGREN as S  # noqa: E402
from cmacionize_amd import engine as E  # noqa: E402
from cmacionize_amd.simulation import (DomainDecomposition,  # noqa: E402
                                       DomainGpuBackend, LocalDomainDriver,
                                       PC)

LEX = [0.1, 2.2e-4, 4.e-5, 3.3e-4, 5.e-5, 9.e-6]


def configure(eng, ncell, offset, size):
    eng.set_sources([[0., 0., 0.]], [1.], 4.26e49)
    eng.set_spectrum_planck(40000.)
    eng.set_cross_sections_verner()
    eng.set_recombination_rates_verner()
    eng.set_abundances(LEX)
    eng.set_reemission(1)
    eng.set_temperature_params(do_temperature_calculation=1,
                               pah_heating_factor=0.)
    ax = [-5. * PC + (np.arange(offset[a], offset[a] + size[a]) + 0.5) *
          (10. * PC / ncell) for a in range(3)]
    r2 = (ax[0][:, None, None] ** 2 + ax[1][None, :, None] ** 2 +
          ax[2][None, None, :] ** 2)
    vacuum = (r2 <= 3.e16 ** 2).ravel()
    n = vacuum.size
    x = np.zeros((14, n))
    x[0] = 1.e-6
    x[1] = 1.e-6
    eng.upload_cells(np.where(vacuum, 0., 1.e8), np.where(vacuum, 0., 8000.), x)


def main():
    ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    npk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20000000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    what = sys.argv[4] if len(sys.argv) > 4 else "both"
    if what in ("whole", "both"):
        eng = GpuEngine((ncell,) * 3, S["anchor"], S["sides"], (0, 0, 0),
                        device=0, track_heating=True)
        configure(eng, ncell, (0, 0, 0), (ncell,) * 3)
        for loop in range(iters):
            eng.reset_grid()
            t0 = time.perf_counter()
            eng.shoot(42, loop, 0, npk)
            tw, tc, ns = eng.get_counters()
            t1 = time.perf_counter()
            eng.update_cells(loop, tw)
            eng.synchronize()
            t2 = time.perf_counter()
            xH = eng.download_field(E.FIELD_IONIC_FRACTION)
            print("whole  it %d shoot %7.1f ms (%6.1f Mpk/s, %5.1f steps/pk) "
                  "update %7.1f ms ion.vol %.4f" %
                  (loop, 1e3 * (t1 - t0), npk / (t1 - t0) / 1e6, ns / npk,
                   1e3 * (t2 - t1), (xH < 0.5).mean()), flush=True)
        eng.close()
    if what in ("blocks", "both"):
        dec = DomainDecomposition((ncell,) * 3, (2, 2, 2))
        backends = []
        for rank in range(8):
            b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                                 track_heating=True, export_capacity=npk)
            off, size = dec.block(rank)
            configure(b.engine, ncell, off, size)
            backends.append(b)
        driver = LocalDomainDriver(backends, dec)
        driver.measure_parallel = True
        for loop in range(iters):
            t0 = time.perf_counter()
            driver.iteration(loop, npk, 42, update=False)
            t1 = time.perf_counter()
            for b in backends:
                b.update_cells(loop, driver.totweight)
            for b in backends:
                b.synchronize()
            t2 = time.perf_counter()
            ion = sum(float((b.engine.download_field(E.FIELD_IONIC_FRACTION)
                             < 0.5).sum()) for b in backends) / ncell ** 3
            print("blocks it %d shoot %7.1f ms (%6.1f Mpk/s, %5.1f steps/pk) "
                  "update %7.1f ms ion.vol %.4f | %d rounds, %.3g flights "
                  "handed over | blocks' calls %7.1f ms in series, %6.1f ms "
                  "with a device per block (per round the slowest)" %
                  (loop, 1e3 * (t1 - t0), npk / (t1 - t0) / 1e6,
                   driver.nsteps / npk, 1e3 * (t2 - t1), ion, driver.rounds,
                   driver.flights_exchanged, 1e3 * driver.serial_s,
                   1e3 * driver.parallel_s), flush=True)
            print("       per round (emission first), ms: " +
                  " ".join("%.1f" % (1e3 * t)
                           for t in driver.round_parallel_s), flush=True)


if __name__ == "__main__":
    main()
