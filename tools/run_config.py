#!/usr/bin/env python3
"""Run one of the scope's configs on the GPU engine and report per-iteration
timings (GPU box):

    python tools/run_config.py stromgren|diffuse|lexington NCELL PACKETS ITERS
        [key=value tuning ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cmacionize_amd import GpuEngine, STROMGREN as S  # noqa: E402
from cmacionize_amd import engine as E  # noqa: E402

PC = 3.086e16
LEX = [0.1, 2.2e-4, 4.e-5, 3.3e-4, 5.e-5, 9.e-6]


def make(config, ncell):
    anchor = (-5. * PC,) * 3
    sides = (10. * PC,) * 3
    n = ncell ** 3
    if config in ("stromgren", "diffuse"):
        eng = GpuEngine((ncell,) * 3, anchor, sides, (0, 0, 0), device=0,
                        track_heating=False)
        eng.set_sources(S["source_position"], S["source_weight"],
                        S["luminosity"])
        eng.set_spectrum_monochromatic(S["frequency"])
        sigma = np.zeros(14)
        sigma[0] = S["sigma_H"]
        alpha = np.zeros(14)
        alpha[0] = S["alpha_H"]
        eng.set_cross_sections_fixed(sigma)
        eng.set_recombination_rates_fixed(alpha)
        if config == "diffuse":
            eng.set_reemission(1)
        x = np.zeros((14, n))
        x[0] = 1.e-6
        x[1] = 1.e-6
        eng.upload_cells(np.full(n, 1.e8), np.full(n, 8000.), x)
        return eng
    eng = GpuEngine((ncell,) * 3, anchor, sides, (0, 0, 0), device=0,
                    track_heating=True)
    eng.set_sources([[0., 0., 0.]], [1.], 4.26e49)
    eng.set_spectrum_planck(40000.)
    eng.set_cross_sections_verner()
    eng.set_recombination_rates_verner()
    eng.set_abundances(LEX)
    eng.set_reemission(1)
    eng.set_temperature_params(do_temperature_calculation=1,
                               pah_heating_factor=0.)
    ax = -5. * PC + (np.arange(ncell) + 0.5) * (10. * PC / ncell)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    r = np.sqrt(X * X + Y * Y + Z * Z).ravel()
    dens = np.where(r <= 3.e16, 0., 1.e8)
    T = np.where(r <= 3.e16, 0., 8000.)
    x = np.zeros((14, n))
    x[0] = 1.e-6
    x[1] = 1.e-6
    eng.upload_cells(dens, T, x)
    return eng


def main():
    config = sys.argv[1]
    ncell = int(sys.argv[2])
    npk = int(float(sys.argv[3]))
    iters = int(sys.argv[4])
    tuning = dict((k, int(v)) for k, v in
                  (a.split("=") for a in sys.argv[5:]))
    eng = make(config, ncell)
    if tuning:
        eng.set_tuning(**tuning)
    for loop in range(iters):
        eng.reset_grid()
        eng.get_timing(reset=True)
        t0 = time.perf_counter()
        eng.shoot(42, loop, 0, npk)
        tw, tc, ns = eng.get_counters()
        t1 = time.perf_counter()
        eng.update_cells(loop, tw)
        eng.synchronize()
        t2 = time.perf_counter()
        launches = eng.get_launch_times()
        tm = eng.get_timing(reset=True)
        na = eng.get_atomic_count()
        xH = eng.download_field(E.FIELD_IONIC_FRACTION)
        T = eng.download_field(E.FIELD_TEMPERATURE)
        print("it %2d shoot %8.1f ms (%7.1f Mpk/s, %5.1f steps/pk, %5.2f "
              "Gstep/s, %.2f atomics/step) update %8.1f ms | abs %.3f difHI "
              "%.3f difHeI %.3f | ion.vol %.4f <T>ion %.0f" %
              (loop, tm["shoot_ms"], npk / tm["shoot_ms"] / 1e3, ns / npk,
               ns / tm["shoot_ms"] / 1e6, na / max(ns, 1), tm["update_ms"],
               tc[3] / tw, tc[1] / tw, tc[2] / tw, (xH < 0.5).mean(),
               T[xH < 0.5].mean() if (xH < 0.5).any() else 0.), flush=True)
        if os.environ.get("CMI_SHOW_LAUNCHES"):
            print("      launches: " + " ".join(
                "%.1fms/%.2gpk" % (ms, pk) for ms, pk in launches), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
