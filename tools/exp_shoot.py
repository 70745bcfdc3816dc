#!/usr/bin/env python3
"""Experiment driver (GPU box): time the transport kernel under different
tuning settings on the converged 256^3 Stromgren field.

    python tools/exp_shoot.py [ncell] [packets]
"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cmacionize_amd import GpuEngine, STROMGREN as S  # noqa: E402
from cmacionize_amd import engine as E  # noqa: E402


def main():
    ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    npk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20000000
    configs = sys.argv[3:]
    eng = GpuEngine((ncell,) * 3, S["anchor"], S["sides"], S["periodic"],
                    device=0, track_heating=False)
    eng.set_sources(S["source_position"], S["source_weight"], S["luminosity"])
    eng.set_spectrum_monochromatic(S["frequency"])
    sigma = np.zeros(14)
    sigma[0] = S["sigma_H"]
    alpha = np.zeros(14)
    alpha[0] = S["alpha_H"]
    eng.set_cross_sections_fixed(sigma)
    eng.set_recombination_rates_fixed(alpha)
    n = ncell ** 3
    x = np.zeros((14, n))
    x[0] = S["xH"]
    x[1] = S["xHe"]
    eng.upload_cells(np.full(n, S["density"]), np.full(n, S["temperature"]), x)
    for loop in range(12):
        eng.reset_grid()
        eng.shoot(42, loop, 0, 10000000)
        tw, _, _ = eng.get_counters()
        eng.update_cells(loop, tw)
    eng.synchronize()

    if not configs:
        configs = [
            "sort_packets=0,aggregate=0",
            "sort_packets=0,aggregate=0,exp_no_atomics=1",
            "sort_packets=1,aggregate=0",
            "sort_packets=1,aggregate=0,exp_no_atomics=1",
            "sort_packets=1,aggregate=1",
            "sort_packets=0,aggregate=1",
        ]
    ref = None
    for cfg in configs:
        kw = dict((k, int(v)) for k, v in
                  (item.split("=") for item in cfg.split(",")))
        base = dict(sort_packets=1, aggregate=2, refill_threshold=64,
                    chunk=64, sort_tau_bits=2, max_blocks_per_cu=8, exp_no_atomics=0,
                    exact_dda=0)
        base.update(kw)
        eng.set_tuning(**base)
        times = []
        for rep in range(3):
            eng.reset_grid()
            eng.get_timing(reset=True)
            eng.shoot(42, 100, 0, npk)
            t = eng.get_timing(reset=True)
            times.append(t["shoot_ms"])
        tw, tc, ns = eng.get_counters()
        J = eng.download_field(E.FIELD_MEAN_INTENSITY)
        ok = ""
        if not base["exp_no_atomics"]:
            if ref is None:
                ref = J
            else:
                ok = "J==ref:%s maxrel=%.2e" % (
                    np.allclose(J, ref, rtol=1e-9, atol=1e-12 * ref.max()),
                    np.max(np.abs(J - ref)) / ref.max())
        ms = min(times)
        na = eng.get_atomic_count()
        nw = eng.get_wave_steps()
        print("%-60s %8.1f ms  %7.1f Mpk/s  %6.2f Gstep/s  steps/pk %.1f "
              "atomics/step %.3f lanes busy %.3f %s" %
              (cfg, ms, npk / ms / 1e3, ns / ms / 1e6, ns / npk, na / max(ns, 1),
               ns / max(64. * nw, 1.), ok),
              flush=True)
    eng.close()


if __name__ == "__main__":
    main()
