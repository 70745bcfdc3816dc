#!/usr/bin/env python3
"""Experiment driver (GPU box): time the multi-ion transport launches under
different tuning settings on the converged 256^3 lexington field.

    python tools/exp_lexington.py [ncell] [packets] cfg1 cfg2 ...
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import run_config  # noqa: E402


def main():
    ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    npk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20000000
    configs = sys.argv[3:] or [""]
    eng = run_config.make("lexington", ncell)
    for loop in range(6):
        eng.reset_grid()
        eng.shoot(42, loop, 0, npk // 2)
        tw, _, _ = eng.get_counters()
        eng.update_cells(loop, tw)
    eng.synchronize()
    base = dict(exp_no_atomics=0, aggregate=2, aggregate_reemit=0,
                refill_threshold_reemit=32)
    for cfg in configs:
        kw = dict(base)
        if cfg:
            kw.update((k, int(v)) for k, v in
                      (item.split("=") for item in cfg.split(",")))
        eng.set_tuning(**kw)
        best = None
        for rep in range(2):
            eng.reset_grid()
            eng.get_timing(reset=True)
            eng.shoot(42, 100, 0, npk)
            launches = eng.get_launch_times()
            t = eng.get_timing(reset=True)
            if best is None or t["shoot_ms"] < best[0]:
                best = (t["shoot_ms"], launches)
        tw, tc, ns = eng.get_counters()
        nw = eng.get_wave_steps()
        print("%-40s shoot %7.1f ms  steps/pk %.1f  lanes busy %.3f  "
              "launches: %s" %
              (cfg, best[0], ns / npk, ns / max(64. * nw, 1.),
               " ".join("%.1f/%.2g" % (ms, pk) for ms, pk in best[1][:4])),
              flush=True)
    eng.close()


if __name__ == "__main__":
    main()
