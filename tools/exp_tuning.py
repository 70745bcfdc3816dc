#!/usr/bin/env python3
"""Experiment driver (GPU box): converge one of the configs once, then time
the transport of an iteration under several tuning settings.

    python tools/exp_tuning.py stromgren|diffuse|lexington NCELL PACKETS \\
        "k=v,k=v" "k=v" ...        ("" = defaults)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from run_config import make  # noqa: E402


def main():
    config, ncell, npk = sys.argv[1], int(sys.argv[2]), int(float(sys.argv[3]))
    settings = sys.argv[4:] or [""]
    eng = make(config, ncell)
    for loop in range(12):
        eng.reset_grid()
        eng.shoot(42, loop, 0, npk // 4)
        tw, _, _ = eng.get_counters()
        eng.update_cells(loop, tw)
    defaults = {}
    for s in settings:
        for item in filter(None, s.split(",")):
            defaults.setdefault(item.split("=")[0], None)
    for s in settings:
        kw = dict((k, int(v)) for k, v in
                  (item.split("=") for item in filter(None, s.split(","))))
        eng.set_tuning(**kw)
        times = []
        for loop in range(12, 15):
            eng.reset_grid()
            eng.get_timing(reset=True)
            eng.shoot(42, loop, 0, npk)
            tw, tc, ns = eng.get_counters()
            tm = eng.get_timing(reset=True)
            times.append(tm["shoot_ms"])
        print("%-60s shoot %s ms  (%.1f Mpk/s)" % (
            s or "(defaults)", " ".join("%7.2f" % t for t in times),
            npk / min(times[1:]) / 1e3), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
