#!/usr/bin/env python3
"""x_H of the converged stromgren run on the GPU engine, saved for
tools/cpu_baseline_scan.py:  python tools/converged_state.py ncell out.npy"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cmacionize_amd import STROMGREN as S  # noqa: E402
from cmacionize_amd import engine as E  # noqa: E402
from cmacionize_amd.simulation import GpuBackend, ReplicaIterationDriver  # noqa: E402

ncell = int(sys.argv[1])
backend = GpuBackend((ncell,) * 3, S["anchor"], S["sides"], S["periodic"],
                     device=0, track_heating=False)
bench.setup_engine(backend, ncell, bench.CONFIGS["stromgren"])
driver = ReplicaIterationDriver(backend, 0, 1, None)
for loop in range(20):
    driver.iteration(loop, 20000000, 42)
np.save(sys.argv[2], backend.engine.download_field(E.FIELD_IONIC_FRACTION))
