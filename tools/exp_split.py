#!/usr/bin/env python3
"""GPU box, experiment build (make -C cmacionize_amd/csrc variant NAME=exp
DEFS=-DCMI_EXPERIMENTS; CMI_GPU_LIBRARY=.../libcmi_gpu_exp.so): what the first
generation's kernel spends where, on a FIXED converged state - the state is
brought up with the unmodified kernel, then one transport step per
`exp_no_atomics` mode is timed without a cell update in between.

    python tools/exp_split.py lexington|diffuse|stromgren [NCELL PACKETS]
"""
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tools")
from run_config import make  # noqa: E402

config = sys.argv[1]
ncell = int(sys.argv[2]) if len(sys.argv) > 2 else 256
npk = int(float(sys.argv[3])) if len(sys.argv) > 3 else 100000000
eng = make(config, ncell)
for loop in range(6):
    eng.reset_grid()
    eng.shoot(42, loop, 0, npk)
    tw, tc, ns = eng.get_counters()
    eng.update_cells(loop, tw)
for mode in (0, 1, 2, 3, 4, 5, 6, 0):
    eng.set_tuning(exp_no_atomics=mode)
    eng.reset_grid()
    eng.get_timing(reset=True)
    eng.shoot(42, 7, 0, npk)
    tw, tc, ns = eng.get_counters()
    launches = eng.get_launch_times()
    first = [ms for ms, pk in launches if pk == npk]
    print("exp_no_atomics=%d  first generation %.1f ms  (all launches %.1f "
          "ms, %.1f steps/packet)" % (mode, first[0] if first else -1.,
                                      sum(ms for ms, _ in launches),
                                      ns / npk), flush=True)
eng.close()
