#!/usr/bin/env python3
"""GPU box, experiment build (make -C cmacionize_amd/csrc variant NAME=exp
DEFS=-DCMI_EXPERIMENTS; CMI_GPU_LIBRARY=.../libcmi_gpu_exp.so): the
hydrogen-only first generation (the padded march) by stages, on a FIXED
converged state of stromgren.param at 256^3 with 1e8 packets - the whole
kernel (0), the march alone (12: cell crossings, record loads, optical depth;
no run sums, no table) and the march with the run sums but without the table
(11). The differences are what the run sums and the combining table cost.
Results of modes 11 / 12 are wrong by design; only the kernel is timed.

    python tools/exp_pad_stages.py [NCELL PACKETS]
"""
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tools")
from run_config import make  # noqa: E402

ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 256
npk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100000000
eng = make("stromgren", ncell)
for loop in range(12):
    eng.reset_grid()
    eng.shoot(42, loop, 0, npk)
    tw, tc, ns = eng.get_counters()
    eng.update_cells(loop, tw)
for mode in (0, 12, 11, 0, 12, 11, 0):
    eng.set_tuning(exp_no_atomics=mode)
    eng.reset_grid()
    eng.get_timing(reset=True)
    eng.shoot(42, 13, 0, npk)
    tw, tc, ns = eng.get_counters()
    launches = eng.get_launch_times()
    first = [ms for ms, pk in launches if pk == npk]
    wave_steps = eng.get_wave_steps()
    print("exp_no_atomics=%2d  first generation %.2f ms  (%.1f steps/packet, "
          "%.3g wave iterations)" % (mode, first[0] if first else -1.,
                                     ns / npk, wave_steps), flush=True)
eng.close()
