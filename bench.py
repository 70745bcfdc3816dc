#!/usr/bin/env python3
"""Benchmark of the hot path: photon packets/s on the 256^3 benchmark grids.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--config stromgren|stromgren_diffuse|lexington]

One "step" is one full iteration of the reference's loop
(src/IonizationSimulation.cpp:359-643) on every rank:
    reset_grid -> shoot `--packets` packets -> [sum all-reduce of the
    accumulators over ranks] -> cell update.
The grid is first brought to its converged ionization state with the
reference's own run - 20 iterations from the fully ionised start of
HomogeneousDensityFunction - which is also what the `whole_run_packets_per_s`
figure times (N_iter x N_p / sum of the shooting times); the
headline `value` is the steady-state iteration rate on the converged state
(the cost of a packet depends on how far it travels).

The default config is the one BASELINE.json quotes the metric on
(configs[1]: stromgren 256^3, 1e8 packets, H-only). --config selects the other
single-GPU configs of the scope (configs[2], configs[3]). The packet count is
BASELINE.json's (1e8 per iteration; the reference's stromgren*.param files
themselves say 1e6).

N > 1 is the replicated-grid mode of the reference's MPI path: every rank
holds the whole grid, the accumulators are sum-reduced with one RCCL
all-reduce. The timed steps are WEAK scaling (every rank shoots `--packets`
packets, `value` counts all of them); the same line carries a
`strong_scaling` object: the iteration with `--packets` packets IN TOTAL.
"""
import argparse
import json
import os
import signal
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.        # HBM3E peak 8.0 TB/s (spec)
N_SIMD = 256 * 4            # 256 CUs x 4 SIMDs
N_CU = 256
MAX_CLOCK_GHZ = 2.4
FP64_VECTOR_PEAK_TFLOPS = 78.6  # half the 157.3 TFLOP/s fp32 vector rate
PC = 3.086e16
LEXINGTON_ABUNDANCES = [0.1, 2.2e-4, 4.e-5, 3.3e-4, 5.e-5, 9.e-6]

# algorithmic HBM bytes per DDA step, SURVEY.md 8(d)'s figures: H-only =
# n, x_H read (16) + J_H read-modify-write (16) = 32; 14 ions = n, x_H, x_He
# read (24) + 16 accumulators read-modify-write (256) = 280. (The engine's
# own record is 8 / 16 B - the pre-multiplied n x_H [, n x_He] - i.e. 24 /
# 272 B; the figures quoted are the survey's, the larger ones.)
CONFIGS = {
    "stromgren": dict(
        name="stromgren.param", bytes_per_step=16. + 16. * 1, diffuse=False,
        lexington=False, converge_iterations=20,
        kernel="shoot_kernel<false, false, false, false, true, false, true, false, false>"),
    "stromgren_diffuse": dict(
        name="stromgren_diffuse.param", bytes_per_step=16. + 16. * 1,
        diffuse=True, lexington=False, converge_iterations=20,
        kernel="shoot_kernel<false, false, false, false, true, false, true, false, false>"),
    "lexington": dict(
        name="lexingtonHII40.param", bytes_per_step=24. + 16. * 16,
        diffuse=True, lexington=True, converge_iterations=20,
        kernel="shoot_kernel<true, true, false, false, true, true, false, false, false>"),
}



def dominant_kernel(cfg, ncell):
    """The first-generation kernel the engine picks for this grid: beyond 2^25
    cells the hydrogen-only padded march is the BIG build (1024 threads, 4096
    table slots; csrc/kernels.h CMI_TABLE_BIG_CELLS, csrc/engine.hip pad_big) -
    the last template argument."""
    name = cfg["kernel"]
    if not cfg["lexington"] and ncell ** 3 > 1 << 25:
        name = name.replace("true, false, false>", "true, false, true>")
    return name


# The CPU baseline (oracle/cmio_transport_fast.c) against the REFERENCE on the
# same cores: 8-thread Xeon 2.1 GHz of the build container, 64^3, the whole
# 20-iteration run of each .param file. The record is committed:
# profiles/<round>/cpu_calibration.json, written by
# tests/calibrate_cpu_baseline.py (the port's side, run in the build
# container) from BASELINE.md section 2 (the reference's side).
CALIBRATION_FILES = [os.path.join("profiles", r, "cpu_calibration.json")
                     for r in ("r05",)]


def load_calibration(config):
    for rel in CALIBRATION_FILES:
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            record = json.load(open(path))
            cal = dict(record["configs"][config])
            cal["record"] = rel
            return cal
    return None


def setup_engine(backend, ncell, cfg, block=None):
    """block = (offset, size): the engine holds only that part of the grid."""
    from cmacionize_amd import STROMGREN as S
    eng = backend.engine
    # CMI_BENCH_TUNE="key=value,key=value": performance knobs of the engine
    # (cmi_gpu_set_tuning) for kernel experiments; nothing is set by default
    for item in filter(None, os.environ.get("CMI_BENCH_TUNE", "").split(",")):
        key, value = item.split("=")
        eng.set_tuning(**{key: int(value)})
    offset, size = block if block else ((0, 0, 0), (ncell,) * 3)
    n = int(np.prod(size))
    x = np.zeros((14, n))
    x[0] = 1.e-6
    x[1] = 1.e-6
    if not cfg["lexington"]:
        eng.set_sources(S["source_position"], S["source_weight"],
                        S["luminosity"])
        eng.set_spectrum_monochromatic(S["frequency"])
        sigma = np.zeros(14)
        sigma[0] = S["sigma_H"]
        alpha = np.zeros(14)
        alpha[0] = S["alpha_H"]
        eng.set_cross_sections_fixed(sigma)
        eng.set_recombination_rates_fixed(alpha)
        if cfg["diffuse"]:
            eng.set_reemission(1)
        eng.upload_cells(np.full(n, S["density"]),
                         np.full(n, S["temperature"]), x)
        return
    # benchmarks/lexingtonHII40.param + .yml
    eng.set_sources([[0., 0., 0.]], [1.], 4.26e49)
    eng.set_spectrum_planck(40000.)
    eng.set_cross_sections_verner()
    eng.set_recombination_rates_verner()
    eng.set_abundances(LEXINGTON_ABUNDANCES)
    eng.set_reemission(1)
    eng.set_temperature_params(do_temperature_calculation=1,
                               pah_heating_factor=0.)
    ax = [-5. * PC + (np.arange(offset[a], offset[a] + size[a]) + 0.5) *
          (10. * PC / ncell) for a in range(3)]
    X, Y, Z = np.meshgrid(ax[0], ax[1], ax[2], indexing="ij")
    vacuum = (np.sqrt(X * X + Y * Y + Z * Z).ravel() <= 3.e16)
    eng.upload_cells(np.where(vacuum, 0., 1.e8), np.where(vacuum, 0., 8000.),
                     x)


def cpu_quota():
    """CPUs the cgroup grants this process (cpu.max: quota / period), or None
    when unlimited / unknown."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(ncell, config, cfg, engine, seconds=12., hint=None):
    """The CPU form of the transport loop (oracle/cmio_transport_fast.c:
    the reference's classic organisation - cells as structures, one lock per
    cell - OpenMP over all host cores) on a bounded sample of the same
    workload: the converged state of the GPU run. `hint` = (threads, radius)
    found by an earlier leg of the same process: the probing of thread counts
    is then not repeated."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from cmacionize_amd import engine as E
    oracle_lib.build()
    if cfg["lexington"]:
        sim = oracle_lib.lexington_simulation(ncell)
        for ion in range(14):
            sim.x[ion][:] = engine.download_field(E.FIELD_IONIC_FRACTION + ion)
        sim.temperature[:] = engine.download_field(E.FIELD_TEMPERATURE)
    else:
        sim = oracle_lib.stromgren_simulation(ncell, diffuse=cfg["diffuse"],
                                              compact=True)
        sim.x[0][:] = engine.download_field(E.FIELD_IONIC_FRACTION)
    # every packet starts in the 8 cells around the star: with many threads
    # the first steps contend for their cache lines. The reference's
    # task-based path replicates the source subgrid
    # (src/DensitySubGridCreator.hpp:437-531); the port keeps private
    # accumulators per thread for a cube around each source
    # (CMIO_FAST_HOT_RADIUS). How many cores the process really gets is the
    # cgroup's CPU quota, not the number of hardware threads it can see (the
    # GPU boxes: 256 threads visible, cpu.max = 16 CPUs; measured with
    # tools/host_cores_probe.c): probe thread counts around the quota.
    all_cores = oracle_lib.num_threads()
    quota = cpu_quota()
    if hint is None:
        budget = min(all_cores, int(round(quota))) if quota else all_cores
        candidates = sorted(set(max(1, min(all_cores, t)) for t in
                                (budget, 2 * budget, max(budget // 2, 1))),
                            reverse=True)
        best = None
        for threads in candidates:
            for radius in (16, 32):
                os.environ["CMIO_FAST_HOT_RADIUS"] = str(radius)
                oracle_lib.set_num_threads(threads)
                n = 10000 * threads
                sim.reset()
                t0 = time.perf_counter()
                sim.shoot_fast(42, 1000, 0, n)
                rate = n / (time.perf_counter() - t0)
                if best is None or rate > best[0]:
                    best = (rate, threads, radius)
        rate, cores, radius = best
    else:
        cores, radius = hint
        os.environ["CMIO_FAST_HOT_RADIUS"] = str(radius)
        oracle_lib.set_num_threads(cores)
        n = 5000 * cores
        # (the first call touches the 256^3 array of cell structures and the
        # threads' private accumulators for the first time: not timed - a
        # cold probe put the lexington leg's rate at 2.0e5 instead of 5.5e5)
        sim.reset()
        sim.shoot_fast(42, 999, 0, n)
        sim.reset()
        t0 = time.perf_counter()
        sim.shoot_fast(42, 1000, 0, n)
        rate = n / (time.perf_counter() - t0)
    os.environ["CMIO_FAST_HOT_RADIUS"] = str(radius)
    oracle_lib.set_num_threads(cores)
    n2 = int(max(10000 * cores, min(rate * seconds, 2e8)))
    sim.reset()
    t0 = time.perf_counter()
    sim.shoot_fast(42, 1001, 0, n2)
    dt = time.perf_counter() - t0
    oracle_lib.set_num_threads(all_cores)
    cal = load_calibration(config)
    sample = ("%d packets on the converged %d^3 %s state, transport "
              "only, %.1f s, at the best of {1/2, 1, 2} x the CPUs "
              "the cgroup grants this process; the reference's "
              "classic loop restated in C (cells as structures, one "
              "lock per cell, per-thread copies of the accumulators "
              "around the source as in the task-based path, OpenMP)." %
              (n2, ncell, cfg["name"], dt))
    if cal:
        sample += (" Calibration against the reference itself (%s), %d "
                   "threads, 64^3, whole 20-iteration run: this port %.3g, "
                   "reference classic %.3g, reference task-based %.3g "
                   "packets/s (port / classic = %.2f)" %
                   (cal["record"], cal.get("threads", 8), cal["port"],
                    cal["reference_classic"], cal["reference_task_based"],
                    cal["port"] / cal["reference_classic"]))
    return {"value": n2 / dt, "unit": "packets/s", "cores": cores,
            "kind": "port",
            "host_threads_available": all_cores,
            "host_cpu_quota": quota,
            "private_accumulator_radius_cells": radius,
            "sample": sample,
            "calibration": cal}


# PMC profiles of this very command, newest first (tools/round_measure.sh)
PROFILES = [os.path.join("profiles", r, "counters.json")
            for r in ("r06", "r05", "r04", "r03")]

# what each unit of the chip can do per second (MI355X_MICROARCH.md; the
# atomic-request rate is measured: profiles/r01/atomic_rates.txt)
UNIT_PEAKS = {
    "valu-issue": (N_SIMD * MAX_CLOCK_GHZ, "G busy SIMD-cycles/s"),
    "lds": (N_CU * MAX_CLOCK_GHZ, "G busy LDS-cycles/s"),
    "atomic-requests": (23.5, "G 64-B requests/s"),
    "hbm": (HBM_PEAK_GBS, "GB/s"),
}


def load_profile(config, ncell, kernel):
    for rel in PROFILES:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        c = json.load(open(path)).get(config)
        if c and c.get("ncell") == ncell and \
                c["dominant"].get("kernel") == kernel:
            return rel, c
    return None, None


def roofline(config, ncell, cfg, steps_per_launch, lanes_per_wave_step,
             first_generation_ms):
    """The dominant kernel (the first generation's transport launch) against
    the unit of the chip that bounds it.

    `bound` / `achieved` / `peak` / `frac` name the unit with the HIGHEST
    measured utilisation in the PMC profile of this very command
    (profiles/rNN/counters.json: separate rocprofv3 --pmc passes,
    tools/pmc_profile.sh + tools/pmc_rooflines.py; numerator and denominator
    of a ratio from the same pass, so frac <= 1 by construction) - for this
    path that is never HBM: direction-sorted packets, cross-lane run sums and
    the LDS combining table keep 85-97 % of a step's algorithmic bytes on
    chip. SURVEY.md 8(d)'s HBM figures are all here too: `algorithmic_GBps`
    (algorithmic bytes per DDA step x the steps one launch executes / the
    launch's duration from the HIP events of THIS run), `traffic` (bytes the
    fabric moved per launch: PMC, 2 x FETCH_SIZE + WRITE_SIZE),
    `traffic_over_algorithmic`, and `hbm` (traffic against the 8 TB/s peak).
    Without a profile of this kernel the bound is unknown and says so."""
    t = first_generation_ms * 1e-3 if first_generation_ms else None
    algorithmic = (steps_per_launch * cfg["bytes_per_step"]
                   if steps_per_launch else None)
    out = {
        "kernel": dominant_kernel(cfg, ncell),
        "bound": None, "achieved": None, "peak": None, "unit": None,
        "frac": None, "traffic": None,
        "kernel_avg_ms": first_generation_ms,
        "dda_steps_per_launch": steps_per_launch,
        "bytes_per_dda_step_algorithmic": cfg["bytes_per_step"],
        "algorithmic_bytes_per_launch": algorithmic,
        "algorithmic_GBps": algorithmic / t / 1e9 if t and algorithmic
        else None,
        "algorithmic_over_hbm_peak": (algorithmic / t / 1e9 / HBM_PEAK_GBS
                                      if t and algorithmic else None),
    }
    rel, c = load_profile(config, ncell, dominant_kernel(cfg, ncell))
    if c is None:
        out["note"] = ("no PMC profile of this kernel at this grid size under "
                       "profiles/: the bounding unit is unmeasured")
        return out
    k = c["dominant"]
    tp = k["kernel_ms"] * 1e-3   # the profiled launch's own duration
    rate = {
        # busy SIMD cycles (4 x SQ_ACTIVE_INST_VALU) per second
        "valu-issue": k["valu_busy_cycles"] / tp / 1e9,
        # busy LDS cycles (SQ_LDS_IDX_ACTIVE) per second
        "lds": k["lds_busy_cycles"] / tp / 1e9,
        # memory-side atomic requests of 64 B (TCC_EA0_ATOMIC) per second
        "atomic-requests": k["atomic_requests"] / tp / 1e9,
        # fabric traffic
        "hbm": k["hbm_bytes"] / tp / 1e9,
    }
    util = {u: rate[u] / UNIT_PEAKS[u][0] for u in rate}
    bound = max(util, key=util.get)
    out.update({
        "bound": bound,
        "achieved": rate[bound],
        "peak": UNIT_PEAKS[bound][0],
        "unit": UNIT_PEAKS[bound][1],
        "frac": util[bound],
        "traffic": k["hbm_bytes"],
        "traffic_over_algorithmic": (k["hbm_bytes"] / algorithmic
                                     if algorithmic else None),
        "hbm": {"achieved": rate["hbm"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": util["hbm"]},
        "bytes_per_dda_step_measured": k["hbm_bytes"] /
        max(steps_per_launch or 1., 1.),
        "utilization": util,
        "waves_waiting_frac": k.get("wave_wait_frac"),
        "profiled_kernel_ms": k["kernel_ms"],
        "profile": rel,
        "profile_commit": c.get("commit"),
        "profile_stale": bool(first_generation_ms) and
        abs(k["kernel_ms"] - first_generation_ms) > 0.1 * first_generation_ms,
    })
    if steps_per_launch and lanes_per_wave_step:
        # instructions executed per 64-lane iteration of the march loop (all
        # of the kernel's instructions over its loop iterations)
        wave_steps = steps_per_launch / lanes_per_wave_step
        out["lanes_stepping_per_wave_iteration"] = lanes_per_wave_step
        if "valu_insts" in k:
            out["valu_insts_per_wave_iteration"] = k["valu_insts"] / wave_steps
        if "salu_insts" in k:
            out["salu_insts_per_wave_iteration"] = k["salu_insts"] / wave_steps
    if "other_kernels" in c:
        out["other_kernels"] = c["other_kernels"]
    return out


def roofline_cell_update(config, ncell, cfg, update_ms):
    """the cell update of the multi-ion config (ionization
    balance + temperature solve, the `temp_*` pipeline kernels) is fp64
    vector-ALU / transcendental bound - reported separately as vector lane
    operations per second (wave instructions x 64 lanes, PMC SQ_INSTS_VALU of
    the same profile) against the vector fp64 rate (78.6 TFLOP/s = 3.93e13
    fused multiply-adds per second)."""
    rel, c = load_profile(config, ncell, dominant_kernel(cfg, ncell))
    if c is None:
        return None
    ks = [k for k in c.get("other_kernels", [])
          if k["kernel"].startswith(("temp_", "temperature_kernel",
                                     "ionization_kernel"))]
    if not ks:
        return None
    ms = sum(k["ms_per_iteration"] for k in ks)
    insts = sum(k["valu_insts_per_iteration"] for k in ks)
    peak = FP64_VECTOR_PEAK_TFLOPS * 1e12 / 2.
    busy = sum(k["valu_busy"] * k["ms_per_iteration"] for k in ks) / ms
    return {
        "kernels": [k["kernel"] for k in ks],
        "bound": "fp64-valu",
        "achieved": insts * 64. / (ms * 1e-3),
        "peak": peak,
        "unit": "vector lane operations/s",
        "frac": insts * 64. / (ms * 1e-3) / peak,
        "valu_issue_busy": busy,
        "profiled_ms_per_update": ms,
        "this_run_ms_per_update": update_ms,
        "traffic": sum(k["hbm_GBps"] * k["ms_per_iteration"] * 1e6
                       for k in ks),
        "profile": rel,
    }


def launch_ranks(n):
    """`python bench.py --gpus N` outside a launcher: run this script as N
    ranks under torch.distributed.run (one per GPU, RCCL over xGMI) and hand
    through their output and exit code. Nothing here initialises a GPU
    (device_count() does not)."""
    import socket
    import subprocess
    import torch
    if os.environ.get("CMI_BENCH_BACKEND", "nccl") == "nccl" and \
            torch.cuda.device_count() < n:
        print("bench.py: --gpus %d but only %d GPU(s) are visible" %
              (n, torch.cuda.device_count()), file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + \
        sys.argv[1:]
    return subprocess.call(cmd, env=env)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="stromgren", choices=sorted(CONFIGS))
    ap.add_argument("--ncell", type=int, default=256)
    ap.add_argument("--packets", type=float, default=None,
                    help="packets per rank per step (default 1e8: "
                         "BASELINE.json's configs)")
    ap.add_argument("--converge-iterations", type=int, default=None)
    ap.add_argument("--converge-packets", type=float, default=None,
                    help="packets per rank of the untimed iterations that "
                         "bring the grid to its converged state (default: "
                         "--packets)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1: skip the strong-scaling leg")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the `also` legs (the other single-GPU configs "
                         "of BASELINE.json, run after the headline one when "
                         "--config is the default and N = 1)")
    ap.add_argument("--also-steps", type=int, default=None,
                    help="timed steps of each `also` leg (default: "
                         "min(--steps, 20))")
    ap.add_argument("--decomposition", default="replica",
                    choices=["replica", "domain"],
                    help="N > 1: every rank holds the whole grid and the "
                         "accumulators are all-reduced (replica), or every "
                         "rank holds one block and flights are exchanged "
                         "(domain)")
    ap.add_argument("--driver", default="torch", choices=["torch", "native"],
                    help="torch: one process per GPU over torch.distributed "
                         "(RCCL) - the form the round's driver launches; "
                         "native: ONE process drives an engine per GPU "
                         "through the C ABI's group API (cmi_gpu_group_*: "
                         "RCCL reduce / device-side routing of flights over "
                         "peer access) - what the C++ host `cmi-gpu` does")
    ap.add_argument("--config5-ncell", type=int, default=None,
                    help="N > 1, default config: cells per axis of the "
                         "`config5` leg (lexingtonHII40 in blocks; default "
                         "512 at 8 ranks, else 256)")
    ap.add_argument("--copies", type=int, default=1,
                    help="engines per block that holds a source (the "
                         "reference's copies of busy subgrids): only 1 here")
    args = ap.parse_args()
    if args.copies != 1:
        # (DomainIterationDriver hands a flight to THE rank that owns the cell
        # it enters; it does not deal flights to several engines of a block)
        raise SystemExit(
            "bench.py: --copies %d: the torch.distributed drivers run one "
            "engine per block (domain) or one per rank (replica) and know no "
            "copies of busy blocks; copies are the C++ host's - "
            "`cmi-gpu --blocks BX,BY,BZ --devices ... --copies K` over the "
            "group API (cmi_gpu_group_*)" % args.copies)
    if args.packets is None:
        args.packets = 1e8
    if args.converge_packets is None:
        args.converge_packets = args.packets
    return args


class Ranks:
    """This process's place among the ranks of the run."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        # (CMI_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on
        # a box with fewer GPUs than ranks - ranks then share devices)
        collective = os.environ.get("CMI_BENCH_BACKEND", "nccl")
        if collective != "nccl":
            self.local_rank %= max(torch.cuda.device_count(), 1)
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(self.local_rank)
            if collective == "nccl":
                dist.init_process_group(
                    "nccl", rank=self.rank, world_size=self.world,
                    device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(collective, rank=self.rank,
                                        world_size=self.world)
        if args.gpus != self.world:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d: start "
                             "one rank per GPU (plain `python bench.py --gpus "
                             "N` does that by itself)" %
                             (args.gpus, self.world))
        if collective == "nccl" and torch.cuda.device_count() < self.world:
            raise SystemExit("bench.py: %d ranks asked for, %d GPUs visible" %
                             (self.world, torch.cuda.device_count()))

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def run_leg(args, ranks, config, steps, warmup, cpu_seconds=None,
            cpu_hint=None):
    """One benchmark configuration from the ionised start to the JSON record:
    the reference's run (untimed for `value`), `warmup` + `steps` iterations
    on the converged state, the roofline of the first-generation kernel and
    the CPU baseline on the same state. Returns the record on rank 0, None on
    the other ranks; the engine is closed before returning."""
    import torch
    from cmacionize_amd.simulation import GpuBackend, ReplicaIterationDriver
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd import engine as E

    cfg = CONFIGS[config]
    world, rank, local_rank, dist = (ranks.world, ranks.rank,
                                     ranks.local_rank, ranks.dist)
    barrier = ranks.barrier
    converge_iterations = (args.converge_iterations
                           if args.converge_iterations is not None
                           else cfg["converge_iterations"])
    ncell = args.ncell
    npk = int(args.packets)
    domain = args.decomposition == "domain"
    if domain:
        from cmacionize_amd.simulation import (DomainDecomposition,
                                               DomainGpuBackend,
                                               DomainIterationDriver,
                                               default_blocks)
        dec = DomainDecomposition((ncell,) * 3, default_blocks(world))
        # domain mode is STRONG scaling: --packets is the iteration's total
        backend = DomainGpuBackend(dec, rank, S["anchor"], S["sides"],
                                   device=local_rank,
                                   track_heating=cfg["lexington"],
                                   export_capacity=max(npk // 4, 1 << 20))
        setup_engine(backend, ncell, cfg, dec.block(rank))
        driver = DomainIterationDriver(backend, dec, rank, world, dist)
    else:
        backend = GpuBackend((ncell,) * 3, S["anchor"], S["sides"],
                             S["periodic"], device=local_rank,
                             track_heating=cfg["lexington"])
        setup_engine(backend, ncell, cfg)
        driver = ReplicaIterationDriver(backend, rank, world, dist)

    def ionized_fraction():
        """V(x_H < 0.5) / V_box; in domain mode every rank counts its block."""
        if not domain and rank != 0:
            return 0.
        xH = backend.engine.download_field(E.FIELD_IONIC_FRACTION)
        count = float((xH < 0.5).sum())
        if domain and world > 1:
            t = torch.tensor([count], dtype=torch.float64, device="cuda")
            dist.all_reduce(t)
            count = float(t.item())
        return count / float(ncell) ** 3

    # The reference's run, untimed as far as `value` goes: from the fully
    # ionised start to the converged state. Rank 0 follows the ionized volume
    # fraction for the iterations-to-converge figure (first
    # iteration after which V(x_H < 0.5) / V_box changes by less than 1 %
    # between consecutive iterations) and the shooting times for the
    # whole-run rate (the reference's "Total photon shooting time",
    # src/IonizationSimulation.cpp:667-674).
    loop = 0
    volume = []
    whole_run_shoot_ms = 0.
    whole_run_wall = 0.
    backend.engine.get_timing(reset=True)
    for _ in range(converge_iterations):
        barrier()
        t0 = time.perf_counter()
        driver.iteration(loop, int(args.converge_packets) *
                         (1 if domain else world), 42)
        barrier()
        whole_run_wall += time.perf_counter() - t0
        whole_run_shoot_ms += backend.engine.get_timing(reset=True)["shoot_ms"]
        loop += 1
        volume.append(ionized_fraction())
    # (the first iterations of a fully ionized start change little too: what
    # counts is the last change of 1 % or more)
    converged_at = None
    if len(volume) > 1:
        last_big = 0
        for k in range(1, len(volume)):
            if volume[k] <= 0. or \
                    abs(volume[k] - volume[k - 1]) >= 0.01 * volume[k]:
                last_big = k
        if last_big + 1 < len(volume):
            converged_at = last_big + 2  # 1-based count of iterations run

    lanes_per_wave_step = None

    def timed(global_packets, nsteps_timed):
        nonlocal loop, lanes_per_wave_step
        barrier()
        backend.engine.get_timing(reset=True)
        if domain:
            driver.idle_s = 0.
            driver.measure_idle = True
        nsteps = 0
        t0 = time.perf_counter()
        for _ in range(nsteps_timed):
            driver.iteration(loop, global_packets, 42)
            nsteps += driver.nsteps
            loop += 1
        barrier()
        elapsed = time.perf_counter() - t0
        # (the last iteration's counters: lanes that step per wave iteration)
        wave_steps = backend.engine.get_wave_steps()
        if wave_steps:
            lanes_per_wave_step = backend.engine.get_counters()[2] / wave_steps
        launches = backend.engine.get_launch_times()
        lsteps = backend.engine.get_launch_steps()
        # (the step counter after each launch; it restarts with every
        # iteration's reset_grid, so after an iteration's first launch it is
        # that launch's own step count)
        launches = [(ms, n, st) for (ms, n), st in zip(launches, lsteps)]
        timing = backend.engine.get_timing(reset=True)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, nsteps, timing, launches

    # replica mode: every rank shoots npk packets, global = npk * world (weak
    # scaling); domain mode: npk packets in total (strong scaling)
    global_packets = npk if domain else npk * world
    for _ in range(warmup):
        driver.iteration(loop, global_packets, 42)
        loop += 1
    elapsed, nsteps_total, timing, launches = timed(global_packets, steps)
    strong = None
    if world > 1 and not domain and not args.no_strong:
        # the same iteration with npk packets IN TOTAL (npk / world per rank)
        for _ in range(warmup):
            driver.iteration(loop, npk, 42)
            loop += 1
        s_elapsed, _, s_timing, _ = timed(npk, steps)
        strong = {"value": float(npk) * steps / s_elapsed,
                  "unit": "packets/s",
                  "packets_per_iteration_all_gpus": npk,
                  "ms_per_step": 1e3 * s_elapsed / steps,
                  "transport_ms_per_step": s_timing["shoot_ms"] / steps,
                  "cell_update_ms_per_step": s_timing["update_ms"] /
                  max(s_timing["update_launches"], 1)}

    final_volume = ionized_fraction()
    idle_ms = None
    if domain:
        # per rank: time blocked in the rounds' collectives per timed step
        idle_ms = [1e3 * driver.idle_s / max(steps, 1)]
        if world > 1:
            t = torch.tensor(idle_ms, dtype=torch.float64, device="cuda")
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            idle_ms = [float(x.item()) for x in parts]
    emitted = None
    if domain:
        # per rank: the packets its block emitted per step (the flights of
        # the first launch after every reset_grid) - a block picks its own
        # packets out of an iteration's ids (block_select_kernel), so this is
        # ~ 1 / world of them for a source on the blocks' common corner
        fg, before = [], None
        for ms, n, st in launches:
            if before is None or st <= before:
                fg.append(n)
            before = st
        emitted = [float(np.mean(fg)) if fg else 0.]
        if world > 1:
            t = torch.tensor(emitted, dtype=torch.float64, device="cuda")
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            emitted = [float(x.item()) for x in parts]
    out = None
    if rank == 0:
        total_packets = float(global_packets) * steps
        value = total_packets / elapsed
        shoot_s = timing["shoot_ms"] * 1e-3
        # the first-generation transport launch of every iteration: the
        # launches that started this rank's full packet count
        if not domain:
            first_gen = [(ms, st) for ms, n, st in launches if n == npk]
        else:
            # a block flies what starts in it: the first launch of an
            # iteration is the one after which the step counter (it restarts
            # with reset_grid) is not larger than after the launch before
            first_gen, before = [], None
            for ms, n, st in launches:
                if before is None or st <= before:
                    first_gen.append((ms, st))
                before = st
        first_gen_ms = float(np.mean([f[0] for f in first_gen])) \
            if first_gen else None
        first_gen_steps = float(np.mean([f[1] for f in first_gen])) \
            if first_gen else None
        kernel_s = timing["kernel_ms"] * 1e-3
        conv_packets = float(args.converge_packets) * \
            (1 if domain else world) * converge_iterations
        out = {
            "metric": "photon packets/sec, %d^3 %s" % (ncell, config),
            "value": value,
            "unit": "packets/s",
            "n_gpus": world,
            "ranks_in_collective": (dist.get_world_size() if world > 1
                                    else 1),
            "packets_per_rank_per_step": (float(npk) / world if domain
                                          else float(npk)),
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "strong" if domain else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s %d^3 grid, %.0e packets per %s, "
                            "converged ionization state; step = "
                            "reset + shoot + %s + cell update" %
                            (cfg["name"], ncell, npk,
                             "iteration (all GPUs together)" if domain
                             else "GPU per iteration",
                             "flight hand-over" if domain else "reduce"),
                "parallelism": ("domain x%d (one block per GPU, all-to-all "
                                "of crossing flights)" if domain else
                                "replica x%d (sum all-reduce of accumulators)")
                               % world,
            },
            "transport_only_packets_per_s": (float(npk) * steps / shoot_s),
            "transport_kernels_ms_per_step": 1e3 * kernel_s / steps,
            "transport_launches_per_step": len(launches) / steps,
            "dda_steps_per_packet": nsteps_total / total_packets,
            "cell_update_ms_per_step": timing["update_ms"] /
            max(timing["update_launches"], 1),
            "ionized_volume_fraction": final_volume,
            # SURVEY 8d's metric: the whole run from the ionised start,
            # N_iter x N_p / sum of the shooting times (and wall clock)
            "whole_run_packets_per_s": (conv_packets /
                                        (whole_run_shoot_ms * 1e-3)
                                        if whole_run_shoot_ms else None),
            "whole_run": {
                "iterations": converge_iterations,
                "packets_per_iteration": args.converge_packets *
                (1 if domain else world),
                "shoot_s": whole_run_shoot_ms * 1e-3,
                "wall_s": whole_run_wall,
                "end_to_end_packets_per_s": (conv_packets / whole_run_wall
                                             if whole_run_wall else None),
            },
            "iterations_to_converge": {
                "value": converged_at,
                "criterion": "first iteration from which on the ionized "
                             "volume fraction V(x_H<0.5)/V_box changes by "
                             "less than 1 % per iteration",
                "packets_per_iteration": args.converge_packets *
                (1 if domain else world),
                "ionized_volume_fraction_by_iteration": volume,
            },
            "roofline": roofline(config, ncell, cfg, first_gen_steps,
                                 lanes_per_wave_step, first_gen_ms),
        }
        if strong is not None:
            out["strong_scaling"] = strong
        if cfg["lexington"]:
            out["roofline_cell_update"] = roofline_cell_update(
                config, ncell, cfg, out["cell_update_ms_per_step"])
        if lanes_per_wave_step:
            # 64-lane iterations of the march loops per packet (all
            # generations of the last timed step)
            out["wave_iterations_per_packet"] = (
                nsteps_total / total_packets / lanes_per_wave_step)
        if cpu_seconds and world == 1:
            # (rank 0 at N = 1 only: the other ranks would wait for it)
            out["cpu_baseline"] = cpu_baseline(ncell, config, cfg,
                                               backend.engine,
                                               seconds=cpu_seconds,
                                               hint=cpu_hint)
        if domain:
            out["exchange_rounds_last_step"] = driver.rounds
            out["flights_exchanged_last_step"] = driver.flights_exchanged
            out["idle_ms_per_step_by_rank"] = idle_ms
            out["emitted_packets_per_step_by_rank"] = emitted
    # the next leg needs the memory (flight slots and queues of 1e8 packets)
    barrier()
    backend.engine.close()
    del driver, backend
    torch.cuda.empty_cache()
    return out


class _Holder:
    """what setup_engine() configures: something with an `engine`"""

    def __init__(self, engine):
        self.engine = engine


def run_native(args, config, ncell, domain, steps, warmup, devices):
    """The same iteration through the C ABI's group API - ONE process, one
    engine per device (include/cmi_gpu.h, cmi_gpu_group_*; the reference:
    src/TaskBasedIonizationSimulation.cpp:514-560,643-1073 for the blocks,
    src/IonizationSimulation.cpp:459-618 for the replicas): replicas shoot
    their share of the packets, cmi_gpu_group_reduce_accumulators (RCCL),
    sharded update; blocks shoot, cmi_gpu_group_exchange_flights until no
    flight moves (rows routed on the device, written into the owner's inbox
    over peer access), every block updates its cells. Engines are driven from
    one host thread each where a call blocks (re-emission reads a few bytes
    back per round). Returns the record."""
    from concurrent.futures import ThreadPoolExecutor
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.engine import EngineGroup
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend, default_blocks,
                                           distribute_packets)
    cfg = CONFIGS[config]
    n = len(devices)
    npk = int(args.packets)
    converge_iterations = (args.converge_iterations
                           if args.converge_iterations is not None
                           else cfg["converge_iterations"])
    backends = []
    if domain:
        dec = DomainDecomposition((ncell,) * 3, default_blocks(n))
        for r in range(n):
            b = DomainGpuBackend(dec, r, S["anchor"], S["sides"],
                                 device=devices[r],
                                 track_heating=cfg["lexington"],
                                 export_capacity=max(npk // 4, 1 << 20))
            setup_engine(b, ncell, cfg, dec.block(r))
            backends.append(b)
    else:
        for r in range(n):
            b = _Holder(GpuEngine((ncell,) * 3, S["anchor"], S["sides"],
                                  S["periodic"], device=devices[r],
                                  track_heating=cfg["lexington"]))
            setup_engine(b, ncell, cfg)
            backends.append(b)
    group = EngineGroup([b.engine for b in backends])
    pool = ThreadPoolExecutor(max_workers=n)
    state = dict(rounds=0, flights=0, nsteps=0)

    def start(job):
        r, loop, total = job
        b = backends[r]
        if domain:
            b.reset_grid()
            b.shoot(42, loop, 0, total)
        else:
            first, count = distribute_packets(total, r, n)
            b.engine.reset_grid()
            b.engine.shoot(42, loop, first, count)
        return b.engine.get_counters()

    def iteration(loop, total):
        counters = list(pool.map(start, [(r, loop, total) for r in range(n)]))
        rounds = flights = 0
        if domain:
            while True:
                moved = group.exchange_flights(42, loop)
                if moved == 0:
                    break
                flights += moved
                rounds += 1
            counters = [b.engine.get_counters() for b in backends]
        else:
            group.reduce_accumulators()
        tw = sum(c[0] for c in counters)
        state.update(rounds=rounds, flights=flights,
                     nsteps=sum(c[2] for c in counters))
        group.update_cells(loop, tw)

    def synchronize():
        for b in backends:
            b.engine.synchronize()

    # replicas: weak scaling, every engine shoots npk; blocks: npk in total
    total = npk if domain else npk * n
    loop = 0
    for _ in range(converge_iterations):
        iteration(loop, int(args.converge_packets) * (1 if domain else n))
        loop += 1
    for _ in range(warmup):
        iteration(loop, total)
        loop += 1
    synchronize()
    group.exchange_stats(reset=True)
    nsteps = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration(loop, total)
        nsteps += state["nsteps"]
        loop += 1
    synchronize()
    elapsed = time.perf_counter() - t0
    st = group.exchange_stats(reset=True)
    xH = np.concatenate([b.engine.download_field(E.FIELD_IONIC_FRACTION)
                         for b in (backends if domain else backends[:1])])
    out = {
        "metric": "photon packets/sec, %d^3 %s" % (ncell, config),
        "value": float(total) * steps / elapsed,
        "unit": "packets/s",
        "driver": "native (cmi_gpu_group_*: one process, %d engines on "
                  "devices %s)" % (n, sorted(set(devices))),
        "n_gpus": len(set(devices)),
        "engines": n,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "scaling": "strong" if domain else "weak",
        "config": {
            "workload": "%s %d^3 grid, %.0e packets per %s" %
                        (cfg["name"], ncell, npk,
                         "iteration (all GPUs together)" if domain
                         else "GPU per iteration"),
            "parallelism": ("domain x%d: blocks %s, flights routed on the "
                            "device into the owner's inbox (peer access)" %
                            (n, "x".join(map(str, default_blocks(n))))
                            if domain else
                            "replica x%d: RCCL all-reduce of the accumulator "
                            "block, sharded cell update" % n),
        },
        "dda_steps_per_packet": nsteps / (float(total) * steps),
        "ionized_volume_fraction": float((xH < 0.5).mean()),
    }
    if domain:
        per = max(st["rounds"], 1)
        out["exchange_rounds_last_step"] = state["rounds"]
        out["flights_exchanged_last_step"] = state["flights"]
        out["exchange_host_us_per_round"] = {
            "until_counts_known": st["counts_us"] / per,
            "owner_threads_beyond_longest_flight": st["threads_us"] / per,
            "whole_round": st["total_us"] / per}
    pool.shutdown()
    group.close()
    for b in backends:
        b.engine.close()
    return out


def native_devices(n):
    """device of each of the n engines: 0 .. n - 1, or - the rehearsal on a
    box with fewer GPUs (CMI_BENCH_BACKEND=gloo) - shared round robin"""
    import torch
    have = torch.cuda.device_count()
    if have >= n:
        return list(range(n))
    if os.environ.get("CMI_BENCH_BACKEND", "nccl") == "nccl":
        raise SystemExit("bench.py: --driver native --gpus %d but only %d "
                         "GPU(s) are visible" % (n, have))
    return [r % max(have, 1) for r in range(n)]


def child_line(args, config, ncell, domain, steps, warmup, driver,
               timeout=300.):
    """One more leg of an N > 1 run as a job of its own, started by rank 0 once
    the ranks' engines are closed (the other ranks wait on the rendezvous
    store): `driver` "native" - this script with --driver native, one process
    for all GPUs - or "torch" - this script under torch.distributed.run, N
    fresh ranks on a port of their own. Whatever happens to that job - RCCL
    inside one process next to torch's, peer access that the box refuses, a
    rank that dies in a collective - the headline line survives; what went
    wrong is recorded instead. (A leg takes 1-2 minutes at 8 ranks; after 5
    it is given up so that three hanging legs cannot hold the line back for
    longer than a scaling run may take.)"""
    import socket
    import subprocess
    tail = ["--gpus", str(args.gpus), "--steps", str(steps), "--warmup",
            str(warmup), "--config", config, "--ncell", str(ncell),
            "--packets", repr(float(args.packets)), "--converge-packets",
            repr(float(args.converge_packets)), "--decomposition",
            "domain" if domain else "replica", "--no-cpu-baseline",
            "--no-also", "--no-strong"]
    if args.converge_iterations is not None:
        tail += ["--converge-iterations", str(args.converge_iterations)]
    if driver == "native":
        cmd = [sys.executable, os.path.abspath(__file__), "--driver",
               "native"] + tail
    else:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + tail
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT",
                        "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                        "ROLE_NAME", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE",
                        "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
                        "TORCHELASTIC_MAX_RESTARTS",
                        "TORCHELASTIC_USE_AGENT_STORE",
                        "TORCH_NCCL_ASYNC_ERROR_HANDLING")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a session (= process group) of its own: a leg that hangs is ended WITH
    # the ranks its launcher started - killing the launcher alone leaves them
    # on their GPUs, and the next leg finds the devices occupied
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True,
                             start_new_session=True)
    try:
        stdout, stderr = child.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(child.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        child.communicate()
        return {"error": "no result after %.0f s" % timeout}
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    if child.returncode != 0 or not lines:
        return {"error": "exit code %d: %s" % (child.returncode,
                                               stderr.strip()[-600:])}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"error": "unreadable result line (%s): %s" %
                         (e, lines[-1][:200])}


# seconds the extra legs of an N > 1 run may take together
EXTRAS_BUDGET_S = 420

# what an `also` leg carries into the headline line
ALSO_KEYS = ("metric", "value", "unit", "steps", "warmup", "ms_per_step",
             "transport_only_packets_per_s", "transport_kernels_ms_per_step",
             "cell_update_ms_per_step", "dda_steps_per_packet",
             "ionized_volume_fraction", "whole_run_packets_per_s",
             "wave_iterations_per_packet", "config")


def main():
    args = parse_args()
    if args.driver == "native":
        # one process for all GPUs; under a launcher only rank 0 works
        if int(os.environ.get("RANK", "0")) == 0:
            out = run_native(args, args.config, args.ncell,
                             args.decomposition == "domain", args.steps,
                             args.warmup, native_devices(args.gpus))
            print(json.dumps(out))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as
        # child processes, before anything in this process touches a GPU
        # (one process per GPU; the reference: one MPI rank per node,
        # src/MPICommunicator.hpp:207-222)
        sys.exit(launch_ranks(args.gpus))

    t_start = time.perf_counter()
    ranks = Ranks(args)
    also = (ranks.world == 1 and args.config == "stromgren" and
            not args.no_also and args.decomposition == "replica")
    cpu = not args.no_cpu_baseline and ranks.world == 1
    # the CPU sample: ~12 s when the headline config runs alone, ~6 s per leg
    # when the line carries all three (the driver's default run: <= 90 s)
    out = run_leg(args, ranks, args.config, args.steps, args.warmup,
                  cpu_seconds=(6. if also else 12.) if cpu else None)
    if also and ranks.rank == 0:
        # BASELINE.json configs[2] and configs[3] - the other benchmarks
        # north_star names - under the same clock, one after the other on the
        # same GPU: each a full leg of its own (converge, warm up, time,
        # roofline, CPU baseline); the headline keys above stay config 2's
        hint = None
        if "cpu_baseline" in out:
            hint = (out["cpu_baseline"]["cores"],
                    out["cpu_baseline"]["private_accumulator_radius_cells"])
        steps = args.also_steps or min(args.steps, 20)
        out["also"] = {}
        for config in ("stromgren_diffuse", "lexington"):
            leg = run_leg(args, ranks, config, steps, args.warmup,
                          cpu_seconds=6. if cpu else None, cpu_hint=hint)
            rec = {k: leg[k] for k in ALSO_KEYS if k in leg}
            rec["iterations_to_converge"] = \
                leg["iterations_to_converge"]["value"]
            rec["roofline"] = {k: v for k, v in leg["roofline"].items()
                               if k != "other_kernels"}
            if "roofline_cell_update" in leg:
                rec["roofline_cell_update"] = leg["roofline_cell_update"]
            if "cpu_baseline" in leg:
                rec["cpu_baseline"] = {k: v for k, v in
                                       leg["cpu_baseline"].items()
                                       if k != "calibration"}
            out["also"][config] = rec
        out["bench_wall_s"] = time.perf_counter() - t_start
    extras = (ranks.world > 1 and args.config == "stromgren" and
              not args.no_also and args.decomposition == "replica")
    if extras:
        # N > 1, the driver's command: the line also carries config 5's shape
        # (lexingtonHII40 in one block per GPU, flights exchanged over RCCL)
        # and the product's own multi-GPU path (the group API) on both
        # shapes - each as a job of its own (child_line), so that nothing
        # that happens there can take the headline line with it. They need
        # the GPUs to themselves: the other ranks wait on the rendezvous
        # store (a host-side wait: a collective would spin on their devices).
        from datetime import timedelta
        steps = args.also_steps or min(args.steps, 20)
        ncell5 = args.config5_ncell or (512 if ranks.world == 8 else 256)
        store = ranks.dist.distributed_c10d._get_default_store()
        # (the three legs together get EXTRAS_BUDGET_S: a leg that hangs must
        # not hold the headline line back beyond what a scaling run may take)
        t_extras = time.perf_counter()

        def extra(config, ncell, domain, driver):
            left = EXTRAS_BUDGET_S - (time.perf_counter() - t_extras)
            if left < 30.:
                return {"error": "not run: the extras' time budget of %d s "
                                 "was spent" % EXTRAS_BUDGET_S}
            return child_line(args, config, ncell, domain, steps, args.warmup,
                              driver, timeout=min(300., left))

        if ranks.rank == 0:
            # (whatever happens here, the other ranks are released and the
            # headline line is printed)
            try:
                leg = extra("lexington", ncell5, True, "torch")
                keep = ALSO_KEYS + (
                    "n_gpus", "ranks_in_collective",
                    "packets_per_rank_per_step", "scaling",
                    "exchange_rounds_last_step",
                    "flights_exchanged_last_step", "idle_ms_per_step_by_rank",
                    "emitted_packets_per_step_by_rank",
                    "error")
                rec = {k: leg[k] for k in keep if k in leg}
                if "iterations_to_converge" in leg:
                    rec["iterations_to_converge"] = \
                        leg["iterations_to_converge"]["value"]
                out["config5"] = rec
                out["native"] = {
                    "replica": extra("stromgren", args.ncell, False,
                                     "native"),
                    "config5": extra("lexington", ncell5, True, "native")}
            except Exception as e:  # noqa: BLE001 - recorded, not raised
                out.setdefault("config5", {})["error"] = \
                    "extras failed: %s: %s" % (type(e).__name__, e)
            finally:
                out["bench_wall_s"] = time.perf_counter() - t_start
                store.set("cmi_bench_extras_done", "1")
        else:
            store.wait(["cmi_bench_extras_done"],
                       timedelta(seconds=EXTRAS_BUDGET_S + 600))
    if ranks.rank == 0:
        print(json.dumps(out))
    if ranks.world > 1:
        ranks.dist.barrier()
        ranks.dist.destroy_process_group()


if __name__ == "__main__":
    main()
