#!/usr/bin/env python3
"""Benchmark of the hot path: photon packets/s on the 256^3 benchmark grids.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--config stromgren|stromgren_diffuse|lexington]

One "step" is one full iteration of the reference's loop
(src/IonizationSimulation.cpp:359-643) on every rank:
    reset_grid -> shoot `--packets` packets -> [sum all-reduce of the
    accumulators over ranks] -> cell update.
The grid is first brought to its converged ionization state with untimed
low-statistics iterations (the cost of a packet depends on how far it travels,
so a fully ionised start would be a different workload from the one the
metric is quoted on).

The default config is the one BASELINE.json quotes the metric on
(configs[1]: stromgren 256^3, 1e8 packets, H-only). --config selects the other
single-GPU configs of the scope (configs[2], configs[3]).

N > 1 is the replicated-grid mode of the reference's MPI path: every rank
holds the whole grid and shoots `--packets` packets of its own (weak scaling),
the [16 x ncell] accumulator block is sum-reduced with one RCCL all-reduce.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
PC = 3.086e16
LEXINGTON_ABUNDANCES = [0.1, 2.2e-4, 4.e-5, 3.3e-4, 5.e-5, 9.e-6]

# algorithmic HBM bytes per DDA step: one 16 B
# cell record {n x_H, n x_He} read + 8 B read + 8 B write per accumulator
CONFIGS = {
    "stromgren": dict(
        name="stromgren.param", bytes_per_step=16. + 16. * 1, diffuse=False,
        lexington=False, converge_iterations=20,
        kernel="shoot_kernel<H-only, fast marcher>"),
    "stromgren_diffuse": dict(
        name="stromgren_diffuse.param", bytes_per_step=16. + 16. * 1,
        diffuse=True, lexington=False, converge_iterations=20,
        kernel="shoot_kernel<H-only, re-emission passes>"),
    "lexington": dict(
        name="lexingtonHII40.param", bytes_per_step=16. + 16. * 16,
        diffuse=True, lexington=True, converge_iterations=12,
        kernel="shoot_kernel<14 ions + heating, re-emission passes>"),
}


def setup_engine(backend, ncell, cfg, block=None):
    """block = (offset, size): the engine holds only that part of the grid."""
    from cmacionize_amd import STROMGREN as S
    eng = backend.engine
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


def cpu_baseline(ncell, cfg, engine, seconds=12.):
    """The oracle's transport loop (OpenMP, all host cores) on a bounded
    sample of the same workload: the converged state of the GPU run."""
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
    cores = oracle_lib.num_threads()
    n = 10000 * cores
    t0 = time.perf_counter()
    sim.shoot(42, 1000, 0, n)
    dt = time.perf_counter() - t0
    rate = n / dt
    n2 = int(max(n, min(rate * seconds, 5e7)))
    sim.reset()
    t0 = time.perf_counter()
    sim.shoot(42, 1001, 0, n2)
    dt = time.perf_counter() - t0
    return {"value": n2 / dt, "unit": "packets/s", "cores": cores,
            "kind": "port",
            "sample": "%d packets on the converged %d^3 %s state, transport "
                      "only, OpenMP C oracle, %.1f s" %
                      (n2, ncell, cfg["name"], dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="stromgren", choices=sorted(CONFIGS))
    ap.add_argument("--ncell", type=int, default=256)
    ap.add_argument("--packets", type=float, default=None,
                    help="packets per rank per step (default 1e8, the "
                         "number of photons of all three .param files)")
    ap.add_argument("--converge-iterations", type=int, default=None)
    ap.add_argument("--converge-packets", type=float, default=None,
                    help="packets per rank of the untimed iterations that "
                         "bring the grid to its converged state (default: "
                         "--packets, the reference's run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decomposition", default="replica",
                    choices=["replica", "domain"],
                    help="N > 1: every rank holds the whole grid and the "
                         "accumulators are all-reduced (replica), or every "
                         "rank holds one block and flights are exchanged "
                         "(domain)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.packets is None:
        args.packets = 1e8
    if args.converge_iterations is None:
        args.converge_iterations = cfg["converge_iterations"]
    if args.converge_packets is None:
        args.converge_packets = args.packets

    import torch
    from cmacionize_amd.simulation import GpuBackend, ReplicaIterationDriver
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd import engine as E

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # (CMI_BENCH_BACKEND=gloo: rehearsal of the multi-rank control flow on a
    # box with fewer GPUs than ranks - ranks then share devices)
    collective = os.environ.get("CMI_BENCH_BACKEND", "nccl")
    if collective != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if collective == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda",
                                                           local_rank))
        else:
            dist.init_process_group(collective, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" %
              (args.gpus, world), file=sys.stderr)

    ncell = args.ncell
    npk = int(args.packets)
    domain = args.decomposition == "domain"
    if domain:
        from cmacionize_amd.simulation import (DomainDecomposition,
                                               DomainGpuBackend,
                                               DomainIterationDriver,
                                               default_blocks)
        dec = DomainDecomposition((ncell,) * 3, default_blocks(world))
        # domain mode is STRONG scaling: --packets is the iteration's total.
        # (All packets start in the one block that holds the source, which
        # hands 7/8 of them over in the first round: with N x packets that
        # single hand-over would be N x 7/8 x packets x 128 B.)
        backend = DomainGpuBackend(dec, rank, S["anchor"], S["sides"],
                                   device=local_rank,
                                   track_heating=cfg["lexington"],
                                   export_capacity=npk + 1024)
        setup_engine(backend, ncell, cfg, dec.block(rank))
        driver = DomainIterationDriver(backend, dec, rank, world, dist)
    else:
        backend = GpuBackend((ncell,) * 3, S["anchor"], S["sides"],
                             S["periodic"], device=local_rank,
                             track_heating=cfg["lexington"])
        setup_engine(backend, ncell, cfg)
        driver = ReplicaIterationDriver(backend, rank, world, dist)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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

    # bring the grid to the converged state (untimed); rank 0 follows the
    # ionized volume fraction for the iterations-to-converge figure
    # (first iteration after which V(x_H < 0.5) / V_box changes
    # by less than 1 % between consecutive iterations), at the reference's
    # packet count
    loop = 0
    volume = []
    for _ in range(args.converge_iterations):
        driver.iteration(loop, int(args.converge_packets) *
                         (1 if domain else world), 42)
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
    # replica mode: every rank shoots npk packets, global = npk * world (weak
    # scaling); domain mode: npk packets in total (strong scaling)
    global_packets = npk if domain else npk * world
    for _ in range(args.warmup):
        driver.iteration(loop, global_packets, 42)
        loop += 1
    barrier()
    backend.engine.get_timing(reset=True)
    nsteps_total = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        driver.iteration(loop, global_packets, 42)
        nsteps_total += driver.nsteps
        loop += 1
    barrier()
    elapsed = time.perf_counter() - t0
    timing = backend.engine.get_timing(reset=True)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    final_volume = ionized_fraction()
    if rank == 0:
        total_packets = float(global_packets) * args.steps
        value = total_packets / elapsed
        shoot_s = timing["shoot_ms"] * 1e-3
        # the transport kernel alone (HIP events around each launch on the
        # engine's stream); shoot_ms also holds the packet ordering kernels
        kernel_s = timing["kernel_ms"] * 1e-3
        launches = max(timing["kernel_launches"], 1)
        # DDA steps executed by THIS rank's launches (nsteps is the global sum)
        steps_per_launch = nsteps_total / world / launches
        achieved = (steps_per_launch * cfg["bytes_per_step"] /
                    (kernel_s / launches)) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath)).get(args.config)
            if tj and tj.get("ncell") == ncell:
                # bytes per DDA step measured with rocprofv3 --pmc (separate
                # passes, profiles/README.md), scaled to this launch
                traffic = (tj["hbm_bytes_per_dda_step"] * steps_per_launch)
        out = {
            "metric": "photon packets/sec, 256^3 stromgren",
            "value": value,
            "unit": "packets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
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
                             "iteration (all GPUs together)" if domain else "GPU per iteration",
                             "flight hand-over" if domain else "reduce"),
                "parallelism": ("domain x%d (one block per GPU, all-to-all "
                                "of crossing flights)" if domain else
                                "replica x%d (sum all-reduce of accumulators)")
                               % world,
            },
            "transport_only_packets_per_s": (float(npk) * args.steps /
                                             shoot_s),
            "dda_steps_per_packet": nsteps_total / total_packets,
            "cell_update_ms_per_step": timing["update_ms"] /
            max(timing["update_launches"], 1),
            "ionized_volume_fraction": final_volume,
            "iterations_to_converge": {
                "value": converged_at,
                "criterion": "first iteration from which on the ionized "
                             "volume fraction V(x_H<0.5)/V_box changes by "
                             "less than 1 % per iteration",
                "packets_per_iteration": args.converge_packets *
                (1 if domain else world),
                "ionized_volume_fraction_by_iteration": volume,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": cfg["kernel"],
                "kernel_avg_ms": 1e3 * kernel_s / launches,
                "kernel_launches": launches,
                "bytes_per_dda_step": cfg["bytes_per_step"],
                "dda_steps_per_launch": steps_per_launch,
            },
        }
        if args.config != "stromgren":
            out["metric"] = "photon packets/sec, 256^3 " + args.config
        if not args.no_cpu_baseline and not (domain and world > 1):
            # (needs the whole grid's state on this rank)
            out["cpu_baseline"] = cpu_baseline(ncell, cfg, backend.engine)
        if domain:
            out["exchange_rounds_last_step"] = driver.rounds
            out["flights_exchanged_last_step"] = driver.flights_exchanged
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
