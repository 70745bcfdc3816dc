"""GPU rehearsal of bench.py's multi-rank control flow: two ranks (sharing
device 0, collectives over gloo - CMI_BENCH_BACKEND) in replica and in domain
mode must run to the end, print one JSON line from rank 0 and reach the same
converged state as the one-rank run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--ncell", "64", "--packets", "4e5",
          "--converge-iterations", "8", "--no-cpu-baseline", "--no-also"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(ranks, extra, launcher=False):
    env = dict(os.environ, CMI_BENCH_BACKEND="gloo")
    if ranks == 1 or not launcher:
        # plain `python bench.py --gpus N`: bench.py starts its N ranks itself
        cmd = [sys.executable, "bench.py", "--gpus", str(ranks)]
    else:
        # the driver's own command line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), "bench.py", "--gpus",
               str(ranks)]
    r = subprocess.run(cmd + COMMON + extra, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    return json.loads(lines[0])


def test_two_ranks_replica_and_domain_reach_the_one_rank_state():
    one = run_bench(1, [])
    assert one["n_gpus"] == 1 and one["scaling"] == "weak"
    for key in ("roofline", "config", "iterations_to_converge"):
        assert key in one
    replica = run_bench(2, [])
    assert replica["n_gpus"] == 2 and replica["scaling"] == "weak"
    assert replica["ranks_in_collective"] == 2
    assert replica["packets_per_rank_per_step"] == 4e5
    launched = run_bench(2, [], launcher=True)
    assert launched["n_gpus"] == 2 and launched["ranks_in_collective"] == 2
    assert launched["ionized_volume_fraction"] == \
        replica["ionized_volume_fraction"]
    assert "replica x2" in replica["config"]["parallelism"]
    domain = run_bench(2, ["--decomposition", "domain"])
    assert domain["n_gpus"] == 2 and domain["scaling"] == "strong"
    assert domain["packets_per_rank_per_step"] == 2e5
    assert domain["flights_exchanged_last_step"] >= 0
    # replica: twice the packets per iteration - the same physical state up to
    # Monte Carlo noise; domain: the same packets as the one-rank run
    ref = one["ionized_volume_fraction"]
    # (at 4e5 packets on 64^3 the noise of the estimator biases the volume)
    assert abs(replica["ionized_volume_fraction"] - ref) < 0.05 * ref
    assert abs(domain["ionized_volume_fraction"] - ref) < 1e-4 * ref
    assert abs(domain["dda_steps_per_packet"] -
               one["dda_steps_per_packet"]) < 1e-6 * one["dda_steps_per_packet"]
    assert replica["value"] > 0. and domain["value"] > 0.


def test_default_line_carries_the_other_single_gpu_configs():
    """The driver's command (`bench.py --gpus 1`, default config) prints the
    headline config's line with the other two single-GPU configs of
    BASELINE.json under `also`, each with its own value, roofline and CPU
    baseline, timed one after the other on the same device."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps",
                        "3", "--warmup", "1", "--ncell", "48", "--packets",
                        "2e5", "--converge-iterations", "6"],
                       cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["metric"] == "photon packets/sec, 48^3 stromgren"
    assert line["cpu_baseline"]["calibration"]["record"].endswith(
        "cpu_calibration.json")
    assert sorted(line["also"]) == ["lexington", "stromgren_diffuse"]
    for name, leg in line["also"].items():
        assert name in leg["metric"]
        assert leg["value"] > 0. and leg["steps"] == 3
        assert abs(leg["ms_per_step"] * 1e-3 * leg["value"] - 2e5) < 1.
        assert leg["transport_only_packets_per_s"] >= leg["value"]
        assert "roofline" in leg and "kernel_avg_ms" in leg["roofline"]
        assert leg["cpu_baseline"]["value"] > 0.
        assert "iterations_to_converge" in leg
    # re-emitted packets fly further than those of the headline config
    assert line["also"]["stromgren_diffuse"]["dda_steps_per_packet"] > \
        line["dda_steps_per_packet"]
    assert "roofline_cell_update" in line["also"]["lexington"]
    assert line["bench_wall_s"] > 0.


def test_more_ranks_than_gpus_is_refused():
    """`--gpus N` with fewer than N devices fails loudly instead of running
    fewer ranks and printing an N-GPU line."""
    import torch
    n = torch.cuda.device_count() + 1
    env = dict(os.environ)
    env.pop("CMI_BENCH_BACKEND", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n)] + COMMON,
                       cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_eight_ranks_on_one_device():
    """The driver's 8-GPU command line with all eight ranks on device 0
    (collectives over gloo): replica mode on the Stromgren config and domain
    mode (2 x 2 x 2 blocks, flights exchanged between ranks, per-rank cell
    update) on lexingtonHII40 - config 5's shape. Eight ranks in every
    collective, rank 0 prints the line, and the state is the one-rank run's
    (domain mode flies the same packets)."""
    small = ["--steps", "2", "--warmup", "1", "--ncell", "48", "--packets",
             "2e5", "--converge-iterations", "6", "--no-cpu-baseline",
             "--no-also"]

    def run(ranks, extra):
        env = dict(os.environ, CMI_BENCH_BACKEND="gloo")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), "bench.py", "--gpus",
               str(ranks)] if ranks > 1 else \
            [sys.executable, "bench.py", "--gpus", "1"]
        r = subprocess.run(cmd + small + extra, cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        return json.loads(lines[0])

    # the driver's command as it stands (no --no-also): the line of N > 1
    # ranks also carries config 5's shape over torch.distributed and the
    # native group driver (one process, eight engines) on both shapes
    import time
    small.remove("--no-also")
    t0 = time.perf_counter()
    replica = run(8, ["--config5-ncell", "48"])
    # the driver allows a scaling run 25 minutes; the extras have a budget of
    # 420 s of their own (bench.py EXTRAS_BUDGET_S)
    assert time.perf_counter() - t0 < 25 * 60
    assert replica["bench_wall_s"] < 25 * 60
    small.append("--no-also")
    assert replica["n_gpus"] == 8 and replica["ranks_in_collective"] == 8
    assert replica["scaling"] == "weak"
    assert replica["packets_per_rank_per_step"] == 2e5
    assert "strong_scaling" in replica
    c5 = replica["config5"]
    assert "error" not in c5, c5
    assert "lexington" in c5["metric"] and "48^3" in c5["metric"]
    assert c5["scaling"] == "strong" and c5["n_gpus"] == 8
    assert c5["exchange_rounds_last_step"] >= 2
    assert c5["flights_exchanged_last_step"] > 0
    assert len(c5["idle_ms_per_step_by_rank"]) == 8
    assert all(ms >= 0. for ms in c5["idle_ms_per_step_by_rank"])
    # every rank emits its own eighth of the packets (the star sits on the
    # blocks' common corner), not all of them
    emitted = c5["emitted_packets_per_step_by_rank"]
    assert len(emitted) == 8 and abs(sum(emitted) - 2e5) < 1.
    assert max(emitted) < 1.1 * 2e5 / 8
    native = replica["native"]
    for shape in ("replica", "config5"):
        assert "error" not in native[shape], native[shape]
        assert native[shape]["engines"] == 8 and native[shape]["value"] > 0.
        assert native[shape]["driver"].startswith("native")
    assert native["config5"]["exchange_rounds_last_step"] >= 2
    assert native["config5"]["exchange_host_us_per_round"]["whole_round"] > 0.
    # the same packets through both drivers: the same state
    assert abs(native["config5"]["ionized_volume_fraction"] -
               c5["ionized_volume_fraction"]) < \
        2e-3 * c5["ionized_volume_fraction"]
    assert abs(native["config5"]["dda_steps_per_packet"] -
               c5["dda_steps_per_packet"]) < 1e-3 * c5["dda_steps_per_packet"]
    one = run(1, ["--config", "lexington"])
    domain = run(8, ["--config", "lexington", "--decomposition", "domain"])
    assert domain["n_gpus"] == 8 and domain["ranks_in_collective"] == 8
    assert domain["scaling"] == "strong"
    assert domain["packets_per_rank_per_step"] == 2e5 / 8
    assert "domain x8" in domain["config"]["parallelism"]
    assert domain["exchange_rounds_last_step"] >= 2
    assert domain["flights_exchanged_last_step"] > 0
    ref = one["ionized_volume_fraction"]
    assert abs(domain["ionized_volume_fraction"] - ref) < 2e-3 * ref
    assert abs(domain["dda_steps_per_packet"] -
               one["dda_steps_per_packet"]) < 1e-3 * one["dda_steps_per_packet"]
