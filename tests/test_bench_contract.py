"""bench.py's host-side logic without a GPU: the command line the round's
driver uses parses to the documented defaults, the CPU calibration record is
the committed one, the roofline record degrades loudly without a profile, and
a plain `--gpus N` with too few GPUs refuses to run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_calibration_record_is_committed_and_complete():
    import bench
    for config in ("stromgren", "stromgren_diffuse", "lexington"):
        cal = bench.load_calibration(config)
        assert cal is not None
        assert cal["record"].startswith("profiles/") and \
            cal["record"].endswith("cpu_calibration.json")
        # the reference's own rates on the same cores
        assert cal["reference_classic"] > 0.9e6
        assert cal["reference_task_based"] > cal["reference_classic"]
        # the port sits between the reference's two paths (8 threads, 64^3)
        assert 0.8 * cal["reference_classic"] < cal["port"] < \
            1.2 * cal["reference_task_based"]
    record = json.load(open(os.path.join(ROOT, cal["record"])))
    assert record["threads"] == 8
    assert record["configs"]["stromgren"]["iterations"] == 20


def test_driver_command_line_defaults(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps",
                                      "20", "--warmup", "5"])
    args = bench.parse_args()
    assert args.config == "stromgren" and args.ncell == 256
    assert args.packets == 1e8 and args.converge_packets == 1e8
    assert args.driver == "torch" and args.decomposition == "replica"
    assert not args.no_also and not args.no_cpu_baseline
    # BASELINE.json's three single-GPU configs
    assert sorted(bench.CONFIGS) == ["lexington", "stromgren",
                                     "stromgren_diffuse"]
    assert bench.CONFIGS["stromgren"]["bytes_per_step"] == 32.
    assert bench.CONFIGS["lexington"]["bytes_per_step"] == 280.


def test_roofline_without_a_profile_says_so():
    import bench
    cfg = bench.CONFIGS["stromgren"]
    r = bench.roofline("stromgren", 48, cfg, 1.e6, 50., 1.0)
    assert r["bound"] is None and r["frac"] is None
    assert "unmeasured" in r["note"]
    assert r["algorithmic_bytes_per_launch"] == 32.e6
    assert abs(r["algorithmic_GBps"] - 32.) < 1e-9


def test_roofline_from_the_committed_profile():
    """the profile of the headline kernel at 256^3: the bounding unit is the
    one with the highest utilisation, frac <= 1"""
    import bench
    cfg = bench.CONFIGS["stromgren"]
    r = bench.roofline("stromgren", 256, cfg, 1.277e10, 57.5, 30.5)
    assert r["profile"] is not None
    assert r["bound"] in bench.UNIT_PEAKS
    assert 0. < r["frac"] <= 1.
    assert r["frac"] == max(r["utilization"].values())
    assert r["traffic"] > 0. and r["traffic_over_algorithmic"] < 1.


def test_too_few_gpus_is_refused_before_anything_runs():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("CMI_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                        "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) are visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
