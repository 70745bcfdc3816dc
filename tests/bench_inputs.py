"""The reference's benchmark inputs as templates for tests that edit them.

benchmarks/ holds the reference's files byte for byte (comments, blank lines,
the block of the task-based driver: tests/test_reference_benchmark_files.py).
Tests that replace whole blocks of a file work on `bench_text(name)`: the same
keys and values with the comment lines, the blank lines and the
`TaskBasedIonizationSimulation:` block dropped, one blank line between two
blocks, and "number of photons" spelled the same way in all three files."""
import os

BENCH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(
    __file__))), "benchmarks")


def bench_text(name):
    out, skip = [], False
    for line in open(os.path.join(BENCH, name)).read().split("\n"):
        line = line.rstrip()
        if not line.strip() or line.strip().startswith("#"):
            continue
        if not line.startswith(" "):
            skip = line.startswith("TaskBasedIonizationSimulation:")
            if not skip and out:
                out.append("")
        if not skip:
            # (stromgren_diffuse.param spells its 1e6 packets "1000000")
            out.append(line.replace("number of photons: 1000000",
                                    "number of photons: 1e6"))
    return "\n".join(out) + "\n"
