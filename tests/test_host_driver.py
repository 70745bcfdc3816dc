"""Tests of the C++ host layer above the C ABI (cmacionize_amd/host): the
.param parser with units and defaults, the plugin factories and their
lowering, and - on the GPU - the cmi-gpu executable end to end against the
oracle.

Mirrors the reference's testParameterFile.cpp / testUnitConverter.cpp in
spirit (same grammar, same unit arithmetic)."""
import json
import os
import subprocess

import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "cmacionize_amd", "cmi-gpu")
BENCH = os.path.join(ROOT, "benchmarks")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "cmacionize_amd",
                                                   "csrc")], check=True)
        subprocess.run(["make", "-C", os.path.join(ROOT, "cmacionize_amd",
                                                   "host")], check=True)
    return EXE


from bench_inputs import bench_text  # noqa: E402


def describe(exe, param, cwd):
    out = subprocess.run([exe, "--params", param, "--dry-run", "--describe"],
                         check=True, capture_output=True, text=True, cwd=cwd)
    return json.loads(out.stdout)


def unit_power(value, power):
    """Unit::operator^=, src/Unit.hpp:124-150"""
    if power >= 0:
        v = value
        for _ in range(1, power):
            v *= value
        return v
    v = 1.
    for _ in range(-power):
        v /= value
    return v


def test_stromgren_param_is_lowered_like_the_reference(exe, tmp_path):
    d = describe(exe, os.path.join(BENCH, "stromgren.param"), str(tmp_path))
    pc = 3.086e16
    assert d["anchor"] == [-5. * pc] * 3
    assert d["sides"] == [10. * pc] * 3
    assert d["periodicity"] == [0, 0, 0]
    assert d["ncell"] == [64, 64, 64]
    assert d["number_of_iterations"] == 20
    assert d["number_of_photons"] == 1000000
    assert d["random_seed"] == 42  # default
    assert d["sources"] == [{"position": [0, 0, 0], "weight": 1}]
    assert d["total_luminosity"] == 4.26e49
    eV, h = 1.6021766208e-19, 6.626070040e-34
    assert d["spectrum"] == {"type": "Monochromatic",
                             "frequency": 13.6 * eV * (1. / h) / 1.}
    assert d["cross_sections"][0] == 6.3e-18 * unit_power(0.01, 2)
    assert d["cross_sections"][1:] == [0] * 13
    assert d["recombination_rates"][0] == 4.e-13 * (unit_power(0.01, 3) * 1.)
    assert d["abundances"] == [0] * 6  # FixedValueAbundanceModel default 0
    assert d["reemission"]["type"] == 0
    assert d["temperature"]["do"] == 0
    assert d["temperature"]["cr_scale"] == 1.33333 * 3.086e19
    assert not os.path.exists(os.path.join(BENCH,
                                           "stromgren.param.used-values"))


def test_diffuse_and_lexington_params(exe, tmp_path):
    d = describe(exe, os.path.join(BENCH, "stromgren_diffuse.param"),
                 str(tmp_path))
    assert d["reemission"]["type"] == 1
    d = describe(exe, os.path.join(BENCH, "lexingtonHII40.param"),
                 str(tmp_path))
    assert d["spectrum"] == {"type": "Planck", "temperature": 40000.}
    assert d["cross_sections"] == "Verner"
    assert d["recombination_rates"] == "Verner"
    assert d["abundances"] == [0.1, 2.2e-4, 4.e-5, 3.3e-4, 5.e-5, 9.e-6]
    assert d["number_of_photons"] == 100000000  # "1e8"
    assert d["temperature"]["do"] == 1 and d["temperature"]["pah"] == 0
    assert d["reemission"]["type"] == 1


PARAM = """# comment line
SimulationBox:
  anchor: [0. m, -1. cm, 2. km]   # trailing comment
  sides: [1. kpc, 1. pc, 1. au]
  periodicity: [true, no, Y]
DensityGrid:
  number of cells: [8, 4, 2]
DensityFunction:
  type: Homogeneous
  density: 5. cm^-3
PhotonSourceSpectrum:
  type: Monochromatic
  frequency: 912. angstrom
IonizationSimulation:
  number of photons: 2e3
  random seed: 7
CrossSections:
  type: FixedValue
RecombinationRates:
  type: FixedValue
DiffuseReemissionHandler:
  type: FixedValue
  reemission probability: 0.25
DensityGridWriter:
  type: AsciiFile
"""


def test_parameter_grammar_units_and_defaults(exe, tmp_path):
    p = tmp_path / "t.param"
    p.write_text(PARAM)
    d = describe(exe, str(p), str(tmp_path))
    assert d["anchor"] == [0., -0.01, 2000.]
    assert d["sides"] == [3.086e19, 3.086e16, 149597870700.]
    assert d["periodicity"] == [1, 0, 1]
    assert d["ncell"] == [8, 4, 2]
    assert d["number_of_photons"] == 2000
    assert d["number_of_iterations"] == 10  # default
    assert d["random_seed"] == 7
    # wavelength -> frequency: c / lambda (src/UnitConverter.hpp:276-280)
    assert d["spectrum"]["frequency"] == (1. / (912. * 1.e-10)) * 299792458.
    # defaults of the FixedValue plugins
    assert d["cross_sections"][0] == 6.3e-18 * unit_power(0.01, 2)
    assert d["recombination_rates"][0] == 4.e-13 * unit_power(0.01, 3)
    assert d["reemission"] == {"type": 2, "probability": 0.25,
                               "frequency": 19.8 * 1.6021766208e-19 *
                               (1. / 6.626070040e-34)}


def test_errors_are_reported_not_aborted(exe, tmp_path):
    p = tmp_path / "bad.param"
    p.write_text("DensityGrid:\n  type: Voronoi\nDensityGridWriter:\n"
                 "  type: AsciiFile\n")
    r = subprocess.run([exe, "--params", str(p), "--dry-run"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "not on this path" in r.stderr
    p.write_text("SimulationBox:\n  anchor: [0. furlong, 0. m, 0. m]\n")
    r = subprocess.run([exe, "--params", str(p), "--dry-run"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "Unknown unit" in r.stderr
    r = subprocess.run([exe, "--params", str(tmp_path / "missing.param")],
                       capture_output=True, text=True)
    assert r.returncode == 1


@pytest.mark.gpu
def test_cmi_gpu_executable_end_to_end(exe, tmp_path, oracle):
    """stromgren at 16^3 through the executable: used-values file, initial and
    final AsciiFile snapshots (reference column layout), result equal to the
    oracle driven with the same lowered values and seeds."""
    text = bench_text("stromgren.param")
    text = text.replace("[64, 64, 64]", "[16, 16, 16]")
    text = text.replace("number of photons: 1e6", "number of photons: 20000")
    text = text.replace("number of iterations: 20", "number of iterations: 3")
    text = text.replace("type: Gadget", "type: AsciiFile")
    p = tmp_path / "small.param"
    p.write_text(text)
    r = subprocess.run([exe, "--params", str(p), "--output-statistics"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "Total photon shooting time" in r.stdout
    assert "Escape fraction" in r.stdout
    used = open(str(p) + ".used-values").read()
    assert "random seed: 42 # (default value)" in used
    assert "number of photons: 20000 # (20000)" in used
    first = np.loadtxt(tmp_path / "stromgren_000.txt")
    last = np.loadtxt(tmp_path / "stromgren_003.txt")
    assert first.shape == (4096, 6) and last.shape == (4096, 6)
    assert np.all(first[:, 5] == 1.e-6)
    d = describe(exe, str(p), str(tmp_path))
    sim = oracle.OracleSimulation((16,) * 3, d["anchor"], d["sides"])
    sim.set_sources([[0., 0., 0.]], [1.], d["total_luminosity"])
    sim.set_homogeneous(100. * (1. / 0.01 / 0.01 / 0.01), 8000.)
    m = sim.model
    m.spectrum_type = oracle.SPECTRUM_MONOCHROMATIC
    m.mono_frequency = d["spectrum"]["frequency"]
    m.xsec_type = oracle.XSEC_FIXED
    m.recomb_type = oracle.RECOMB_FIXED
    for i in range(14):
        m.xsec_fixed[i] = d["cross_sections"][i]
        m.recomb_fixed[i] = d["recombination_rates"][i]
    sim.run(20000, 3, seed=42)
    # text output has 6 significant digits; iteration-to-iteration rounding
    # feedback as in test_replica_distributed
    assert np.allclose(last[:, 5], sim.x[0], rtol=2e-3, atol=0.)
    mid = (np.arange(16) + 0.5) * (d["sides"][0] / 16) + d["anchor"][0]
    assert np.allclose(last[:16, 2], mid, rtol=1e-5)
    assert np.allclose(last[:, 4], (d["sides"][0] / 16) ** 3, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("bench,blocks", [("stromgren.param", "2,1,2"),
                                          ("stromgren_diffuse.param", "2,2,2"),
                                          ("lexingtonHII40.param", "1,3,1")])
def test_cmi_gpu_executable_with_blocks(exe, tmp_path, bench, blocks):
    """--blocks: the C++ host drives one engine per block of the grid; the
    flights are handed over device to device (cmi_gpu_group_exchange_flights:
    a routing kernel writes them into the owner's inbox); the snapshots equal
    those of the undivided run of the same parameter file."""
    text = bench_text(bench)
    text = text.replace("[64, 64, 64]", "[18, 18, 18]")
    for old in ("number of photons: 1e6", "number of photons: 1e8"):
        text = text.replace(old, "number of photons: 20000")
    text = text.replace("number of iterations: 20", "number of iterations: 5")
    text = text.replace("type: Gadget", "type: AsciiFile")
    outputs = {}
    for label, extra in (("whole", []), ("blocks", ["--blocks", blocks])):
        d = tmp_path / label
        d.mkdir()
        if bench.startswith("lexington"):
            import shutil
            shutil.copy(os.path.join(BENCH, "lexingtonHII40.yml"), d)
        p = d / "run.param"
        p.write_text(text)
        r = subprocess.run([exe, "--params", str(p), "--output-statistics"] +
                           extra, capture_output=True, text=True, cwd=str(d))
        assert r.returncode == 0, r.stderr
        if extra:
            assert "Flights handed over between blocks" in r.stdout
        snapshots = sorted(f for f in os.listdir(d) if f.endswith("005.txt"))
        assert len(snapshots) == 1, os.listdir(d)
        outputs[label] = (np.loadtxt(d / snapshots[0]), r.stdout)
    whole, blocks_out = outputs["whole"][0], outputs["blocks"][0]
    assert whole.shape == blocks_out.shape == (18 ** 3, 6)
    assert np.array_equal(whole[:, :5], blocks_out[:, :5])
    # identical packets and path lengths; 6 printed digits. The balance
    # amplifies the 1e-15 differences of the sums from iteration to iteration
    # until, after a few iterations, a packet near the ionization front is
    # absorbed one cell earlier or later: most cells agree to the printed
    # digits, a few differ by one packet's worth
    rel = np.abs(whole[:, 5] - blocks_out[:, 5]) / whole[:, 5]
    assert np.median(rel) < 1e-5
    assert (rel < 1e-2).mean() > 0.97, (rel > 1e-2).sum()
    # the reference's statistics lines agree to the printed precision
    stats = [[l for l in out.splitlines() if "Escape fraction" in l][-1]
             for _, out in outputs.values()]
    assert stats[0] == stats[1]


@pytest.mark.gpu
@pytest.mark.parametrize("bench", ["stromgren.param", "lexingtonHII40.param"])
def test_cmi_gpu_executable_with_replicas(exe, tmp_path, bench):
    """--devices D0,D1 without --blocks: the reference's MPI scheme in one
    process - every device holds the whole grid and flies its share of the
    packets (disjoint Philox counters), the accumulators are summed over the
    replicas (cmi_gpu_group_reduce_accumulators: RCCL all-reduce between
    distinct devices; replicas that share this box's one device are summed by
    a kernel), every replica updates its cells. Same packets as the
    single-engine run: equal snapshots."""
    text = bench_text(bench)
    text = text.replace("[64, 64, 64]", "[18, 18, 18]")
    for old in ("number of photons: 1e6", "number of photons: 1e8"):
        text = text.replace(old, "number of photons: 20001")
    text = text.replace("number of iterations: 20", "number of iterations: 5")
    text = text.replace("type: Gadget", "type: AsciiFile")
    outputs = {}
    for label, extra in (("one", []), ("three", ["--devices", "0,0,0"])):
        d = tmp_path / label
        d.mkdir()
        if bench.startswith("lexington"):
            import shutil
            shutil.copy(os.path.join(BENCH, "lexingtonHII40.yml"), d)
        p = d / "run.param"
        p.write_text(text)
        r = subprocess.run([exe, "--params", str(p), "--output-statistics"] +
                           extra, capture_output=True, text=True, cwd=str(d))
        assert r.returncode == 0, r.stderr
        if extra:
            assert "Replica mode: 3 devices" in r.stdout
        snapshots = sorted(f for f in os.listdir(d) if f.endswith("005.txt"))
        assert len(snapshots) == 1, os.listdir(d)
        outputs[label] = (np.loadtxt(d / snapshots[0]), r.stdout)
    one, three = outputs["one"][0], outputs["three"][0]
    assert np.array_equal(one[:, :5], three[:, :5])
    rel = np.abs(one[:, 5] - three[:, 5]) / one[:, 5]
    assert np.median(rel) < 1e-5
    assert (rel < 1e-2).mean() > 0.97, (rel > 1e-2).sum()
    stats = [[l for l in out.splitlines() if "Escape fraction" in l][-1]
             for _, out in outputs.values()]
    assert stats[0] == stats[1]


def test_parameter_file_fixture_of_the_reference(tmp_path):
    """test/testParameterFile.cpp:78-150 on the reference's own test.param
    (tests/golden/test.param): integers in decimal, hexadecimal and exponent
    notation, the eight spellings of a boolean, units, vectors of numbers with
    and without units, groups in groups, inline comments, defaults."""
    out = tmp_path / "parameter_file_cli"
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror",
                    "-I", os.path.join(ROOT, "cmacionize_amd", "host"),
                    "-o", str(out),
                    os.path.join(ROOT, "tests", "support",
                                 "parameter_file_cli.cpp"), "-lz"], check=True)
    r = subprocess.run([str(out), os.path.join(ROOT, "tests", "golden",
                                               "test.param")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    v = json.loads(r.stdout)
    assert [v["test_integer%d" % i] for i in range(1, 6)] == \
        [42, 42, 42, 1000000, 1000000]
    assert v["test_float"] == 3.14 and v["test_unit"] == 3.086e16
    assert [v["test_bool%d" % i] for i in range(1, 9)] == \
        [True] * 4 + [False] * 4
    assert v["test_string"] == "This is a test string."
    assert v["group_member"] == 42 and v["group_group_member"] == 42
    assert v["comments_value"] == "test comments string"
    assert v["vector_unit"] == [3.086e16, 2., 2.4e19]
    assert v["vector_int"] == [42, 42, 40]
    assert v["vector_bool"] == [False, True, True]
    assert v["not_in_file1"] == 42 and v["not_in_file2"] == 3.14
    assert v["unit_not_in_file"] == 3.086e16 and v["not_in"] == "file?"
    assert v["not_in_file3"] is True


def test_block_syntax_fixture_of_the_reference(exe, tmp_path):
    """test/testBlockSyntaxDensityFunction.cpp:30-50 with the reference's own
    block file (tests/golden/blocksyntaxtest.yml = test/blocksyntaxtest.yml: a
    cube, two spheres and a rhombus): a 64^3 grid over the unit box holds the
    analytic number of hydrogen atoms within 0.003 (assert_values_equal_rel)."""
    import struct
    text = bench_text("lexingtonHII40.param")
    text = text.replace("anchor: [-5. pc, -5. pc, -5. pc]",
                        "anchor: [0. m, 0. m, 0. m]")
    text = text.replace("sides: [10. pc, 10. pc, 10. pc]",
                        "sides: [1. m, 1. m, 1. m]")
    text = text.replace("filename: lexingtonHII40.yml",
                        "filename: " + os.path.join(ROOT, "tests", "golden",
                                                    "blocksyntaxtest.yml"))
    text = text.replace("type: Gadget\n", "type: Binary\n")
    assert "[64, 64, 64]" in text and "anchor: [0. m" in text
    assert "blocksyntaxtest.yml" in text and "type: Binary" in text
    (tmp_path / "run.param").write_text(text)
    r = subprocess.run([exe, "--params", "run.param", "--dry-run",
                        "--dry-run-snapshot"], capture_output=True, text=True,
                       cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    blob = open(tmp_path / "lexingtonHII40_000.bin", "rb").read()
    assert struct.unpack_from("<3q", blob) == (64, 64, 64)
    fields = np.frombuffer(blob, dtype="<f8", offset=24).reshape(16, -1)
    total = fields[0].sum() / 64. ** 3
    expect = (4. * np.pi * 0.25 ** 3 / 3 + 4. * np.pi * 0.125 ** 3 / 3. +
              4. * 0.125 ** 3 / 3.)
    # assert_values_equal_rel(a, b, 0.003), test/Assert.hpp:63-68
    assert abs(total - expect) <= 0.003 * abs(total + expect)
    # the blocks' temperatures: 0 K outside, 100 K in the big sphere and the
    # rhombus, 200 K in the inner sphere
    assert set(np.unique(fields[1])) == {0., 100., 200.}


def test_ascii_file_source_fixture_of_the_reference(exe, tmp_path):
    """test/testAsciiFilePhotonSourceDistribution.cpp:30-49 with the
    reference's own file (tests/golden/
    test_asciifilephotonsourcedistribution.yml): three sources, 2.4e49 s^-1
    in total, weights that sum to one, the first source at the origin."""
    text = bench_text("stromgren.param")
    old = ("PhotonSourceDistribution:\n  type: SingleStar\n"
           "  position: [0. pc, 0. pc, 0. pc]\n  luminosity: 4.26e49 s^-1\n")
    assert old in text
    text = text.replace(
        old, "PhotonSourceDistribution:\n  type: AsciiFile\n  filename: " +
        os.path.join(ROOT, "tests", "golden",
                     "test_asciifilephotonsourcedistribution.yml") + "\n")
    (tmp_path / "run.param").write_text(text)
    d = describe(exe, str(tmp_path / "run.param"), str(tmp_path))
    assert len(d["sources"]) == 3
    assert d["total_luminosity"] == 2.4e49
    assert abs(sum(s["weight"] for s in d["sources"]) - 1.) <= 1e-14
    assert d["sources"][0]["position"] == [0., 0., 0.]
    kpc = 3.086e19
    assert np.allclose(d["sources"][1]["position"],
                       [0.5 * kpc, 0.2 * kpc, 0.4 * kpc], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("bench", ["stromgren_diffuse.param",
                                   "lexingtonHII40.param"])
def test_cmi_gpu_executable_with_copies_of_the_source_block(exe, tmp_path,
                                                            bench):
    """--copies: a star in the middle of ONE octant, so that every packet
    starts in one block of a 2x2x2 decomposition. Three engines hold that
    block (DensitySubGridCreator::create_copies,
    src/DensitySubGridCreator.hpp:437-531): they share its packets by packet
    id, their integrals are summed into each of them after the transport
    (update_original_counters, :556-574) and each solves the block's cells.
    The snapshots equal those of the undivided run."""
    text = bench_text(bench)
    text = text.replace("[64, 64, 64]", "[18, 18, 18]")
    for old in ("number of photons: 1e6", "number of photons: 1e8"):
        text = text.replace(old, "number of photons: 20000")
    text = text.replace("number of iterations: 20", "number of iterations: 5")
    text = text.replace("type: Gadget", "type: AsciiFile")
    assert "position: [0. pc, 0. pc, 0. pc]" in text
    text = text.replace("position: [0. pc, 0. pc, 0. pc]",
                        "position: [1.3 pc, -1.2 pc, 1.1 pc]")
    outputs = {}
    for label, extra in (("whole", []),
                         ("copies", ["--blocks", "2,2,2", "--copies", "3"])):
        d = tmp_path / label
        d.mkdir()
        if bench.startswith("lexington"):
            import shutil
            shutil.copy(os.path.join(BENCH, "lexingtonHII40.yml"), d)
        p = d / "run.param"
        p.write_text(text)
        r = subprocess.run([exe, "--params", str(p), "--output-statistics"] +
                           extra, capture_output=True, text=True, cwd=str(d))
        assert r.returncode == 0, r.stderr
        if extra:
            assert "8 blocks and 2 copies of source blocks" in r.stdout
        snapshots = sorted(f for f in os.listdir(d) if f.endswith("005.txt"))
        assert len(snapshots) == 1, os.listdir(d)
        outputs[label] = (np.loadtxt(d / snapshots[0]), r.stdout)
    whole, copies = outputs["whole"][0], outputs["copies"][0]
    assert np.array_equal(whole[:, :5], copies[:, :5])
    rel = np.abs(whole[:, 5] - copies[:, 5]) / whole[:, 5]
    assert np.median(rel) < 1e-5
    assert (rel < 1e-2).mean() > 0.97, (rel > 1e-2).sum()
    stats = [[l for l in out.splitlines() if "Escape fraction" in l][-1]
             for _, out in outputs.values()]
    assert stats[0] == stats[1]


@pytest.mark.gpu
def test_copies_cascade_to_the_neighbours_of_the_source_block(exe, tmp_path,
                                                              oracle):
    """An off-centre star on a grid of 4 x 1 x 1 blocks with --copies 4: the
    block with the star gets 4 engines, the blocks next to it 2 each (the
    reference's restriction of the copy levels to one level per neighbour,
    src/TaskBasedIonizationSimulation.cpp:533-556 and
    src/DensitySubGridCreator.hpp:437-531), the far block 1: nine engines,
    flights routed to the copies of the block they enter by packet id.
    Against the oracle on the undivided grid, same seed."""
    text = bench_text("stromgren_diffuse.param")
    text = text.replace("[64, 64, 64]", "[24, 12, 12]")
    text = text.replace("number of photons: 1e6", "number of photons: 30000")
    text = text.replace("number of iterations: 20", "number of iterations: 4")
    text = text.replace("type: Gadget", "type: AsciiFile")
    assert "position: [0. pc, 0. pc, 0. pc]" in text
    # in the second of the four blocks along x
    text = text.replace("position: [0. pc, 0. pc, 0. pc]",
                        "position: [-1.1 pc, 0.3 pc, -0.2 pc]")
    p = tmp_path / "run.param"
    p.write_text(text)
    r = subprocess.run([exe, "--params", str(p), "--output-statistics",
                        "--blocks", "4,1,1", "--copies", "4"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert ("4 blocks and 5 copies of source blocks (2 of them of their "
            "neighbours)") in r.stdout, r.stdout
    snapshots = sorted(f for f in os.listdir(tmp_path) if f.endswith("004.txt"))
    assert len(snapshots) == 1, os.listdir(tmp_path)
    last = np.loadtxt(tmp_path / snapshots[0])
    d = describe(exe, str(p), str(tmp_path))
    sim = oracle.OracleSimulation((24, 12, 12), d["anchor"], d["sides"])
    sim.set_sources([d["sources"][0]["position"]], [1.],
                    d["total_luminosity"])
    sim.set_homogeneous(100. * (1. / 0.01 / 0.01 / 0.01), 8000.)
    m = sim.model
    m.spectrum_type = oracle.SPECTRUM_MONOCHROMATIC
    m.mono_frequency = d["spectrum"]["frequency"]
    m.xsec_type = oracle.XSEC_FIXED
    m.recomb_type = oracle.RECOMB_FIXED
    m.reemit_type = oracle.REEMIT_PHYSICAL
    m.do_temperature = 0
    for i in range(14):
        m.xsec_fixed[i] = d["cross_sections"][i]
        m.recomb_fixed[i] = d["recombination_rates"][i]
    sim.run(30000, 4, seed=42)
    x = np.asarray(sim.x[0])
    assert 0.02 < (x < 0.5).mean() < 0.98
    # (equal seeds: the differences are rounding, amplified through four
    # iterations of the ionization balance)
    rel = np.abs(last[:, 5] - x) / x
    assert np.median(rel) < 1e-5
    assert (rel < 1e-2).mean() > 0.97, (rel > 1e-2).sum()


@pytest.mark.gpu
def test_cmi_gpu_executable_with_a_continuous_source(exe, tmp_path, oracle):
    """stromgren.param plus `ContinuousPhotonSource: type: Isotropic` with a
    monochromatic ContinuousPhotonSourceSpectrum of given total flux: the host
    computes the continuous luminosity as surface area x flux
    (src/PhotonSource.cpp:104-111), half of the packets come from either kind
    of source; result equal to the oracle with the same mix."""
    text = bench_text("stromgren.param")
    text = text.replace("[64, 64, 64]", "[16, 16, 16]")
    text = text.replace("number of photons: 1e6", "number of photons: 20000")
    text = text.replace("number of iterations: 20", "number of iterations: 3")
    text = text.replace("type: Gadget", "type: AsciiFile")
    assert "ContinuousPhotonSource" not in text
    flux = 2.e14  # m^-2 s^-1: about 2.7 times the star over the 10 pc box
    text += ("\nContinuousPhotonSource:\n  type: Isotropic\n"
             "\nContinuousPhotonSourceSpectrum:\n  type: Monochromatic\n"
             "  frequency: 3.6e15 Hz\n  total flux: %g m^-2 s^-1\n" % flux)
    p = tmp_path / "small.param"
    p.write_text(text)
    r = subprocess.run([exe, "--params", str(p), "--output-statistics"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    last = np.loadtxt(tmp_path / "stromgren_003.txt")
    d = describe(exe, str(tmp_path / "small.param"), str(tmp_path))
    sim = oracle.OracleSimulation((16,) * 3, d["anchor"], d["sides"])
    sim.set_sources([[0., 0., 0.]], [1.], d["total_luminosity"])
    sim.set_homogeneous(100. * (1. / 0.01 / 0.01 / 0.01), 8000.)
    m = sim.model
    m.spectrum_type = oracle.SPECTRUM_MONOCHROMATIC
    m.mono_frequency = d["spectrum"]["frequency"]
    m.xsec_type = oracle.XSEC_FIXED
    m.recomb_type = oracle.RECOMB_FIXED
    for i in range(14):
        m.xsec_fixed[i] = d["cross_sections"][i]
        m.recomb_fixed[i] = d["recombination_rates"][i]
    area = 6. * d["sides"][0] ** 2
    sim.set_continuous_source(area * flux, frequency=3.6e15)
    assert 2. < m.continuous_photon_weight < 3.5
    sim.run(20000, 3, seed=42)
    assert np.allclose(last[:, 5], sim.x[0], rtol=2e-3, atol=0.)
    # the background keeps the corners of the box ionized too
    assert last[0, 5] < 0.5


def test_ascii_file_source_distribution(exe, tmp_path):
    """PhotonSourceDistribution type AsciiFile
    (src/AsciiFilePhotonSourceDistribution.hpp:45-120): several stars from a
    YAML file; weights = luminosities over their sum."""
    text = bench_text("stromgren.param")
    old = ("PhotonSourceDistribution:\n  type: SingleStar\n"
           "  position: [0. pc, 0. pc, 0. pc]\n  luminosity: 4.26e49 s^-1\n")
    assert old in text
    text = text.replace(old, "PhotonSourceDistribution:\n  type: AsciiFile\n"
                             "  filename: stars.yml\n")
    (tmp_path / "stars.yml").write_text(
        "number of sources: 3\n"
        "source[0]:\n  position: [1. pc, 0. pc, 0. pc]\n"
        "  luminosity: 1.e49 s^-1\n"
        "source[1]:\n  position: [-2. pc, 1. pc, 0.5 pc]\n"
        "  luminosity: 3.e49 s^-1\n"
        "source[2]:\n  position: [0. pc, -3. pc, 2. pc]\n"
        "  luminosity: 4.e49 s^-1\n")
    p = tmp_path / "multi.param"
    p.write_text(text)
    d = describe(exe, str(p), str(tmp_path))
    pc = 3.086e16
    assert abs(d["total_luminosity"] - 8.e49) <= 1e-15 * 8.e49
    assert np.allclose([s["weight"] for s in d["sources"]],
                       [0.125, 0.375, 0.5], rtol=1e-15)
    assert np.allclose([s["position"] for s in d["sources"]],
                       [[pc, 0., 0.], [-2. * pc, pc, 0.5 * pc],
                        [0., -3. * pc, 2. * pc]], rtol=1e-12)
    assert os.path.exists(tmp_path / "stars.yml.used-values")
    # a file without the count is an error message, not an abort
    (tmp_path / "stars.yml").write_text("source[0]:\n  luminosity: 1 s^-1\n")
    r = subprocess.run([exe, "--params", str(p), "--dry-run"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "number of sources" in r.stderr


def tracker_run(exe, tmp_path, label, blocks, hdf5=False, block_file=None):
    """lexingtonHII40.param at 16^3 with trackers in the last iteration"""
    import shutil
    text = bench_text("lexingtonHII40.param")
    text = text.replace("[64, 64, 64]", "[16, 16, 16]")
    text = text.replace("number of photons: 1e8", "number of photons: 30000")
    text = text.replace("number of iterations: 20", "number of iterations: 4")
    text = text.replace("type: Gadget", "type: AsciiFile")
    text = text.replace("random seed: 42",
                        "random seed: 42\n  enable trackers: true")
    text += ("\nTrackerManager:\n  filename: trackers.yml\n"
             "  minimum number of photon packets: 50000\n")
    if hdf5:
        text += "  HDF5 output: true\n  HDF5 output name: absorbed.hdf5\n"
    assert "enable trackers: true" in text
    d = tmp_path / label
    d.mkdir()
    if block_file is not None:
        (d / "trackers.yml").write_text(block_file)
    elif hdf5:
        (d / "trackers.yml").write_text(
            "number of trackers: 2\n"
            "tracker[0]:\n"
            "  position: [1.3 pc, 0.4 pc, -0.7 pc]\n"
            "  type: Absorption\n"
            "tracker[1]:\n"
            "  position: [-2.1 pc, 1.9 pc, 0.2 pc]\n"
            "  type: Absorption\n"
            "  output name: far side\n")
    else:
        (d / "trackers.yml").write_text(
            "number of trackers: 3\n"
            "tracker[0]:\n"
            "  position: [1.3 pc, 0.4 pc, -0.7 pc]\n"
            "  type: Spectrum\n"
            "  number of bins: 50\n"
            "tracker[1]:\n"
            "  position: [-2.1 pc, 1.9 pc, 0.2 pc]\n"
            "  number of bins: 50\n"
            "  opening angle: 60. degrees\n"
            "  reference direction: [-1., 1., 0.]\n"
            "  output name: second.txt\n"
            "tracker[2]:\n"
            "  position: [1.3 pc, 0.4 pc, -0.7 pc]\n"
            "  type: Absorption\n")
    shutil.copy(os.path.join(BENCH, "lexingtonHII40.yml"), d)
    (d / "run.param").write_text(text)
    r = subprocess.run([exe, "--params", "run.param"] +
                       (["--blocks", blocks] if blocks else []),
                       capture_output=True, text=True, cwd=str(d))
    assert r.returncode == 0, r.stderr
    # the last iteration used the trackers' packet count
    assert "Start shooting 50000 photons" in r.stdout
    assert r.stdout.count("Start shooting 30000 photons") == 3
    assert os.path.exists(d / "trackers.yml.used-values")
    return d


def read_absorption(path):
    lines = open(path).read().splitlines()
    assert lines[0] == ("# Ion \tsource photon\tdiffuse H photon\t"
                        "diffuse He photon\tabsorbed photon")
    names = [l.split("\t")[0] for l in lines[1:]]
    values = np.array([[float(v) for v in l.split("\t")[1:]]
                       for l in lines[1:]])
    return names, values


@pytest.mark.gpu
def test_trackers_through_the_driver(exe, tmp_path):
    """IonizationSimulation:enable trackers + a TrackerManager block file
    (src/TrackerManager.hpp, src/SpectrumTracker.hpp,
    src/AbsorptionTracker.hpp): the trackers count in the last iteration and
    are written as the reference's text files - on the undivided grid and on a
    grid in blocks (every block counts the trackers in its cells, the driver
    merges them)."""
    whole = tracker_run(exe, tmp_path, "whole", None)
    first = open(whole / "Tracker0.txt").read().splitlines()
    assert first[0] == ("# frequency (Hz)\tprimary count\tdiffuse H count\t"
                        "diffuse He count")
    a = np.loadtxt(whole / "Tracker0.txt")
    b = np.loadtxt(whole / "second.txt")
    assert a.shape == b.shape == (50, 4)
    width = 3. * 3.289e15 / 50
    assert np.allclose(a[:, 0], 3.289e15 + (np.arange(50) + 0.5) * width,
                       rtol=1e-5)
    # a 40 000 K star: primaries fall off with frequency; the diffuse hydrogen
    # photons sit just above the threshold
    assert a[:, 1].sum() > 100 and a[:5, 1].sum() > a[-20:, 1].sum()
    assert a[:, 2].sum() > 10 and a[:3, 2].sum() >= 0.8 * a[:, 2].sum()
    # the cone around the outward direction sees the star's light, not all of
    # the diffuse field
    assert 0 < b[:, 1:].sum() < a[:, 1:].sum() * 3
    # the AbsorptionTracker of the first tracker's cell (normalised: x
    # luminosity / total weight): ion names down, photon types across
    names, absorbed = read_absorption(whole / "Tracker2.txt")
    # (get_ion_name, src/ElementNames.hpp:208-258)
    assert names[:4] == ["H", "He", "C+", "C++"] and len(names) == 14
    assert absorbed.shape == (14, 4)
    assert absorbed[0, 0] > 0. and absorbed[0, 1] > 0.
    assert not absorbed[:, 3].any()
    # hydrogen absorbs source photons over more of the spectrum than He does
    assert absorbed[0, 0] > absorbed[1, 0] > 0.
    # the same run on 2 x 2 x 1 blocks
    blocks = tracker_run(exe, tmp_path, "blocks", "2,2,1")
    a2 = np.loadtxt(blocks / "Tracker0.txt")
    b2 = np.loadtxt(blocks / "second.txt")
    # (the blocks fly in the incremental marcher, the undivided grid in the
    # exact one while trackers count: the same cells except on corner ties,
    # and the states the two runs reached in three iterations differ by
    # rounding)
    assert np.abs(a2[:, 1:] - a[:, 1:]).max() <= 2
    assert np.abs(b2[:, 1:] - b[:, 1:]).max() <= 2
    names2, absorbed2 = read_absorption(blocks / "Tracker2.txt")
    assert names2 == names
    assert np.allclose(absorbed2, absorbed, rtol=1e-3,
                       atol=1e-6 * absorbed.max())


@pytest.mark.gpu
def test_absorption_trackers_as_one_hdf5_file(exe, tmp_path):
    """TrackerManager:HDF5 output (src/TrackerManager.hpp:330-367 with
    AbsorptionTracker::create_group / append_to_group,
    src/AbsorptionTracker.hpp:182-223): one group per tracker type with its
    shared datasets; equal to the text files of the same run."""
    import hdf5_mini
    d = tracker_run(exe, tmp_path, "hdf5", None, hdf5=True)
    f = hdf5_mini.read(str(d / "absorbed.hdf5"))
    g = f["/Group0"]
    assert g.attrs["type"] == "Absorption"
    assert g.attrs["position unit"] == "m"
    assert f["/Group0/ion name"].data[:3] == ["H", "He", "C+"]
    assert len(f["/Group0/ion name"].data) == 14
    assert f["/Group0/tracker labels"].data == ["Tracker0", "far side"]
    pc = 3.086e16
    assert np.allclose(f["/Group0/positions"].data,
                       [[1.3 * pc, 0.4 * pc, -0.7 * pc],
                        [-2.1 * pc, 1.9 * pc, 0.2 * pc]], rtol=1e-12)
    t = tracker_run(exe, tmp_path, "text", None)
    _, absorbed = read_absorption(t / "Tracker2.txt")
    for column, name in enumerate(("source photon", "diffuse H photon",
                                   "diffuse He photon", "absorbed photon")):
        table = np.asarray(f["/Group0/" + name + " absorption"].data)
        assert table.shape == (2, 14)
        # (text files hold 6 significant digits)
        assert np.allclose(table[0], absorbed[:, column], rtol=1e-5, atol=0.)
    assert not os.path.exists(d / "Tracker0.txt")


WEIGHTED_BLOCK_FILE = (
    "number of trackers: 4\n"
    "tracker[0]:\n"
    "  position: [1.3 pc, 0.4 pc, -0.7 pc]\n"
    "  type: WeightedSpectrum\n"
    "tracker[1]:\n"
    "  position: [1.3 pc, 0.4 pc, -0.7 pc]\n"
    "  type: WeightedSpectrum\n"
    "  output name: levels\n"
    "  FrequencyBins:\n"
    "    type: Level\n"
    "tracker[2]:\n"
    "  position: [-2.1 pc, 1.9 pc, 0.2 pc]\n"
    "  type: WeightedSpectrum\n"
    "  FrequencyBins:\n"
    "    type: Linear\n"
    "    number of bins: 20\n"
    "    minimum frequency: 13.6 eV\n"
    "    maximum frequency: 24.6 eV\n"
    "tracker[3]:\n"
    "  position: [-2.1 pc, 1.9 pc, 0.2 pc]\n"
    "  type: WeightedSpectrum\n"
    "  output name: far levels\n"
    "  FrequencyBins:\n"
    "    type: Level\n")


def test_weighted_spectrum_block_file_is_parsed(exe, tmp_path):
    """TrackerFactory's "WeightedSpectrum" (src/TrackerFactory.hpp:60-77) with
    its FrequencyBins block (src/FrequencyBinsFactory.hpp:57-72,
    src/LinearFrequencyBins.hpp:80-88 for the defaults). Dry run: no GPU."""
    (tmp_path / "weighted.yml").write_text(WEIGHTED_BLOCK_FILE)
    multi_tracker_param(tmp_path, "weighted.yml")
    run = [exe, "--params", "run.param", "--dry-run"]
    r = subprocess.run(run, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    used = open(tmp_path / "weighted.yml.used-values").read()
    assert used.count("type: WeightedSpectrum") == 4
    assert used.count("type: Level") == 2
    # the first tracker's bins are all defaults, the third's are given
    assert "number of bins: 100" in used and "number of bins: 20" in used
    # (used values are written in SI units: 13.6, 54.4 and 24.6 eV in Hz)
    assert "minimum frequency: 3.28847e+15 Hz # (default value)" in used
    assert "maximum frequency: 1.31539e+16 Hz # (default value)" in used
    assert "maximum frequency: 5.94825e+15 Hz # (24.6 eV)" in used
    (tmp_path / "weighted.yml").write_text(
        WEIGHTED_BLOCK_FILE.replace("type: Level", "type: Logarithmic", 1))
    r = subprocess.run(run, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode != 0
    assert 'Unknown FrequencyBins type: "Logarithmic".' in r.stderr
    # HDF5 output: weighted trackers have a form, Spectrum trackers do not
    text = open(tmp_path / "run.param").read()
    (tmp_path / "run.param").write_text(text + "  HDF5 output: true\n")
    (tmp_path / "weighted.yml").write_text(WEIGHTED_BLOCK_FILE)
    r = subprocess.run(run, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    (tmp_path / "weighted.yml").write_text(
        WEIGHTED_BLOCK_FILE.replace("type: WeightedSpectrum",
                                    "type: Spectrum", 1))
    r = subprocess.run(run, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode != 0 and "HDF5 form" in r.stderr


@pytest.mark.gpu
def test_weighted_spectrum_trackers_through_the_driver(exe, tmp_path):
    """WeightedSpectrum trackers (src/WeightedSpectrumTracker.hpp) with the
    two kinds of FrequencyBins (src/FrequencyBinsFactory.hpp:57-72) as text
    files (output_tracker, :325-341) and as groups of one HDF5 file
    (create_group / append_to_group, :366-428; trackers with the same bins
    share a group, src/TrackerManager.hpp:141-161): fluxes per area, the same
    numbers in both forms."""
    import hdf5_mini
    d = tracker_run(exe, tmp_path, "text", None, block_file=WEIGHTED_BLOCK_FILE)
    head = open(d / "Tracker0.txt").read().splitlines()[0]
    assert head == ("# frequency (Hz)\tsource photon flux (s^-1 m^-2)\t"
                    "diffuse H photon flux (s^-1 m^-2)\t"
                    "diffuse He photon flux (s^-1 m^-2)\t"
                    "absorbed photon flux (s^-1 m^-2)")
    a = np.loadtxt(d / "Tracker0.txt")
    lv = np.loadtxt(d / "levels")
    c = np.loadtxt(d / "Tracker2.txt")
    assert a.shape == (100, 5) and lv.shape == (14, 5) and c.shape == (20, 5)
    ev = 1.6021766208e-19 / 6.626070040e-34
    width = (54.4 - 13.6) * ev / 100
    assert np.allclose(a[:, 0], 13.6 * ev + (np.arange(100) + 0.5) * width,
                       rtol=1e-5)
    # LevelFrequencyBins::get_frequency: the ionization energies, ascending
    assert np.allclose(lv[:3, 0], [3.28810e15, 3.29285e15, 3.51436e15],
                       rtol=1e-5)
    assert np.all(np.diff(lv[:, 0]) > 0.)
    # the same crossings in both trackers of the first cell
    assert np.allclose(a[:, 1:].sum(axis=0), lv[:, 1:].sum(axis=0), rtol=1e-4)
    assert a[:, 1].sum() > 0. and a[:, 2].sum() > 0. and not a[:, 4].any()
    # a flux: crossings / projected area x luminosity / total weight / cell
    # side^2 - of the order of Q / (4 pi r^2) at r = 1.5 pc from a star of
    # 4.26e49 photons per second (the cell's midpoint lies at 1.85 pc, and
    # some of the light has been absorbed on the way)
    pc = 3.086e16
    r2 = (1.3 ** 2 + 0.4 ** 2 + 0.7 ** 2) * pc * pc
    geometric = 4.26e49 / (4. * np.pi * r2)
    assert 0.3 * geometric < a[:, 1].sum() < 1.0 * geometric
    # the narrow range collects the harder photons in its last bin
    assert c[-1, 1] > c[-2, 1]
    # the same on 2 x 2 x 1 blocks: every block counts the trackers in its
    # cells (TRACK builds of the incremental marcher), the driver merges them
    # (the two runs' states differ by rounding after three iterations, and the
    # undivided grid counts in the exact marcher: the same cells except on
    # corner ties)
    b = tracker_run(exe, tmp_path, "blocks", "2,2,1",
                    block_file=WEIGHTED_BLOCK_FILE)
    a2 = np.loadtxt(b / "Tracker0.txt")
    lv2 = np.loadtxt(b / "levels")
    assert np.allclose(a2[:, 1:].sum(axis=0), a[:, 1:].sum(axis=0), rtol=2e-2)
    assert np.allclose(lv2[:, 1], lv[:, 1], rtol=5e-2,
                       atol=2e-3 * lv[:, 1].max())
    h = tracker_run(exe, tmp_path, "hdf5", None, hdf5=True,
                    block_file=WEIGHTED_BLOCK_FILE)
    f = hdf5_mini.read(str(h / "absorbed.hdf5"))
    # groups in the order of their first trackers: [0], [1, 3], [2]
    g0, g1, g2 = f["/Group0"], f["/Group1"], f["/Group2"]
    for g in (g0, g1, g2):
        assert g.attrs["type"] == "WeightedSpectrum"
        assert g.attrs["frequency unit"] == "s^-1"
        assert g.attrs["flux unit"] == "m^-2 s^-1"
        assert g.attrs["position unit"] == "m"
    assert f["/Group0/tracker labels"].data == ["Tracker0"]
    assert f["/Group1/tracker labels"].data == ["levels", "far levels"]
    assert f["/Group2/tracker labels"].data == ["Tracker2"]
    # (get_ion_name of the ions in the order of their ionization energies)
    assert f["/Group1/bin labels"].data[:4] == ["H", "O", "N", "Ne"]
    assert len(f["/Group1/bin labels"].data) == 14
    assert np.allclose(f["/Group1/positions"].data,
                       [[1.3 * pc, 0.4 * pc, -0.7 * pc],
                        [-2.1 * pc, 1.9 * pc, 0.2 * pc]], rtol=1e-12)
    assert np.allclose(f["/Group0/frequencies"].data, a[:, 0], rtol=1e-5)
    assert np.allclose(f["/Group1/frequencies"].data, lv[:, 0], rtol=1e-5)
    for column, name in enumerate(("source photon", "diffuse H photon",
                                   "diffuse He photon", "absorbed photon")):
        t0 = np.asarray(f["/Group0/" + name + " flux"].data)
        t1 = np.asarray(f["/Group1/" + name + " flux"].data)
        t2 = np.asarray(f["/Group2/" + name + " flux"].data)
        assert t0.shape == (1, 100) and t1.shape == (2, 14)
        assert t2.shape == (1, 20)
        # (text files hold 6 significant digits)
        assert np.allclose(t0[0], a[:, 1 + column], rtol=1e-5, atol=0.)
        assert np.allclose(t1[0], lv[:, 1 + column], rtol=1e-5, atol=0.)
        assert np.allclose(t2[0], c[:, 1 + column], rtol=1e-5, atol=0.)
    assert np.allclose(np.asarray(f["/Group1/source photon flux"].data)[1],
                       np.loadtxt(d / "far levels")[:, 1], rtol=1e-5)


@pytest.mark.gpu
def test_tracker_manager_fixture_of_the_reference(exe, tmp_path):
    """test/testTrackerManager.cpp:30-56 with the reference's own block file
    (tests/golden/test_tracker_manager.yml = test/test_tracker_manager.yml:
    three Spectrum trackers with 100, 1000 and 100 bins, one with its own
    output name) on the box of that test (10 pc around the origin, 64^3):
    all three are placed and written, each with its own number of bins."""
    import shutil
    text = bench_text("stromgren.param")
    text = text.replace("number of photons: 1e6", "number of photons: 100000")
    text = text.replace("number of iterations: 20", "number of iterations: 2")
    text = text.replace("type: Gadget", "type: AsciiFile")
    assert "IonizationSimulation:" in text
    text = text.replace("IonizationSimulation:",
                        "IonizationSimulation:\n  enable trackers: true")
    text += ("\nTrackerManager:\n  filename: test_tracker_manager.yml\n"
             "  minimum number of photon packets: 99\n")
    shutil.copy(os.path.join(ROOT, "tests", "golden",
                             "test_tracker_manager.yml"), tmp_path)
    (tmp_path / "run.param").write_text(text)
    r = subprocess.run([exe, "--params", "run.param"], capture_output=True,
                       text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    a = np.loadtxt(tmp_path / "Tracker0.txt")
    b = np.loadtxt(tmp_path / "Tracker1.txt")
    c = np.loadtxt(tmp_path / "special_position.txt")
    assert a.shape == (100, 4) and b.shape == (1000, 4) and c.shape == (100, 4)
    # a monochromatic 13.6 eV source: every count sits in the bin of that
    # frequency; the star's own cell sees every packet, 4 pc away fewer
    nu = 13.6 * 1.6021766208e-19 / 6.626070040e-34
    for table, nbins in ((a, 100), (b, 1000), (c, 100)):
        k = int((nu - 3.289e15) / (3. * 3.289e15 / nbins))
        assert table[:, 1].sum() == table[k, 1] > 0
        assert not table[:, 2:].any()
    assert a[:, 1].sum() >= 100000 > b[:, 1].sum() > 0
    assert abs(b[:, 1].sum() - c[:, 1].sum()) < 0.5 * b[:, 1].sum()


def test_tracker_block_file_is_parsed(exe, tmp_path):
    """TrackerManager's block file (src/TrackerManager.hpp:98-170): parsed
    with the parameter file grammar, used values written back, unknown types
    and missing keys reported (no GPU needed: dry run)."""
    text = bench_text("stromgren.param")
    text = text.replace("random seed: 42",
                        "random seed: 42\n  enable trackers: true")
    if "enable trackers" not in text:
        text = text.replace("IonizationSimulation:",
                            "IonizationSimulation:\n  enable trackers: true")
    text += "\nTrackerManager:\n  filename: trackers.yml\n"
    (tmp_path / "run.param").write_text(text)

    def dry_run():
        return subprocess.run([exe, "--params", "run.param", "--dry-run"],
                              capture_output=True, text=True,
                              cwd=str(tmp_path))
    (tmp_path / "trackers.yml").write_text(
        "number of trackers: 1\ntracker[0]:\n"
        "  position: [1. pc, 0.5 pc, -2. pc]\n"
        "  opening angle: 45. degrees\n")
    r = dry_run()
    assert r.returncode == 0, r.stderr
    used = open(tmp_path / "trackers.yml.used-values").read()
    assert "number of bins: 100" in used and "Tracker0.txt" in used
    assert "0.785398" in used  # 45 degrees in radians
    (tmp_path / "trackers.yml").write_text(
        "number of trackers: 1\ntracker[0]:\n"
        "  position: [1. pc, 0.5 pc, -2. pc]\n  type: Absorption\n")
    r = dry_run()
    assert r.returncode == 0, r.stderr
    (tmp_path / "trackers.yml").write_text(
        "number of trackers: 1\ntracker[0]:\n"
        "  position: [1. pc, 0.5 pc, -2. pc]\n  type: Polarization\n")
    r = dry_run()
    assert r.returncode != 0 and "Polarization" in r.stderr
    # HDF5 output exists for Absorption trackers only (src/Tracker.hpp:112-130)
    (tmp_path / "trackers.yml").write_text(
        "number of trackers: 1\ntracker[0]:\n"
        "  position: [1. pc, 0.5 pc, -2. pc]\n")
    (tmp_path / "run.param").write_text(text + "  HDF5 output: true\n")
    r = dry_run()
    assert r.returncode != 0 and "Absorption" in r.stderr
    (tmp_path / "run.param").write_text(text)
    (tmp_path / "trackers.yml").write_text("tracker[0]:\n  type: Spectrum\n")
    r = dry_run()
    assert r.returncode != 0 and "number of trackers" in r.stderr


def multi_tracker_param(tmp_path, block_file):
    text = bench_text("stromgren_diffuse.param")
    text = text.replace("number of photons: 1e6", "number of photons: 100000")
    text = text.replace("number of iterations: 20", "number of iterations: 2")
    text = text.replace("[64, 64, 64]", "[16, 16, 16]")
    text = text.replace("type: Gadget", "type: AsciiFile")
    assert "IonizationSimulation:" in text
    text = text.replace("IonizationSimulation:",
                        "IonizationSimulation:\n  enable trackers: true")
    text += "\nTrackerManager:\n  filename: %s\n" % block_file
    (tmp_path / "run.param").write_text(text)


def test_multi_tracker_fixture_is_parsed(exe, tmp_path):
    """test/testMultiTracker.cpp:36-44 with the reference's block file
    (tests/golden/test_multi_tracker.yml = test/test_multi_tracker.yml): its
    tracker[0] is a Multi tracker of two Spectrum trackers. (The file
    announces three trackers and holds one - the reference's test builds only
    `tracker[0]:` -, so the manager is given its first line as 1.) Dry run:
    no GPU."""
    fixture = open(os.path.join(ROOT, "tests", "golden",
                                "test_multi_tracker.yml")).read()
    assert fixture.startswith("number of trackers: 3")
    (tmp_path / "multi.yml").write_text(
        fixture.replace("number of trackers: 3", "number of trackers: 1", 1))
    multi_tracker_param(tmp_path, "multi.yml")
    r = subprocess.run([exe, "--params", "run.param", "--dry-run"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    used = open(tmp_path / "multi.yml.used-values").read()
    assert used.count("type: Spectrum") == 2 and "type: Multi" in used
    assert used.count("number of bins: 100") == 2
    # as it stands the file is short of two trackers: reported, like the
    # reference's YAMLDictionary does for a missing key
    (tmp_path / "multi.yml").write_text(fixture)
    r = subprocess.run([exe, "--params", "run.param", "--dry-run"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode != 0 and "tracker[1]:position" in r.stderr
    # a Multi tracker without its count
    (tmp_path / "multi.yml").write_text(
        "number of trackers: 1\ntracker[0]:\n  type: Multi\n"
        "  position: [0. pc, 0. pc, 0. pc]\n")
    r = subprocess.run([exe, "--params", "run.param", "--dry-run"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode != 0 and "number of trackers" in r.stderr


@pytest.mark.gpu
def test_multi_tracker_counts_equal_separate_trackers(exe, tmp_path):
    """MultiTracker (src/MultiTracker.hpp, src/MultiTracker.cpp): every
    tracker of a Multi tracker counts every photon of the cell
    (count_photon, :104-124), and its output is one file per tracker plus a
    table of contents (output_tracker, :127-143). The same three trackers
    once as one Multi tracker and once as three trackers of their own at the
    same position: identical files, on an undivided grid and in blocks."""
    # (key: value lines of each tracker, without indentation)
    leaves = [
        ["type: Spectrum", "number of bins: 50"],
        ["type: Spectrum", "opening angle: 60. degrees",
         "reference direction: [0., 0., 2.]"],
        ["type: Absorption"],
    ]
    position = "position: [1.1 pc, -0.4 pc, 0.7 pc]"
    multi = ("number of trackers: 2\n"
             "tracker[0]:\n  type: Multi\n  " + position +
             "\n  output name: cell.txt\n  number of trackers: 3\n")
    single = "number of trackers: 3\n"
    for i, leaf in enumerate(leaves):
        multi += "  tracker[%d]:\n" % i
        multi += "".join("    %s\n" % line for line in leaf)
        if i == 1:
            multi += "    output name: cone.txt\n"
        single += "tracker[%d]:\n  %s\n" % (i, position)
        single += "".join("  %s\n" % line for line in leaf)
    # a second, plain tracker after the Multi one keeps its own index
    multi += "tracker[1]:\n  position: [-2. pc, 2. pc, 0.3 pc]\n"
    (tmp_path / "multi.yml").write_text(multi)
    (tmp_path / "single.yml").write_text(single)
    for blocks in (None, "2,1,2"):
        for name in ("multi", "single"):
            d = tmp_path / ("%s_%s" % (name, "blocks" if blocks else "whole"))
            d.mkdir()
            multi_tracker_param(d, "../%s.yml" % name)
            cmd = [exe, "--params", "run.param"]
            if blocks:
                cmd += ["--blocks", blocks]
            r = subprocess.run(cmd, capture_output=True, text=True,
                               cwd=str(d))
            assert r.returncode == 0, r.stderr
        dm = tmp_path / ("multi_" + ("blocks" if blocks else "whole"))
        ds = tmp_path / ("single_" + ("blocks" if blocks else "whole"))
        pairs = (("cell.txt.0.txt", "Tracker0.txt"),
                 ("cone.txt", "Tracker1.txt"),
                 ("cell.txt.2.txt", "Tracker2.txt"))
        for a, b in pairs:
            ta, tb = np.loadtxt(dm / a, usecols=(1, 2, 3)), \
                np.loadtxt(ds / b, usecols=(1, 2, 3))
            assert np.array_equal(ta, tb), (blocks, a)
            assert ta.sum() > 0
        assert np.loadtxt(dm / "cell.txt.0.txt").shape == (50, 4)
        assert np.loadtxt(dm / "Tracker1.txt").shape == (100, 4)
        # the cone sees fewer photons than the whole sphere
        assert np.loadtxt(dm / "cone.txt")[:, 1:].sum() < \
            np.loadtxt(dm / "cell.txt.0.txt")[:, 1:].sum()
        toc = open(dm / "cell.txt").read()
        assert toc == (
            "tracker[0]:\n  output name: cell.txt.0.txt\n  type: Spectrum\n"
            "  number of bins: 50\n  opening angle: 3.14159 radians\n"
            "  reference direction: [0, 0, 0]\n"
            "tracker[1]:\n  output name: cone.txt\n  type: Spectrum\n"
            "  number of bins: 100\n  opening angle: 1.0472 radians\n"
            "  reference direction: [0, 0, 1]\n"
            "tracker[2]:\n  output name: cell.txt.2.txt\n"
            "  type: AbsorptionTracker\n"), toc


@pytest.mark.gpu
def test_copies_cascade_is_cut_back_to_the_group_limit(exe, tmp_path):
    """Two stars in different blocks of a 4 x 4 x 2 decomposition with
    --copies 8: the full cascade (8 engines per source block, 4 and 2 for the
    rings of neighbours) needs far more than the 64 engines of a group. The
    reference has no such limit, so the cascade is cut back - outermost rings
    first, then fewer copies per source block - instead of refusing the run;
    the result is the undivided grid's (copies only share work)."""
    text = bench_text("stromgren.param")
    text = text.replace("[64, 64, 64]", "[24, 24, 12]")
    text = text.replace("number of photons: 1e6", "number of photons: 40000")
    text = text.replace("number of iterations: 20", "number of iterations: 3")
    text = text.replace("type: Gadget", "type: AsciiFile")
    old = text[text.index("PhotonSourceDistribution:"):]
    old = old[:old.index("\n\n")]
    text = text.replace(old, "PhotonSourceDistribution:\n  type: AsciiFile\n"
                        "  filename: stars.yml")
    stars = ("number of sources: 2\n"
             "source[0]:\n  position: [-3.1 pc, -3.3 pc, -1.2 pc]\n"
             "  luminosity: 3.e49 s^-1\n"
             "source[1]:\n  position: [1.4 pc, 3.2 pc, 2.1 pc]\n"
             "  luminosity: 1.26e49 s^-1\n")
    runs = {}
    for label, extra in (("whole", []),
                         ("blocks", ["--blocks", "4,4,2", "--copies", "8"])):
        d = tmp_path / label
        d.mkdir()
        (d / "run.param").write_text(text)
        (d / "stars.yml").write_text(stars)
        r = subprocess.run([exe, "--params", "run.param",
                            "--output-statistics"] + extra,
                           capture_output=True, text=True, cwd=str(d))
        assert r.returncode == 0, r.stderr
        snapshots = sorted(f for f in os.listdir(d) if f.endswith("003.txt"))
        assert len(snapshots) == 1, os.listdir(d)
        runs[label] = (np.loadtxt(d / snapshots[0]), r.stdout)
    out = runs["blocks"][1]
    assert "8 engines per source block asked for" in out, out
    assert "granted" in out and "at most 64 engines" in out
    line = [l for l in out.splitlines() if "Domain decomposition" in l][0]
    # "Domain decomposition: 32 blocks and N copies of source blocks ..."
    words = line.split()
    nblocks, ncopies = int(words[2]), int(words[5])
    assert nblocks == 32 and 0 < ncopies <= 32
    whole, blocks = runs["whole"][0], runs["blocks"][0]
    x = whole[:, 5]
    assert 0.02 < (x < 0.5).mean() < 0.98
    rel = np.abs(blocks[:, 5] - x) / x
    assert np.median(rel) < 1e-5
    assert (rel < 1e-2).mean() > 0.97, (rel > 1e-2).sum()


@pytest.mark.gpu
@pytest.mark.parametrize("name,flags,prefix,ncell,nphoton,niter", [
    ("test_ionizationsimulation.param", [], "test_ionizationsimulation",
     32, 50000, 4),
    ("test_taskbasedionizationsimulation.param", ["--task-based"],
     "test_taskbasedionizationsimulation", 16, 100000, 10),
])
def test_reference_integration_inputs_run_unchanged(exe, tmp_path, oracle,
                                                    name, flags, prefix,
                                                    ncell, nphoton, niter):
    """test/testIonizationSimulation.cpp and
    test/testTaskBasedIonizationSimulation.cpp construct a simulation from
    their parameter file, initialize and run it - nothing else. The same two
    files (tests/golden/, byte for byte; the second holds two
    PhotonSourceSpectrum blocks, the later one counts) through the executable:
    it runs to the end, writes the reference's snapshots, and the final
    neutral fractions are the oracle's for the lowered values."""
    import shutil
    src = os.path.join(ROOT, "tests", "golden", name)
    p = tmp_path / name
    shutil.copy(src, p)
    r = subprocess.run([exe, "--params", str(p)] + flags,
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(src, "rb").read() == open(p, "rb").read()  # unchanged
    first = np.loadtxt(tmp_path / (prefix + "000.txt"))
    last = np.loadtxt(tmp_path / (prefix + "%03d.txt" % niter))
    n = ncell ** 3
    assert first.shape == (n, 6) and last.shape == (n, 6)
    d = describe(exe, str(p), str(tmp_path))
    assert d["number_of_photons"] == nphoton
    assert d["number_of_iterations"] == niter
    sim = oracle.OracleSimulation((ncell,) * 3, d["anchor"], d["sides"])
    sim.set_sources([[0., 0., 0.]], [1.], d["total_luminosity"])
    sim.set_homogeneous(100. * (1. / 0.01 / 0.01 / 0.01), 8000.)
    m = sim.model
    if d["spectrum"]["type"] == "Planck":
        m.spectrum_type = oracle.SPECTRUM_PLANCK
        m.planck_temperature = d["spectrum"]["temperature"]
    else:
        m.spectrum_type = oracle.SPECTRUM_MONOCHROMATIC
        m.mono_frequency = d["spectrum"]["frequency"]
    if d["cross_sections"] == "Verner":
        m.xsec_type = oracle.XSEC_VERNER
        m.recomb_type = oracle.RECOMB_VERNER
    else:
        m.xsec_type = oracle.XSEC_FIXED
        m.recomb_type = oracle.RECOMB_FIXED
        for i in range(14):
            m.xsec_fixed[i] = d["cross_sections"][i]
            m.recomb_fixed[i] = d["recombination_rates"][i]
    sim.run(nphoton, niter, seed=42)
    # (text output has 6 significant digits; the iterations feed the rounded
    # state back - as in test_cmi_gpu_executable_end_to_end)
    ionized = np.asarray(sim.x[0]) < 0.5
    assert 0.2 < ionized.mean() < 0.6
    assert np.allclose(last[:, 5], sim.x[0], rtol=5e-3, atol=0.)
