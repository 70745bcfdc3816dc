"""The Gadget / HDF5 snapshot writer of the C++ host (the reference's default
DensityGridWriter, src/GadgetDensityGridWriter.cpp:107-358) - written without
an HDF5 library. Checked on the CPU (cmi-gpu --dry-run-snapshot evaluates the
DensityFunction on the host and writes snapshot 0):

 * a pure-Python reader of the HDF5 structures (tests/hdf5_mini.py) finds
   every group, attribute and dataset the reference's analysis scripts read
   (benchmarks/stromgren.py:75-83, lexingtonHII40.py:86-108), with the values
   of the lowered parameter file;
 * if an HDF5 library is present on the machine (this image has conda's
   libhdf5.so, no headers / h5py), the REAL library opens the file and returns
   the same numbers - through ctypes.
"""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

from test_host_driver import BENCH, exe  # noqa: F401 (fixture)
import hdf5_mini

PC = 3.086e16


def make_snapshot(exe, tmp_path, bench, ncell=12):
    text = open(os.path.join(BENCH, bench)).read()
    text = text.replace("[64, 64, 64]", "[%d, %d, %d]" % ((ncell,) * 3))
    if bench.startswith("lexington"):
        import shutil
        shutil.copy(os.path.join(BENCH, "lexingtonHII40.yml"), tmp_path)
        text = text.replace("NumberDensity: 0", "NumberDensity: 1")
    p = tmp_path / "run.param"
    p.write_text(text)
    r = subprocess.run([exe, "--params", str(p), "--dry-run-snapshot"],
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout
    files = sorted(glob.glob(str(tmp_path / "*000.hdf5")))
    assert len(files) == 1, os.listdir(tmp_path)
    return files[0]


def expected_coordinates(ncell):
    side = 10. * PC / ncell
    ax = (np.arange(ncell) * side + 0.5 * side)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)


def test_gadget_snapshot_structure(exe, tmp_path):  # noqa: F811
    ncell = 12
    path = make_snapshot(exe, tmp_path, "lexingtonHII40.param", ncell)
    f = hdf5_mini.read(path)
    n = ncell ** 3
    header = f["/Header"].attrs
    assert np.allclose(header["BoxSize"], [10. * PC] * 3, rtol=1e-15)
    assert header["Dimension"] == 3
    assert list(header["NumPart_ThisFile"]) == [n, 0, 0, 0, 0, 0]
    assert list(header["NumPart_Total"]) == [n, 0, 0, 0, 0, 0]
    assert list(header["NumPart_Total_HighWord"]) == [0] * 6
    assert header["NumFilesPerSnapshot"] == 1 and header["Time"] == 0.
    assert list(header["MassTable"]) == [0.] * 6
    units = f["/Units"].attrs
    assert units["Unit length in cgs (U_L)"] == 100.
    assert units["Unit mass in cgs (U_M)"] == 1000.
    assert units["Unit time in cgs (U_t)"] == 1.
    assert f["/RuntimePars"].attrs["Iteration"] == 0
    params = f["/Parameters"].attrs
    assert params["DensityGrid:number of cells"] == "[12, 12, 12]"
    assert params["IonizationSimulation:random seed"] == "42"
    assert "Code" in f.root.members and "Configuration" in f.root.members
    part = f["/PartType0"].members
    ions = ["H", "He", "C+", "C++", "N", "N+", "N++", "O", "O+", "Ne", "Ne+",
            "S+", "S++", "S+++"]  # the fields lexingtonHII40.py reads
    assert sorted(part) == sorted(
        ["Coordinates", "NumberDensity", "Temperature"] +
        ["NeutralFraction" + ion for ion in ions])
    coords = part["Coordinates"].data
    assert coords.shape == (n, 3)
    # cell midpoints relative to the box anchor, row-major x, y, z
    assert np.allclose(coords, expected_coordinates(ncell), rtol=1e-14)
    r = np.linalg.norm(coords - 5. * PC, axis=1)
    dens = part["NumberDensity"].data
    assert dens.shape == (n,)
    assert np.array_equal(dens, np.where(r <= 3.e16, 0., 1.e8))
    assert np.array_equal(part["Temperature"].data,
                          np.where(r <= 3.e16, 0., 8000.))
    assert np.all(part["NeutralFractionH"].data == 1.e-6)


def test_default_fields_are_the_reference_defaults(exe, tmp_path):  # noqa: F811
    """stromgren.param switches NumberDensity off: Coordinates and the
    hydrogen neutral fraction remain (DensityGridWriterFields::default_flag)"""
    path = make_snapshot(exe, tmp_path, "stromgren.param", 8)
    f = hdf5_mini.read(path)
    assert sorted(f["/PartType0"].members) == ["Coordinates",
                                               "NeutralFractionH"]
    assert os.path.basename(path) == "stromgren_000.hdf5"


def find_libhdf5():
    for pattern in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/*/libhdf5*.so*",
                    "/usr/lib/libhdf5*.so*"):
        for p in sorted(glob.glob(pattern)):
            if "_hl" in p or "fortran" in p or "cpp" in p:
                continue
            try:
                return C.CDLL(p)
            except OSError:
                continue
    return None


def test_real_hdf5_library_reads_the_file(exe, tmp_path):  # noqa: F811
    lib = find_libhdf5()
    if lib is None:
        pytest.skip("no HDF5 library on this machine")
    ncell = 10
    path = make_snapshot(exe, tmp_path, "lexingtonHII40.param", ncell)
    n = ncell ** 3
    hid = C.c_int64
    lib.H5open()
    lib.H5Fopen.restype = hid
    lib.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    lib.H5Dopen2.restype = hid
    lib.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Dget_space.restype = hid
    lib.H5Dget_space.argtypes = [hid]
    lib.H5Sget_simple_extent_ndims.argtypes = [hid]
    lib.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint64)]
    lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    lib.H5Aopen_by_name.restype = hid
    lib.H5Aopen_by_name.argtypes = [hid, C.c_char_p, C.c_char_p, hid, hid]
    lib.H5Aread.argtypes = [hid, hid, C.c_void_p]
    lib.H5Gopen2.restype = hid
    lib.H5Gopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Gget_num_objs.argtypes = [hid, C.POINTER(C.c_uint64)]
    for name in ("H5Dclose", "H5Sclose", "H5Aclose", "H5Fclose", "H5Gclose"):
        getattr(lib, name).argtypes = [hid]
    native_double = hid.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
    native_uint = hid.in_dll(lib, "H5T_NATIVE_UINT_g").value
    f = lib.H5Fopen(path.encode(), 0, 0)  # H5F_ACC_RDONLY, H5P_DEFAULT
    assert f >= 0, "the HDF5 library refuses the file"

    def dataset(name):
        d = lib.H5Dopen2(f, name.encode(), 0)
        assert d >= 0, name
        s = lib.H5Dget_space(d)
        rank = lib.H5Sget_simple_extent_ndims(s)
        dims = (C.c_uint64 * rank)()
        lib.H5Sget_simple_extent_dims(s, dims, None)
        out = np.zeros(tuple(dims))
        assert lib.H5Dread(d, native_double, 0, 0, 0,
                           out.ctypes.data_as(C.c_void_p)) >= 0
        lib.H5Sclose(s)
        lib.H5Dclose(d)
        return out

    mine = hdf5_mini.read(path)
    for name in ("Coordinates", "NumberDensity", "Temperature",
                 "NeutralFractionH", "NeutralFractionO+",
                 "NeutralFractionS+++"):
        got = dataset("/PartType0/" + name)
        assert np.array_equal(got, mine["/PartType0/" + name].data), name
    assert dataset("/PartType0/Coordinates").shape == (n, 3)
    # attributes
    a = lib.H5Aopen_by_name(f, b"/Header", b"BoxSize", 0, 0)
    assert a >= 0
    box = np.zeros(3)
    assert lib.H5Aread(a, native_double, box.ctypes.data_as(C.c_void_p)) >= 0
    lib.H5Aclose(a)
    assert np.allclose(box, 10. * PC, rtol=1e-15)
    a = lib.H5Aopen_by_name(f, b"/Header", b"NumPart_Total", 0, 0)
    assert a >= 0
    numpart = np.zeros(6, dtype=np.uint32)
    assert lib.H5Aread(a, native_uint, numpart.ctypes.data_as(C.c_void_p)) >= 0
    lib.H5Aclose(a)
    assert list(numpart) == [n, 0, 0, 0, 0, 0]
    g = lib.H5Gopen2(f, b"/PartType0", 0)
    assert g >= 0
    count = C.c_uint64()
    assert lib.H5Gget_num_objs(g, C.byref(count)) >= 0 and count.value == 17
    lib.H5Gclose(g)
    lib.H5Fclose(f)


@pytest.mark.gpu
def test_gadget_snapshots_of_a_run(exe, tmp_path):  # noqa: F811
    """cmi-gpu with the reference's default writer: the final snapshot holds
    the state of the engine - equal to what the AsciiFile writer prints for
    the same run (6 digits) - and the analysis of benchmarks/stromgren.py
    (radial profile of the neutral fraction around the box centre) works on
    it."""
    text = open(os.path.join(BENCH, "stromgren.param")).read()
    text = text.replace("[64, 64, 64]", "[16, 16, 16]")
    text = text.replace("number of photons: 1e6", "number of photons: 20000")
    text = text.replace("number of iterations: 20", "number of iterations: 3")
    out = {}
    for kind in ("Gadget", "AsciiFile"):
        d = tmp_path / kind
        d.mkdir()
        p = d / "run.param"
        p.write_text(text.replace("type: Gadget", "type: " + kind))
        r = subprocess.run([exe, "--params", str(p)], capture_output=True,
                           text=True, cwd=str(d))
        assert r.returncode == 0, r.stderr
        out[kind] = d
    f = hdf5_mini.read(str(out["Gadget"] / "stromgren_003.hdf5"))
    assert f["/RuntimePars"].attrs["Iteration"] == 3
    ascii_ = np.loadtxt(out["AsciiFile"] / "stromgren_003.txt")
    xH = f["/PartType0/NeutralFractionH"].data
    # two runs of the same packets: the order of the atomic sums differs, the
    # balance feeds the 1e-15 differences back from iteration to iteration
    assert np.allclose(xH, ascii_[:, 5], rtol=2e-3, atol=0.)
    assert np.median(np.abs(xH / ascii_[:, 5] - 1.)) < 2e-6
    coords = f["/PartType0/Coordinates"].data
    box = f["/Header"].attrs["BoxSize"]
    # benchmarks/stromgren.py:78-83
    radius = np.sqrt(((coords - 0.5 * box) ** 2).sum(axis=1))
    assert xH[radius < 0.2 * box[0]].mean() < 0.1 < xH[radius > 0.45 * box[0]].mean()
    # initial and final snapshot (no --every-iteration-output)
    assert sorted(n for n in os.listdir(out["Gadget"])
                  if n.endswith(".hdf5")) == ["stromgren_000.hdf5",
                                              "stromgren_003.hdf5"]


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", [None, "2,1,2"])
def test_emission_lines_in_the_final_snapshot(exe, tmp_path, oracle, blocks):
    """"EmissivityValues:<name>: true" (the switches of the reference's
    emission mode, src/EmissivityCalculationSimulation.cpp:70-74): the lines
    are computed on the device from the final state and written as the
    datasets that mode appends to a snapshot (:181-193) - checked against the
    oracle's calculate_emissivities of the snapshot's own fields."""
    import shutil
    text = open(os.path.join(BENCH, "lexingtonHII40.param")).read()
    text = text.replace("[64, 64, 64]", "[16, 16, 16]")
    text = text.replace("number of photons: 1e8", "number of photons: 30000")
    text = text.replace("number of iterations: 20", "number of iterations: 6")
    text = text.replace("NumberDensity: 0", "NumberDensity: 1")
    text += ("\nEmissivityValues:\n  Halpha: true\n  OIII_5007: true\n"
             "  SIII_6213: true\n  BaHigh: true\n  NII_6584: false\n")
    shutil.copy(os.path.join(BENCH, "lexingtonHII40.yml"), tmp_path)
    p = tmp_path / "run.param"
    p.write_text(text)
    r = subprocess.run([exe, "--params", str(p)] +
                       (["--blocks", blocks] if blocks else []),
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    first = hdf5_mini.read(str(tmp_path / "lexingtonHII40_000.hdf5"))
    assert "Halpha" not in first["/PartType0"].members
    f = hdf5_mini.read(str(tmp_path / "lexingtonHII40_006.hdf5"))
    assert "NII_6584" not in f["/PartType0"].members
    ions = ["H", "He", "C+", "C++", "N", "N+", "N++", "O", "O+", "Ne", "Ne+",
            "S+", "S++", "S+++"]
    x = np.array([f["/PartType0/NeutralFraction" + i].data for i in ions])
    n = f["/PartType0/NumberDensity"].data
    T = f["/PartType0/Temperature"].data
    sim = oracle.lexington_simulation(4)
    index = {"Halpha": "HAlpha", "OIII_5007": "OIII_5007",
             "SIII_6213": "SIII_6312", "BaHigh": "BALMER_JUMP_HIGH"}
    lit = np.flatnonzero((x[0] < 0.2) & (T > 3000.))
    assert len(lit) > 100
    for name, line in index.items():
        values = f["/PartType0/" + name].data
        assert values.shape == (16 ** 3,)
        assert not values[(x[0] >= 0.2) | (T <= 3000.)].any()
        k = oracle.EMISSION_LINES.index(line)
        for c in lit[::5]:
            ref = oracle.emissivities(sim.model, n[c], T[c], x[:, c])[k]
            assert abs(values[c] - ref) <= 1e-10 * ref, (name, c)


@pytest.mark.gpu
def test_emission_mode_on_a_snapshot(exe, tmp_path, oracle):
    """`cmi-gpu --emission --params lines.param --file snapshot.hdf5` (the
    reference's emission mode, src/EmissivityCalculationSimulation.cpp): the
    flagged lines are added to the snapshot, everything else in it stays; the
    values equal the oracle's and those the run itself writes with the same
    switches in its parameter file."""
    import shutil
    text = open(os.path.join(BENCH, "lexingtonHII40.param")).read()
    text = text.replace("[64, 64, 64]", "[14, 14, 14]")
    text = text.replace("number of photons: 1e8", "number of photons: 30000")
    text = text.replace("number of iterations: 20", "number of iterations: 6")
    text = text.replace("NumberDensity: 0", "NumberDensity: 1")
    switches = ("\nEmissivityValues:\n  Hbeta: true\n  OIII_5007: true\n"
                "  NeIII_3869: true\n  avg_T: true\n  WFC2_F555W: true\n")
    names = ["Hbeta", "OIII_5007", "NeIII_3869", "avg_T", "WFC2_F555W"]
    results = {}
    for label, extra in (("plain", ""), ("in_run", switches)):
        d = tmp_path / label
        d.mkdir()
        shutil.copy(os.path.join(BENCH, "lexingtonHII40.yml"), d)
        (d / "run.param").write_text(text + extra)
        r = subprocess.run([exe, "--params", "run.param"], capture_output=True,
                           text=True, cwd=str(d))
        assert r.returncode == 0, r.stderr
        results[label] = str(d / "lexingtonHII40_006.hdf5")
    before = hdf5_mini.read(results["plain"])
    lines = tmp_path / "lines.param"
    lines.write_text(switches)
    for repeat in range(2):
        r = subprocess.run([exe, "--emission", "--params", str(lines),
                            "--file", results["plain"]], capture_output=True,
                           text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr
        assert ("already exists" in r.stdout) == (repeat == 1)
    after = hdf5_mini.read(results["plain"])
    in_run = hdf5_mini.read(results["in_run"])
    # everything that was there still is
    assert sorted(after.root.members) == sorted(before.root.members)
    for g in before.root.members:
        assert dict(after["/" + g].attrs).keys() == \
            dict(before["/" + g].attrs).keys()
        for k, v in before["/" + g].attrs.items():
            assert np.array_equal(np.asarray(after["/" + g].attrs[k]),
                                  np.asarray(v)), (g, k)
    for name, node in before["/PartType0"].members.items():
        assert np.array_equal(after["/PartType0/" + name].data, node.data)
    assert sorted(after["/PartType0"].members) == \
        sorted(list(before["/PartType0"].members) + names)
    # the lines the run wrote itself (same switches in its parameter file)
    # are what the emission mode computes from that snapshot
    r = subprocess.run([exe, "--emission", "--params", str(lines), "--file",
                        results["in_run"]], capture_output=True, text=True,
                       cwd=str(tmp_path))
    assert r.returncode == 0 and "already exists" in r.stdout, r.stderr
    redone = hdf5_mini.read(results["in_run"])
    for name in names:
        assert np.array_equal(redone["/PartType0/" + name].data,
                              in_run["/PartType0/" + name].data), name
        assert in_run["/PartType0/" + name].data.max() > 0.
    # ... and the oracle's
    ions = ["H", "He", "C+", "C++", "N", "N+", "N++", "O", "O+", "Ne", "Ne+",
            "S+", "S++", "S+++"]
    x = np.array([after["/PartType0/NeutralFraction" + i].data for i in ions])
    n = after["/PartType0/NumberDensity"].data
    T = after["/PartType0/Temperature"].data
    sim = oracle.lexington_simulation(4)
    index = {"Hbeta": "HBeta", "OIII_5007": "OIII_5007",
             "NeIII_3869": "NeIII_3869", "avg_T": "avg_T",
             "WFC2_F555W": "WFC2_F555W"}
    lit = np.flatnonzero((x[0] < 0.2) & (T > 3000.))
    assert len(lit) > 50
    for name in names:
        values = after["/PartType0/" + name].data
        k = oracle.EMISSION_LINES.index(index[name])
        for c in lit[::5]:
            ref = oracle.emissivities(sim.model, n[c], T[c], x[:, c])[k]
            assert abs(values[c] - ref) <= 1e-10 * ref, (name, c)
    # a snapshot without the ionic fractions is refused
    r = subprocess.run([exe, "--emission", "--params", str(lines), "--file",
                        str(tmp_path / "nowhere.hdf5")], capture_output=True,
                       text=True, cwd=str(tmp_path))
    assert r.returncode != 0 and "Could not open" in r.stderr


def test_snapshot_opens_with_libhdf5_when_there_is_one(exe, tmp_path):  # noqa: F811
    """The compatibility claim the reference's analysis scripts depend on
    (benchmarks/*.py read the snapshots with h5py): where h5py is installed, a
    snapshot laid out by the host's writer opens with the real library and
    holds the same groups, attributes and datasets the independent Python
    reader finds. (No h5py in the build image: skipped there.)"""
    h5py = pytest.importorskip("h5py")
    path = make_snapshot(exe, tmp_path, "lexingtonHII40.param", 8)
    mine = hdf5_mini.read(path)
    with h5py.File(path, "r") as f:
        assert sorted(f.keys()) == sorted(mine.root.members)
        for group in ("Header", "Units", "RuntimePars", "Parameters"):
            for name, value in mine["/" + group].attrs.items():
                got = f[group].attrs[name]
                if isinstance(value, str):
                    got = got.decode() if isinstance(got, bytes) else str(got)
                    assert got == value, (group, name)
                else:
                    assert np.array_equal(np.asarray(got), np.asarray(value))
        for name, node in mine["/PartType0"].members.items():
            assert np.array_equal(f["PartType0"][name][...],
                                  mine["/PartType0/" + name].data), name
