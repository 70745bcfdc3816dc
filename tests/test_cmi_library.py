"""The reference's LIBRARY mode (src/CMILibrary.cpp: cmi_init,
cmi_compute_neutral_fraction_*, cmi_destroy) on top of the GPU engine:
cmacionize_amd/libcmi_gpu_library.so exports the same C entry points; an SPH
code hands over particle arrays and gets neutral fractions back
(SPHArrayInterface: particles -> cell densities, simulation, cells ->
particles).

Checked against the oracle: the same mapping formulas in numpy (the
reference's "centroid" mapping, src/SPHArrayInterface.cpp:943-959 and
.hpp:156-200, with the cubic spline kernel of src/CubicSplineKernel.hpp),
the oracle's run on the mapped density field, the inverse mapping in numpy.

The "Petkova" mapping (src/SPHArrayInterface.cpp:208-925,960-1003) is pinned
by the reference's own known answers: the three total hydrogen numbers of
test/testSPHArrayInterface.cpp:95,123,151 (1000 particles at positions drawn
from glibc's unseeded rand(), mapped onto a 16^3 grid) at that test's
tolerances. The C ABI itself is exercised by a C program typed from the
reference's header (tests/support/cmi_library_caller.c) with guard words
round every output buffer."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cmacionize_amd", "libcmi_gpu_library.so")
BENCH = os.path.join(ROOT, "benchmarks")
PC = 3.086e16
M_H = 1.6737236e-27


@pytest.fixture(scope="module")
def library():
    subprocess.run(["make", "-C", os.path.join(ROOT, "cmacionize_amd",
                                               "csrc")], check=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "cmacionize_amd",
                                               "host"), "all"], check=True)
    L = C.CDLL(LIB)
    dp = C.POINTER(C.c_double)
    fp = C.POINTER(C.c_float)
    L.cmi_init.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_double,
                           C.c_char_p, C.c_int]
    L.cmi_init_periodic_dp.argtypes = [C.c_char_p, C.c_int, C.c_double,
                                       C.c_double, dp, dp, C.c_char_p, C.c_int]
    L.cmi_compute_neutral_fraction_dp.argtypes = [dp] * 6 + [C.c_size_t]
    L.cmi_compute_neutral_fraction_mp.argtypes = [dp] * 3 + [fp] * 3 + [
        C.c_size_t]
    vp = C.c_void_p
    L.cmi_gpu_library_map_to_cells.argtypes = [
        C.c_char_p, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, C.c_double,
        C.c_double, vp, vp, vp, vp, vp]
    L.cmi_gpu_library_map_to_particles.argtypes = [
        C.c_char_p, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, C.c_double,
        C.c_double, vp, vp, vp, vp, vp, vp]
    L.cmi_compute_neutral_fraction_sp.argtypes = [fp] * 6 + [C.c_size_t]
    return L


def test_library_exports_the_reference_entry_points(library, tmp_path):
    """src/CMILibrary.hpp:46-72 (no compute calls: runs without a GPU)."""
    for name in ("cmi_init", "cmi_init_periodic_dp", "cmi_init_periodic_sp",
                 "cmi_destroy", "cmi_compute_neutral_fraction_dp",
                 "cmi_compute_neutral_fraction_mp",
                 "cmi_compute_neutral_fraction_sp"):
        getattr(library, name)
    # a wrong mapping type is reported, not fatal
    p = tmp_path / "lib.param"
    p.write_text(open(os.path.join(BENCH, "stromgren.param")).read())
    library.cmi_init(str(p).encode(), 1, 1., 1., b"Voronoi", 0)
    assert library.cmi_gpu_library_status() == 1
    library.cmi_destroy()


def map_to_cells(library, mapping, precision, x, y, z, h, m, ncell,
                 anchor=(0., 0., 0.), sides=(1., 1., 1.), periodic_box=None):
    """SPHArrayInterface as a DensityFunction, host only: reset, initialize,
    one evaluation per cell of a Cartesian grid."""
    anchor = np.array(anchor, dtype=np.float64)
    sides = np.array(sides, dtype=np.float64)
    nc = np.array([ncell] * 3, dtype=np.int32)
    out = np.zeros(ncell ** 3)
    box = None if periodic_box is None else np.array(periodic_box,
                                                      dtype=np.float64)
    rc = library.cmi_gpu_library_map_to_cells(
        mapping, precision, x.ctypes.data, y.ctypes.data, z.ctypes.data,
        h.ctypes.data, m.ctypes.data, len(x), 1., 1.,
        None if box is None else box.ctypes.data, anchor.ctypes.data,
        sides.ctypes.data, nc.ctypes.data, out.ctypes.data)
    assert rc == 0
    return out


def test_petkova_mapping_known_answers(library):
    """test/testSPHArrayInterface.cpp:70-155: 1000 particles (h = 0.2, m =
    0.001) at Utilities::random_double() positions - rand() / RAND_MAX of the
    C library, never seeded, so the default seed 1; the three blocks draw from
    the one stream one after the other - on a 16^3 grid over the unit box:
    total hydrogen numbers 8.89848e26 (double arrays, 1e-6), 8.76356e26 (float
    h and m, 1e-5), 8.90243e26 (all float, 1e-6)."""
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    rand_max = 2147483647
    libc.srand(1)
    for precision, known, tolerance in ((0, 8.89848e26, 1.e-6),
                                        (1, 8.76356e26, 1.e-5),
                                        (2, 8.90243e26, 1.e-6)):
        r = np.array([libc.rand() / rand_max for _ in range(3000)])
        r = r.reshape(1000, 3)
        x, y, z = (np.ascontiguousarray(r[:, a]) for a in range(3))
        h = np.full(1000, 0.2)
        m = np.full(1000, 0.001)
        if precision >= 1:
            h, m = h.astype(np.float32), m.astype(np.float32)
        if precision == 2:
            x, y, z = (a.astype(np.float32) for a in (x, y, z))
        n = map_to_cells(library, b"Petkova", precision, x, y, z, h, m, 16)
        total = n.sum() / 16 ** 3   # get_total_hydrogen_number: sum n V
        assert abs(total - known) <= tolerance * (total + known), \
            (precision, total, known)


def test_petkova_mapping_conserves_mass(library):
    """"Petkova_oriented" (every face's normal into the cell - on a Cartesian
    grid the reference's own sum is not the integral, see
    PetkovaMapping::mass_fraction) integrates every kernel over every cell
    exactly: particles whose kernels lie inside the grid put all their mass on
    it (the interpolated vertex integrals are good to ~1e-3), and the mapping
    back hands every cell's ionized fraction out in shares that add up to
    one. Same vertex integrals, table and neighbour search as "Petkova"."""
    rng = np.random.default_rng(5)
    n = 400
    # (which side of an edge the projected particle lies on comes from
    # determinants of position vectors - src/SPHArrayInterface.cpp:789-795,
    # 831-837 - and is lost for a face in a plane through the origin: the
    # reference's behaviour, kept. Here the box is kept off those planes.)
    origin = np.array([1.03, 2.07, 3.01])
    x, y, z = (np.ascontiguousarray(rng.uniform(0.25, 0.75, n) + origin[a])
               for a in range(3))
    h = rng.uniform(0.1, 0.2, n)
    m = rng.uniform(0.5, 1.5, n)
    ncell = 12
    dens = map_to_cells(library, b"Petkova_oriented", 0, x, y, z, h, m, ncell,
                        anchor=origin)
    mass = dens.sum() * M_H / ncell ** 3
    # cells no kernel reaches get the reference's floor m[0] / V * 1e-6
    floor = (dens * M_H / ncell ** 3 < 2.e-6 * m[0]).sum() * 1.e-6 * m[0]
    assert abs(mass - floor - m.sum()) < 2.e-3 * m.sum(), (mass, m.sum())

    xH = rng.uniform(0., 1., ncell ** 3)
    nH = np.zeros(n)
    anchor, sides = origin.copy(), np.ones(3)
    nc = np.array([ncell] * 3, dtype=np.int32)
    rc = library.cmi_gpu_library_map_to_particles(
        b"Petkova_oriented", 0, x.ctypes.data, y.ctypes.data, z.ctypes.data,
        h.ctypes.data, m.ctypes.data, n, 1., 1., None, anchor.ctypes.data,
        sides.ctypes.data, nc.ctypes.data, xH.ctypes.data, nH.ctypes.data)
    assert rc == 0
    # sum over particles of what they lost = sum over the cells that have a
    # neighbour of their ionized fraction
    mid = (np.arange(ncell) + 0.5) / ncell
    CX, CY, CZ = np.meshgrid(mid + origin[0], mid + origin[1],
                             mid + origin[2], indexing="ij")
    cells = np.stack([CX.ravel(), CY.ravel(), CZ.ravel()], axis=1)
    pos = np.stack([x, y, z], axis=1)
    radius = 0.5 * np.sqrt(3.) / ncell
    r = np.linalg.norm(cells[:, None, :] - pos[None, :, :], axis=2)
    has_neighbour = (r <= h[None, :] + radius).any(axis=1)
    assert np.isclose((1. - nH).sum(), (1. - xH)[has_neighbour].sum(),
                      rtol=1e-10)
    # a neutral grid leaves the particles neutral
    xH[:] = 1.
    library.cmi_gpu_library_map_to_particles(
        b"Petkova_oriented", 0, x.ctypes.data, y.ctypes.data, z.ctypes.data,
        h.ctypes.data, m.ctypes.data, n, 1., 1., None, anchor.ctypes.data,
        sides.ctypes.data, nc.ctypes.data, xH.ctypes.data, nH.ctypes.data)
    assert np.array_equal(nH, np.ones(n))


def test_vertex_integrals_add_up_to_the_kernel(library):
    """One particle in the middle of one big cell: the sum of the 48 signed
    vertex integrals is the whole kernel, 1; moved so that a face cuts the
    kernel in half: 1/2 (symmetry); a corner at the particle: 1/8."""
    h = np.array([0.4])
    m = np.array([M_H])   # number density x volume = fraction inside
    origin = (1., 2., 3.)   # (no face in a plane through the origin)
    for pos, expect in (((0.5, 0.5, 0.5), 1.), ((1., 0.5, 0.5), 0.5),
                        ((0.3, 0.6, 0.45), 1.), ((0.9, 0.5, 0.5), None),
                        ((1., 1., 0.5), 0.25), ((1., 1., 1.), 0.125)):
        x, y, z = (np.array([c + o]) for c, o in zip(pos, origin))
        got = map_to_cells(library, b"Petkova_oriented", 0, x, y, z, h, m, 1,
                           anchor=origin)[0]
        if expect is None:
            # a kernel cut by one face 0.1 from its centre: shells of radius
            # r > 0.1 have the fraction (1 + 0.1 / r) / 2 of their area inside
            r = np.linspace(0., 0.4, 400001)[1:]
            shell = kernel(r / 0.4, 0.4) * 4. * np.pi * r * r
            inside = np.where(r > 0.1, 0.5 * (1. + 0.1 / r), 1.)
            expect = np.trapezoid(shell * inside, r) / np.trapezoid(shell, r)
            assert 0.6 < expect < 0.9
        # (trilinear interpolation of the vertex integrals: ~1e-3)
        assert abs(got - expect) < 3.e-3, (pos, got)


def kernel(u, h):
    """src/CubicSplineKernel.hpp:36-59"""
    KC1, KC2, KC5 = 2.546479089470, 15.278874536822, 5.092958178941
    w = np.where(u < 0.5, KC1 + KC2 * (u - 1.) * u * u,
                 KC5 * (1. - u) ** 3)
    return np.where(u < 1., w, 0.) / h ** 3


@pytest.mark.gpu
def test_library_mode_matches_the_oracle(library, oracle, tmp_path):
    ncell, npart_side, npacket, iterations = 16, 16, 40000, 6
    side = 10. * PC
    text = open(os.path.join(BENCH, "stromgren.param")).read()
    text = text.replace("[64, 64, 64]", "[%d, %d, %d]" % ((ncell,) * 3))
    text = text.replace("number of photons: 1e6",
                        "number of photons: %d" % npacket)
    text = text.replace("number of iterations: 20",
                        "number of iterations: %d" % iterations)
    text = text.replace("type: Gadget", "type: AsciiFile")
    p = tmp_path / "lib.param"
    p.write_text(text)
    # a slightly perturbed lattice of particles filling the box
    rng = np.random.default_rng(3)
    d = side / npart_side
    ax = (np.arange(npart_side) + 0.5) * d - 0.5 * side
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    pos = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    pos += rng.uniform(-0.2, 0.2, pos.shape) * d
    n = pos.shape[0]
    h = np.full(n, 1.9 * d)
    m = np.full(n, 100. * 1.e6 * M_H * d ** 3)

    # caller's units: parsec and solar masses
    ul, um = PC, 1.98855e30
    x, y, z = (np.ascontiguousarray(pos[:, a] / ul) for a in range(3))
    hh = np.ascontiguousarray(h / ul)
    mm = np.ascontiguousarray(m / um)
    nH = np.full(n, -1.)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        library.cmi_init(str(p).encode(), 1, ul, um, b"centroid", 0)
        assert library.cmi_gpu_library_status() == 0
        ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        library.cmi_compute_neutral_fraction_dp(ptr(x), ptr(y), ptr(z),
                                                ptr(hh), ptr(mm), ptr(nH), n)
        assert library.cmi_gpu_library_status() == 0
        # the mixed precision entry point: same particles, float h and m
        # (src/CMILibrary.hpp:66-68: its nH is FLOAT; n elements and a guard)
        nH_mp = np.full(n + 16, -1., dtype=np.float32)
        h32, m32 = hh.astype(np.float32), mm.astype(np.float32)
        fptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        library.cmi_compute_neutral_fraction_mp(ptr(x), ptr(y), ptr(z),
                                                fptr(h32), fptr(m32),
                                                fptr(nH_mp), n)
        assert np.all(nH_mp[n:] == -1.)
        nH_mp = nH_mp[:n].astype(np.float64)
        library.cmi_destroy()
    finally:
        os.chdir(cwd)

    # the same chain with numpy + the oracle
    cax = (np.arange(ncell) + 0.5) * (side / ncell) - 0.5 * side
    CX, CY, CZ = np.meshgrid(cax, cax, cax, indexing="ij")
    mid = np.stack([CX.ravel(), CY.ravel(), CZ.ravel()], axis=1)
    r = np.linalg.norm(mid[:, None, :] - pos[None, :, :], axis=2)
    W = m[None, :] * kernel(r / h[None, :], h[None, :])   # [cell, particle]
    rho = W.sum(axis=1)
    assert rho.min() > 0.
    sim = oracle.stromgren_simulation(ncell)
    sim.number_density[:] = rho / M_H
    sim.run(npacket, iterations, seed=42)
    xH = np.asarray(sim.x[0])
    expect = 1. - (W / rho[:, None] * (1. - xH)[:, None]).sum(axis=0)

    assert np.allclose(nH, expect, rtol=0, atol=5e-3), \
        np.abs(nH - expect).max()
    # (the reference's inverse mapping gives every particle the kernel-weighted
    # ionized fractions of the cells it covers, normalised per CELL: with about
    # one particle per cell, as here, the result stays close to [0, 1])
    rp = np.linalg.norm(pos, axis=1)
    assert nH[rp < 1.5 * PC].max() < 0.3
    assert nH[rp > 6. * PC].min() > 0.9
    assert np.allclose(nH_mp, nH, rtol=0, atol=5e-3)


@pytest.mark.gpu
def test_c_caller_typed_from_the_reference_header(library, tmp_path):
    """tests/support/cmi_library_caller.c: gcc-compiled C, prototypes typed
    from src/CMILibrary.hpp:46-72, the call sequence of the reference's own C
    caller (cmi_init_periodic_dp(..., "Petkova", 0) then _dp) and the other
    two precisions, every nH buffer exactly N elements between guard words."""
    exe = tmp_path / "cmi_library_caller"
    subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-o", str(exe),
                    os.path.join(ROOT, "tests", "support",
                                 "cmi_library_caller.c"),
                    "-L" + os.path.join(ROOT, "cmacionize_amd"),
                    "-lcmi_gpu_library", "-lcmi_gpu",
                    "-Wl,-rpath," + os.path.join(ROOT, "cmacionize_amd"),
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    text = open(os.path.join(BENCH, "stromgren.param")).read()
    text = text.replace("[64, 64, 64]", "[16, 16, 16]")
    text = text.replace("number of photons: 1e6", "number of photons: 40000")
    text = text.replace("number of iterations: 20", "number of iterations: 6")
    text = text.replace("type: Gadget", "type: AsciiFile")
    p = tmp_path / "caller.param"
    p.write_text(text)
    out = tmp_path / "caller.txt"
    for mapping in ("Petkova", "Petkova_oriented", "centroid"):
        run = subprocess.run([str(exe), str(p), mapping, str(out)],
                             cwd=tmp_path, capture_output=True, text=True,
                             timeout=600)
        # 0 = every guard word intact and every nH element written
        assert run.returncode == 0, (mapping, run.returncode, run.stderr)
        data = np.loadtxt(out)
        pos, nH = data[:, :3], data[:, 3:]
        assert np.all(np.isfinite(nH))
        # float positions against double positions, same periodic box
        assert np.abs(nH[:, 2] - nH[:, 0]).max() < \
            0.02 * max(1., np.abs(nH[:, 0]).max()), mapping
        if mapping == "Petkova":
            # the reference's sum of vertex integrals is not a volume integral
            # on a Cartesian grid (PetkovaMapping::mass_fraction): its numbers
            # are reproduced (test_petkova_mapping_known_answers), not physics
            continue
        radius = np.linalg.norm(pos, axis=1)
        # a Stromgren sphere of ~3 pc (stromgren.param in a 10 pc box). The
        # reference's mapping back takes from a particle its share of the
        # ionized fraction of EVERY cell it covers (shares add up to one per
        # cell, not per particle): with 19 cells per particle, as here, the
        # particles in the sphere end far below zero, those whose kernels do
        # not reach it stay at one
        # (the Petkova mappings do not wrap a kernel that sticks out of the
        # box - the reference passes the particle's own position - so the
        # density falls off towards the walls and the front reaches further
        # there: the corner particles are mostly, not fully, neutral)
        assert nH[radius < 1.5].max() < 0., mapping
        assert nH[radius > 6.5].min() > 0.2, mapping
        assert nH[radius > 6.5][:, 0].mean() > 0.75, mapping
        # float h and m: the non-periodic middle call sees the same particles
        # through a box of their own extent
        near = nH[radius < 2.]
        assert np.abs(near[:, 1] - near[:, 0]).max() < \
            0.25 * np.abs(near[:, 0]).max(), mapping


@pytest.mark.gpu
def test_the_references_library_test_input_runs_unchanged(tmp_path):
    """test/testCMICLibrary.c initialises the library with
    test_CMI_library.param (tests/golden/, byte for byte) and the "Petkova"
    mapping on a lattice of particles in its box, computes the neutral
    fractions and destroys it: the same through libcmi_gpu_library.so with
    the gcc-compiled caller (6^3 particles) - every buffer intact, every
    value finite, the particles near the star more ionized than the far
    ones."""
    import shutil
    exe = tmp_path / "cmi_library_caller"
    subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-o", str(exe),
                    os.path.join(ROOT, "tests", "support",
                                 "cmi_library_caller.c"),
                    "-L" + os.path.join(ROOT, "cmacionize_amd"),
                    "-lcmi_gpu_library", "-lcmi_gpu",
                    "-Wl,-rpath," + os.path.join(ROOT, "cmacionize_amd"),
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    src = os.path.join(ROOT, "tests", "golden", "test_CMI_library.param")
    p = tmp_path / "test_CMI_library.param"
    shutil.copy(src, p)
    out = tmp_path / "caller.txt"
    run = subprocess.run([str(exe), str(p), "Petkova", str(out)],
                         cwd=tmp_path, capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0, (run.returncode, run.stderr)
    assert open(src, "rb").read() == open(p, "rb").read()
    data = np.loadtxt(out)
    pos, nH = data[:, :3], data[:, 3:]
    assert nH.shape[0] == 216 and np.all(np.isfinite(nH))
    radius = np.linalg.norm(pos, axis=1)
    assert nH[radius < 1.5][:, 0].mean() < nH[radius > 6.5][:, 0].mean()
