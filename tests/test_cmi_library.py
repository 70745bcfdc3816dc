"""The reference's LIBRARY mode (src/CMILibrary.cpp: cmi_init,
cmi_compute_neutral_fraction_*, cmi_destroy) on top of the GPU engine:
cmacionize_amd/libcmi_gpu_library.so exports the same C entry points; an SPH
code hands over particle arrays and gets neutral fractions back
(SPHArrayInterface: particles -> cell densities, simulation, cells ->
particles).

Checked against the oracle: the same mapping formulas in numpy (the
reference's "centroid" mapping, src/SPHArrayInterface.cpp:943-959 and
.hpp:156-200, with the cubic spline kernel of src/CubicSplineKernel.hpp),
the oracle's run on the mapped density field, the inverse mapping in numpy."""
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
                           C.c_char_p]
    L.cmi_init_periodic_dp.argtypes = [C.c_char_p, C.c_int, C.c_double,
                                       C.c_double, dp, dp, C.c_char_p, C.c_int]
    L.cmi_compute_neutral_fraction_dp.argtypes = [dp] * 6 + [C.c_size_t]
    L.cmi_compute_neutral_fraction_mp.argtypes = [dp] * 3 + [fp] * 2 + [
        dp, C.c_size_t]
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
    library.cmi_init(str(p).encode(), 1, 1., 1., b"Petkova")
    assert library.cmi_gpu_library_status() == 1
    library.cmi_destroy()


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
        library.cmi_init(str(p).encode(), 1, ul, um, b"centroid")
        assert library.cmi_gpu_library_status() == 0
        ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        library.cmi_compute_neutral_fraction_dp(ptr(x), ptr(y), ptr(z),
                                                ptr(hh), ptr(mm), ptr(nH), n)
        assert library.cmi_gpu_library_status() == 0
        # the mixed precision entry point: same particles, float h and m
        nH_mp = np.full(n, -1.)
        h32, m32 = hh.astype(np.float32), mm.astype(np.float32)
        fptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        library.cmi_compute_neutral_fraction_mp(ptr(x), ptr(y), ptr(z),
                                                fptr(h32), fptr(m32),
                                                ptr(nH_mp), n)
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
