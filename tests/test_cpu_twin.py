"""The CPU twin of the C ABI (oracle/cmi_cpu.c -> oracle/libcmi_cpu.so, test
infrastructure): the core entry points of include/cmi_gpu.h
under cmi_cpu_* names with the same argument lists (checked at compile time),
error codes and call-sequence rules, on top of the oracle. The GPU tests run
ONE sequence of ABI calls through both libraries."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import abi_driver as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True,
                   capture_output=True)


def test_twin_exports_the_core_entry_points():
    lib = C.CDLL(A.TWIN)
    lib.cmi_cpu_has.argtypes = [C.c_char_p]
    core = ["create", "destroy", "last_error", "number_of_cells"] + \
        list(A.CALLS)
    for name in core:
        assert hasattr(lib, "cmi_cpu_" + name), name
        assert lib.cmi_cpu_has(("cmi_gpu_" + name).encode()) == 1
    # what the twin does not have says so
    assert lib.cmi_cpu_has(b"cmi_gpu_set_trackers") == 0
    assert lib.cmi_cpu_has(b"cmi_gpu_group_create") == 0


def test_twin_is_the_oracle(oracle):
    """the same run through the twin's ABI and through the oracle's own API:
    the same numbers (it is the same code underneath; its OpenMP threads add
    in a different order from run to run)"""
    twin = A.twin(12)
    steps, (xH, xHe, T) = A.run_benchmark(twin, "stromgren_diffuse", 12,
                                          20000, 3)
    twin.close()
    sim = oracle.stromgren_simulation(12, diffuse=True)
    for loop, (tw, tc, JH, JHe, hH) in enumerate(steps):
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, loop, 0, 20000)
        assert tw == sim.totweight and np.array_equal(tc, sim.typecount)
        # (each run feeds its own state back: the rounding noise of one
        # iteration is the next one's input)
        # (measured: 3e-15 / 3e-12 / 1e-8 of the largest value in iterations
        # 0 / 1 / 2)
        assert np.allclose(JH, sim.J[0], rtol=1e-6,
                           atol=1e-7 * sim.J[0].max())
        assert np.allclose(hH, sim.heating[0], rtol=1e-6,
                           atol=1e-7 * np.abs(sim.heating[0]).max())
        sim.update(loop, sim.totweight)
    assert np.allclose(xH, sim.x[0], rtol=1e-5, atol=0.)


def test_twin_error_behaviour_is_the_engines():
    """call-sequence and argument errors: the ABI's codes and a message"""
    twin = A.twin(4)
    with pytest.raises(A.AbiError) as err:
        twin.call("shoot", 1, 0, 0, 10)   # nothing set yet
    assert err.value.code == 3            # CMI_GPU_ESTATE
    assert "must be set first" in str(err.value)
    with pytest.raises(A.AbiError) as err:
        twin.call("upload_field", 99, None)
    assert err.value.code == 1            # CMI_GPU_EINVAL
    with pytest.raises(A.AbiError) as err:
        twin.call("update_cells_range", 0, 1., 0, 10 ** 6)
    assert err.value.code == 1
    twin.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,ncell,npacket,iterations,rtol", [
    ("stromgren", 16, 30000, 3, 1e-9),
    ("stromgren_diffuse", 16, 30000, 3, 1e-9),
    ("lexington", 16, 30000, 5, 1e-6),
])
def test_engine_and_twin_through_the_same_calls(kind, ncell, npacket,
                                                iterations, rtol):
    """ONE sequence of ABI calls, two libraries: packet counters identical,
    integrals within the order of the sums (and libm ulps), the state after
    the cell updates - each library feeding its own state back - within the
    same bounds as the oracle tests (hydrogen-only: the closed form from
    integrals that differ by 1e-9)."""
    eng = A.engine(ncell)
    cpu = A.twin(ncell)
    got, (xH, xHe, T) = A.run_benchmark(eng, kind, ncell, npacket, iterations)
    ref, (xH0, xHe0, T0) = A.run_benchmark(cpu, kind, ncell, npacket,
                                           iterations)
    eng.close()
    cpu.close()
    for (tw, tc, JH, JHe, hH), (tw0, tc0, JH0, JHe0, hH0) in zip(got, ref):
        assert tw == tw0 == npacket
        if kind == "lexington":
            # (a frequency within an ulp of a threshold: tests/
            # test_gpu_physics.py::test_lexington_iteration_matches_oracle)
            assert np.abs(tc - tc0).max() <= 3
        else:
            assert np.array_equal(tc, tc0)
        # measured (tools/debug/twin_diffs.py): 2e-15 of the largest value in
        # the first iteration, 1.5e-9 in the third of the Stromgren runs -
        # each library feeds its own state back, the rounding noise of one
        # iteration is the next one's input
        for a, b in ((JH, JH0), (JHe, JHe0), (hH, hH0)):
            assert np.allclose(a, b, rtol=1e-5,
                               atol=1e-7 * max(np.abs(b).max(), 1e-300))
    first, first0 = got[0], ref[0]
    for a, b in zip(first[2:], first0[2:]):
        assert np.allclose(a, b, rtol=rtol,
                           atol=1e-12 * max(np.abs(b).max(), 1e-300))
    # (measured: x_H 7e-6 at the ionization front of the Stromgren runs,
    # 4e-13 / 3e-13 for x_H / T of lexingtonHII40)
    assert np.allclose(xH, xH0, rtol=1e-4, atol=0.)
    assert np.allclose(T, T0, rtol=1e-4)
