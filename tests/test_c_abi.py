"""CPU tests of the drop-in boundary itself: libcmi_gpu.so loads without a
GPU, exports exactly the entry points include/cmi_gpu.h declares (and the
Python binding lists), and fails loudly - no CPU fallback - when asked for an
engine on a machine without a HIP device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cmi_gpu.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cmi_gpu_[a-z_0-9]+)\s*\(", text)))


def test_header_binding_and_library_agree():
    from cmacionize_amd import engine
    declared = declared_symbols()
    assert len(declared) > 30
    # the binding knows every declared entry point, and nothing else
    assert set(declared) == set(engine.EXPORTED_SYMBOLS), \
        set(declared) ^ set(engine.EXPORTED_SYMBOLS)
    lib = C.CDLL(engine.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    # every entry point cites the reference interface it replaces
    text = open(HEADER).read()
    assert text.count("src/") >= 40


def test_library_mode_header_and_library_agree():
    """include/cmi_library.h: the reference's seven entry points
    (src/CMILibrary.hpp:46-72) with the reference's argument types, plus the
    status and mapping probes; libcmi_gpu_library.so exports all of them and
    is compiled against this header."""
    text = open(os.path.join(ROOT, "include", "cmi_library.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(cmi_[a-z_0-9]+)\s*\(", code)))
    assert declared == sorted([
        "cmi_init", "cmi_init_periodic_dp", "cmi_init_periodic_sp",
        "cmi_destroy", "cmi_compute_neutral_fraction_dp",
        "cmi_compute_neutral_fraction_mp", "cmi_compute_neutral_fraction_sp",
        "cmi_gpu_library_status", "cmi_gpu_library_map_to_cells",
        "cmi_gpu_library_map_to_particles"])
    lib = C.CDLL(os.path.join(ROOT, "cmacionize_amd",
                              "libcmi_gpu_library.so"))
    for name in declared:
        assert hasattr(lib, name), name
    flat = " ".join(code.split())
    # the two places round 3 got wrong
    assert ("void cmi_init(const char *parameter_file, const int num_thread, "
            "const double unit_length_in_SI, const double unit_mass_in_SI, "
            "const char *mapping_type, const int talk);") in flat
    assert ("void cmi_compute_neutral_fraction_mp(const double *x, "
            "const double *y, const double *z, const float *h, "
            "const float *m, float *nH, const size_t N);") in flat
    source = open(os.path.join(ROOT, "cmacionize_amd", "host",
                               "CMILibrary.cpp")).read()
    assert '#include "../../include/cmi_library.h"' in source


def test_no_cpu_fallback():
    """Without a HIP device cmi_gpu_create returns an error and a message;
    with one, an impossible device ordinal does."""
    from cmacionize_amd import engine
    with pytest.raises(engine.EngineError) as info:
        engine.GpuEngine((4, 4, 4), (0., 0., 0.), (1., 1., 1.), device=9999)
    assert "device" in str(info.value).lower()


def test_flight_record_constants_match():
    from cmacionize_amd import simulation
    text = open(HEADER).read()
    assert "#define CMI_GPU_FLIGHT_DOUBLES %d" % simulation.FLIGHT_DOUBLES in text
    assert "[12] int64" in text and simulation.FLIGHT_CELL == 12
