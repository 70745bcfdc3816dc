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
