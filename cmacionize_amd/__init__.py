"""cmacionize_amd - MI355X-native engine for the photon-transport +
ionization-balance hot path of CMacIonize.

The product is the HIP library behind include/cmi_gpu.h
(cmacionize_amd/libcmi_gpu.so, sources in cmacionize_amd/csrc). This package
only holds the thin host-side mirror of the reference's driver loop.
"""
from .engine import EngineError, GpuEngine, load_library  # noqa: F401
from .simulation import (ReplicaIterationDriver, distribute_packets,  # noqa: F401
                         STROMGREN)
