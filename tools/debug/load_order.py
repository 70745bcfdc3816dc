#!/usr/bin/env python3
"""GPU box: does loading a library that links libcmi_gpu.so before / after the
engine's own load change what HIP sees? usage: load_order.py LIB first|last"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cmacionize_amd import engine as E, GpuEngine
lib, order = sys.argv[1], sys.argv[2]
if order == "first":
    C.CDLL(lib)
E.load_library()
if order == "last":
    C.CDLL(lib)
try:
    eng = GpuEngine((4, 4, 4), (0, 0, 0), (1, 1, 1), (0, 0, 0), device=0)
    print(order, "OK")
except Exception as e:
    print(order, "FAIL", e)
