#!/bin/bash
# GPU box: the xin_slots episode (DESIGN_LOG.md "A result that changed with one
# more kernel argument") on the tree of 3e4ff5c exported to gpurun_in/old, with
# variant builds of its libcmi_gpu.so. usage: episode.sh [variant ...]
cd gpurun_in/old || exit 1
for v in "$@"; do
  lib=cmacionize_amd/libcmi_gpu.so
  [ "$v" != base ] && lib=cmacionize_amd/variants/libcmi_gpu_$v.so
  [ -f "$lib" ] || continue
  echo "== variant '$v'"
  CMI_GPU_LIBRARY=$PWD/$lib timeout 300 python3 tools/debug/cont_decomposed.py 2>&1 | grep -v amdgpu.ids
done
