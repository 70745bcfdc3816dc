#!/bin/bash
# GPU box: what leaves every block after every stage of the failing set-up
# (exports sorted by packet id, counters per block), per variant library, into
# gpurun_out/trace_<variant>.npz. usage: episode_trace.sh variant ...
cd gpurun_in/old || exit 1
for v in "$@"; do
  lib=cmacionize_amd/libcmi_gpu.so
  [ "$v" != base ] && lib=cmacionize_amd/variants/libcmi_gpu_$v.so
  [ -f "$lib" ] || continue
  echo "== variant '$v'"
  CMI_TRACE_OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$v.npz CMI_GPU_LIBRARY=$PWD/$lib timeout 300 python3 tools/debug/cont_trace.py 2>&1 | grep -v amdgpu.ids
done
