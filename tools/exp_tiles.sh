#!/bin/bash
# GPU box: tile-shape experiment builds (cmacionize_amd/variants/, made with
# `make -C cmacionize_amd/csrc variant NAME=... DEFS="-DCMI_TILE_L?_FULL=..."`)
# against the default build: the re-emission invariance test as the parity
# check, then the launch list of one converged iteration.
#   tools/exp_tiles.sh OUTDIR lexington|diffuse NAME [NAME ...]
OUT=$1; CFG=$2; shift 2
mkdir -p "$OUT"
for NAME in default "$@"; do
  if [ "$NAME" = default ]; then unset CMI_GPU_LIBRARY; else
    export CMI_GPU_LIBRARY=$PWD/cmacionize_amd/variants/libcmi_gpu_$NAME.so; fi
  echo "== $NAME"
  python3 -m pytest tests/test_gpu_fullsize_physics.py -x -q -m gpu -k "reemission_reordering" 2>&1 | tail -n 2
  ITERS=7; [ "$CFG" = diffuse ] && ITERS=9
  CMI_SHOW_LAUNCHES=1 python3 tools/run_config.py $CFG 256 1e8 $ITERS 2>&1 | tail -n 2 | cut -c1-1700
done > "$OUT/tiles_$CFG.txt" 2>&1
