#!/bin/bash
# GPU box: one experiment build after the other (cmacionize_amd/variants/)
# against the default library: a parity test, then the converged iteration.
#   tools/exp_batch.sh OUT CONFIG ITERS "PYTEST -k EXPR" NAME [NAME ...]
OUT=$1; CFG=$2; ITERS=$3; KEXPR=$4; shift 4
mkdir -p "$(dirname "$OUT")"
for NAME in default "$@"; do
  if [ "$NAME" = default ]; then unset CMI_GPU_LIBRARY; else
    export CMI_GPU_LIBRARY=$PWD/cmacionize_amd/variants/libcmi_gpu_$NAME.so; fi
  echo "== $NAME"
  python3 -m pytest tests/test_gpu_transport.py -x -q -m gpu -k "$KEXPR" 2>&1 | tail -n 1
  python3 tools/run_config.py $CFG 256 1e8 $ITERS 2>&1 | tail -n 2 | cut -c1-150
done > "$OUT" 2>&1
