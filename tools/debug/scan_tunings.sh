#!/bin/bash
# GPU box: a config's converged iteration under one tuning after the other.
#   tools/debug/scan_tunings.sh OUT CONFIG ITERS "key=value ..." "key=value ..." ...
OUT=$1; CFG=$2; ITERS=$3; shift 3
mkdir -p "$(dirname "$OUT")"
for T in "" "$@"; do
  echo "== ${T:-default}"
  python3 tools/run_config.py $CFG 256 1e8 $ITERS $T 2>&1 | tail -n 1 | cut -c1-110
done > "$OUT" 2>&1
