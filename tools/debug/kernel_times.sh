#!/bin/bash
# GPU box: per-kernel times of a few iterations of a config, default library
# against experiment builds.  tools/debug/kernel_times.sh OUT CONFIG ITERS NAME...
OUT=$1; CFG=$2; ITERS=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p "$OUT"
for NAME in default "$@"; do
  if [ "$NAME" = default ]; then unset CMI_GPU_LIBRARY; else
    export CMI_GPU_LIBRARY=$PWD/cmacionize_amd/variants/libcmi_gpu_$NAME.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$NAME" -- \
    python3 tools/run_config.py $CFG 256 1e8 $ITERS > "$OUT/run_$NAME.log" 2>&1
  f=$(find "$OUT/stats_$NAME" -name '*kernel_stats.csv' | head -n 1)
  echo "== $NAME" >> "$OUT/summary.txt"
  python3 - "$f" >> "$OUT/summary.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-60s %6s %10.3f ms avg %9.3f" % (r["Name"][:60], r["Calls"],
          float(r["TotalDurationNs"]) * 1e-6, float(r["AverageNs"]) * 1e-6))
PY
  rm -rf "$OUT/stats_$NAME"
done
