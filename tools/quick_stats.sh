#!/bin/bash
# GPU box: kernel statistics of one bench configuration, compact.
#   tools/quick_stats.sh CONFIG [STEPS]   -> gpurun_out/quick_CONFIG.txt
CFG=${1:-stromgren_diffuse}
STEPS=${2:-20}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/quick_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/quick_stats -- \
  python3 "$REPO/bench.py" --config $CFG --steps $STEPS --no-cpu-baseline > /tmp/quick_stats.log 2>&1
python3 "$REPO/tools/profile_summary.py" /tmp/quick_stats /tmp/quick_stats.log > "$OUT/quick_$CFG.txt"
head -${3:-22} "$OUT/quick_$CFG.txt" | cut -c1-150
