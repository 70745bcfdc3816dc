#!/bin/bash
# GPU box: the round's measurement set. usage: tools/round_measure.sh OUTDIR
# - bench.py line of every single-GPU config
# - rocprofv3 --kernel-trace --stats of the same command
# - FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs) for roofline.traffic
set -u
REPO=$(pwd)
OUT=$REPO/$1
mkdir -p "$OUT"
export TMPDIR=/tmp
for CFG in stromgren stromgren_diffuse lexington; do
  python3 bench.py --config $CFG > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
  echo "bench $CFG rc=$?"
done
cd /tmp
for CFG in stromgren stromgren_diffuse lexington; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$CFG" -- \
    python3 "$REPO/bench.py" --config $CFG --no-cpu-baseline > "$OUT/stats_$CFG.log" 2>&1
  echo "stats $CFG rc=$?"
  for PMC in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc_${CFG}_$PMC" -- \
      python3 "$REPO/bench.py" --config $CFG --no-cpu-baseline --steps 2 --warmup 0 > "$OUT/pmc_${CFG}_$PMC.log" 2>&1
    echo "pmc $CFG $PMC rc=$?"
  done
done
# keep only the small summaries (the traces are large)
cd "$OUT"
find . -name "*kernel_trace.csv" -size +8M -delete
find . -name "*counter_collection.csv" -size +8M -delete
du -sh .
