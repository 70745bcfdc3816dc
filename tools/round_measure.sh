#!/bin/bash
# GPU box: the round's measurement set. usage: tools/round_measure.sh OUTDIR
#  1. PMC passes of `bench.py --config C --steps 3 --warmup 0` per config
#     (tools/pmc_profile.sh: separate rocprofv3 --pmc runs) and from them
#     OUTDIR/counters.json (tools/pmc_rooflines.py) - copy it to
#     profiles/rNN/counters.json, bench.py's roofline reads it from there
#  2. the bench.py line of every single-GPU config
#  4. the three .param files through cmi-gpu (OUTDIR/cmi_gpu_param_files.txt)
#  3. rocprofv3 --kernel-trace --stats of the same bench command, summarised
#     (tools/profile_summary.py) into OUTDIR/bench_<config>_kernel_stats.txt
set -u
REPO=$(pwd)
OUT=$REPO/$1
ROUND=${2:-r06}
mkdir -p "$OUT"
export TMPDIR=/tmp
# (the GPU box has no .git: pass the commit the counters belong to, e.g.
#  CMI_PROFILE_COMMIT=$(git rev-parse --short HEAD) in the gpurun command)
# (SKIP_PMC=1: keep profiles/rNN/counters.json as it is)
if [ -z "${SKIP_PMC:-}" ]; then
  for CFG in stromgren stromgren_diffuse lexington; do
    tools/pmc_profile.sh "$1/pmc_$CFG" -- bench.py --config $CFG --steps 3 --warmup 0 --no-cpu-baseline --no-also
  done
  python3 tools/pmc_rooflines.py "$OUT" 3 > "$OUT/counters.json" || exit 1
  mkdir -p "$REPO/profiles/$ROUND"
  cp "$OUT/counters.json" "$REPO/profiles/$ROUND/counters.json"
fi
for CFG in stromgren stromgren_diffuse lexington; do
  python3 bench.py --config $CFG --no-also > "$OUT/bench_$CFG.json" 2> "$OUT/bench_$CFG.err"
  echo "bench $CFG rc=$?"
done
# the round driver's own command: the headline line with the other two
# single-GPU configs under "also"
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench default rc=$?"
cd /tmp
for CFG in stromgren stromgren_diffuse lexington; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$CFG" -- \
    python3 "$REPO/bench.py" --config $CFG --no-cpu-baseline --no-also > "$OUT/stats_$CFG.log" 2>&1
  echo "stats $CFG rc=$?"
  python3 "$REPO/tools/profile_summary.py" "$OUT/stats_$CFG" "$OUT/stats_$CFG.log" \
    > "$OUT/bench_${CFG}_kernel_stats.txt"
done
# 4. the reference's three .param files, unchanged, through the cmi-gpu
#    executable (Gadget / HDF5 snapshots, written and deleted)
mkdir -p "$OUT/params" && cd "$OUT/params" && cp "$REPO/benchmarks/lexingtonHII40.yml" .
{
  for F in stromgren stromgren_diffuse lexingtonHII40; do
    echo "== cmi-gpu --params benchmarks/$F.param --output-statistics"
    "$REPO/cmacionize_amd/cmi-gpu" --params "$REPO/benchmarks/$F.param" --output-statistics 2>&1 |
      grep -E "Escape fraction|scattered|non-ionizing|Maximum number|Total photon|Total cell|Done shooting" | tail -9
    ls -l *.hdf5 | awk '{print $5, $9}'
    rm -f *.hdf5
  done
} > "$OUT/cmi_gpu_param_files.txt" 2>&1
cd "$OUT" && rm -rf params
# keep only the small summaries (the traces are large)
cd "$OUT"
find . -name "*kernel_trace.csv" -delete
find . -name "*counter_collection.csv" -size +2M -delete
du -sh .
